// host_io_check.cpp — reads BAM files with the product's reader and prints what it got; built with AddressSanitizer and
// UndefinedBehaviorSanitizer by tools/asan_host_io.sh (CPU only: no GPU sanitizers on this pool).
#include "../bamqc_amd/host/bam_io.h"
#include <cstdio>
int main(int argc, char** argv)
{
    for (int a = 1; a < argc; ++a) {
        BamReader rd;
        std::string err;
        if (!rd.open(argv[a], err)) { printf("%s: open failed: %s\n", argv[a], err.c_str()); continue; }
        rd.set_main_chrom(std::vector<uint8_t>(rd.header().ref_names.size(), 1));
        HostBatch hb; size_t n = 0; int code = 0, rc;
        while ((rc = rd.next_batch(hb, 100000, 1 << 26, err, code)) > 0) n += hb.n();
        printf("%s: %zu records rc %d %s\n", argv[a], n, rc, rc < 0 ? err.c_str() : "");
    }
}
