// driver.cpp — the `bamqualcheck` program as a library function: drop-in for the reference's main()
// (src/bamqualcheck.cpp:239-457) and command line (src/CommandLineParser.hpp:43-149), with the
// record loop's body replaced by the GPU aggregation of include/bamqc.h.  Also: FASTA loader
// (replaces Genome / SequenceStream, src/TripletCounting.hpp:60-104) and the C wrappers of
// include/bamqc_host.h around the BAM reader.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <chrono>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/bamqc_host.h"
#include "bam_io.h"
#include "raw_vector.h"
#include "parallel.h"
#include "crc32_fast.h"
#include "inflate_fast.h"
#include "host_tools.h"
#include "../csrc/gpu_bam.h"
#include <hip/hip_runtime_api.h>

// ---------------------------------------------------------------------------------------------------
// FASTA
// ---------------------------------------------------------------------------------------------------
struct FastaRecord { std::string name; raw_vector<uint8_t> codes; };

static inline uint8_t dna5_of_char(char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 4;
    }
}

// An uncompressed FASTA file, mapped: where its records are (header lines found by all host threads: '>' outside a header line; a
// few thousand at most), and any record's sequence as Dna5 codes on demand — counted (bytes other than '\n' / '\r') and encoded
// chunk by chunk straight into the record's array by all host threads (a human genome is 3 GB of text: 2.2 s on one thread).
// The id is cut at the first space or tab (TripletCounting.hpp:99-102).  Same rules as the sequential loader below.
struct MappedFasta {
    struct Rec { std::string name; size_t lo, hi; }; // [lo, hi): the record's sequence lines in the file
    std::vector<Rec> recs;
    const char* m = nullptr;
    size_t size = 0;
    ~MappedFasta() { if (m) munmap((void*)m, size); }
    MappedFasta() = default;
    MappedFasta(const MappedFasta&) = delete;
    MappedFasta& operator=(const MappedFasta&) = delete;
    // 1: indexed, 0: not this kind of file (compressed, a pipe, empty)
    int open(const char* path)
    {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return 0;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 2) { close(fd); return 0; }
        size = (size_t)st.st_size;
        void* mp = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (mp == MAP_FAILED) { size = 0; return 0; }
        m = (const char*)mp;
        if ((uint8_t)m[0] == 0x1f && (uint8_t)m[1] == 0x8b) return 0; // gzip: the sequential loader inflates it
        (void)madvise(mp, size, MADV_WILLNEED);
        const unsigned nt = bqc_host_threads();
        // 1. every '>' of the file
        const size_t n_chunks = std::max<size_t>(1, std::min<size_t>((size + (1u << 20) - 1) >> 20, (size_t)nt * 4));
        std::vector<std::vector<size_t>> gts(n_chunks);
        parallel_ranges(n_chunks, nt, 1, [&](unsigned, size_t lo, size_t hi) {
            for (size_t c = lo; c < hi; ++c) {
                const char* p = m + size * c / n_chunks;
                const char* const e = m + size * (c + 1) / n_chunks;
                while (p < e) {
                    const char* g = (const char*)memchr(p, '>', (size_t)(e - p));
                    if (!g) break;
                    gts[c].push_back((size_t)(g - m));
                    p = g + 1;
                }
            }
        });
        // 2. those outside a header line start one; the line ends at the next '\n' (or with the file)
        struct Hdr { size_t gt, nl; };
        std::vector<Hdr> hdrs;
        size_t limit = 0;
        for (const auto& v : gts)
            for (size_t g : v) {
                if (g < limit) continue;
                const char* nl = (const char*)memchr(m + g, '\n', size - g);
                const size_t e = nl ? (size_t)(nl - m) : size;
                hdrs.push_back(Hdr{g, e});
                limit = e + 1;
            }
        recs.reserve(hdrs.size());
        for (size_t i = 0; i < hdrs.size(); ++i) {
            std::string hdr(m + hdrs[i].gt + 1, hdrs[i].nl - hdrs[i].gt - 1);
            const size_t cut = hdr.find_first_of(" \t");
            std::string name = hdr.substr(0, cut);
            if (!name.empty() && name.back() == '\r') name.pop_back();
            recs.push_back(Rec{name, std::min(size, hdrs[i].nl + 1), i + 1 < hdrs.size() ? hdrs[i + 1].gt : size});
        }
        return 1;
    }
    // the sequences of the records `which` (indices into recs), encoded together by all host threads; false: out of memory
    bool encode(const std::vector<size_t>& which, std::vector<raw_vector<uint8_t>*> const& out) const
    {
        const unsigned nt = bqc_host_threads();
        struct Task { size_t k, lo, hi, count, at; };
        std::vector<Task> tasks;
        const size_t kTask = 4u << 20;
        for (size_t k = 0; k < which.size(); ++k)
            for (size_t a = recs[which[k]].lo; a < recs[which[k]].hi; a += kTask) tasks.push_back(Task{k, a, std::min(recs[which[k]].hi, a + kTask), 0, 0});
        parallel_ranges(tasks.size(), nt, 1, [&](unsigned, size_t lo, size_t hi) {
            for (size_t t = lo; t < hi; ++t) {
                size_t n = 0;
                for (const char* q = m + tasks[t].lo; q < m + tasks[t].hi; ++q) n += (*q != '\n') & (*q != '\r');
                tasks[t].count = n;
            }
        });
        std::vector<size_t> total(which.size(), 0);
        for (auto& t : tasks) { t.at = total[t.k]; total[t.k] += t.count; }
        try {
            for (size_t k = 0; k < which.size(); ++k) { out[k]->clear(); if (total[k]) { out[k]->resize(total[k]); advise_huge(*out[k]); } }
        } catch (const std::bad_alloc&) { return false; }
        uint8_t lut[256];
        for (int i = 0; i < 256; ++i) lut[i] = dna5_of_char((char)i);
        parallel_ranges(tasks.size(), nt, 1, [&](unsigned, size_t lo, size_t hi) {
            for (size_t t = lo; t < hi; ++t) {
                uint8_t* w = out[tasks[t].k]->data() + tasks[t].at;
                const char* q = m + tasks[t].lo;
                const char* const hi_q = m + tasks[t].hi;
                while (q < hi_q) { // line by line: a run without '\n' is translated as a whole; a '\r' inside it (rare) takes the byte-wise path
                    const char* nl = (const char*)memchr(q, '\n', (size_t)(hi_q - q));
                    const char* const e = nl ? nl : hi_q;
                    const size_t n = (size_t)(e - q);
                    if (n && memchr(q, '\r', n)) {
                        for (size_t i = 0; i < n; ++i) if (q[i] != '\r') *w++ = lut[(uint8_t)q[i]];
                    } else {
                        for (size_t i = 0; i < n; ++i) w[i] = lut[(uint8_t)q[i]];
                        w += n;
                    }
                    q = e + 1;
                }
            }
        });
        return true;
    }
};

// Loads every record (`want`, optional, limits which sequences are kept in memory; all records keep their FASTA position) of a
// mapped file.  1: loaded, 0: not this kind of file (compressed, a pipe, empty), -1: error.
static int load_fasta_mapped(const char* path, const std::vector<std::string>* want, std::vector<FastaRecord>& out)
{
    MappedFasta mf;
    if (mf.open(path) != 1) return 0;
    out.clear();
    out.reserve(mf.recs.size());
    std::vector<size_t> which;
    for (size_t i = 0; i < mf.recs.size(); ++i) {
        bool keep = true;
        if (want) {
            keep = false;
            for (const auto& w : *want) if (w == mf.recs[i].name) { keep = true; break; }
        }
        out.push_back(FastaRecord{mf.recs[i].name, {}});
        if (keep) which.push_back(i);
    }
    std::vector<raw_vector<uint8_t>*> dst;
    for (size_t i : which) dst.push_back(&out[i].codes);
    return mf.encode(which, dst) ? 1 : -1;
}

static bool load_fasta(const char* path, const std::vector<std::string>* want, std::vector<FastaRecord>& out, std::string& err)
{
    if (!getenv("BQC_FASTA_SEQUENTIAL")) {
        const int rc = load_fasta_mapped(path, want, out);
        if (rc > 0) return true;
        if (rc < 0) { err = std::string("ERROR: out of memory while loading ") + path; return false; }
        out.clear();
    }
    gzFile f = gzopen(path, "rb");
    if (!f) { err = std::string("ERROR: Could not open fasta file ") + path; return false; }
    gzbuffer(f, 1 << 20);
    std::vector<char> buf(1 << 22);
    // An uncompressed file bounds every contig by the bytes that are left: reserving that much (address space only) saves
    // growing a multi-hundred-megabyte vector through a dozen reallocations, each faulting in fresh pages.
    uint64_t file_size = 0, consumed = 0, reserve_hint = 0; // (consumed: bytes of the buffers before the current one)
    {
        struct stat st;
        if (gzdirect(f) && stat(path, &st) == 0 && st.st_size > 0) file_size = (uint64_t)st.st_size;
    }
    FastaRecord* cur = nullptr;
    bool keep = false, in_header = false;
    std::string hdr;
    uint8_t lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = dna5_of_char((char)i);
    auto finish_header = [&]() {
        size_t cut = hdr.find_first_of(" \t");
        std::string name = hdr.substr(0, cut);
        if (!name.empty() && name.back() == '\r') name.pop_back();
        out.push_back(FastaRecord{name, {}});
        cur = &out.back();
        keep = true;
        if (want) {
            keep = false;
            for (const auto& w : *want) if (w == name) { keep = true; break; }
        }
        if (keep && reserve_hint) { try { cur->codes.reserve((size_t)reserve_hint); } catch (const std::bad_alloc&) {} }
        hdr.clear();
    };
    int n;
    while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) {
        // Same rules as a byte-at-a-time scan ('>' anywhere outside a header line starts one; '\n' and '\r' are dropped from
        // sequence), but header lines and runs of sequence are handled in bulk: a whole genome is a few GB of text.
        const char* p = buf.data();
        const char* const end = p + n;
        while (p < end) {
            if (in_header) {
                const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
                if (!nl) { hdr.append(p, (size_t)(end - p)); break; }
                hdr.append(p, (size_t)(nl - p));
                in_header = false;
                { const uint64_t at = consumed + (uint64_t)(nl - buf.data()); reserve_hint = file_size > at ? file_size - at : 0; }
                finish_header();
                p = nl + 1;
                continue;
            }
            const char* gt = (const char*)memchr(p, '>', (size_t)(end - p));
            const char* stop = gt ? gt : end;
            if (cur && keep && stop > p) {
                raw_vector<uint8_t>& codes = cur->codes;
                const size_t at = codes.size();
                codes.resize(at + (size_t)(stop - p));
                uint8_t* w = codes.data() + at;
                for (const char* q = p; q < stop; ++q) {
                    const char c = *q;
                    *w = lut[(uint8_t)c];
                    w += (c != '\n') & (c != '\r');
                }
                codes.resize((size_t)(w - codes.data()));
            }
            p = stop;
            if (gt) { in_header = true; ++p; }
        }
        consumed += (uint64_t)n;
    }
    if (in_header) finish_header();
    gzclose(f);
    return true;
}

// ---------------------------------------------------------------------------------------------------
// C wrappers around the reader (python drivers, tests)
// ---------------------------------------------------------------------------------------------------
struct bqc_bam {
    BamReader bam;
    SamReader sam;
    bool is_sam = false;
    FILE* sam_file = nullptr;
    std::unique_ptr<GpuBamReader> gpu; // bqc_bam_open_gpu: records decoded on the card
    uint64_t g_b0 = 0, g_b1 = UINT64_MAX; // bqc_bam_open_gpu_range: the shard's block boundaries
    RecordReader& rd() { return gpu ? static_cast<RecordReader&>(*gpu) : is_sam ? static_cast<RecordReader&>(sam) : static_cast<RecordReader&>(bam); }
    const BamHeader& hdr() const { return const_cast<bqc_bam*>(this)->rd().header(); }
    ~bqc_bam() { if (sam_file && sam_file != stdin) fclose(sam_file); }
    HostBatch hb;
    bqc_batch view;
    std::string err;
    std::vector<std::string> lane_sorted;
    std::vector<uint32_t> lane_idx;
    void refresh_lanes()
    {
        lane_sorted.clear(); lane_idx.clear();
        for (auto& kv : rd().header().lane_names) { lane_sorted.push_back(kv.first); lane_idx.push_back(kv.second); }
    }
};

extern "C" int bqc_bam_open(const char* path, bqc_bam** out)
{
    if (!path || !out) return BQC_ERR_ARG;
    auto* b = new bqc_bam();
    const size_t n = strlen(path);
    if (strcmp(path, "-") == 0 || (n > 4 && strcmp(path + n - 4, ".sam") == 0)) { // SAM text: stdin or a file
        b->is_sam = true;
        b->sam_file = strcmp(path, "-") == 0 ? stdin : fopen(path, "r");
        if (!b->sam_file) { b->err = std::string("could not open ") + path; *out = b; return BQC_ERR_IO; }
        if (!b->sam.open(b->sam_file, b->err)) { *out = b; return BQC_ERR_IO; }
    } else if (!b->bam.open(path, b->err)) { *out = b; return BQC_ERR_IO; }
    b->refresh_lanes();
    *out = b;
    return 0;
}
extern "C" int bqc_bam_open_gpu(const char* path, int device, bqc_bam** out)
{
    if (!path || !out) return BQC_ERR_ARG;
    auto* b = new bqc_bam();
    *out = b;
    if (!b->bam.open(path, b->err, true)) return BQC_ERR_IO; // the header, on the host
    b->gpu.reset(new GpuBamReader());
    const uint64_t first_record = b->bam.stream_pos();
    b->bam.close(); // (the host reader's read-ahead stops here)
    if (!b->gpu->open(path, device, b->bam.header(), first_record, 1u << 20, 256u << 20, b->err)) { b->gpu.reset(); return BQC_ERR_DEVICE; }
    b->gpu->allow_kernels();
    b->refresh_lanes();
    return 0;
}
extern "C" int bqc_bam_open_range(const char* path, uint64_t begin_hint, uint64_t end_hint, bqc_bam** out)
{
    if (!path || !out) return BQC_ERR_ARG;
    auto* b = new bqc_bam();
    *out = b;
    if (!b->bam.open_range(path, begin_hint, end_hint, b->err)) return BQC_ERR_IO;
    b->refresh_lanes();
    return 0;
}
// the block boundaries of a shard, as BamReader::open_range finds them; false: an empty shard (or an error: err set)
static bool shard_blocks(const char* path, uint64_t begin_hint, uint64_t end_hint, uint64_t& b0, uint64_t& b1, std::string& err)
{
    const uint64_t size = bgzf_file_size(path);
    b1 = end_hint >= size ? UINT64_MAX : bgzf_find_block(path, end_hint, err);
    if (b1 >= size) b1 = UINT64_MAX;
    b0 = begin_hint == 0 ? 0 : bgzf_find_block(path, begin_hint, err);
    return err.empty() && b0 < std::min(b1, size);
}
extern "C" int bqc_bam_open_gpu_range(const char* path, int device, uint64_t begin_hint, uint64_t end_hint, bqc_bam** out)
{
    if (!path || !out) return BQC_ERR_ARG;
    auto* b = new bqc_bam();
    *out = b;
    if (!shard_blocks(path, begin_hint, end_hint, b->g_b0, b->g_b1, b->err)) { if (b->err.empty()) b->err = "empty shard"; return BQC_ERR_ARG; }
    if (!b->bam.open(path, b->err, true)) return BQC_ERR_IO; // the header, on the host
    b->gpu.reset(new GpuBamReader());
    const uint64_t first_record = b->bam.stream_pos();
    b->bam.close();
    b->gpu->set_range(b->g_b0, b->g_b1);
    if (!b->gpu->open(path, device, b->bam.header(), first_record, 1u << 20, 256u << 20, b->err)) { b->gpu.reset(); return BQC_ERR_DEVICE; }
    b->gpu->allow_kernels();
    b->refresh_lanes();
    return 0;
}
extern "C" uint64_t bqc_bam_range_begin_block(const bqc_bam* b) { return b->gpu ? b->g_b0 : b->bam.range_begin_block(); }
extern "C" uint64_t bqc_bam_range_end_block(const bqc_bam* b) { return b->gpu ? b->g_b1 : b->bam.range_end_block(); }
extern "C" uint64_t bqc_bam_range_first(const bqc_bam* b) { return b->gpu ? b->gpu->range_first() : b->bam.range_first(); }
extern "C" uint64_t bqc_bam_range_over(const bqc_bam* b) { return b->gpu ? b->gpu->range_over() : b->bam.range_over(); }
extern "C" uint64_t bqc_bam_batches_handed_over(const bqc_bam* b) { return b && b->gpu ? b->gpu->batches_handed_over() : 0; }
extern "C" uint64_t bqc_file_size(const char* path) { return path ? bgzf_file_size(path) : 0; }
extern "C" void bqc_gpu_inflate_device(int device) { bgzf_gpu_inflate_device(device); }
extern "C" uint64_t bqc_gpu_inflated_blocks(void) { return bgzf_gpu_inflated_blocks(); }
extern "C" int bqc_inflate_raw(const uint8_t* in, uint64_t in_n, uint8_t* out, uint64_t out_n)
{
    static thread_local Inflater inf;
    return inf.run(in, (size_t)in_n, out, (size_t)out_n) ? 1 : 0;
}
extern "C" uint32_t bqc_crc32(const uint8_t* p, uint64_t n) { return bqc_crc32_fast(p, (size_t)n); }
extern "C" void bqc_bam_close(bqc_bam* b) { delete b; }
extern "C" const char* bqc_bam_error(const bqc_bam* b) { return b ? b->err.c_str() : ""; }
extern "C" uint32_t bqc_bam_n_refs(const bqc_bam* b) { return (uint32_t)b->hdr().ref_names.size(); }
extern "C" const char* bqc_bam_ref_name(const bqc_bam* b, uint32_t i) { return b->hdr().ref_names[i].c_str(); }
extern "C" uint32_t bqc_bam_ref_len(const bqc_bam* b, uint32_t i) { return b->hdr().ref_lens[i]; }
extern "C" const char* bqc_bam_sample_id(const bqc_bam* b) { return b->hdr().sample_id.c_str(); }
extern "C" uint32_t bqc_bam_lane_count(const bqc_bam* b) { return b->hdr().lane_count; }
extern "C" uint32_t bqc_bam_n_lane_names(bqc_bam* b) { b->refresh_lanes(); return (uint32_t)b->lane_sorted.size(); }
extern "C" const char* bqc_bam_lane_name(const bqc_bam* b, uint32_t i) { return b->lane_sorted[i].c_str(); }
extern "C" uint32_t bqc_bam_lane_index(const bqc_bam* b, uint32_t i) { return b->lane_idx[i]; }
extern "C" int bqc_bam_set_main_chrom(bqc_bam* b, const uint8_t* mc)
{
    if (!b || !mc) return BQC_ERR_ARG;
    b->rd().set_main_chrom(std::vector<uint8_t>(mc, mc + b->hdr().ref_names.size()));
    return 0;
}
extern "C" int bqc_bam_set_rid_filter(bqc_bam* b, const uint8_t* keep, int keep_unplaced)
{
    if (!b || !keep) return BQC_ERR_ARG;
    if (b->is_sam) return BQC_ERR_ARG; // (chromosome sharding needs the BAM reader)
    b->bam.set_rid_filter(std::vector<uint8_t>(keep, keep + b->hdr().ref_names.size()), keep_unplaced != 0);
    return 0;
}
extern "C" int bqc_bam_next(bqc_bam* b, uint32_t max_reads, uint64_t max_bases, const bqc_batch** out)
{
    if (!b || !out) return -BQC_ERR_ARG;
    int code = 0;
    const int rc = b->rd().next_batch(b->hb, max_reads, max_bases, b->err, code);
    if (rc < 0) return code < 0 ? code : -code; // (GpuBamReader::kUnsupported is negative already)
    if (b->gpu && b->hb.d_seq) { // callers of this wrapper read the columns on the host: fetch the payload
        HostBatch& h = b->hb;
        uint64_t so = 0, qo = 0, co = 0;
        for (size_t i = 0; i < h.n(); ++i) { so += (h.l_seq[i] + 1u) / 2u; qo += h.l_seq[i]; co += h.n_cigar[i]; }
        h.seq.resize(so); h.qual.resize(qo); h.cigar.resize(co);
        if ((so && hipMemcpy(h.seq.data(), h.d_seq, so, hipMemcpyDeviceToHost) != hipSuccess) || (qo && hipMemcpy(h.qual.data(), h.d_qual, qo, hipMemcpyDeviceToHost) != hipSuccess) ||
            (co && hipMemcpy(h.cigar.data(), h.d_cigar, 4 * co, hipMemcpyDeviceToHost) != hipSuccess)) { b->err = "copy from the device failed"; return -BQC_ERR_DEVICE; }
        h.d_seq = h.d_qual = nullptr; h.d_cigar = nullptr;
    }
    b->view = b->hb.view();
    *out = &b->view;
    return rc;
}

extern "C" int bqc_fasta_load(const char* path, uint32_t* n_records, char*** names, uint8_t*** codes, uint64_t** lens)
{
    std::vector<FastaRecord> recs;
    std::string err;
    if (!load_fasta(path, nullptr, recs, err)) return BQC_ERR_IO;
    *n_records = (uint32_t)recs.size();
    *names = (char**)calloc(recs.size() + 1, sizeof(char*));
    *codes = (uint8_t**)calloc(recs.size() + 1, sizeof(uint8_t*));
    *lens = (uint64_t*)calloc(recs.size() + 1, sizeof(uint64_t));
    for (size_t i = 0; i < recs.size(); ++i) {
        (*names)[i] = strdup(recs[i].name.c_str());
        (*codes)[i] = (uint8_t*)malloc(recs[i].codes.size() + 1);
        memcpy((*codes)[i], recs[i].codes.data(), recs[i].codes.size());
        (*lens)[i] = recs[i].codes.size();
    }
    return 0;
}
extern "C" void bqc_fasta_free(uint32_t n, char** names, uint8_t** codes, uint64_t* lens)
{
    for (uint32_t i = 0; i < n; ++i) { free(names[i]); free(codes[i]); }
    free(names); free(codes); free(lens);
}

// ---------------------------------------------------------------------------------------------------
// synthetic BAM + FASTA files
// ---------------------------------------------------------------------------------------------------
extern "C" int bqc_synth_write(const bqc_synth_params* p, const char* const* ref_names, const char* bam_path, const char* fasta_path,
                               uint32_t batch_reads)
{
    if (!p || !bam_path || !ref_names) return BQC_ERR_ARG;
    std::vector<std::vector<uint8_t>> refs(p->n_refs);
    std::vector<const uint8_t*> rp(p->n_refs);
    std::vector<std::string> names;
    std::vector<uint32_t> lens;
    for (uint32_t c = 0; c < p->n_refs; ++c) {
        refs[c].resize(p->ref_len[c]);
        bqc_synth_reference(p->seed, (int32_t)c, p->ref_len[c], refs[c].data());
        rp[c] = refs[c].data();
        names.push_back(ref_names[c]);
        lens.push_back(p->ref_len[c]);
    }
    if (fasta_path) {
        FILE* f = fopen(fasta_path, "wb");
        if (!f) return BQC_ERR_IO;
        std::string line;
        for (uint32_t c = 0; c < p->n_refs; ++c) {
            fprintf(f, ">%s synthetic contig %u\n", ref_names[c], c);
            for (uint64_t i = 0; i < refs[c].size(); i += 60) {
                line.clear();
                for (uint64_t k = i; k < std::min<uint64_t>(refs[c].size(), i + 60); ++k) line.push_back("ACGTN"[refs[c][k]]);
                line.push_back('\n');
                fwrite(line.data(), 1, line.size(), f);
            }
        }
        fclose(f);
    }
    std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
    for (uint32_t c = 0; c < p->n_refs; ++c) text += "@SQ\tSN:" + names[c] + "\tLN:" + std::to_string(lens[c]) + "\n";
    std::vector<std::string> lane_ids;
    for (uint32_t l = 0; l < std::max(1u, p->n_lanes); ++l) {
        lane_ids.push_back("L" + std::to_string(l + 1));
        text += "@RG\tID:" + lane_ids.back() + "\tSM:SYN\tPL:ILLUMINA\n";
    }
    BamWriter w;
    std::string err;
    if (!w.open(bam_path, text, names, lens, err)) return BQC_ERR_IO;
    // One generator call for all reads keeps the coordinate order; written in slices.
    bqc_batch* b = nullptr;
    int rc = bqc_synth_batch(p, rp.data(), &b);
    if (rc) return rc;
    (void)batch_reads;
    const bool ok = w.write_batch(*b, lane_ids, p->first_read_index) && w.close();
    bqc_synth_batch_free(b);
    return ok ? 0 : BQC_ERR_IO;
}

static std::string synth_header_text(const std::vector<std::string>& names, const std::vector<uint32_t>& lens, uint32_t n_lanes, std::vector<std::string>& lane_ids)
{
    std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
    for (size_t c = 0; c < names.size(); ++c) text += "@SQ\tSN:" + names[c] + "\tLN:" + std::to_string(lens[c]) + "\n";
    for (uint32_t l = 0; l < std::max(1u, n_lanes); ++l) {
        lane_ids.push_back("L" + std::to_string(l + 1));
        text += "@RG\tID:" + lane_ids.back() + "\tSM:SYN\tPL:ILLUMINA\n";
    }
    return text;
}

extern "C" int bqc_bam_write(const char* path, const bqc_batch* b, uint32_t n_refs, const char* const* ref_names, const uint32_t* ref_lens,
                             uint32_t n_lanes, uint64_t first_read_index, int level)
{
    if (!path || !b || (n_refs && (!ref_names || !ref_lens))) return BQC_ERR_ARG;
    std::vector<std::string> names, lane_ids;
    std::vector<uint32_t> lens;
    for (uint32_t c = 0; c < n_refs; ++c) { names.push_back(ref_names[c]); lens.push_back(ref_lens[c]); }
    const std::string text = synth_header_text(names, lens, n_lanes, lane_ids);
    for (uint32_t i = 0; i < b->n_reads; ++i) if (b->lane[i] >= lane_ids.size()) return BQC_ERR_ARG;
    BamWriter w;
    std::string err;
    if (!w.open(path, text, names, lens, err, level)) return BQC_ERR_IO;
    return w.write_batch_parallel(*b, lane_ids, first_read_index) && w.close() ? 0 : BQC_ERR_IO;
}

// The plan of bqc_synth_write, streamed: slices of the plan are generated (bqc_synth_slice), serialised into BAM records by
// all host threads (sizes -> prefix sum -> every thread writes its records in place) and handed to the BGZF writer.
extern "C" int bqc_synth_stream(const bqc_synth_params* p, const char* const* ref_names, const char* bam_path, const char* fasta_path,
                                uint32_t slice_reads, int level)
{
    if (!p || !bam_path || !ref_names || p->n_refs == 0) return BQC_ERR_ARG;
    if (!slice_reads) slice_reads = 1u << 21;
    std::vector<std::vector<uint8_t>> refs(p->n_refs);
    std::vector<const uint8_t*> rp(p->n_refs);
    std::vector<std::string> names;
    std::vector<uint32_t> lens;
    for (uint32_t c = 0; c < p->n_refs; ++c) {
        refs[c].resize(p->ref_len[c]);
        bqc_synth_reference(p->seed, (int32_t)c, p->ref_len[c], refs[c].data());
        rp[c] = refs[c].data();
        names.push_back(ref_names[c]);
        lens.push_back(p->ref_len[c]);
    }
    if (fasta_path) {
        FILE* f = fopen(fasta_path, "wb");
        if (!f) return BQC_ERR_IO;
        setvbuf(f, nullptr, _IOFBF, 1 << 22);
        std::vector<char> text;
        for (uint32_t c = 0; c < p->n_refs; ++c) {
            fprintf(f, ">%s synthetic contig %u\n", ref_names[c], c);
            const std::vector<uint8_t>& r = refs[c];
            const size_t n = r.size(), lines = (n + 59) / 60;
            text.resize(n + lines);
            parallel_ranges(lines, bqc_host_threads(), 1 << 14, [&](unsigned, size_t lo, size_t hi) {
                for (size_t l = lo; l < hi; ++l) {
                    const size_t a = l * 60, z = std::min(n, a + 60);
                    char* w = text.data() + a + l;
                    for (size_t k = a; k < z; ++k) *w++ = "ACGTN"[r[k]];
                    *w = '\n';
                }
            });
            if (fwrite(text.data(), 1, text.size(), f) != text.size()) { fclose(f); return BQC_ERR_IO; }
        }
        if (fclose(f) != 0) return BQC_ERR_IO;
    }
    std::vector<std::string> lane_ids;
    const std::string text = synth_header_text(names, lens, p->n_lanes, lane_ids);
    BamWriter w;
    std::string err;
    if (!w.open(bam_path, text, names, lens, err, level)) return BQC_ERR_IO;
    const bool timing = getenv("BQC_TIMING") != nullptr;
    double t_gen = 0, t_write = 0;
    for (uint64_t lo = 0; lo < p->n_reads; lo += slice_reads) {
        const uint32_t cnt = (uint32_t)std::min<uint64_t>(slice_reads, p->n_reads - lo);
        bqc_batch* b = nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = bqc_synth_slice(p, lo, cnt, rp.data(), &b);
        if (rc) return rc;
        const auto t1 = std::chrono::steady_clock::now();
        const bool ok = w.write_batch_parallel(*b, lane_ids, p->first_read_index + lo);
        bqc_synth_batch_free(b);
        const auto t2 = std::chrono::steady_clock::now();
        t_gen += std::chrono::duration<double>(t1 - t0).count();
        t_write += std::chrono::duration<double>(t2 - t1).count();
        if (!ok) return BQC_ERR_IO;
    }
    if (timing) fprintf(stderr, "[timing] synthetic stream of %u reads: generate %.2f s, serialise + compress + write %.2f s\n", p->n_reads, t_gen, t_write);
    return w.close() ? 0 : BQC_ERR_IO;
}

// ---------------------------------------------------------------------------------------------------
// command line (CommandLineParser.hpp:43-149)
// ---------------------------------------------------------------------------------------------------
namespace {
struct ProgramOptions {
    std::string bamFile, referenceFile = "genome.fa", outputFile;
    std::string chroms = "chr1,chr2,chr3,chr4,chr5,chr6,chr7,chr8,chr9,chr10,chr11,chr12,chr13,chr14,chr15,chr16,chr17,chr18,chr19,chr20,chr21,chr22";
    std::vector<int32_t> klist;
    std::vector<uint32_t> q_cutoff;
    double e = 0.01;
    int seed = 1;
    int isize = 1000;
    // extensions (not in the reference)
    uint32_t max_read_len = 65536, hist_cap = 65536;
    int device = 0;
    uint32_t batch_reads = [] { const char* e = getenv("BQC_BATCH_READS"); return e && atoi(e) > 0 ? (uint32_t)atoi(e) : 1u << 20; }(); // (the variable: experiments; --batch-reads)
    bool no_sketch = false;
};

const char* kVersion = "dev"; // src/version.h:12

void usage(FILE* f)
{
    fprintf(f,
            "bamqualcheck - String Modifier\n==============================\n\nSYNOPSIS\n    bamqualcheck [OPTIONS] BAMFILE\n\n"
            "DESCRIPTION\n    Program for bam quality checks. The program can read from bamfile or stdin(sam format)\n\n"
            "    -h, --help\n          Displays this help message.\n    --version\n          Display version information\n\n"
            "  General options:\n    -r, --reference FILENAME\n          Reference genome filename. Default: genome.fa.\n"
            "    -i, --insert-size INT\n          Upper bound for the insert size in insert size histogram. Default: 1000.\n"
            "    -c, --chromosomes STRING\n          Comma separated list of the main chromosome names.\n"
            "    -o, --output-file OUT\n          Output filename.\n\n"
            "  Kmerstream options:\n    -k, --kmer-size STRING\n          Comma-separated list of k-mer sizes. Default: 32.\n"
            "    -q, --quality-cutoff STRING\n          Comma-separated list of PHRED quality thresholds. Default: 17.\n"
            "    -e, --error-rate DOUBLE\n          Error rate guaranteed. Default: 0.01.\n"
            "    -s, --seed INT\n          Seed value for the randomness. Default: 1.\n\n"
            "  MI355X options (not in the reference):\n    --device INT, --max-read-len INT, --hist-cap INT, --batch-reads INT, --no-sketch\n\n"
            "VERSION\n    bamqualcheck version: %s\n    Last update August 2019\n",
            kVersion);
}

// returns 0 = ok, 1 = parse error, 2 = help/version printed (exit 0)
int parse_args(int argc, const char** argv, ProgramOptions& o, std::string& err)
{
    std::string kstr = "32", qstr = "17";
    bool have_r = false, have_o = false;
    std::vector<std::string> pos;
    auto need = [&](int& i, const std::string& name, std::string& val, const char* inl) -> bool {
        if (inl) { val = inl; return true; }
        if (i + 1 >= argc) { err = "bamqualcheck: option requires an argument -- " + name; return false; }
        val = argv[++i];
        return true;
    };
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-") { pos.push_back(a); continue; }
        if (a == "-h" || a == "--help") { usage(stdout); return 2; }
        if (a == "--version") { printf("bamqualcheck version: %s\nLast update August 2019\n", kVersion); return 2; }
        const char* inl = nullptr;
        std::string key = a;
        if (a.rfind("--", 0) == 0) {
            size_t eq = a.find('=');
            if (eq != std::string::npos) { key = a.substr(0, eq); inl = argv[i] + eq + 1; }
        } else if (a.size() > 2 && a[0] == '-') { // -ovalue
            key = a.substr(0, 2);
            inl = argv[i] + 2;
        }
        std::string v;
        if (key == "-r" || key == "--reference") { if (!need(i, key, v, inl)) return 1; o.referenceFile = v; have_r = true; }
        else if (key == "-o" || key == "--output-file") { if (!need(i, key, v, inl)) return 1; o.outputFile = v; have_o = true; }
        else if (key == "-c" || key == "--chromosomes") { if (!need(i, key, v, inl)) return 1; o.chroms = v; }
        else if (key == "-i" || key == "--insert-size") {
            if (!need(i, key, v, inl)) return 1;
            char* end = nullptr;
            long x = strtol(v.c_str(), &end, 10);
            if (!end || *end || v.empty()) { err = "bamqualcheck: the given value '" + v + "' cannot be casted to integer"; return 1; }
            o.isize = (int)x;
        }
        else if (key == "-k" || key == "--kmer-size") { if (!need(i, key, v, inl)) return 1; kstr = v; }
        else if (key == "-q" || key == "--quality-cutoff") { if (!need(i, key, v, inl)) return 1; qstr = v; }
        else if (key == "-e" || key == "--error-rate") {
            if (!need(i, key, v, inl)) return 1;
            char* end = nullptr;
            o.e = strtod(v.c_str(), &end);
            if (!end || *end || v.empty()) { err = "bamqualcheck: the given value '" + v + "' cannot be casted to double"; return 1; }
            if (o.e < 0) { err = "bamqualcheck: the given value '" + v + "' is smaller than the minimum value of 0"; return 1; }
        }
        else if (key == "-s" || key == "--seed") { if (!need(i, key, v, inl)) return 1; o.seed = atoi(v.c_str()); }
        else if (key == "--device") { if (!need(i, key, v, inl)) return 1; o.device = atoi(v.c_str()); }
        else if (key == "--max-read-len") { if (!need(i, key, v, inl)) return 1; o.max_read_len = (uint32_t)strtoul(v.c_str(), nullptr, 10); }
        else if (key == "--hist-cap") { if (!need(i, key, v, inl)) return 1; o.hist_cap = (uint32_t)strtoul(v.c_str(), nullptr, 10); }
        else if (key == "--batch-reads") { if (!need(i, key, v, inl)) return 1; o.batch_reads = (uint32_t)strtoul(v.c_str(), nullptr, 10); }
        else if (key == "--no-sketch") { o.no_sketch = true; }
        else if (a[0] == '-') { err = "bamqualcheck: illegal option -- " + a.substr(a.rfind("--", 0) == 0 ? 2 : 1); return 1; }
        else pos.push_back(a);
    }
    if (!have_r) { err = "bamqualcheck: option requires an argument -- r (required option -r/--reference missing)"; return 1; }
    if (!have_o) { err = "bamqualcheck: option requires an argument -- o (required option -o/--output-file missing)"; return 1; }
    if (pos.size() != 1) { err = pos.empty() ? "bamqualcheck: Not enough arguments were provided." : "bamqualcheck: Too many arguments were provided!"; return 1; }
    o.bamFile = pos[0];
    if (o.bamFile != "-") { // valid values: "- bam sam" (file extension), CommandLineParser.hpp:59-60
        size_t dot = o.bamFile.rfind('.');
        std::string ext = dot == std::string::npos ? "" : o.bamFile.substr(dot + 1);
        if (ext != "bam" && ext != "sam") { err = "bamqualcheck: the given path '" + o.bamFile + "' does not have one of the valid file extensions [*.-, *.bam, *.sam]"; return 1; }
    }
    // comma separated lists, parsed like the reference's stringstream loops (:121-143)
    auto split_nums = [](const std::string& s, auto& out) {
        size_t p = 0;
        while (p < s.size()) {
            char* end = nullptr;
            long v = strtol(s.c_str() + p, &end, 10);
            if (end == s.c_str() + p) break;
            out.push_back((typename std::remove_reference<decltype(out)>::type::value_type)v);
            p = (size_t)(end - s.c_str());
            if (p < s.size() && s[p] == ',') ++p;
        }
    };
    split_nums(kstr, o.klist);
    split_nums(qstr, o.q_cutoff);
    return 0;
}

// Page-locked decode buffers: the columns of every HostBatch the decoder fills are registered with the HIP runtime once
// (hipHostRegister, ~40 ms per GB of touched pages), so that bqc_submit_async copies them to the device without a staging
// copy; a block that a growing vector frees is released first (bqc_raw_vector_free_hook).
struct PinRegistry {
    std::mutex m;
    std::unordered_map<void*, size_t> blocks;
    double t_register = 0;
    void pin(const void* p, size_t bytes)
    {
        if (!p || bytes < (1u << 16)) return;
        std::lock_guard<std::mutex> lk(m);
        auto it = blocks.find((void*)p);
        if (it != blocks.end() && it->second >= bytes) return;
        if (it != blocks.end()) { (void)bqc_host_unregister((void*)p); blocks.erase(it); }
        const auto t0 = std::chrono::steady_clock::now();
        if (bqc_host_register((void*)p, bytes) == 0) blocks[(void*)p] = bytes;
        t_register += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    void release(void* p)
    {
        std::lock_guard<std::mutex> lk(m);
        auto it = blocks.find(p);
        if (it == blocks.end()) return;
        (void)bqc_host_unregister(p);
        blocks.erase(it);
    }
    void release_all()
    {
        std::lock_guard<std::mutex> lk(m);
        for (auto& kv : blocks) (void)bqc_host_unregister(kv.first);
        blocks.clear();
    }
};
PinRegistry g_pins;
void pin_free_hook(void* p) { g_pins.release(p); }
void pin_hook(const void* p, size_t bytes) { g_pins.pin(p, bytes); }
template <typename V> void pin_column(const V& v) { g_pins.pin(v.data(), v.capacity() * sizeof(typename V::value_type)); }
void pin_batch(const HostBatch& hb)
{
    pin_column(hb.flag); pin_column(hb.n_cigar); pin_column(hb.mapq); pin_column(hb.lane); pin_column(hb.seq); pin_column(hb.qual);
    pin_column(hb.rid); pin_column(hb.pos); pin_column(hb.tlen); pin_column(hb.nm); pin_column(hb.as); pin_column(hb.l_seq); pin_column(hb.cigar);
}

struct BatchQueue { // decode thread -> submit thread
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::unique_ptr<HostBatch>> q;
    std::vector<std::unique_ptr<HostBatch>> spare; // submitted batches go back to the decoder: their ~300 MB of columns are reused
    bool done = false, stop = false;
    int err_code = 0;
    std::string err;
};
} // namespace

namespace {
struct ShardArgs { uint32_t index, count; bqc_shard_hook hook; void* user; };
}

static int run_program(int argc, const char** argv, const ShardArgs* shard, bool host_reader_only = false)
{
    ProgramOptions opt;
    std::string perr;
    const int pr = parse_args(argc, argv, opt, perr);
    if (pr == 2) return 0;
    if (pr == 1) { fprintf(stderr, "%s\n", perr.c_str()); return 1; }
    using clk = std::chrono::steady_clock;
    const bool timing = getenv("BQC_TIMING") && getenv("BQC_TIMING")[0] == '1'; // BQC_TIMING=1: where the wall time of a run goes (stderr)
    auto since_launch = [&](const char* what) { // (BQC_T0: the launcher's CLOCK_MONOTONIC seconds when it started the program)
        if (!timing || !getenv("BQC_T0")) return;
        const double now = std::chrono::duration<double>(clk::now().time_since_epoch()).count();
        fprintf(stderr, "[timing] %s: %.3f s after launch\n", what, now - atof(getenv("BQC_T0")));
    };
    since_launch("main entered");
    setenv("GPU_MAX_HW_QUEUES", "8", 0); // (before the runtime starts: the batch pipeline, the reader and its producer each keep a hardware queue of their own)
    // the HIP runtime starts (~0.1 s) while the inputs are opened; from then on the reader's runs are inflated on the card
    const char* gi_env = getenv("BQC_GPU_INFLATE");
    const bool gpu_inflate = gi_env && atoi(gi_env) != 0;
    std::thread warm([dev = opt.device, gpu_inflate] {
        if (bqc_warmup(dev) != 0 || !gpu_inflate) return;
        bqc_raw_vector_free_hook = pin_free_hook; // the reader page-locks the buffers the card copies into
        bqc_raw_vector_pin_hook = pin_hook;
        bgzf_gpu_inflate_device(dev);
    });
    struct InflateOff { ~InflateOff() { bgzf_gpu_inflate_device(-1); bqc_raw_vector_pin_hook = nullptr; } } inflate_off; // (destroyed after the joiner below has joined the thread that sets it)
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } warm_joiner{warm};
    // "-": SAM text from stdin; everything else is opened as BAM (bamqualcheck.cpp:252-262: a .sam path fails to open there too)
    BamReader bam_rd;
    SamReader sam_rd;
    std::string err;
    const bool from_stdin = opt.bamFile == "-";
    // a failure before the hook is reached must still reach it: the other processes wait for this one
    auto shard_abort = [&]() { if (shard) { bqc_shard_info si{}; si.status = 1; bqc_shard_result sr{}; (void)shard->hook(shard->user, &si, &sr); } return 1; };
    if (shard && from_stdin) { fprintf(stderr, "ERROR: a stream on stdin cannot be split between processes\n"); return shard_abort(); }
    bool opened;
    // a shard of the file: its block boundaries are found here (BamReader::open_range finds the same ones: bgzf_find_block is a
    // function of the hint), so that either reader can take the range
    uint64_t shard_b0 = 0, shard_b1 = UINT64_MAX;
    bool shard_gpu = false;
    if (from_stdin) opened = sam_rd.open(stdin, err);
    else if (shard) { // this process's part of the compressed file
        const uint64_t size = bqc_file_size(opt.bamFile.c_str());
        const uint64_t lo = size / shard->count * shard->index, hi = shard->index + 1 == shard->count ? UINT64_MAX : size / shard->count * (shard->index + 1);
        const char* gd = getenv("BQC_GPU_DECODE");
        struct stat st;
        if (!host_reader_only && !(gd && atoi(gd) == 0) && stat(opt.bamFile.c_str(), &st) == 0 && S_ISREG(st.st_mode)) {
            std::string e2;
            // (an empty part, or a small one, is the host reader's: the GPU reader's set-up costs ~50 ms)
            shard_gpu = shard_blocks(opt.bamFile.c_str(), lo, hi, shard_b0, shard_b1, e2) && (gd || std::min(shard_b1, size) - shard_b0 >= (256ull << 20));
        }
        opened = shard_gpu ? bam_rd.open(opt.bamFile.c_str(), err, true) /* the header; the records come from the card */ : bam_rd.open_range(opt.bamFile.c_str(), lo, hi, err);
    } else {
        // (a file the GPU reader will probably take: the host reader is only asked for the header — a small first run)
        const char* gd = getenv("BQC_GPU_DECODE");
        struct stat st;
        const bool likely_gpu = !host_reader_only && !(gd && atoi(gd) == 0) && stat(opt.bamFile.c_str(), &st) == 0 && S_ISREG(st.st_mode) && (gd || (uint64_t)st.st_size >= (256ull << 20));
        opened = bam_rd.open(opt.bamFile.c_str(), err, likely_gpu);
    }
    since_launch("input opened");
    if (!opened) {
        fprintf(stderr, "ERROR: Could not open %s for reading.\n", opt.bamFile.c_str()); // bamqualcheck.cpp:265
        return shard_abort();
    }
    // A BAM file read as a whole is inflated and decoded on the GPU (csrc/gpu_bam.hip): the host reader has read the header, the
    // records come from the card.  Whatever that reader does not decode itself (a read group that is not in the header, a record
    // the host reader would report, ...) ends this pass before anything has been reported, and the program starts over with the
    // host reader (host_reader_only).  BQC_GPU_DECODE=0: the host reader from the start.
    GpuBamReader gpu_rd;
    const char* gd_env = getenv("BQC_GPU_DECODE"); // 1: whenever possible, 0: never; unset: files of 256 MB and more (its set-up costs ~50 ms)
    bool use_gpu_reader = !from_stdin && (!shard || shard_gpu) && !host_reader_only && bam_rd.header().lane_count != 0 && !(gd_env && atoi(gd_env) == 0);
    if (shard && shard_gpu && !use_gpu_reader) { // (no @RG line: the host reader, over its range, decides what that means)
        bam_rd.~BamReader();
        new (&bam_rd) BamReader();
        const uint64_t size = bqc_file_size(opt.bamFile.c_str());
        const uint64_t lo = size / shard->count * shard->index, hi = shard->index + 1 == shard->count ? UINT64_MAX : size / shard->count * (shard->index + 1);
        if (!bam_rd.open_range(opt.bamFile.c_str(), lo, hi, err)) { fprintf(stderr, "ERROR: Could not open %s for reading.\n", opt.bamFile.c_str()); return shard_abort(); }
        shard_gpu = false;
    }
    if (use_gpu_reader && !shard) { // a regular file: the reader opens it a second time
        struct stat st;
        use_gpu_reader = stat(opt.bamFile.c_str(), &st) == 0 && S_ISREG(st.st_mode) && (gd_env || (uint64_t)st.st_size >= (256ull << 20));
    }
    uint64_t first_record_u = 0;
    if (use_gpu_reader) {
        first_record_u = bam_rd.stream_pos();
        bam_rd.close(); // (its read-ahead has inflated the first run of blocks for the header; no more)
        if (shard) gpu_rd.set_range(shard_b0, shard_b1);
        bqc_raw_vector_free_hook = pin_free_hook;
        bqc_raw_vector_pin_hook = pin_hook;
    }
    RecordReader& rd = from_stdin ? static_cast<RecordReader&>(sam_rd) : use_gpu_reader ? static_cast<RecordReader&>(gpu_rd) : static_cast<RecordReader&>(bam_rd);
    if (use_gpu_reader) gpu_rd.header() = bam_rd.header(); // (complete once opened, in the decode thread: the device is not up yet)
    if (!shard || shard->index == 0) {
        FILE* of = fopen(opt.outputFile.c_str(), "wb"); // opened (truncated) before the scan, :278-283
        if (!of) { fprintf(stderr, "ERROR: Could not open output file %s\n", opt.outputFile.c_str()); return shard_abort(); }
        fclose(of);
    }
    const BamHeader& H = rd.header();
    const uint32_t n_refs = (uint32_t)H.ref_names.size();
    // initChroms (:106-123): names that are not BAM references are silently dropped
    std::vector<uint8_t> main_chrom(std::max(1u, n_refs), 0);
    {
        size_t p = 0;
        while (p <= opt.chroms.size()) {
            size_t e = opt.chroms.find(',', p);
            if (e == std::string::npos) e = opt.chroms.size();
            const std::string name = opt.chroms.substr(p, e - p);
            for (uint32_t r = 0; r < n_refs; ++r) if (H.ref_names[r] == name) { main_chrom[r] = 1; break; }
            p = e + 1;
        }
    }
    rd.set_main_chrom(main_chrom);
    since_launch("input opened, header read");
    const auto t_begin = std::chrono::steady_clock::now(); // (BQC_TIMING=1 also reports the phases around the record loop)
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    double t_wait = 0, t_submit = 0, t_decode = 0;
    uint64_t n_total = 0;
    // decode thread feeds batches; this thread submits them.  It starts right away: the first batches are inflated and
    // decoded while the FASTA file is read and the device is set up.
    BatchQueue Q;
    std::atomic<bool> gpu_reader_opened{false};
    std::thread dec;
    auto stop_decoder = [&]() {
        if (!dec.joinable()) return;
        { std::lock_guard<std::mutex> lk(Q.m); Q.stop = true; }
        Q.cv.notify_all();
        dec.join();
    };
    if (H.lane_count != 0) dec = std::thread([&]() {
        if (use_gpu_reader) {
            std::string e;
            const bool ok = gpu_rd.open(opt.bamFile.c_str(), opt.device, gpu_rd.header(), first_record_u, opt.batch_reads, 256ull << 20, e);
            gpu_reader_opened = true; // (the context is created after this: see there)
            if (!ok) {
                std::lock_guard<std::mutex> lk(Q.m);
                Q.err = e; Q.err_code = GpuBamReader::kUnsupported; Q.done = true; Q.cv.notify_all();
                return;
            }
            gpu_rd.set_main_chrom(main_chrom);
        }
        for (;;) {
            std::unique_ptr<HostBatch> hb;
            {
                std::lock_guard<std::mutex> lk(Q.m);
                if (!Q.spare.empty()) { hb = std::move(Q.spare.back()); Q.spare.pop_back(); }
            }
            if (!hb) hb = std::make_unique<HostBatch>();
            int code = 0;
            std::string e;
            const auto d0 = clk::now();
            const int r = rd.next_batch(*hb, opt.batch_reads, 256ull << 20, e, code);
            t_decode += secs(d0, clk::now());
            std::unique_lock<std::mutex> lk(Q.m);
            if (r < 0) { Q.err = e; Q.err_code = code; Q.done = true; Q.cv.notify_all(); return; }
            if (r == 0) { Q.done = true; Q.cv.notify_all(); return; }
            Q.cv.wait(lk, [&] { return Q.q.size() < 3 || Q.stop; });
            if (Q.stop) return;
            Q.q.push_back(std::move(hb));
            Q.cv.notify_all();
        }
    });
    if (H.lane_count == 0) { // no @RG: counts is empty; any record is an error, none means an empty output file
        std::vector<FastaRecord> fa0;
        std::string ferr0;
        if (!load_fasta(opt.referenceFile.c_str(), &H.ref_names, fa0, ferr0)) fprintf(stderr, "%s\n", ferr0.c_str()); // return value ignored (:291)
        HostBatch hb;
        int code = 0;
        const int rc = rd.next_batch(hb, 1, 1 << 20, err, code);
        if (rc != 0) { fprintf(stderr, "%s\n", rc < 0 ? err.c_str() : "ERROR: records present but the header has no @RG line"); return shard_abort(); }
        if (shard) { bqc_shard_info si{}; bqc_shard_result sr{}; (void)shard->hook(shard->user, &si, &sr); } // (nothing to merge: no counters at all)
        return 0;
    }
    // the context is created (device memory, streams, sketch tables: ~0.1 s) by a thread of its own while this one reads the FASTA file
    bqc_options bo;
    memset(&bo, 0, sizeof bo);
    bo.struct_size = sizeof bo;
    bo.n_lanes = H.lane_count; bo.n_refs = n_refs; bo.isize = opt.isize;
    bo.max_read_len = opt.max_read_len; bo.hist_cap = opt.hist_cap;
    bo.main_chrom = main_chrom.data(); bo.fasta_index = nullptr; bo.device = opt.device; // (FASTA order: bqc_set_fasta_index below)
    bo.shard_tail = shard && shard->index > 0 ? 1u : 0u;
    if (!opt.no_sketch) {
        bo.sketch.n_k = (uint32_t)opt.klist.size(); bo.sketch.klist = opt.klist.data();
        bo.sketch.n_q = (uint32_t)opt.q_cutoff.size(); bo.sketch.qlist = opt.q_cutoff.data();
        bo.sketch.e = opt.e; bo.sketch.seed = opt.seed;
    }
    bqc_ctx* ctx = nullptr;
    int rc = 0;
    std::string create_err;
    double t_create_s = 0, t_c_warm = 0, t_c_reader = 0, t_c_create = 0;
    std::thread creator([&] {
        const auto c0 = clk::now();
        // Who sets up when (round 4): the warm-up thread starts the runtime (~60 ms) and then makes the program's four streams one after the
        // other (20 + 3 x 8-10 ms: hardware queues, serial whoever asks); the reader allocates its buffers as soon as the runtime is up
        // and takes the SECOND stream; the context is created here as soon as the FIRST stream exists — a few milliseconds of allocations
        // and memsets, behind the reader's allocations by then (side by side the two were measured to hold each other up at the runtime's
        // locks in round 3, when each also made its own streams and the reader page-locked 96 MB with hipHostMalloc) — and the references
        // go to the card while the reader still waits for its stream.
        const auto c1 = clk::now();
        const auto c2 = clk::now();
        rc = bqc_create(&bo, &ctx);
        if (rc) create_err = bqc_last_error(nullptr);
        else if (!shard) { // one allocation for the contigs, sized from the BAM header, while nothing runs on the card yet (an allocation beside running kernels waits for them)
            uint64_t total = 0;
            for (uint32_t r = 0; r < n_refs; ++r) total += H.ref_lens[r];
            (void)bqc_reserve_references(ctx, total, n_refs);
        }
        t_create_s = secs(c0, clk::now());
        t_c_warm = secs(c0, c1); t_c_reader = secs(c1, c2); t_c_create = secs(c2, clk::now());
    });
    // reference genome: all contigs that are BAM references, FASTA order kept for the cursor rule
    // One of several processes over a file needs the contigs ITS records lie on, not the genome: the FASTA file is only indexed
    // here (where its records are), and a contig is encoded and uploaded when the first batch with a read on it arrives
    // (ensure_refs below) — eight workers parsing 3 GB of text each on shared host cores would cost more than their record loops.
    std::vector<FastaRecord> fa;
    std::string ferr;
    MappedFasta lazy_fa;
    const bool lazy_refs = shard && !(getenv("BQC_EAGER_REFS") && getenv("BQC_EAGER_REFS")[0] == '1') && lazy_fa.open(opt.referenceFile.c_str()) == 1;
    if (lazy_refs) { for (const auto& r : lazy_fa.recs) fa.push_back(FastaRecord{r.name, {}}); }
    else if (!load_fasta(opt.referenceFile.c_str(), &H.ref_names, fa, ferr)) fprintf(stderr, "%s\n", ferr.c_str()); // return value ignored (:291)
    std::vector<int32_t> fasta_index(std::max(1u, n_refs), -1);
    for (uint32_t r = 0; r < n_refs; ++r)
        for (size_t i = 0; i < fa.size(); ++i) if (fa[i].name == H.ref_names[r]) { fasta_index[r] = (int32_t)i; break; }
    const auto t_fasta = clk::now();
    creator.join();
    // The reader may launch its first inflate kernel once the context exists AND its own set-up is done (it is still waiting for its
    // stream at this point, as a rule): a thread of its own says so, this one goes on to the references.
    std::thread allow_thr;
    struct AllowJoiner { std::thread& t; ~AllowJoiner() { if (t.joinable()) t.join(); } } allow_joiner{allow_thr};
    if (use_gpu_reader) allow_thr = std::thread([&gpu_reader_opened, &gpu_rd, ctx, rc] {
        while (!gpu_reader_opened.load()) std::this_thread::sleep_for(std::chrono::microseconds(100));
        // the batches' coverage anchors are made on the card from the first batch on (BQC_DEVICE_ANCHORS=0: the host's pass, columns and all)
        const char* da = getenv("BQC_DEVICE_ANCHORS");
        if (!rc && ctx && !(da && da[0] == '0')) gpu_rd.set_anchor_context(ctx);
        gpu_rd.allow_kernels(); // (the context exists: the card may get busy)
    });
    const auto t_create = clk::now();
    if (rc) { fprintf(stderr, "ERROR: %s\n", create_err.c_str()); stop_decoder(); return shard_abort(); }
    if ((rc = bqc_set_fasta_index(ctx, fasta_index.data()))) { fprintf(stderr, "ERROR: %s\n", bqc_last_error(ctx)); stop_decoder(); bqc_destroy(ctx); return shard_abort(); }
    // The references go to the card BEHIND the start of the record loop: a thread of its own uploads the contigs in the BAM header's order
    // (a human genome is 3.1 GB: 0.35-0.45 s that the loop used to wait for) and the submitting thread only waits when a batch holds a
    // read on a contig that is not there yet — the first one, as a rule.  (bqc_set_reference may be called beside bqc_submit* for contigs
    // no submitted batch refers to: include/bamqc.h.)
    std::vector<uint8_t> ref_loaded(std::max(1u, n_refs), 0); // lazy_refs: 1 loaded; background upload: see ref_state
    std::unique_ptr<std::atomic<uint8_t>[]> ref_state(new std::atomic<uint8_t>[std::max(1u, n_refs)]); // 0 on its way, 1 on the card, 2 failed
    for (uint32_t r = 0; r < std::max(1u, n_refs); ++r) ref_state[r] = 0;
    std::mutex ref_m;
    std::condition_variable ref_cv;
    std::string ref_err;
    std::atomic<bool> ref_stop{false};
    std::thread ref_loader;
    const bool bg_refs = !lazy_refs;
    // BQC_BG_REFS=1: the uploader runs BESIDE the record loop (which then starts 0.35 s earlier).  Measured on a 100 M-read file over a
    // human-sized genome: the loop itself becomes 0.5-0.6 s longer (1.45-1.55 s against 0.9), from pageable and from page-locked memory
    // alike — page-locking and large copies beside running inflate kernels hold up the other threads' calls into the runtime — so by
    // default the uploader is waited for before the loop starts; what is kept of it: one allocation for all contigs, made with the context.
    const bool refs_beside_loop = bg_refs && getenv("BQC_BG_REFS") && getenv("BQC_BG_REFS")[0] == '1';
    auto upload_all = [&] {
        for (uint32_t r = 0; r < n_refs; ++r) {
            uint8_t st = 1;
            if (fasta_index[r] >= 0 && !ref_stop.load()) {
                auto& c = fa[fasta_index[r]].codes;
                // (from pageable memory: page-locking the 3.1 GB first — hipHostRegister — and copying at the link's speed measured SLOWER
                // for the whole genome, 0.63-0.82 s against 0.35-0.46 s)
                const int src = bqc_set_reference(ctx, (int32_t)r, c.data(), c.size());
                if (src) { std::lock_guard<std::mutex> lk(ref_m); if (ref_err.empty()) ref_err = bqc_last_error(ctx); st = 2; }
                // (the host copy is NOT given back here: returning a human genome's 3.1 GB to the kernel costs 0.35-0.4 s — measured:
                // "references" 0.57 s with the contigs freed one by one, 0.20 s without — and the program's exit does it for nothing)
            }
            { std::lock_guard<std::mutex> lk(ref_m); ref_state[r] = st; }
            ref_cv.notify_all();
        }
    };
    if (bg_refs && refs_beside_loop) ref_loader = std::thread(upload_all);
    else if (bg_refs) upload_all(); // (in this thread, before the loop: the default)
    auto destroy_ctx = [&]() { // (every way out: the uploader first, it uses the context)
        ref_stop = true;
        if (ref_loader.joinable()) ref_loader.join();
        bqc_destroy(ctx);
    };
    std::vector<raw_vector<uint8_t>> uploaded_codes; // (lazy_refs) host copies of contigs that are on the card: kept, see above
    double t_lazy_refs = 0;
    uint32_t n_lazy_refs = 0;
    double t_wait_refs = 0;
    std::vector<int32_t> rid_range; // (a batch whose columns stayed on the card names the range of its reference ids)
    auto ensure_refs = [&](const HostBatch& hb) -> int { // the contigs this batch's reads lie on are on the card before it is submitted
        const int32_t* rid_begin = hb.rid.data();
        const int32_t* rid_end = hb.rid.data() + hb.rid.size();
        if (hb.anchored) {
            rid_range.clear();
            for (int32_t r = hb.rid_min; r <= hb.rid_max; ++r) rid_range.push_back(r);
            rid_begin = rid_range.data(); rid_end = rid_range.data() + rid_range.size();
        }
        if (bg_refs) { // wait for the uploader where it has not got to yet
            int32_t last = -1;
            for (const int32_t* q = rid_begin; q != rid_end; ++q) {
                const int32_t rid = *q;
                if (rid == last || rid < 0 || (uint32_t)rid >= n_refs) continue;
                last = rid;
                if (ref_state[rid].load() == 1) continue;
                const auto w0 = clk::now();
                std::unique_lock<std::mutex> lk(ref_m);
                ref_cv.wait(lk, [&] { return ref_state[rid].load() != 0; });
                t_wait_refs += secs(w0, clk::now());
                if (ref_state[rid].load() == 2) { fprintf(stderr, "ERROR: %s\n", ref_err.c_str()); return 1; }
            }
            return 0;
        }
        std::vector<size_t> which;
        std::vector<uint32_t> rids;
        int32_t last = -1;
        for (const int32_t* q = rid_begin; q != rid_end; ++q) {
            const int32_t rid = *q;
            if (rid == last || rid < 0 || (uint32_t)rid >= n_refs) continue;
            last = rid;
            if (ref_loaded[rid] || fasta_index[rid] < 0) continue;
            ref_loaded[rid] = 1;
            which.push_back((size_t)fasta_index[rid]);
            rids.push_back((uint32_t)rid);
        }
        if (which.empty()) return 0;
        const auto l0 = clk::now();
        std::vector<raw_vector<uint8_t>> codes(which.size());
        std::vector<raw_vector<uint8_t>*> dst;
        for (auto& c : codes) dst.push_back(&c);
        if (!lazy_fa.encode(which, dst)) { fprintf(stderr, "ERROR: out of memory while loading %s\n", opt.referenceFile.c_str()); return 1; }
        for (size_t k = 0; k < rids.size(); ++k)
            if (bqc_set_reference(ctx, (int32_t)rids[k], codes[k].data(), codes[k].size())) { fprintf(stderr, "ERROR: %s\n", bqc_last_error(ctx)); return 1; }
        for (auto& c : codes) uploaded_codes.push_back(std::move(c));
        t_lazy_refs += secs(l0, clk::now());
        n_lazy_refs += (uint32_t)rids.size();
        return 0;
    };
    const auto t_setup = clk::now();
    since_launch("record loop starts");

    int status = 0;
    uint64_t n_noqual_deferred = 0;
    const bool pinned = !(getenv("BQC_NO_PINNED") && getenv("BQC_NO_PINNED")[0] == '1'); // (1: the staging path of bqc_submit, for comparison)
    if (pinned) bqc_raw_vector_free_hook = pin_free_hook;
    struct InFlight { uint64_t ticket; std::unique_ptr<HostBatch> hb; };
    std::deque<InFlight> inflight; // batches whose columns the device may still be reading
    auto recycle = [&](bool wait_all) {
        while (!inflight.empty()) {
            const int up = bqc_batch_uploaded(ctx, inflight.front().ticket, wait_all || inflight.size() > 3 ? 1 : 0);
            if (up == 0) break;
            std::lock_guard<std::mutex> lk(Q.m);
            if (Q.spare.size() < 8) Q.spare.push_back(std::move(inflight.front().hb));
            inflight.pop_front();
        }
    };
    for (;;) {
        std::unique_ptr<HostBatch> hb;
        const auto w0 = clk::now();
        {
            std::unique_lock<std::mutex> lk(Q.m);
            Q.cv.wait(lk, [&] { return !Q.q.empty() || Q.done; });
            if (Q.q.empty()) break;
            hb = std::move(Q.q.front());
            Q.q.pop_front();
            Q.cv.notify_all();
        }
        if (status) continue; // drain
        { // check_read_len (QualityCheck.hpp:70-79, called at bamqualcheck.cpp:357,378; return value ignored): one line per primary
          // record with a first / last flag whose quality string is empty while its sequence is not
            size_t n_noqual = hb->anchored ? hb->n_noqual : 0; // (a batch whose columns stayed on the card: counted there)
            for (uint16_t f : hb->flag) n_noqual += (f & BQC_FLAG_NO_QUAL) && !(f & 0x900) && (f & 0xC0);
            // (the reader on the card: printed once the pass is known to be the one that counts — it may still hand the file over to the host
            // reader, which then prints them itself — i.e. behind the loop and in front of a record error: the lines of the batches BEFORE
            // a failing record all come first, where the reference interleaves them with nothing either; the failing batch's own lines
            // are the host decoder's to print)
            if (use_gpu_reader) { n_noqual_deferred += n_noqual; n_noqual = 0; }
            if (n_noqual) {
                static const char msg[] = "ERROR: length of sequence and quality is not the same\n";
                std::string out;
                out.reserve(std::min<size_t>(n_noqual, 4096) * (sizeof msg - 1));
                for (size_t k = 0; k < n_noqual; ++k) {
                    out.append(msg, sizeof msg - 1);
                    if (out.size() >= 4096 * (sizeof msg - 1)) { fwrite(out.data(), 1, out.size(), stderr); out.clear(); }
                }
                fwrite(out.data(), 1, out.size(), stderr);
            }
        }
        if (ensure_refs(*hb)) { status = 1; continue; }
        const bqc_batch v = hb->view();
        const auto s0 = clk::now();
        t_wait += secs(w0, s0);
        n_total += v.n_reads;
        if (pinned || hb->d_seq) { // (BQC_NO_PINNED=1 is about host columns: a payload that lives on the card can only be taken from there)
            if (!hb->d_seq) pin_batch(*hb); // (a batch decoded on the card: its payload is there already, its small fixed columns are copied staged)
            uint64_t ticket = 0;
            if (hb->anchored) { rc = bqc_submit_anchored(ctx, &v, (bqc_anchored*)hb->anchored, &ticket); hb->anchored = nullptr; }
            else rc = bqc_submit_async(ctx, &v, &ticket);
            if (rc) { fprintf(stderr, "%s\n", bqc_last_error(ctx)); status = 1; }
            inflight.push_back(InFlight{ticket, std::move(hb)});
            recycle(false);
        } else {
            if ((rc = bqc_submit(ctx, &v))) { fprintf(stderr, "%s\n", bqc_last_error(ctx)); status = 1; }
            std::lock_guard<std::mutex> lk(Q.m); // (bqc_submit has staged the batch)
            if (Q.spare.size() < 4) Q.spare.push_back(std::move(hb));
        }
        t_submit += secs(s0, clk::now());
    }
    recycle(true);
    ref_stop = true; // (contigs no read has asked for are not waited for)
    if (ref_loader.joinable()) ref_loader.join();
    if (!status && (rc = bqc_sync(ctx))) { fprintf(stderr, "%s\n", bqc_last_error(ctx)); status = 1; } // what the device found in the last batches
    dec.join();
    if (timing) fprintf(stderr, "[timing] records decoded %s\n", use_gpu_reader ? "on the GPU (csrc/gpu_bam.hip)" : "on the host");
    if (timing) { // (every buffer of the run exists at this point: what is in use now is the run's peak)
        size_t free_b = 0, total_b = 0, pinned = 0;
        { std::lock_guard<std::mutex> lk(g_pins.m); for (auto& kv : g_pins.blocks) pinned += kv.second; }
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            fprintf(stderr, "[timing] device memory in use at the end of the record loop: %.2f GB of %.0f GB; page-locked decode buffers %.2f GB\n", (total_b - free_b) / 1e9, total_b / 1e9, pinned / 1e9);
    }
    if (timing && refs_beside_loop) fprintf(stderr, "[timing] references uploaded beside the record loop; the submitting thread waited %.3f s for them\n", t_wait_refs);
    if (timing && lazy_refs) fprintf(stderr, "[timing] %u of %u contigs loaded, when their first reads arrived: %.3f s\n", n_lazy_refs, n_refs, t_lazy_refs);
    if (timing && use_gpu_reader) fprintf(stderr, "[timing] %llu batches anchored on the card (fixed columns never on the host)\n", (unsigned long long)gpu_rd.batches_anchored());
    if (timing && use_gpu_reader)
        fprintf(stderr, "[timing] reader on the card: pread %.2f s summed over its reader threads, the producer waited %.2f s for them, the decode thread waited %.2f s for inflated runs\n",
                gpu_rd.seconds_reading(), gpu_rd.seconds_producer_waiting_for_chunks(), gpu_rd.seconds_waiting_for_runs());
    if (timing && use_gpu_reader && gpu_rd.batches_handed_over()) fprintf(stderr, "[timing] %llu batches held records the card does not decode and went through the host decoder\n", (unsigned long long)gpu_rd.batches_handed_over());
    if (timing)
        fprintf(stderr, "[timing] %llu records: decode thread busy %.2f s, submit thread (host pass + enqueue; page-locking %.2f s) %.2f s, waiting for the decoder %.2f s, loop %.2f s\n",
                (unsigned long long)n_total, t_decode, g_pins.t_register, t_submit, t_wait, secs(t_setup, clk::now()));
    if (Q.err_code == GpuBamReader::kUnsupported && status) Q.err_code = 0; // (an error of the records before it has been reported: that is the run's result)
    if (Q.err_code == GpuBamReader::kUnsupported) { // nothing has been reported yet: the same file again, through the host reader
        if (timing) fprintf(stderr, "[timing] %s\n", Q.err.c_str());
        destroy_ctx();
        g_pins.release_all();
        return run_program(argc, argv, shard, true);
    }
    for (uint64_t k = 0; k < n_noqual_deferred; ++k) fputs("ERROR: length of sequence and quality is not the same\n", stderr);
    if (!status && Q.err_code) {
        if (Q.err_code == BQC_ERR_IO) fprintf(stderr, "ERROR: Could not read record from BAM File %s\n", opt.bamFile.c_str()); // :308
        else fprintf(Q.err == "Read does not have Z" ? stdout : stderr, "%s\n", Q.err.c_str());
        status = 1;
    }
    // writeOutput iterates laneNames (std::map: lexicographic), including IDs inserted by getLane
    std::vector<const char*> names;
    std::vector<uint32_t> idx;
    for (auto& kv : rd.header().lane_names) { names.push_back(kv.first.c_str()); idx.push_back(kv.second); }
    if (shard) { // what needs the other processes: split check, coverage hand-over, sum of the state vectors, lane names
        bqc_shard_info si{};
        si.ctx = ctx; si.status = status;
        if (use_gpu_reader) { si.begin_block = shard_b0; si.end_block = shard_b1; si.first = gpu_rd.range_first(); si.over = gpu_rd.range_over(); }
        else { si.begin_block = bam_rd.range_begin_block(); si.end_block = bam_rd.range_end_block(); si.first = bam_rd.range_first(); si.over = bam_rd.range_over(); }
        si.sample_id = rd.header().sample_id.c_str();
        si.n_lane_names = (uint32_t)names.size(); si.lane_names = names.data(); si.lane_index = idx.data();
        bqc_shard_result sr{};
        const int verdict = shard->hook(shard->user, &si, &sr);
        if (verdict == BQC_SHARD_FALLBACK) { // the split could not be verified: the whole file again, in this process alone
            destroy_ctx();
            g_pins.release_all();
            return run_program(argc, argv, nullptr);
        }
        if (verdict != BQC_SHARD_WRITE) { destroy_ctx(); g_pins.release_all(); return verdict == BQC_SHARD_DONE && !status ? 0 : 1; }
        names.assign(sr.lane_names, sr.lane_names + sr.n_lane_names);
        idx.assign(sr.lane_index, sr.lane_index + sr.n_lane_names);
    }
    if (status) { destroy_ctx(); return 1; }
    const bqc_counts* counts = nullptr;
    const auto t_loop_end = clk::now();
    since_launch("record loop and its checks over");
    if ((rc = bqc_finalize(ctx, &counts))) { fprintf(stderr, "ERROR: %s\n", bqc_last_error(ctx)); destroy_ctx(); return 1; }
    const auto t_final = clk::now();
    bqc_header_info hi;
    hi.sample_id = rd.header().sample_id.c_str();
    hi.n_names = (uint32_t)names.size();
    hi.lane_names = names.data();
    hi.lane_index = idx.data();
    rc = bqc_write_bamqc(counts, &hi, opt.outputFile.c_str());
    const auto t_write = clk::now();
    // the program proper (tools/bamqualcheck.cpp sets BQC_FAST_EXIT) leaves without the static destructors of the HIP runtime:
    // the output file is complete and closed, the process is about to end anyway
    const bool fast_exit = !rc && getenv("BQC_FAST_EXIT") && getenv("BQC_FAST_EXIT")[0] == '1';
    const char* done_fd = fast_exit ? getenv("BQC_DONE_FD") : nullptr; // the front end (tools/bamqualcheck.cpp) waits for a byte there
    auto teardown = [&]() { // (device memory and page locks are released explicitly: left to the kernel's process teardown they cost 0.2 s more)
        destroy_ctx();
        g_pins.release_all();
    };
    if (!done_fd) teardown();
    if (timing) {
        fprintf(stderr, "[timing] start-up: waiting for the HIP runtime %.3f s, then for the GPU reader's set-up %.3f s, bqc_create %.3f s\n", t_c_warm, t_c_reader, t_c_create);
        fprintf(stderr, "[timing] phases: FASTA %.2f s (context created meanwhile in %.2f s), context %.2f s, references %.2f s, record loop %.2f s, finalize %.2f s, write %.2f s, destroy %.2f s%s\n",
                secs(t_begin, t_fasta), t_create_s, secs(t_fasta, t_create), secs(t_create, t_setup), secs(t_setup, t_loop_end), secs(t_loop_end, t_final), secs(t_final, t_write),
                secs(t_write, clk::now()), done_fd ? " (the context is released after the run has been reported complete)" : "");
    }
    if (rc) { fprintf(stderr, "ERROR: Could not write output file %s\n", opt.outputFile.c_str()); return 1; }
    since_launch("done");
    if (fast_exit) { // the program proper leaves without the static destructors of the HIP runtime: the output file is complete and closed
        fflush(stdout); fflush(stderr);
        if (done_fd) { // nothing is printed from here on: the run is reported complete, then this process cleans up on its own
            const int fd = atoi(done_fd);
            const unsigned char st = 0;
            (void)prctl(PR_SET_PDEATHSIG, 0); // (the front end leaves now; this process still hands its memory back)
            if (write(fd, &st, 1) == 1) {
                close(fd);
                const int nul = open("/dev/null", O_WRONLY);
                if (nul >= 0) { dup2(nul, 1); dup2(nul, 2); }
            }
            teardown();
        }
        _exit(0);
    }
    return 0;
}

extern "C" int bqc_main(int argc, const char** argv) { return run_program(argc, argv, nullptr); }

// The front end of `--gpus N` checks the command line ONCE, before it forks (a usage error is then printed once, and no worker
// starts RCCL for it), and learns the input path the workers will use from the same parser they use.
extern "C" int bqc_program_args(int argc, const char** argv, char* input_path, uint64_t cap)
{
    ProgramOptions opt;
    std::string perr;
    const int pr = parse_args(argc, argv, opt, perr);
    if (pr == 1) fprintf(stderr, "%s\n", perr.c_str());
    if (pr == 0 && input_path && cap) snprintf(input_path, (size_t)cap, "%s", opt.bamFile.c_str());
    return pr;
}

extern "C" int bqc_main_shard(int argc, const char** argv, uint32_t shard_index, uint32_t shard_count, bqc_shard_hook hook, void* user)
{
    if (!hook || shard_count == 0 || shard_index >= shard_count) return 1;
    const ShardArgs sa{shard_index, shard_count, hook, user};
    return run_program(argc, argv, &sa);
}
