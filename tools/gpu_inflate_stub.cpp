// Sanitizer builds of the host input code (tools/asan_host_io.sh) have no HIP: the BGZF reader's hooks into csrc/gpu_inflate.hip
// resolve to these — "no such device", so that the reader stays on its CPU decoder.  Not part of the product.
#include "../bamqc_amd/csrc/gpu_inflate.h"
extern "C" GpuInflater* bqc_gpu_inflater_create(int) { return nullptr; }
extern "C" void bqc_gpu_inflater_destroy(GpuInflater*) {}
extern "C" int bqc_gpu_inflate(GpuInflater*, const uint8_t*, size_t, const GiBlock*, size_t, uint8_t*, size_t) { return -1; }
