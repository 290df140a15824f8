// copy_probe.cpp — what the FIRST copies of a process cost on this card, in fresh processes (round 4: bqc_create's first copy of a few bytes from
// pageable memory took 50 ms).  usage: ./copy_probe ORDER   ORDER = letters, executed in order, each timed:
//   p pageable H2D copy of 256 B (hipMemcpy)      P the same, 64 MB      r H2D of 256 B from hipHostRegister'ed memory, async + sync
//   h H2D of 256 B from hipHostMalloc'ed memory    s hipMemsetAsync 4 KB + sync      d D2H 256 B pageable      k an empty kernel
//   build: hipcc --offload-arch=gfx950 -O2 -o copy_probe copy_probe.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 1234567) *p = 1; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const char* order = argc > 1 ? argv[1] : "pprsdk";
    double t0 = now();
    (void)hipSetDevice(0); (void)hipFree(nullptr);
    printf("runtime init %.1f ms\n", now() - t0);
    t0 = now();
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    printf("stream %.1f ms\n", now() - t0);
    t0 = now();
    void* d = nullptr; (void)hipMalloc(&d, 64u << 20);
    printf("hipMalloc 64 MB %.1f ms\n", now() - t0);
    std::vector<unsigned char> big(64u << 20, 1);
    static unsigned char small[4096];
    for (const char* c = order; *c; ++c) {
        t0 = now();
        switch (*c) {
        case 'p': (void)hipMemcpy(d, small, 256, hipMemcpyHostToDevice); break;
        case 'P': (void)hipMemcpy(d, big.data(), big.size(), hipMemcpyHostToDevice); break;
        case 'r': { static unsigned char reg[8192]; (void)hipHostRegister(reg, sizeof reg, hipHostRegisterDefault); (void)hipMemcpyAsync(d, reg, 256, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); break; }
        case 'h': { void* h = nullptr; (void)hipHostMalloc(&h, 4096, hipHostMallocDefault); (void)hipMemcpyAsync(d, h, 256, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); break; }
        case 's': (void)hipMemsetAsync(d, 0, 4096, s); (void)hipStreamSynchronize(s); break;
        case 'd': (void)hipMemcpy(small, d, 256, hipMemcpyDeviceToHost); break;
        case 'k': hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (int*)nullptr); (void)hipStreamSynchronize(s); break;
        }
        printf("  %c %.1f ms\n", *c, now() - t0);
    }
    return 0;
}
