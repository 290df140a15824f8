// pinned_probe.cpp — what page-locked staging costs and buys on this box: hipHostMalloc / hipHostRegister time per GB,
// H2D bandwidth from pageable, registered and allocated host memory, and the fixed costs of starting HIP.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using clk = std::chrono::steady_clock;
static double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    auto t0 = clk::now();
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    auto t1 = clk::now();
    printf("hip init %.3f s\n", secs(t0, t1));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t N = 512ull << 20;
    void* d; t0 = clk::now(); CK(hipMalloc(&d, N)); t1 = clk::now(); printf("hipMalloc 512 MiB %.4f s\n", secs(t0, t1));
    void* d2; t0 = clk::now(); CK(hipMalloc(&d2, N)); t1 = clk::now(); printf("hipMalloc 512 MiB (2nd) %.4f s\n", secs(t0, t1));
    // pageable
    char* p = (char*)malloc(N); memset(p, 1, N);
    for (int r = 0; r < 3; ++r) { t0 = clk::now(); CK(hipMemcpyAsync(d, p, N, hipMemcpyHostToDevice, s)); auto tm = clk::now(); CK(hipStreamSynchronize(s)); t1 = clk::now(); printf("H2D pageable: call returns after %.4f s, done after %.4f s = %.1f GB/s\n", secs(t0, tm), secs(t0, t1), N / secs(t0, t1) / 1e9); }
    // registered
    t0 = clk::now(); CK(hipHostRegister(p, N, hipHostRegisterDefault)); t1 = clk::now(); printf("hipHostRegister 512 MiB (touched) %.4f s\n", secs(t0, t1));
    for (int r = 0; r < 3; ++r) { t0 = clk::now(); CK(hipMemcpyAsync(d, p, N, hipMemcpyHostToDevice, s)); auto tm = clk::now(); CK(hipStreamSynchronize(s)); t1 = clk::now(); printf("H2D registered: call returns after %.5f s, done after %.4f s = %.1f GB/s\n", secs(t0, tm), secs(t0, t1), N / secs(t0, t1) / 1e9); }
    t0 = clk::now(); CK(hipHostUnregister(p)); t1 = clk::now(); printf("hipHostUnregister %.4f s\n", secs(t0, t1));
    // allocated
    void* h; t0 = clk::now(); CK(hipHostMalloc(&h, N, hipHostMallocDefault)); t1 = clk::now(); printf("hipHostMalloc 512 MiB %.4f s\n", secs(t0, t1));
    t0 = clk::now(); memset(h, 2, N); t1 = clk::now(); printf("first touch of it %.4f s\n", secs(t0, t1));
    t0 = clk::now(); memcpy(h, p, N); t1 = clk::now(); printf("memcpy pageable -> pinned %.4f s = %.1f GB/s (1 thread)\n", secs(t0, t1), N / secs(t0, t1) / 1e9);
    for (int r = 0; r < 3; ++r) { t0 = clk::now(); CK(hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, s)); auto tm = clk::now(); CK(hipStreamSynchronize(s)); t1 = clk::now(); printf("H2D pinned: call returns after %.5f s, done after %.4f s = %.1f GB/s\n", secs(t0, tm), secs(t0, t1), N / secs(t0, t1) / 1e9); }
    // two copies on two streams at once
    hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    t0 = clk::now(); CK(hipMemcpyAsync(d, h, N / 2, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync(d2, (char*)h + N / 2, N / 2, hipMemcpyHostToDevice, s2)); CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2)); t1 = clk::now();
    printf("H2D pinned, two streams: %.1f GB/s\n", N / secs(t0, t1) / 1e9);
    // many small copies: per-call overhead
    t0 = clk::now(); for (int i = 0; i < 100; ++i) CK(hipMemcpyAsync((char*)d + i * 4096, (char*)h + i * 4096, 4096, hipMemcpyHostToDevice, s)); auto tm = clk::now(); CK(hipStreamSynchronize(s)); t1 = clk::now();
    printf("100 x 4 KiB H2D: enqueue %.1f us each, all done after %.1f us\n", secs(t0, tm) * 1e4, secs(t0, t1) * 1e6);
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    t0 = clk::now(); for (int i = 0; i < 100; ++i) { CK(hipEventRecord(ev, s)); CK(hipStreamWaitEvent(s2, ev, 0)); } t1 = clk::now(); printf("event record + stream wait: %.1f us per pair\n", secs(t0, t1) * 1e4);
    t0 = clk::now(); CK(hipHostFree(h)); t1 = clk::now(); printf("hipHostFree %.4f s\n", secs(t0, t1));
    return 0;
}
