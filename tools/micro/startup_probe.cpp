// startup_probe — what the pieces of a process's GPU start-up cost on this box (each as a fresh process: run it several times)
//   usage: startup_probe MODE    MODE = seq | par | one | hostmem
// seq: runtime init, then 4 streams one after the other; par: the 4 streams by 4 threads at once; one: 1 stream, then 3 more;
// hostmem: hipHostMalloc of 6 x 16 MB against malloc + touch + hipHostRegister
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_nop(int* p) { if (p) *p = 1; }
int main(int argc, char** argv)
{
    const char* mode = argc > 1 ? argv[1] : "seq";
    if (getenv("PROBE_QUEUES")) setenv("GPU_MAX_HW_QUEUES", getenv("PROBE_QUEUES"), 1);
    const double t0 = now();
    hipSetDevice(0);
    hipFree(nullptr);
    const double t1 = now();
    printf("%s: runtime init %.1f ms\n", mode, (t1 - t0) * 1e3);
    if (!strcmp(mode, "hostmem")) {
        double a = now();
        void* p[6];
        for (int i = 0; i < 6; ++i) hipHostMalloc(&p[i], 16u << 20, hipHostMallocDefault);
        double b = now();
        printf("  6 x hipHostMalloc(16 MB): %.1f ms\n", (b - a) * 1e3);
        void* q[6];
        a = now();
        for (int i = 0; i < 6; ++i) { q[i] = aligned_alloc(1 << 21, 16u << 20); memset(q[i], 0, 16u << 20); }
        b = now();
        for (int i = 0; i < 6; ++i) hipHostRegister(q[i], 16u << 20, hipHostRegisterDefault);
        double c = now();
        printf("  6 x (aligned_alloc + memset 16 MB): %.1f ms, 6 x hipHostRegister: %.1f ms\n", (b - a) * 1e3, (c - b) * 1e3);
        a = now();
        void* big;
        hipHostMalloc(&big, 96u << 20, hipHostMallocDefault);
        b = now();
        printf("  1 x hipHostMalloc(96 MB): %.1f ms\n", (b - a) * 1e3);
        return 0;
    }
    hipStream_t s[4];
    if (!strcmp(mode, "par")) {
        std::vector<std::thread> th;
        double dt[4];
        for (int i = 0; i < 4; ++i) th.emplace_back([&, i] { hipSetDevice(0); const double a = now(); hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); dt[i] = now() - a; });
        for (auto& t : th) t.join();
        printf("  4 streams by 4 threads: %.1f ms in all (%.1f %.1f %.1f %.1f)\n", (now() - t1) * 1e3, dt[0] * 1e3, dt[1] * 1e3, dt[2] * 1e3, dt[3] * 1e3);
    } else {
        for (int i = 0; i < 4; ++i) { const double a = now(); hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); printf("  stream %d: %.1f ms\n", i, (now() - a) * 1e3); }
    }
    // first kernel on each stream (code object load on the first one)
    int* d;
    double a = now();
    hipMalloc(&d, 64);
    printf("  hipMalloc(64): %.1f ms\n", (now() - a) * 1e3);
    for (int i = 0; i < 4; ++i) { a = now(); hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s[i], d); hipStreamSynchronize(s[i]); printf("  first kernel on stream %d: %.1f ms\n", i, (now() - a) * 1e3); }
    a = now();
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, 0, d); hipStreamSynchronize(0);
    printf("  first kernel on the null stream: %.1f ms\n", (now() - a) * 1e3);
    a = now();
    hipStream_t s5; hipStreamCreateWithFlags(&s5, hipStreamNonBlocking);
    printf("  a fifth stream: %.1f ms\n", (now() - a) * 1e3);
    printf("  total %.1f ms\n", (now() - t0) * 1e3);
    return 0;
}
