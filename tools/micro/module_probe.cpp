// module_probe — what the first use of the library's kernels costs in a fresh process (code object load): run it a few times
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
extern "C" hipError_t bqc_short_init();
extern "C" hipError_t bqc_long_init();
extern "C" int bqc_calib_read4(unsigned long long bytes, int repeat);
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    double t = now();
    hipSetDevice(0); hipFree(nullptr);
    printf("runtime init %.1f ms\n", (now() - t) * 1e3); t = now();
    bqc_short_init();
    printf("hipFuncSetAttribute(k_short): %.1f ms\n", (now() - t) * 1e3); t = now();
    bqc_long_init();
    printf("hipFuncSetAttribute(k_long): %.1f ms\n", (now() - t) * 1e3); t = now();
    bqc_calib_read4(1 << 20, 1);
    printf("first launch of a kernel of another source file (k_cov.hip): %.1f ms\n", (now() - t) * 1e3); t = now();
    bqc_calib_read4(1 << 20, 1);
    printf("the same again: %.1f ms\n", (now() - t) * 1e3);
    return 0;
}
