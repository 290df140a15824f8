#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_u;
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 __attribute__((aligned(1))) u32x3_u;
__global__ void k(uint32_t* out, const uint8_t* in)
{
    const uint32_t x = threadIdx.x;
    out[x] = __builtin_amdgcn_alignbyte(1u, 1u, x * 0x01010101u + (x >> 3));
    out[64 + x] = __builtin_amdgcn_perm(0xAABBCCDDu, 0x11223344u, (x & 1) ? 0x07060504u : 0x00010203u);
    const u32x4 v = *(const __attribute__((address_space(1))) u32x4_u*)(uintptr_t)(in + x);
    const u32x3 u = *(const __attribute__((address_space(1))) u32x3_u*)(uintptr_t)(in + x + 1);
    out[128 + x] = v.x ^ v.y ^ v.z ^ v.w;
    out[192 + x] = u.x + u.y + u.z;
}
int main()
{
    uint32_t* d; uint8_t* in; uint32_t h[256]; uint8_t hin[256];
    for (int i = 0; i < 256; ++i) hin[i] = (uint8_t)(i * 7 + 1);
    hipMalloc(&d, 1024); hipMalloc(&in, 256); hipMemcpy(in, hin, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, in);
    hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 12; ++i) printf("alignbyte(1,1,%08x)=%08x\n", i * 0x01010101u + (i >> 3), h[i]);
    printf("perm fwd %08x rc %08x\n", h[64], h[65]);
    int bad = 0;
    for (int x = 0; x < 64; ++x) {
        uint32_t w[4], e = 0, s = 0;
        for (int j = 0; j < 4; ++j) { memcpy(&w[j], hin + x + 4 * j, 4); e ^= w[j]; }
        for (int j = 0; j < 3; ++j) { uint32_t t; memcpy(&t, hin + x + 1 + 4 * j, 4); s += t; }
        if (e != h[128 + x] || s != h[192 + x]) ++bad;
    }
    printf("unaligned vector loads bad=%d\n", bad);
    return 0;
}
