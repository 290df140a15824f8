// lds_atomic.hip — what a wave64 LDS atomic add costs on CDNA4, by address pattern (round 4: the 16 packed 8-mer counter updates of a
// k_long row cost 0.32 of the kernel's 1.42 ms, the 32 triplet-bin updates of the same row 0.02).  One workgroup of W waves per CU
// (k_long's shape: 12), every wave runs ITER rounds of 16 `ds_add_u32` / `ds_add_rtn_u32`; the addresses of a round are computed
// beforehand (a hash of lane, round and slot), the time is the shader clock around the loop, averaged over the waves.
//   patterns: 0 every lane its own dword, consecutive (no conflict)       1 random dwords in 64 KB (the 8-mer table)
//             2 random dwords in 1 KB (the triplet bins)                   3 random among 64 dwords of 1 KB (bins used in practice)
//             4 all lanes one address                                       5 random dwords in 64 KB, bank = lane (no bank conflict, distinct addresses)
//             6 random in 64 KB, 31 of 64 lanes active                      7 random in 64 KB, 8 of 64 lanes active
//   build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic lds_atomic.hip      run: ./lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define ITER 1000

__device__ __forceinline__ unsigned mix(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int PAT, bool RTN> __global__ __launch_bounds__(1024) void k_atom(unsigned long long* out, unsigned* sink, unsigned seed)
{
    extern __shared__ unsigned lds[];
    for (unsigned i = threadIdx.x; i < 16384u; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u;
    unsigned acc = 0;
    const bool active = PAT == 6 ? (lane & 1u) != 0u && lane < 63u : PAT == 7 ? (lane & 7u) == 0u : true;
    // the addresses of round 0; every round moves them by a lane-dependent step that keeps the pattern (two VALU instructions per atomic)
    unsigned a[16], step[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const unsigned r = mix(seed + threadIdx.x * 977u + (unsigned)s + blockIdx.x * 7919u), q = mix(r + 99u) | 1u;
        if (PAT == 0) { a[s] = (lane + 64u * (unsigned)s) & 16383u; step[s] = 64u; }
        else if (PAT == 1 || PAT == 6 || PAT == 7) { a[s] = r & 16383u; step[s] = q & 16383u; }
        else if (PAT == 2) { a[s] = r & 255u; step[s] = q & 255u; }
        else if (PAT == 3) { a[s] = (r & 63u) * 4u; step[s] = (q & 63u) * 4u; }
        else if (PAT == 4) { a[s] = (unsigned)s * 64u; step[s] = 1u; }
        else { a[s] = ((r & 511u) * 32u + (lane & 31u)) & 16383u; step[s] = (q & 511u) * 32u; }
    }
    const unsigned amask = PAT == 2 || PAT == 3 ? 255u : 16383u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
        if (active) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (RTN) acc += __hip_atomic_fetch_add(&lds[a[s]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_fetch_add(&lds[a[s]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                a[s] = (a[s] + step[s]) & amask;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 0x12345u) sink[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && lds[seed & 16383u] == 0xFFFFFFFFu) sink[1] = 1;
}

// the same loop without the atomics: the cost of computing the addresses
template <int PAT> __global__ __launch_bounds__(1024) void k_base(unsigned long long* out, unsigned* sink, unsigned seed)
{
    const unsigned lane = threadIdx.x & 63u;
    unsigned acc = 0;
    unsigned a[16], step[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const unsigned r = mix(seed + threadIdx.x * 977u + (unsigned)s + blockIdx.x * 7919u), q = mix(r + 99u) | 1u;
        a[s] = r & 16383u; step[s] = q & 16383u;
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) { asm volatile("" : "+v"(a[s])); a[s] = (a[s] + step[s]) & 16383u; }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc += a[s];
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 0x12345u) sink[0] = acc;
}

template <typename F> static double run(F launch, int waves, int n_cu, unsigned long long* d_out)
{
    std::vector<unsigned long long> h((size_t)n_cu * waves);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        launch();
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += (double)v;
        s /= (double)h.size();
        if (s < best) best = s;
    }
    return best;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount;
    unsigned long long* d_out;
    unsigned* d_sink;
    hipMalloc(&d_out, (size_t)n_cu * 16 * 8);
    hipMalloc(&d_sink, 64);
    printf("%d CUs; ITER %d rounds of 16 atomics; shader-clock ticks per wave-level atomic (address arithmetic subtracted), W waves per CU\n", n_cu, ITER);
    for (int waves : {4, 12, 16}) {
        const dim3 g(n_cu), b(waves * 64);
        const double base = run([&] { hipLaunchKernelGGL(k_base<1>, g, b, 0, 0, d_out, d_sink, 17u); }, waves, n_cu, d_out);
#define ONE(PAT, RTN, NAME) { \
        const double t = run([&] { hipLaunchKernelGGL((k_atom<PAT, RTN>), g, b, 65536, 0, d_out, d_sink, 17u); }, waves, n_cu, d_out); \
        printf("W=%2d %-46s %-7s %7.1f ticks per atomic and wave = %6.1f per atomic and CU\n", waves, NAME, RTN ? "rtn" : "no-rtn", (t - base) / (ITER * 16.0), (t - base) / (ITER * 16.0) / waves); }
        ONE(0, false, "own dword, consecutive (conflict-free)") ONE(0, true, "own dword, consecutive (conflict-free)")
        ONE(1, false, "random in 64 KB") ONE(1, true, "random in 64 KB")
        ONE(2, false, "random in 1 KB") ONE(2, true, "random in 1 KB")
        ONE(3, false, "random among 64 dwords") ONE(4, false, "all lanes one address")
        ONE(5, false, "random in 64 KB, bank = lane") ONE(5, true, "random in 64 KB, bank = lane")
        ONE(6, false, "random in 64 KB, 31 lanes") ONE(7, false, "random in 64 KB, 8 lanes")
        printf("W=%2d address arithmetic alone: %.1f ticks per round of 16\n", waves, base / ITER);
    }
    return 0;
}
