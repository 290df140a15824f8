// kernels_common.h — device helpers shared by the HIP kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_types.h"
#include "../../include/bamqc.h"

#define WAVE 64

// ---------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// Workgroup barrier that first drains this wave's outstanding LDS operations (s_waitcnt lgkmcnt(0)).  With this toolchain a
// plain __syncthreads() reached through a loop back-edge was observed WITHOUT that wait in front of its s_barrier, so that
// LDS atomics issued just before it (non-returning ds_add) could land after another wave had already read the counters
// behind the barrier (k_cov lost histogram counts that way, about once per 25 000 tiles).  Use this instead of
// __syncthreads() in every kernel.
__device__ __forceinline__ void block_sync()
{
    __builtin_amdgcn_s_waitcnt(0xC07F); // vmcnt / expcnt: no wait, lgkmcnt(0)
    __syncthreads();
}

// Atomic adds to counters in DEVICE memory, said so to the compiler (round 4): a pointer it cannot trace to a kernel argument — one
// loaded from a table of pointers, as the sketch's are — otherwise gives a FLAT atomic, which counts on vmcnt AND lgkmcnt and completes
// out of order, so that every later wait for an LDS read also waits for the atomic's trip to memory.
typedef __attribute__((address_space(1))) unsigned long long gmem_u64;
typedef __attribute__((address_space(1))) uint32_t gmem_u32;
__device__ __forceinline__ void gadd(uint64_t* p, uint64_t v)
{
    __hip_atomic_fetch_add((gmem_u64*)(uintptr_t)p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gadd32(uint32_t* p, uint32_t v)
{
    __hip_atomic_fetch_add((gmem_u32*)(uintptr_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Add 1 to *addr for every lane with pred, aggregating lanes that hit the same address
// (hot histogram bins: mapQ 60, mismatch 0, ...) into one atomic per distinct address.
__device__ __forceinline__ void wave_inc(bool pred, uint64_t* addr)
{
    uint64_t m = __ballot(pred);
    const uint64_t a = (uint64_t)addr;
    while (m) {
        const int leader = __ffsll((unsigned long long)m) - 1;
        const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)a, leader);
        const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(a >> 32), leader);
        const bool same = pred && (uint32_t)a == lo && (uint32_t)(a >> 32) == hi;
        const uint64_t sm = __ballot(same);
        if (lane_id() == leader) gadd(addr, (uint64_t)__popcll((unsigned long long)sm));
        m &= ~sm;
    }
}

// inclusive prefix sum over the 64 lanes on the VALU (DPP row shifts / broadcasts): no LDS round trips
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) 
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// nibble -> Dna5 ordinal (A0 C1 G2 T3, everything else 4) as a 16 x 4-bit table; BAM code "=ACMGRSVTWYHKDBN"
//   nib:  0 1 2 3 4 5 6 7 8 9 a b c d e f
//   fwd:  4 0 1 4 2 4 4 4 3 4 4 4 4 4 4 4
//   rc :  4 3 2 4 1 4 4 4 0 4 4 4 4 4 4 4   (complement; non-ACGT stays "other")
#define LUT5_FWD 0x4444444344424104ull
#define LUT5_RC  0x4444444044414234ull
__device__ __forceinline__ uint32_t lut5(uint64_t lut, uint32_t nib) { return (uint32_t)(lut >> (nib * 4)) & 7u; }

__device__ __forceinline__ uint32_t reverse8x2(uint32_t h) // reverse the order of 8 packed 2-bit bases
{
    const uint32_t x = __brev(h) >> 16;                   // reverses bit order: base order reversed, bits in pair swapped
    return ((x & 0xAAAAu) >> 1) | ((x & 0x5555u) << 1);   // swap the two bits of every base back
}

