// swar.h — register-level helpers shared by the per-base kernels (k_short.hip: reads of up to 255 bases; k_long.hip: any
// length): funnel shifts / byte permutes, DPP lane moves, the one-hot base planes of 8 packed BAM nibbles, the packed u8
// 8-mer counters in LDS with exact wrap accounting, unaligned multi-dword global loads.
#pragma once
#include "kernels_common.h"

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
__device__ __forceinline__ uint32_t vperm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
__device__ __forceinline__ uint32_t bfe(uint32_t x, uint32_t off, uint32_t w) { return __builtin_amdgcn_ubfe(x, off, w); }

// cross-lane moves on the VALU (DPP) instead of ds_bpermute: no LDS round trip
__device__ __forceinline__ uint32_t lane_next(uint32_t x) // value of lane + 1 (0 for lane 63)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t lane_prev(uint32_t x) // value of lane - 1 (0 for lane 0)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
}
// 8 nibble-spaced 2-bit values (bits [1:0] of every nibble) -> 16 contiguous bits in the low half, first nibble on top.
// Every step takes exactly the bits it needs (v_bfi), so the upper half of the result is junk.
__device__ __forceinline__ uint32_t squeeze2(uint32_t c)
{
    c = (c & 0x33333333u) | ((c >> 2) & ~0x33333333u);
    c = (c & 0x0F0F0F0Fu) | ((c >> 4) & ~0x0F0F0F0Fu);
    return (c & 0x00FF00FFu) | ((c >> 8) & ~0x00FF00FFu);
}

// a / b for a < 2^20, 0 < b < 2^10: float reciprocal estimate (off by at most one either way at these sizes), then exact
// correction with the remainder — a fraction of the instructions of the generic 32-bit division
__device__ __forceinline__ uint32_t small_div(uint32_t a, uint32_t b)
{
    uint32_t q = (uint32_t)((float)a * __builtin_amdgcn_rcpf((float)b));
    int32_t r = (int32_t)(a - q * b);
    if (r < 0) { --q; r += (int32_t)b; }
    if (r >= (int32_t)b) ++q;
    return q;
}

struct Planes { uint32_t a, c, g, t, oh, n; }; // one-hot masked planes, one-hot mask, literal-N mask (nibble LSBs)
__device__ __forceinline__ Planes planes_of(uint32_t x)
{
    const uint32_t M = 0x11111111u;
    const uint32_t p0 = x & M, p1 = (x >> 1) & M, p2 = (x >> 2) & M, p3 = (x >> 3) & M;
    const uint32_t s = p0 + p1 + p2 + p3;          // per-nibble popcount (0..4)
    Planes P;
    P.oh = __builtin_amdgcn_bitop3_b32(s, s >> 1, s >> 2, 0x10) & M; // a & ~b & ~c: popcount == 1
    P.n = (s >> 2) & M;                             // popcount == 4: literal 'N' (code 15)
    P.a = p0 & P.oh; P.c = p1 & P.oh; P.g = p2 & P.oh; P.t = p3 & P.oh;
    return P;
}

// Packed u8 8-mer counters: bin h lives in dword h >> 2, byte (4 - (h & 3)) & 3 — the byte that v_alignbyte_b32(1, 1, h)
// sets.  Exact accounting when a field wraps: every wrap of byte b is worth +256 for its bin and, because the carry
// spills into byte b + 1, -1 for that byte's bin.
__device__ __forceinline__ uint32_t t8_byte(uint32_t h) { return (4u - (h & 3u)) & 3u; }
__device__ __noinline__ void t8_wrap(uint64_t* __restrict__ em, uint32_t h, uint32_t old)
{
    const uint32_t d = h & ~3u;
    uint32_t b = t8_byte(h);
    while (b < 4u && ((old >> (8u * b)) & 0xFFu) == 0xFFu) {
        gadd(em + d + ((4u - b) & 3u), 256);
        if (b < 3u) gadd(em + d + ((3u - b) & 3u), (uint64_t)-1ll);
        ++b;
    }
}

// Rare: some old value of a batch of 8 window atomics has a byte >= 128.  A real function (by-value arguments) that recomputes
// the windows, so that the hot loop does not keep them in registers.  HB = first window of the batch, f = its count flags.
template <int HB>
__device__ __noinline__ void t8_check(uint64_t* __restrict__ em, uint32_t c32, uint32_t cx, uint32_t f, uint32_t o0, uint32_t o1, uint32_t o2,
                                      uint32_t o3, uint32_t o4, uint32_t o5, uint32_t o6, uint32_t o7)
{
    const uint32_t old[8] = {o0, o1, o2, o3, o4, o5, o6, o7};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int kw = HB + kk;
        const uint32_t h = (kw < 8 ? c32 >> (16 - 2 * kw) : kw == 8 ? c32 : __builtin_amdgcn_alignbit(c32, cx, 48 - 2 * kw)) & 0xFFFFu;
        const bool counted = (f >> (28 - 4 * kk)) & 1u; // a blocked window added 0
        if (counted && ((old[kk] >> (8u * t8_byte(h))) & 0xFFu) == 0xFFu) t8_wrap(em, h, old[kk]);
    }
}


// read groups of the rows a workgroup has written into its slot of the scratch table (round 4: every read group's counters may leave
// through the rows, not only the one with the most reads; k_t8_fold runs once per read group present)
struct T8Tags { uint32_t lo = 0, hi = 0; };
__device__ __forceinline__ void t8_tag(T8Tags& t, uint32_t row, uint32_t lane)
{
    if (row < 4u) t.lo |= lane << (8u * row); else t.hi |= lane << (8u * (row - 4u));
}
__device__ __forceinline__ void t8_directory(uint32_t* __restrict__ used /* this slot's BQC_T8_USED words */, uint32_t n_rows, const T8Tags& t)
{
    used[0] = n_rows; used[1] = t.lo; used[2] = t.hi;
}

// The packed counters of a workgroup added to the 64-bit counters in memory by atomics — for read groups that do not own the scratch
// rows.  One BIN per lane: 64 consecutive counters = 512 contiguous bytes per wave instruction (round 4; a dword of four bins per lane
// before: four instructions whose lanes lay 32 bytes apart, the shape the float-atomics section of the MI355X guide calls an order of
// magnitude slower).  Called by every thread of the workgroup, between barriers; leaves the table zeroed.
__device__ __forceinline__ void t8_atomics_out(uint32_t* t8 /* LDS [16384] */, uint64_t* __restrict__ em)
{
    for (uint32_t b = threadIdx.x; b < 65536u; b += blockDim.x) {
        const uint32_t v = (t8[b >> 2] >> (8u * t8_byte(b))) & 0xFFu;
        if (v) gadd(em + b, v);
    }
    block_sync();
    for (uint32_t i = threadIdx.x; i < 16384u; i += blockDim.x) t8[i] = 0;
}

// explicit global-address-space loads (generic/flat loads would count on lgkmcnt and make every LDS wait also wait for
// the prefetch); the 12- and 16-byte loads are unaligned
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ lds_u32* lds_at(uint32_t byte_addr) { return (lds_u32*)(uintptr_t)byte_addr; }


// N dwords from global memory with one or two vector loads; U: any byte address, A: dword-aligned address
typedef u32x3 __attribute__((aligned(1))) u32x3_u;
typedef u32x4 __attribute__((aligned(1))) u32x4_u;
typedef uint32_t __attribute__((aligned(1))) u32_u;
typedef u32x3 __attribute__((aligned(4))) u32x3_a;
typedef u32x4 __attribute__((aligned(4))) u32x4_a;
#define KS_GLOBAL(T, p) (*(const __attribute__((address_space(1))) T*)(uintptr_t)(p))
template <int N> struct GVec;
template <> struct GVec<3> {
    static __device__ __forceinline__ void ldu(uint32_t* d, const uint8_t* p) { const u32x3 v = KS_GLOBAL(u32x3_u, p); d[0] = v.x; d[1] = v.y; d[2] = v.z; }
    static __device__ __forceinline__ void lda(uint32_t* d, const uint8_t* p) { const u32x3 v = KS_GLOBAL(u32x3_a, p); d[0] = v.x; d[1] = v.y; d[2] = v.z; }
};
template <> struct GVec<4> {
    static __device__ __forceinline__ void ldu(uint32_t* d, const uint8_t* p) { const u32x4 v = KS_GLOBAL(u32x4_u, p); d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
    static __device__ __forceinline__ void lda(uint32_t* d, const uint8_t* p) { const u32x4 v = KS_GLOBAL(u32x4_a, p); d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
};
template <> struct GVec<5> {
    static __device__ __forceinline__ void ldu(uint32_t* d, const uint8_t* p) { GVec<4>::ldu(d, p); d[4] = KS_GLOBAL(u32_u, p + 16); }
    static __device__ __forceinline__ void lda(uint32_t* d, const uint8_t* p) { GVec<4>::lda(d, p); d[4] = KS_GLOBAL(uint32_t, p + 16); }
};
template <> struct GVec<8> {
    static __device__ __forceinline__ void ldu(uint32_t* d, const uint8_t* p) { GVec<4>::ldu(d, p); GVec<4>::ldu(d + 4, p + 16); }
};

