// k_short.hip — short-read fast path (reads up to 255 bases, i.e. all Illumina-style data).
//
// One workgroup (16 waves) per CU walks lane-uniform chunks.  Inside a wave, `rpw = 64 / W` reads are processed at
// once; lane (s, w) owns the 16 SEQUENCING CYCLES 16w..16w+15 of read slot s.  Everything is loaded straight from
// global memory in cycle orientation — for a reverse-strand read the lane reads the mirrored 16-base window and
// reverses it in registers — one group ahead of its use, and all per-base work is SWAR on those registers:
//
//   orientation  the BAM base codes are one-hot nibbles (A1 C2 G4 T8), so v_bfrev_b32 yields the reversed AND
//                complemented window in one instruction; qualities are reversed by the byte selector of v_perm_b32;
//                the reference is stored as nibbles  r1 r0 ~r0 ~r1  (r = 2-bit code), which bit reversal also turns
//                into the reverse complement (k_ref_nibbles).
//   cycles       four one-hot planes are added into 4-bit vertical counters in registers (one add per plane and 8
//                cycles), spilled to the LDS cycle tile every 15 groups; quality sums as packed 16-bit fields;
//                per-read sums (quality | N << 16 | GC << 24) by one DPP prefix scan (QualityCheck.hpp:122-166).
//   8-mers       2-bit codes of the lane's 16 cycles in one register (+ the next lane's by DPP); the 16 windows are
//                funnel shifts; counters are 65 536 packed u8 fields in 64 KiB of LDS, incremented by returning
//                atomics whose increment 1 << 8*byte comes from v_alignbyte_b32(1, 1, h); a wrapping field is
//                detected from the returned value and accounted exactly (OverallNumbers.hpp:137-168).
//   triplets     in cycle space as well: read and reference codes interleaved per nibble [r c]; flank equality,
//                N exclusion, position range and quality thresholds are nibble-flag SWAR; the LDS bin index
//                c(j-1) r(j) c(j) r(j+1) is ONE bit-field extract; ks_flush maps the bins (reverse-strand groups:
//                complemented and mirrored) to the reference's layout (TripletCounting.hpp:195-236).  A read's record
//                covers its first CIGAR operation; further match-like operations are triplet-segment entries of the
//                chunk (TripSeg), handled by the same code with cycles / 8-mers switched off.
//   per-read     flag cascade / scalars / per-read histograms (read_stats.h, the body of k_reads) run thread-per-read in
//                phase A, where the read's columns are loaded anyway (bamqualcheck.cpp:318-434).
#include "kernels_common.h"
#include "read_stats.h"

#define KS_WAVES BQC_FAST_WAVES
#define KS_THREADS (KS_WAVES * 64)
#define KS_CT 256                                  // cycles held in LDS ( > BQC_FAST_MAXLEN )
// LDS map (uint32 words)
#define KS_T8    0                                 // 16384: 65536 u8 8-mer counters, four per dword
#define KS_TRIP  (KS_T8 + 16384)                   // [4 groups][256]
#define KS_CYC   (KS_TRIP + 1024)                  // [2 mates][6: A C G T other qual][KS_CT], cycle c at (c & 15) * 16 + (c >> 4):
                                                   // the lanes of a read (c >> 4 = w) hit different banks when they add the same c & 15
#define KS_NC    (KS_CYC + 2 * 6 * KS_CT)          // [2 mates][KS_CT + 1]
#define KS_GC    (KS_NC + 2 * (KS_CT + 1))
#define KS_AQ    (KS_GC + 2 * (KS_CT + 1))         // [2][256]
#define KS_AC    (KS_AQ + 512)
#define KS_LUT   (KS_AC + 512)                     // [17][8] masks for "the first n of 16 cycles": nibbles (2), pad (2), bytes (4)
#define KS_META  (KS_LUT + 17 * 8)                 // per-read records: 64 per wave (the reads the wave is working on)
#define KS_MW    8
#define KS_RS    (KS_META + (KS_WAVES * 64 + 1) * KS_MW)  // (16 waves x 64 records + one dummy record); per-read statistics (read_stats.h)
#define KS_WORDS (KS_RS + RS_WORDS)
#define KS_T8_PERIOD 16                            // chunks between two flushes of the 8-mer counters (~35 counts per bin)
#define KS_BIAS  512u                              // seq / qual offsets in the records are biased so that they stay non-negative

static_assert((KS_LUT % 4) == 0 && (KS_META % 4) == 0, "16-byte alignment of the LDS tables");

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

// The packed 8-mer counters of this workgroup go to its own row of a global scratch table (plain read-modify-write of 16 B
// per thread, no atomics; k_t8_reduce folds the rows into the state vector), or with global atomics when the read group is
// not the one the scratch table is collecting in this launch.  Called workgroup-uniformly between barriers.
__device__ __forceinline__ void t8_flush(uint32_t* lds, uint64_t* __restrict__ em, uint4* __restrict__ row /* nullptr: atomics */)
{
    for (uint32_t i = threadIdx.x; i < 16384; i += blockDim.x) {
        const uint32_t v = lds[KS_T8 + i];
        if (!v) continue;
        lds[KS_T8 + i] = 0;
        if (row) { // row[i] = counts of the dword's bytes 0..3, i.e. of bins 4i + (0, 3, 2, 1)
            uint4 x = row[i];
            x.x += v & 0xFFu; x.y += (v >> 8) & 0xFFu; x.z += (v >> 16) & 0xFFu; x.w += v >> 24;
            row[i] = x;
        } else {
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b)
                if ((v >> (8u * b)) & 0xFFu) gadd(em + 4u * i + ((4u - b) & 3u), (v >> (8u * b)) & 0xFFu);
        }
    }
}

__device__ __forceinline__ void ks_flush(uint32_t* lds, const StateLayout& sl, uint64_t* __restrict__ state, uint32_t lane, uint4* __restrict__ row)
{
    const uint64_t lb = sl.lane_base(lane);
    t8_flush(lds, state + lb + sl.o_eightmer, row);
    for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) { // bin = c(j-1) r(j) c(j) r(j+1) of group i >> 8, in cycle space
        const uint32_t v = lds[KS_TRIP + i];
        if (!v) continue;
        lds[KS_TRIP + i] = 0;
        const uint32_t grp = i >> 8, f3 = (i >> 6) & 3u, f2 = (i >> 4) & 3u, f1 = (i >> 2) & 3u, f0 = i & 3u;
        uint32_t ctx, base;
        if (grp < 2u) { ctx = (f3 << 4) | (f2 << 2) | f0; base = f1; }                               // forward: as is
        else { ctx = ((3u - f0) << 4) | ((3u - f2) << 2) | (3u - f3); base = 3u - f1; }                // reverse: complement, mirrored
        gadd(state + lb + sl.o_triplet + ctx * 16u + grp * 4u + base, v);
    }
    for (uint32_t i = threadIdx.x; i < 2 * 6 * KS_CT; i += blockDim.x) { // per-cycle counters [2 mates][A C G T other qual][KS_CT]
        const uint32_t v = lds[KS_CYC + i];
        if (!v) continue;
        lds[KS_CYC + i] = 0;
        const uint32_t m = i / (6 * KS_CT), c = (i / KS_CT) % 6, jj = i % KS_CT, j = (jj & 15u) * 16u + (jj >> 4);
        if (j < sl.lcap) gadd(state + sl.mate_base(lane, m) + (c < 5 ? sl.m_dnacount + c * sl.lcap : sl.m_qualcount) + j, v);
    }
    for (uint32_t i = threadIdx.x; i < 2 * (KS_CT + 1); i += blockDim.x) {
        const uint32_t m = i / (KS_CT + 1), j = i % (KS_CT + 1);
        const uint64_t mb = sl.mate_base(lane, m);
        uint32_t v = lds[KS_NC + i];
        if (v && j <= sl.lcap) gadd(state + mb + sl.m_ncount + j, v);
        lds[KS_NC + i] = 0;
        v = lds[KS_GC + i];
        if (v && j <= sl.lcap) gadd(state + mb + sl.m_gccount + j, v);
        lds[KS_GC + i] = 0;
    }
    for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x) {
        const uint64_t mb = sl.mate_base(lane, i >> 8);
        uint32_t v = lds[KS_AQ + i];
        if (v) gadd(state + mb + sl.m_avgqual + (i & 255), v);
        lds[KS_AQ + i] = 0;
        v = lds[KS_AC + i];
        if (v) gadd(state + mb + sl.m_avgceil + (i & 255), v);
        lds[KS_AC + i] = 0;
    }
}

// Per-cycle accumulators of one lane (its 16 cycles), for the reads of ONE mate.
struct CycAcc {
    uint32_t l1[2][4];   // [half][A C G T]: 4-bit vertical counters, nibble t <-> cycle 8*half + 7 - t
    uint32_t qo[4], qe[4]; // quality sums of dword d (cycles 4d..4d+3), 16-bit fields: qo = cycles 4d | 4d+2, qe = 4d+1 | 4d+3
};
__device__ __forceinline__ void cyc_zero(CycAcc& A)
{
#pragma unroll
    for (int p = 0; p < 4; ++p) { A.l1[0][p] = 0; A.l1[1][p] = 0; A.qo[p] = 0; A.qe[p] = 0; }
}
// The counters go to the LDS cycle tile through real functions with by-value arguments (registers, no scratch): rare.
__device__ __noinline__ void cyc_spill_lds(uint32_t a0, uint32_t a1, uint32_t c0, uint32_t c1, uint32_t g0, uint32_t g1, uint32_t t0, uint32_t t1,
                                           uint32_t* base /* lds + KS_CYC + mate * 6 * KS_CT + w */)
{
    const uint32_t v[8] = {a0, a1, c0, c1, g0, g1, t0, t1};
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < 8; ++t) atomicAdd(base + p * KS_CT + 16 * (8 * h + 7 - t), (v[2 * p + h] >> (4 * t)) & 15u);
}
__device__ __noinline__ void cyc_qflush_lds(uint32_t o0, uint32_t e0, uint32_t o1, uint32_t e1, uint32_t o2, uint32_t e2, uint32_t o3, uint32_t e3,
                                            uint32_t* base /* lds + KS_CYC + (mate * 6 + 5) * KS_CT + w */)
{
    const uint32_t vo[4] = {o0, o1, o2, o3}, ve[4] = {e0, e1, e2, e3};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        atomicAdd(base + 16 * (4 * d + 0), vo[d] >> 16);
        atomicAdd(base + 16 * (4 * d + 1), ve[d] >> 16);
        atomicAdd(base + 16 * (4 * d + 2), vo[d] & 0xFFFFu);
        atomicAdd(base + 16 * (4 * d + 3), ve[d] & 0xFFFFu);
    }
}
__device__ __forceinline__ void cyc_spill(CycAcc& A, uint32_t* lds, uint32_t mate, uint32_t w)
{
    cyc_spill_lds(A.l1[0][0], A.l1[1][0], A.l1[0][1], A.l1[1][1], A.l1[0][2], A.l1[1][2], A.l1[0][3], A.l1[1][3], lds + KS_CYC + mate * 6 * KS_CT + w);
#pragma unroll
    for (int p = 0; p < 4; ++p) { A.l1[0][p] = 0; A.l1[1][p] = 0; }
}
__device__ __forceinline__ void cyc_qflush(CycAcc& A, uint32_t* lds, uint32_t mate, uint32_t w)
{
    cyc_qflush_lds(A.qo[0], A.qe[0], A.qo[1], A.qe[1], A.qo[2], A.qe[2], A.qo[3], A.qe[3], lds + KS_CYC + (mate * 6 + 5) * KS_CT + w);
#pragma unroll
    for (int p = 0; p < 4; ++p) { A.qo[p] = 0; A.qe[p] = 0; }
}
__device__ __forceinline__ void cyc_add(CycAcc& A, const Planes& P0, const Planes& P1, const uint32_t (&Q)[4])
{
    A.l1[0][0] += P0.a; A.l1[0][1] += P0.c; A.l1[0][2] += P0.g; A.l1[0][3] += P0.t;
    A.l1[1][0] += P1.a; A.l1[1][1] += P1.c; A.l1[1][2] += P1.g; A.l1[1][3] += P1.t;
#pragma unroll
    for (int d = 0; d < 4; ++d) { A.qe[d] += Q[d] & 0x00FF00FFu; A.qo[d] += (Q[d] >> 8) & 0x00FF00FFu; }
}

// explicit global-address-space loads (generic/flat loads would count on lgkmcnt and make every LDS wait also wait for
// the prefetch); the 12- and 16-byte loads are unaligned
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x3 __attribute__((aligned(1))) u32x3_u;
typedef u32x4 __attribute__((aligned(1))) u32x4_u;
typedef u32x3 __attribute__((aligned(4))) u32x3_a;
typedef const __attribute__((address_space(1))) uint8_t* g_u8p;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ lds_u32* lds_at(uint32_t byte_addr) { return (lds_u32*)(uintptr_t)byte_addr; }

// record of one read in the chunk's LDS table (written by phase A)
#define KM_PRIM   0x10000u   // primary record with first/last flag: reaches get_count / count8mers
#define KM_TRIP   0x20000u   // triplets are evaluated for the position range in word 5 (needs a loaded reference)
#define KM_SEG    0x40000u   // triplet-segment entry: the read's bases / qualities for triplets only (no KM_PRIM)
// word 0: BAM flag (low 16 bits) | KM_* | L << 20 (L = 0 unless KM_PRIM)
//      1: seq byte offset of the window of lane w = 0 (+ KS_BIAS)      -> after the group was computed: packed sums before the read
//      2: qual byte offset of the window of lane w = 0 (+ KS_BIAS)     ->                               packed sums after the read
//      3: reference nibble index of that window (pos + o0 + 15; the table has 16 pad nibbles in front)
//      4: last loadable dword index of the contig's table     5: ja | jb << 8 | seq funnel shift << 16
//      6-7: pointer to the contig's nibble table
// Lane w of a forward read loads the window starting at base 16 w, of a reverse read the one starting at L - 16 - 16 w.

struct Pre { u32x3 s; u32x4 q; u32x3 e; uint32_t m0, pp, w5; }; // raw data of the NEXT group, in flight while the current one is computed

// Issue the global loads of one group for this lane.  Branch-free: records of non-primary reads, padding entries and the
// dummy record used by lanes past the end of the chunk carry L = 0 and offsets / pointers that are safe to load from
// (the buffers are padded on both sides), so every lane always loads; what must not be used is masked when consumed.
__device__ __forceinline__ Pre ks_prefetch(const uint32_t* rec, uint32_t w, g_u8p seqb, g_u8p qualb)
{
    const uint4 ma = *(const uint4*)rec, mb = *(const uint4*)(rec + 4);
    const int32_t sw = (ma.x & 0x10u) ? -(int32_t)w : (int32_t)w;
    Pre P;
    P.m0 = ma.x;
    P.w5 = mb.y;
    P.pp = ma.w + (uint32_t)(16 * sw);
    P.s = *(const __attribute__((address_space(1))) u32x3_u*)(seqb + (uint32_t)(ma.y + (uint32_t)(8 * sw)));
    P.q = *(const __attribute__((address_space(1))) u32x4_u*)(qualb + (uint32_t)(ma.z + (uint32_t)(16 * sw)));
    const uint32_t* rn = (const uint32_t*)(uintptr_t)((uint64_t)mb.z | ((uint64_t)mb.w << 32));
    const int32_t di = min(max((int32_t)P.pp >> 3, 0), (int32_t)mb.x);
    P.e = *(const __attribute__((address_space(1))) u32x3_a*)(uintptr_t)(rn + di);
    return P;
}

__global__ __launch_bounds__(KS_THREADS) void k_short(DevBatch b, StateLayout sl, uint64_t* __restrict__ state, DevRefs refs,
                                                         uint32_t* __restrict__ err, uint32_t parts, uint4* __restrict__ t8rows,
                                                         uint32_t t8_lane, uint32_t t8_period)
{
    extern __shared__ uint32_t lds[];
    if ((uint32_t)(uintptr_t)(lds_u32*)lds != 0u) { // the 8-mer atomics address LDS directly (KS_T8 at LDS address 0)
        if (threadIdx.x == 0) atomicOr(err, BQC_DEVERR_INTERNAL);
        return;
    }
    for (uint32_t i = threadIdx.x; i < KS_WORDS; i += blockDim.x) lds[i] = 0;
    block_sync();
    const uint32_t M = 0x11111111u;
    const uint32_t ln = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t W = b.fast_w, rpw = 64u / W;
    const uint32_t slot = ln / W, w = ln % W, w16 = 16u * w;
    const bool lane_used = slot < rpw;
    const bool last_w = (w + 1u >= W) || ln == 63u;   // the next lane belongs to another read
    const uint32_t tile_cap = rpw * (64u / rpw);        // reads a wave takes at a time: one record per lane, whole groups
    uint32_t* WM = lds + KS_META + wave * 64u * KS_MW;   // this wave's records
    const uint32_t* DUMMY = lds + KS_META + KS_WAVES * 64u * KS_MW; // record for lanes behind the last slot
    const uint32_t* LUT = lds + KS_LUT;
    if (threadIdx.x < 17u) { // masks for "the first n of the lane's 16 cycles"
        const uint32_t n = threadIdx.x, n0 = n < 8u ? n : 8u, n1 = n - n0;
        uint32_t* e = lds + KS_LUT + 8 * n;
        e[0] = n0 ? 0xFFFFFFFFu << (4u * (8u - n0)) : 0u;
        e[1] = n1 ? 0xFFFFFFFFu << (4u * (8u - n1)) : 0u;
        for (uint32_t d = 0; d < 4; ++d) {
            const uint32_t v = n > 4u * d ? (n - 4u * d < 4u ? n - 4u * d : 4u) : 0u;
            e[4 + d] = v ? 0xFFFFFFFFu << (8u * (4u - v)) : 0u; // qualities are kept big-endian: first cycle in the top byte
        }
    }
    if (threadIdx.x < KS_MW) { // dummy record: L = 0, loadable offsets / pointer
        const uint32_t t = threadIdx.x;
        lds[KS_META + KS_WAVES * 64u * KS_MW + t] = t == 1 || t == 2 ? KS_BIAS : t == 3 ? 15u : t == 5 ? (24u << 16) : t == 6 ? (uint32_t)(uintptr_t)state
                                   : t == 7 ? (uint32_t)((uintptr_t)state >> 32) : 0u;
    }
    block_sync();
    CycAcc A; // per-cycle accumulators of this lane: reads of ONE mate (fixed per slot, see the chunk layout below)
    cyc_zero(A);
    const uint32_t mate = slot < (rpw + 1u) / 2u ? 0u : 1u;
    uint32_t n1 = 0, n2 = 0; // groups since the last counter spill / quality flush (wave-uniform)
    uint32_t cur_lane = 0xFFFFFFFFu, since_t8 = 0;

    const g_u8p g_seq = (g_u8p)(uintptr_t)(b.seq - KS_BIAS);
    const g_u8p g_qual = (g_u8p)(uintptr_t)(b.qual - KS_BIAS);
    const uint32_t n_chunks = b.n_chunks_fast;
    for (uint32_t ci = blockIdx.x;; ci += gridDim.x) { // one extra pass at the end flushes the last lane (single call site)
        const bool done = ci >= n_chunks;
        Chunk ch{0, 0, 0xFFFFFFFFu, 0, 0, 0, 0, 0};
        if (!done) ch = b.chunks_fast[ci];
        if (ch.lane != cur_lane) { // block-uniform
            if (cur_lane != 0xFFFFFFFFu) {
                if (lane_used) { cyc_spill(A, lds, mate, w); cyc_qflush(A, lds, mate, w); }
                n1 = n2 = 0;
                block_sync();
                ks_flush(lds, sl, state, cur_lane, cur_lane == t8_lane ? t8rows + (size_t)blockIdx.x * 16384u : nullptr);
                rs_flush(lds + KS_RS, sl, state, cur_lane);
            }
            cur_lane = ch.lane;
            since_t8 = 0;
        }
        if (done) break;
        if (++since_t8 > t8_period) { // keep the packed u8 counters small, so that the wrap check below stays on its fast path
            since_t8 = 1;
            block_sync();
            t8_flush(lds, state + sl.lane_base(cur_lane) + sl.o_eightmer, cur_lane == t8_lane ? t8rows + (size_t)blockIdx.x * 16384u : nullptr);
            block_sync();
        }
        const uint64_t lb = sl.lane_base(cur_lane);
        uint64_t* em = state + lb + sl.o_eightmer;
        // The chunk is [read groups | triplet-segment groups]; in a read group the first h0 slots hold first-mate reads and the
        // others second-mate reads (or null entries), so every lane only ever sees reads of ONE mate (`mate`) and keeps that
        // mate's per-cycle counters in registers.  Wave v takes the read tiles v, v + 16, ... and the segment tiles starting at
        // a wave that rotates with the chunk.
        const uint32_t n_seg = ch.count - ch.aux;
        const uint32_t my_tiles = ch.aux > wave * tile_cap ? (ch.aux - wave * tile_cap + KS_WAVES * tile_cap - 1u) / (KS_WAVES * tile_cap) : 0u;
        const uint32_t sw0 = (wave + KS_WAVES - ci % KS_WAVES) % KS_WAVES;
        const uint32_t seg_tiles = n_seg > sw0 * tile_cap ? (n_seg - sw0 * tile_cap + KS_WAVES * tile_cap - 1u) / (KS_WAVES * tile_cap) : 0u;
        for (uint32_t tix = 0; tix < my_tiles + seg_tiles; ++tix) {
        const bool seg_tile = tix >= my_tiles; // wave-uniform
        const uint32_t tb = seg_tile ? (sw0 + (tix - my_tiles) * KS_WAVES) * tile_cap : (wave + tix * KS_WAVES) * tile_cap;
        const uint32_t p_first = seg_tile ? ch.aux : 0u, p_n = seg_tile ? n_seg : ch.aux;
        const uint32_t tn = min(tile_cap, p_n - tb); // entries of this wave's tile (a multiple of rpw)
        // ---- phase A: lane per read — per-read statistics, and the read's record for phase B into LDS
        __builtin_amdgcn_s_setprio(3); // few instructions between long waits: let them issue ahead of the other waves' phase B
        uint32_t r;
        bool stat; // this lane's entry is a read (per-read statistics below)
        {
            const uint32_t t = ch.first + p_first + tb + ln;
            r = ln < tn ? (b.perm ? b.perm[t] : t) : 0xFFFFFFFFu;
            const bool live = r != 0xFFFFFFFFu;                // not a padding entry
            const bool seg = live && (r & BQC_ENTRY_SEG);      // triplet segment of a read (the read itself is another entry)
            stat = live && !seg;
            TripSeg sg{0, 0, 0, 0};
            if (seg) { sg = b.segs[r & ~BQC_ENTRY_SEG]; r = sg.r; }
            uint4 R0 = make_uint4(0u, KS_BIAS, KS_BIAS, 15u);
            uint4 R1 = make_uint4(0u, 24u << 16, (uint32_t)(uintptr_t)state, (uint32_t)((uintptr_t)state >> 32));
            if (live) {
                const uint32_t fl = b.flag[r];
                R0.x = fl & 0xFFFFu;
                if (!(fl & 0x900u) && (fl & 0xC0u)) { // reaches get_count / count8mers
                    const uint32_t L = b.l_seq[r];
                    const bool rc = fl & 0x10u, noq = fl & BQC_FLAG_NO_QUAL;
                    const int32_t o0 = rc ? (int32_t)L - 16 : 0;
                    R0.x |= (seg ? KM_SEG : KM_PRIM) | (L << 20);
                    R0.y = b.seq_off[r] + (uint32_t)((o0 - 1) >> 1) + KS_BIAS; // the window is loaded from one byte (odd o0: one nibble) earlier
                    R0.z = (noq ? 0u : b.qual_off[r]) + (uint32_t)o0 + KS_BIAS;
                    R1.y = ((o0 & 1) ? 28u : 24u) << 16;
                    const int32_t rid = b.rid[r];
                    if ((fl & BQC_FLAG_TRIPLET) && b.n_cigar[r] >= 1 && L >= 3 && !noq && rid >= 0 && (uint32_t)rid < refs.n_refs &&
                        refs.refn[rid] != nullptr) {
                        // read positions ia <= i < ib with chromPos = pos + i: inside the read (1 .. L-2), inside the first CIGAR
                        // operation (assumed match-like, TripletCounting.hpp:203) or the segment, context pos+i-1 .. pos+i+1 inside the contig
                        const int64_t pos = seg ? (int64_t)sg.posv : (int64_t)b.pos[r];
                        const int64_t reflen = (int64_t)refs.len[rid];
                        int64_t ia = 1, ib = (int64_t)L - 1;
                        if (seg) { ia = sg.range & 0xFFu; ib = (sg.range >> 8) & 0xFFu; }
                        else {
                            const uint32_t n0 = b.cigar[b.cigar_off[r]] >> 4;
                            if (n0 != 0u && (int64_t)n0 < ib) ib = n0;
                        }
                        if (1 - pos > ia) ia = 1 - pos;
                        if (reflen - 1 - pos < ib) ib = reflen - 1 - pos;
                        if (ib > ia && pos > -(1 << 30) && pos < (1 << 30)) {
                            const uint32_t ja = rc ? L - (uint32_t)ib : (uint32_t)ia, jb = rc ? L - (uint32_t)ia : (uint32_t)ib; // the same range in cycles
                            const uint64_t nd8 = (uint64_t)(reflen + 7) >> 3;
                            R0.x |= KM_TRIP;
                            R0.w = (uint32_t)((int32_t)pos + o0 + 15);
                            R1.x = (uint32_t)(nd8 + 1);
                            R1.y |= ja | (jb << 8);
                            const uint64_t rn = (uint64_t)(uintptr_t)refs.refn[rid];
                            R1.z = (uint32_t)rn; R1.w = (uint32_t)(rn >> 32);
                        }
                    }
                }
            }
            uint4* Mr = (uint4*)(WM + ln * KS_MW);
            Mr[0] = R0; Mr[1] = R1;
        }
        // LDS operations of one wave execute in order, so the records are visible to the reads below; only the COMPILER must not
        // reorder them
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // ---- phase B: groups of rpw reads; the next group's data is loaded while this one is processed
        const uint32_t n_groups = tn / rpw;
        Pre cur = ks_prefetch(lane_used ? WM + slot * KS_MW : DUMMY, w, g_seq, g_qual);
        // per-read statistics of the tile's reads, while the first group's data is on its way
        if ((parts & 8u) && __ballot(stat)) read_stats(b, sl, state, refs, err, lds + KS_RS, stat ? r : 0u, stat, stat);
        __builtin_amdgcn_s_setprio(0);
        for (uint32_t g = 0; g < n_groups; ++g) {
            const uint32_t k = g * rpw + slot; // this lane's record (lanes behind the last slot: unused)
            const uint32_t m0 = cur.m0, w5 = cur.w5, L = (m0 >> 20) & 0xFFu; // L = 0 unless the record reaches get_count
            const bool rc = m0 & 0x10u;
            const bool segg = (uint32_t)__builtin_amdgcn_readfirstlane((int)m0) & KM_SEG; // a group of triplet segments: triplets only
            const uint32_t nv = (uint32_t)min(max((int32_t)L - (int32_t)w16, 0), 16);   // valid cycles of this lane
            const uint32_t nvq = (m0 & BQC_FLAG_NO_QUAL) ? 0u : nv;
            const uint2 xm = *(const uint2*)(LUT + 8u * nv);
            const uint4 qm = *(const uint4*)(LUT + 8u * nvq + 4u);
            // ---------------- the lane's 16 cycles: one-hot base nibbles X0 | X1 (first cycle in the top nibble) ...
            uint32_t X0, X1;
            {
                const uint32_t sh = w5 >> 16; // 24 / 28: the loaded bytes start 2 / 1 nibbles before the window
                const uint32_t b0 = bswap32(cur.s.x), b1 = bswap32(cur.s.y), b2 = bswap32(cur.s.z);
                const uint32_t F0 = alignbit(b0, b1, sh), F1 = alignbit(b1, b2, sh);
                X0 = (rc ? __brev(F1) : F0) & xm.x; // bit reversal = reversed base order and complemented one-hot codes (IUPAC too)
                X1 = (rc ? __brev(F0) : F1) & xm.y;
            }
            // ... and qualities Q[d] = cycles 4d..4d+3, first cycle in the top byte
            uint32_t Q[4];
            {
                const uint32_t sel = rc ? 0x07060504u : 0x00010203u; // reverse read: dwords in reverse order; forward read: bytes swapped
                Q[0] = vperm(cur.q.w, cur.q.x, sel) & qm.x;
                Q[1] = vperm(cur.q.z, cur.q.y, sel) & qm.y;
                Q[2] = vperm(cur.q.y, cur.q.z, sel) & qm.z;
                Q[3] = vperm(cur.q.x, cur.q.w, sel) & qm.w;
                const uint32_t hi = (Q[0] | Q[1] | Q[2] | Q[3]) & 0x80808080u;
                if (hi) { // some Phred >= 128: check the 222 limit precisely
                    bool bad = false;
#pragma unroll
                    for (int d = 0; d < 4; ++d)
#pragma unroll
                        for (int k8 = 0; k8 < 4; ++k8) bad |= ((Q[d] >> (8 * k8)) & 0xFFu) > 222u;
                    if (bad) atomicOr(err, BQC_DEVERR_QUAL);
                }
            }
            // ... and the reference window as nibbles r1 r0 ~r0 ~r1: bit reversal = reverse complement here too
            uint32_t E0, E1;
            {
                const uint32_t sh = 28u - 4u * (cur.pp & 7u);
                const uint32_t F0 = alignbit(cur.e.x, cur.e.y, sh), F1 = alignbit(cur.e.y, cur.e.z, sh);
                E0 = rc ? __brev(F1) : F0;
                E1 = rc ? __brev(F0) : F1;
            }
            { // the raw registers are free again: issue the loads of this wave's next group
                cur = ks_prefetch(lane_used && g + 1u < n_groups ? WM + (k + rpw) * KS_MW : DUMMY, w, g_seq, g_qual);
            }
            const Planes P0 = planes_of(X0), P1 = planes_of(X1);
            // ---- per-cycle counters (the group's mate selects the register set: wave-uniform branch)
            if ((parts & 1u) && !segg) {
                cyc_add(A, P0, P1, Q);
                if (++n1 == 15u) { if (lane_used) cyc_spill(A, lds, mate, w); n1 = 0; }
                if (++n2 == 255u) { if (lane_used) cyc_qflush(A, lds, mate, w); n2 = 0; }
                // cycles holding anything but A/C/G/T (Dna5 'N' bin) are rare: counted directly
                uint32_t o0 = xm.x & M & ~P0.oh, o1 = xm.y & M & ~P1.oh;
                if (o0 | o1) {
                    uint32_t* ob = lds + KS_CYC + (mate * 6 + 4) * KS_CT + w;
                    while (o0) { const uint32_t bit = (uint32_t)__ffs((int)o0) - 1u; o0 &= o0 - 1u; atomicAdd(ob + 16u * (7u - (bit >> 2)), 1u); }
                    while (o1) { const uint32_t bit = (uint32_t)__ffs((int)o1) - 1u; o1 &= o1 - 1u; atomicAdd(ob + 16u * (15u - (bit >> 2)), 1u); }
                }
                // per-read sums: quality | N << 16 | GC << 24 (L <= 255), one wave-wide prefix scan; the lane at the start of
                // a slot leaves the running sum before its read in record word 1 and, for the previous slot, after it in word 2
                uint32_t v = __builtin_amdgcn_sad_u8(Q[0], 0u, 0u);
                v = __builtin_amdgcn_sad_u8(Q[1], 0u, v); v = __builtin_amdgcn_sad_u8(Q[2], 0u, v); v = __builtin_amdgcn_sad_u8(Q[3], 0u, v);
                v |= ((uint32_t)__popc(P0.n) + (uint32_t)__popc(P1.n)) << 16;
                v += ((uint32_t)__popc(P0.c | P0.g) + (uint32_t)__popc(P1.c | P1.g)) << 24;
                const uint32_t si = wave_scan_incl(v);
                const uint32_t bs = lane_prev(si);
                const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)si, 63);
                if (lane_used && w == 0u) {
                    WM[k * KS_MW + 1] = bs;
                    if (slot) WM[(k - 1u) * KS_MW + 2] = bs;
                    if (slot + 1u == rpw) WM[k * KS_MW + 2] = tot; // lanes behind the last slot hold only zeros
                }
            }
            // ---- 2-bit codes per nibble; non-ACGT -> A (char -> Dna after the reverse complement)
            const uint32_t cn0 = (P0.c | P0.t) | ((P0.g | P0.t) << 1), cn1 = (P1.c | P1.t) | ((P1.g | P1.t) << 1);
            // literal N or past the end of the read: blocks 8-mer windows and triplet flanks
            const uint32_t nb0 = P0.n | (~xm.x & M), nb1 = P1.n | (~xm.y & M);
            // ---- 8-mers: windows starting at the lane's 16 cycles
            if ((parts & 2u) && !segg) {
                const uint32_t c32 = vperm(squeeze2(cn0), squeeze2(cn1), 0x05040100u); // cycle 16w in the top two bits
                const uint32_t cx = lane_next(c32);
                uint32_t nbx = lane_next(nb0);
                if (last_w) nbx = M;
                // a window is blocked when any of its 8 cycles is: OR-smear over the next 7 positions of the flag stream nb0 nb1 nbx
                uint32_t f0, f1;
                {
                    const uint32_t a0 = nb0 | alignbit(nb0, nb1, 28), a1 = nb1 | alignbit(nb1, nbx, 28), ax = nbx | (nbx << 4);
                    const uint32_t b0 = a0 | alignbit(a0, a1, 24), b1 = a1 | alignbit(a1, ax, 24), bx = ax | (ax << 8);
                    f0 = ~(b0 | alignbit(b0, b1, 16)); // nibble LSB set <=> the window starting there is counted
                    f1 = ~(b1 | alignbit(b1, bx, 16));
                }
                // Branch-free: every lane issues all 16 returning atomics; a blocked window (and every window of a lane without
                // valid cycles) adds 0.  The address is the LDS byte address itself: KS_T8 = 0 and the dynamic LDS block starts
                // at 0 (checked at kernel entry), which saves the base addition per window.
                if (nv) { // all 16 atomics are issued before the first returned value is looked at
                    uint32_t old[16];
#pragma unroll
                    for (int kw = 0; kw < 16; ++kw) {
                        const uint32_t h = kw < 8 ? c32 >> (16 - 2 * kw) : kw == 8 ? c32 : alignbit(c32, cx, 48 - 2 * kw); // window in the low 16 bits
                        const uint32_t one = bfe(kw < 8 ? f0 : f1, 28 - 4 * (kw & 7), 1);
                        old[kw] = __hip_atomic_fetch_add(lds_at(h & 0xFFFCu), alignbyte(one, one, h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    uint32_t hot0 = 0, hot1 = 0;
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) { hot0 |= old[kk]; hot1 |= old[8 + kk]; }
                    if (hot0 & 0x80808080u) // some counter of a touched dword is >= 128: look precisely (rare)
                        t8_check<0>(em, c32, cx, f0, old[0], old[1], old[2], old[3], old[4], old[5], old[6], old[7]);
                    if (hot1 & 0x80808080u)
                        t8_check<8>(em, c32, cx, f1, old[8], old[9], old[10], old[11], old[12], old[13], old[14], old[15]);
                }
            }
            // ---- triplets in cycle space (single-operation CIGAR: chromPos = pos + i)
            if ((parts & 4u) && __ballot(m0 & KM_TRIP)) {
                // The reference converts the BAM-orientation char to Dna (anything but A/C/G/T -> A); in the cycle space of a
                // reverse read that 'A' is the complement's code 3.  (For 8-mers the conversion comes after the reverse complement.)
                const uint32_t x0 = ~P0.oh & M, x1 = ~P1.oh & M, rcm = rc ? 0x33333333u : 0u;
                const uint32_t ct0 = cn0 | ((x0 | (x0 << 1)) & rcm), ct1 = cn1 | ((x1 | (x1 << 1)) & rcm);
                const uint32_t I0 = (E0 & 0xCCCCCCCCu) | ct0, I1 = (E1 & 0xCCCCCCCCu) | ct1; // nibble = [r c]
                const uint32_t t0 = I0 ^ (I0 >> 2), t1 = I1 ^ (I1 >> 2);
                const uint32_t bad0 = ((t0 | (t0 >> 1)) & M) | nb0, bad1 = ((t1 | (t1 >> 1)) & M) | nb1; // as a flank: mismatch or N
                const uint32_t badp = lane_prev(bad1), badn = lane_next(bad0); // (cycle 0 / L-1 are never evaluated)
                const uint32_t fl0 = alignbit(badp, bad0, 4) | alignbit(bad0, bad1, 28);
                const uint32_t fl1 = alignbit(bad0, bad1, 4) | alignbit(bad1, badn, 28);
                // position range [ja, jb) of the read in cycles
                const uint32_t ja = (uint32_t)min(max((int32_t)(w5 & 0xFFu) - (int32_t)w16, 0), 16);
                const uint32_t jb = (uint32_t)min(max((int32_t)((w5 >> 8) & 0xFFu) - (int32_t)w16, 0), 16);
                const uint2 pa = *(const uint2*)(LUT + 8u * ja), pb = *(const uint2*)(LUT + 8u * jb);
                // quality 20..94 <=> (signed char)(q + 33) >= '5'; flags at the byte MSBs, then compressed to nibble LSBs
                uint32_t qf[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint32_t x = Q[d] & 0x7F7F7F7Fu;
                    const uint32_t f = (x + 0x6C6C6C6Cu) & ~(x + 0x21212121u) & ~Q[d]; // >= 20, not >= 95, not >= 128
                    uint32_t y = (f >> 3) & 0x10101010u;              // cycle 4d+k: bit 28 - 8k
                    y = (y | (y << 4)) & 0xFF00FF00u;                  // 28 24 | 12 8
                    qf[d] = y | (y << 8);                              // top half: 28 24 20 16
                }
                const uint32_t ok0 = P0.oh & ~fl0 & pb.x & ~pa.x & vperm(qf[0], qf[1], 0x07060302u);
                const uint32_t ok1 = P1.oh & ~fl1 & pb.y & ~pa.y & vperm(qf[2], qf[3], 0x07060302u);
                const uint32_t Ip = lane_prev(I1), In = lane_next(I0); // (cross-lane: outside the divergent branch)
                if (ok0 | ok1) {
                    uint32_t* tbin = lds + KS_TRIP + ((rc ? 2u : 0u) + ((m0 & 0x40u) ? 0u : 1u)) * 256u; // fwd1st fwd2nd rev1st rev2nd
                    const uint32_t SA0 = alignbit(Ip, I0, 6), SB0 = alignbit(I0, I1, 22), SA1 = alignbit(I0, I1, 6), SB1 = alignbit(I1, In, 22);
#pragma unroll
                    for (int t = 0; t < 8; ++t) { // bin = c(j-1) r(j) c(j) r(j+1): 8 contiguous bits of the [r c] stream
                        const uint32_t i0 = t < 6 ? bfe(SA0, 20 - 4 * t, 8) : bfe(SB0, 12 - 4 * (t - 6), 8);
                        if (ok0 & (1u << (28 - 4 * t))) atomicAdd(tbin + i0, 1u);
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const uint32_t i1 = t < 6 ? bfe(SA1, 20 - 4 * t, 8) : bfe(SB1, 12 - 4 * (t - 6), 8);
                        if (ok1 & (1u << (28 - 4 * t))) atomicAdd(tbin + i1, 1u);
                    }
                }
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // ---- phase C: lane per read — per-read histograms from the sums left in the records (QualityCheck.hpp:157-165)
        if ((parts & 1u) && ln < tn) {
            const uint32_t m0 = WM[ln * KS_MW];
            if (m0 & KM_PRIM) {
                const uint32_t L = (m0 >> 20) & 0xFFu, sum = WM[ln * KS_MW + 2] - WM[ln * KS_MW + 1];
                const uint32_t qs = sum & 0xFFFFu, nN = (sum >> 16) & 0xFFu, nGC = sum >> 24, rm = (m0 & 0x40u) ? 0u : 1u;
                atomicAdd(&lds[KS_NC + rm * (KS_CT + 1) + nN], 1u);
                atomicAdd(&lds[KS_GC + rm * (KS_CT + 1) + nGC], 1u);
                if (L > 0) { // round-half-away and ceil of qs/L, exact: small_div corrects the reciprocal estimate
                    atomicAdd(&lds[KS_AQ + rm * 256 + (small_div(2u * qs + L, 2u * L) & 255u)], 1u);
                    atomicAdd(&lds[KS_AC + rm * 256 + (small_div(qs + L - 1u, L) & 255u)], 1u);
                }
            }
        }
        asm volatile("" ::: "memory"); // the records are rewritten by the next tile
        __builtin_amdgcn_wave_barrier();
        } // tiles
    }
}


// Dna5 bytes -> reference table of the fast path: one nibble  r1 r0 ~r0 ~r1  per base (r = Dna5 code & 3, i.e. N -> A like
// Dna5 -> Dna), 8 bases per dword, first base in the top nibble; bases 0.. start at dword 2 (two zero dwords in front, at
// least two behind): nd8 + 4 dwords for nd8 = ceil(len / 8).  Bit-reversing a dword yields the reverse complement.
__global__ void k_ref_nibbles(const uint8_t* __restrict__ dna5, uint64_t len, uint32_t* __restrict__ out, uint64_t nd8)
{
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nd8 + 4) return;
    uint32_t v = 0;
    if (d >= 2)
        for (uint32_t t = 0; t < 8; ++t) {
            const uint64_t p = (d - 2) * 8 + t;
            if (p < len) {
                const uint32_t r = dna5[p] & 3u;
                v |= ((r << 2) | ((~r & 1u) << 1) | ((~r >> 1) & 1u)) << (28u - 4u * t);
            }
        }
    out[d] = v;
}

extern "C" void bqc_launch_ref_nibbles(const uint8_t* dna5, uint64_t len, uint32_t* out, uint64_t nd8, hipStream_t s)
{
    hipLaunchKernelGGL(k_ref_nibbles, dim3((uint32_t)((nd8 + 4 + 255) / 256)), dim3(256), 0, s, dna5, len, out, nd8);
}

// BQC_SHORT_PARTS: ablation switch for profiling (1 cycles + per-read sums, 2 8-mers, 4 triplets, 8 per-read statistics;
// without 8 the caller runs k_reads over the fast chunks instead)
extern "C" uint32_t bqc_short_parts()
{
    static uint32_t parts = 0xFFFFFFFFu;
    if (parts == 0xFFFFFFFFu) { const char* e = getenv("BQC_SHORT_PARTS"); parts = e ? (uint32_t)atoi(e) : 15u; }
    return parts;
}

extern "C" hipError_t bqc_short_init()
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_short), hipFuncAttributeMaxDynamicSharedMemorySize, KS_WORDS * 4);
}

// Fold the per-workgroup 8-mer rows into the state vector and clear them: thread per LDS dword (4 bins) and slice of 16 rows.
__global__ __launch_bounds__(256) void k_t8_reduce(uint4* __restrict__ rows, uint32_t n_rows, uint64_t* __restrict__ em)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; // < 16384
    uint64_t A = 0, B = 0, C = 0, D = 0;
    const uint32_t r0 = blockIdx.y * 16u, r1 = min(n_rows, r0 + 16u);
    for (uint32_t r = r0; r < r1; ++r) {
        const uint4 x = rows[(size_t)r * 16384u + i];
        if (x.x | x.y | x.z | x.w) rows[(size_t)r * 16384u + i] = make_uint4(0, 0, 0, 0);
        A += x.x; B += x.y; C += x.z; D += x.w;
    }
    if (A) gadd(em + 4u * i + 0u, A);
    if (B) gadd(em + 4u * i + 3u, B);
    if (C) gadd(em + 4u * i + 2u, C);
    if (D) gadd(em + 4u * i + 1u, D);
}

extern "C" void bqc_launch_short(const DevBatch& b, const StateLayout& sl, uint64_t* state, const DevRefs& refs, uint32_t* err,
                                 uint32_t grid, uint32_t* t8rows, uint32_t t8_lane, hipStream_t s)
{
    if (b.n_chunks_fast == 0) return;
    if (grid > b.n_chunks_fast) grid = b.n_chunks_fast;
    static uint32_t period = 0;
    if (!period) { const char* e = getenv("BQC_T8_PERIOD"); period = e && atoi(e) > 0 ? (uint32_t)atoi(e) : KS_T8_PERIOD; } // tuning knob
    hipLaunchKernelGGL(k_short, dim3(grid), dim3(KS_THREADS), KS_WORDS * 4, s, b, sl, state, refs, err, bqc_short_parts(), (uint4*)t8rows, t8_lane, period);
}

// fold the scratch rows into the 8-mer counters of `lane` (the rows are zero afterwards)
extern "C" void bqc_launch_t8_reduce(uint32_t* t8rows, uint32_t n_rows, const StateLayout& sl, uint64_t* state, uint32_t lane, hipStream_t s)
{
    hipLaunchKernelGGL(k_t8_reduce, dim3(64, (n_rows + 15) / 16), dim3(256), 0, s, (uint4*)t8rows, n_rows, state + sl.lane_base(lane) + sl.o_eightmer);
}
