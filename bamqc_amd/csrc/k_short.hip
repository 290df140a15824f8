// k_short.hip — short-read fast path (reads up to 256 bases, i.e. all Illumina-style data).
//
// One workgroup (16 waves) per CU walks lane-uniform chunks.  Inside a wave, `rpw = 64 / W` reads are
// processed at once; lane (s, w) owns the 8 sequencing cycles 8w..8w+7 of read slot s, so a whole
// dword of 4-bit bases / two dwords of qualities are handled per lane with SWAR arithmetic:
//
//   staging    each lane copies its dword of packed bases, its two dwords of qualities and (for
//              triplets) the matching 8 reference bases into a per-wave LDS tile (wave-local: no
//              workgroup barrier in the main loop); the next group is prefetched into registers.
//   cycles     reverse-strand reads are turned into sequencing orientation by a funnel shift +
//              v_bfrev (bit reversal = reversed base order AND complemented one-hot nibbles).  Base
//              counts per cycle are accumulated bit-sliced in registers (4-bit then 8-bit vertical
//              counters, flushed to LDS every 255 reads), quality sums as packed 16-bit sums
//              (QualityCheck.hpp:122-166).
//   8-mers     2-bit codes of 16 consecutive cycles are packed into one register; the 8 windows of a
//              lane are bit-field extracts; counters live in a 64 KiB LDS table of packed u8 fields with
//              exact carry accounting on the (rare) wrap (OverallNumbers.hpp:137-168).
//   triplets   BAM orientation: read and reference as one-hot nibbles, flank test = XOR + zero-nibble
//              detection on 8 positions at once; context index from a 2-bit reference stream
//              (TripletCounting.hpp:195-236).  Reads whose CIGAR has more than one operation go to
//              the generic kernel (k_bases_generic.hip).
//   per-read   flag cascade / histograms stay in k_reads (k_reads.hip), launched over the same chunks.
#include "kernels_common.h"

#ifndef KS_THREADS
#define KS_THREADS 1024
#endif
#define KS_WAVES (KS_THREADS / 64)
#define KS_CT 256                                  // cycles held in LDS ( = BQC_FAST_MAXLEN )
// LDS map (uint32 words)
#define KS_T8    0                                 // 16384: 65536 u8 8-mer counters, four per dword
#define KS_TRIP  (KS_T8 + 16384)                   // 1024
#define KS_CYC   (KS_TRIP + 1024)                  // [2 mates][6: A C G T other qual][KS_CT]
#define KS_NC    (KS_CYC + 2 * 6 * KS_CT)          // [2 mates][KS_CT + 1]
#define KS_GC    (KS_NC + 2 * (KS_CT + 1))
#define KS_AQ    (KS_GC + 2 * (KS_CT + 1))         // [2][256]
#define KS_AC    (KS_AQ + 512)
#define KS_STAGE (KS_AC + 512)                     // per-wave staging tiles
#define KS_WS    384                               // words per wave: rpw * (5W + 10) <= 360 for 10 <= W <= 32
#define KS_META  (KS_STAGE + KS_WAVES * KS_WS)      // per-read records of the current chunk
#define KS_CHUNK 1008
#define KS_MW    8                                 // flags|L<<20, pos, n0, ref limit | seq_off, qual_off, refn pointer (2)
#define KS_WORDS (KS_META + (KS_CHUNK + 1) * KS_MW)  // + one dummy record for lanes past the end of a chunk

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
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
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) // inclusive prefix sum over the 64 lanes
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);
    return v;
}

// 8 nibble-spaced 2-bit values (bits [1:0] of every nibble) -> 16 contiguous bits, first nibble on top
__device__ __forceinline__ uint32_t squeeze2(uint32_t c)
{
    c = (c | (c >> 2)) & 0x0F0F0F0Fu;
    c = (c | (c >> 4)) & 0x00FF00FFu;
    return (c | (c >> 8)) & 0xFFFFu;
}
// 8 nibble-LSB flags -> 8 contiguous bits, first nibble on top
__device__ __forceinline__ uint32_t squeeze1(uint32_t m)
{
    m = (m | (m >> 3)) & 0x03030303u;
    m = (m | (m >> 6)) & 0x000F000Fu;
    return (m | (m >> 12)) & 0xFFu;
}

struct Planes { uint32_t a, c, g, t, oh, n; }; // one-hot masked planes, one-hot mask, literal-N mask (nibble LSBs)
__device__ __forceinline__ Planes planes_of(uint32_t x)
{
    const uint32_t M = 0x11111111u;
    const uint32_t p0 = x & M, p1 = (x >> 1) & M, p2 = (x >> 2) & M, p3 = (x >> 3) & M;
    const uint32_t s = p0 + p1 + p2 + p3;          // per-nibble popcount (0..4)
    Planes P;
    P.oh = s & ~(s >> 1) & ~(s >> 2) & M;           // popcount == 1
    P.n = (s >> 2) & M;                             // popcount == 4: literal 'N' (code 15)
    P.a = p0 & P.oh; P.c = p1 & P.oh; P.g = p2 & P.oh; P.t = p3 & P.oh;
    return P;
}

// exact accounting when a packed u8 counter wraps: every wrap of field f is worth +256 for its bin and,
// because the carry spills into field f+1, -1 for the next bin (see header)
__device__ __noinline__ void t8_wrap(uint64_t* __restrict__ em, uint32_t h, uint32_t old)
{
    uint32_t f = h & 3u;
    uint32_t bin = h;
    while (f < 4u && ((old >> (8u * f)) & 0xFFu) == 0xFFu) {
        gadd(em + bin, 256);
        if (f < 3u) gadd(em + bin + 1, (uint64_t)-1ll);
        ++f; ++bin;
    }
}

__device__ __forceinline__ void ks_flush(uint32_t* lds, const StateLayout& sl, uint64_t* __restrict__ state, uint32_t lane)
{
    const uint64_t lb = sl.lane_base(lane);
    for (uint32_t i = threadIdx.x; i < 65536; i += blockDim.x) {
        const uint32_t v = (lds[KS_T8 + (i >> 2)] >> (8u * (i & 3u))) & 0xFFu;
        if (v) gadd(state + lb + sl.o_eightmer + i, v);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 16384; i += blockDim.x) lds[KS_T8 + i] = 0;
    for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) {
        const uint32_t v = lds[KS_TRIP + i];
        if (v) { gadd(state + lb + sl.o_triplet + i, v); lds[KS_TRIP + i] = 0; }
    }
    for (uint32_t i = threadIdx.x; i < 2 * 6 * KS_CT; i += blockDim.x) { // per-cycle counters [2 mates][A C G T other qual][KS_CT]
        const uint32_t v = lds[KS_CYC + i];
        if (!v) continue;
        lds[KS_CYC + i] = 0;
        const uint32_t m = i / (6 * KS_CT), c = (i / KS_CT) % 6, j = i % KS_CT;
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

// bit-sliced per-cycle counters of one lane (cycle group w of its slot), for the reads of ONE mate
struct CycAcc {
    uint32_t l1[4];      // 4-bit vertical counters (nibble t <-> cycle 8w + 7 - t): A C G T
    uint32_t l2[4][2];   // 8-bit: [..][0] nibbles 0,2,4,6  [..][1] nibbles 1,3,5,7
    uint32_t q[4];       // quality sums, 16-bit fields: e0 o0 e1 o1
};

__device__ __forceinline__ void cyc_zero(CycAcc& A)
{
#pragma unroll
    for (int p = 0; p < 4; ++p) { A.l1[p] = 0; A.l2[p][0] = 0; A.l2[p][1] = 0; A.q[p] = 0; }
}

__device__ __forceinline__ void cyc_spill(CycAcc& A)
{
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        A.l2[p][0] += A.l1[p] & 0x0F0F0F0Fu;
        A.l2[p][1] += (A.l1[p] >> 4) & 0x0F0F0F0Fu;
        A.l1[p] = 0;
    }
}

// Rare (every 255 groups / at a lane switch): the counters are copied to a small local array and handed to a real
// function, so that the accumulators themselves never have their address taken (which would move them to scratch).
__device__ __noinline__ void cyc_flush_arr(const uint32_t* v, uint32_t* base /* lds + KS_CYC + mate * 6 * KS_CT */, uint32_t w)
{
    const uint32_t c0 = 8u * w;
    for (int p = 0; p < 4; ++p) {
        const uint32_t a0 = v[p * 2], a1 = v[p * 2 + 1];
        for (int b = 0; b < 4; ++b) {
            const uint32_t v0 = (a0 >> (8 * b)) & 0xFFu, v1 = (a1 >> (8 * b)) & 0xFFu;
            const uint32_t cy0 = c0 + 7u - 2u * b, cy1 = c0 + 6u - 2u * b;
            if (v0 && cy0 < KS_CT) atomicAdd(base + p * KS_CT + cy0, v0);
            if (v1 && cy1 < KS_CT) atomicAdd(base + p * KS_CT + cy1, v1);
        }
    }
    // quality: qa bytes = cycles c0..c0+3 (e0: +0,+2  o0: +1,+3), qb bytes = c0+4..c0+7
    for (int k = 0; k < 4; ++k) {
        const uint32_t ca = c0 + (k == 0 ? 0 : k == 1 ? 1 : k == 2 ? 4 : 5), cb = ca + 2;
        const uint32_t x = v[8 + k], lo = x & 0xFFFFu, hi = x >> 16;
        if (lo && ca < KS_CT) atomicAdd(base + 5 * KS_CT + ca, lo);
        if (hi && cb < KS_CT) atomicAdd(base + 5 * KS_CT + cb, hi);
    }
}
__device__ __forceinline__ void cyc_flush(CycAcc& A, uint32_t* lds, uint32_t mate, uint32_t w)
{
    cyc_spill(A);
    uint32_t v[12];
#pragma unroll
    for (int p = 0; p < 4; ++p) { v[p * 2] = A.l2[p][0]; v[p * 2 + 1] = A.l2[p][1]; v[8 + p] = A.q[p]; }
    cyc_flush_arr(v, lds + KS_CYC + mate * 6 * KS_CT, w);
    cyc_zero(A);
}

__device__ __forceinline__ void cyc_add(CycAcc& A, const uint32_t pa, const uint32_t pc, const uint32_t pg, const uint32_t pt,
                                        const uint32_t qa, const uint32_t qb)
{
    A.l1[0] += pa; A.l1[1] += pc; A.l1[2] += pg; A.l1[3] += pt;
    A.q[0] += qa & 0x00FF00FFu; A.q[1] += (qa >> 8) & 0x00FF00FFu; A.q[2] += qb & 0x00FF00FFu; A.q[3] += (qb >> 8) & 0x00FF00FFu;
}

// explicit global-address-space loads: pointers that were themselves loaded from memory (refs.refn[rid]) are generic to
// the compiler, and generic (flat) loads count on lgkmcnt too, which would make every LDS wait also wait for the prefetch
typedef const __attribute__((address_space(1))) uint32_t* g_u32p;
typedef const __attribute__((address_space(1))) uint8_t* g_u8p;
__device__ __forceinline__ uint32_t gld32(const uint32_t* p) { return *(g_u32p)(uintptr_t)p; }
__device__ __forceinline__ uint32_t gld8(const uint8_t* p) { return *(g_u8p)(uintptr_t)p; }

struct Pre { uint32_t sv, q0, q1, d0, d1, c0, c1; }; // raw dwords of the NEXT group, in flight while the current one is computed

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) // unaligned little-endian dword: one global_load_dword
{
    return *(const __attribute__((address_space(1))) u32_unaligned*)(uintptr_t)p;
}

// record of one read in the chunk's LDS table (written by phase A)
#define KM_PRIM   0x10000u   // primary record with first/last flag: reaches get_count / count8mers
#define KM_TRIP   0x20000u   // triplet-eligible with a single CIGAR operation and a loaded reference
// word 0: BAM flag (low 16 bits) | KM_* | L << 20 (L = 0 unless KM_PRIM), 1: pos, 2: n0, 3: index of the contig's first pad dword,
// 4: seq_off, 5: qual_off, 6-7: pointer to the contig's nibble table

// Issue the global loads of one group for this lane.  Branch-free: records of non-primary reads (and the dummy record
// used by lanes past the end of the chunk) carry L = 0 and offsets / pointers that are safe to load from, and the
// buffers are padded, so every lane always loads; what must not be used is masked when it is staged.
__device__ __forceinline__ Pre ks_prefetch(const uint32_t* META, uint32_t k, bool in_chunk, uint32_t w, const uint8_t* seq,
                                           const uint8_t* qual)
{
    const uint32_t kk = in_chunk ? k : (uint32_t)KS_CHUNK;
    const uint4 ma = *(const uint4*)(META + kk * KS_MW), mb = *(const uint4*)(META + kk * KS_MW + 4);
    Pre P;
    P.sv = ld32u(seq + mb.x + 4u * w);
    const uint8_t* qp = qual + mb.y + 8u * w;
    P.q0 = ld32u(qp); P.q1 = ld32u(qp + 4);
    const uint32_t* rn = (const uint32_t*)(uintptr_t)((uint64_t)mb.z | ((uint64_t)mb.w << 32));
    const uint32_t di = min((ma.y + 8u * w) >> 3, ma.w); // ma.w: index of the first of the two zero dwords behind the contig
    P.d0 = gld32(rn + di); P.d1 = gld32(rn + di + 1);
    // the same bases as 2-bit codes (16 per dword) live right behind the nibble table: offset in record word 2's top bits
    const uint32_t* r2 = rn + ma.w + 2u;
    const uint32_t dj = min((ma.y + 8u * w) >> 4, (ma.w + 1u) >> 1);
    P.c0 = gld32(r2 + dj); P.c1 = gld32(r2 + dj + 1);
    return P;
}

__global__ __launch_bounds__(KS_THREADS) void k_short(DevBatch b, StateLayout sl, uint64_t* __restrict__ state, DevRefs refs,
                                                         uint32_t* __restrict__ err, uint32_t parts)
{
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < KS_WORDS; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const uint32_t ln = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t W = b.fast_w, rpw = 64u / W;
    const uint32_t slot = ln / W, w = ln % W;
    const bool lane_used = slot < rpw;
    // per-wave staging tile: SEQ [rpw][W+2] | QUAL [rpw][2W+4] | REFN [rpw][W+2] | REF2 [rpw][W+2]
    uint32_t* T = lds + KS_STAGE + wave * KS_WS;
    uint32_t* SEQ = T + slot * (W + 2);
    uint32_t* QUAL = T + rpw * (W + 2) + slot * (2 * W + 4);
    uint32_t* REFN = T + rpw * (3 * W + 6) + slot * (W + 2);
    uint32_t* REF2 = T + rpw * (4 * W + 8) + slot * (W + 2);
    uint32_t* META = lds + KS_META;
    CycAcc A0, A1; // first-mate / second-mate reads (chunks are mate-uniform: Chunk::huge carries the mate)
    cyc_zero(A0); cyc_zero(A1);
    uint32_t n1[2] = {0, 0}, n2[2] = {0, 0}; // groups since the last level-1 spill / level-2 flush (wave-uniform)
    uint32_t cur_lane = 0xFFFFFFFFu;
    if (threadIdx.x < KS_MW) // dummy record: L = 0, offsets 0, reference pointer -> any loadable memory
        META[KS_CHUNK * KS_MW + threadIdx.x] = threadIdx.x == 6 ? (uint32_t)(uintptr_t)state : threadIdx.x == 7 ? (uint32_t)((uintptr_t)state >> 32) : 0u;

    const uint8_t* const g_seq = b.seq;
    const uint8_t* const g_qual = b.qual;
    const uint32_t n_chunks = b.n_chunks_fast;
    for (uint32_t ci = blockIdx.x;; ci += gridDim.x) { // one extra pass at the end flushes the last lane (single call site)
        const bool done = ci >= n_chunks;
        Chunk ch{0, 0, 0xFFFFFFFFu, 0};
        if (!done) ch = b.chunks_fast[ci];
        if (ch.lane != cur_lane) { // block-uniform
            if (cur_lane != 0xFFFFFFFFu) {
                if (lane_used) { cyc_flush(A0, lds, 0, w); cyc_flush(A1, lds, 1, w); }
                n1[0] = n1[1] = n2[0] = n2[1] = 0;
                __syncthreads();
                ks_flush(lds, sl, state, cur_lane);
                __syncthreads();
            }
            cur_lane = ch.lane;
        }
        if (done) break;
        const uint64_t lb = sl.lane_base(cur_lane);
        uint64_t* em = state + lb + sl.o_eightmer;
        // ---- phase A: thread per read — the read's record for phase B into LDS
        for (uint32_t t = threadIdx.x; t < ch.count; t += blockDim.x) {
            const uint32_t r = b.perm ? b.perm[ch.first + t] : ch.first + t;
            if (r == 0xFFFFFFFFu) { // padding entry
                uint4* M = (uint4*)(META + t * KS_MW);
                M[0] = make_uint4(0, 0, 0, 0);
                M[1] = make_uint4(0, 0, (uint32_t)(uintptr_t)state, (uint32_t)((uintptr_t)state >> 32));
                continue;
            }
            const uint32_t fl = b.flag[r], nc = b.n_cigar[r], L = b.l_seq[r];
            const int32_t rid = b.rid[r];
            uint32_t m0 = fl & 0xFFFFu, n0 = 0, maxd = 0, so = 0, qo = 0, Lr = 0;
            uint64_t rn = (uint64_t)(uintptr_t)state; // any loadable address for reads without triplets
            if (!(fl & 0x900u) && (fl & 0xC0u)) { // reaches get_count / count8mers
                m0 |= KM_PRIM; Lr = L; so = b.seq_off[r]; qo = b.qual_off[r];
                if ((fl & BQC_FLAG_TRIPLET) && nc == 1 && L >= 3 && !(fl & BQC_FLAG_NO_QUAL) && rid >= 0 && (uint32_t)rid < refs.n_refs &&
                    refs.refn[rid] != nullptr) {
                    m0 |= KM_TRIP;
                    n0 = b.cigar[b.cigar_off[r]] >> 4;
                    rn = (uint64_t)(uintptr_t)refs.refn[rid];
                    const uint64_t md = (refs.len[rid] + 7u) >> 3; // first of the two zero dwords behind the contig
                    maxd = md > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)md;
                }
            }
            uint4* M = (uint4*)(META + t * KS_MW);
            M[0] = make_uint4(m0 | (Lr << 20), (m0 & KM_TRIP) ? (uint32_t)b.pos[r] : 0u, n0, maxd);
            M[1] = make_uint4(so, (fl & BQC_FLAG_NO_QUAL) ? 0u : qo, (uint32_t)rn, (uint32_t)(rn >> 32));
        }
        __syncthreads();
        // ---- phase B: groups of rpw reads per wave; the next group's data is loaded while this one is processed
        const uint32_t n_groups = (ch.count + rpw - 1) / rpw; // host: every group of rpw consecutive records is mate-uniform
        Pre nxt = ks_prefetch(META, wave * rpw + slot, lane_used && wave * rpw + slot < ch.count, w, g_seq, g_qual);
        for (uint32_t g = wave; g < n_groups; g += KS_WAVES) {
            const Pre cur = nxt;
            {
                const uint32_t kn = (g + KS_WAVES) * rpw + slot;
                nxt = ks_prefetch(META, kn, lane_used && kn < ch.count, w, g_seq, g_qual);
            }
            const uint32_t k = g * rpw + slot;
            const bool have = lane_used && k < ch.count;
            // mate of every read of this group: the record of slot 0 (padding entries sit at the end of a run)
            const uint32_t cm8 = (META[g * rpw * KS_MW] & 0x40u) ? 0u : 1u;
            const uint4 ma = *(const uint4*)(META + (have ? k : (uint32_t)KS_CHUNK) * KS_MW);
            const uint32_t flag = ma.x, L = ma.x >> 20, pos = ma.y, n0 = ma.z; // L = 0 unless the record reaches get_count
            const bool prim = flag & KM_PRIM;
            const bool rc = flag & 0x10u;
            const uint32_t mate = cm8;
            const uint32_t nd = (L + 7u) >> 3;           // dwords / cycle groups of this read
            const bool trip = (parts & 4u) && (flag & KM_TRIP);
            const uint32_t nv = (w < nd) ? min(8u, L - 8u * w) : 0u; // valid cycles of this lane
            // ---------------- staging (registers -> LDS tile of this wave), with the tails masked to zero
            if (lane_used) {
                // keep the top nv nibbles (also clears the pad nibble of an odd-length read) / the low nv quality bytes
                const uint32_t sv = nv ? (bswap32(cur.sv) & (0xFFFFFFFFu << (4u * (8u - nv)))) : 0u; // big-endian: base 8w on top
                const uint64_t qm = nv >= 8u ? ~0ull : ((1ull << (8u * nv)) - 1ull);
                const bool hasq = !(flag & BQC_FLAG_NO_QUAL);
                const uint32_t q0 = hasq ? cur.q0 & (uint32_t)qm : 0u, q1 = hasq ? cur.q1 & (uint32_t)(qm >> 32) : 0u;
                if ((q0 | q1) & 0x80808080u) { // some Phred >= 128: check the 222 limit precisely
                    bool bad = false;
#pragma unroll
                    for (int k8 = 0; k8 < 4; ++k8) bad |= ((q0 >> (8 * k8)) & 0xFFu) > 222u || ((q1 >> (8 * k8)) & 0xFFu) > 222u;
                    if (bad) atomicOr(err, BQC_DEVERR_QUAL);
                }
                SEQ[1 + w] = sv;
                QUAL[2 + 2 * w] = q0;
                QUAL[3 + 2 * w] = q1;
                // reference bases pos+8w .. pos+8w+7 as one-hot nibbles and as 2-bit codes (garbage for reads without
                // triplets: never read).  Position pos-1 is not needed: read position 0 is never evaluated.
                const uint32_t sh = ((pos + 8u * w) & 7u) * 4u;
                const uint32_t v = (uint32_t)((((uint64_t)cur.d0 << 32) | cur.d1) >> (32u - sh));
                REFN[1 + w] = v;
                REF2[1 + w] = (uint32_t)((((uint64_t)cur.c0 << 32) | cur.c1) >> (48u - ((pos + 8u * w) & 15u) * 2u)) & 0xFFFFu;
            }
            // LDS operations of one wave execute in order, so other lanes' ds_writes above are visible to the ds_reads
            // below; only the COMPILER must not reorder them.  (A fence or volatile accesses would insert
            // s_waitcnt vmcnt(0) and so wait for the prefetch loads that were just issued.)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();

            // ---------------- sequencing-orientation dword X and qualities qa/qb (cycles 8w .. 8w+7), one code path for
            //                  both strands: an unaligned 8-base / 8-byte window, bit- / byte-reversed for reverse reads
            uint32_t X, qa, qb;
            {
                const int32_t o = nv ? (rc ? (int32_t)L - 8 - 8 * (int32_t)w : 8 * (int32_t)w) : 0; // first base, > -8
                const int32_t d0i = o >> 3;                                                       // floor: -1 = zero pad
                const uint32_t shn = ((uint32_t)o & 7u) * 4u;
                const uint32_t hi = SEQ[1 + d0i], lo = SEQ[2 + d0i];
                const uint32_t Y = (uint32_t)((((uint64_t)hi << 32) | lo) >> (32u - shn));
                X = rc ? __brev(Y) : Y; // bit reversal = reversed base order and complemented one-hot codes (IUPAC too)
                const uint32_t bo = (uint32_t)(8 + o);                                            // byte offset into QUAL
                const uint32_t qd = bo >> 2, bs = bo & 3u;
                const uint32_t a0 = QUAL[qd], a1 = QUAL[qd + 1], a2 = QUAL[qd + 2];
                const uint32_t y0 = alignbyte(a1, a0, bs), y1 = alignbyte(a2, a1, bs);
                qa = rc ? bswap32(y1) : y0; qb = rc ? bswap32(y0) : y1;
                if (!nv) { X = 0; qa = 0; qb = 0; }
            }
            const Planes P = planes_of(X);
            // ---- bit-sliced accumulation (the chunk's mate selects the register set: wave-uniform branch)
            if (parts & 1u) {
                if (cm8 == 0) cyc_add(A0, P.a, P.c, P.g, P.t, qa, qb); else cyc_add(A1, P.a, P.c, P.g, P.t, qa, qb);
                if (++n1[cm8] == 15u) { if (cm8 == 0) cyc_spill(A0); else cyc_spill(A1); n1[cm8] = 0; }
                if (++n2[cm8] == 255u) { if (cm8 == 0) cyc_flush(A0, lds, 0, w); else cyc_flush(A1, lds, 1, w); n1[cm8] = 0; n2[cm8] = 0; }
                // cycles holding anything but A/C/G/T (Dna5 'N' bin) are rare: counted directly
                uint32_t other = nv ? (0x11111111u & (0xFFFFFFFFu << (4u * (8u - nv))) & ~P.oh) : 0u;
                while (other) {
                    const uint32_t bit = (uint32_t)__ffs((int)other) - 1u;
                    other &= other - 1u;
                    atomicAdd(&lds[KS_CYC + (cm8 * 6 + 4) * KS_CT + 8u * w + 7u - (bit >> 2)], 1u);
                }
            }
            // ---- per-read sums: N count, GC count, quality sum.  Wave-wide inclusive scans on the VALU; the total of a
            //      slot is the scan value at its last lane minus the one at the previous slot's last lane.
            if (parts & 1u) {
                const uint32_t s1 = wave_scan_incl(__builtin_amdgcn_sad_u8(qa, 0u, 0u) + __builtin_amdgcn_sad_u8(qb, 0u, 0u));
                const uint32_t s2 = wave_scan_incl((uint32_t)__popc(P.n) | ((uint32_t)__popc(P.c | P.g) << 16)); // <= 64*8 per field
                const uint32_t last = (slot + 1u) * W - 1u;                    // last lane of this slot (< 64 for used lanes)
                const uint32_t e1 = (uint32_t)__shfl((int)s1, (int)(last & 63u)), e2 = (uint32_t)__shfl((int)s2, (int)(last & 63u));
                const uint32_t b1 = lane_prev(s1), b2 = lane_prev(s2);         // scan value just before this lane (w == 0: slot start)
                if (prim && w == 0) { // keep the read's sums in its LDS record (pos / n0 are no longer needed by this lane)
                    META[k * KS_MW + 1] = e1 - b1;  // quality sum
                    META[k * KS_MW + 2] = e2 - b2;  // N count | GC count << 16
                }
            }
            // ---- 8-mers: windows starting at cycles 8w .. 8w+7
            const uint32_t cn = (P.c | P.t) | ((P.g | P.t) << 1);  // 2-bit code per nibble; non-ACGT -> A (char -> Dna after RC)
            const uint32_t c16 = squeeze2(cn);
            if (parts & 2u) {
                const uint32_t n8 = squeeze1(P.n) | (nv < 8u ? (0xFFu >> nv) : 0u); // literal N or past the end blocks a window
                uint32_t cx = lane_next(c16), nx = lane_next(n8);
                if (w + 1u >= W || ln == 63u) { cx = 0; nx = 0xFFu; }
                const uint32_t c32 = (c16 << 16) | cx, n16 = (n8 << 8) | nx;
                if (nv) {
                    uint32_t old[8];
                    uint32_t ovf = 0;
#pragma unroll
                    for (int kw = 0; kw < 8; ++kw) { // issue all returning atomics first, look at the old values afterwards
                        old[kw] = 0;
                        if (bfe(n16, 8 - kw, 8) == 0u) {
                            const uint32_t h = bfe(c32, 16 - 2 * kw, 16);
                            old[kw] = atomicAdd(&lds[KS_T8 + (h >> 2)], 1u << (8u * (h & 3u)));
                        }
                    }
                    uint32_t hot = 0;
#pragma unroll
                    for (int kw = 0; kw < 8; ++kw) hot |= old[kw];
                    if (hot & 0x80808080u) { // some counter of a touched dword is >= 128: look precisely (rare)
#pragma unroll
                        for (int kw = 0; kw < 8; ++kw) {
                            const uint32_t h = bfe(c32, 16 - 2 * kw, 16);
                            ovf |= (((old[kw] >> (8u * (h & 3u))) & 0xFFu) == 0xFFu) ? (1u << kw) : 0u; // old == 0 for skipped windows
                        }
                    }
                    if (ovf) { // rare: some packed u8 counter wrapped
#pragma unroll
                        for (int kw = 0; kw < 8; ++kw)
                            if ((ovf >> kw) & 1u) t8_wrap(em, bfe(c32, 16 - 2 * kw, 16), old[kw]);
                    }
                }
            }
            // ---- triplets (BAM orientation, single-operation CIGAR: chromPos = pos + i)
            if ((parts & 4u) && __ballot(trip)) {
                uint32_t Xf = X, c16f = c16;
                Planes F = P;
                if (rc) { // forward-orientation dword
                    Xf = nv ? SEQ[1 + w] : 0u;
                    F = planes_of(Xf);
                    c16f = squeeze2((F.c | F.t) | ((F.g | F.t) << 1));
                }
                // canonical nibbles: one-hot kept, literal N -> 0 (never matches), other -> A (char -> Dna)
                const uint32_t Z = (Xf & (F.oh * 15u)) | (0x11111111u & ~F.oh & ~F.n & (nv ? 0xFFFFFFFFu : 0u));
                uint32_t Zp = lane_prev(Z), Zn = lane_next(Z);
                if (w == 0) Zp = 0;
                if (w + 1u >= W || ln == 63u) Zn = 0;
                if (trip && nv) {
                    const uint32_t ZL = alignbit(Zp, Z, 4), ZR = alignbit(Z, Zn, 28);           // neighbours i-1 / i+1
                    const uint32_t r0 = REFN[w], r1 = REFN[1 + w], r2 = REFN[2 + w];
                    const uint32_t RL = alignbit(r0, r1, 4), RR = alignbit(r1, r2, 28);
                    uint32_t u = (ZL ^ RL) | (ZR ^ RR);
                    u |= u >> 1; u |= u >> 2;
                    uint32_t cm = ~u & F.oh;                                                     // flanks match, base is A/C/G/T
                    // positions 1 <= i <= L-2, and i < n0 when the single CIGAR op is shorter than the read
                    uint32_t lim = L - 1u;
                    if (n0 != 0u && n0 < lim) lim = n0;
                    const uint32_t cnt = lim > 8u * w ? min(8u, lim - 8u * w) : 0u;
                    uint32_t pm = cnt ? (0xFFFFFFFFu << (4u * (8u - cnt))) : 0u;
                    if (w == 0) pm &= 0x0FFFFFFFu;
                    cm &= pm;
                    if (cm) {
                        const uint32_t f0 = QUAL[2 + 2 * w], f1 = QUAL[3 + 2 * w];                  // forward qualities
                        const uint32_t r2s = ((REF2[w] & 3u) << 18) | (REF2[1 + w] << 2) | (REF2[2 + w] >> 14); // codes of pos-1 .. pos+8
                        const uint32_t grp = (rc ? 2u : 0u) + mate;
#pragma unroll
                        for (int kp = 0; kp < 8; ++kp) {
                            const uint32_t q = bfe(kp < 4 ? f0 : f1, 8 * (kp & 3), 8);
                            if (((cm >> (28 - 4 * kp)) & 1u) && (q - 20u) <= 74u) {             // (signed char)(q+33) >= '5'
                                const uint32_t ctx = bfe(r2s, 14 - 2 * kp, 6), base = bfe(c16f, 14 - 2 * kp, 2);
                                atomicAdd(&lds[KS_TRIP + ctx * 16u + grp * 4u + base], 1u);
                            }
                        }
                    }
                }
            }
            asm volatile("" ::: "memory"); // the tile is rewritten by the next group
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        // ---- phase C: thread per read — per-read histograms from the sums left in the records (QualityCheck.hpp:157-165)
        if (parts & 1u)
            for (uint32_t t = threadIdx.x; t < ch.count; t += blockDim.x) {
                const uint32_t m0 = META[t * KS_MW];
                if (!(m0 & KM_PRIM)) continue;
                const uint32_t L = m0 >> 20, qs = META[t * KS_MW + 1], v2 = META[t * KS_MW + 2];
                const uint32_t mate = (m0 & 0x40u) ? 0u : 1u;
                atomicAdd(&lds[KS_NC + mate * (KS_CT + 1) + (v2 & 0xFFFFu)], 1u);
                atomicAdd(&lds[KS_GC + mate * (KS_CT + 1) + (v2 >> 16)], 1u);
                if (L > 0) { // round-half-away and ceil of qs/L in exact integer arithmetic
                    atomicAdd(&lds[KS_AQ + mate * 256 + (((2u * qs + L) / (2u * L)) & 255u)], 1u);
                    atomicAdd(&lds[KS_AC + mate * 256 + (((qs + L - 1u) / L) & 255u)], 1u);
                }
            }
        __syncthreads(); // META is rewritten by the next chunk
    }
}

// Dna5 bytes -> (a) one-hot nibbles, 8 bases per dword, first base in the top nibble, followed by two zero dwords;
//               (b) right behind: 2-bit codes, 16 bases per dword, first base in the top bits, followed by two zero dwords.
// N (4) -> A like Dna5 -> Dna.  nd8 = ceil(len / 8): table (a) has nd8 + 2 dwords, table (b) starts at out + nd8 + 2.
__global__ void k_ref_nibbles(const uint8_t* __restrict__ dna5, uint64_t len, uint32_t* __restrict__ out, uint64_t nd8)
{
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nd16 = (nd8 + 1) / 2;
    if (d < nd8 + 2) {
        uint32_t v = 0;
        for (uint32_t t = 0; t < 8; ++t) {
            const uint64_t p = d * 8 + t;
            if (p < len) v |= (1u << (dna5[p] & 3u)) << (28u - 4u * t);
        }
        out[d] = v;
    }
    if (d < nd16 + 2) {
        uint32_t v = 0;
        for (uint32_t t = 0; t < 16; ++t) {
            const uint64_t p = d * 16 + t;
            if (p < len) v |= (uint32_t)(dna5[p] & 3u) << (30u - 2u * t);
        }
        out[nd8 + 2 + d] = v;
    }
}

extern "C" void bqc_launch_ref_nibbles(const uint8_t* dna5, uint64_t len, uint32_t* out, uint64_t nd8, hipStream_t s)
{
    hipLaunchKernelGGL(k_ref_nibbles, dim3((uint32_t)((nd8 + 2 + 255) / 256)), dim3(256), 0, s, dna5, len, out, nd8);
}

extern "C" hipError_t bqc_short_init()
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_short), hipFuncAttributeMaxDynamicSharedMemorySize, KS_WORDS * 4);
}

extern "C" void bqc_launch_short(const DevBatch& b, const StateLayout& sl, uint64_t* state, const DevRefs& refs, uint32_t* err,
                                 uint32_t grid, hipStream_t s)
{
    if (b.n_chunks_fast == 0) return;
    if (grid > b.n_chunks_fast) grid = b.n_chunks_fast;
    static uint32_t parts = 0xFFFFFFFFu; // BQC_SHORT_PARTS: ablation switch for profiling (1 cycles, 2 8-mers, 4 triplets, 8 per-read)
    if (parts == 0xFFFFFFFFu) { const char* e = getenv("BQC_SHORT_PARTS"); parts = e ? (uint32_t)atoi(e) : 15u; }
    hipLaunchKernelGGL(k_short, dim3(grid), dim3(KS_THREADS), KS_WORDS * 4, s, b, sl, state, refs, err, parts);
}
