// k_short.hip — short-read fast path (reads up to 255 bases, i.e. all Illumina-style data).
//
// One workgroup (16 waves) per CU walks lane-uniform chunks.  Inside a wave, `rpw = 64 / W` reads are processed at
// once; lane (s, w) owns 8 * BQC_FAST_NH = 16 SEQUENCING CYCLES (16w..16w+15) of read slot s.  Everything is loaded straight from
// global memory in cycle orientation — for a reverse-strand read the lane reads the mirrored window and
// reverses it in registers — one group ahead of its use, and all per-base work is SWAR on those registers:
//
//   orientation  the BAM base codes are one-hot nibbles (A1 C2 G4 T8), so v_bfrev_b32 yields the reversed AND
//                complemented window in one instruction; qualities are reversed by the byte selector of v_perm_b32;
//                the reference is stored as nibbles  r1 r0 ~r0 ~r1  (r = 2-bit code), which bit reversal also turns
//                into the reverse complement (k_ref_nibbles).
//   cycles       four one-hot planes are added into 4-bit vertical counters in registers (one add per plane and 8
//                cycles), spilled to the LDS cycle tile every 15 groups; quality sums as packed 16-bit fields;
//                per-read sums (quality | N << 16 | GC << 24) added into the read's LDS record (QualityCheck.hpp:122-166).
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
#include "swar.h"

#define KS_NH BQC_FAST_NH                          // 8-cycle halves per lane
#define KS_NB (8 * KS_NH)                         // sequencing cycles per lane
#define KS_ND (2 * KS_NH)                         // quality dwords per lane
#define KS_CSTRIDE (256 / KS_NB)                   // LDS cycle tile: cycle c at (c % KS_NB) * KS_CSTRIDE + c / KS_NB
#define KS_LUTW (KS_NH == 2 ? 8 : 16)              // words per mask-table entry: KS_NH nibble masks, then (at word 4) KS_ND byte masks
#define KS_WAVES BQC_FAST_WAVES
#define KS_THREADS (KS_WAVES * 64)
#define KS_CT 256                                  // cycles held in LDS ( > BQC_FAST_MAXLEN )
// LDS map (uint32 words)
#define KS_T8    0                                 // 16384: 65536 u8 8-mer counters, four per dword
#define KS_TRIP  (KS_T8 + 16384)                   // [4 groups][256]
#define KS_CYC   (KS_TRIP + 1024)                  // [2 mates][6: A C G T other qual][KS_CT], cycle c at (c % KS_NB) * KS_CSTRIDE + c / KS_NB:
                                                   // the lanes of a read (c / KS_NB = w) hit different banks when they add the same c % KS_NB
#define KS_NC    (KS_CYC + 2 * 6 * KS_CT)          // [2 mates][KS_CT + 1]
#define KS_GC    (KS_NC + 2 * (KS_CT + 1))
#define KS_AQ    (KS_GC + 2 * (KS_CT + 1))         // [2][256]
#define KS_AC    (KS_AQ + 512)
#define KS_LUT   (KS_AC + 512)                     // [KS_NB + 1][KS_LUTW] masks for "the first n of the lane's cycles"
#define KS_META  (KS_LUT + (KS_NB + 1) * KS_LUTW + (((KS_NB + 1) * KS_LUTW) % 4 ? 4 - ((KS_NB + 1) * KS_LUTW) % 4 : 0))                 // per-read records: 64 per wave (the reads the wave is working on)
#define KS_MW    8
#define KS_RS    (KS_META + (KS_WAVES * 64 + 1) * KS_MW)  // (16 waves x 64 records + one dummy record); per-read statistics (read_stats.h)
#define KS_WORDS (KS_RS + RS_WORDS)
#define KS_T8_PERIOD 16                            // chunks between two flushes of the 8-mer counters (~35 counts per bin)
#define KS_BIAS  512u                              // seq / qual offsets in the records are biased so that they stay non-negative

static_assert((KS_LUT % 4) == 0 && (KS_META % 4) == 0, "16-byte alignment of the LDS tables");

// The packed 8-mer counters of this workgroup are emptied into the next free row of its slot of a global scratch table: plain
// 16-byte stores of the LDS image, nothing to wait for (k_t8_fold unpacks and sums the rows into the state vector later);
// or with global atomics when the read group is not the one the scratch table is collecting in this launch, or the slot is
// full.  Called workgroup-uniformly between barriers; returns true when a row was written.
__device__ __forceinline__ bool t8_flush(uint32_t* lds, uint64_t* __restrict__ em, uint4* __restrict__ slot /* nullptr: atomics */, uint32_t n_rows_used)
{
    if (slot && n_rows_used < BQC_T8_SPW) {
        uint4* row = slot + (size_t)n_rows_used * 4096u;
        uint4* src = (uint4*)(lds + KS_T8);
        for (uint32_t i = threadIdx.x; i < 4096u; i += blockDim.x) { row[i] = src[i]; src[i] = make_uint4(0, 0, 0, 0); }
        return true;
    }
    t8_atomics_out(lds + KS_T8, em);
    return false;
}

__device__ __forceinline__ bool ks_flush(uint32_t* lds, const StateLayout& sl, uint64_t* __restrict__ state, uint32_t lane, uint4* __restrict__ slot, uint32_t n_rows_used)
{
    const uint64_t lb = sl.lane_base(lane);
    const bool wrote_row = t8_flush(lds, state + lb + sl.o_eightmer, slot, n_rows_used);
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
        const uint32_t m = i / (6 * KS_CT), c = (i / KS_CT) % 6, jj = i % KS_CT, j = (jj % KS_CSTRIDE) * KS_NB + jj / KS_CSTRIDE;
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
    return wrote_row;
}

// Per-cycle accumulators of one lane (its KS_NB cycles), for the reads of ONE mate.
struct CycAcc {
    uint32_t l1[KS_NH][4];       // [half][A C G T]: 4-bit vertical counters, nibble t <-> cycle 8*half + 7 - t
    uint32_t qo[KS_ND], qe[KS_ND]; // quality sums of dword d (cycles 4d..4d+3), 16-bit fields: qo = cycles 4d | 4d+2, qe = 4d+1 | 4d+3
};
__device__ __forceinline__ void cyc_zero(CycAcc& A)
{
#pragma unroll
    for (int h = 0; h < KS_NH; ++h)
#pragma unroll
        for (int p = 0; p < 4; ++p) A.l1[h][p] = 0;
#pragma unroll
    for (int d = 0; d < KS_ND; ++d) { A.qo[d] = 0; A.qe[d] = 0; }
}
// The counters go to the LDS cycle tile through real functions with by-value arguments (registers, no scratch): rare.
// (round 4: these two helpers make FLAT atomics — `base` is a generic pointer —; taking the tile's LDS byte address instead, as k_long's now do,
// measured 3 % SLOWER here (0.821 against 0.795 ms per 4 M reads, gpurun_out/r7g): left as they are)
__device__ __noinline__ void cyc_spill_half(uint32_t a, uint32_t c, uint32_t g, uint32_t t, uint32_t* base /* lds + KS_CYC + mate * 6 * KS_CT + w + KS_CSTRIDE * 8 * half */)
{
    const uint32_t v[4] = {a, c, g, t};
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int n = 0; n < 8; ++n) atomicAdd(base + p * KS_CT + KS_CSTRIDE * (7 - n), (v[p] >> (4 * n)) & 15u);
}
__device__ __noinline__ void cyc_qflush_pair(uint32_t o0, uint32_t e0, uint32_t o1, uint32_t e1, uint32_t* base /* lds + KS_CYC + (mate * 6 + 5) * KS_CT + w + KS_CSTRIDE * 8 * half */)
{
    const uint32_t vo[2] = {o0, o1}, ve[2] = {e0, e1};
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        atomicAdd(base + KS_CSTRIDE * (4 * d + 0), vo[d] >> 16);
        atomicAdd(base + KS_CSTRIDE * (4 * d + 1), ve[d] >> 16);
        atomicAdd(base + KS_CSTRIDE * (4 * d + 2), vo[d] & 0xFFFFu);
        atomicAdd(base + KS_CSTRIDE * (4 * d + 3), ve[d] & 0xFFFFu);
    }
}
__device__ __forceinline__ void cyc_spill(CycAcc& A, uint32_t* lds, uint32_t mate, uint32_t w)
{
#pragma unroll
    for (int h = 0; h < KS_NH; ++h) {
        cyc_spill_half(A.l1[h][0], A.l1[h][1], A.l1[h][2], A.l1[h][3], lds + KS_CYC + mate * 6 * KS_CT + w + KS_CSTRIDE * 8 * h);
#pragma unroll
        for (int p = 0; p < 4; ++p) A.l1[h][p] = 0;
    }
}
__device__ __forceinline__ void cyc_qflush(CycAcc& A, uint32_t* lds, uint32_t mate, uint32_t w)
{
#pragma unroll
    for (int h = 0; h < KS_NH; ++h) {
        cyc_qflush_pair(A.qo[2 * h], A.qe[2 * h], A.qo[2 * h + 1], A.qe[2 * h + 1], lds + KS_CYC + (mate * 6 + 5) * KS_CT + w + KS_CSTRIDE * 8 * h);
        A.qo[2 * h] = A.qe[2 * h] = A.qo[2 * h + 1] = A.qe[2 * h + 1] = 0;
    }
}
__device__ __forceinline__ void cyc_add(CycAcc& A, const Planes (&P)[KS_NH], const uint32_t (&Q)[KS_ND])
{
#pragma unroll
    for (int h = 0; h < KS_NH; ++h) { A.l1[h][0] += P[h].a; A.l1[h][1] += P[h].c; A.l1[h][2] += P[h].g; A.l1[h][3] += P[h].t; }
#pragma unroll
    for (int d = 0; d < KS_ND; ++d) { A.qe[d] += Q[d] & 0x00FF00FFu; A.qo[d] += __builtin_amdgcn_perm(0u, Q[d], 0x0C030C01u); } // (Q >> 8) & 0x00FF00FF
}

// record of one read in the chunk's LDS table (written by phase A)
#define KM_PRIM   0x10000u   // primary record with first/last flag: reaches get_count / count8mers
#define KM_TRIP   0x20000u   // triplets are evaluated for the position range in word 5 (needs a loaded reference)
#define KM_SEG    0x40000u   // triplet-segment entry: the read's bases / qualities for triplets only (no KM_PRIM)
// word 0: BAM flag (low 16 bits) | KM_* | L << 20 (L = 0 unless KM_PRIM)
//      1: seq byte offset of the window of lane w = 0 (+ KS_BIAS)      -> after the group was computed: the read's packed sums
//      2: qual byte offset of the window of lane w = 0 (+ KS_BIAS)
//      3: reference nibble index of that window, minus one (pos + o0 + NB - 1; the table has NB = 8 * BQC_FAST_NH pad nibbles in front)
//      4: last loadable dword index of the contig's table     5: ja | jb << 8 | seq funnel shift << 16
//      6-7: pointer to the contig's nibble table
// Lane w of a forward read loads the window starting at base NB w, of a reverse read the one starting at L - NB - NB w.

// NH nibble masks / ND byte masks of one mask-table entry (16-byte LDS reads)
__device__ __forceinline__ void lut_nib(uint32_t (&d)[KS_NH], const uint32_t* e)
{
    if (KS_NH == 2) { const uint2 v = *(const uint2*)e; d[0] = v.x; d[1] = v.y; }
    else { const uint4 v = *(const uint4*)e; d[0] = v.x; d[1] = v.y; d[KS_NH - 2] = v.z; d[KS_NH - 1] = v.w; }
}
__device__ __forceinline__ void lut_byte(uint32_t (&d)[KS_ND], const uint32_t* e)
{
#pragma unroll
    for (int q = 0; q < KS_ND / 4; ++q) { const uint4 v = *(const uint4*)(e + 4 * q); d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w; }
}

// raw data of the NEXT group, in flight while the current one is computed: KS_NH + 1 dwords of packed bases (the window may start
// up to two nibbles into them), KS_ND dwords of qualities, KS_NH + 1 dwords of reference nibbles
struct Pre { uint32_t s[KS_NH + 1], q[KS_ND], e[KS_NH + 1]; uint32_t m0, pp, w5; };

// Issue the global loads of one group for this lane.  Branch-free: records of non-primary reads, padding entries and the
// dummy record used by lanes past the end of the chunk carry L = 0 and offsets / pointers that are safe to load from
// (the buffers are padded on both sides), so every lane always loads; what must not be used is masked when consumed.
__device__ __forceinline__ Pre ks_prefetch(const uint32_t* rec, uint32_t w, const uint8_t* seqb, const uint8_t* qualb)
{
    const uint4 ma = *(const uint4*)rec, mb = *(const uint4*)(rec + 4);
    const int32_t sw = (ma.x & 0x10u) ? -(int32_t)w : (int32_t)w;
    Pre P;
    P.m0 = ma.x;
    P.w5 = mb.y;
    P.pp = ma.w + (uint32_t)(KS_NB * sw);
    GVec<KS_NH + 1>::ldu(P.s, seqb + (uint32_t)(ma.y + (uint32_t)(KS_NB / 2 * sw)));
    GVec<KS_ND>::ldu(P.q, qualb + (uint32_t)(ma.z + (uint32_t)(KS_NB * sw)));
    const uint32_t* rn = (const uint32_t*)(uintptr_t)((uint64_t)mb.z | ((uint64_t)mb.w << 32));
    const int32_t di = min(max((int32_t)P.pp >> 3, 0), (int32_t)mb.x);
    GVec<KS_NH + 1>::lda(P.e, (const uint8_t*)(rn + di));
    return P;
}

__global__ __launch_bounds__(KS_THREADS) void k_short(DevBatch b, StateLayout sl, uint64_t* __restrict__ state, DevRefs refs,
                                                         uint32_t* __restrict__ err, uint32_t parts, uint4* __restrict__ t8rows,
                                                         uint32_t* __restrict__ t8_used, uint32_t t8_period)
{
    extern __shared__ uint32_t lds[];
    if ((uint32_t)(uintptr_t)(lds_u32*)lds != 0u) { // the 8-mer atomics address LDS directly (KS_T8 at LDS address 0)
        if (threadIdx.x == 0) { atomicOr(err, BQC_DEVERR_INTERNAL); t8_used[blockIdx.x * BQC_T8_USED] = 0; }
        return;
    }
    for (uint32_t i = threadIdx.x; i < KS_WORDS; i += blockDim.x) lds[i] = 0;
    block_sync();
    const uint32_t M = 0x11111111u;
    const uint32_t ln = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const BatchDesc desc = *b.desc; // written by the batch's k_build_plan
    const uint32_t W = desc.fast_w, rpw = 64u / W;
    const uint32_t slot = ln / W, w = ln % W, wnb = KS_NB * w;
    const bool lane_used = slot < rpw;
    const bool last_w = (w + 1u >= W) || ln == 63u;   // the next lane belongs to another read
    const uint32_t tile_cap = rpw * (64u / rpw);        // reads a wave takes at a time: one record per lane, whole groups
    uint32_t* WM = lds + KS_META + wave * 64u * KS_MW;   // this wave's records
    const uint32_t* DUMMY = lds + KS_META + KS_WAVES * 64u * KS_MW; // record for lanes behind the last slot
    const uint32_t* LUT = lds + KS_LUT;
    if (threadIdx.x <= KS_NB) { // masks for "the first n of the lane's cycles"
        const uint32_t n = threadIdx.x;
        uint32_t* e = lds + KS_LUT + KS_LUTW * n;
        for (uint32_t h = 0; h < KS_NH; ++h) {
            const uint32_t v = n > 8u * h ? (n - 8u * h < 8u ? n - 8u * h : 8u) : 0u;
            e[h] = v ? 0xFFFFFFFFu << (4u * (8u - v)) : 0u;
        }
        for (uint32_t d = 0; d < KS_ND; ++d) {
            const uint32_t v = n > 4u * d ? (n - 4u * d < 4u ? n - 4u * d : 4u) : 0u;
            e[4 + d] = v ? 0xFFFFFFFFu << (8u * (4u - v)) : 0u; // qualities are kept big-endian: first cycle in the top byte
        }
    }
    if (threadIdx.x < KS_MW) { // dummy record: L = 0, loadable offsets / pointer
        const uint32_t t = threadIdx.x;
        lds[KS_META + KS_WAVES * 64u * KS_MW + t] = t == 1 || t == 2 ? KS_BIAS : t == 3 ? 8u * KS_NH - 1u : t == 5 ? (24u << 16) : t == 6 ? (uint32_t)(uintptr_t)state
                                   : t == 7 ? (uint32_t)((uintptr_t)state >> 32) : 0u;
    }
    block_sync();
    CycAcc A; // per-cycle accumulators of this lane: reads of ONE mate (fixed per slot, see the chunk layout below)
    cyc_zero(A);
    const uint32_t mate = slot < (rpw + 1u) / 2u ? 0u : 1u;
    uint32_t n1 = 0, n2 = 0; // groups since the last counter spill / quality flush (wave-uniform)
    uint32_t cur_lane = 0xFFFFFFFFu, since_t8 = 0;
    uint4* const t8_slot = t8rows + (size_t)blockIdx.x * BQC_T8_SPW * 4096u; // this workgroup's rows of the scratch table
    uint32_t t8_n = 0;                                                          // rows written so far
    T8Tags t8_tags;                                                             // ... and their read groups

    const uint8_t* g_seq = b.seq - KS_BIAS;
    const uint8_t* g_qual = b.qual - KS_BIAS;
    const uint32_t n_chunks = desc.n_chunks_fast;
    {   // chunks between two flushes of the packed 8-mer counters: as small as the rows of a workgroup's slot allow (the last row
        // is for the final flush; beyond the slot the kernel falls back to global atomics)
        const uint32_t cpw = (n_chunks + gridDim.x - 1) / gridDim.x;
        t8_period = max(t8_period, (cpw + BQC_T8_SPW - 2) / (BQC_T8_SPW - 1));
    }
    // Which chunks a workgroup walks.  Workgroup ids go round the eight XCDs (id % 8), each with an L2 of its own, and neighbouring
    // chunks share lines (the columns of one stretch of the stream, the reference under it): with parts bit 16 an XCD's workgroups
    // take one CONTIGUOUS eighth of the chunk table, so that a line two chunks share is fetched into one L2, not two.
    uint32_t ci = blockIdx.x, ci_step = gridDim.x, ci_end = n_chunks;
    if ((parts & 16u) && (gridDim.x & 7u) == 0u && n_chunks >= gridDim.x) {
        const uint32_t per = (n_chunks + 7u) >> 3, lo = (blockIdx.x & 7u) * per;
        ci = lo + (blockIdx.x >> 3); ci_step = gridDim.x >> 3; ci_end = min(n_chunks, lo + per);
    }
    for (;; ci += ci_step) { // one extra pass at the end flushes the last lane (single call site)
        const bool done = ci >= ci_end;
        Chunk ch{0, 0, 0xFFFFFFFFu, 0, 0, 0, 0, 0};
        if (!done) ch = b.chunks_fast[ci];
        if (ch.lane != cur_lane) { // block-uniform
            if (cur_lane != 0xFFFFFFFFu) {
                if (lane_used) { cyc_spill(A, lds, mate, w); cyc_qflush(A, lds, mate, w); }
                n1 = n2 = 0;
                block_sync();
                if (ks_flush(lds, sl, state, cur_lane, t8_slot, t8_n)) { t8_tag(t8_tags, t8_n, cur_lane); ++t8_n; }
                rs_flush(lds + KS_RS, sl, state, cur_lane);
            }
            cur_lane = ch.lane;
            since_t8 = 0;
        }
        if (done) break;
        if (++since_t8 > t8_period) { // keep the packed u8 counters small, so that the wrap check below stays on its fast path
            // No barrier: every wave empties ITS sixteenth of the table with atomic exchanges, whenever it gets here; increments
            // the other waves are still making for earlier chunks simply land in a later row.  All waves walk the same chunk
            // sequence, so they agree on the row index.
            since_t8 = 1;
            const bool to_row = t8_n < BQC_T8_SPW;
            uint32_t* row = (uint32_t*)(t8_slot + (size_t)t8_n * 4096u);
            uint64_t* emf = state + sl.lane_base(cur_lane) + sl.o_eightmer;
#pragma unroll 4
            for (uint32_t k = 0; k < 16384u / (KS_WAVES * 64u); ++k) {
                const uint32_t i = (wave * (16384u / (KS_WAVES * 64u)) + k) * 64u + ln;
                const uint32_t v = atomicExch(&lds[KS_T8 + i], 0u);
                if (to_row) row[i] = v;
                else if (v) {
#pragma unroll
                    for (uint32_t b8 = 0; b8 < 4; ++b8)
                        if ((v >> (8u * b8)) & 0xFFu) gadd(emf + 4u * i + ((4u - b8) & 3u), (v >> (8u * b8)) & 0xFFu);
                }
            }
            if (to_row) { t8_tag(t8_tags, t8_n, cur_lane); ++t8_n; }
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
            r = ln < tn ? b.perm[t] : 0xFFFFFFFFu;
            const bool live = r != 0xFFFFFFFFu;                // not a padding entry
            const bool seg = live && (r & BQC_ENTRY_SEG);      // triplet segment of a read (the read itself is another entry)
            stat = live && !seg;
            TripSeg sg{0, 0, 0, 0};
            if (seg) { sg = b.segs[r & ~BQC_ENTRY_SEG]; r = sg.r; }
            uint4 R0 = make_uint4(0u, KS_BIAS, KS_BIAS, 8u * KS_NH - 1u);
            uint4 R1 = make_uint4(0u, 24u << 16, (uint32_t)(uintptr_t)state, (uint32_t)((uintptr_t)state >> 32));
            if (live) {
                const uint32_t fl = b.flag[r];
                R0.x = fl & 0xFFFFu;
                if (!(fl & 0x900u) && (fl & 0xC0u)) { // reaches get_count / count8mers
                    const uint32_t L = b.l_seq[r];
                    const bool rc = fl & 0x10u, noq = fl & BQC_FLAG_NO_QUAL;
                    const int32_t o0 = rc ? (int32_t)L - KS_NB : 0;
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
                            R0.w = (uint32_t)((int32_t)pos + o0 + 8 * KS_NH - 1);
                            R1.x = (uint32_t)(nd8 + KS_NH - 1);
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
            const uint32_t nv = (uint32_t)min(max((int32_t)L - (int32_t)wnb, 0), KS_NB);   // valid cycles of this lane
            const uint32_t nvq = (m0 & BQC_FLAG_NO_QUAL) ? 0u : nv;
            uint32_t xm[KS_NH], qm[KS_ND];
            lut_nib(xm, LUT + KS_LUTW * nv);
            lut_byte(qm, LUT + KS_LUTW * nvq + 4u);
            // ---------------- the lane's cycles: one-hot base nibbles X[h] = cycles 8h..8h+7 (first cycle in the top nibble) ...
            uint32_t X[KS_NH];
            {
                const uint32_t sh = w5 >> 16; // 24 / 28: the loaded bytes start 2 / 1 nibbles before the window
                uint32_t bs[KS_NH + 1], F[KS_NH];
#pragma unroll
                for (int h = 0; h <= KS_NH; ++h) bs[h] = bswap32(cur.s[h]);
#pragma unroll
                for (int h = 0; h < KS_NH; ++h) F[h] = alignbit(bs[h], bs[h + 1], sh);
#pragma unroll
                for (int h = 0; h < KS_NH; ++h) // bit reversal = reversed base order and complemented one-hot codes (IUPAC too)
                    X[h] = (rc ? __brev(F[KS_NH - 1 - h]) : F[h]) & xm[h];
            }
            // ... and qualities Q[d] = cycles 4d..4d+3, first cycle in the top byte
            uint32_t Q[KS_ND];
            {
                const uint32_t sel = rc ? 0x07060504u : 0x00010203u; // reverse read: dwords in reverse order; forward read: bytes swapped
                uint32_t any = 0;
#pragma unroll
                for (int d = 0; d < KS_ND; ++d) { Q[d] = vperm(cur.q[KS_ND - 1 - d], cur.q[d], sel) & qm[d]; any |= Q[d]; }
                if (any & 0x80808080u) { // some Phred >= 128: check the 222 limit precisely
                    bool bad = false;
#pragma unroll
                    for (int d = 0; d < KS_ND; ++d)
#pragma unroll
                        for (int k8 = 0; k8 < 4; ++k8) bad |= ((Q[d] >> (8 * k8)) & 0xFFu) > 222u;
                    if (bad) atomicOr(err, BQC_DEVERR_QUAL);
                }
            }
            // ... and the reference window as nibbles r1 r0 ~r0 ~r1: bit reversal = reverse complement here too
            uint32_t E[KS_NH];
            {
                const uint32_t sh = 28u - 4u * (cur.pp & 7u);
                uint32_t F[KS_NH];
#pragma unroll
                for (int h = 0; h < KS_NH; ++h) F[h] = alignbit(cur.e[h], cur.e[h + 1], sh);
#pragma unroll
                for (int h = 0; h < KS_NH; ++h) E[h] = rc ? __brev(F[KS_NH - 1 - h]) : F[h];
            }
            { // the raw registers are free again: issue the loads of this wave's next group
                cur = ks_prefetch(lane_used && g + 1u < n_groups ? WM + (k + rpw) * KS_MW : DUMMY, w, g_seq, g_qual);
            }
            Planes P[KS_NH];
#pragma unroll
            for (int h = 0; h < KS_NH; ++h) P[h] = planes_of(X[h]);
            // ---- per-cycle counters (this lane's mate is fixed: one register set)
            if ((parts & 1u) && !segg) {
                cyc_add(A, P, Q);
                if (++n1 == 15u) { if (lane_used) cyc_spill(A, lds, mate, w); n1 = 0; }
                if (++n2 == 255u) { if (lane_used) cyc_qflush(A, lds, mate, w); n2 = 0; }
                // cycles holding anything but A/C/G/T (Dna5 'N' bin) are rare: counted directly
                uint32_t oth[KS_NH], oany = 0;
#pragma unroll
                for (int h = 0; h < KS_NH; ++h) { oth[h] = xm[h] & M & ~P[h].oh; oany |= oth[h]; }
                if (oany) {
                    uint32_t* ob = lds + KS_CYC + (mate * 6 + 4) * KS_CT + w;
#pragma unroll
                    for (int h = 0; h < KS_NH; ++h) {
                        uint32_t o = oth[h];
                        while (o) { const uint32_t bit = (uint32_t)__ffs((int)o) - 1u; o &= o - 1u; atomicAdd(ob + KS_CSTRIDE * (8u * h + 7u - (bit >> 2)), 1u); }
                    }
                }
                // per-read sums: quality | N << 16 | GC << 24 (L <= 255), added up in word 1 of the read's record
                uint32_t v = 0, vn = 0, vgc = 0;
#pragma unroll
                for (int d = 0; d < KS_ND; ++d) v = __builtin_amdgcn_sad_u8(Q[d], 0u, v);
#pragma unroll
                for (int h = 0; h < KS_NH; ++h) { vn += (uint32_t)__popc(P[h].n); vgc += (uint32_t)__popc(P[h].c | P[h].g); }
                v |= vn << 16;
                v += vgc << 24;
                if (lane_used) { // (the DS operations of a wave execute in order: the store precedes the adds)
                    uint32_t* sw = WM + k * KS_MW + 1;
                    if (w == 0u) *sw = 0u;
                    atomicAdd(sw, v);
                }
            }
            // ---- 2-bit codes per nibble; non-ACGT -> A (char -> Dna after the reverse complement)
            uint32_t cn[KS_NH], nb[KS_NH + 1];
#pragma unroll
            for (int h = 0; h < KS_NH; ++h) {
                cn[h] = (P[h].c | P[h].t) | ((P[h].g | P[h].t) << 1);
                nb[h] = P[h].n | (~xm[h] & M); // literal N or past the end of the read: blocks 8-mer windows and triplet flanks
            }
            // ---- 8-mers: windows starting at the lane's cycles
            if ((parts & 2u) && !segg) {
                uint32_t S[KS_NH / 2 + 1]; // 2-bit codes of 16 cycles per register, first cycle in the top two bits; then the next lane's
#pragma unroll
                for (int j = 0; j < KS_NH / 2; ++j) S[j] = vperm(squeeze2(cn[2 * j]), squeeze2(cn[2 * j + 1]), 0x05040100u);
                S[KS_NH / 2] = lane_next(S[0]);
                nb[KS_NH] = lane_next(nb[0]);
                if (last_w) nb[KS_NH] = M;
                // a window is blocked when any of its 8 cycles is: OR-smear over the next 7 positions of the flag stream
                uint32_t f[KS_NH];
                {
                    uint32_t sa[KS_NH + 1], sb[KS_NH + 1];
#pragma unroll
                    for (int h = 0; h < KS_NH; ++h) sa[h] = nb[h] | alignbit(nb[h], nb[h + 1], 28);
                    sa[KS_NH] = nb[KS_NH] | (nb[KS_NH] << 4);
#pragma unroll
                    for (int h = 0; h < KS_NH; ++h) sb[h] = sa[h] | alignbit(sa[h], sa[h + 1], 24);
                    sb[KS_NH] = sa[KS_NH] | (sa[KS_NH] << 8);
#pragma unroll
                    for (int h = 0; h < KS_NH; ++h) f[h] = ~(sb[h] | alignbit(sb[h], sb[h + 1], 16)); // nibble LSB set <=> the window starting there is counted
                }
                // Branch-free inside a batch of 16 windows: every lane with a valid cycle there issues all 16 returning atomics;
                // a blocked window adds 0.  The address is the LDS byte address itself: KS_T8 = 0 and the dynamic LDS block
                // starts at 0 (checked at kernel entry), which saves the base addition per window.
#pragma unroll
                for (int j = 0; j < KS_NH / 2; ++j) {
                    if (nv > 16u * j) { // all 16 atomics are issued before the first returned value is looked at
                        const uint32_t c32 = S[j], cx = S[j + 1];
                        uint32_t old[16];
#pragma unroll
                        for (int kw = 0; kw < 16; ++kw) {
                            const uint32_t h = kw < 8 ? c32 >> (16 - 2 * kw) : kw == 8 ? c32 : alignbit(c32, cx, 48 - 2 * kw); // window in the low 16 bits
                            const uint32_t one = bfe(f[2 * j + (kw >> 3)], 28 - 4 * (kw & 7), 1);
                            old[kw] = __hip_atomic_fetch_add(lds_at(h & 0xFFFCu), alignbyte(one, one, h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        uint32_t hot0 = 0, hot1 = 0;
#pragma unroll
                        for (int kk = 0; kk < 8; ++kk) { hot0 |= old[kk]; hot1 |= old[8 + kk]; }
                        if (hot0 & 0x80808080u) // some counter of a touched dword is >= 128: look precisely (rare)
                            t8_check<0>(em, c32, cx, f[2 * j], old[0], old[1], old[2], old[3], old[4], old[5], old[6], old[7]);
                        if (hot1 & 0x80808080u)
                            t8_check<8>(em, c32, cx, f[2 * j + 1], old[8], old[9], old[10], old[11], old[12], old[13], old[14], old[15]);
                    }
                }
            }
            // ---- triplets in cycle space (single-operation CIGAR: chromPos = pos + i)
            if ((parts & 4u) && __ballot(m0 & KM_TRIP)) {
                // The reference converts the BAM-orientation char to Dna (anything but A/C/G/T -> A); in the cycle space of a
                // reverse read that 'A' is the complement's code 3.  (For 8-mers the conversion comes after the reverse complement.)
                const uint32_t rcm = rc ? 0x33333333u : 0u;
                uint32_t I[KS_NH + 2], bad[KS_NH + 2]; // [1 .. KS_NH] = this lane, [0] / [KS_NH + 1] = the last / first dword of its neighbours
#pragma unroll
                for (int h = 0; h < KS_NH; ++h) {
                    const uint32_t x = ~P[h].oh & M;
                    const uint32_t ct = cn[h] | ((x | (x << 1)) & rcm);
                    I[h + 1] = (E[h] & 0xCCCCCCCCu) | ct; // nibble = [r c]
                    const uint32_t t = I[h + 1] ^ (I[h + 1] >> 2);
                    bad[h + 1] = ((t | (t >> 1)) & M) | nb[h]; // as a flank: mismatch or N
                }
                bad[0] = lane_prev(bad[KS_NH]); bad[KS_NH + 1] = lane_next(bad[1]); // (cycle 0 / L-1 are never evaluated)
                I[0] = lane_prev(I[KS_NH]); I[KS_NH + 1] = lane_next(I[1]);          // (cross-lane: outside the divergent branch)
                // position range [ja, jb) of the read in cycles
                const uint32_t ja = (uint32_t)min(max((int32_t)(w5 & 0xFFu) - (int32_t)wnb, 0), KS_NB);
                const uint32_t jb = (uint32_t)min(max((int32_t)((w5 >> 8) & 0xFFu) - (int32_t)wnb, 0), KS_NB);
                uint32_t pa[KS_NH], pb[KS_NH];
                lut_nib(pa, LUT + KS_LUTW * ja);
                lut_nib(pb, LUT + KS_LUTW * jb);
                // quality 20..94 <=> (signed char)(q + 33) >= '5'; flags at the byte MSBs, then moved next to each other in pairs:
                // bytes 3 and 1 of qf[d] hold the flags of cycles 4d, 4d+1 and 4d+2, 4d+3 at their bits 7 and 3
                uint32_t qf[KS_ND];
#pragma unroll
                for (int d = 0; d < KS_ND; ++d) {
                    const uint32_t x = Q[d] & 0x7F7F7F7Fu;
                    const uint32_t fq = (x + 0x6C6C6C6Cu) & ~(x + 0x21212121u) & ~Q[d]; // >= 20, not >= 95, not >= 128
                    const uint32_t y = fq & 0x80808080u;
                    qf[d] = y | (y << 4);
                }
                uint32_t ok[KS_NH], okany = 0;
#pragma unroll
                for (int h = 0; h < KS_NH; ++h) {
                    const uint32_t fl = alignbit(bad[h], bad[h + 1], 4) | alignbit(bad[h + 1], bad[h + 2], 28);
                    ok[h] = P[h].oh & ~fl & pb[h] & ~pa[h] & (vperm(qf[2 * h], qf[2 * h + 1], 0x07050301u) >> 3); // nibble MSB -> LSB
                    okany |= ok[h];
                }
                if (okany) {
                    uint32_t* tbin = lds + KS_TRIP + ((rc ? 2u : 0u) + ((m0 & 0x40u) ? 0u : 1u)) * 256u; // fwd1st fwd2nd rev1st rev2nd
#pragma unroll
                    for (int h = 0; h < KS_NH; ++h) {
                        if (!ok[h]) continue;
                        const uint32_t SA = alignbit(I[h], I[h + 1], 6), SB = alignbit(I[h + 1], I[h + 2], 22);
#pragma unroll
                        for (int t = 0; t < 8; ++t) { // bin = c(j-1) r(j) c(j) r(j+1): 8 contiguous bits of the [r c] stream
                            const uint32_t ix = t < 6 ? bfe(SA, 20 - 4 * t, 8) : bfe(SB, 12 - 4 * (t - 6), 8);
                            if (ok[h] & (1u << (28 - 4 * t))) atomicAdd(tbin + ix, 1u);
                        }
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
                const uint32_t L = (m0 >> 20) & 0xFFu, sum = WM[ln * KS_MW + 1];
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
    if (threadIdx.x == 0) t8_directory(t8_used + blockIdx.x * BQC_T8_USED, t8_n, t8_tags); // rows of this workgroup's slot that k_t8_fold has to read
}


// Dna5 bytes -> reference table of the fast path: one nibble  r1 r0 ~r0 ~r1  per base (r = Dna5 code & 3, i.e. N -> A like
// Dna5 -> Dna), 8 bases per dword, first base in the top nibble; bases 0.. start at dword BQC_FAST_NH (that many zero dwords in
// front, at least as many behind): nd8 + 2 * BQC_FAST_NH dwords for nd8 = ceil(len / 8).  Bit-reversing a dword yields the
// reverse complement.
__global__ void k_ref_nibbles(const uint8_t* __restrict__ dna5, uint64_t len, uint32_t* __restrict__ out, uint64_t nd8)
{
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nd8 + 2 * KS_NH) return;
    uint32_t v = 0;
    if (d >= KS_NH)
        for (uint32_t t = 0; t < 8; ++t) {
            const uint64_t p = (d - KS_NH) * 8 + t;
            if (p < len) {
                const uint32_t r = dna5[p] & 3u;
                v |= ((r << 2) | ((~r & 1u) << 1) | ((~r >> 1) & 1u)) << (28u - 4u * t);
            }
        }
    out[d] = v;
}

extern "C" void bqc_launch_ref_nibbles(const uint8_t* dna5, uint64_t len, uint32_t* out, uint64_t nd8, hipStream_t s)
{
    hipLaunchKernelGGL(k_ref_nibbles, dim3((uint32_t)((nd8 + 2 * KS_NH + 255) / 256)), dim3(256), 0, s, dna5, len, out, nd8);
}

// BQC_SHORT_PARTS: ablation switch for profiling (1 cycles + per-read sums, 2 8-mers, 4 triplets, 8 per-read statistics;
// without 8 the caller runs k_reads over the fast chunks instead)
extern "C" uint32_t bqc_short_parts()
{
    static uint32_t parts = 0xFFFFFFFFu;
    if (parts == 0xFFFFFFFFu) {
        const char* e = getenv("BQC_SHORT_PARTS");
        parts = e ? (uint32_t)atoi(e) : 15u;
        const char* x = getenv("BQC_SHORT_XCD"); // 1: an XCD's workgroups take a contiguous eighth of the chunks (see k_short)
        if (x && atoi(x) != 0) parts |= 16u;
    }
    return parts;
}

extern "C" hipError_t bqc_short_init()
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_short), hipFuncAttributeMaxDynamicSharedMemorySize, KS_WORDS * 4);
}

// Sum the packed rows the workgroups of one or more k_short launches have written (slot s: rows s * BQC_T8_SPW .. + used[s]) into
// the 8-mer counters of the state vector: thread per LDS dword (4 bins) and slice of 64 slots.  Rows are read once and not
// written: the next launches overwrite them.
__global__ __launch_bounds__(256) void k_t8_fold(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ used, uint32_t n_slots,
                                                    uint64_t* __restrict__ em, uint32_t lane)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; // < 16384
    uint32_t A = 0, B = 0, C = 0, D = 0; // bytes 0..3 of the dword = bins 4i + (0, 3, 2, 1); at most 64 * BQC_T8_SPW * 255 each
    const uint32_t s0 = blockIdx.y * 64u, s1 = min(n_slots, s0 + 64u); // (few slices: every slice ends in 65 536 x 4 global atomics)
    for (uint32_t sidx = s0; sidx < s1; ++sidx) {
        const uint32_t n = min(used[sidx * BQC_T8_USED], (uint32_t)BQC_T8_SPW);
        const uint64_t tags = (uint64_t)used[sidx * BQC_T8_USED + 1] | ((uint64_t)used[sidx * BQC_T8_USED + 2] << 32);
        const uint32_t* slot = rows + (size_t)sidx * BQC_T8_SPW * 16384u + i;
        uint32_t v[BQC_T8_SPW];
#pragma unroll
        for (uint32_t f = 0; f < BQC_T8_SPW; ++f) v[f] = f < n && ((uint32_t)(tags >> (8u * f)) & 0xFFu) == lane ? slot[(size_t)f * 16384u] : 0u; // independent loads, issued together
#pragma unroll
        for (uint32_t f = 0; f < BQC_T8_SPW; ++f) { A += v[f] & 0xFFu; B += (v[f] >> 8) & 0xFFu; C += (v[f] >> 16) & 0xFFu; D += v[f] >> 24; }
    }
    if (A) gadd(em + 4u * i + 0u, A);
    if (B) gadd(em + 4u * i + 3u, B);
    if (C) gadd(em + 4u * i + 2u, C);
    if (D) gadd(em + 4u * i + 1u, D);
}

extern "C" void bqc_launch_short(const DevBatch& b, const StateLayout& sl, uint64_t* state, const DevRefs& refs, uint32_t* err,
                                 uint32_t grid, uint32_t* t8rows, uint32_t* t8_used, hipStream_t s)
{
    if (grid == 0) return;
    static const hipError_t attr_once = bqc_short_init(); // (at the first launch, not in bqc_create: the call loads the code object — 40-60 ms that
    (void)attr_once;                                      //  the compute stream, which has slack at a run's start, can take; bqc_create's caller cannot)
    static uint32_t env_period = 0xFFFFFFFFu;
    if (env_period == 0xFFFFFFFFu) { const char* e = getenv("BQC_T8_PERIOD"); env_period = e && atoi(e) > 0 ? (uint32_t)atoi(e) : 0u; } // tuning knob
    hipLaunchKernelGGL(k_short, dim3(grid), dim3(KS_THREADS), KS_WORDS * 4, s, b, sl, state, refs, err, bqc_short_parts(), (uint4*)t8rows, t8_used,
                       env_period ? env_period : KS_T8_PERIOD);
}

// sum the rows of `n_slots` workgroup slots that belong to read group `lane` into its 8-mer counters
extern "C" void bqc_launch_t8_fold(const uint32_t* t8rows, const uint32_t* t8_used, uint32_t n_slots, const StateLayout& sl, uint64_t* state, uint32_t lane,
                                   hipStream_t s)
{
    if (!n_slots) return;
    hipLaunchKernelGGL(k_t8_fold, dim3(64, (n_slots + 63) / 64), dim3(256), 0, s, t8rows, t8_used, n_slots, state + sl.lane_base(lane) + sl.o_eightmer, lane);
}
