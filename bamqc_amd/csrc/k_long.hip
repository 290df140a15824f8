// k_long.hip — per-base statistics for reads that do not fit the short-read fast path (longer than 255 bases; every read
// when BQC_NO_FAST=1), any read length, on the register machinery of k_short (swar.h).
//
// A wave handles one ROW of one read at a time: 992 consecutive sequencing cycles, 16 per lane for the lanes 1..62; lane 0
// holds the 16 cycles in front of the row and lane 63 the 16 behind it (needed for the 8-mer windows that run out of the row
// and the flanking bases of a triplet; what those two lanes hold is counted by the neighbouring rows).  The per-cycle
// histograms of a 10 kb read (2 mates x 6 counters x 10 000 cycles) do not fit in LDS, so workgroup (x, y) walks the
// chunks x, x + gridDim.x, ... and handles row y of every read — for a reverse-strand read that is the mirrored base
// range, loaded mirrored and turned in registers.  Per lane, as in k_short: 12 B of bases and 16 B of qualities straight
// from global memory, four one-hot planes added into vertical 4-bit counters (spilled to the LDS cycle tile every 15 rows),
// the 16 8-mer windows by funnel shifts into packed u8 LDS counters, triplets as nibble-flag SWAR.  The CIGAR is turned into
// its match-like segments ONCE per row by the whole wave (operation k in lane k, prefix sums by DPP scans); every segment
// that touches the row is one triplet pass with its own alignment offset.  Per-read sums (quality, N, GC) are combined
// across rows through a small per-read scratch array and turned into histograms by k_long_finish.
//
// Reference: QualityCheck.hpp:122-166 (read_counts), OverallNumbers.hpp:137-168 (count8mers),
// TripletCounting.hpp:195-236 (countBasesInTriplets).
#include "kernels_common.h"
#include "swar.h"

#define KL_ROW    992                              // cycles a row owns (lanes 1..62)
#define KL_T8     0                                // 16384 words: 65536 packed u8 8-mer counters (LDS address 0: see k_short)
#define KL_TRIP   16384                            // [4 groups][256] bins c(j-1) r(j) c(j) r(j+1) in cycle space
#define KL_CYC    (KL_TRIP + 1024)                 // [2 mates][A C G T other qual][1024]: cycle t of the row (lane w = 1 + t / 16) at (t % 16) * 64 + w
#define KL_LUT    (KL_CYC + 2 * 6 * 1024)          // [17][8] masks for "the first n of the lane's cycles" (k_short's table)
#define KL_WORDS  (KL_LUT + 17 * 8)
#ifndef KL_WAVES
#define KL_WAVES  12
#endif
#ifndef KL_EXPER
#define KL_EXPER  0                                // timing experiments only (results are wrong): 1 no triplets, 2 no 8-mers, 4 no per-cycle counters / per-read sums
#endif

// The packed 8-mer counters go into the next free row of the workgroup's slot of the scratch table k_short's workgroups use (plain
// 16-byte stores of the LDS image; k_t8_fold sums the rows of all slots into the state vector later) while the slot has rows left; with
// global atomics after that.  Round 4: with atomics only, every workgroup ended in 65 536 scattered 8-byte
// atomic adds — 16.6 M per launch on config 5, 0.3 of the kernel's 1.4 ms.  Returns true when a row was written.
// The per-cycle tile likewise: its first flush for that read group goes, as it is, into the workgroup's place of a scratch array that
// k_long_cyc_fold sums over the workgroups of a row (they all count the same cycles: 23 workgroups' atomics on the same words).
__device__ __forceinline__ bool kl_flush(uint32_t* lds, const StateLayout& sl, uint64_t* __restrict__ state, uint32_t lane, uint32_t cyc0,
                                         uint4* __restrict__ slot /* nullptr: atomics */, uint32_t n_rows_used, uint4* __restrict__ cyc_tile /* nullptr: atomics */)
{
    const uint64_t lb = sl.lane_base(lane);
    const bool to_row = slot && n_rows_used < BQC_T8_SPW;
    if (to_row) {
        uint4* row = slot + (size_t)n_rows_used * 4096u;
        uint4* src = (uint4*)(lds + KL_T8);
        for (uint32_t i = threadIdx.x; i < 4096u; i += blockDim.x) { row[i] = src[i]; src[i] = make_uint4(0, 0, 0, 0); }
    } else t8_atomics_out(lds + KL_T8, state + lb + sl.o_eightmer);
    for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) { // bin = c(j-1) r(j) c(j) r(j+1) of group i >> 8, in cycle space
        const uint32_t v = lds[KL_TRIP + i];
        if (!v) continue;
        lds[KL_TRIP + i] = 0;
        const uint32_t grp = i >> 8, f3 = (i >> 6) & 3u, f2 = (i >> 4) & 3u, f1 = (i >> 2) & 3u, f0 = i & 3u;
        uint32_t ctx, base;
        if (grp < 2u) { ctx = (f3 << 4) | (f2 << 2) | f0; base = f1; }                               // forward: as is
        else { ctx = ((3u - f0) << 4) | ((3u - f2) << 2) | (3u - f3); base = 3u - f1; }                // reverse: complement, mirrored
        gadd(state + lb + sl.o_triplet + ctx * 16u + grp * 4u + base, v);
    }
    if (cyc_tile) {
        uint4* src = (uint4*)(lds + KL_CYC);
        for (uint32_t i = threadIdx.x; i < 2 * 6 * 1024 / 4; i += blockDim.x) { cyc_tile[i] = src[i]; src[i] = make_uint4(0, 0, 0, 0); }
    } else
    for (uint32_t i = threadIdx.x; i < ((KL_EXPER & 256) ? 0 : 2 * 6 * 1024); i += blockDim.x) { // (256: timing without this flush)
        const uint32_t v = lds[KL_CYC + i];
        if (!v) continue;
        lds[KL_CYC + i] = 0;
        const uint32_t m = i / (6 * 1024), c = (i / 1024) % 6, jj = i % 1024, w = jj % 64, t = jj / 64;
        if (w == 0 || w == 63) continue; // (never written)
        const uint32_t j = cyc0 + 16u * (w - 1u) + t;
        if (j < sl.lcap) gadd(state + sl.mate_base(lane, m) + (c < 5 ? sl.m_dnacount + c * sl.lcap : sl.m_qualcount) + j, v);
    }
    return to_row;
}

// Per-cycle accumulators of one lane (its 16 cycles), for the reads of ONE mate: 4-bit vertical counters per base plane
// and packed 16-bit quality sums (k_short's CycAcc for two 8-cycle halves).
struct KlAcc { uint32_t l1[2][4]; uint32_t qo[4], qe[4]; };
__device__ __forceinline__ void kl_zero(KlAcc& A)
{
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int p = 0; p < 4; ++p) A.l1[h][p] = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) { A.qo[d] = 0; A.qe[d] = 0; }
}
// (the tiles are addressed by their LDS byte address — the dynamic LDS starts at address 0, checked at the kernel's start —: as
// generic pointers the seven tile addresses a lane keeps were 14 registers, all of them spilled)
template <uint32_t TILE /* word index of tile + 64 * 8 * half */> __device__ __noinline__ void kl_spill_half(uint32_t a, uint32_t c, uint32_t g, uint32_t t, uint32_t w)
{
    const uint32_t v[4] = {a, c, g, t};
    const uint32_t base = 4u * (TILE + w); // (computed here: kept by the caller, the compiler hoisted the eight addresses out of the read loop and spilled them)
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int n = 0; n < 8; ++n) __hip_atomic_fetch_add(lds_at(base + 4u * (p * 1024 + 64 * (7 - n))), (v[p] >> (4 * n)) & 15u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <uint32_t TILE /* word index of tile + 5 * 1024 + 64 * 8 * half */> __device__ __noinline__ void kl_qflush_pair(uint32_t o0, uint32_t e0, uint32_t o1, uint32_t e1, uint32_t w)
{
    const uint32_t vo[2] = {o0, o1}, ve[2] = {e0, e1};
    const uint32_t base = 4u * (TILE + w);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        __hip_atomic_fetch_add(lds_at(base + 4u * (64 * (4 * d + 0))), vo[d] >> 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(lds_at(base + 4u * (64 * (4 * d + 1))), ve[d] >> 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(lds_at(base + 4u * (64 * (4 * d + 2))), vo[d] & 0xFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(lds_at(base + 4u * (64 * (4 * d + 3))), ve[d] & 0xFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
template <uint32_t MATE> __device__ __forceinline__ void kl_spill(KlAcc& A, uint32_t w)
{
    kl_spill_half<KL_CYC + MATE * 6 * 1024>(A.l1[0][0], A.l1[0][1], A.l1[0][2], A.l1[0][3], w);
    kl_spill_half<KL_CYC + MATE * 6 * 1024 + 64 * 8>(A.l1[1][0], A.l1[1][1], A.l1[1][2], A.l1[1][3], w);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int p = 0; p < 4; ++p) A.l1[h][p] = 0;
}
template <uint32_t MATE> __device__ __forceinline__ void kl_qflush(KlAcc& A, uint32_t w)
{
    kl_qflush_pair<KL_CYC + (MATE * 6 + 5) * 1024>(A.qo[0], A.qe[0], A.qo[1], A.qe[1], w);
    kl_qflush_pair<KL_CYC + (MATE * 6 + 5) * 1024 + 64 * 8>(A.qo[2], A.qe[2], A.qo[3], A.qe[3], w);
#pragma unroll
    for (int d = 0; d < 4; ++d) { A.qo[d] = 0; A.qe[d] = 0; }
}
__device__ __forceinline__ void kl_add(KlAcc& A, const Planes (&P)[2], const uint32_t (&Q)[4])
{
#pragma unroll
    for (int h = 0; h < 2; ++h) { A.l1[h][0] += P[h].a; A.l1[h][1] += P[h].c; A.l1[h][2] += P[h].g; A.l1[h][3] += P[h].t; }
#pragma unroll
    for (int d = 0; d < 4; ++d) { A.qe[d] += Q[d] & 0x00FF00FFu; A.qo[d] += __builtin_amdgcn_perm(0u, Q[d], 0x0C030C01u); }
}
__device__ __forceinline__ void kl_lut_nib(uint32_t (&d)[2], const uint32_t* e) { const uint2 v = *(const uint2*)e; d[0] = v.x; d[1] = v.y; }
__device__ __forceinline__ void kl_lut_byte(uint32_t (&d)[4], const uint32_t* e) { const uint4 v = *(const uint4*)e; d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }

__global__ __launch_bounds__(KL_WAVES * 64) void k_long(DevBatch b, StateLayout sl, uint64_t* __restrict__ state, DevRefs refs,
                                                            uint32_t* __restrict__ err, uint32_t* __restrict__ rsum /* [n_reads][3] */,
                                                            uint4* __restrict__ t8rows, uint32_t* __restrict__ t8_used, uint32_t t8_lane,
                                                            uint4* __restrict__ cyc_tiles, uint32_t* __restrict__ cyc_used)
{
    extern __shared__ uint32_t lds[];
    const uint32_t wg = blockIdx.y * gridDim.x + blockIdx.x; // this workgroup's slot of the scratch rows
    if ((uint32_t)(uintptr_t)(lds_u32*)lds != 0u) { // the 8-mer atomics address LDS directly (KL_T8 at LDS address 0)
        if (threadIdx.x == 0) { atomicOr(err, BQC_DEVERR_INTERNAL); t8_used[wg * BQC_T8_USED] = 0; cyc_used[wg] = 0; }
        return;
    }
    const uint32_t cyc0 = blockIdx.y * KL_ROW;
    if (cyc0 >= b.desc->long_max_len) { // no read of the batch reaches this row (the grid is sized from an upper bound)
        if (threadIdx.x == 0) { t8_used[wg * BQC_T8_USED] = 0; cyc_used[wg] = 0; }
        return;
    }
    uint4* const t8_slot = t8rows + (size_t)wg * BQC_T8_SPW * 4096u;
    uint32_t t8_n = 0; // rows written so far
    T8Tags t8_tags;    // ... and their read groups
    uint4* const cyc_tile = cyc_tiles + (size_t)wg * (2 * 6 * 1024 / 4);
    bool cyc_written = false;
    for (uint32_t i = threadIdx.x; i < KL_WORDS; i += blockDim.x) lds[i] = 0;
    block_sync();
    const uint32_t M = 0x11111111u;
    const uint32_t ln = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t w = ln;                         // lane w holds the cycles cyc0 - 16 + 16 w .. + 15
    const bool own = w >= 1u && w <= 62u;          // ... and counts them, unless it is one of the two flank lanes
    const uint32_t* LUT = lds + KL_LUT;
    if (threadIdx.x <= 16u) { // masks for "the first n of the lane's cycles" (n nibbles from the top / n bytes from the top)
        const uint32_t n = threadIdx.x;
        uint32_t* e = lds + KL_LUT + 8u * n;
        for (uint32_t h = 0; h < 2; ++h) {
            const uint32_t v = n > 8u * h ? (n - 8u * h < 8u ? n - 8u * h : 8u) : 0u;
            e[h] = v ? 0xFFFFFFFFu << (4u * (8u - v)) : 0u;
        }
        for (uint32_t d = 0; d < 4; ++d) {
            const uint32_t v = n > 4u * d ? (n - 4u * d < 4u ? n - 4u * d : 4u) : 0u;
            e[4 + d] = v ? 0xFFFFFFFFu << (8u * (4u - v)) : 0u;
        }
    }
    block_sync();
    KlAcc A0, A1; // per-cycle accumulators of this lane, first / second mate
    kl_zero(A0); kl_zero(A1);
    uint32_t n1[2] = {0, 0}, n2[2] = {0, 0}; // rows since the last counter spill / quality flush, per mate (wave-uniform)
    uint32_t cur_lane = 0xFFFFFFFFu;
    const int32_t c_lo = (int32_t)cyc0 - 16 + 16 * (int32_t)w; // this lane's first cycle
    const uint8_t* g_seq = b.seq;
    const uint8_t* g_qual = b.qual;
    const uint32_t n_chunks = b.desc->n_chunks_slow;
    for (uint32_t ci = blockIdx.x;; ci += gridDim.x) { // one extra pass at the end flushes the last lane
        const bool done = ci >= n_chunks;
        Chunk ch{0, 0, 0xFFFFFFFFu, 0, 0, 0, 0, 0};
        if (!done) ch = b.chunks[ci];
        if (ch.lane != cur_lane) { // block-uniform
            if (cur_lane != 0xFFFFFFFFu) {
                if (own) { kl_spill<0>(A0, w); kl_qflush<0>(A0, w); kl_spill<1>(A1, w); kl_qflush<1>(A1, w); }
                n1[0] = n1[1] = n2[0] = n2[1] = 0;
                block_sync();
                const bool tile_out = cur_lane == t8_lane && !cyc_written;
                if (kl_flush(lds, sl, state, cur_lane, cyc0, t8_slot, t8_n, tile_out ? cyc_tile : nullptr)) { t8_tag(t8_tags, t8_n, cur_lane); ++t8_n; }
                cyc_written |= tile_out;
                block_sync();
            }
            cur_lane = ch.lane;
        }
        if (done) break;
        uint64_t* em = state + sl.lane_base(cur_lane) + sl.o_eightmer;
        // The reads of the chunk this wave takes (k = wave, wave + 16, ...) go through a two-deep pipeline, so that a row's
        // instructions run while the next rows' loads are in flight: the columns of read k + 16 and the bases / qualities of its
        // row are requested before read k is computed (a wave alone would wait for four dependent round trips per row).
        struct Meta { uint32_t r, flag, L, so, qo, co, ncig; int32_t rid, pos; };
        struct RefOf { const uint32_t* rn; uint64_t len; }; // the read's contig (nibble table, length): asked for with the row's bases — looked up
                                                            // where it is needed it cost every row two dependent round trips behind a vmcnt(0)
        auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }; // the same in every lane: keep it in a scalar register
        // (the read's index is asked for an iteration before its columns are: read where it is needed it was a round trip of its own
        // per row, behind a vmcnt(0) that also waited for the row loads issued just before)
        auto load_perm = [&](uint32_t kk) { return kk < ch.count ? b.perm[ch.first + kk] : 0u; };
        auto load_meta = [&](uint32_t kk, uint32_t r_raw) {
            Meta m{0, 0x900u, 0, 0, 0, 0, 0, -1, 0};
            if (kk < ch.count) {
                m.r = uni(r_raw);
                m.flag = b.flag[m.r]; m.L = b.l_seq[m.r]; m.so = b.seq_off[m.r]; m.qo = b.qual_off[m.r];
                m.co = b.cigar_off[m.r]; m.ncig = b.n_cigar[m.r]; m.rid = b.rid[m.r]; m.pos = b.pos[m.r];
            }
            return m;
        };
        auto settle = [&](Meta& m) { // (called when the values are about to be used: the loads have had an iteration to arrive)
            m.flag = uni(m.flag); m.L = uni(m.L); m.so = uni(m.so); m.qo = uni(m.qo); m.co = uni(m.co); m.ncig = uni(m.ncig);
            m.rid = (int32_t)uni((uint32_t)m.rid); m.pos = (int32_t)uni((uint32_t)m.pos);
        };
        // bases / qualities of the row, and the read's first 64 CIGAR operations (operation ln in lane ln)
        auto load_row = [&](const Meta& m, uint32_t (&s)[3], uint32_t (&q)[4], uint32_t& cw, RefOf& ro) { // (rows that are skipped load harmlessly from the buffers' start)
            const bool rc_ = m.flag & 0x10u;
            const bool use = !(m.flag & 0x900u) && (m.flag & 0xC0u) && m.L > cyc0;
            const int32_t o0_ = rc_ ? (int32_t)m.L - (int32_t)cyc0 : (int32_t)cyc0 - 16;
            const int32_t sw_ = rc_ ? -(int32_t)w : (int32_t)w;
            // the window is loaded from one byte (odd o0: one nibble) earlier; lanes whose window lies outside the read load
            // neighbouring data (the buffers are padded) and mask it
            const int64_t so = use ? (int64_t)m.so + ((o0_ - 1) >> 1) + 8 * sw_ : 0;
            const int64_t qo = use ? (int64_t)((m.flag & BQC_FLAG_NO_QUAL) ? 0u : m.qo) + o0_ + 16 * sw_ : 0;
            GVec<3>::ldu(s, g_seq + so);
            GVec<4>::ldu(q, g_qual + qo);
            cw = use && ln < m.ncig ? b.cigar[(uint64_t)m.co + ln] : 0u;
            const bool hasref = use && m.rid >= 0 && (uint32_t)m.rid < refs.n_refs;
            ro.rn = hasref ? refs.refn[m.rid] : nullptr;
            ro.len = hasref ? refs.len[m.rid] : 0;
        };
        const uint32_t r_0 = load_perm(wave), r_1 = load_perm(wave + KL_WAVES);
        uint32_t r_nn = load_perm(wave + 2 * KL_WAVES);
        Meta cur_m = load_meta(wave, r_0), nxt_m = load_meta(wave + KL_WAVES, r_1);
        uint32_t s[3], q[4], cw, s_n[3], q_n[4], cw_n;
        RefOf ro, ro_n;
        settle(cur_m);
        load_row(cur_m, s, q, cw, ro);
        // Every load of the prologue has arrived before the loop is entered: the compiler's wait-count pass otherwise carries "pending
        // since the prologue, three loads behind it" into the loop and waits — in every iteration, in the middle of the row, for all but
        // the last three of the loads the iteration has just issued (s_waitcnt vmcnt(3) in front of the first use of the contig's length).
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
        for (uint32_t k = wave; k < ch.count; k += KL_WAVES) {
            settle(nxt_m);                                  // (its columns were requested one iteration ago)
            load_row(nxt_m, s_n, q_n, cw_n, ro_n);
            const Meta nn_m = load_meta(k + 2 * KL_WAVES, r_nn);
            r_nn = load_perm(k + 3 * KL_WAVES);
            {
            const uint32_t r = cur_m.r, flag = cur_m.flag;
            if ((flag & 0x900u) || !(flag & 0xC0u)) goto next_read;  // skipped records; a missing mate flag is raised by the pre-pass / k_reads
            const uint32_t L = cur_m.L;
            if (L <= cyc0) goto next_read;                            // no base of this read falls into this row
            const uint32_t mate = (flag & 0x40u) ? 0u : 1u;
            const bool rc = flag & 0x10u, noq = flag & BQC_FLAG_NO_QUAL;
            // BAM index of the first base of lane 0's window, and of this lane's: forward reads run with the cycles, reverse reads
            // against them (lane w then takes the mirrored window and turns it in registers)
            const int32_t o0 = rc ? (int32_t)L - (int32_t)cyc0 : (int32_t)cyc0 - 16;
            const uint32_t nv = c_lo < 0 ? 0u : (uint32_t)min(max((int32_t)L - c_lo, 0), 16); // valid cycles of this lane
            const uint32_t nvq = noq ? 0u : nv;
            uint32_t xm[2], qm[4];
            kl_lut_nib(xm, LUT + 8u * nv);
            kl_lut_byte(qm, LUT + 8u * nvq + 4u);
            // ---- triplets in cycle space (TripletCounting.hpp:195-236), first half: the CIGAR's segments and the reference windows they
            // need are worked out BEFORE the row's own arithmetic, which then runs while the windows are on their way
            const uint32_t ncig = cur_m.ncig;
            const int32_t rid = cur_m.rid;
            const uint32_t* rn = (const uint32_t*)(((uint64_t)uni((uint32_t)((uint64_t)ro.rn >> 32)) << 32) | uni((uint32_t)(uint64_t)ro.rn)); // (nullptr: no such contig, or not loaded)
            const int64_t reflen = (int64_t)(((uint64_t)uni((uint32_t)(ro.len >> 32)) << 32) | uni((uint32_t)ro.len));
            const bool trip = !(KL_EXPER & 1) && (flag & BQC_FLAG_TRIPLET) && L >= 3 && ncig > 0 && !noq && rid >= 0 && rn != nullptr;
            const uint32_t* __restrict__ cg = b.cigar + cur_m.co;
            const uint32_t nd8p1 = (uint32_t)((reflen + 7) >> 3) + 1u; // (the table: 2 pad dwords, nd8, 2 pad dwords)
            const int64_t pos0 = cur_m.pos;
            // the row's bases in BAM orientation (flank lanes included: their cycles are never evaluated, only looked at)
            const int64_t I0 = rc ? (int64_t)L - cyc0 - KL_ROW : (int64_t)cyc0, I1 = I0 + KL_ROW;
            const uint32_t w16 = 16u * w;
            // Round 4: one triplet pass per LANE, not per segment.  A match-like CIGAR segment aligns read positions [ia, ib) at
            // chromPos = posv + i; the lanes holding the operations clip it to the contig and the row and turn it into CYCLES of this
            // row (+ 16: lane w holds 16 w .. 16 w + 15) and the reference nibble index of lane 0's window; every cycle lane then
            // picks the (at most two, nearly always) segments that touch its 16 cycles and evaluates each with its own alignment —
            // two passes per row whatever the number of segments (round 3: a pass of the whole wave per segment, three per row on
            // config 5's reads), both reference windows in flight together, 32-bit arithmetic throughout the pass.  The window
            // has 18 nibbles now (the reference bases in front of the lane's first cycle and behind its last one under THIS
            // lane's alignment; the neighbour may sit in another segment).
            auto seg_pack = [&](bool valid, int64_t ia, int64_t ib, int64_t posv, uint32_t& kab_s, uint32_t& U_s) {
                if (1 - posv > ia) ia = 1 - posv;                       // context posv+i-1 .. posv+i+1 inside the contig
                if (reflen - 1 - posv < ib) ib = reflen - 1 - posv;
                valid = valid && ia < ib && ia < I1 && ib > I0;
                int64_t ka = (rc ? (int64_t)L - ib : ia) - (int64_t)cyc0 + 16, kb = (rc ? (int64_t)L - ia : ib) - (int64_t)cyc0 + 16;
                ka = ka < 0 ? 0 : ka > 1024 ? 1024 : ka;
                kb = kb < 0 ? 0 : kb > 1024 ? 1024 : kb;
                kab_s = (uint32_t)ka | ((uint32_t)kb << 16);
                U_s = (uint32_t)(uint64_t)(posv + o0 + 15);             // (the table has 16 pad nibbles in front; the window starts one nibble early)
                return valid;
            };
            // The walk of :207-222, by the whole wave: operation kb + ln in lane ln.  Read / chromosome advance of every operation
            // (the first one is taken as match-like whatever it is), exclusive prefix sums by DPP scans (the chromosome advance as two
            // 16-bit halves: 64 operations of up to 2^28 positions); the match-like ones that reach into the row are the result.
            int64_t rp_c = 0, cp_c = 0; // positions in front of the current block of 64 operations
            auto scan_block = [&](uint32_t kb, uint32_t& kab_s, uint32_t& U_s) {
                const uint32_t kk = kb + ln;
                const uint32_t wv = kb == 0u ? cw : (kk < ncig ? cg[kk] : 0u), op = wv & 15u, nn = wv >> 4;
                const bool live = kk < ncig;
                const bool ref_only = live && kk != 0u && (op == 2u || op == 3u || op == 5u || op == 6u); // D N H P
                const bool read_only = live && kk != 0u && (op == 4u || op == 1u);                         // S I
                const bool match = live && !ref_only && !read_only;
                const uint32_t ra = (live && !ref_only) ? min(nn, L + 1u) : 0u, ca = (live && !read_only) ? nn : 0u;
                const uint32_t ra_i = wave_scan_incl(ra), cl_i = wave_scan_incl(ca & 0xFFFFu), ch_i = wave_scan_incl(ca >> 16);
                const int64_t rp = rp_c + (ra_i - ra), cp = cp_c + (int64_t)(cl_i - (ca & 0xFFFFu)) + ((int64_t)(ch_i - (ca >> 16)) << 16);
                // segment of a match-like operation: read positions [ia, ib) at chromPos = posv + i
                const int64_t ia = rp > 1 ? rp : 1, ib = rp + nn < (int64_t)L - 1 ? rp + nn : (int64_t)L - 1;
                const int64_t posv = pos0 + cp - rp;
                const bool seg = seg_pack(match && rp < (int64_t)L && ia < ib && posv > INT32_MIN / 2 && posv < INT32_MAX / 2, ia, ib, posv, kab_s, U_s);
                rp_c += (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)ra_i, 63);
                cp_c += (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)cl_i, 63) + ((int64_t)(uint32_t)__builtin_amdgcn_readlane((int)ch_i, 63) << 16);
                return (uint64_t)__ballot(seg);
            };
            // the segments `t` (operation lanes) that touch this lane's cycles, at most two per call, in operation order from
            // `wm` on; returns the ones some lane had no room for
            auto collect = [&](uint64_t t, uint32_t kab_s, uint32_t U_s, uint32_t& wm, uint32_t (&kab)[2], uint32_t (&U)[2], uint32_t& have) {
                kab[0] = kab[1] = 0; U[0] = U[1] = 0; have = 0;
                uint64_t rest = 0;
                while (t) {
                    const uint32_t src = (uint32_t)__ffsll((unsigned long long)t) - 1u;
                    t &= t - 1;
                    const uint32_t K = (uint32_t)__builtin_amdgcn_readlane((int)kab_s, (int)src), UU = (uint32_t)__builtin_amdgcn_readlane((int)U_s, (int)src);
                    const bool ov = (K & 0xFFFFu) < w16 + 16u && (K >> 16) > w16 && src >= wm;
                    const bool take = ov && have < 2u;
                    if (take) {
                        if (have == 0u) { kab[0] = K; U[0] = UU; } else { kab[1] = K; U[1] = UU; }
                        ++have; wm = src + 1u;
                    }
                    if (__ballot(ov && !take)) rest |= 1ull << src;
                }
                return rest;
            };
            auto issue = [&](uint32_t U, uint32_t& pp, uint32_t (&e)[4]) {
                pp = rc ? U - w16 : U + w16;                            // nibble index of the lane's window, minus one (a lane that takes the segment: 0 <= pp < reflen + 16)
                const uint32_t di = min(pp >> 3, nd8p1);
                GVec<3>::lda(e, (const uint8_t*)(rn + di));
                e[3] = KS_GLOBAL(uint32_t, rn + min(di + 3u, nd8p1 + 2u)); // (the nibble behind the window when it starts at a dword's last nibble; said to be
                                                                            //  global: a plain load through this pointer — itself loaded — is a FLAT load)
            };
            // the first 64 operations' segments, the first two per lane: requested here, evaluated at the end
            uint32_t t_kab_s = 0, t_U_s = 0, t_wm = 0, t_kab[2] = {0, 0}, t_pp[2] = {0, 0}, t_e0[4] = {0, 0, 0, 0}, t_e1[4] = {0, 0, 0, 0};
            uint64_t t_todo = 0;
            bool t_two = false, t_any = false;
            if (trip) {
                const uint32_t n0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cw) >> 4; // operation 0 sits in lane 0
                if (n0 == 0u) {
                    // cigarCount starts at (size_t)-1: every position counts as inside the first operation (:203)
                    t_todo = seg_pack(pos0 > -(1 << 30) && pos0 < (1 << 30), 1, (int64_t)L - 1, pos0, t_kab_s, t_U_s) ? 1ull : 0ull; // (the same in every lane)
                    rp_c = (int64_t)L; // (no further operations)
                } else t_todo = scan_block(0, t_kab_s, t_U_s);
                if (t_todo) {
                    uint32_t U[2], have;
                    t_todo = collect(t_todo, t_kab_s, t_U_s, t_wm, t_kab, U, have);
                    t_two = !(KL_EXPER & 64) && __ballot(have >= 2u) != 0; // (64: no second pass)
                    t_any = true;
                    issue(U[0], t_pp[0], t_e0);
                    if (t_two) issue(U[1], t_pp[1], t_e1);
                }
            }
            // ---------------- the lane's cycles: one-hot base nibbles X[h] = cycles 8h..8h+7 (first cycle in the top nibble) ...
            uint32_t X[2];
            {
                const uint32_t sh = (o0 & 1) ? 28u : 24u; // the loaded bytes start 1 / 2 nibbles before the window
                uint32_t bs[3], F[2];
#pragma unroll
                for (int h = 0; h < 3; ++h) bs[h] = bswap32(s[h]);
#pragma unroll
                for (int h = 0; h < 2; ++h) F[h] = alignbit(bs[h], bs[h + 1], sh);
#pragma unroll
                for (int h = 0; h < 2; ++h) X[h] = (rc ? __brev(F[1 - h]) : F[h]) & xm[h]; // bit reversal = reversed order, complemented one-hot codes
            }
            // ... and qualities Q[d] = cycles 4d..4d+3, first cycle in the top byte
            uint32_t Q[4];
            {
                const uint32_t sel = rc ? 0x07060504u : 0x00010203u;
                uint32_t any = 0;
#pragma unroll
                for (int d = 0; d < 4; ++d) { Q[d] = vperm(q[3 - d], q[d], sel) & qm[d]; any |= Q[d]; }
                if (own && (any & 0x80808080u)) { // some Phred >= 128: check the 222 limit precisely
                    bool bad = false;
#pragma unroll
                    for (int d = 0; d < 4; ++d)
#pragma unroll
                        for (int k8 = 0; k8 < 4; ++k8) bad |= ((Q[d] >> (8 * k8)) & 0xFFu) > 222u;
                    if (bad) atomicOr(err, BQC_DEVERR_QUAL);
                }
            }
            Planes P[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) P[h] = planes_of(X[h]);
            // ---- 2-bit codes per nibble; non-ACGT -> A (char -> Dna after the reverse complement)
            uint32_t cn[2], nb[3];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                cn[h] = (P[h].c | P[h].t) | ((P[h].g | P[h].t) << 1);
                nb[h] = P[h].n | (~xm[h] & M); // literal N or outside the read: blocks 8-mer windows and triplet flanks
            }
            // ---- 8-mers: windows starting at the lane's cycles (OverallNumbers.hpp:137-168)
            uint32_t old[16], f[2], c32 = 0, cx = 0;
            const bool t8_go = !(KL_EXPER & 2) && own && nv > 0u;
            {
                uint32_t S[2];
                S[0] = vperm(squeeze2(cn[0]), squeeze2(cn[1]), 0x05040100u);
                S[1] = lane_next(S[0]);
                nb[2] = lane_next(nb[0]);
                if (w == 63u) nb[2] = M;
                {
                    uint32_t sa[3], sb[3];
#pragma unroll
                    for (int h = 0; h < 2; ++h) sa[h] = nb[h] | alignbit(nb[h], nb[h + 1], 28);
                    sa[2] = nb[2] | (nb[2] << 4);
#pragma unroll
                    for (int h = 0; h < 2; ++h) sb[h] = sa[h] | alignbit(sa[h], sa[h + 1], 24);
                    sb[2] = sa[2] | (sa[2] << 8);
#pragma unroll
                    for (int h = 0; h < 2; ++h) f[h] = ~(sb[h] | alignbit(sb[h], sb[h + 1], 16)); // nibble LSB set <=> the window starting there is counted
                }
                if (t8_go) { // all 16 atomics are issued, the returned values looked at further down; a blocked window adds 0
                    c32 = S[0]; cx = S[1];
#pragma unroll
                    for (int kw = 0; kw < 16; ++kw) {
                        const uint32_t h = kw < 8 ? c32 >> (16 - 2 * kw) : kw == 8 ? c32 : alignbit(c32, cx, 48 - 2 * kw);
                        const uint32_t one = bfe(f[kw >> 3], 28 - 4 * (kw & 7), 1);
#if KL_EXPER & 8   // the atomics without their returned values
                        old[kw] = 0; __hip_atomic_fetch_add(lds_at(h & 0xFFFCu), alignbyte(one, one, h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#elif KL_EXPER & 16 // the arithmetic without the atomics (and without the checks: the values never look hot)
                        old[kw] = ((h & 0xFFFCu) + alignbyte(one, one, h)) & 0x7F7F7F7Fu;
#elif KL_EXPER & 128 // neither: what is in front of the loop only
                        old[kw] = 0; if (kw == 0) old[0] = (c32 ^ cx ^ f[0] ^ f[1]) & 0x7F7F7F7Fu;
#else
                        old[kw] = __hip_atomic_fetch_add(lds_at(h & 0xFFFCu), alignbyte(one, one, h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
                    }
                }
            }
            // ---- per-cycle counters and the per-read sums (QualityCheck.hpp:122-166) — while the 8-mer counters' old values come back
            if (!(KL_EXPER & 4)) {
                if (own) { if (mate) kl_add(A1, P, Q); else kl_add(A0, P, Q); }
                if (++n1[mate] == 15u) { if (own) { if (mate) kl_spill<1>(A1, w); else kl_spill<0>(A0, w); } n1[mate] = 0; }
                if (++n2[mate] == 255u) { if (own) { if (mate) kl_qflush<1>(A1, w); else kl_qflush<0>(A0, w); } n2[mate] = 0; }
                uint32_t oany = 0, oth[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) { oth[h] = own ? xm[h] & M & ~P[h].oh : 0u; oany |= oth[h]; }
                if (oany) { // cycles holding anything but A/C/G/T (Dna5 'N' bin) are rare: counted directly
                    uint32_t* ob = lds + KL_CYC + (mate * 6 + 4) * 1024 + w;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        uint32_t o = oth[h];
                        while (o) { const uint32_t bit = (uint32_t)__ffs((int)o) - 1u; o &= o - 1u; atomicAdd(ob + 64u * (8u * h + 7u - (bit >> 2)), 1u); }
                    }
                }
                uint32_t v = 0, vn = 0, vgc = 0;
                if (own) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) v = __builtin_amdgcn_sad_u8(Q[d], 0u, v);
#pragma unroll
                    for (int h = 0; h < 2; ++h) { vn += (uint32_t)__popc(P[h].n); vgc += (uint32_t)__popc(P[h].c | P[h].g); }
                }
                // (sums over the wave by DPP adds, the total in lane 63: no LDS round trips; <= 62 * 4080 < 2^20, <= 992)
                const uint32_t s1 = wave_scan_incl(v | (vn << 20)), s2 = wave_scan_incl(vgc);
                if (ln == 63u) {
                    if (s1 & 0xFFFFFu) atomicAdd(&rsum[3 * (uint64_t)r], s1 & 0xFFFFFu);
                    if (s1 >> 20) atomicAdd(&rsum[3 * (uint64_t)r + 1], s1 >> 20);
                    if (s2) atomicAdd(&rsum[3 * (uint64_t)r + 2], s2);
                }
            }
            if (t8_go) { // a counter that wrapped (exact accounting: swar.h)
                uint32_t hot0 = 0, hot1 = 0;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) { hot0 |= old[kk]; hot1 |= old[8 + kk]; }
                if (hot0 & 0x80808080u) t8_check<0>(em, c32, cx, f[0], old[0], old[1], old[2], old[3], old[4], old[5], old[6], old[7]);
                if (hot1 & 0x80808080u) t8_check<8>(em, c32, cx, f[1], old[8], old[9], old[10], old[11], old[12], old[13], old[14], old[15]);
            }
            // ---- triplets, second half: the passes
            if (trip) {
                const uint32_t rcm = rc ? 0x33333333u : 0u;
                // quality 20..94 <=> (signed char)(q + 33) >= '5'; flags at the byte MSBs, then moved next to each other in pairs
                uint32_t qf[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint32_t x = Q[d] & 0x7F7F7F7Fu;
                    const uint32_t fq = (x + 0x6C6C6C6Cu) & ~(x + 0x21212121u) & ~Q[d];
                    const uint32_t y = fq & 0x80808080u;
                    qf[d] = y | (y << 4);
                }
                uint32_t ct[2]; // read codes as the triplet test sees them (BAM-orientation char -> Dna: non-ACGT -> A, i.e. 3 in the cycle space of a reverse read)
#pragma unroll
                for (int h = 0; h < 2; ++h) { const uint32_t x = ~P[h].oh & M; ct[h] = cn[h] | ((x | (x << 1)) & rcm); }
                // what a lane needs from its neighbours does not depend on the alignment: the read's code and the "N or outside the
                // read" flag of the cycle in front of its first and behind its last one — once per row (round 3: four lane moves per pass)
                const uint32_t ctp = lane_prev(ct[1]) & 3u, nbp = lane_prev(nb[1]) & 1u;
                const uint32_t ctn = lane_next(ct[0]) & 0x30000000u, nbn = lane_next(nb[0]) & 0x10000000u;
                uint32_t* tbin = lds + KL_TRIP + ((rc ? 2u : 0u) + mate) * 256u; // fwd1st fwd2nd rev1st rev2nd
                auto eval = [&](uint32_t kab, uint32_t pp, const uint32_t (&e)[4]) {
                    const uint32_t ja = (uint32_t)min(max((int32_t)(kab & 0xFFFFu) - (int32_t)w16, 0), 16), jb = (uint32_t)min(max((int32_t)(kab >> 16) - (int32_t)w16, 0), 16);
                    const uint32_t sh = 28u - 4u * (pp & 7u);
                    uint32_t G[3], E[2], Ep, En; // Ep: the reference nibble of the cycle in front of the lane's first (bits 3:0), En: behind its last (bits 31:28)
#pragma unroll
                    for (int h = 0; h < 3; ++h) G[h] = alignbit(e[h], e[h + 1], sh);
                    const uint32_t pv = e[0] >> sh;
                    if (rc) { E[0] = __brev(G[1]); E[1] = __brev(G[0]); Ep = __brev(G[2]); En = __brev(pv); }
                    else { E[0] = G[0]; E[1] = G[1]; Ep = pv; En = G[2]; }
                    uint32_t I[4], bad[4]; // [1 .. 2] = this lane, [0] / [3] = the cycle in front / behind (low / top nibble)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        I[h + 1] = (E[h] & 0xCCCCCCCCu) | ct[h]; // nibble = [r c]
                        const uint32_t t = I[h + 1] ^ (I[h + 1] >> 2);
                        bad[h + 1] = ((t | (t >> 1)) & M) | nb[h]; // as a flank: mismatch or N
                    }
                    I[0] = (Ep & 0xCu) | ctp;
                    { const uint32_t t = I[0] ^ (I[0] >> 2); bad[0] = ((t | (t >> 1)) & 1u) | nbp; }
                    I[3] = (En & 0xC0000000u) | ctn;
                    { const uint32_t t = I[3] ^ (I[3] >> 2); bad[3] = ((t | (t >> 1)) & 0x10000000u) | nbn; }
                    uint32_t pa[2], pb[2];
                    kl_lut_nib(pa, LUT + 8u * ja);
                    kl_lut_nib(pb, LUT + 8u * jb);
                    uint32_t ok[2], okany = 0;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t fl = alignbit(bad[h], bad[h + 1], 4) | alignbit(bad[h + 1], bad[h + 2], 28);
                        ok[h] = own ? P[h].oh & ~fl & pb[h] & ~pa[h] & (vperm(qf[2 * h], qf[2 * h + 1], 0x07050301u) >> 3) : 0u; // nibble MSB -> LSB
                        okany |= ok[h];
                    }
                    if (okany) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            if (!ok[h]) continue;
                            const uint32_t SA = alignbit(I[h], I[h + 1], 6), SB = alignbit(I[h + 1], I[h + 2], 22);
#pragma unroll
                            for (int t = 0; t < 8; ++t) { // bin = c(j-1) r(j) c(j) r(j+1): 8 contiguous bits of the [r c] stream
                                const uint32_t ix = t < 6 ? bfe(SA, 20 - 4 * t, 8) : bfe(SB, 12 - 4 * (t - 6), 8);
                                // (a cycle that does not count adds 0: a branch per cycle was two scalar and two vector instructions more
                                // than the atomic itself, sixteen times a pass)
#if KL_EXPER & 32 // the passes without their atomics
                                okany += ix + bfe(ok[h], 28 - 4 * t, 1);
#else
                                __hip_atomic_fetch_add(tbin + ix, bfe(ok[h], 28 - 4 * t, 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
                            }
                        }
#if KL_EXPER & 32
                        if (okany == 0x12345u) atomicAdd(tbin, 1u); // (keeps the arithmetic alive)
#endif
                    }
                };
                if (t_any) {
                    eval(t_kab[0], t_pp[0], t_e0);
                    if (t_two) eval(t_kab[1], t_pp[1], t_e1);
                }
                // what is left: lanes touched by three or more segments (rare), operations 64 and up (long CIGARs)
                auto passes = [&](uint64_t todo, uint32_t kab_s, uint32_t U_s, uint32_t wm) {
                    while (todo) {
                        uint32_t kab[2], U[2], have, pp0, pp1 = 0, e0[4], e1[4] = {0, 0, 0, 0};
                        todo = collect(todo, kab_s, U_s, wm, kab, U, have);
                        const bool two = __ballot(have >= 2u) != 0;
                        issue(U[0], pp0, e0);
                        if (two) issue(U[1], pp1, e1);
                        eval(kab[0], pp0, e0);
                        if (two) eval(kab[1], pp1, e1);
                    }
                };
                passes(t_todo, t_kab_s, t_U_s, t_wm);
                for (uint32_t kb = 64u; kb < ncig && rp_c < (int64_t)L; kb += 64u) {
                    uint32_t kab_s, U_s;
                    const uint64_t todo = scan_block(kb, kab_s, U_s);
                    passes(todo, kab_s, U_s, 0u);
                }
            }
            }
        next_read:
            cur_m = nxt_m;
#pragma unroll
            for (int h = 0; h < 3; ++h) s[h] = s_n[h];
#pragma unroll
            for (int d = 0; d < 4; ++d) q[d] = q_n[d];
            cw = cw_n;
            ro = ro_n;
            nxt_m = nn_m;
        }
    }
    if (threadIdx.x == 0) { t8_directory(t8_used + wg * BQC_T8_USED, t8_n, t8_tags); cyc_used[wg] = cyc_written ? 1u : 0u; } // rows of this workgroup's slot that k_t8_fold has to read; its tile
}

// the per-cycle tiles of the workgroups of one row (blockIdx.y), summed and added to the read group's counters: thread per tile word
__global__ __launch_bounds__(256) void k_long_cyc_fold(const uint32_t* __restrict__ tiles, const uint32_t* __restrict__ used, uint32_t gx, StateLayout sl,
                                                          uint64_t* __restrict__ state, uint32_t lane)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y; // i < 2 * 6 * 1024
    uint64_t v = 0;
    for (uint32_t x = 0; x < gx; ++x) {
        const uint32_t wg = y * gx + x;
        if (used[wg]) v += tiles[(size_t)wg * (2 * 6 * 1024) + i];
    }
    if (!v) return;
    const uint32_t m = i / (6 * 1024), c = (i / 1024) % 6, jj = i % 1024, w = jj % 64, t = jj / 64;
    if (w == 0 || w == 63) return; // (never written)
    const uint32_t j = y * KL_ROW + 16u * (w - 1u) + t;
    if (j < sl.lcap) gadd(state + sl.mate_base(lane, m) + (c < 5 ? sl.m_dnacount + c * sl.lcap : sl.m_qualcount) + j, v);
}

// per-read histograms from the sums (QualityCheck.hpp:157-165): thread per read of the generic chunks
__global__ __launch_bounds__(256) void k_long_finish(DevBatch b, StateLayout sl, uint64_t* __restrict__ state, const uint32_t* __restrict__ rsum)
{
    const uint32_t n_chunks = b.desc->n_chunks_slow;
    for (uint32_t ci = blockIdx.x; ci < n_chunks; ci += gridDim.x) {
        const Chunk ch = b.chunks[ci];
        for (uint32_t t0 = 0; t0 < ch.count; t0 += blockDim.x) {
            const uint32_t t = t0 + threadIdx.x;
            const bool live = t < ch.count;
            const uint32_t r = live ? b.perm[ch.first + t] : 0;
            const uint32_t flag = live ? b.flag[r] : 0x900u;
            const bool prim = live && !(flag & 0x900u) && (flag & 0xC0u);
            const uint32_t L = prim ? b.l_seq[r] : 0;
            const uint32_t qs = prim ? rsum[3 * (uint64_t)r] : 0, nN = prim ? rsum[3 * (uint64_t)r + 1] : 0, nGC = prim ? rsum[3 * (uint64_t)r + 2] : 0;
            uint64_t* M = state + sl.mate_base(ch.lane, (flag & 0x40u) ? 0u : 1u);
            wave_inc(prim && nN <= sl.lcap, M + sl.m_ncount + (nN <= sl.lcap ? nN : 0));
            wave_inc(prim && nGC <= sl.lcap, M + sl.m_gccount + (nGC <= sl.lcap ? nGC : 0));
            const bool hq = prim && L > 0; // round-half-away and ceil of qs/L in exact integer arithmetic
            const uint32_t rnd = hq ? (uint32_t)((2ull * qs + L) / (2ull * L)) : 0, cl = hq ? (uint32_t)(((uint64_t)qs + L - 1) / L) : 0;
            wave_inc(hq, M + sl.m_avgqual + (rnd & 255u));
            wave_inc(hq, M + sl.m_avgceil + (cl & 255u));
        }
    }
}

extern "C" hipError_t bqc_long_init()
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_long), hipFuncAttributeMaxDynamicSharedMemorySize, KL_WORDS * 4);
}

static void kl_grid(uint32_t max_len_ub, uint32_t n_chunks_ub, uint32_t n_cu, uint32_t& gx, uint32_t& rows)
{
    rows = max_len_ub ? (max_len_ub + KL_ROW - 1) / KL_ROW : 1u;
    gx = n_cu / rows ? n_cu / rows : 1u; // one workgroup per CU (119 KB of LDS each): gx * rows <= n_cu where possible
    if (gx > n_chunks_ub) gx = n_chunks_ub;
}

// workgroups (= slots of the 8-mer scratch rows) a launch with these bounds has
extern "C" uint32_t bqc_long_slots(uint32_t max_len_ub, uint32_t n_chunks_ub, uint32_t n_cu)
{
    if (n_chunks_ub == 0) return 0;
    uint32_t gx, rows;
    kl_grid(max_len_ub, n_chunks_ub, n_cu, gx, rows);
    return gx * rows;
}

// n_chunks_ub / max_len_ub: host-side upper bounds (the exact values are in the batch descriptor on the device); t8rows / t8_used:
// bqc_long_slots() slots of the scratch table
extern "C" void bqc_launch_long(const DevBatch& b, const StateLayout& sl, uint64_t* state, const DevRefs& refs, uint32_t* err,
                                uint32_t* rsum, uint32_t max_len_ub, uint32_t n_chunks_ub, uint32_t n_cu, uint32_t* t8rows, uint32_t* t8_used,
                                uint32_t t8_lane, uint32_t* cyc_tiles /* [slots][2 * 6 * 1024] */, uint32_t* cyc_used, hipStream_t s)
{
    if (n_chunks_ub == 0) return;
    static const hipError_t attr_once = bqc_long_init(); // (at the first launch: see bqc_launch_short)
    (void)attr_once;
    uint32_t gx, rows;
    kl_grid(max_len_ub, n_chunks_ub, n_cu, gx, rows);
    hipLaunchKernelGGL(k_long, dim3(gx, rows), dim3(KL_WAVES * 64), KL_WORDS * 4, s, b, sl, state, refs, err, rsum, (uint4*)t8rows, t8_used, t8_lane, (uint4*)cyc_tiles, cyc_used);
    hipLaunchKernelGGL(k_long_cyc_fold, dim3(2 * 6 * 1024 / 256, rows), dim3(256), 0, s, cyc_tiles, cyc_used, gx, sl, state, t8_lane);
    const uint32_t g2 = n_chunks_ub < n_cu * 8 ? n_chunks_ub : n_cu * 8;
    hipLaunchKernelGGL(k_long_finish, dim3(g2), dim3(256), 0, s, b, sl, state, rsum);
}
