// k_long.hip — per-base statistics for reads that do not fit the short-read fast path (longer than 256 bases;
// every read when BQC_NO_FAST=1): wave per read, lane per base, any read length.
//
// The per-cycle histograms of a 10 kb read (2 mates x 6 counters x 10 000 cycles) do not fit in LDS, so the work is
// tiled over sequencing cycles: workgroup (x, y) walks the chunks x, x + gridDim.x, ... and handles only the bases whose
// cycle lies in [y * 1024, (y + 1) * 1024) — for a reverse-strand read that is the mirrored base range.  Each base
// belongs to exactly one cycle tile, so the 8-mer windows starting there and its triplet evaluation are done by the
// same workgroup.  LDS: 65 536 packed u8 8-mer counters (exact carry accounting on wrap, as in k_short), one cycle tile
// of [2 mates][A C G T N qualsum][1024] and the 1 024 triplet counters.  Per-read sums (quality, N, GC) are combined
// across cycle tiles through a small per-read scratch array and turned into histograms by k_long_finish.
//
// Reference: QualityCheck.hpp:122-166 (read_counts), OverallNumbers.hpp:137-168 (count8mers),
// TripletCounting.hpp:195-236 (countBasesInTriplets).
#include "kernels_common.h"

#define KL_CT    1024
#define KL_T8    0
#define KL_CYC   16384
#define KL_TRIP  (KL_CYC + 2 * 6 * KL_CT)
#define KL_WORDS (KL_TRIP + 1024)

__device__ __noinline__ void kl_t8_wrap(uint64_t* __restrict__ em, uint32_t h, uint32_t old)
{
    uint32_t f = h & 3u, bin = h; // every wrap of field f: +256 for its bin, and -1 for the next bin (the carry spilled into it)
    while (f < 4u && ((old >> (8u * f)) & 0xFFu) == 0xFFu) {
        gadd(em + bin, 256);
        if (f < 3u) gadd(em + bin + 1, (uint64_t)-1ll);
        ++f; ++bin;
    }
}

__device__ __forceinline__ void kl_flush(uint32_t* lds, const StateLayout& sl, uint64_t* __restrict__ state, uint32_t lane, uint32_t cyc0)
{
    const uint64_t lb = sl.lane_base(lane);
    for (uint32_t i = threadIdx.x; i < 65536; i += blockDim.x) {
        const uint32_t v = (lds[KL_T8 + (i >> 2)] >> (8u * (i & 3u))) & 0xFFu;
        if (v) gadd(state + lb + sl.o_eightmer + i, v);
    }
    block_sync();
    for (uint32_t i = threadIdx.x; i < 16384; i += blockDim.x) lds[KL_T8 + i] = 0;
    for (uint32_t i = threadIdx.x; i < 2 * 6 * KL_CT; i += blockDim.x) {
        const uint32_t v = lds[KL_CYC + i];
        if (!v) continue;
        lds[KL_CYC + i] = 0;
        const uint32_t m = i / (6 * KL_CT), c = (i / KL_CT) % 6, j = cyc0 + i % KL_CT;
        if (j < sl.lcap) gadd(state + sl.mate_base(lane, m) + (c < 5 ? sl.m_dnacount + c * sl.lcap : sl.m_qualcount) + j, v);
    }
    for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) {
        const uint32_t v = lds[KL_TRIP + i];
        if (v) { gadd(state + lb + sl.o_triplet + i, v); lds[KL_TRIP + i] = 0; }
    }
}

__global__ __launch_bounds__(1024) void k_long(DevBatch b, StateLayout sl, uint64_t* __restrict__ state, DevRefs refs,
                                                   uint32_t* __restrict__ err, uint32_t* __restrict__ rsum /* [n_reads][3] */)
{
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < KL_WORDS; i += blockDim.x) lds[i] = 0;
    block_sync();
    const uint32_t ln = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const uint32_t cyc0 = blockIdx.y * KL_CT;
    uint32_t cur_lane = 0xFFFFFFFFu;
    const uint32_t n_chunks = b.desc->n_chunks_slow;
    if (cyc0 >= b.desc->long_max_len) return; // no read of the batch reaches this cycle tile (the grid is sized from an upper bound)
    for (uint32_t ci = blockIdx.x;; ci += gridDim.x) { // one extra pass at the end flushes the last lane
        const bool done = ci >= n_chunks;
        Chunk ch{0, 0, 0xFFFFFFFFu, 0, 0, 0, 0, 0};
        if (!done) ch = b.chunks[ci];
        if (ch.lane != cur_lane) { // block-uniform
            if (cur_lane != 0xFFFFFFFFu) {
                block_sync();
                kl_flush(lds, sl, state, cur_lane, cyc0);
                block_sync();
            }
            cur_lane = ch.lane;
        }
        if (done) break;
        uint64_t* em = state + sl.lane_base(cur_lane) + sl.o_eightmer;
        for (uint32_t k = wave; k < ch.count; k += nwaves) {
            const uint32_t r = b.perm[ch.first + k];
            const uint32_t flag = b.flag[r];
            if ((flag & 0x900u) || !(flag & 0xC0u)) continue;  // skipped records; missing mate flag is raised by k_reads / host
            const uint32_t L = b.l_seq[r];
            if (L <= cyc0) continue;                              // no base of this read falls into this cycle tile
            const uint32_t mate = (flag & 0x40u) ? 0u : 1u;
            const bool rc = flag & 0x10u, noqual = flag & BQC_FLAG_NO_QUAL;
            const uint8_t* __restrict__ sq = b.seq + b.seq_off[r];
            const uint8_t* __restrict__ ql = b.qual + b.qual_off[r];
            const uint64_t lut_seq = rc ? LUT5_RC : LUT5_FWD;   // seq-orient code (after reverseComplement)
            const uint32_t cyc1 = min(L, cyc0 + KL_CT);           // cycles [cyc0, cyc1)
            const uint32_t lo_i = rc ? L - cyc1 : cyc0, hi_i = rc ? L - cyc0 : cyc1; // the bases with those cycles
            // triplets (BAM orientation)
            const uint32_t ncig = b.n_cigar[r];
            const uint32_t* __restrict__ cg = b.cigar + b.cigar_off[r];
            const int32_t rid = b.rid[r];
            const bool trip = (flag & BQC_FLAG_TRIPLET) && L >= 3 && ncig > 0 && !noqual && rid >= 0 && (uint32_t)rid < refs.n_refs &&
                              refs.ref[rid] != nullptr;
            const uint8_t* __restrict__ ref = trip ? refs.ref[rid] : nullptr;
            const int64_t reflen = trip ? (int64_t)refs.len[rid] : 0;
            const int64_t pos = b.pos[r];
            const uint32_t grp = (rc ? 2u : 0u) + mate; // fwd1st, fwd2nd, rev1st, rev2nd (TripletCounting.hpp:174-189)
            const uint32_t n0 = trip ? cg[0] >> 4 : 0u;
            // CIGAR cursor of the triplet walk (first op assumed match-like, :203), resumed from tile to tile
            uint32_t w_kk = 1;
            uint64_t w_rp = n0;
            int64_t w_c = pos + (int64_t)n0;
            uint32_t nN = 0, nGC = 0, qs = 0;
            bool bad_q = false;

            uint32_t by_nx = 0, q_nx = 0; // loads of the next 64 positions are issued one iteration ahead
            if (lo_i + ln < L) { by_nx = sq[(lo_i + ln) >> 1]; if (!noqual) q_nx = ql[lo_i + ln]; }
            for (uint32_t t0 = lo_i; t0 < hi_i; t0 += BQC_TILE_STRIDE) {
                const uint32_t i = t0 + ln;
                const bool in = i < L;
                const uint32_t nib = in ? ((i & 1u) ? (by_nx & 15u) : (by_nx >> 4)) : 0u, q = in ? q_nx : 0u;
                {
                    const uint32_t j = i + BQC_TILE_STRIDE;
                    if (t0 + BQC_TILE_STRIDE < hi_i && j < L) { by_nx = sq[j >> 1]; if (!noqual) q_nx = ql[j]; }
                }
                const bool own = in && ln < BQC_TILE_STRIDE && i < hi_i;
                const bool isN = nib == 15u;
                { // read_counts (sequencing orientation)
                    bad_q |= own && q > 222u;
                    if (own) {
                        const uint32_t c5 = lut5(lut_seq, nib);
                        const uint32_t cyc = (rc ? (L - 1 - i) : i) - cyc0; // in [0, KL_CT)
                        atomicAdd(&lds[KL_CYC + (mate * 6 + c5) * KL_CT + cyc], 1u);
                        if (!noqual) atomicAdd(&lds[KL_CYC + (mate * 6 + 5) * KL_CT + cyc], q);
                        qs += q;
                    }
                    nN += (uint32_t)__popcll((unsigned long long)__ballot(own && isN));                 // literal 'N'
                    nGC += (uint32_t)__popcll((unsigned long long)__ballot(own && (nib == 2u || nib == 4u))); // 'C' / 'G'
                }
                { // count8mers; window starts at i (BAM orientation), bases already complemented for reverse reads
                    const uint32_t c2 = lut5(lut_seq, nib) & 3u;
                    const uint32_t v = in ? (c2 | (isN ? 0x10000u : 0u)) : 0x10000u; // past the end blocks the window
                    const uint32_t p2 = (v << 2) | (uint32_t)__shfl_down((int)v, 1);
                    const uint32_t p4 = (p2 << 4) | (uint32_t)__shfl_down((int)p2, 2);
                    const uint32_t p8 = (p4 << 8) | (uint32_t)__shfl_down((int)p4, 4);
                    if (own && (p8 >> 16) == 0) {
                        uint32_t h = p8 & 0xFFFFu;
                        if (rc) h = reverse8x2(h);
                        const uint32_t old = atomicAdd(&lds[KL_T8 + (h >> 2)], 1u << (8u * (h & 3u)));
                        if (((old >> (8u * (h & 3u))) & 0xFFu) == 0xFFu) kl_t8_wrap(em, h, old);
                    }
                }
                if (trip) { // countBasesInTriplets (BAM orientation)
                    const uint32_t nib_next = (uint32_t)__shfl_down((int)nib, 1);
                    uint32_t nib_prev = (uint32_t)__shfl_up((int)nib, 1);
                    if (ln == 0 && i > 0) { const uint32_t pb = sq[(i - 1) >> 1]; nib_prev = ((i - 1) & 1u) ? (pb & 15u) : (pb >> 4); }
                    bool cand = own && i >= 1 && i + 1 < L;
                    cand = cand && q >= 20u && q <= 94u; // (signed char)(q+33) >= '5'
                    const uint32_t base = lut5(LUT5_FWD, nib);
                    cand = cand && base != 4u && nib_prev != 15u && nib_next != 15u;
                    int64_t cp = -1;
                    bool found = false;
                    if (n0 == 0 || i < n0) { cp = pos + (int64_t)i; found = true; }
                    if (n0 != 0) { // advance the shared cursor past everything that ends before this tile, then scan the tile
                        while (w_kk < ncig) {
                            const uint32_t wv = cg[w_kk], op = wv & 15u, n = wv >> 4;
                            const bool m = !(op == 2u || op == 3u || op == 5u || op == 6u || op == 4u || op == 1u);
                            if (m && w_rp + n > (uint64_t)t0) break; // this match segment reaches into the tile
                            if (op == 2u || op == 3u || op == 5u || op == 6u) w_c += n;    // D N H P
                            else if (op == 4u || op == 1u) w_rp += n;                       // S I
                            else { w_rp += n; w_c += n; }                                   // M = X entirely before the tile
                            ++w_kk;
                        }
                        uint64_t rp = w_rp;
                        int64_t c = w_c;
                        for (uint32_t kk = w_kk; kk < ncig && rp <= (uint64_t)t0 + 63u; ++kk) {
                            const uint32_t wv = cg[kk], op = wv & 15u, n = wv >> 4;
                            if (op == 2u || op == 3u || op == 5u || op == 6u) c += n;
                            else if (op == 4u || op == 1u) rp += n;
                            else {
                                if ((uint64_t)i >= rp && (uint64_t)i < rp + n) { cp = c + (int64_t)((uint64_t)i - rp); found = true; }
                                rp += n; c += n;
                            }
                        }
                    }
                    cand = cand && found && cp >= 1 && cp + 1 < reflen;
                    if (cand) {
                        const uint32_t r0 = ref[cp - 1] & 3u, r1 = ref[cp] & 3u, r2 = ref[cp + 1] & 3u; // Dna5 -> Dna: N -> A
                        if ((lut5(LUT5_FWD, nib_prev) & 3u) == r0 && (lut5(LUT5_FWD, nib_next) & 3u) == r2)
                            atomicAdd(&lds[KL_TRIP + ((r0 << 4) | (r1 << 2) | r2) * 16 + grp * 4 + base], 1u);
                    }
                }
            }
            // per-read sums of this cycle tile -> scratch (combined over tiles by k_long_finish)
            qs = wave_sum(qs);
            if (__ballot(bad_q)) { if (ln == 0) atomicOr(err, BQC_DEVERR_QUAL); }
            if (ln == 0) {
                if (qs) atomicAdd(&rsum[3 * (uint64_t)r], qs);
                if (nN) atomicAdd(&rsum[3 * (uint64_t)r + 1], nN);
                if (nGC) atomicAdd(&rsum[3 * (uint64_t)r + 2], nGC);
            }
        }
    }
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

// n_chunks_ub / max_len_ub: host-side upper bounds (the exact values are in the batch descriptor on the device)
extern "C" void bqc_launch_long(const DevBatch& b, const StateLayout& sl, uint64_t* state, const DevRefs& refs, uint32_t* err,
                                uint32_t* rsum, uint32_t max_len_ub, uint32_t n_chunks_ub, uint32_t n_cu, hipStream_t s)
{
    if (n_chunks_ub == 0) return;
    const uint32_t tiles = max_len_ub ? (max_len_ub + KL_CT - 1) / KL_CT : 1u;
    uint32_t gx = n_cu / tiles ? n_cu / tiles : 1u; // one workgroup per CU (116 KB of LDS each): gx * tiles <= n_cu where possible
    if (gx > n_chunks_ub) gx = n_chunks_ub;
    hipLaunchKernelGGL(k_long, dim3(gx, tiles), dim3(1024), KL_WORDS * 4, s, b, sl, state, refs, err, rsum);
    const uint32_t g2 = n_chunks_ub < n_cu * 8 ? n_chunks_ub : n_cu * 8;
    hipLaunchKernelGGL(k_long_finish, dim3(g2), dim3(256), 0, s, b, sl, state, rsum);
}
