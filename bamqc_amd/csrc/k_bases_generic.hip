// k_bases_generic.hip — generic per-base kernel: wave per read, lane per base (any read length).
// Fallback for reads that do not fit the short-read fast path (k_short.hip).
#include "kernels_common.h"

// ---------------------------------------------------------------------------------------------------
// k_bases
// ---------------------------------------------------------------------------------------------------
// LDS map (uint32 words)
#define L_T8    0                                  // 32768: 65536 u16 8-mer counters packed two per dword
#define L_CYC   (L_T8 + 32768)                     // [2 mates][6: A C G T N qualsum][BQC_CT]
#define L_NC    (L_CYC + 2 * 6 * BQC_CT)           // [2][BQC_CT + 1] N-count histogram
#define L_GC    (L_NC + 2 * (BQC_CT + 1))          // [2][BQC_CT + 1]
#define L_AQ    (L_GC + 2 * (BQC_CT + 1))          // [2][256] round(mean quality)
#define L_AC    (L_AQ + 512)                       // [2][256] ceil(mean quality) presence
#define L_TRIP  (L_AC + 512)                       // [1024]
#define L_MISC  (L_TRIP + 1024)                    // [8]
#define L_WORDS (L_MISC + 8)
extern "C" __host__ uint32_t bqc_k_bases_lds_bytes() { return L_WORDS * 4; }

__device__ void bases_flush(uint32_t* lds, const StateLayout& sl, uint64_t* state, uint32_t lane, bool t8, bool rest)
{
    const uint64_t lb = sl.lane_base(lane);
    if (t8) {
        for (uint32_t i = threadIdx.x; i < 32768; i += blockDim.x) {
            const uint32_t v = lds[L_T8 + i];
            if (v & 0xFFFFu) gadd(state + lb + sl.o_eightmer + 2 * i, v & 0xFFFFu);
            if (v >> 16) gadd(state + lb + sl.o_eightmer + 2 * i + 1, v >> 16);
            lds[L_T8 + i] = 0;
        }
    }
    if (!rest) return;
    for (uint32_t i = threadIdx.x; i < 2 * 6 * BQC_CT; i += blockDim.x) {
        const uint32_t v = lds[L_CYC + i];
        if (v) {
            const uint32_t m = i / (6 * BQC_CT), c = (i / BQC_CT) % 6, j = i % BQC_CT;
            if (j < sl.lcap) {
                const uint64_t mb = sl.mate_base(lane, m);
                gadd(state + mb + (c < 5 ? sl.m_dnacount + c * sl.lcap : sl.m_qualcount) + j, v);
            }
            lds[L_CYC + i] = 0;
        }
    }
    for (uint32_t i = threadIdx.x; i < 2 * (BQC_CT + 1); i += blockDim.x) {
        const uint32_t m = i / (BQC_CT + 1), j = i % (BQC_CT + 1);
        const uint64_t mb = sl.mate_base(lane, m);
        uint32_t v = lds[L_NC + i];
        if (v && j <= sl.lcap) gadd(state + mb + sl.m_ncount + j, v);
        lds[L_NC + i] = 0;
        v = lds[L_GC + i];
        if (v && j <= sl.lcap) gadd(state + mb + sl.m_gccount + j, v);
        lds[L_GC + i] = 0;
    }
    for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x) {
        const uint64_t mb = sl.mate_base(lane, i >> 8);
        uint32_t v = lds[L_AQ + i];
        if (v) gadd(state + mb + sl.m_avgqual + (i & 255), v);
        lds[L_AQ + i] = 0;
        v = lds[L_AC + i];
        if (v) gadd(state + mb + sl.m_avgceil + (i & 255), v);
        lds[L_AC + i] = 0;
    }
    for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) {
        const uint32_t v = lds[L_TRIP + i];
        if (v) gadd(state + lb + sl.o_triplet + i, v);
        lds[L_TRIP + i] = 0;
    }
}

template <bool DO_CYC, bool DO_8MER, bool DO_TRIP>
__global__ __launch_bounds__(1024) void k_bases(DevBatch b, StateLayout sl, uint64_t* __restrict__ state, DevRefs refs,
                                                    uint32_t* __restrict__ err)
{
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < L_WORDS; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const int ln = lane_id();
    const uint32_t wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    uint32_t cur_lane = 0xFFFFFFFFu;
    uint32_t t8max = 0; // per-thread running max of the u16 8-mer counters this thread touched

    for (uint32_t ci = blockIdx.x; ci < b.n_chunks; ci += gridDim.x) {
        const Chunk ch = b.chunks[ci];
        if (ch.lane != cur_lane) { // block-uniform
            if (cur_lane != 0xFFFFFFFFu) {
                __syncthreads();
                bases_flush(lds, sl, state, cur_lane, DO_8MER, true);
                t8max = 0;
                __syncthreads();
            }
            cur_lane = ch.lane;
        }
        const uint64_t lb = sl.lane_base(cur_lane);
        for (uint32_t k = wave; k < ch.count; k += nwaves) {
            const uint32_t r = b.perm ? b.perm[ch.first + k] : ch.first + k;
            const uint32_t flag = b.flag[r];
            if (flag & 0x900) continue;                      // supplementary / secondary: bamqualcheck.cpp:318-327
            const uint32_t mate = (flag & 0x40) ? 0u : 1u;
            if (!(flag & 0xC0)) continue;                    // error raised by k_reads / host
            const uint32_t L = b.l_seq[r];
            const bool rc = flag & 0x10, noqual = flag & BQC_FLAG_NO_QUAL;
            const uint8_t* __restrict__ sq = b.seq + b.seq_off[r];
            const uint8_t* __restrict__ ql = b.qual + b.qual_off[r];
            const uint64_t mb = sl.mate_base(cur_lane, mate);
            const uint64_t lut_seq = rc ? LUT5_RC : LUT5_FWD; // seq-orient code (after reverseComplement)
            // triplets (BAM orientation)
            const uint32_t ncig = b.n_cigar[r];
            const uint32_t* __restrict__ cg = b.cigar + b.cigar_off[r];
            const int32_t rid = b.rid[r];
            const bool trip = DO_TRIP && (flag & BQC_FLAG_TRIPLET) && L >= 3 && ncig > 0 && !noqual && rid >= 0 &&
                              (uint32_t)rid < refs.n_refs && refs.ref[rid] != nullptr;
            const uint8_t* __restrict__ ref = trip ? refs.ref[rid] : nullptr;
            const int64_t reflen = trip ? (int64_t)refs.len[rid] : 0;
            const int64_t pos = b.pos[r];
            const uint32_t grp = (rc ? 2u : 0u) + mate; // fwd1st, fwd2nd, rev1st, rev2nd (TripletCounting.hpp:174-189)
            uint32_t nN = 0, nGC = 0, qs = 0;
            uint32_t prev_nib = 0;
            bool bad_q = false;

            for (uint32_t t0 = 0; t0 < L; t0 += BQC_TILE_STRIDE) {
                const uint32_t i = t0 + ln;
                const bool in = i < L;
                uint32_t nib = 0, q = 0;
                if (in) {
                    const uint32_t by = sq[i >> 1];
                    nib = (i & 1u) ? (by & 15u) : (by >> 4);
                    if (!noqual) q = ql[i];
                }
                const bool own = in && ln < BQC_TILE_STRIDE;
                const bool isN = nib == 15u;
                if (DO_CYC) { // read_counts, QualityCheck.hpp:122-166 (sequencing orientation)
                    bad_q |= q > 222u;
                    if (own) {
                        const uint32_t c5 = lut5(lut_seq, nib);
                        const uint32_t cyc = rc ? (L - 1 - i) : i;
                        if (cyc < BQC_CT) {
                            atomicAdd(&lds[L_CYC + (mate * 6 + c5) * BQC_CT + cyc], 1u);
                            if (!noqual) atomicAdd(&lds[L_CYC + (mate * 6 + 5) * BQC_CT + cyc], q);
                        } else if (cyc < sl.lcap) {
                            gadd(state + mb + sl.m_dnacount + c5 * sl.lcap + cyc, 1);
                            if (!noqual) gadd(state + mb + sl.m_qualcount + cyc, q);
                        }
                        qs += q;
                    }
                    nN += (uint32_t)__popcll((unsigned long long)__ballot(own && isN));                 // literal 'N'
                    nGC += (uint32_t)__popcll((unsigned long long)__ballot(own && (nib == 2u || nib == 4u))); // 'C' / 'G'
                }
                if (DO_8MER) { // count8mers, OverallNumbers.hpp:137-168; window starts at i (BAM orientation)
                    // char -> Dna AFTER the reverse complement: complemented code, non-ACGT -> A either way
                    const uint32_t c2 = lut5(lut_seq, nib) & 3u;
                    const uint32_t v = in ? (c2 | (isN ? 0x10000u : 0u)) : 0x10000u; // past the end blocks the window
                    const uint32_t p2 = (v << 2) | (uint32_t)__shfl_down((int)v, 1);
                    const uint32_t p4 = (p2 << 4) | (uint32_t)__shfl_down((int)p2, 2);
                    const uint32_t p8 = (p4 << 8) | (uint32_t)__shfl_down((int)p4, 4);
                    if (own && (p8 >> 16) == 0) {
                        uint32_t h = p8 & 0xFFFFu;
                        if (rc) h = reverse8x2(h); // bases are already complemented: the 8-mer as read off the RC'd sequence
                        if (!ch.huge) {
                            const uint32_t old = atomicAdd(&lds[L_T8 + (h >> 1)], (h & 1u) ? 0x10000u : 1u);
                            t8max = max(t8max, max(old >> 16, old & 0xFFFFu));
                        } else {
                            gadd(state + lb + sl.o_eightmer + h, 1);
                        }
                    }
                }
                if (DO_TRIP && trip) { // countBasesInTriplets, TripletCounting.hpp:195-236 (BAM orientation)
                    uint32_t nib_next = (uint32_t)__shfl_down((int)nib, 1);
                    uint32_t nib_prev = (uint32_t)__shfl_up((int)nib, 1);
                    if (ln == 0) nib_prev = prev_nib;
                    bool cand = own && i >= 1 && i + 1 < L;
                    cand = cand && q >= 20u && q <= 94u; // (signed char)(q+33) >= '5'
                    const uint32_t base = lut5(LUT5_FWD, nib);
                    cand = cand && base != 4u && nib_prev != 15u && nib_next != 15u;
                    // CIGAR walk -> chromPos for this lane's read position (first op assumed match-like, :203)
                    int64_t cp = -1;
                    bool found = false;
                    if (__ballot(cand)) {
                        const uint32_t n0 = cg[0] >> 4;
                        if (n0 == 0 || i < n0) { cp = pos + (int64_t)i; found = true; }
                        if (n0 != 0) {
                            uint64_t rp = n0;
                            int64_t c = pos + (int64_t)n0;
                            for (uint32_t kk = 1; kk < ncig; ++kk) {
                                const uint32_t w = cg[kk], op = w & 15u, n = w >> 4;
                                if (op == 2u || op == 3u || op == 5u || op == 6u) c += n;      // D N H P
                                else if (op == 4u || op == 1u) rp += n;                           // S I
                                else {                                                            // M = X (and unknown)
                                    if ((uint64_t)i >= rp && (uint64_t)i < rp + n) { cp = c + (int64_t)((uint64_t)i - rp); found = true; }
                                    rp += n; c += n;
                                }
                                if (rp > (uint64_t)t0 + 63u) break; // later segments lie beyond this tile
                            }
                        }
                    }
                    cand = cand && found && cp >= 1 && cp + 1 < reflen;
                    if (cand) {
                        const uint32_t r0 = ref[cp - 1] & 3u, r1 = ref[cp] & 3u, r2 = ref[cp + 1] & 3u; // Dna5 -> Dna: N -> A
                        if ((lut5(LUT5_FWD, nib_prev) & 3u) == r0 && (lut5(LUT5_FWD, nib_next) & 3u) == r2)
                            atomicAdd(&lds[L_TRIP + ((r0 << 4) | (r1 << 2) | r2) * 16 + grp * 4 + base], 1u);
                    }
                    prev_nib = __builtin_amdgcn_readlane(nib, BQC_TILE_STRIDE - 1);
                }
            }
            if (DO_CYC) { // per-read histograms, QualityCheck.hpp:157-165
                qs = wave_sum(qs);
                if (__ballot(bad_q)) { if (ln == 0) atomicOr(err, BQC_DEVERR_QUAL); }
                if (ln == 0) {
                    if (nN <= BQC_CT) atomicAdd(&lds[L_NC + mate * (BQC_CT + 1) + nN], 1u);
                    else if (nN <= sl.lcap) gadd(state + mb + sl.m_ncount + nN, 1);
                    if (nGC <= BQC_CT) atomicAdd(&lds[L_GC + mate * (BQC_CT + 1) + nGC], 1u);
                    else if (nGC <= sl.lcap) gadd(state + mb + sl.m_gccount + nGC, 1);
                    if (L > 0) { // round-half-away and ceil of qs/L in exact integer arithmetic
                        const uint32_t rnd = (uint32_t)((2ull * qs + L) / (2ull * L));
                        const uint32_t cl = (uint32_t)(((uint64_t)qs + L - 1) / L);
                        atomicAdd(&lds[L_AQ + mate * 256 + (rnd & 255u)], 1u);
                        atomicAdd(&lds[L_AC + mate * 256 + (cl & 255u)], 1u);
                    }
                }
            }
        }
        if (DO_8MER) { // keep every packed u16 counter below 65535 - (largest chunk): flush the table when needed
            atomicMax(&lds[L_MISC], t8max);
            __syncthreads();
            const uint32_t m = lds[L_MISC];
            __syncthreads();
            if (m + 1u + BQC_CHUNK_BASES >= 65535u) {
                bases_flush(lds, sl, state, cur_lane, true, false);
                if (threadIdx.x == 0) lds[L_MISC] = 0;
                t8max = 0;
                __syncthreads();
            }
        }
    }
    __syncthreads();
    if (cur_lane != 0xFFFFFFFFu) bases_flush(lds, sl, state, cur_lane, DO_8MER, true);
}

extern "C" void bqc_launch_bases(const DevBatch& b, const StateLayout& sl, uint64_t* state, const DevRefs& refs, uint32_t* err,
                                 uint32_t grid, int variant, hipStream_t s)
{
    if (b.n_chunks == 0) return;
    const uint32_t lds = L_WORDS * 4;
    if (grid > b.n_chunks) grid = b.n_chunks;
    switch (variant) {
    case 0: hipLaunchKernelGGL((k_bases<true, true, true>), dim3(grid), dim3(1024), lds, s, b, sl, state, refs, err); break;
    case 1: hipLaunchKernelGGL((k_bases<true, false, false>), dim3(grid), dim3(1024), lds, s, b, sl, state, refs, err); break;
    case 2: hipLaunchKernelGGL((k_bases<false, true, false>), dim3(grid), dim3(1024), lds, s, b, sl, state, refs, err); break;
    case 3: hipLaunchKernelGGL((k_bases<false, false, true>), dim3(grid), dim3(1024), lds, s, b, sl, state, refs, err); break;
    }
}

extern "C" hipError_t bqc_kernels_init()
{
    // k_bases needs > 64 KiB of dynamic LDS
    hipError_t e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bases<true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, L_WORDS * 4);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bases<true, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, L_WORDS * 4);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bases<false, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, L_WORDS * 4);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bases<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, L_WORDS * 4);
    return e;
}

