// kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for BamQC's per-read aggregation.
//
//   k_reads  : thread per read   — flag cascade + scalar counters (bamqualcheck.cpp:318-434), per-read
//              histograms of QualityCheck (QualityCheck.hpp:168-271)
//   k_bases  : wave per read, lane per base — per-cycle base/quality histograms (QualityCheck.hpp:122-166),
//              8-mer spectrum (OverallNumbers.hpp:137-168), reference-context triplets
//              (TripletCounting.hpp:195-236); LDS-privatised, one flush per workgroup
//   k_cov    : workgroup per 4 coverage windows — depth by difference array + scan + clamp-100 histogram
//              (OverallNumbers.hpp:59-135)
//
// All accumulators are integers; results are bit-exact sums, independent of scheduling.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_types.h"
#include "../../include/bamqc.h"

#define WAVE 64

// ---------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

__device__ __forceinline__ void gadd(uint64_t* p, uint64_t v)
{
    atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v);
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

// ---------------------------------------------------------------------------------------------------
// k_reads
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reads(DevBatch b, StateLayout sl, uint64_t* __restrict__ state, DevRefs refs,
                                                  uint32_t* __restrict__ err)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < b.n_reads;
    uint32_t flag = 0, L = 0, mapq = 0, ncig = 0, lane = 0;
    int32_t rid = -1, tlen = 0, nm = BQC_NM_ABSENT;
    if (live) {
        flag = b.flag[r]; L = b.l_seq[r]; mapq = b.mapq[r]; ncig = b.n_cigar[r]; lane = b.lane[r];
        rid = b.rid[r]; tlen = b.tlen[r]; nm = b.nm[r];
    }
    uint64_t* S = state + sl.lane_base(lane) + sl.o_scalars;
    // ---- flag cascade, bamqualcheck.cpp:318-335
    const bool supp = live && (flag & 0x800);
    const bool sec = live && !supp && (flag & 0x100);
    const bool prim = live && !(flag & 0x900);
    const bool dup = prim && (flag & 0x400), qcf = prim && (flag & 0x200);
    wave_inc(supp, S + BQC_S_SUPPLEMENTARY);
    wave_inc(sec, S + BQC_S_NOT_PRIMARY);
    wave_inc(dup, S + BQC_S_DUPLICATES);
    wave_inc(qcf, S + BQC_S_QCFAILED);
    // ---- :353-389
    const bool first = prim && (flag & 0x40);
    const bool last = prim && !first && (flag & 0x80);
    if (prim && !first && !last) atomicOr(err, BQC_DEVERR_MATE);
    const bool unm = flag & 0x4, nunm = flag & 0x8, proper = flag & 0x2;
    const bool rc = flag & 0x10, nrc = flag & 0x20;
    wave_inc(prim, S + BQC_S_READCOUNT);
    { // totalbps += L : wave-sum when every live lane targets the same lane block, else per-thread
        const uint64_t m = __ballot(prim);
        if (m) {
            const int leader = __ffsll((unsigned long long)m) - 1;
            const uint32_t l0 = __builtin_amdgcn_readlane(lane, leader);
            const bool uniform = __ballot(prim && lane != l0) == 0;
            if (uniform) {
                uint32_t lo = wave_sum(prim ? (L & 0xFFFFu) : 0u), hi = wave_sum(prim ? (L >> 16) : 0u);
                if (lane_id() == leader) gadd(S + BQC_S_TOTALBPS, (uint64_t)lo + ((uint64_t)hi << 16));
            } else if (prim) {
                gadd(S + BQC_S_TOTALBPS, L);
            }
        }
    }
    wave_inc(first && unm, S + BQC_S_FIRSTUNMAPPED);
    wave_inc(first && unm && nunm, S + BQC_S_BOTHUNMAPPED);
    wave_inc(first && proper, S + BQC_S_PROPERPAIR);
    wave_inc(first && proper && (rc == nrc), S + BQC_S_FF_RR);
    wave_inc(last && unm, S + BQC_S_SECONDUNMAPPED);
    const bool mated = first || last;
    const uint32_t mate = first ? 0u : 1u;
    uint64_t* M = state + sl.mate_base(lane, mate);
    // read_length + qualcount_readnr (QualityCheck.hpp:130,168-176)
    wave_inc(mated, M + sl.m_readnr);
    wave_inc(mated && L <= sl.lcap, M + sl.m_readlen + (L <= sl.lcap ? L : 0));
    // ---- main chromosomes only, :392-434
    const bool in_main = mated && rid >= 0 && (uint32_t)rid < refs.n_refs && refs.main_chrom[rid];
    const bool mapped_main = in_main && !unm;
    uint32_t del = 0, ins = 0;
    if (mapped_main) { // cigar_count (QualityCheck.hpp:222-271) on the seq-oriented (reversed for RC) CIGAR
        const uint32_t* cg = b.cigar + b.cigar_off[r];
        if (ncig > 0) {
            const uint32_t c_first = rc ? cg[ncig - 1] : cg[0];
            const uint32_t c_last = rc ? cg[0] : cg[ncig - 1];
            if ((c_first & 15u) == 4u) { // 'S'
                uint32_t n = c_first >> 4;
                if (n > L) n = L;
                gadd(M + sl.m_sc5hist + n, 1); // sc5[j]++ for j < n  <=>  histogram of n, suffix-summed at finalize
            } else if ((c_last & 15u) == 4u) {
                const uint32_t n = c_last >> 4;
                if (n <= L && n > 0) { // for (j = L-n; j < L; ++j) sc3[j]++   as a difference array
                    gadd(M + sl.m_sc3diff + (L - n), 1);
                    gadd(M + sl.m_sc3diff + L, (uint64_t)-1ll);
                }
            }
            for (uint32_t k = 0; k < ncig; ++k) {
                const uint32_t c = cg[k], op = c & 15u;
                if (op == 2u) del += c >> 4;       // 'D'
                else if (op == 1u) ins += c >> 4;  // 'I'
            }
        }
        if (del >= sl.hcap || ins >= sl.hcap) atomicOr(err, BQC_DEVERR_RANGE);
    }
    const bool hist_ok = mapped_main && del < sl.hcap && ins < sl.hcap;
    wave_inc(hist_ok, M + sl.m_delhist + (hist_ok ? del : 0));
    wave_inc(hist_ok, M + sl.m_inshist + (hist_ok ? ins : 0));
    wave_inc(mapped_main, M + sl.m_mapq + mapq);                         // map_Q :178-185
    { // mis_match :198-220
        const bool has = mapped_main && nm != BQC_NM_ABSENT;
        const uint32_t mm = (uint32_t)nm - del - ins; // unsigned arithmetic (:210)
        if (has && mm >= sl.hcap) atomicOr(err, BQC_DEVERR_RANGE);
        const bool ok = has && mm < sl.hcap;
        wave_inc(ok, M + sl.m_mismatch + (ok ? mm : 0));
    }
    if (first && mapped_main && !nunm && (flag & BQC_FLAG_MATE_MAIN)) { // insert_size :187-196
        uint32_t idx = tlen < 0 ? (uint32_t)0 - (uint32_t)tlen : (uint32_t)tlen; // abs(INT_MIN) -> 2^31
        if (idx >= sl.icap) idx = sl.icap - 1;
        gadd(M + sl.m_insert + idx, 1);
    }
    wave_inc(first && in_main && (!unm || !nunm) && !(flag & 0x400), S + BQC_S_FIRST_AND_OR_SECOND_MAPPED);
    wave_inc(first && in_main && proper && !(flag & 0x400), S + BQC_S_AUTO_PROPERPAIR);
}

// further integer NM tags of a record: mis_match counts once per tag (QualityCheck.hpp:201-218)
__global__ void k_nm_extra(DevBatch b, StateLayout sl, uint64_t* __restrict__ state, DevRefs refs, uint32_t* __restrict__ err)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= b.n_nm_extra) return;
    const uint32_t r = b.nm_extra_read[e];
    const uint32_t flag = b.flag[r];
    if (flag & 0x900) return;
    const bool first = flag & 0x40, last = !first && (flag & 0x80);
    if (!first && !last) return;
    const int32_t rid = b.rid[r];
    if (!(rid >= 0 && (uint32_t)rid < refs.n_refs && refs.main_chrom[rid]) || (flag & 0x4)) return;
    const uint32_t* cg = b.cigar + b.cigar_off[r];
    uint32_t del = 0, ins = 0;
    for (uint32_t k = 0; k < b.n_cigar[r]; ++k) {
        const uint32_t c = cg[k], op = c & 15u;
        if (op == 2u) del += c >> 4; else if (op == 1u) ins += c >> 4;
    }
    const uint32_t mm = (uint32_t)b.nm_extra_val[e] - del - ins;
    if (mm >= sl.hcap) { atomicOr(err, BQC_DEVERR_RANGE); return; }
    gadd(state + sl.mate_base(b.lane[r], first ? 0u : 1u) + sl.m_mismatch + mm, 1);
}

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

// ---------------------------------------------------------------------------------------------------
// k_cov — coverage depth histogram
// ---------------------------------------------------------------------------------------------------
// Virtual coordinates: the host runs the order-dependent anchor recurrence (OverallNumbers.hpp:84-110)
// and numbers every 1000-position window in flush order; a read contributes to
// [win*1000 + off + c, ...) truncated at (win+2)*1000.  Depth is then order-free.
__global__ __launch_bounds__(256) void k_cov(DevBatch b, StateLayout sl, uint64_t* __restrict__ state,
                                                uint32_t* __restrict__ carry /* [lane][2][2000] */, const uint32_t* __restrict__ parity)
{
    __shared__ int32_t diff[BQC_COV_TILE + 8];
    __shared__ uint32_t hist[BQC_COVSIZE + 1];
    __shared__ uint32_t wsum[4];
    const CovTile t = b.cov_tiles[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < BQC_COV_TILE + 8; i += blockDim.x) diff[i] = 0;
    for (uint32_t i = threadIdx.x; i <= BQC_COVSIZE; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)t.win_lo * BQC_VSIZE, hi = lo + BQC_COV_TILE;
    for (uint32_t e = t.list_begin + threadIdx.x; e < t.list_end; e += blockDim.x) {
        const uint32_t r = b.cov_list[e];
        const uint32_t flag = b.flag[r];
        const bool rc = flag & 0x10;
        const int64_t base = (int64_t)b.cov_win[r] * BQC_VSIZE;
        const int64_t limit = base + 2 * BQC_VSIZE; // DEFINED: increments at window offset >= 2000 are dropped
        const int64_t p0 = base + b.cov_off[r];
        const uint32_t ncig = b.n_cigar[r];
        const uint32_t* cg = b.cigar + b.cigar_off[r];
        uint32_t c = 0; // `int c` in the reference; wraps identically
        for (uint32_t k = 0; k < ncig; ++k) { // seq-oriented CIGAR: reversed for RC reads (bamqualcheck.cpp:349)
            const uint32_t w = cg[rc ? ncig - 1 - k : k], op = w & 15u, n = w >> 4;
            if (op == 4u) c += n;                    // 'S'
            if (op == 0u || op == 2u) {              // 'M' or 'D'
                int64_t a = p0 + c, z = a + n;
                if (z > limit) z = limit;
                if (a < lo) a = lo;
                if (z > hi) z = hi;
                if (a < z) {
                    atomicAdd(&diff[a - lo], 1);
                    atomicAdd(&diff[z - lo], -1);
                }
                c += n;
            }
        }
    }
    __syncthreads();
    // block scan of diff: 16 consecutive entries per thread (4000 <= 256 * 16)
    const uint32_t per = (BQC_COV_TILE + 255) / 256;
    const uint32_t s0 = threadIdx.x * per;
    int32_t loc = 0;
    for (uint32_t j = 0; j < per; ++j) if (s0 + j < BQC_COV_TILE) loc += diff[s0 + j];
    // inclusive wave scan of thread totals
    int32_t inc = loc;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const int32_t v = __shfl_up(inc, o);
        if (lane_id() >= o) inc += v;
    }
    if (lane_id() == WAVE - 1) wsum[threadIdx.x >> 6] = (uint32_t)inc;
    __syncthreads();
    int32_t off = inc - loc;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) off += (int32_t)wsum[w];
    const uint32_t par = parity[t.lane] & 1u; // flipped by k_cov_flip after every batch that owns tiles of this lane
    const uint32_t* cin = carry + ((uint64_t)t.lane * 2 + par) * 2000;
    uint32_t* cout = carry + ((uint64_t)t.lane * 2 + (par ^ 1u)) * 2000;
    int32_t run = off;
    for (uint32_t j = 0; j < per; ++j) {
        const uint32_t p = s0 + j;
        if (p >= BQC_COV_TILE) break;
        run += diff[p];
        const int64_t vp = lo + p;
        uint32_t depth = (uint32_t)run;
        if (vp < 2 * BQC_VSIZE) depth += cin[vp]; // partial windows carried over from the previous batch
        const uint32_t win = t.win_lo + p / BQC_VSIZE;
        if (win < t.win_final) atomicAdd(&hist[depth > BQC_COVSIZE ? BQC_COVSIZE : depth], 1u); // update_coverage :66-77
        else if (win < t.win_final + 2) cout[(win - t.win_final) * BQC_VSIZE + p % BQC_VSIZE] = depth;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= BQC_COVSIZE; i += blockDim.x)
        if (hist[i]) gadd(state + sl.lane_base(t.lane) + sl.o_poscov + i, hist[i]);
}

// end of stream: histogram the two live windows of every started lane (bamqualcheck.cpp:447-453)
__global__ __launch_bounds__(256) void k_cov_final(StateLayout sl, uint64_t* __restrict__ state, const uint32_t* __restrict__ carry,
                                                      const uint32_t* __restrict__ parity, const uint8_t* __restrict__ started)
{
    __shared__ uint32_t hist[BQC_COVSIZE + 1];
    const uint32_t lane = blockIdx.x;
    if (!started[lane]) return;
    for (uint32_t i = threadIdx.x; i <= BQC_COVSIZE; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const uint32_t* c = carry + ((uint64_t)lane * 2 + (parity[lane] & 1u)) * 2000;
    for (uint32_t i = threadIdx.x; i < 2000; i += blockDim.x) {
        const uint32_t d = c[i];
        atomicAdd(&hist[d > BQC_COVSIZE ? BQC_COVSIZE : d], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= BQC_COVSIZE; i += blockDim.x)
        if (hist[i]) gadd(state + sl.lane_base(lane) + sl.o_poscov + i, hist[i]);
    if (threadIdx.x == 0) gadd(state + sl.lane_base(lane) + sl.o_covstart, 1);
}

__global__ void k_cov_flip(uint32_t* __restrict__ parity, const uint8_t* __restrict__ lane_mask, uint32_t n_lanes)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l < n_lanes && lane_mask[l]) parity[l] ^= 1u;
}

__global__ void k_or_bytes(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && src[i]) dst[i] = 1;
}

__global__ void k_add_words(uint64_t* __restrict__ state, const uint64_t* __restrict__ idx, const uint64_t* __restrict__ val, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gadd(state + idx[i], val[i]);
}

// ---------------------------------------------------------------------------------------------------
// launchers (called from bqc_api.cpp)
// ---------------------------------------------------------------------------------------------------
extern "C" void bqc_launch_reads(const DevBatch& b, const StateLayout& sl, uint64_t* state, const DevRefs& refs, uint32_t* err,
                                 hipStream_t s)
{
    if (b.n_reads == 0) return;
    hipLaunchKernelGGL(k_reads, dim3((b.n_reads + 255) / 256), dim3(256), 0, s, b, sl, state, refs, err);
    if (b.n_nm_extra)
        hipLaunchKernelGGL(k_nm_extra, dim3((b.n_nm_extra + 255) / 256), dim3(256), 0, s, b, sl, state, refs, err);
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

extern "C" void bqc_launch_cov(const DevBatch& b, const StateLayout& sl, uint64_t* state, uint32_t* carry, const uint32_t* parity,
                               hipStream_t s)
{
    if (b.n_cov_tiles == 0) return;
    hipLaunchKernelGGL(k_cov, dim3(b.n_cov_tiles), dim3(256), 0, s, b, sl, state, carry, parity);
}

extern "C" void bqc_launch_cov_flip(uint32_t* parity, const uint8_t* lane_mask, uint32_t n_lanes, hipStream_t s)
{
    hipLaunchKernelGGL(k_cov_flip, dim3((n_lanes + 255) / 256), dim3(256), 0, s, parity, lane_mask, n_lanes);
}

extern "C" void bqc_launch_cov_final(const StateLayout& sl, uint64_t* state, const uint32_t* carry, const uint32_t* parity,
                                     const uint8_t* started, hipStream_t s)
{
    hipLaunchKernelGGL(k_cov_final, dim3(sl.n_lanes), dim3(256), 0, s, sl, state, carry, parity, started);
}

extern "C" void bqc_launch_add_words(uint64_t* state, const uint64_t* idx, const uint64_t* val, uint32_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_add_words, dim3((n + 255) / 256), dim3(256), 0, s, state, idx, val, n);
}

extern "C" void bqc_launch_or_bytes(uint8_t* dst, const uint8_t* src, uint32_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_or_bytes, dim3((n + 255) / 256), dim3(256), 0, s, dst, src, n);
}
