// read_stats.h — per-read statistics as a device function with LDS-privatised counters, shared by
// k_reads (thread per read of the generic chunks) and k_short (lane per read of a wave's tile, phase A).
#pragma once
#include "kernels_common.h"

// LDS map (uint32 words) of the per-read counters of ONE read group ("lane")
#define RS_CT     BQC_CT                      // read lengths / clip lengths up to here stay in LDS
#define RS_HB     64                          // mismatch / deletion / insertion bins kept in LDS
#define RS_INS    1024                        // insert-size bins kept in LDS (mate 0 only)
#define RS_SCAL   0                           // [16] scalars
#define RS_MATE   16                          // per-mate block follows
#define RS_M_READNR 0
#define RS_M_READLEN (RS_M_READNR + 1)        // [RS_CT + 1]
#define RS_M_MAPQ   (RS_M_READLEN + RS_CT + 1)  // [256]
#define RS_M_MM     (RS_M_MAPQ + 256)         // [RS_HB]
#define RS_M_DEL    (RS_M_MM + RS_HB)
#define RS_M_INS    (RS_M_DEL + RS_HB)
#define RS_M_SC5    (RS_M_INS + RS_HB)        // [RS_CT + 1] histogram of min(leading clip, L)
#define RS_M_SC3    (RS_M_SC5 + RS_CT + 1)    // [RS_CT + 1] difference array (+1 / -1 as wrapping u32)
#define RS_M_WORDS  (RS_M_SC3 + RS_CT + 1)
#define RS_INSERT   (RS_MATE + 2 * RS_M_WORDS) // [RS_INS]
#define RS_WORDS    (RS_INSERT + RS_INS)

// count lanes with pred into one LDS word (pure scalars: every lane hits the same address)
__device__ __forceinline__ void rs_count(bool pred, uint32_t* w)
{
    const uint64_t m = __ballot(pred);
    if (m && lane_id() == (__ffsll((unsigned long long)m) - 1)) atomicAdd(w, (uint32_t)__popcll((unsigned long long)m));
}

// histogram increment: LDS if this thread is on the privatised lane and the bin fits, else global
__device__ __forceinline__ void rs_hist(bool pred, bool use_lds, uint32_t bin, uint32_t cap, uint32_t* lds_base, uint64_t* g_base)
{
    const bool in_lds = pred && use_lds && bin < cap;
    if (in_lds) atomicAdd(lds_base + bin, 1u);
    wave_inc(pred && !in_lds, g_base + bin);
}

__device__ __forceinline__ void read_stats(const DevBatch& b, const StateLayout& sl, uint64_t* __restrict__ state, const DevRefs& refs,
                                           uint32_t* __restrict__ err, uint32_t* lds, uint32_t r, bool live, bool use_lds)
{
    uint32_t flag = 0, L = 0, mapq = 0, ncig = 0, lane = 0;
    int32_t rid = -1, tlen = 0, nm = BQC_NM_ABSENT;
    if (live) {
        flag = b.flag[r]; L = b.l_seq[r]; mapq = b.mapq[r]; ncig = b.n_cigar[r]; lane = b.lane[r];
        rid = b.rid[r]; tlen = b.tlen[r]; nm = b.nm[r];
    }
    uint64_t* S = state + sl.lane_base(lane) + sl.o_scalars;
    uint32_t* LS = lds + RS_SCAL;
    const bool gl = live && !use_lds; // global path (other lane inside a mixed wave)
    // ---- flag cascade, bamqualcheck.cpp:318-335
    const bool supp = live && (flag & 0x800);
    const bool sec = live && !supp && (flag & 0x100);
    const bool prim = live && !(flag & 0x900);
    const bool dup = prim && (flag & 0x400), qcf = prim && (flag & 0x200);
    // ---- :353-389
    const bool first = prim && (flag & 0x40);
    const bool last = prim && !first && (flag & 0x80);
    if (prim && !first && !last) atomicOr(err, BQC_DEVERR_MATE);
    const bool unm = flag & 0x4, nunm = flag & 0x8, proper = flag & 0x2;
    const bool rc = flag & 0x10, nrc = flag & 0x20;
    const bool mated = first || last;
    const uint32_t mate = first ? 0u : 1u;
    const int32_t nrefs = (int32_t)refs.n_refs;
    const bool in_main = mated && rid >= 0 && rid < nrefs && refs.main_chrom[rid];
    const bool fasm = first && in_main && (!unm || !nunm) && !(flag & 0x400);
    const bool autop = first && in_main && proper && !(flag & 0x400);
#define RS_SC(pred, idx)                       \
    do {                                       \
        rs_count((pred) && use_lds, LS + (idx)); \
        wave_inc((pred) && gl, S + (idx));     \
    } while (0)
    RS_SC(supp, BQC_S_SUPPLEMENTARY);
    RS_SC(sec, BQC_S_NOT_PRIMARY);
    RS_SC(dup, BQC_S_DUPLICATES);
    RS_SC(qcf, BQC_S_QCFAILED);
    RS_SC(prim, BQC_S_READCOUNT);
    RS_SC(first && unm, BQC_S_FIRSTUNMAPPED);
    RS_SC(first && unm && nunm, BQC_S_BOTHUNMAPPED);
    RS_SC(first && proper, BQC_S_PROPERPAIR);
    RS_SC(first && proper && (rc == nrc), BQC_S_FF_RR);
    RS_SC(last && unm, BQC_S_SECONDUNMAPPED);
    RS_SC(fasm, BQC_S_FIRST_AND_OR_SECOND_MAPPED);
    RS_SC(autop, BQC_S_AUTO_PROPERPAIR);
#undef RS_SC
    { // totalbps += L (:354): 64-bit sum kept in two LDS words [14] (low 16 bits summed) and [15] (high bits)
        const bool p = prim && use_lds;
        if (__ballot(p)) {
            const uint32_t lo = wave_sum(p ? (L & 0xFFFFu) : 0u), hi = wave_sum(p ? (L >> 16) : 0u);
            if (lane_id() == 0) { // per wave at most 64 * 65535 < 2^22: the LDS word cannot overflow before ~1000 waves
                const uint32_t o = atomicAdd(LS + 14, lo);
                if (o + lo < o) atomicAdd(LS + 15, 1u << 16); // carry of the low word, kept in units of 2^16
                atomicAdd(LS + 15, hi);
            }
        }
        if (prim && gl) gadd(S + BQC_S_TOTALBPS, L);
    }
    uint64_t* M = state + sl.mate_base(lane, mate);
    uint32_t* LM = lds + RS_MATE + mate * RS_M_WORDS;
    // read_length + qualcount_readnr (QualityCheck.hpp:130,168-176)
    rs_count(first && use_lds, lds + RS_MATE + RS_M_READNR);              // rs_count: one address per call
    rs_count(last && use_lds, lds + RS_MATE + RS_M_WORDS + RS_M_READNR);
    wave_inc(mated && gl, M + sl.m_readnr);
    rs_hist(mated && L <= sl.lcap, use_lds, L <= sl.lcap ? L : 0, RS_CT + 1, LM + RS_M_READLEN, M + sl.m_readlen);
    // ---- main chromosomes only, :392-434
    const bool mapped_main = in_main && !unm;
    uint32_t del = 0, ins = 0;
    bool sc5 = false, sc3 = false;
    uint32_t n5 = 0, n3 = 0;
    if (mapped_main) { // cigar_count (QualityCheck.hpp:222-271) on the seq-oriented (reversed for RC) CIGAR
        const uint32_t* cg = b.cigar + b.cigar_off[r];
        if (ncig > 0) {
            const uint32_t c_first = rc ? cg[ncig - 1] : cg[0];
            const uint32_t c_last = rc ? cg[0] : cg[ncig - 1];
            if ((c_first & 15u) == 4u) { // 'S': sc5[j]++ for j < n  <=>  histogram of n, suffix-summed at finalize
                sc5 = true;
                n5 = min(c_first >> 4, L);
            } else if ((c_last & 15u) == 4u) { // for (j = L-n; j < L; ++j) sc3[j]++   as a difference array
                n3 = c_last >> 4;
                sc3 = n3 <= L && n3 > 0;
            }
            for (uint32_t k = 0; k < ncig; ++k) {
                const uint32_t c = cg[k], op = c & 15u;
                if (op == 2u) del += c >> 4;       // 'D'
                else if (op == 1u) ins += c >> 4;  // 'I'
            }
        }
        if (del >= sl.hcap || ins >= sl.hcap) atomicOr(err, BQC_DEVERR_RANGE);
    }
    if (sc5) {
        if (use_lds && n5 <= RS_CT) atomicAdd(LM + RS_M_SC5 + n5, 1u);
        else gadd(M + sl.m_sc5hist + n5, 1);
    }
    if (sc3) {
        if (use_lds && L <= RS_CT) { atomicAdd(LM + RS_M_SC3 + (L - n3), 1u); atomicAdd(LM + RS_M_SC3 + L, 0xFFFFFFFFu); }
        else { gadd(M + sl.m_sc3diff + (L - n3), 1); gadd(M + sl.m_sc3diff + L, (uint64_t)-1ll); }
    }
    const bool hist_ok = mapped_main && del < sl.hcap && ins < sl.hcap;
    rs_hist(hist_ok, use_lds, hist_ok ? del : 0, RS_HB, LM + RS_M_DEL, M + sl.m_delhist);
    rs_hist(hist_ok, use_lds, hist_ok ? ins : 0, RS_HB, LM + RS_M_INS, M + sl.m_inshist);
    rs_hist(mapped_main, use_lds, mapq, 256, LM + RS_M_MAPQ, M + sl.m_mapq); // map_Q :178-185
    { // mis_match :198-220
        const bool has = mapped_main && nm != BQC_NM_ABSENT;
        const uint32_t mm = (uint32_t)nm - del - ins; // unsigned arithmetic (:210)
        if (has && mm >= sl.hcap) atomicOr(err, BQC_DEVERR_RANGE);
        const bool ok = has && mm < sl.hcap;
        rs_hist(ok, use_lds, ok ? mm : 0, RS_HB, LM + RS_M_MM, M + sl.m_mismatch);
    }
    if (first && mapped_main && !nunm && (flag & BQC_FLAG_MATE_MAIN)) { // insert_size :187-196
        uint32_t idx = tlen < 0 ? (uint32_t)0 - (uint32_t)tlen : (uint32_t)tlen; // abs(INT_MIN) -> 2^31
        if (idx >= sl.icap) idx = sl.icap - 1;
        if (use_lds && idx < RS_INS) atomicAdd(lds + RS_INSERT + idx, 1u);
        else gadd(M + sl.m_insert + idx, 1);
    }
}

// flush the privatised counters of `lane` to the global state and zero them (call block-uniformly,
// preceded and followed by __syncthreads by the caller as needed)
__device__ __forceinline__ void rs_flush(uint32_t* lds, const StateLayout& sl, uint64_t* __restrict__ state, uint32_t lane)
{
    block_sync();
    const uint64_t lb = sl.lane_base(lane);
    for (uint32_t i = threadIdx.x; i < RS_WORDS; i += blockDim.x) {
        const uint32_t v = lds[i];
        if (!v) continue;
        lds[i] = 0;
        if (i < RS_MATE) {
            if (i < 14) { if (i != BQC_S_TOTALBPS) gadd(state + lb + sl.o_scalars + i, v); }
            else if (i == 14) gadd(state + lb + sl.o_scalars + BQC_S_TOTALBPS, v);
            else gadd(state + lb + sl.o_scalars + BQC_S_TOTALBPS, (uint64_t)v << 16);
        } else if (i < RS_INSERT) {
            const uint32_t m = (i - RS_MATE) / RS_M_WORDS, j = (i - RS_MATE) % RS_M_WORDS;
            uint64_t* M = state + sl.mate_base(lane, m);
            if (j == RS_M_READNR) gadd(M + sl.m_readnr, v);
            else if (j < RS_M_MAPQ) gadd(M + sl.m_readlen + (j - RS_M_READLEN), v);
            else if (j < RS_M_MM) gadd(M + sl.m_mapq + (j - RS_M_MAPQ), v);
            else if (j < RS_M_DEL) gadd(M + sl.m_mismatch + (j - RS_M_MM), v);
            else if (j < RS_M_INS) gadd(M + sl.m_delhist + (j - RS_M_DEL), v);
            else if (j < RS_M_SC5) gadd(M + sl.m_inshist + (j - RS_M_INS), v);
            else if (j < RS_M_SC3) gadd(M + sl.m_sc5hist + (j - RS_M_SC5), v);
            else gadd(M + sl.m_sc3diff + (j - RS_M_SC3), (uint64_t)(int64_t)(int32_t)v); // signed net of +1 / -1
        } else {
            gadd(state + sl.mate_base(lane, 0) + sl.m_insert + (i - RS_INSERT), v);
        }
    }
    block_sync();
}
