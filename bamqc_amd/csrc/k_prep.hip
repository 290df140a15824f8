// k_prep.hip — the per-read pre-pass and the work decomposition of a batch, on the device.
//
// What the reference does record by record before its counters see a read is done here for a whole batch:
//   k_prep_sizes / k_prep_scan   payload offsets of every read (prefix sums of ceil(L/2), L, n_cigar)
//   k_prep_reads                 lane per read: quality-missing flag, checkFlagsAndQuality (TripletCounting.hpp:136-168)
//                                incl. the forward-only FASTA scan (:254-259) inside a block, the covered interval(s) of
//                                OverallNumbers::coverage (OverallNumbers.hpp:112-131) from the host's anchor (win, pos), the triplet
//                                segments of multi-operation CIGARs (TripletCounting.hpp:203-232), and the checks whose failure ends
//                                the reference's run (bamqualcheck.cpp:340,385-389)
//   k_build_count / _plan / _scatter   chunk tables and the entry order (`perm`) of k_short / k_reads / k_long: reads are ranked by
//                                mate slot inside super-windows of BQC_SW_READS stream positions, so that a read group of k_short holds
//                                first-mate and second-mate reads of one short stretch of the stream (shared 128-byte lines) while
//                                every lane of k_short only ever sees one mate.
// Only the O(1)-per-read coverage anchor recurrence (OverallNumbers.hpp:84-110) stays on the host (bqc_pipeline.cpp).
#include <algorithm>
#include "kernels_common.h"
#include "prep.h"

#define PR_THREADS 256
#define PR_PER_THREAD 4
#define PR_BLOCK (PR_THREADS * PR_PER_THREAD) // reads per workgroup of the per-read kernels

// ---------------------------------------------------------------------------------------------------
// scans over the 256 threads of a workgroup (4 waves): DPP inside a wave, four LDS words across
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_scan_incl_max(uint32_t v) // unsigned, identity 0
{
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
    return v;
}
// exclusive sum over the workgroup's threads in thread order; *total = sum over all threads.  `sh`: 2 * waves words of LDS.
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* sh, uint32_t* total)
{
    const uint32_t inc = wave_scan_incl(v), w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    block_sync();
    if (lane_id() == WAVE - 1) sh[w] = inc;
    block_sync();
    uint32_t base = 0, tot = 0;
    for (uint32_t k = 0; k < nw; ++k) { const uint32_t x = sh[k]; if (k < w) base += x; tot += x; }
    *total = tot;
    return base + inc - v;
}
__device__ __forceinline__ uint32_t block_scan_excl_max(uint32_t v, uint32_t* sh, uint32_t* total)
{
    const uint32_t inc = wave_scan_incl_max(v), w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    block_sync();
    if (lane_id() == WAVE - 1) sh[w] = inc;
    block_sync();
    uint32_t base = 0, tot = 0;
    for (uint32_t k = 0; k < nw; ++k) { const uint32_t x = sh[k]; if (k < w) base = max(base, x); tot = max(tot, x); }
    *total = tot;
    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x138 /* wave_shr:1 */, 0xF, 0xF, true); // inclusive value of the lane before
    return max(base, prev);
}
__device__ __forceinline__ void err_key(ErrRec* e, uint32_t read, uint32_t order)
{
    atomicMin(&e->first_key, ((unsigned long long)read << 3) | order);
}

// ---------------------------------------------------------------------------------------------------
// payload sizes -> offsets
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PR_THREADS) void k_prep_sizes(PrepArgs a)
{
    __shared__ unsigned long long sh[3 * (PR_THREADS / 64)];
    if (blockIdx.x == 0 && threadIdx.x == 0) { // first kernel of the batch: its records start empty; a replayed batch starts from its own cursor
        a.err->first_key = BQC_ERRKEY_NONE; a.err->flags = 0; a.err->aux0 = a.err->aux1 = 0;
        a.desc->n_cov_extra = 0;
        if (a.pend_extra_n) *a.pend_extra_n = 0;
        if (a.replay) *a.cursor = *a.cursor_save; else *a.cursor_save = *a.cursor;
    }
    for (uint32_t k = blockIdx.x * PR_THREADS + threadIdx.x; k < a.n_sw; k += gridDim.x * PR_THREADS) a.sw_counts[k] = SwCounts{0, 0, 0, 0};
    const uint32_t b0 = blockIdx.x * PR_BLOCK;
    unsigned long long s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
    for (int j = 0; j < PR_PER_THREAD; ++j) {
        const uint32_t i = b0 + j * PR_THREADS + threadIdx.x;
        if (i < a.n) { const uint32_t L = a.l_seq[i]; s1 += (L + 1) / 2; s2 += L; s3 += a.n_cigar[i]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); s3 += __shfl_xor(s3, o); }
    if (lane_id() == 0) { const uint32_t w = threadIdx.x >> 6; sh[3 * w] = s1; sh[3 * w + 1] = s2; sh[3 * w + 2] = s3; }
    block_sync();
    if (threadIdx.x < 3) {
        unsigned long long t = 0;
        for (uint32_t w = 0; w < PR_THREADS / 64; ++w) t += sh[3 * w + threadIdx.x];
        a.blk_sizes[3ull * blockIdx.x + threadIdx.x] = t;
    }
}

// one workgroup: block sums -> exclusive bases, in place; a batch whose payload does not fit 32-bit offsets is an error
__global__ __launch_bounds__(1024) void k_prep_scan(PrepArgs a)
{
    __shared__ unsigned long long sh[3 * 1024];
    const uint32_t nblk = (a.n + PR_BLOCK - 1) / PR_BLOCK;
    const uint32_t per = (nblk + blockDim.x - 1) / blockDim.x;
    const uint32_t lo = min(nblk, threadIdx.x * per), hi = min(nblk, lo + per);
    unsigned long long t[3] = {0, 0, 0};
    for (uint32_t b = lo; b < hi; ++b)
        for (int k = 0; k < 3; ++k) t[k] += a.blk_sizes[3ull * b + k];
    for (int k = 0; k < 3; ++k) sh[3 * threadIdx.x + k] = t[k];
    block_sync();
    for (uint32_t d = 1; d < blockDim.x; d <<= 1) { // Hillis-Steele over the thread totals
        unsigned long long v[3] = {0, 0, 0};
        if (threadIdx.x >= d) for (int k = 0; k < 3; ++k) v[k] = sh[3 * (threadIdx.x - d) + k];
        block_sync();
        for (int k = 0; k < 3; ++k) sh[3 * threadIdx.x + k] += v[k];
        block_sync();
    }
    unsigned long long run[3];
    for (int k = 0; k < 3; ++k) run[k] = sh[3 * threadIdx.x + k] - t[k];
    for (uint32_t b = lo; b < hi; ++b) {
        bool over = false;
        for (int k = 0; k < 3; ++k) {
            const unsigned long long s = a.blk_sizes[3ull * b + k];
            a.blk_sizes[3ull * b + k] = run[k];
            run[k] += s;
            over |= run[k] > 0xFFFFFFFFull;
        }
        if (over) err_key(a.err, b * PR_BLOCK, 3);
    }
}

// ---------------------------------------------------------------------------------------------------
// per read
// ---------------------------------------------------------------------------------------------------
// checkFlagsAndQuality (TripletCounting.hpp:136-168): 1 eligible, 0 not, -1 fatal
// A read's CIGAR as the pre-pass sees it: the first four words in registers — requested for all of a thread's reads at once, before any
// of them is looked at (round 4: read where they were needed, behind the stores and atomics of the read before, they were a chain of
// dependent round trips per thread: the kernel's whole time) —, the rest from memory.
struct CigarView {
    uint32_t w[4];
    const uint32_t* cg;
    __device__ __forceinline__ uint32_t at(uint32_t k) const
    {
        if (k < 4u) return k == 0u ? w[0] : k == 1u ? w[1] : k == 2u ? w[2] : w[3];
        return cg[k];
    }
};
__device__ __forceinline__ int triplet_eligible(uint32_t flag, uint32_t mapq, int32_t as, const CigarView& cv, uint32_t ncig)
{
    if (!(flag & 0x1) || !(flag & 0x2) || (flag & 0x4) || (flag & 0x8) || (flag & 0x100)) return 0;
    if (mapq < 60) return 0;
    if (as == BQC_AS_ABSENT || as < 0) return -1;
    if (as < 50) return 0;
    uint32_t clipped = 0;
    for (uint32_t k = 0; k < ncig; ++k) {
        const uint32_t wd = cv.at(k), op = wd & 15u;
        if (op == 4u || op == 5u) clipped += wd >> 4;
    }
    return clipped > 0 ? 0 : 1;
}

// One read's share of the pre-pass.  Everything but the payload offsets and the cross-read part of the FASTA scan.
struct ReadIn { uint32_t L, nc, flag, lane, mapq; int32_t rid, pos, as; CovEntry ce; };
struct ReadOut { uint32_t flag, cls, tgt1; CovEntry ce; };
__device__ __forceinline__ ReadOut prep_one(const PrepArgs& a, const DevRefs& refs, uint32_t i, const ReadIn& in, uint32_t co, bool offsets_ok,
                                            const CigarView& cv, bool ref_loaded /* refs.ref[rid] != nullptr (rid in range) */, int32_t target_in /* the contig's FASTA record, -1: none */)
{
    ReadOut o;
    uint32_t flag = in.flag & (0x0FFFu | BQC_FLAG_MATE_MAIN | BQC_FLAG_NO_QUAL);
    const uint32_t L = in.L, lane = in.lane;
    uint32_t nc = in.nc;
    o.tgt1 = 0;
    if (L > a.max_read_len) err_key(a.err, i, 1);
    else if (lane >= a.n_lanes) err_key(a.err, i, 2);
    // (BQC_FLAG_NO_QUAL — quality block starting with 0xFF, SURVEY U1 — is a fact of the record's decoding and arrives with the
    // flag column: probing qual[qo] here cost one 128-byte line per read, more than every other byte this kernel reads)
    if (!offsets_ok) nc = 0; // (the batch fails: k_prep_scan)
    const bool fast = !a.no_fast && L <= BQC_FAST_MAXLEN;
    uint32_t nseg = 0;
    CovEntry ce = in.ce;
    if (!(flag & 0x900u)) { // primary record: bamqualcheck.cpp:318-327
        const bool dup = flag & 0x400u, qcf = flag & 0x200u;
        if (!dup && !qcf) { // tripletCounting, :338-342
            const int e = triplet_eligible(flag, in.mapq, in.as, cv, nc);
            if (e < 0) err_key(a.err, i, 4);
            if (e > 0) { // Genome: forward-only FASTA scan (TripletCounting.hpp:254-259)
                const int32_t target = target_in;
                if (target < 0 || !ref_loaded) err_key(a.err, i, 5);
                else { o.tgt1 = (uint32_t)target + 1u; flag |= BQC_FLAG_TRIPLET; }
            }
        }
        if (!(flag & 0xC0u)) err_key(a.err, i, 6);
    }
    // The read's covered interval(s) relative to its first live window (OverallNumbers.hpp:112-131): `c` runs over the
    // seq-oriented CIGAR (reversed for reverse reads, bamqualcheck.cpp:349) and advances on S, M and D; M and D add
    // coverage.  DEFINED: increments at window offset >= 2000 are dropped.  One interval unless a clip sits between
    // two match operations.  The host decided WHETHER the read enters coverage() and where its window starts.
    if (ce.win != BQC_COV_NONE) {
        flag |= BQC_FLAG_COV;
        const bool rc = flag & 0x10u;
        const bool pending = ce.win == BQC_COV_PENDING; // shard mode: the window comes later (bqc_shard_resolve): runs relative to beginPos
        const uint32_t pend_idx = ce.off_len;
        const int64_t pos = pending ? 0 : (int64_t)ce.off_len;
        uint32_t cc = 0; // `int c` in the reference; wraps identically
        int64_t run_a = -1, run_z = -1;
        bool first = true;
        ce.off_len = 0;
        auto emit = [&](int64_t lo, int64_t hi) {
            if (pending) { // (pos <= 2000 later: a run that starts 2000 or more behind beginPos can never count)
                if (lo < 0 || lo >= 2 * BQC_VSIZE || lo >= hi) return;
                const uint32_t len = (uint32_t)(hi - lo < 2 * BQC_VSIZE ? hi - lo : 2 * BQC_VSIZE);
                if (first) { a.pend[pend_idx] = PendRun{(uint32_t)lo, len}; first = false; return; }
                const uint32_t k = atomicAdd(a.pend_extra_n, 1u);
                if (k < a.pend_extra_cap) a.pend_extra[k] = PendExtra{pend_idx, (uint32_t)lo, len, 0};
                else atomicOr(&a.err->flags, BQC_DEVERR_INTERNAL);
                return;
            }
            hi = hi < 2 * BQC_VSIZE ? hi : 2 * BQC_VSIZE;
            if (lo < 0 || lo >= hi) return;
            const uint32_t v = (uint32_t)lo | ((uint32_t)(hi - lo) << 16);
            if (first) { ce.off_len = v; first = false; return; }
            const uint32_t k = atomicAdd(&a.desc->n_cov_extra, 1u);
            if (k < a.cov_extra_cap) a.cov_extra[k] = CovExtra{ce.win, v, lane, 0};
            else atomicOr(&a.err->flags, BQC_DEVERR_INTERNAL); // (capacity = CIGAR words / 2 + 1: cannot happen)
        };
        if (nc == 1u) { // (nearly every read: no loop)
            const uint32_t w = cv.w[0], op = w & 15u, nn = w >> 4;
            if (op == 0u || op == 2u) emit(pos, pos + nn);
        } else {
            for (uint32_t k = 0; k < nc; ++k) {
                const uint32_t w = cv.at(rc ? nc - 1 - k : k), op = w & 15u, nn = w >> 4;
                if (op == 4u) cc += nn;
                if (op == 0u || op == 2u) {
                    const int64_t lo = pos + (int64_t)cc, hi = lo + nn;
                    if (run_z == lo) run_z = hi;
                    else { if (run_a >= 0) emit(run_a, run_z); run_a = lo; run_z = hi; }
                    cc += nn;
                }
            }
            if (run_a >= 0) emit(run_a, run_z);
        }
        if (pending) { if (first) a.pend[pend_idx] = PendRun{0, 0}; ce = CovEntry{BQC_COV_NONE, 0}; }
    } else ce.off_len = 0;
    // k_short evaluates triplets with chromPos = pos + i inside the first CIGAR operation (assumed match-like,
    // TripletCounting.hpp:203); every further match-like operation becomes a segment entry with its own offset
    if (fast && (flag & BQC_FLAG_TRIPLET) && nc > 1 && L >= 3) {
        const uint32_t n0 = cv.w[0] >> 4;
        if (n0 != 0) { // (n0 == 0: every position counts as inside the first operation, no walk)
            uint64_t rp = n0;
            int64_t cpos = (int64_t)in.pos + n0;
            for (uint32_t k2 = 1; k2 < nc && rp < L; ++k2) {
                const uint32_t wk = cv.at(k2), op = wk & 15u, nn = wk >> 4;
                if (op == 2u || op == 3u || op == 5u || op == 6u) cpos += nn;   // D N H P
                else if (op == 4u || op == 1u) rp += nn;                          // S I
                else {                                                            // M = X (and unknown)
                    const uint64_t ia = rp > 1 ? rp : 1, ib = rp + nn < (uint64_t)L - 1 ? rp + nn : (uint64_t)L - 1;
                    const int64_t posv = cpos - (int64_t)rp;
                    if (ia < ib && posv > INT32_MIN / 2 && posv < INT32_MAX / 2) {
                        a.segs[co + nseg] = TripSeg{i, (int32_t)posv, (uint32_t)ia | ((uint32_t)ib << 8), 0};
                        ++nseg;
                    }
                    rp += nn; cpos += nn;
                }
            }
        }
    }
    o.flag = flag;
    o.ce = ce;
    o.cls = ((fast ? ((flag & 0x40u) ? 0u : 1u) : 2u) << 8) | nseg; // mate slot of a fast read (first mate / everything else) or 2 = generic path; segments
    return o;
}

// Thread t of a workgroup owns the four consecutive reads b0 + 4t .. + 3: their columns are 8- / 16-byte loads, the payload
// offsets are a serial prefix inside the thread plus ONE workgroup scan per quantity, and so is the forward-only FASTA scan.
#ifndef PR_WAVES_PER_EU
#define PR_WAVES_PER_EU 4
#endif
__global__ __launch_bounds__(PR_THREADS) __attribute__((amdgpu_waves_per_eu(PR_WAVES_PER_EU, PR_WAVES_PER_EU))) void k_prep_reads(PrepArgs a, DevRefs refs)
{
    __shared__ uint32_t sh[2 * (PR_THREADS / 64)];
    __shared__ uint32_t red[8];
    const uint32_t i0 = blockIdx.x * PR_BLOCK + 4u * threadIdx.x;
    const unsigned long long gs = a.blk_sizes[3ull * blockIdx.x], gq = a.blk_sizes[3ull * blockIdx.x + 1], gc = a.blk_sizes[3ull * blockIdx.x + 2];
    if (threadIdx.x < 8) red[threadIdx.x] = threadIdx.x == 0 ? 0xFFFFFFFFu : 0u;
    ReadIn in[4];
    const uint32_t nlive = i0 >= a.n ? 0u : min(4u, a.n - i0);
    if (nlive == 4u) {
        const uint4 vl = *(const uint4*)(a.l_seq + i0);
        const uint2 vc = *(const uint2*)(a.n_cigar + i0), vf = *(const uint2*)(a.flag_in + i0);
        const uint32_t vlane = *(const uint32_t*)(a.lane + i0), vmq = *(const uint32_t*)(a.mapq + i0);
        const int4 vr = *(const int4*)(a.rid + i0), vp = *(const int4*)(a.pos + i0), va = *(const int4*)(a.as_ + i0);
        const uint4 c01 = *(const uint4*)(a.cov_in + i0), c23 = *(const uint4*)(a.cov_in + i0 + 2);
        const uint32_t Ls[4] = {vl.x, vl.y, vl.z, vl.w}, ncs[4] = {vc.x & 0xFFFFu, vc.x >> 16, vc.y & 0xFFFFu, vc.y >> 16};
        const uint32_t fs[4] = {vf.x & 0xFFFFu, vf.x >> 16, vf.y & 0xFFFFu, vf.y >> 16};
        const int32_t rs[4] = {vr.x, vr.y, vr.z, vr.w}, ps[4] = {vp.x, vp.y, vp.z, vp.w}, as[4] = {va.x, va.y, va.z, va.w};
        const CovEntry cs[4] = {{c01.x, c01.y}, {c01.z, c01.w}, {c23.x, c23.y}, {c23.z, c23.w}};
#pragma unroll
        for (int j = 0; j < 4; ++j) in[j] = ReadIn{Ls[j], ncs[j], fs[j], (vlane >> (8 * j)) & 0xFFu, (vmq >> (8 * j)) & 0xFFu, rs[j], ps[j], as[j], cs[j]};
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            in[j] = ReadIn{0, 0, 0x900u, 0, 0, -1, 0, 0, CovEntry{BQC_COV_NONE, 0}};
            if ((uint32_t)j < nlive) { const uint32_t i = i0 + j; in[j] = ReadIn{a.l_seq[i], a.n_cigar[i], a.flag_in[i], a.lane[i], a.mapq[i], a.rid[i], a.pos[i], a.as_[i], a.cov_in[i]}; }
        }
    }
    // payload offsets: prefix inside the thread, then across the workgroup
    uint32_t ls[4], lq[4], lc[4], ts = 0, tq = 0, tc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { ls[j] = ts; lq[j] = tq; lc[j] = tc; ts += (in[j].L + 1) / 2; tq += in[j].L; tc += in[j].nc; }
    uint32_t tot;
    const unsigned long long bs = gs + block_scan_excl(ts, sh, &tot), bq = gq + block_scan_excl(tq, sh, &tot), bc = gc + block_scan_excl(tc, sh, &tot);
    ReadOut out[4];
    // what the four reads need from memory besides their columns, all of it requested before the first read is looked at
    CigarView cv[4];
    bool ref_loaded[4];
    int32_t target[4];
    unsigned long long cos[4];
    bool oks[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned long long so = bs + ls[j], qo = bq + lq[j], co = bc + lc[j];
        cos[j] = co;
        oks[j] = so + (in[j].L + 1) / 2 <= 0xFFFFFFFFull && qo + in[j].L <= 0xFFFFFFFFull && co + in[j].nc <= 0xFFFFFFFFull;
        const bool live = (uint32_t)j < nlive && oks[j];
        cv[j].cg = a.cigar + (uint32_t)co;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) cv[j].w[k] = live && k < in[j].nc ? cv[j].cg[k] : 0u;
        const int32_t rid = in[j].rid;
        const bool rid_ok = (uint32_t)j < nlive && rid >= 0 && (uint32_t)rid < refs.n_refs;
        ref_loaded[j] = rid_ok && refs.ref[rid] != nullptr;
        target[j] = rid_ok ? (a.fasta_index ? a.fasta_index[rid] : rid) : -1;
    }
    uint32_t tmax = 0, tmin = 0xFFFFFFFFu, maxfast = 0, maxlong = 0, cnt[4] = {0, 0, 0, 0};
    bool local_bad[4] = {false, false, false, false};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        out[j] = ReadOut{0x900u, 2u << 8, 0, CovEntry{BQC_COV_NONE, 0}};
        if ((uint32_t)j >= nlive) continue;
        out[j] = prep_one(a, refs, i0 + j, in[j], (uint32_t)cos[j], oks[j], cv[j], ref_loaded[j], target[j]);
        if (out[j].tgt1) { local_bad[j] = out[j].tgt1 < tmax; tmax = max(tmax, out[j].tgt1); tmin = min(tmin, out[j].tgt1); }
        const uint32_t cl = out[j].cls >> 8;
        if (cl == 2u) maxlong = max(maxlong, in[j].L); else maxfast = max(maxfast, in[j].L);
        cnt[cl] += 1; cnt[3] += out[j].cls & 0xFFu;
    }
    if (nlive == 4u) {
        *(uint4*)(a.seq_off + i0) = make_uint4((uint32_t)bs + ls[0], (uint32_t)bs + ls[1], (uint32_t)bs + ls[2], (uint32_t)bs + ls[3]);
        *(uint4*)(a.qual_off + i0) = make_uint4((uint32_t)bq + lq[0], (uint32_t)bq + lq[1], (uint32_t)bq + lq[2], (uint32_t)bq + lq[3]);
        *(uint4*)(a.cigar_off + i0) = make_uint4((uint32_t)bc + lc[0], (uint32_t)bc + lc[1], (uint32_t)bc + lc[2], (uint32_t)bc + lc[3]);
        *(uint2*)(a.flag_out + i0) = make_uint2(out[0].flag | (out[1].flag << 16), out[2].flag | (out[3].flag << 16));
        *(uint2*)(a.cls + i0) = make_uint2(out[0].cls | (out[1].cls << 16), out[2].cls | (out[3].cls << 16));
        *(uint4*)(a.cov_out + i0) = make_uint4(out[0].ce.win, out[0].ce.off_len, out[1].ce.win, out[1].ce.off_len);
        *(uint4*)(a.cov_out + i0 + 2) = make_uint4(out[2].ce.win, out[2].ce.off_len, out[3].ce.win, out[3].ce.off_len);
    } else {
        for (uint32_t j = 0; j < nlive; ++j) {
            const uint32_t i = i0 + j;
            a.seq_off[i] = (uint32_t)bs + ls[j]; a.qual_off[i] = (uint32_t)bq + lq[j]; a.cigar_off[i] = (uint32_t)bc + lc[j];
            a.flag_out[i] = (uint16_t)out[j].flag; a.cls[i] = (uint16_t)out[j].cls; a.cov_out[i] = out[j].ce;
        }
    }
    // forward-only FASTA scan: an eligible read whose position lies before an earlier eligible read's — inside the thread
    // (local_bad), inside the workgroup (here), across workgroups (k_build_plan)
    uint32_t blk_max;
    const uint32_t before = block_scan_excl_max(tmax, sh, &blk_max);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (out[j].tgt1 && (local_bad[j] || out[j].tgt1 < before)) err_key(a.err, i0 + j, 5);
    // block partials: FASTA positions (for the scan across blocks), longest fast / generic read; class counts of the super-window
    if (tmin != 0xFFFFFFFFu) atomicMin(&red[0], tmin);
    if (maxfast) atomicMax(&red[1], maxfast);
    if (maxlong) atomicMax(&red[2], maxlong);
    if (!a.order) { // stream order: the block's reads lie in one super-window
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint32_t v = wave_sum(cnt[k]); if (lane_id() == 0 && v) atomicAdd(&red[4 + k], v); }
    }
    block_sync();
    if (threadIdx.x == 0) {
        a.blk_tgt[2ull * blockIdx.x] = blk_max;
        a.blk_tgt[2ull * blockIdx.x + 1] = red[0];
        a.blk_maxfast[2ull * blockIdx.x] = red[1];
        a.blk_maxfast[2ull * blockIdx.x + 1] = red[2];
    }
    if (!a.order && threadIdx.x < 4 && red[4 + threadIdx.x])
        atomicAdd((uint32_t*)&a.sw_counts[blockIdx.x / (BQC_SW_READS / PR_BLOCK)] + threadIdx.x, red[4 + threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------
// chunk builder
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t read_class(const PrepArgs& a, uint32_t r, uint32_t* nseg) // 0 / 1: fast read by mate slot, 2: generic path
{
    const uint32_t v = a.cls[r];
    *nseg = v & 0xFFu;
    return v >> 8;
}

// (only for batches with several read groups: in stream order k_prep_reads has counted the classes already)
__global__ __launch_bounds__(PR_THREADS) void k_build_count(PrepArgs a)
{
    __shared__ uint32_t sh[4 * (PR_THREADS / 64)];
    const SuperWindow sw = a.sws[blockIdx.x];
    uint32_t c[4] = {0, 0, 0, 0};
    for (uint32_t p = threadIdx.x; p < sw.count; p += PR_THREADS) {
        const uint32_t r = a.order ? a.order[sw.begin + p] : sw.begin + p;
        uint32_t ns;
        const uint32_t cl = read_class(a, r, &ns);
        c[cl] += 1;
        c[3] += ns;
    }
    for (int k = 0; k < 4; ++k) c[k] = wave_sum(c[k]);
    if (lane_id() == 0) for (int k = 0; k < 4; ++k) sh[4 * (threadIdx.x >> 6) + k] = c[k];
    block_sync();
    if (threadIdx.x < 4) {
        uint32_t t = 0;
        for (uint32_t w = 0; w < PR_THREADS / 64; ++w) t += sh[4 * w + threadIdx.x];
        ((uint32_t*)&a.sw_counts[blockIdx.x])[threadIdx.x] = t;
    }
}

// One workgroup.  (1) the forward-only FASTA scan across blocks, and the stream's cursor; (2) lanes per read of k_short;
// (3) where every super-window's entries go and the chunk tables; (4) the batch descriptor.
__global__ __launch_bounds__(1024) void k_build_plan(PrepArgs a)
{
    __shared__ uint32_t sh[2 * 16 + 8];
    __shared__ uint32_t s_fast_w, s_first_bad, s_gmin;
    const uint32_t nblk = (a.n + PR_BLOCK - 1) / PR_BLOCK;
    // ---- (1) + (2)
    {
        uint32_t carry = (uint32_t)(*a.cursor + 1); // position + 1 of the last eligible read of the stream so far (0: none)
        uint32_t mf = 0, gmin = 0xFFFFFFFFu;
        if (threadIdx.x == 0) { s_first_bad = 0xFFFFFFFFu; s_gmin = 0xFFFFFFFFu; }
        for (uint32_t b0 = 0; b0 < nblk; b0 += blockDim.x) {
            const uint32_t b = b0 + threadIdx.x;
            const uint32_t mx = b < nblk ? a.blk_tgt[2ull * b] : 0u, mn = b < nblk ? a.blk_tgt[2ull * b + 1] : 0xFFFFFFFFu;
            if (b < nblk) mf = max(mf, a.blk_maxfast[2ull * b]);
            uint32_t tot;
            const uint32_t before = max(carry, block_scan_excl_max(mx, sh, &tot));
            gmin = min(gmin, mn);
            if (mn < before) atomicMin(&s_first_bad, b); // some eligible read of block b lies before an earlier block's
            carry = max(carry, tot);
        }
        if (gmin != 0xFFFFFFFFu) atomicMin(&s_gmin, gmin);
        block_sync();
        const uint32_t bad = s_first_bad;
        if (bad != 0xFFFFFFFFu) { // find the first such read of that block (rare: the run ends with an error)
            uint32_t before = (uint32_t)(*a.cursor + 1);
            for (uint32_t b = threadIdx.x; b < bad; b += blockDim.x) before = max(before, a.blk_tgt[2ull * b]);
            uint32_t tot;
            (void)block_scan_excl_max(before, sh, &tot);
            for (uint32_t i = bad * PR_BLOCK + threadIdx.x; i < min(a.n, (bad + 1) * PR_BLOCK); i += blockDim.x)
                if (a.flag_out[i] & BQC_FLAG_TRIPLET) {
                    const int32_t rid = a.rid[i];
                    const uint32_t t1 = (uint32_t)(a.fasta_index ? a.fasta_index[rid] : rid) + 1u;
                    if (t1 < tot) err_key(a.err, i, 5);
                }
        }
        for (int o = 32; o > 0; o >>= 1) mf = max(mf, (uint32_t)__shfl_xor((int)mf, o));
        block_sync();
        if (lane_id() == 0) sh[threadIdx.x >> 6] = mf;
        block_sync();
        if (threadIdx.x == 0) {
            uint32_t m = 0;
            for (uint32_t w = 0; w < blockDim.x / 64; ++w) m = max(m, sh[w]);
            s_fast_w = max(1u, (m + 8 * BQC_FAST_NH - 1) / (8 * BQC_FAST_NH));
            *a.cursor = (int32_t)carry - 1;
            if (a.cursor[1] < 0 && s_gmin != 0xFFFFFFFFu) a.cursor[1] = (int32_t)s_gmin - 1; // first eligible read of the stream (shards check their order with it)
        }
        block_sync();
    }
    const uint32_t fast_w = s_fast_w, rpw = 64u / fast_w, h0 = (rpw + 1) / 2, h1 = rpw / 2;
    const uint32_t groups_cap = BQC_FAST_WAVES * (64u / rpw);  // groups per chunk: one tile of whole groups per wave of k_short
    const uint32_t seg_cap = groups_cap * rpw;                 // segment entries per chunk
    // ---- (3) prefix sums over the super-windows of {groups, segments, generic reads}
    uint32_t run_g = 0, run_q = 0, run_s = 0;
    for (uint32_t s0 = 0; s0 < a.n_sw; s0 += blockDim.x) {
        const uint32_t s = s0 + threadIdx.x;
        uint32_t g = 0, q = 0, sl = 0;
        if (s < a.n_sw) {
            const SwCounts c = a.sw_counts[s];
            g = max((c.n0 + h0 - 1) / h0, (c.n1 + h1 - 1) / h1);
            q = c.n_seg; sl = c.n_slow;
        }
        uint32_t tg, tq, ts;
        const uint32_t eg = run_g + block_scan_excl(g, sh, &tg), eq = run_q + block_scan_excl(q, sh, &tq), es = run_s + block_scan_excl(sl, sh, &ts);
        if (s < a.n_sw) a.sw_plan[s] = SwPlan{eg, g, eq, es}; // (relative to the batch; made absolute below)
        run_g += tg; run_q += tq; run_s += ts;
    }
    __threadfence_block();
    block_sync();
    // per stretch (one read group each, in order): [read groups | segment entries, padded to whole groups | generic reads]
    // -> absolute perm positions and chunk tables.  Thread 0 walks the stretches (at most one per read group); all threads fill.
    __shared__ uint32_t s_pb, s_cf, s_cs, s_g0, s_q0, s_s0, s_ng, s_nq, s_ns, s_lane, s_sw0, s_sw1, s_longest;
    if (threadIdx.x == 0) { s_pb = 0; s_cf = 0; s_cs = 0; }
    block_sync();
    for (uint32_t st = 0; st < a.n_stretch; ++st) {
        if (threadIdx.x == 0) {
            const Stretch S = a.stretches[st];
            const SwPlan p0 = a.sw_plan[S.sw_begin];
            const SwPlan pl = a.sw_plan[S.sw_end - 1];
            const SwCounts cl = a.sw_counts[S.sw_end - 1];
            s_g0 = p0.group_base; s_q0 = p0.seg_base; s_s0 = p0.slow_base;
            s_ng = pl.group_base + pl.n_groups - p0.group_base;
            s_nq = pl.seg_base + cl.n_seg - p0.seg_base;
            s_ns = pl.slow_base + cl.n_slow - p0.slow_base;
            s_lane = S.lane; s_sw0 = S.sw_begin; s_sw1 = S.sw_end;
        }
        block_sync();
        const uint32_t pb = s_pb, ng = s_ng, nq = s_nq, ns = s_ns, lane = s_lane;
        const uint32_t nq_pad = (nq + rpw - 1) / rpw * rpw;
        const uint32_t seg0 = pb + ng * rpw, slow0 = seg0 + nq_pad;
        for (uint32_t s = s_sw0 + threadIdx.x; s < s_sw1; s += blockDim.x) {
            SwPlan p = a.sw_plan[s];
            p.group_base = pb + (p.group_base - s_g0) * rpw;
            p.seg_base = seg0 + (p.seg_base - s_q0);
            p.slow_base = slow0 + (p.slow_base - s_s0);
            a.sw_plan[s] = p;
        }
        for (uint32_t k = nq + threadIdx.x; k < nq_pad; k += blockDim.x) a.perm[seg0 + k] = 0xFFFFFFFFu;
        const uint32_t n_rc = (ng + groups_cap - 1) / groups_cap, n_sc = (nq_pad + seg_cap - 1) / seg_cap, n_lc = (ns + BQC_CHUNK_READS - 1) / BQC_CHUNK_READS;
        const uint32_t cf = s_cf, cs = s_cs;
        for (uint32_t k = threadIdx.x; k < n_rc; k += blockDim.x) {
            const uint32_t g = min(groups_cap, ng - k * groups_cap);
            if (cf + k < a.chunks_fast_cap) a.chunks_fast[cf + k] = Chunk{pb + k * groups_cap * rpw, g * rpw, lane, g * rpw, 0, 0, 0, 0};
        }
        for (uint32_t k = threadIdx.x; k < n_sc; k += blockDim.x) {
            const uint32_t e = min(seg_cap, nq_pad - k * seg_cap);
            if (cf + n_rc + k < a.chunks_fast_cap) a.chunks_fast[cf + n_rc + k] = Chunk{seg0 + k * seg_cap, e, lane, 0, 0, 0, 0, 0};
        }
        for (uint32_t k = threadIdx.x; k < n_lc; k += blockDim.x) {
            const uint32_t e = min((uint32_t)BQC_CHUNK_READS, ns - k * BQC_CHUNK_READS);
            if (cs + k < a.chunks_slow_cap) a.chunks_slow[cs + k] = Chunk{slow0 + k * BQC_CHUNK_READS, e, lane, 0, 0, 0, 0, 0};
        }
        block_sync();
        if (threadIdx.x == 0) { s_pb = slow0 + ns; s_cf = cf + n_rc + n_sc; s_cs = cs + n_lc; }
        block_sync();
    }
    // ---- (4) descriptor; the longest generic read (the cycle tiles k_long needs)
    uint32_t longest = 0;
    for (uint32_t b = threadIdx.x; b < nblk; b += blockDim.x) longest = max(longest, a.blk_maxfast[2ull * b + 1]);
    for (int o = 32; o > 0; o >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, o));
    if (threadIdx.x == 0) s_longest = 0;
    block_sync();
    if (lane_id() == 0) atomicMax(&s_longest, longest);
    block_sync();
    if (threadIdx.x == 0) {
        if (s_pb > a.perm_cap || s_cf > a.chunks_fast_cap || s_cs > a.chunks_slow_cap) atomicOr(&a.err->flags, BQC_DEVERR_INTERNAL);
        // A read that ends the reference's run (first_key) poisons the context on the host; until the host gets to know, the
        // hot kernels must not touch such a batch (a lane out of range would index outside the state vector).
        const bool fatal = a.err->first_key != BQC_ERRKEY_NONE || (a.err->flags & BQC_DEVERR_INTERNAL);
        a.desc->fatal = fatal ? 1u : 0u;
        a.desc->n_chunks_fast = fatal ? 0u : min(s_cf, a.chunks_fast_cap);
        a.desc->n_chunks_slow = fatal ? 0u : min(s_cs, a.chunks_slow_cap);
        a.desc->fast_w = fast_w;
        a.desc->n_perm = s_pb;
        a.desc->long_max_len = s_longest;
        if (a.desc->n_cov_extra > a.cov_extra_cap) a.desc->n_cov_extra = a.cov_extra_cap;
        // values for the message of the first failing read
        const unsigned long long key = a.err->first_key;
        if (key != BQC_ERRKEY_NONE) {
            const uint32_t i = (uint32_t)(key >> 3), order = (uint32_t)key & 7u;
            if (i < a.n) { a.err->aux0 = order == 1 ? a.l_seq[i] : order == 2 ? a.lane[i] : (uint32_t)a.rid[i]; a.err->aux1 = a.lane[i]; }
        }
    }
}

// entries of one super-window: read groups (first-mate slots | other slots, missing ones null), segment entries, generic reads
__global__ __launch_bounds__(PR_THREADS) void k_build_scatter(PrepArgs a)
{
    __shared__ uint32_t ent[3 * BQC_SW_READS + 128]; // groups * rpw <= (count / h + 1) * rpw with rpw / h <= 2.5 (rpw = 5: 3 + 2 slots)
    __shared__ uint32_t sh[2 * (PR_THREADS / 64)];
    const SuperWindow sw = a.sws[blockIdx.x];
    const SwPlan pl = a.sw_plan[blockIdx.x];
    const uint32_t fast_w = a.desc->fast_w, rpw = 64u / fast_w, h0 = (rpw + 1) / 2, h1 = rpw / 2;
    const uint32_t n_ent = pl.n_groups * rpw;
    if (n_ent > 3 * BQC_SW_READS + 128) { if (threadIdx.x == 0) atomicOr(&a.err->flags, BQC_DEVERR_INTERNAL); return; }
    for (uint32_t k = threadIdx.x; k < n_ent; k += PR_THREADS) ent[k] = 0xFFFFFFFFu;
    // thread t owns the positions [t * per, (t + 1) * per) of the super-window: ranks inside its run, then across threads
    const uint32_t per = (sw.count + PR_THREADS - 1) / PR_THREADS;
    const uint32_t lo = min(sw.count, threadIdx.x * per), hi = min(sw.count, lo + per);
    uint32_t c[4] = {0, 0, 0, 0};
    for (uint32_t p = lo; p < hi; ++p) {
        const uint32_t r = a.order ? a.order[sw.begin + p] : sw.begin + p;
        uint32_t ns;
        const uint32_t cl = read_class(a, r, &ns);
        c[cl] += 1; c[3] += ns;
    }
    uint32_t base[4], tot;
    for (int k = 0; k < 4; ++k) base[k] = block_scan_excl(c[k], sh, &tot);
    block_sync();
    for (uint32_t p = lo; p < hi; ++p) {
        const uint32_t r = a.order ? a.order[sw.begin + p] : sw.begin + p;
        uint32_t ns;
        const uint32_t cl = read_class(a, r, &ns);
        const uint32_t k = base[cl]++;
        if (cl == 0u) ent[(k / h0) * rpw + k % h0] = r;
        else if (cl == 1u) ent[(k / h1) * rpw + h0 + k % h1] = r;
        else a.perm[pl.slow_base + k] = r;
        if (ns) {
            const uint32_t co = a.cigar_off[r];
            for (uint32_t j = 0; j < ns; ++j) a.perm[pl.seg_base + base[3] + j] = BQC_ENTRY_SEG | (co + j);
            base[3] += ns;
        }
    }
    block_sync();
    for (uint32_t k = threadIdx.x; k < n_ent; k += PR_THREADS) a.perm[pl.group_base + k] = ent[k];
}

// the unordered flags of the hot kernels and the keyed first error, for the host (one small copy per batch)
extern "C" void bqc_launch_prep(const PrepArgs& a, const DevRefs& refs, hipStream_t s)
{
    if (a.n == 0) return;
    const uint32_t nblk = (a.n + PR_BLOCK - 1) / PR_BLOCK;
    hipLaunchKernelGGL(k_prep_sizes, dim3(nblk), dim3(PR_THREADS), 0, s, a);
    hipLaunchKernelGGL(k_prep_scan, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(k_prep_reads, dim3(nblk), dim3(PR_THREADS), 0, s, a, refs);
    if (a.order) hipLaunchKernelGGL(k_build_count, dim3(a.n_sw), dim3(PR_THREADS), 0, s, a);
    hipLaunchKernelGGL(k_build_plan, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(k_build_scatter, dim3(a.n_sw), dim3(PR_THREADS), 0, s, a);
}

// ---------------------------------------------------------------------------------------------------
// shard mode: the set-aside reads of a batch, once their windows are known
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pend_interval(uint32_t pos, uint32_t c0, uint32_t len) // off | len << 16 inside the read's two live windows, 0: nothing
{
    const uint32_t lo = pos + c0;
    if (!len || lo >= 2u * BQC_VSIZE) return 0u;
    const uint32_t hi = min(lo + len, 2u * BQC_VSIZE);
    return lo | ((hi - lo) << 16);
}
__global__ __launch_bounds__(256) void k_pend_cov(uint32_t n, const CovEntry* __restrict__ cov_in, const PendRun* __restrict__ pend, const PendExtra* __restrict__ extra,
                                                     const uint32_t* __restrict__ extra_n, uint32_t extra_cap, const uint8_t* __restrict__ lane,
                                                     CovEntry* __restrict__ cov_out, CovExtra* __restrict__ cov_extra, BatchDesc* __restrict__ desc)
{
    const uint32_t ne = min(*extra_n, extra_cap);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const CovEntry ce = cov_in[i];
        const PendRun r = pend[i];
        cov_out[i] = CovEntry{ce.win, pend_interval(ce.off_len, r.c0, r.len)};
    }
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += gridDim.x * blockDim.x) {
        const PendExtra x = extra[e];
        const CovEntry ce = cov_in[x.idx];
        cov_extra[e] = CovExtra{ce.win, pend_interval(ce.off_len, x.c0, x.len), lane[x.idx], 0};
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { desc->n_cov_extra = ne; desc->fatal = 0; }
}
extern "C" void bqc_launch_pend_cov(uint32_t n, const CovEntry* cov_in, const PendRun* pend, const PendExtra* extra, const uint32_t* extra_n, uint32_t extra_cap,
                                    const uint8_t* lane, CovEntry* cov_out, CovExtra* cov_extra, BatchDesc* desc, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_pend_cov, dim3(std::min<uint32_t>((n + 255) / 256, 4096u)), dim3(256), 0, s, n, cov_in, pend, extra, extra_n, extra_cap, lane, cov_out, cov_extra, desc);
}
