// k_anchor.hip — the order-dependent part of OverallNumbers::coverage (OverallNumbers.hpp:84-110) on the card, for ONE read group.
//
// The reference keeps, per read group, two live windows of 1000 positions: `shift` (where the first one starts), the chromosome
// `id`, and — in this library's virtual coordinates — `win`, the number of windows flushed so far.  A read that enters coverage()
// at beginPos b (bamqualcheck.cpp:318-327,392,430-433 decide which do) moves the state:
//     first read:                         id = rid, shift = b
//     id != rid or b - shift > 2000:      reset   -> id = rid, shift = b, win += 2          (unsigned arithmetic: b < shift resets too)
//     1000 < b - shift < 2000:            slide   -> shift += 1000, win += 1
// and gets {win, b - shift}.  A recurrence over 600 M reads — but almost all of it is arithmetic: behind any read, b - shift lies in
// [0, 1000] (or is 2000 exactly: neither slide nor reset, "stuck"); so a read that is less than 1000 positions behind the read before
// it (same chromosome) can only stay or slide ONCE, and a whole RUN of such reads follows from the state at the run's first read in
// closed form: with x = b - shift_entry, b - shift = x if x <= 1000, else ((x - 1) mod 1000) + 1, and the slides are what was taken
// off, in thousands.  What really is sequential are the BREAKS — a read 1000 or more behind its predecessor, out of order, or on
// another chromosome: a handful per million reads of a 30x genome (chromosome ends, gaps in the assembly), every other read in sparse
// data.  So:
//   k_an_count / k_an_scan / k_an_scatter   the reads that enter coverage(), compacted in stream order (position, chromosome, index);
//                                           the batch's other per-read facts the host used to gather (generic-path reads, records
//                                           without qualities, range of chromosomes) on the way
//   k_an_bcount / k_an_scan / k_an_bscatter the breaks among them, listed; every candidate learns the number of its run
//   k_an_chain                              ONE thread walks the breaks (their operands loaded into LDS by the workgroup, 256 at a
//                                           time): state at the end of the run before, transition of the recurrence, state the new run
//                                           starts with.  More than AN_MAX_BREAKS breaks: the batch is left to the host's recurrence
//                                           (flag in the summary; the state is not touched)
//   k_an_apply                              every candidate: closed form from its run's entry -> {window, offset}; the candidates at
//                                           which the window changes go to the boundary list the host builds the coverage tiles from
// Checked against the host's recurrence (bqc_pipeline.cpp: CovPlanner) read by read on sorted, sparse, unsorted and wild inputs
// (tests/test_gpu_anchor.py), and through every test that runs the program with the reader on the card.
#include "kernels_common.h"
#include "anchor.h"

namespace {
__device__ __forceinline__ bool an_candidate(const AnchorArgs& a, uint32_t i)
{
    const uint32_t flag = a.flag[i];
    const int32_t rid = a.rid[i];
    // primary record with a first / last flag, on a main chromosome, mapped, not a duplicate (bamqualcheck.cpp:318-327,392,430-433)
    return !((flag & 0xD04u) || !(flag & 0xC0u) || (uint32_t)rid >= a.n_refs || !a.main_chrom[rid] || a.lane[i] >= a.n_lanes);
}

// exclusive prefix of `v` over the workgroup's 256 threads, and the workgroup's total
__device__ __forceinline__ uint32_t block_excl(uint32_t v, uint32_t* wsum /* [4] */, uint32_t& total)
{
    const uint32_t inc = wave_scan_incl(v);
    block_sync();
    if (lane_id() == WAVE - 1) wsum[threadIdx.x >> 6] = inc;
    block_sync();
    uint32_t off = inc - v;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) off += wsum[w];
    total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return off;
}

// a read's state inside its run: x = beginPos - shift at the run's entry (for a stuck run: - beginPos of the read that reset)
__device__ __forceinline__ void an_closed(uint32_t x, uint32_t& delta, uint32_t& slides)
{
    if (x <= BQC_VSIZE) { delta = x; slides = 0; return; }
    delta = (x - 1u) % BQC_VSIZE + 1u;
    slides = (x - delta) / BQC_VSIZE;
}
__device__ __forceinline__ void an_in_run(const AnchorRun& r, uint32_t b, uint32_t& rel, uint32_t& delta)
{
    uint32_t slides;
    if (!r.stuck) { an_closed(b - r.s_e, delta, slides); rel = r.rel_e + slides; return; }
    if (b == r.b_e) { delta = 2u * BQC_VSIZE; rel = r.rel_e; return; }
    an_closed(b - r.b_star, delta, slides); // the first read further right reset the windows
    rel = r.rel_e + 2u + slides;
}
} // namespace

// ---- candidates ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_an_count(AnchorArgs a)
{
    __shared__ uint32_t wsum[4];
    const uint32_t i0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    uint32_t c = 0, n_slow = 0, max_slow = 0, n_noqual = 0, s1 = 0, s2 = 0, s3 = 0;
    int32_t rmin = INT32_MAX, rmax = -1;
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t i = i0 + k;
        if (i >= a.n) break;
        c += an_candidate(a, i) ? 1u : 0u;
        const uint32_t L = a.l_seq[i], f = a.flag[i];
        s1 += (L + 1u) / 2u; s2 += L; s3 += a.n_cigar[i]; // (four reads per thread: far below 2^32; summed in 64 bits from the wave on)
        if (a.no_fast || L > BQC_FAST_MAXLEN) { ++n_slow; max_slow = max(max_slow, L); }
        n_noqual += ((f & BQC_FLAG_NO_QUAL) && !(f & 0x900u) && (f & 0xC0u)) ? 1u : 0u;
        const int32_t rid = a.rid[i];
        if ((uint32_t)rid < a.n_refs) { rmin = min(rmin, rid); rmax = max(rmax, rid); }
    }
    uint32_t total;
    (void)block_excl(c, wsum, total);
    if (threadIdx.x == 0) a.blk_a[blockIdx.x] = total;
    // the batch's other facts: the workgroup's partial results (thousands of waves adding to the same few words took longer than the
    // rest of the kernel; k_an_scan sums the workgroups')
    __shared__ unsigned long long part[4][8];
    n_slow = wave_sum(n_slow); n_noqual = wave_sum(n_noqual);
    unsigned long long t1 = s1, t2 = s2, t3 = s3;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o); t2 += __shfl_xor(t2, o); t3 += __shfl_xor(t3, o); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        max_slow = max(max_slow, (uint32_t)__shfl_xor((int)max_slow, o));
        rmin = min(rmin, __shfl_xor(rmin, o));
        rmax = max(rmax, __shfl_xor(rmax, o));
    }
    if (lane_id() == 0) {
        unsigned long long* q = part[threadIdx.x >> 6];
        q[0] = n_slow; q[1] = max_slow; q[2] = n_noqual; q[3] = (unsigned long long)(long long)rmin; q[4] = (unsigned long long)(long long)rmax; q[5] = t1; q[6] = t2; q[7] = t3;
    }
    block_sync();
    if (threadIdx.x == 0) {
        AnchorPart P;
        P.n_slow = (uint32_t)(part[0][0] + part[1][0] + part[2][0] + part[3][0]);
        P.max_slow = (uint32_t)max(max(part[0][1], part[1][1]), max(part[2][1], part[3][1]));
        P.n_noqual = (uint32_t)(part[0][2] + part[1][2] + part[2][2] + part[3][2]);
        P.rid_min = min(min((int32_t)(long long)part[0][3], (int32_t)(long long)part[1][3]), min((int32_t)(long long)part[2][3], (int32_t)(long long)part[3][3]));
        P.rid_max = max(max((int32_t)(long long)part[0][4], (int32_t)(long long)part[1][4]), max((int32_t)(long long)part[2][4], (int32_t)(long long)part[3][4]));
        P.s1 = part[0][5] + part[1][5] + part[2][5] + part[3][5];
        P.s2 = part[0][6] + part[1][6] + part[2][6] + part[3][6];
        P.s3 = part[0][7] + part[1][7] + part[2][7] + part[3][7];
        P.first_certain = 0xFFFFFFFFu;
        a.parts[blockIdx.x] = P;
    }
}

// exclusive scan of up to 4096 block counts by one workgroup of 1024 threads; the total goes to *total_out
// (which: 0 = the candidates' counts — it also starts the batch's summary and sums the workgroups' partial facts into it; 1 = the
// breaks' counts — and the first certain reset, when a shard is still setting reads aside)
__global__ __launch_bounds__(1024) void k_an_scan(uint32_t* __restrict__ blk, uint32_t nblk, AnchorSummary* __restrict__ sum, AnchorPart* __restrict__ parts, int which)
{
    __shared__ uint32_t wsum[16];
    __shared__ unsigned long long red[16][8];
    if (which == 0) {
        unsigned long long v[8] = {0, 0, 0, (unsigned long long)(long long)INT32_MAX, (unsigned long long)(long long)-1, 0, 0, 0};
        for (uint32_t i = threadIdx.x; i < nblk; i += 1024u) {
            const AnchorPart P = parts[i];
            v[0] += P.n_slow; v[1] = max(v[1], (unsigned long long)P.max_slow); v[2] += P.n_noqual;
            v[3] = (unsigned long long)(long long)min((int32_t)(long long)v[3], P.rid_min); v[4] = (unsigned long long)(long long)max((int32_t)(long long)v[4], P.rid_max);
            v[5] += P.s1; v[6] += P.s2; v[7] += P.s3;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            v[0] += __shfl_xor(v[0], o); v[2] += __shfl_xor(v[2], o); v[5] += __shfl_xor(v[5], o); v[6] += __shfl_xor(v[6], o); v[7] += __shfl_xor(v[7], o);
            v[1] = max(v[1], (unsigned long long)__shfl_xor(v[1], o));
            v[3] = (unsigned long long)(long long)min((int32_t)(long long)v[3], (int32_t)(long long)__shfl_xor(v[3], o));
            v[4] = (unsigned long long)(long long)max((int32_t)(long long)v[4], (int32_t)(long long)__shfl_xor(v[4], o));
        }
        if (lane_id() == 0) for (int k = 0; k < 8; ++k) red[threadIdx.x >> 6][k] = v[k];
        block_sync();
        if (threadIdx.x == 0) {
            AnchorSummary z{};
            z.rid_min = INT32_MAX; z.rid_max = -1; z.first_certain = 0xFFFFFFFFu;
            for (int w = 0; w < 16; ++w) {
                z.n_slow += (uint32_t)red[w][0]; z.max_len_slow = max(z.max_len_slow, (uint32_t)red[w][1]); z.n_noqual += (uint32_t)red[w][2];
                z.rid_min = min(z.rid_min, (int32_t)(long long)red[w][3]); z.rid_max = max(z.rid_max, (int32_t)(long long)red[w][4]);
                z.seq_bytes += red[w][5]; z.qual_bytes += red[w][6]; z.cigar_words += red[w][7];
            }
            *sum = z;
        }
        block_sync();
    } else {
        uint32_t fc = 0xFFFFFFFFu;
        for (uint32_t i = threadIdx.x; i < nblk; i += 1024u) fc = min(fc, parts[i].first_certain);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) fc = min(fc, (uint32_t)__shfl_xor((int)fc, o));
        if (lane_id() == 0 && fc != 0xFFFFFFFFu) atomicMin(&sum->first_certain, fc); // (sixteen waves)
    }
    uint32_t* const total_out = which == 0 ? &sum->n_cand : &sum->n_breaks;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblk; base += 1024u) { // (one round for batches of up to 4 M reads)
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nblk ? blk[i] : 0u;
        const uint32_t inc = wave_scan_incl(v);
        block_sync();
        if (lane_id() == WAVE - 1) wsum[threadIdx.x >> 6] = inc;
        block_sync();
        uint32_t off = inc - v, tot = 0;
        for (uint32_t w = 0; w < 16u; ++w) { if (w < (threadIdx.x >> 6)) off += wsum[w]; tot += wsum[w]; }
        if (i < nblk) blk[i] = carry + off;
        carry += tot;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(256) void k_an_scatter(AnchorArgs a)
{
    __shared__ uint32_t wsum[4];
    const uint32_t i0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    bool cand[4];
    uint32_t c = 0;
    for (uint32_t k = 0; k < 4u; ++k) { cand[k] = i0 + k < a.n && an_candidate(a, i0 + k); c += cand[k] ? 1u : 0u; }
    uint32_t total;
    uint32_t j = a.blk_a[blockIdx.x] + block_excl(c, wsum, total);
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t i = i0 + k;
        if (i >= a.n) break;
        if (cand[k]) { a.cpos[j] = (uint32_t)a.pos[i]; a.crid[j] = a.rid[i]; a.cidx[j] = i; ++j; }
        else a.cov_out[i] = CovEntry{BQC_COV_NONE, 0u};
    }
}

// ---- breaks ------------------------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ bool an_break(const AnchorArgs& a, uint32_t j)
{
    if (j == 0) return true; // the batch's first candidate: its state comes from the batch before
    return a.crid[j] != a.crid[j - 1] || a.cpos[j] - a.cpos[j - 1] >= BQC_VSIZE; // (unsigned: a read in front of its predecessor is a break)
}
}
__global__ __launch_bounds__(256) void k_an_bcount(AnchorArgs a)
{
    __shared__ uint32_t wsum[4];
    const uint32_t nc = a.sum->n_cand, j0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    uint32_t c = 0;
    for (uint32_t k = 0; k < 4u; ++k) c += (j0 + k < nc && an_break(a, j0 + k)) ? 1u : 0u;
    uint32_t total;
    (void)block_excl(c, wsum, total);
    if (threadIdx.x == 0) a.blk_b[blockIdx.x] = total;
    if (a.state->pending) { // a shard in the middle of the stream: the first read that resets the windows whatever their state
        uint32_t fc = 0xFFFFFFFFu;
        for (uint32_t k = 0; k < 4u && fc == 0xFFFFFFFFu; ++k) {
            const uint32_t j = j0 + k;
            if (j >= nc) break;
            const bool has_prev = j ? true : a.state->has_prev != 0;
            const int32_t prid = j ? a.crid[j - 1] : a.state->prev_rid;
            const uint32_t d = a.cpos[j] - (j ? a.cpos[j - 1] : a.state->prev_bp);
            if (has_prev && (a.crid[j] != prid || (d > 2u * BQC_VSIZE && d <= 0xFFFFFFFFu - 2u * BQC_VSIZE))) fc = j;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) fc = min(fc, (uint32_t)__shfl_xor((int)fc, o));
        if (lane_id() == 0 && fc != 0xFFFFFFFFu) atomicMin(&a.parts[blockIdx.x].first_certain, fc); // (the workgroup's four waves; k_an_scan takes the minimum)
    }
}
__global__ __launch_bounds__(256) void k_an_bscatter(AnchorArgs a)
{
    __shared__ uint32_t wsum[4];
    const uint32_t nc = a.sum->n_cand, j0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    bool br[4];
    uint32_t c = 0;
    for (uint32_t k = 0; k < 4u; ++k) { br[k] = j0 + k < nc && an_break(a, j0 + k); c += br[k] ? 1u : 0u; }
    uint32_t total;
    uint32_t r = a.blk_b[blockIdx.x] + block_excl(c, wsum, total); // breaks in front of this thread's first candidate
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t j = j0 + k;
        if (j >= nc) break;
        if (br[k]) { if (r < AN_MAX_BREAKS) a.bj[r] = j; ++r; }
        a.crun[j] = r - 1u; // (candidate 0 is a break: r >= 1)
    }
}

// ---- the chain over the breaks -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_an_chain(AnchorArgs a)
{
    __shared__ uint32_t s_j[256], s_b[256], s_bl[256], s_jn[256];
    __shared__ int32_t s_rid[256];
    const uint32_t nc = a.sum->n_cand, nb = a.sum->n_breaks;
    AnchorState st = *a.state;
    if (threadIdx.x == 0) { a.sum->before = st; a.sum->after = st; a.sum->last_rel = 0; a.sum->n_pending = 0; }
    if (nb > AN_MAX_BREAKS) { if (threadIdx.x == 0) atomicOr(&a.sum->flags, AN_FLAG_TOO_MANY_BREAKS); return; }
    if (nc == 0) return;
    // setting aside: the candidates in front of the first certain reset are pending; the chain starts AT that read — a break — from the
    // context's own state (a stream that begins there)
    uint32_t k0 = 0, j_first = 0;
    if (st.pending) {
        const uint32_t fc = a.sum->first_certain;
        j_first = fc == 0xFFFFFFFFu ? nc : fc;
        if (threadIdx.x == 0) a.sum->n_pending = j_first;
        if (j_first == nc) { // every candidate of the batch is set aside
            if (threadIdx.x == 0) {
                AnchorState out = st;
                out.has_prev = 1; out.prev_rid = a.crid[nc - 1]; out.prev_bp = a.cpos[nc - 1];
                *a.state = out;
                a.sum->after = out;
            }
            return;
        }
        k0 = a.crun[j_first];
    }
    // the state between two reads: first / id / absolute shift / windows flushed in this batch, and the run it belongs to
    uint32_t first = st.first, s = (uint32_t)st.shift, rel = 0;
    int32_t id = st.id;
    AnchorRun run{};
    for (uint32_t base = k0; base < nb; base += 256u) {
        block_sync();
        {
            const uint32_t k = base + threadIdx.x;
            if (k < nb) {
                const uint32_t j = a.bj[k];
                s_j[threadIdx.x] = j; s_b[threadIdx.x] = a.cpos[j]; s_rid[threadIdx.x] = a.crid[j];
                s_bl[threadIdx.x] = j ? a.cpos[j - 1] : 0u;            // the last read of the run before
                s_jn[threadIdx.x] = k + 1 < nb ? a.bj[k + 1] : nc;      // where this run ends
            }
        }
        block_sync();
        if (threadIdx.x == 0) {
            const uint32_t m = min(256u, nb - base);
            for (uint32_t t = 0; t < m; ++t) {
                const uint32_t j = s_j[t], b = s_b[t];
                const int32_t rid = s_rid[t];
                if (j != j_first) { // where the run before has got to at its last read (the first read of all: the state that came in)
                    uint32_t r2, d2;
                    an_in_run(run, s_bl[t], r2, d2);
                    rel = r2; s = s_bl[t] - d2;
                }
                // the recurrence itself (OverallNumbers.hpp:84-110)
                if (first) { first = 0; id = rid; s = b; }
                if (id != rid || b - s > 2u * BQC_VSIZE) { id = rid; rel += 2u; s = b; }
                uint32_t p = b - s;
                if (p > BQC_VSIZE && p < 2u * BQC_VSIZE) { rel += 1u; s += BQC_VSIZE; p -= BQC_VSIZE; }
                run.b_e = b; run.s_e = s; run.rel_e = rel; run.stuck = p == 2u * BQC_VSIZE ? 1u : 0u; run.b_star = b;
                if (run.stuck) // the first read of the run further right (reads at the same position come first: the run is sorted)
                    for (uint32_t q = j + 1; q < s_jn[t]; ++q) { const uint32_t bq = a.cpos[q]; if (bq != b) { run.b_star = bq; break; } }
                a.runs[base + t] = run;
            }
        }
    }
    if (threadIdx.x == 0) { // behind the batch's last candidate
        uint32_t r2, d2;
        const uint32_t bl = a.cpos[nc - 1];
        an_in_run(run, bl, r2, d2);
        AnchorState out = st;
        out.first = 0; out.id = id; out.shift = (int32_t)(bl - d2); out.pad = 0; out.win = st.win + r2;
        if (st.pending) { out.pending = 0; out.has_prev = 1; out.prev_rid = a.crid[j_first]; out.prev_bp = a.cpos[j_first]; }
        *a.state = out;
        a.sum->after = out;
        a.sum->last_rel = r2;
    }
}

// ---- every candidate's anchor ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_an_apply(AnchorArgs a)
{
    if (a.sum->flags & AN_FLAG_TOO_MANY_BREAKS) return;
    const uint32_t nc = a.sum->n_cand, j = blockIdx.x * 256u + threadIdx.x;
    if (j >= nc) return;
    const uint32_t i = a.cidx[j], np = a.sum->n_pending;
    if (j < np) { a.cov_out[i] = CovEntry{BQC_COV_PENDING, j}; return; } // set aside: its place in the batch's pending log
    uint32_t rel, delta;
    an_in_run(a.runs[a.crun[j]], a.cpos[j], rel, delta);
    a.cov_out[i] = CovEntry{rel, delta};
    bool boundary = j == np;
    if (j != np) {
        uint32_t relp, dp;
        an_in_run(a.runs[a.crun[j - 1]], a.cpos[j - 1], relp, dp);
        boundary = relp != rel;
    }
    if (boundary) {
        if (rel < a.first_cap) a.first_of[rel] = i;
        else atomicOr(&a.sum->flags, AN_FLAG_BOUND_OVERFLOW);
    }
}

extern "C" void bqc_launch_anchor(const AnchorArgs& a, hipStream_t s)
{
    const uint32_t nblk = (a.n + 1023u) / 1024u; // (also the grid of the candidates' passes: n_cand <= n is only known on the card)
    if (a.n) hipLaunchKernelGGL(k_an_count, dim3(nblk), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_an_scan, dim3(1), dim3(1024), 0, s, a.blk_a, nblk, a.sum, a.parts, 0); // (starts the summary)
    if (!a.n) { hipLaunchKernelGGL(k_an_chain, dim3(1), dim3(256), 0, s, a); return; }
    hipLaunchKernelGGL(k_an_scatter, dim3(nblk), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_an_bcount, dim3(nblk), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_an_scan, dim3(1), dim3(1024), 0, s, a.blk_b, nblk, a.sum, a.parts, 1);
    hipLaunchKernelGGL(k_an_bscatter, dim3(nblk), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_an_chain, dim3(1), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_an_apply, dim3((a.n + 255u) / 256u), dim3(256), 0, s, a);
}
