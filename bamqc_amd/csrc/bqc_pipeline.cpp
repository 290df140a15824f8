// bqc_pipeline.cpp — how a batch of decoded records gets to the kernels (include/bamqc.h: bqc_submit, bqc_submit_async,
// bqc_upload / bqc_process).  Replaces the body of the reference's record loop (src/bamqualcheck.cpp:303-444) together with
// the kernels; the only per-read work left on the host is the O(1) coverage anchor recurrence
// (src/OverallNumbers.hpp:84-110) — everything else a read needs before its counters (payload offsets, flag annotation,
// checkFlagsAndQuality, covered intervals, triplet segments, chunk tables) is computed on the device (k_prep.hip).
//
// bqc_submit is a pipeline of three page-locked slots: the host pass of batch i + 1 runs while batch i is copied to the device
// on the copy stream and batch i - 1 is processed on the compute stream; the call returns when the batch is queued, and what
// the device finds wrong with a batch (first failing read in stream order, as the reference would have met it) surfaces at
// a later call.  There is NO CPU fallback: every path below ends in HIP launches.
#include "bqc_ctx.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdlib>
#include <cstring>

#include "../host/parallel.h"

extern "C" {
void bqc_launch_reads_chunks(const DevBatch&, const StateLayout&, uint64_t*, const DevRefs&, uint32_t*, uint32_t grid, uint32_t fast_table, hipStream_t);
void bqc_launch_nm_extra(const DevBatch&, const StateLayout&, uint64_t*, const DevRefs&, uint32_t*, hipStream_t);
void bqc_launch_long(const DevBatch&, const StateLayout&, uint64_t*, const DevRefs&, uint32_t*, uint32_t*, uint32_t, uint32_t, uint32_t, uint32_t* t8rows, uint32_t* t8_used, uint32_t t8_lane, uint32_t* cyc_tiles, uint32_t* cyc_used, hipStream_t);
uint32_t bqc_long_slots(uint32_t max_len_ub, uint32_t n_chunks_ub, uint32_t n_cu);
void bqc_launch_cov(const DevBatch&, const StateLayout&, uint64_t*, uint32_t* carry, uint32_t* parity, const uint8_t* lane_mask, uint8_t* started,
                    const uint8_t* started_after, uint32_t n_lanes, hipStream_t);
void bqc_launch_add_words(uint64_t* state, const uint64_t* idx, const uint64_t* val, uint32_t n, hipStream_t);
void bqc_launch_or_bytes(uint8_t* dst, const uint8_t* src, uint32_t n, hipStream_t);
void bqc_launch_short(const DevBatch&, const StateLayout&, uint64_t*, const DevRefs&, uint32_t*, uint32_t grid, uint32_t* t8rows, uint32_t* t8_used, hipStream_t);
uint32_t bqc_short_parts();
void bqc_launch_t8_fold(const uint32_t* t8rows, const uint32_t* t8_used, uint32_t n_slots, const StateLayout&, uint64_t* state, uint32_t lane, hipStream_t);
void bqc_launch_err_merge(ErrRec* dst, const ErrRec* src, hipStream_t);
void bqc_launch_cov_final(const StateLayout&, uint64_t*, const uint32_t* carry, const uint32_t* parity, const uint8_t* started, const uint8_t* sel, uint32_t count_start, hipStream_t);
}

using clk = std::chrono::steady_clock;
static double secs_since(clk::time_point a) { return std::chrono::duration<double>(clk::now() - a).count(); }
static int timing_level()
{
    static const int lv = [] { const char* e = getenv("BQC_TIMING"); return e ? atoi(e) : 0; }();
    return lv;
}

int bqc_copy_stream(bqc_ctx* c)
{
    if (c->copy_stream) return 0;
    c->copy_stream = bqc_pool_stream(c->device, 3);
    return c->copy_stream ? 0 : bqc_fail(c, BQC_ERR_DEVICE, "hipStreamCreate failed");
}

void bqc_state_ready(bqc_ctx* c)
{
    if (!c->t8_slots_used) return;
    for (uint32_t l = 0; l < c->opt.n_lanes; ++l) // one pass over the slots' directory per read group that may have rows
        if (c->t8_lanes[l >> 6] >> (l & 63u) & 1ull) bqc_launch_t8_fold(c->d_t8rows, c->d_t8used, c->t8_slots_used, c->sl, c->d_state, l, c->stream);
    c->t8_lanes[0] = c->t8_lanes[1] = c->t8_lanes[2] = c->t8_lanes[3] = 0;
    c->t8_slots_used = 0;
}

// ---------------------------------------------------------------------------------------------------
// coverage anchors and tiles of a batch (host)
// ---------------------------------------------------------------------------------------------------
// The order-dependent part of OverallNumbers::coverage (OverallNumbers.hpp:84-110): every 1000-position window is numbered in
// flush order ("virtual coordinates"): a reset advances the window index by 2, a slide by 1.  A read that enters coverage()
// gets {window, position in it}; the covered interval comes from its CIGAR on the device.  Then the coverage tiles: those that
// hold a live window of some read (its first and the next one), the two windows carried in from the previous batch and the two
// carried out (W1, W1 + 1 = the last read's).  Used for the reads of a batch (host_pass) and for the set-aside reads of a
// shard (bqc_shard_resolve).
namespace {
struct CovPlanner {
    bqc_ctx* c;
    HostPass& H;
    uint32_t n, nl;
    std::vector<uint8_t> started_before;
    std::vector<uint32_t> last_rel;
    CovPlanner(bqc_ctx* c_, HostPass& H_, uint32_t n_) : c(c_), H(H_), n(n_), nl(c_->opt.n_lanes), started_before(nl), last_rel(nl, 0)
    {
        H.lane_mask.assign(nl, 0);
        H.lane_first.resize(nl);
        for (uint32_t l = 0; l < nl; ++l) {
            started_before[l] = !c->cov[l].first;
            c->cov[l].batch_base = c->cov[l].win;
            H.lane_first[l].clear();
        }
    }
    void step(uint32_t i, uint32_t lane, int32_t rid, uint32_t beginpos, uint64_t& rel, uint32_t& pos)
    {
        LaneCov& s = c->cov[lane];
        if (s.first) { s.first = false; s.id = rid; s.shift = (int32_t)beginpos; }
        if (s.id != rid || (uint32_t)(beginpos - (uint32_t)s.shift) > 2u * BQC_VSIZE) { // reset: two windows flushed
            s.id = rid; s.win += 2; s.shift = (int32_t)beginpos;
        }
        pos = beginpos - (uint32_t)s.shift;
        if (pos > BQC_VSIZE && pos < 2u * BQC_VSIZE) { // slide: one window flushed
            s.win += 1; s.shift += BQC_VSIZE; pos = beginpos - (uint32_t)s.shift;
        }
        rel = s.win - s.batch_base;
        if (rel > 0xFFFFFFF0ull) return;
        std::vector<uint32_t>& first = H.lane_first[lane]; // first[k] = first read of the lane whose window is >= k
        if (first.size() <= rel) first.resize((size_t)rel + 1, i);
        last_rel[lane] = (uint32_t)rel;
    }
    void finish()
    {
        H.tiles.clear(); H.add_idx.clear(); H.add_val.clear();
        for (uint32_t l = 0; l < nl; ++l) {
            const std::vector<uint32_t>& first = H.lane_first[l];
            if (first.empty()) continue;
            H.lane_mask[l] = 1;
            const uint32_t W1 = last_rel[l]; // windows < W1 are complete after this batch
            auto first_at = [&](uint64_t k) { return k < first.size() ? first[k] : n; };
            uint32_t last_tile = 0xFFFFFFFFu;
            uint64_t covered_final = 0;
            auto push_tile = [&](uint32_t w) {
                const uint32_t t = w / BQC_COV_TILE_WINDOWS;
                if (last_tile != 0xFFFFFFFFu && t <= last_tile) return;
                last_tile = t;
                const uint32_t wlo = t * BQC_COV_TILE_WINDOWS;
                CovTile ct{};
                ct.lane = l; ct.win_lo = wlo; ct.win_final = W1; ct.mixed = H.multi_lane ? 1u : 0u;
                ct.list_begin = first_at(wlo == 0 ? 0 : wlo - 1);
                ct.list_end = first_at((uint64_t)wlo + BQC_COV_TILE_WINDOWS);
                H.tiles.push_back(ct);
                const uint64_t hi = std::min<uint64_t>((uint64_t)wlo + BQC_COV_TILE_WINDOWS, W1);
                if (hi > wlo) covered_final += hi - wlo;
            };
            if (started_before[l]) { push_tile(0); push_tile(1); }
            // a window value w is live for some read iff a read's first window is w: first[w] != first[w + 1] (or w is the last)
            for (uint64_t w = 0; w < first.size(); ++w) {
                const bool present = w + 1 == first.size() || first[w] != first[w + 1];
                if (present) { push_tile((uint32_t)w); push_tile((uint32_t)w + 1); }
            }
            if (W1 > covered_final) { // complete windows nobody touched: depth 0 everywhere
                H.add_idx.push_back(c->sl.lane_base(l) + c->sl.o_poscov + 0);
                H.add_val.push_back((uint64_t)(W1 - covered_final) * BQC_VSIZE);
            }
            c->cov[l].batch_base = c->cov[l].win; // the next batch numbers its windows from this batch's last live window
        }
        H.started_after.resize(nl);
        for (uint32_t l = 0; l < nl; ++l) H.started_after[l] = !c->cov[l].first; // lanes that have seen a coverage read so far
    }
};
} // namespace

// ---------------------------------------------------------------------------------------------------
// host pass: sizes, read groups, coverage anchors
// ---------------------------------------------------------------------------------------------------
static int host_pass(bqc_ctx* c, const bqc_batch* b, HostPass& H)
{
    const uint32_t n = b->n_reads, nl = c->opt.n_lanes;
    H.n = n;
    // (a) on the host's cores: payload sizes, reads per read group, reads for the generic kernels
    const unsigned nt_max = std::min(16u, bqc_host_threads());
    struct Part { uint64_t s1 = 0, s2 = 0, s3 = 0; uint32_t n_slow = 0, max_slow = 0; std::vector<uint64_t> lanes; };
    std::vector<Part> part(nt_max);
    const unsigned nt = parallel_ranges(n, nt_max, 1 << 17, [&](unsigned t, size_t lo, size_t hi) {
        Part& P = part[t];
        P.lanes.assign(nl, 0);
        uint64_t s1 = 0, s2 = 0, s3 = 0;
        uint32_t ns = 0, ms = 0;
        for (size_t i = lo; i < hi; ++i) {
            const uint32_t L = b->l_seq[i], lane = b->lane[i];
            s1 += (L + 1) / 2; s2 += L; s3 += b->n_cigar[i];
            if (lane < nl) P.lanes[lane]++;
            if (c->no_fast || L > BQC_FAST_MAXLEN) { ++ns; ms = std::max(ms, L); }
        }
        P.s1 = s1; P.s2 = s2; P.s3 = s3; P.n_slow = ns; P.max_slow = ms;
    });
    H.seq_bytes = H.qual_bytes = H.cigar_words = 0;
    H.n_slow = H.max_len_slow = 0;
    H.lane_count.assign(nl, 0);
    for (unsigned t = 0; t < nt; ++t) {
        H.seq_bytes += part[t].s1; H.qual_bytes += part[t].s2; H.cigar_words += part[t].s3;
        H.n_slow += part[t].n_slow; H.max_len_slow = std::max(H.max_len_slow, part[t].max_slow);
        for (uint32_t l = 0; l < nl; ++l) H.lane_count[l] += part[t].lanes[l];
    }
    uint32_t lanes_present = 0;
    H.t8_lane = 0;
    for (uint32_t l = 0; l < nl; ++l) {
        if (H.lane_count[l]) ++lanes_present;
        if (H.lane_count[l] > H.lane_count[H.t8_lane]) H.t8_lane = l;
    }
    uint64_t in_lanes = 0;
    for (uint32_t l = 0; l < nl; ++l) in_lanes += H.lane_count[l];
    // (a read whose lane is out of range ends the run on the device — check 2 — and the batch then contributes nothing; such
    // reads still need a place in the decomposition: they go in front of read group 0)
    const uint64_t n_bad = n - in_lanes;
    H.multi_lane = lanes_present > 1 || (n_bad > 0 && in_lanes > 0);
    // (b) processing order: reads grouped by read group (stable); lane stretches and super-windows
    H.stretches.clear(); H.sws.clear();
    auto add_stretch = [&](uint32_t lane, uint32_t begin, uint32_t count) {
        if (!count) return;
        Stretch S{lane, (uint32_t)H.sws.size(), 0, 0};
        for (uint32_t p = 0; p < count; p += BQC_SW_READS)
            H.sws.push_back(SuperWindow{lane, begin + p, std::min<uint32_t>(BQC_SW_READS, count - p), (uint32_t)H.stretches.size()});
        S.sw_end = (uint32_t)H.sws.size();
        H.stretches.push_back(S);
    };
    if (!H.multi_lane) {
        H.order.clear();
        uint32_t lane = 0;
        for (uint32_t l = 0; l < nl; ++l) if (H.lane_count[l]) lane = l;
        add_stretch(lane, 0, n);
    } else {
        std::vector<uint64_t> w(nl); // where the next read of each read group goes
        uint64_t at = n_bad, w_bad = 0;
        for (uint32_t l = 0; l < nl; ++l) { w[l] = at; at += H.lane_count[l]; }
        for (uint32_t l = 0; l < nl; ++l) add_stretch(l, (uint32_t)(l == 0 ? 0 : w[l]), (uint32_t)(H.lane_count[l] + (l == 0 ? n_bad : 0)));
        { const size_t cap = H.order.capacity(); H.order.resize(n); if (H.order.capacity() != cap) advise_huge(H.order); }
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t lane = b->lane[i];
            if (lane < nl) H.order[w[lane]++] = i; else H.order[w_bad++] = i;
        }
    }
    // (c) the order-dependent part of OverallNumbers::coverage, in stream order (CovPlanner)
    { const size_t cap = H.cov.capacity(); H.cov.resize(n); if (H.cov.capacity() != cap) advise_huge(H.cov); }
    CovPlanner plan(c, H, n);
    ShardCtx& sh = c->shard;
    const bool set_aside = sh.tail && !sh.resolved;
    PendBatch* pb = nullptr;
    H.n_pending = 0;
    const uint32_t n_refs = c->opt.n_refs;
    const uint8_t* main_chrom = c->main_chrom.data();
    if (!set_aside) {
        // The common case, kept tight — this loop is the one serial pass over every read of the run (the submitting thread's time is what
        // bounds a whole-genome file once the card reads and inflates it): raw pointers, the read group's state in locals when there is
        // one read group, the window list touched only when the window changes.
        const uint16_t* const flags = b->flag;
        const uint8_t* const lanes = b->lane;
        const int32_t* const rids = b->rid;
        const int32_t* const poss = b->pos;
        CovEntry* const out = H.cov.data();
        LaneCov* const cov = c->cov.data();
        std::vector<uint32_t>* const lane_first = H.lane_first.data();
        uint32_t* const last_rel = plan.last_rel.data();
        LaneCov local = cov[0];
        const bool one = nl == 1;
        bool too_many = false;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t flag = flags[i], lane = lanes[i];
            const int32_t rid = rids[i];
            // primary record with a first / last flag, on a main chromosome, mapped, not a duplicate (bamqualcheck.cpp:318-327,392,430-433)
            if ((flag & 0xD04u) || !(flag & 0xC0u) || (uint32_t)rid >= n_refs || !main_chrom[rid] || lane >= nl) { out[i] = CovEntry{BQC_COV_NONE, 0}; continue; }
            LaneCov& st = one ? local : cov[lane];
            const uint32_t beginpos = (uint32_t)poss[i];
            // CovPlanner::step, inlined (OverallNumbers.hpp:84-110)
            if (st.first) { st.first = false; st.id = rid; st.shift = (int32_t)beginpos; }
            if (st.id != rid || (uint32_t)(beginpos - (uint32_t)st.shift) > 2u * BQC_VSIZE) { st.id = rid; st.win += 2; st.shift = (int32_t)beginpos; } // reset: two windows flushed
            uint32_t pos = beginpos - (uint32_t)st.shift;
            if (pos > BQC_VSIZE && pos < 2u * BQC_VSIZE) { st.win += 1; st.shift += BQC_VSIZE; pos = beginpos - (uint32_t)st.shift; } // slide: one window flushed
            const uint64_t rel = st.win - st.batch_base;
            if (rel > 0xFFFFFFF0ull) { too_many = true; break; }
            std::vector<uint32_t>& first = lane_first[lane]; // first[k] = first read of the lane whose window is >= k
            if (first.size() <= rel) first.resize((size_t)rel + 1, i);
            last_rel[lane] = (uint32_t)rel;
            out[i] = CovEntry{(uint32_t)rel, pos};
        }
        if (one) cov[0] = local;
        if (too_many) return bqc_fail(c, BQC_ERR_ARG, "batch spans too many coverage windows (split the batch)");
    } else
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t flag = b->flag[i], lane = b->lane[i];
        const int32_t rid = b->rid[i];
        // primary record with a first / last flag, on a main chromosome, mapped, not a duplicate (bamqualcheck.cpp:318-327,392,430-433)
        const bool cand = !(flag & 0x900u) && (flag & 0xC0u) && !(flag & 0x4u) && !(flag & 0x400u) && rid >= 0 && (uint32_t)rid < n_refs && main_chrom[rid] &&
                          lane < nl;
        if (!cand) { H.cov[i] = CovEntry{BQC_COV_NONE, 0}; continue; }
        const uint32_t beginpos = (uint32_t)b->pos[i];
        if (set_aside && sh.pending[lane]) {
            // Does the state machine reset at this read WHATEVER its state?  Its position in the live windows is
            // (beginPos - previous beginPos) + (previous read's position, 0..2000): a reset for certain iff the chromosome changes or
            // that whole range lies above 2000 in the reference's unsigned arithmetic.
            const uint32_t d = beginpos - sh.prev_bp[lane];
            const bool certain = sh.has_prev[lane] && (rid != sh.prev_rid[lane] || (d > 2u * BQC_VSIZE && d <= 0xFFFFFFFFu - 2u * BQC_VSIZE));
            sh.has_prev[lane] = 1; sh.prev_rid[lane] = rid; sh.prev_bp[lane] = beginpos;
            if (!certain) {
                if (!pb) { sh.batches.emplace_back(); pb = &sh.batches.back(); }
                H.cov[i] = CovEntry{BQC_COV_PENDING, pb->n()};
                pb->lane.push_back((uint8_t)lane); pb->rid.push_back(rid); pb->bp.push_back(beginpos);
                continue;
            }
            sh.pending[lane] = 0; // from here on this read group runs as a stream of its own would from this read
        }
        uint64_t rel;
        uint32_t pos;
        plan.step(i, lane, rid, beginpos, rel, pos);
        if (rel > 0xFFFFFFF0ull) return bqc_fail(c, BQC_ERR_ARG, "batch spans too many coverage windows (split the batch)");
        H.cov[i] = CovEntry{(uint32_t)rel, pos};
    }
    H.n_pending = pb ? pb->n() : 0;
    plan.finish();
    for (uint32_t e = 0; e < b->n_nm_extra; ++e)
        if (b->nm_extra_read[e] >= n) return bqc_fail(c, BQC_ERR_ARG, "nm_extra_read out of range");
    return 0;
}

// The same for a batch that was anchored on the card (k_anchor.hip): no pass over the reads — the batch's sizes, the read group's
// state before and behind it and the reads at which the window index changes come with the summary; the tiles follow from those.
static int host_pass_anchored(bqc_ctx* c, uint32_t n, const bqc_anchored* a, HostPass& H)
{
    const AnchorSummary& S = *a->h_sum;
    H.n = n;
    H.seq_bytes = S.seq_bytes; H.qual_bytes = S.qual_bytes; H.cigar_words = S.cigar_words;
    H.n_slow = S.n_slow; H.max_len_slow = S.max_len_slow;
    H.lane_count.assign(1, n);
    H.t8_lane = 0;
    H.multi_lane = false;
    H.order.clear();
    H.stretches.clear(); H.sws.clear();
    {
        Stretch St{0, 0, 0, 0};
        for (uint32_t p = 0; p < n; p += BQC_SW_READS) H.sws.push_back(SuperWindow{0, p, std::min<uint32_t>(BQC_SW_READS, n - p), 0});
        St.sw_end = (uint32_t)H.sws.size();
        H.stretches.push_back(St);
    }
    H.n_pending = 0;
    if (S.n_pending) { // a shard in the middle of the stream: the reads set aside go to the pending log, as host_pass would have put them there
        ShardCtx& sh = c->shard;
        sh.batches.emplace_back();
        PendBatch& pb = sh.batches.back();
        pb.lane.assign(S.n_pending, 0);
        pb.rid = a->pend_rid;
        pb.bp = a->pend_bp;
        H.n_pending = S.n_pending;
    }
    if (c->shard.tail) { c->shard.pending[0] = S.after.pending ? 1 : 0; c->shard.has_prev[0] = S.after.has_prev ? 1 : 0; c->shard.prev_rid[0] = S.after.prev_rid; c->shard.prev_bp[0] = S.after.prev_bp; }
    // the read group's state in front of the batch, as the planner wants to find it
    LaneCov& lc = c->cov[0];
    lc.first = S.before.first != 0; lc.id = S.before.id; lc.shift = S.before.shift; lc.win = S.before.win;
    CovPlanner plan(c, H, n);
    if (S.n_cand > S.n_pending) {
        // first[k] = the first read whose window is >= k: the card's first_of[] (the first read whose window IS k, AN_NO_READ where a
        // reset skipped k), filled from the back
        const size_t total = (size_t)S.last_rel + 1, inl = std::min<size_t>(total, bqc_anchored::kInline);
        if (total > inl + a->rest.size()) return bqc_fail(c, BQC_ERR_DEVICE, "internal error: the anchors' window table is shorter than the last window");
        std::vector<uint32_t>& first = H.lane_first[0];
        first.assign(total, AN_NO_READ);
        uint32_t cur = AN_NO_READ;
        for (size_t k = total; k-- > 0;) {
            const uint32_t v = k < inl ? a->h_bound[k] : a->rest[k - inl];
            if (v != AN_NO_READ) cur = v;
            first[k] = cur;
        }
        plan.last_rel[0] = S.last_rel;
        if (first[total - 1] == AN_NO_READ) return bqc_fail(c, BQC_ERR_DEVICE, "internal error: the anchors' window table does not end at the last window");
    }
    lc.first = S.after.first != 0; lc.id = S.after.id; lc.shift = S.after.shift; lc.win = S.after.win; // (batch_base: set by the planner, advanced by finish())
    plan.finish();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// device image of a batch
// ---------------------------------------------------------------------------------------------------
namespace {
struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; }
};
}

// carve m.dmem (allocating / growing it) for a batch of the given sizes and fill in the DevBatch / PrepArgs pointers
// in_place (an anchored batch: every column and the anchors live in the caller's device memory until the batch's kernels are through):
// the image holds no copy of them — the kernels read them where they are (250 MB per million reads that used to be copied once more)
static int layout_batch(bqc_ctx* c, BatchMem& m, const bqc_batch* b, const HostPass& H, bool from_pool, const CovEntry* in_place = nullptr)
{
    const uint64_t n = H.n;
    const uint32_t nl = c->opt.n_lanes;
    Carver cv;
    cv.take(256);
    m.h2d_begin = cv.off;
    const size_t cb[13] = {2 * n, n, n, 4 * n, 4 * n, 4 * n, 4 * n, 4 * n, 4 * n, 2 * n, H.seq_bytes, H.qual_bytes, 4 * H.cigar_words};
    for (int k = 0; k < 13; ++k) { m.col_bytes[k] = in_place ? 0 : cb[k]; m.o_col[k] = cv.take(in_place ? 0 : cb[k] + (k >= 10 ? 512 : 0)); }
    m.o_xr = cv.take(4ull * b->n_nm_extra); m.o_xv = cv.take(4ull * b->n_nm_extra);
    m.o_cov_in = cv.take(in_place ? 0 : sizeof(CovEntry) * n);
    m.o_order = cv.take(H.multi_lane ? 4 * n : 0);
    m.o_sws = cv.take(sizeof(SuperWindow) * H.sws.size());
    m.o_stretch = cv.take(sizeof(Stretch) * H.stretches.size());
    m.o_tiles = cv.take(sizeof(CovTile) * H.tiles.size());
    m.o_mask = cv.take(nl); m.o_started = cv.take(nl);
    m.o_aidx = cv.take(8 * H.add_idx.size()); m.o_aval = cv.take(8 * H.add_val.size());
    m.h2d_end = cv.off;
    // outputs of the device pre-pass
    const uint64_t nblk = (n + 1023) / 1024, n_sw = H.sws.size(), n_st = H.stretches.size();
    const uint64_t perm_cap = 3 * n + H.cigar_words + 128 * (n_sw + n_st + 1);
    const uint64_t cf_cap = (2 * n + 128 * n_sw) / 256 + H.cigar_words / 256 + 4 * n_st + 16;
    const uint64_t cs_cap = H.n_slow / BQC_CHUNK_READS + n_st + 4;
    const uint64_t cx_cap = H.cigar_words / 2 + 1;
    if (perm_cap > 0xFFFFFFF0ull) return bqc_fail(c, BQC_ERR_ARG, "batch too large (split the batch)");
    const size_t o_flag2 = cv.take(2 * n), o_soff = cv.take(4 * n), o_qoff = cv.take(4 * n), o_cgoff = cv.take(4 * n), o_cov = cv.take(sizeof(CovEntry) * n),
                 o_covx = cv.take(sizeof(CovExtra) * cx_cap), o_cls = cv.take(2 * n), o_segs = cv.take(sizeof(TripSeg) * (H.cigar_words + 1)),
                 o_perm = cv.take(4 * perm_cap), o_cf = cv.take(sizeof(Chunk) * cf_cap), o_cs = cv.take(sizeof(Chunk) * cs_cap),
                 o_desc = cv.take(sizeof(BatchDesc)), o_err = cv.take(sizeof(ErrRec)), o_cursave = cv.take(8), o_bsz = cv.take(24 * nblk), o_btgt = cv.take(8 * nblk),
                 o_bmf = cv.take(8 * nblk), o_swc = cv.take(sizeof(SwCounts) * n_sw), o_swp = cv.take(sizeof(SwPlan) * n_sw),
                 o_rsum = cv.take(H.n_slow ? 12 * n : 0);
    const size_t need = cv.off + 256;
    if (m.dcap < need) {
        if (m.dmem) { (void)hipFree(m.dmem); m.dmem = nullptr; m.dcap = 0; }
        if (from_pool)
            for (size_t k = 0; k < c->pool.size(); ++k)
                if (c->pool[k].second >= need && c->pool[k].second <= 2 * need + (64u << 20)) { // a freed buffer of a similar size
                    m.dmem = c->pool[k].first; m.dcap = c->pool[k].second;
                    c->pool.erase(c->pool.begin() + k);
                    break;
                }
        if (!m.dmem) {
            const size_t cap = need + need / 8; // a little slack, so that the next batch of about this size fits as well
            const hipError_t he = hipMalloc(&m.dmem, cap);
            if (he != hipSuccess) { m.dmem = nullptr; return bqc_fail(c, BQC_ERR_DEVICE, "hipMalloc(%zu) failed: %s", cap, hipGetErrorString(he)); }
            m.dcap = cap;
        }
    }
    char* base = (char*)m.dmem;
    DevBatch& d = m.d;
    d = DevBatch{};
    d.n_reads = (uint32_t)n;
    d.flag = (const uint16_t*)(base + o_flag2);
    d.mapq = (const uint8_t*)(base + m.o_col[1]); d.lane = (const uint8_t*)(base + m.o_col[2]);
    d.rid = (const int32_t*)(base + m.o_col[3]); d.pos = (const int32_t*)(base + m.o_col[4]); d.tlen = (const int32_t*)(base + m.o_col[5]);
    d.nm = (const int32_t*)(base + m.o_col[6]); d.as_ = (const int32_t*)(base + m.o_col[7]); d.l_seq = (const uint32_t*)(base + m.o_col[8]);
    d.n_cigar = (const uint16_t*)(base + m.o_col[9]);
    d.seq = (const uint8_t*)(base + m.o_col[10]); d.qual = (const uint8_t*)(base + m.o_col[11]); d.cigar = (const uint32_t*)(base + m.o_col[12]);
    d.seq_off = (const uint32_t*)(base + o_soff); d.qual_off = (const uint32_t*)(base + o_qoff); d.cigar_off = (const uint32_t*)(base + o_cgoff);
    d.order = H.multi_lane ? (const uint32_t*)(base + m.o_order) : nullptr;
    d.perm = (const uint32_t*)(base + o_perm);
    d.chunks = (const Chunk*)(base + o_cs); d.chunks_fast = (const Chunk*)(base + o_cf);
    d.desc = (const BatchDesc*)(base + o_desc);
    d.nm_extra_read = (const uint32_t*)(base + m.o_xr); d.nm_extra_val = (const int32_t*)(base + m.o_xv); d.n_nm_extra = b->n_nm_extra;
    d.cov = (const CovEntry*)(base + o_cov); d.cov_extra = (const CovExtra*)(base + o_covx);
    d.cov_tiles = (const CovTile*)(base + m.o_tiles); d.n_cov_tiles = (uint32_t)H.tiles.size();
    d.segs = (const TripSeg*)(base + o_segs);
    PrepArgs& p = m.prep;
    p = PrepArgs{};
    p.n = (uint32_t)n; p.n_lanes = nl; p.max_read_len = c->opt.max_read_len; p.no_fast = c->no_fast ? 1u : 0u; p.replay = 0;
    p.flag_in = (const uint16_t*)(base + m.o_col[0]); p.mapq = d.mapq; p.lane = d.lane; p.rid = d.rid; p.pos = d.pos; p.as_ = d.as_;
    p.l_seq = d.l_seq; p.n_cigar = d.n_cigar; p.qual = d.qual; p.cigar = d.cigar;
    p.cov_in = (const CovEntry*)(base + m.o_cov_in); p.order = d.order;
    p.sws = (const SuperWindow*)(base + m.o_sws); p.n_sw = (uint32_t)n_sw;
    p.stretches = (const Stretch*)(base + m.o_stretch); p.n_stretch = (uint32_t)n_st;
    p.fasta_index = c->d_fasta_index;
    p.flag_out = (uint16_t*)(base + o_flag2); p.seq_off = (uint32_t*)(base + o_soff); p.qual_off = (uint32_t*)(base + o_qoff);
    p.cigar_off = (uint32_t*)(base + o_cgoff); p.cov_out = (CovEntry*)(base + o_cov); p.cov_extra = (CovExtra*)(base + o_covx);
    p.cov_extra_cap = (uint32_t)std::min<uint64_t>(cx_cap, 0xFFFFFFFFull); p.cls = (uint16_t*)(base + o_cls); p.segs = (TripSeg*)(base + o_segs);
    p.perm = (uint32_t*)(base + o_perm); p.perm_cap = (uint32_t)perm_cap;
    p.chunks_fast = (Chunk*)(base + o_cf); p.chunks_fast_cap = (uint32_t)cf_cap;
    p.chunks_slow = (Chunk*)(base + o_cs); p.chunks_slow_cap = (uint32_t)cs_cap;
    p.desc = (BatchDesc*)(base + o_desc); p.err = (ErrRec*)(base + o_err);
    p.cursor = c->d_cursor; p.cursor_save = (int32_t*)(base + o_cursave);
    p.blk_sizes = (unsigned long long*)(base + o_bsz); p.blk_tgt = (uint32_t*)(base + o_btgt); p.blk_maxfast = (uint32_t*)(base + o_bmf);
    p.sw_counts = (SwCounts*)(base + o_swc); p.sw_plan = (SwPlan*)(base + o_swp);
    m.d_lane_mask = (uint8_t*)(base + m.o_mask); m.d_started_after = (uint8_t*)(base + m.o_started);
    m.d_add_idx = (uint64_t*)(base + m.o_aidx); m.d_add_val = (uint64_t*)(base + m.o_aval); m.n_add = (uint32_t)H.add_idx.size();
    m.d_rsum = H.n_slow ? (uint32_t*)(base + o_rsum) : nullptr;
    m.d_err = p.err;
    m.algo_bytes = 48ull * n + H.seq_bytes + H.qual_bytes + 4 * H.cigar_words; // A(L,n) of SURVEY.md §8d summed over the batch
    m.n_slow = H.n_slow; m.max_len_slow = H.max_len_slow; m.n_chunks_slow_ub = (uint32_t)cs_cap; m.t8_lane = H.t8_lane;
    m.lane_bits[0] = m.lane_bits[1] = m.lane_bits[2] = m.lane_bits[3] = 0;
    for (size_t l = 0; l < H.lane_count.size() && l < 256; ++l) if (H.lane_count[l]) m.lane_bits[l >> 6] |= 1ull << (l & 63);
    if (in_place) { // the columns where the caller has them
        d.mapq = b->mapq; d.lane = b->lane; d.rid = b->rid; d.pos = b->pos; d.tlen = b->tlen; d.nm = b->nm; d.as_ = b->as; d.l_seq = b->l_seq; d.n_cigar = b->n_cigar;
        d.seq = b->seq; d.qual = b->qual; d.cigar = b->cigar;
        p.flag_in = b->flag; p.mapq = d.mapq; p.lane = d.lane; p.rid = d.rid; p.pos = d.pos; p.as_ = d.as_; p.l_seq = d.l_seq; p.n_cigar = d.n_cigar; p.qual = d.qual; p.cigar = d.cigar;
        p.cov_in = in_place;
    }
    m.processed = false;
    return 0;
}

// the tables of the host pass into the image (host memory laid out like the device buffer from h2d_begin on)
static void fill_tables(const BatchMem& m, const bqc_batch* b, const HostPass& H, char* img, size_t img_begin /* device offset of img[0]: <= o_xr */, bool anchors_on_device = false)
{
    auto at = [&](size_t off) { return img + (off - img_begin); };
    if (b->n_nm_extra) { memcpy(at(m.o_xr), b->nm_extra_read, 4ull * b->n_nm_extra); memcpy(at(m.o_xv), b->nm_extra_val, 4ull * b->n_nm_extra); }
    if (!anchors_on_device) memcpy(at(m.o_cov_in), H.cov.data(), sizeof(CovEntry) * (size_t)H.n);
    if (H.multi_lane) memcpy(at(m.o_order), H.order.data(), 4ull * H.n);
    memcpy(at(m.o_sws), H.sws.data(), sizeof(SuperWindow) * H.sws.size());
    memcpy(at(m.o_stretch), H.stretches.data(), sizeof(Stretch) * H.stretches.size());
    if (!H.tiles.empty()) memcpy(at(m.o_tiles), H.tiles.data(), sizeof(CovTile) * H.tiles.size());
    memcpy(at(m.o_mask), H.lane_mask.data(), H.lane_mask.size());
    memcpy(at(m.o_started), H.started_after.data(), H.started_after.size());
    if (!H.add_idx.empty()) { memcpy(at(m.o_aidx), H.add_idx.data(), 8 * H.add_idx.size()); memcpy(at(m.o_aval), H.add_val.data(), 8 * H.add_val.size()); }
}

static const void* column_ptr(const bqc_batch* b, int k)
{
    switch (k) {
    case 0: return b->flag; case 1: return b->mapq; case 2: return b->lane; case 3: return b->rid; case 4: return b->pos; case 5: return b->tlen;
    case 6: return b->nm; case 7: return b->as; case 8: return b->l_seq; case 9: return b->n_cigar; case 10: return b->seq; case 11: return b->qual;
    default: return b->cigar;
    }
}

// ---------------------------------------------------------------------------------------------------
// launches of one batch (compute stream, in batch order)
// ---------------------------------------------------------------------------------------------------
static void tick(bqc_ctx* c, const char* name)
{
    if (!c->timing || c->n_timed + 1 >= (int)c->ev.size()) return;
    (void)hipEventRecord(c->ev[c->n_timed + 1], c->stream);
    c->tnames.push_back(name);
    c->n_timed++;
}

static int enqueue_kernels(bqc_ctx* c, BatchMem& m)
{
    DevRefs refs{(const uint8_t* const*)c->d_ref_ptrs, c->d_ref_len, c->d_main, c->opt.n_refs, (const uint32_t* const*)c->d_refn_ptrs};
    uint32_t* err = &m.d_err->flags;
    const DevBatch& d = m.d;
    if (d.n_reads == 0) { if (c->timing) { c->n_timed = 0; c->tnames.clear(); } return 0; }
    if (c->timing) { c->n_timed = 0; c->tnames.clear(); (void)hipEventRecord(c->ev[0], c->stream); }
    m.prep.replay = m.processed ? 1u : 0u;
    bqc_launch_prep(m.prep, refs, c->stream);
    m.processed = true;
    tick(c, "k_prep");
    if (!d.n_cov_tiles) bqc_launch_or_bytes(c->d_started, m.d_started_after, c->opt.n_lanes, c->stream); // (else: k_cov's epilogue)
    if (d.n_reads > m.n_slow) { // reads on the short-read fast path
        const uint32_t grid = c->n_cu; // one workgroup per CU (the chunk count is on the device); every workgroup owns a slot of scratch rows
        if (!(bqc_short_parts() & 8u)) { // (profiling only: per-read statistics of the fast chunks as a separate kernel)
            bqc_launch_reads_chunks(d, c->sl, c->d_state, refs, err, c->n_cu * 8, 1, c->stream);
            tick(c, "k_reads");
        }
        if (c->t8_slots_used + grid > c->t8_slots_cap) bqc_state_ready(c);
        for (int k = 0; k < 4; ++k) c->t8_lanes[k] |= m.lane_bits[k];
        bqc_launch_short(d, c->sl, c->d_state, refs, err, grid, c->d_t8rows + (size_t)c->t8_slots_used * BQC_T8_SPW * 16384u,
                         c->d_t8used + (size_t)c->t8_slots_used * BQC_T8_USED, c->stream);
        c->t8_slots_used += grid;
        tick(c, "k_short");
    }
    if (m.n_slow) {
        bqc_launch_reads_chunks(d, c->sl, c->d_state, refs, err, std::min(m.n_chunks_slow_ub, c->n_cu * 8), 0, c->stream);
        tick(c, "k_reads(generic)");
        HIPCHK(c, hipMemsetAsync(m.d_rsum, 0, 12ull * d.n_reads, c->stream));
        const uint32_t slots = bqc_long_slots(m.max_len_slow, m.n_chunks_slow_ub, c->n_cu); // a slot of 8-mer scratch rows per workgroup, as for k_short
        if (c->t8_slots_used + slots > c->t8_slots_cap) bqc_state_ready(c);
        for (int k = 0; k < 4; ++k) c->t8_lanes[k] |= m.lane_bits[k];
        bqc_launch_long(d, c->sl, c->d_state, refs, err, m.d_rsum, m.max_len_slow, m.n_chunks_slow_ub, c->n_cu,
                        c->d_t8rows + (size_t)c->t8_slots_used * BQC_T8_SPW * 16384u, c->d_t8used + (size_t)c->t8_slots_used * BQC_T8_USED, m.t8_lane, c->d_kl_cyc,
                        c->d_kl_cyc_used, c->stream);
        c->t8_slots_used += slots;
        tick(c, "k_long");
    }
    if (d.n_nm_extra) bqc_launch_nm_extra(d, c->sl, c->d_state, refs, err, c->stream);
    if (d.n_cov_tiles) bqc_launch_cov(d, c->sl, c->d_state, c->d_carry, c->d_parity, m.d_lane_mask, c->d_started, m.d_started_after, c->opt.n_lanes, c->stream);
    bqc_launch_add_words(c->d_state, m.d_add_idx, m.d_add_val, m.n_add, c->stream);
    tick(c, "k_cov");
    if (c->sketch) { sketch_process(c->sketch, d, c->stream); tick(c, "k_sketch"); }
    bqc_launch_err_merge(c->d_err0, m.d_err, c->stream); // the first batch with an error defines the stream's error
    HIPCHK(c, hipGetLastError());
    return 0;
}

// what the device found wrong with a batch -> error code and message (the reference prints a message and exits 1)
int bqc_report_errors(bqc_ctx* c, const ErrRec& e)
{
    if (e.first_key != BQC_ERRKEY_NONE) {
        const uint32_t i = (uint32_t)(e.first_key >> 3), order = (uint32_t)(e.first_key & 7u);
        switch (order) {
        case 1: return bqc_fail(c, BQC_ERR_RANGE, "read %u is %u bases long; max_read_len is %u", i, e.aux0, c->opt.max_read_len);
        case 2: return bqc_fail(c, BQC_ERR_ARG, "read %u: lane %u out of range", i, e.aux0);
        case 3: return bqc_fail(c, BQC_ERR_ARG, "batch too large: payload offsets exceed 32 bits (split the batch)");
        case 4: return bqc_fail(c, BQC_ERR_AS_TAG, "ERROR: read %u has no usable AS tag.", i);
        case 5: return bqc_fail(c, BQC_ERR_FASTA, "ERROR: Could not read fasta record for reference id %d (read %u)", (int32_t)e.aux0, i);
        default: return bqc_fail(c, BQC_ERR_NO_MATE_FLAG, "ERROR: No first or second flag in read %u", i);
        }
    }
    if (!e.flags) return 0;
    if (e.flags & BQC_DEVERR_INTERNAL) return bqc_fail(c, BQC_ERR_DEVICE, "internal error: kernel layout assumption violated");
    if (e.flags & BQC_DEVERR_MATE) return bqc_fail(c, BQC_ERR_NO_MATE_FLAG, "ERROR: No first or second flag in read");
    if (e.flags & BQC_DEVERR_RANGE) return bqc_fail(c, BQC_ERR_RANGE, "mismatch/deletion/insertion count exceeds hist_cap (or NM < D+I)");
    return bqc_fail(c, BQC_ERR_RANGE, "base quality above 222 cannot be represented by the reference (q+33 wraps)");
}
static int poison(bqc_ctx* c, int code) { c->poisoned = true; c->poison_code = code; return code; }

// ---------------------------------------------------------------------------------------------------
// resident batches: bqc_upload / bqc_process (benchmark, tests; everything synchronous on the compute stream)
// ---------------------------------------------------------------------------------------------------
extern "C" void bqc_dbatch_free(bqc_ctx* c, bqc_dbatch* db)
{
    if (!db) return;
    if (c) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
    if (c && db->m.dmem && c->pool.size() < 3) c->pool.emplace_back(db->m.dmem, db->m.dcap);
    else (void)hipFree(db->m.dmem);
    delete db;
}
extern "C" uint64_t bqc_dbatch_bytes(const bqc_dbatch* db) { return db ? db->m.algo_bytes : 0; }

static int check_batch_args(bqc_ctx* c, const bqc_batch* b, const char* who)
{
    if (!c || !b) return bqc_fail(c, BQC_ERR_ARG, "%s: null argument", who);
    if (c->poisoned) return bqc_fail(c, BQC_ERR_STATE, "context is in an error state: %s", c->err.c_str());
    if (c->flushed) return bqc_fail(c, BQC_ERR_STATE, "%s after bqc_flush/bqc_finalize (call bqc_reset first)", who);
    if (b->n_reads && (!b->flag || !b->mapq || !b->lane || !b->rid || !b->pos || !b->tlen || !b->nm || !b->as || !b->l_seq || !b->n_cigar))
        return bqc_fail(c, BQC_ERR_ARG, "%s: null column", who);
    if (b->n_nm_extra && (!b->nm_extra_read || !b->nm_extra_val)) return bqc_fail(c, BQC_ERR_ARG, "%s: null nm_extra column", who);
    return 0;
}

extern "C" int bqc_upload(bqc_ctx* c, const bqc_batch* b, bqc_dbatch** out)
{
    if (!out) return bqc_fail(c, BQC_ERR_ARG, "bqc_upload: null argument");
    int rc = check_batch_args(c, b, "bqc_upload");
    if (rc) return rc;
    if (c->shard.tail && !c->shard.resolved) return bqc_fail(c, BQC_ERR_STATE, "bqc_upload: resident batches are not available to a shard_tail context (use bqc_submit)");
    HIPCHK(c, hipSetDevice(c->device));
    c->anchor.mode = 2; // (the host keeps the window state)
    const auto t0 = clk::now();
    HostPass& H = c->hp;
    if ((rc = host_pass(c, b, H))) return poison(c, rc);
    const double t_pass = secs_since(t0);
    bqc_dbatch* db = new bqc_dbatch();
    if ((rc = layout_batch(c, db->m, b, H, true))) { delete db; return poison(c, rc); }
    BatchMem& m = db->m;
    char* base = (char*)m.dmem;
    std::vector<char> img(m.h2d_end - m.o_xr);
    fill_tables(m, b, H, img.data(), m.o_xr);
    hipError_t he = hipSuccess;
    for (int k = 0; k < 13 && he == hipSuccess; ++k)
        if (m.col_bytes[k]) he = hipMemcpyAsync(base + m.o_col[k], column_ptr(b, k), m.col_bytes[k], hipMemcpyHostToDevice, c->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(base + m.o_xr, img.data(), img.size(), hipMemcpyHostToDevice, c->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream); // buffers may be reused by the caller on return
    if (he != hipSuccess) { bqc_dbatch_free(c, db); poison(c, BQC_ERR_DEVICE); return bqc_fail(c, BQC_ERR_DEVICE, "upload failed: %s", hipGetErrorString(he)); }
    if (timing_level() == 2)
        fprintf(stderr, "[timing] upload of %u reads: host pass %.4f s, H2D %.4f s (%.1f MB)\n", b->n_reads, t_pass, secs_since(t0) - t_pass, (m.h2d_end - m.h2d_begin) / 1e6);
    db->cov_after = c->cov;
    db->seq = ++c->upload_counter;
    c->state_seq = db->seq;
    *out = db;
    return 0;
}

extern "C" int bqc_process(bqc_ctx* c, bqc_dbatch* db)
{
    if (!c || !db) return bqc_fail(c, BQC_ERR_ARG, "bqc_process: null argument");
    if (c->poisoned) return bqc_fail(c, BQC_ERR_STATE, "context is in an error state: %s", c->err.c_str());
    if (c->flushed) return bqc_fail(c, BQC_ERR_STATE, "bqc_process after bqc_flush (call bqc_reset first)");
    HIPCHK(c, hipSetDevice(c->device));
    if (db->seq > c->state_seq) { // re-processing after bqc_reset: this batch (uploaded on a fresh context) defines the stream state again
        c->cov = db->cov_after;
        c->state_seq = db->seq;
    }
    return enqueue_kernels(c, db->m);
}

// ---------------------------------------------------------------------------------------------------
// the submit pipeline
// ---------------------------------------------------------------------------------------------------
static int retire_slot(bqc_ctx* c, Slot& s) // wait for the slot's batch and read what the device found
{
    if (!s.busy) return 0;
    HIPCHK(c, hipEventSynchronize(s.ev_done));
    s.busy = false;
    if (s.ticket > c->checked_ticket) c->checked_ticket = s.ticket;
    const int rc = bqc_report_errors(c, *s.h_err);
    return rc ? poison(c, rc) : 0;
}

int bqc_drain(bqc_ctx* c)
{
    int first = 0;
    for (uint64_t t = c->checked_ticket + 1; t < c->next_ticket; ++t) { // in submission order: the first failing batch wins
        Slot& s = c->slots[t % bqc_ctx::kSlots];
        if (!s.busy || s.ticket != t) continue;
        if (first) { (void)hipEventSynchronize(s.ev_done); s.busy = false; continue; }
        const int rc = retire_slot(c, s);
        if (rc) first = rc;
    }
    if (c->next_ticket) c->checked_ticket = c->next_ticket - 1;
    return first;
}

void bqc_pipeline_destroy(bqc_ctx* c)
{
    for (Slot& s : c->slots) {
        if (s.busy && s.ev_done) (void)hipEventSynchronize(s.ev_done);
        if (s.m.dmem) (void)hipFree(s.m.dmem);
        if (s.hmem) (void)hipHostFree(s.hmem);
        if (s.h_err) (void)hipHostFree(s.h_err);
        if (s.ev_h2d) (void)hipEventDestroy(s.ev_h2d);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
        s = Slot();
    }
    for (auto& pb : c->pool) (void)hipFree(pb.first);
    c->pool.clear();
    for (PendBatch& pb : c->shard.batches) if (pb.dmem) (void)hipFree(pb.dmem);
    c->shard.batches.clear();
}

static int submit_impl(bqc_ctx* c, const bqc_batch* b, bool pinned_columns, uint64_t* ticket_out, const bqc_anchored* anchored = nullptr)
{
    int rc = check_batch_args(c, b, "bqc_submit");
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if (ticket_out) *ticket_out = 0;
    if (!anchored) c->anchor.mode = 2; // (the host keeps the window state from here on: bqc_anchor_enqueue refuses)
    if (b->n_reads == 0) return 0;
    if ((rc = bqc_copy_stream(c))) return rc;
    const uint64_t ticket = c->next_ticket;
    Slot& s = c->slots[ticket % bqc_ctx::kSlots];
    const auto t0 = clk::now();
    if ((rc = retire_slot(c, s))) return rc; // the batch submitted kSlots calls ago: its errors surface here
    const double t_wait = secs_since(t0);
    if (!s.ev_done) {
        HIPCHK(c, hipEventCreateWithFlags(&s.ev_h2d, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
        HIPCHK(c, hipHostMalloc((void**)&s.h_err, sizeof(ErrRec), hipHostMallocDefault));
    }
    HostPass& H = c->hp;
    if ((rc = anchored ? host_pass_anchored(c, b->n_reads, anchored, H) : host_pass(c, b, H))) return poison(c, rc);
    const double t_pass = secs_since(t0) - t_wait;
    static const bool copy_anyway = getenv("BQC_ANCHORED_COPY") && getenv("BQC_ANCHORED_COPY")[0] == '1'; // (A/B: the columns copied into the image as bqc_submit_async does)
    const bool in_place = anchored && !copy_anyway;
    if ((rc = layout_batch(c, s.m, b, H, false, in_place ? anchored->d_cov : nullptr))) return poison(c, rc);
    s.in_place = in_place;
    BatchMem& m = s.m;
    if (H.n_pending) { // shard mode: the covered runs of the reads set aside stay on the device until bqc_shard_resolve
        PendBatch& pb = c->shard.batches.back();
        pb.extra_cap = (uint32_t)std::min<uint64_t>(H.cigar_words / 2 + 1, 0xFFFFFFF0ull);
        const size_t bytes = sizeof(PendRun) * (size_t)pb.n() + 256 + sizeof(PendExtra) * (size_t)pb.extra_cap;
        const hipError_t he = hipMalloc(&pb.dmem, bytes);
        if (he != hipSuccess) { poison(c, BQC_ERR_DEVICE); return bqc_fail(c, BQC_ERR_DEVICE, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(he)); }
        m.prep.pend = (PendRun*)pb.dmem;
        m.prep.pend_extra_n = (uint32_t*)((char*)pb.dmem + sizeof(PendRun) * (size_t)pb.n());
        m.prep.pend_extra = (PendExtra*)((char*)m.prep.pend_extra_n + 256);
        m.prep.pend_extra_cap = pb.extra_cap;
    }
    // page-locked image: the host pass's tables, and the columns too unless the caller's are page-locked already
    const size_t img_begin = pinned_columns ? m.o_xr : m.h2d_begin, img_bytes = m.h2d_end - img_begin;
    if (s.hcap < img_bytes) {
        if (s.hmem) (void)hipHostFree(s.hmem);
        s.hmem = nullptr; s.hcap = 0;
        const size_t cap = img_bytes + img_bytes / 8 + 4096;
        const hipError_t he = hipHostMalloc((void**)&s.hmem, cap, hipHostMallocDefault);
        if (he != hipSuccess) { poison(c, BQC_ERR_DEVICE); return bqc_fail(c, BQC_ERR_DEVICE, "hipHostMalloc(%zu) failed: %s", cap, hipGetErrorString(he)); }
        s.hcap = cap;
    }
    fill_tables(m, b, H, s.hmem, img_begin, anchored != nullptr);
    char* base = (char*)m.dmem;
    hipError_t he = hipSuccess;
    if (pinned_columns) {
        for (int k = 0; k < 13 && he == hipSuccess; ++k)
            if (m.col_bytes[k]) he = hipMemcpyAsync(base + m.o_col[k], column_ptr(b, k), m.col_bytes[k], hipMemcpyDefault, c->copy_stream); // (page-locked host memory or device memory)
    } else { // stage the columns: the caller may reuse its buffers on return
        struct Piece { char* dst; const char* src; size_t n; };
        Piece pc[13];
        size_t total = 0;
        for (int k = 0; k < 13; ++k) { pc[k] = Piece{s.hmem + (m.o_col[k] - m.h2d_begin), (const char*)column_ptr(b, k), m.col_bytes[k]}; total += m.col_bytes[k]; }
        parallel_ranges(total, std::min(8u, bqc_host_threads()), 16u << 20, [&](unsigned, size_t lo, size_t hi) {
            size_t at = 0;
            for (int k = 0; k < 13; ++k) { // the byte range [lo, hi) of the concatenated columns
                const size_t a = std::max(lo, at), z = std::min(hi, at + pc[k].n);
                if (a < z) memcpy(pc[k].dst + (a - at), pc[k].src + (a - at), z - a);
                at += pc[k].n;
            }
        });
    }
    // (an anchored batch: columns and anchors stay where the caller has them — the image is the host pass's small tables)
    if (anchored && !in_place) { // BQC_ANCHORED_COPY=1: the anchors into their place in the image, the tables in front of and behind it
        const size_t n_cov = sizeof(CovEntry) * (size_t)b->n_reads, o_behind = m.o_cov_in + ((n_cov + 255) & ~(size_t)255);
        if (he == hipSuccess && m.o_cov_in > img_begin) he = hipMemcpyAsync(base + img_begin, s.hmem, m.o_cov_in - img_begin, hipMemcpyHostToDevice, c->copy_stream);
        if (he == hipSuccess && n_cov) he = hipMemcpyAsync(base + m.o_cov_in, anchored->d_cov, n_cov, hipMemcpyDeviceToDevice, c->copy_stream);
        if (he == hipSuccess && m.h2d_end > o_behind) he = hipMemcpyAsync(base + o_behind, s.hmem + (o_behind - img_begin), m.h2d_end - o_behind, hipMemcpyHostToDevice, c->copy_stream);
    } else
    if (he == hipSuccess) he = hipMemcpyAsync(base + img_begin, s.hmem, img_bytes, hipMemcpyHostToDevice, c->copy_stream);
    if (he == hipSuccess) he = hipEventRecord(s.ev_h2d, c->copy_stream);
    if (he == hipSuccess) he = hipStreamWaitEvent(c->stream, s.ev_h2d, 0);
    if (he != hipSuccess) { poison(c, BQC_ERR_DEVICE); return bqc_fail(c, BQC_ERR_DEVICE, "upload failed: %s", hipGetErrorString(he)); }
    if ((rc = enqueue_kernels(c, m))) return poison(c, rc);
    HIPCHK(c, hipMemcpyAsync(s.h_err, m.d_err, sizeof(ErrRec), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(s.ev_done, c->stream));
    s.busy = true;
    s.ticket = ticket;
    c->next_ticket = ticket + 1;
    c->state_seq = ++c->upload_counter;
    if (ticket_out) *ticket_out = ticket;
    if (timing_level() == 2)
        fprintf(stderr, "[timing] submit of %u reads: waited %.4f s for the slot, host pass %.4f s, staging + enqueue %.4f s (%.1f MB to the device)\n", b->n_reads, t_wait,
                t_pass, secs_since(t0) - t_wait - t_pass, (m.h2d_end - m.h2d_begin) / 1e6);
    return 0;
}

// (the staged path copies the columns with the host's memcpy: device-resident payload columns are bqc_submit_async's)
static bool lives_on_device(const void* p)
{
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; } // (ordinary host memory is unknown to the runtime)
    return a.type == hipMemoryTypeDevice;
}
extern "C" int bqc_submit(bqc_ctx* c, const bqc_batch* b)
{
    if (c && b && b->n_reads && (lives_on_device(b->seq) || lives_on_device(b->qual) || lives_on_device(b->cigar)))
        return bqc_fail(c, BQC_ERR_ARG, "bqc_submit: payload columns in device memory need bqc_submit_async");
    return submit_impl(c, b, false, nullptr);
}
extern "C" int bqc_submit_async(bqc_ctx* c, const bqc_batch* b, uint64_t* ticket) { return submit_impl(c, b, true, ticket); }

// ---------------------------------------------------------------------------------------------------
// anchors made on the card (include/bamqc.h: bqc_anchor_*; anchor.h, k_anchor.hip)
// ---------------------------------------------------------------------------------------------------
static int anchor_fail(bqc_ctx* c, const char* what) { c->anchor.err = what; return -BQC_ERR_DEVICE; }
extern "C" const char* bqc_anchor_error(const bqc_ctx* c) { return c ? c->anchor.err.c_str() : ""; }

extern "C" int bqc_anchor_enqueue(bqc_ctx* c, const bqc_batch* b, void* d_cov, void* stream, bqc_anchored** out)
{
    if (!c || !b || !out || (b->n_reads && !d_cov)) return -BQC_ERR_ARG;
    *out = nullptr;
    AnchorEngine& E = c->anchor;
    // one read group, and the whole stream from its first batch on (a shard that starts inside the stream sets its first reads aside on
    // the card as the host's pass would: AnchorState::pending)
    if (c->opt.n_lanes != 1 || (c->shard.tail && c->shard.resolved) || E.mode.load() == 2) return 1;
    if (hipSetDevice(c->device) != hipSuccess) return anchor_fail(c, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = b->n_reads;
    if (!E.d_state) {
        if (hipMalloc((void**)&E.d_state, sizeof(AnchorState)) != hipSuccess || hipMalloc((void**)&E.d_sum, sizeof(AnchorSummary)) != hipSuccess) return anchor_fail(c, "out of device memory");
        AnchorState s0{};
        s0.first = 1;
        s0.pending = c->shard.tail ? 1u : 0u;
        if (hipMemcpy(E.d_state, &s0, sizeof s0, hipMemcpyHostToDevice) != hipSuccess) return anchor_fail(c, "copy failed");
    }
    if (E.cap_n < n) { // scratch: [cpos crid cidx crun](4 B x n) [bound](8 B x n) [bj][runs][blk_a][blk_b]
        if (hipStreamSynchronize(st) != hipSuccess) return anchor_fail(c, "stream failed");
        if (E.d_scratch) (void)hipFree(E.d_scratch);
        E.d_scratch = nullptr; E.cap_n = 0;
        const size_t cap = std::max<size_t>(n + n / 8, 1u << 20);
        const size_t bytes = cap * 24 + 64 + AN_MAX_BREAKS * (4 + sizeof(AnchorRun)) + (cap / 1024 + 4) * (2 * 4 + sizeof(AnchorPart)) + 4096;
        if (hipMalloc(&E.d_scratch, bytes) != hipSuccess) return anchor_fail(c, "out of device memory");
        E.cap_n = cap;
    }
    bqc_anchored* a = nullptr;
    {
        std::lock_guard<std::mutex> lk(E.m);
        if (!E.free_list.empty()) { a = E.free_list.back(); E.free_list.pop_back(); }
    }
    if (!a) {
        a = new bqc_anchored();
        if (hipHostMalloc((void**)&a->h_sum, sizeof(AnchorSummary), hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void**)&a->h_bound, sizeof(uint32_t) * bqc_anchored::kInline, hipHostMallocDefault) != hipSuccess) {
            if (a->h_sum) (void)hipHostFree(a->h_sum);
            delete a;
            return anchor_fail(c, "out of page-locked memory");
        }
        std::lock_guard<std::mutex> lk(E.m);
        E.all.push_back(a);
    }
    a->rest.clear(); a->pend_rid.clear(); a->pend_bp.clear(); a->completed = false; a->n = (uint32_t)n; a->d_cov = (const CovEntry*)d_cov;
    AnchorArgs A{};
    A.n = (uint32_t)n; A.n_refs = c->opt.n_refs; A.n_lanes = 1; A.no_fast = c->no_fast ? 1u : 0u;
    A.flag = b->flag; A.lane = b->lane; A.rid = b->rid; A.pos = b->pos; A.l_seq = b->l_seq; A.n_cigar = b->n_cigar;
    A.main_chrom = c->d_main;
    A.cov_out = (CovEntry*)d_cov; A.state = E.d_state; A.sum = E.d_sum;
    char* q = (char*)E.d_scratch;
    const size_t cap = E.cap_n;
    A.cpos = (uint32_t*)q; q += 4 * cap; A.crid = (int32_t*)q; q += 4 * cap; A.cidx = (uint32_t*)q; q += 4 * cap; A.crun = (uint32_t*)q; q += 4 * cap;
    A.first_of = (uint32_t*)q; q += 8 * cap + 64; A.first_cap = (uint32_t)std::min<size_t>(2 * cap + 16, 0xFFFFFFFFu);
    A.bj = (uint32_t*)q; q += 4 * AN_MAX_BREAKS; A.runs = (AnchorRun*)q; q += sizeof(AnchorRun) * AN_MAX_BREAKS;
    q = (char*)(((uintptr_t)q + 255) & ~(uintptr_t)255);
    A.blk_a = (uint32_t*)q; q += 4 * (cap / 1024 + 4); A.blk_b = (uint32_t*)q; q += 4 * (cap / 1024 + 4);
    q = (char*)(((uintptr_t)q + 255) & ~(uintptr_t)255);
    A.parts = (AnchorPart*)q;
    E.d_bound = A.first_of;
    const size_t n_first = std::min<size_t>(2 * n + 16, A.first_cap); // (windows a batch of n reads can reach)
    if (hipMemsetAsync(A.first_of, 0xFF, 4 * n_first, st) != hipSuccess) { std::lock_guard<std::mutex> lk(E.m); E.free_list.push_back(a); return anchor_fail(c, "memset failed"); }
    bqc_launch_anchor(A, st);
    if (hipMemcpyAsync(a->h_sum, E.d_sum, sizeof(AnchorSummary), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(a->h_bound, A.first_of, sizeof(uint32_t) * std::min<size_t>(bqc_anchored::kInline, n_first), hipMemcpyDeviceToHost, st) != hipSuccess) {
        std::lock_guard<std::mutex> lk(E.m);
        E.free_list.push_back(a);
        return anchor_fail(c, "copy failed");
    }
    E.mode = 1;
    *out = a;
    return 0;
}

extern "C" int bqc_anchor_complete(bqc_ctx* c, bqc_anchored* a, bqc_anchor_info* info)
{
    if (!c || !a) return -BQC_ERR_ARG;
    AnchorEngine& E = c->anchor;
    const AnchorSummary& S = *a->h_sum;
    if (info) { info->n_noqual = S.n_noqual; info->rid_min = S.rid_min; info->rid_max = S.rid_max; }
    if (S.flags & AN_FLAG_BOUND_OVERFLOW) return anchor_fail(c, "internal error: boundary list overflow");
    if (S.flags & AN_FLAG_TOO_MANY_BREAKS) { // the card has left its state alone: this batch and what follows are the host's
        E.mode = 2;
        std::lock_guard<std::mutex> lk(E.m);
        E.free_list.push_back(a);
        return 1;
    }
    if (S.n_cand > S.n_pending && (size_t)S.last_rel + 1 > bqc_anchored::kInline) { // (sparse data: the rest of the table, before the next batch's kernels reuse the buffer)
        a->rest.resize((size_t)S.last_rel + 1 - bqc_anchored::kInline);
        if (hipSetDevice(c->device) != hipSuccess ||
            hipMemcpy(a->rest.data(), E.d_bound + bqc_anchored::kInline, sizeof(uint32_t) * a->rest.size(), hipMemcpyDeviceToHost) != hipSuccess) return anchor_fail(c, "copy failed");
    }
    if (S.n_pending) { // the reads set aside: chromosome and position of the batch's first n_pending candidates (the scratch's crid / cpos)
        a->pend_rid.resize(S.n_pending); a->pend_bp.resize(S.n_pending);
        const char* q = (const char*)E.d_scratch;
        if (hipSetDevice(c->device) != hipSuccess || hipMemcpy(a->pend_bp.data(), q, 4ull * S.n_pending, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(a->pend_rid.data(), q + 4 * E.cap_n, 4ull * S.n_pending, hipMemcpyDeviceToHost) != hipSuccess) return anchor_fail(c, "copy failed");
    }
    a->completed = true;
    return 0;
}

extern "C" void bqc_anchor_discard(bqc_ctx* c, bqc_anchored* a)
{
    if (!c || !a) return;
    std::lock_guard<std::mutex> lk(c->anchor.m);
    c->anchor.free_list.push_back(a);
}

extern "C" int bqc_submit_anchored(bqc_ctx* c, const bqc_batch* b, bqc_anchored* a, uint64_t* ticket)
{
    if (!c || !b || !a || !a->completed || a->n != b->n_reads) return bqc_fail(c, BQC_ERR_ARG, "bqc_submit_anchored: bad argument");
    if (b->n_nm_extra) return bqc_fail(c, BQC_ERR_ARG, "bqc_submit_anchored: further NM values belong to batches decoded on the host");
    const int rc = submit_impl(c, b, true, ticket, a);
    bqc_anchor_discard(c, a);
    return rc;
}

void bqc_anchor_destroy(bqc_ctx* c)
{
    AnchorEngine& E = c->anchor;
    for (bqc_anchored* a : E.all) { if (a->h_sum) (void)hipHostFree(a->h_sum); if (a->h_bound) (void)hipHostFree(a->h_bound); delete a; }
    E.all.clear(); E.free_list.clear();
    if (E.d_state) (void)hipFree(E.d_state);
    if (E.d_sum) (void)hipFree(E.d_sum);
    if (E.d_scratch) (void)hipFree(E.d_scratch);
    E.d_state = nullptr; E.d_sum = nullptr; E.d_scratch = nullptr; E.cap_n = 0;
}

extern "C" int bqc_batch_uploaded(bqc_ctx* c, uint64_t ticket, int wait)
{
    if (!c) return -BQC_ERR_ARG;
    if (ticket == 0 || ticket >= c->next_ticket) return 1;
    Slot& s = c->slots[ticket % bqc_ctx::kSlots];
    if (!s.busy || s.ticket != ticket) return 1; // retired (or overwritten by a later batch, which waited for it)
    // (a batch whose columns are read in place — bqc_submit_anchored — is done with them when its kernels are)
    hipEvent_t ev = s.in_place ? s.ev_done : s.ev_h2d;
    if (wait) return hipEventSynchronize(ev) == hipSuccess ? 1 : -BQC_ERR_DEVICE;
    const hipError_t q = hipEventQuery(ev);
    return q == hipSuccess ? 1 : q == hipErrorNotReady ? 0 : -BQC_ERR_DEVICE;
}

extern "C" int bqc_host_register(void* p, uint64_t bytes)
{
    if (!p || !bytes) return BQC_ERR_ARG;
    return hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault) == hipSuccess ? 0 : BQC_ERR_DEVICE;
}
extern "C" int bqc_host_unregister(void* p) { return p && hipHostUnregister(p) == hipSuccess ? 0 : BQC_ERR_DEVICE; }

// ---------------------------------------------------------------------------------------------------
// shards of one record stream (multi-GPU): hand-over of the coverage state machine
// ---------------------------------------------------------------------------------------------------
// exported block: int32 first / last FASTA position of the shard's triplet-eligible reads, then per read group
// {first, started, pad, pad, id, shift} (12 bytes) and the 2000 depths of its two live windows
namespace {
struct LaneWire { uint8_t first, started, pad0, pad1; int32_t id, shift; };
}
static size_t shard_bytes(const bqc_ctx* c) { return 8 + (size_t)c->opt.n_lanes * (sizeof(LaneWire) + 2 * BQC_VSIZE * 4); }
extern "C" uint64_t bqc_shard_state_bytes(const bqc_ctx* c) { return c ? shard_bytes(c) : 0; }

extern "C" int bqc_shard_fasta_span(bqc_ctx* c, int32_t span[2])
{
    if (!c || !span) return BQC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = c->poisoned ? 0 : bqc_sync(c);
    if (rc) return rc;
    int32_t cur[2];
    HIPCHK(c, hipMemcpy(cur, c->d_cursor, 8, hipMemcpyDeviceToHost));
    span[0] = cur[1]; span[1] = cur[0];
    return 0;
}

extern "C" int bqc_shard_export(bqc_ctx* c, void* out)
{
    if (!c || !out) return bqc_fail(c, BQC_ERR_ARG, "bqc_shard_export: null argument");
    if (c->poisoned) return bqc_fail(c, BQC_ERR_STATE, "context is in an error state: %s", c->err.c_str());
    if (c->shard.tail && !c->shard.resolved) return bqc_fail(c, BQC_ERR_STATE, "bqc_shard_export before bqc_shard_resolve");
    if (c->flushed && !c->shard.exported) return bqc_fail(c, BQC_ERR_STATE, "bqc_shard_export after bqc_flush");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = bqc_sync(c);
    if (rc) return rc;
    const uint32_t nl = c->opt.n_lanes;
    std::vector<uint32_t> parity(nl + 1);
    std::vector<uint8_t> started(nl);
    int32_t cur[2];
    HIPCHK(c, hipMemcpy(parity.data(), c->d_parity, 4 * (size_t)nl, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(started.data(), c->d_started, nl, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(cur, c->d_cursor, 8, hipMemcpyDeviceToHost));
    char* w = (char*)out;
    const int32_t head[2] = {cur[1], cur[0]}; // first, last FASTA position of the eligible reads (-1: none)
    memcpy(w, head, 8); w += 8;
    for (uint32_t l = 0; l < nl; ++l) {
        const LaneCov& s = c->cov[l];
        const LaneWire lw{(uint8_t)s.first, started[l], 0, 0, s.id, s.shift};
        memcpy(w, &lw, sizeof lw); w += sizeof lw;
        HIPCHK(c, hipMemcpy(w, c->d_carry + ((size_t)l * 2 + (parity[l] & 1u)) * 2 * BQC_VSIZE, 2 * BQC_VSIZE * 4, hipMemcpyDeviceToHost));
        w += 2 * BQC_VSIZE * 4;
    }
    c->shard.exported = true;
    c->flushed = true; // the live windows are the successor's now: this state vector holds complete windows only
    return 0;
}

extern "C" int bqc_shard_resolve(bqc_ctx* c, const void* pred)
{
    if (!c || !pred) return bqc_fail(c, BQC_ERR_ARG, "bqc_shard_resolve: null argument");
    if (c->poisoned) return bqc_fail(c, BQC_ERR_STATE, "context is in an error state: %s", c->err.c_str());
    ShardCtx& sh = c->shard;
    if (!sh.tail || sh.resolved) return bqc_fail(c, BQC_ERR_STATE, "bqc_shard_resolve: not a shard_tail context, or resolved already");
    if (c->flushed) return bqc_fail(c, BQC_ERR_STATE, "bqc_shard_resolve after bqc_flush");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = bqc_sync(c); // every batch of the shard is in: the pending logs are complete
    if (rc) return rc;
    const uint32_t nl = c->opt.n_lanes;
    const size_t carry_lane = 2 * 2 * BQC_VSIZE; // words per read group in d_carry: [2 halves][2000]
    // the shard's own ("main") trajectories aside; the predecessor's final state in their place
    const std::vector<LaneCov> main_cov = c->cov;
    uint32_t* d_save = nullptr; // carry | parity | started of the main trajectories
    const size_t save_bytes = carry_lane * 4 * nl + 4 * ((size_t)nl + 1) + nl;
    HIPCHK(c, hipMalloc(&d_save, save_bytes + 256));
    uint32_t* d_save_par = d_save + carry_lane * nl;
    uint8_t* d_save_started = (uint8_t*)(d_save_par + nl + 1);
    HIPCHK(c, hipMemcpyAsync(d_save, c->d_carry, carry_lane * 4 * nl, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_save_par, c->d_parity, 4 * ((size_t)nl + 1), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_save_started, c->d_started, nl, hipMemcpyDeviceToDevice, c->stream));
    const char* r = (const char*)pred + 8;
    std::vector<uint32_t> carry_in(carry_lane * nl, 0);
    std::vector<uint8_t> started_in(nl);
    for (uint32_t l = 0; l < nl; ++l) {
        LaneWire lw;
        memcpy(&lw, r, sizeof lw); r += sizeof lw;
        LaneCov s;
        s.first = lw.first != 0; s.id = lw.id; s.shift = lw.shift; s.win = 0; s.batch_base = 0;
        c->cov[l] = s;
        started_in[l] = lw.started;
        memcpy(carry_in.data() + carry_lane * l, r, 2 * BQC_VSIZE * 4); r += 2 * BQC_VSIZE * 4; // live half -> half 0
    }
    HIPCHK(c, hipMemcpyAsync(c->d_carry, carry_in.data(), carry_lane * 4 * nl, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_parity, 0, 4 * ((size_t)nl + 1), c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_started, started_in.data(), nl, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // the reads set aside, batch by batch, through the same planner and k_cov
    HostPass& H = c->hp;
    void* d_work = nullptr;
    size_t work_cap = 0;
    for (PendBatch& pb : sh.batches) {
        const uint32_t n = pb.n();
        H.n = n; H.multi_lane = nl > 1;
        H.cov.resize(n);
        CovPlanner plan(c, H, n);
        for (uint32_t i = 0; i < n; ++i) {
            uint64_t rel;
            uint32_t pos;
            plan.step(i, pb.lane[i], pb.rid[i], pb.bp[i], rel, pos);
            if (rel > 0xFFFFFFF0ull) { (void)hipFree(d_save); (void)hipFree(d_work); return poison(c, bqc_fail(c, BQC_ERR_ARG, "shard spans too many coverage windows")); }
            H.cov[i] = CovEntry{(uint32_t)rel, pos};
        }
        plan.finish();
        Carver cv;
        const size_t o_cin = cv.take(sizeof(CovEntry) * (size_t)n), o_lane = cv.take(n), o_tiles = cv.take(sizeof(CovTile) * H.tiles.size()), o_mask = cv.take(nl),
                     o_started = cv.take(nl), o_aidx = cv.take(8 * H.add_idx.size()), o_aval = cv.take(8 * H.add_val.size()), o_h2d_end = cv.off,
                     o_cov = cv.take(sizeof(CovEntry) * (size_t)n), o_covx = cv.take(sizeof(CovExtra) * ((size_t)pb.extra_cap + 1)), o_desc = cv.take(sizeof(BatchDesc));
        if (work_cap < cv.off) {
            if (d_work) (void)hipFree(d_work);
            d_work = nullptr;
            const hipError_t he = hipMalloc(&d_work, cv.off + cv.off / 4);
            if (he != hipSuccess) { (void)hipFree(d_save); return poison(c, bqc_fail(c, BQC_ERR_DEVICE, "hipMalloc failed: %s", hipGetErrorString(he))); }
            work_cap = cv.off + cv.off / 4;
        }
        std::vector<char> img(o_h2d_end);
        memcpy(img.data() + o_cin, H.cov.data(), sizeof(CovEntry) * (size_t)n);
        memcpy(img.data() + o_lane, pb.lane.data(), n);
        if (!H.tiles.empty()) memcpy(img.data() + o_tiles, H.tiles.data(), sizeof(CovTile) * H.tiles.size());
        memcpy(img.data() + o_mask, H.lane_mask.data(), nl);
        memcpy(img.data() + o_started, H.started_after.data(), nl);
        if (!H.add_idx.empty()) { memcpy(img.data() + o_aidx, H.add_idx.data(), 8 * H.add_idx.size()); memcpy(img.data() + o_aval, H.add_val.data(), 8 * H.add_val.size()); }
        char* base = (char*)d_work;
        HIPCHK(c, hipMemcpyAsync(base, img.data(), o_h2d_end, hipMemcpyHostToDevice, c->stream));
        const PendRun* pend = (const PendRun*)pb.dmem;
        const uint32_t* extra_n = (const uint32_t*)((const char*)pb.dmem + sizeof(PendRun) * (size_t)n);
        const PendExtra* extra = (const PendExtra*)((const char*)extra_n + 256);
        bqc_launch_pend_cov(n, (const CovEntry*)(base + o_cin), pend, extra, extra_n, pb.extra_cap, (const uint8_t*)(base + o_lane), (CovEntry*)(base + o_cov),
                            (CovExtra*)(base + o_covx), (BatchDesc*)(base + o_desc), c->stream);
        DevBatch d{};
        d.n_reads = n; d.lane = (const uint8_t*)(base + o_lane); d.cov = (const CovEntry*)(base + o_cov); d.cov_extra = (const CovExtra*)(base + o_covx);
        d.desc = (const BatchDesc*)(base + o_desc); d.cov_tiles = (const CovTile*)(base + o_tiles); d.n_cov_tiles = (uint32_t)H.tiles.size();
        if (!d.n_cov_tiles) bqc_launch_or_bytes(c->d_started, (const uint8_t*)(base + o_started), nl, c->stream);
        else bqc_launch_cov(d, c->sl, c->d_state, c->d_carry, c->d_parity, (const uint8_t*)(base + o_mask), c->d_started, (const uint8_t*)(base + o_started), nl, c->stream);
        bqc_launch_add_words(c->d_state, (const uint64_t*)(base + o_aidx), (const uint64_t*)(base + o_aval), (uint32_t)H.add_idx.size(), c->stream);
        HIPCHK(c, hipStreamSynchronize(c->stream)); // (img and the tables are reused by the next batch)
        (void)hipFree(pb.dmem);
        pb.dmem = nullptr;
    }
    sh.batches.clear();
    // Read groups that met a certain reset in this shard: the trajectory that came in ends there — the reset flushes its two live
    // windows (bamqualcheck's update_coverage on both) — and the read group goes on with its main trajectory.  The others (every
    // read set aside, or none at all) simply continue from the state that came in.
    std::vector<uint8_t> sel(nl, 0);
    bool any = false;
    for (uint32_t l = 0; l < nl; ++l) { sel[l] = sh.pending[l] ? 0 : 1; any |= sel[l] != 0; }
    if (any) {
        uint8_t* d_sel = (uint8_t*)d_save + save_bytes; // (256 spare bytes behind the saved arrays; nl <= 256)
        HIPCHK(c, hipMemcpyAsync(d_sel, sel.data(), nl, hipMemcpyHostToDevice, c->stream));
        bqc_launch_cov_final(c->sl, c->d_state, c->d_carry, c->d_parity, c->d_started, d_sel, 0, c->stream);
        for (uint32_t l = 0; l < nl; ++l)
            if (sel[l]) {
                c->cov[l] = main_cov[l];
                HIPCHK(c, hipMemcpyAsync(c->d_carry + carry_lane * l, d_save + carry_lane * l, carry_lane * 4, hipMemcpyDeviceToDevice, c->stream));
                HIPCHK(c, hipMemcpyAsync(c->d_parity + l, d_save_par + l, 4, hipMemcpyDeviceToDevice, c->stream));
                HIPCHK(c, hipMemcpyAsync(c->d_started + l, d_save_started + l, 1, hipMemcpyDeviceToDevice, c->stream));
            }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(d_save);
    if (d_work) (void)hipFree(d_work);
    for (uint32_t l = 0; l < nl; ++l) sh.pending[l] = 0;
    sh.resolved = true;
    return 0;
}
