// bqc_ctx.h — the context behind include/bamqc.h, shared by bqc_api.cpp (life cycle, state vector, finalisation) and
// bqc_pipeline.cpp (host pass, staging ring, uploads and launches).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bamqc.h"
#include "device_types.h"
#include "prep.h"
#include "anchor.h"
#include "sketch.h"
#include "../host/raw_vector.h"

struct LaneCov { // host side of OverallNumbers' window state machine (OverallNumbers.hpp:84-110)
    bool first = true;
    int32_t id = 0;
    int32_t shift = 0;
    uint64_t win = 0;        // absolute index (flush order) of the window currently held in v1
    uint64_t batch_base = 0; // absolute window index that is batch-relative window 0 (= carry windows 0,1)
};

// What the host knows about a batch after its pass over the fixed columns (bqc_pipeline.cpp: host_pass).
struct HostPass {
    uint32_t n = 0;
    uint64_t seq_bytes = 0, qual_bytes = 0, cigar_words = 0;
    uint32_t n_slow = 0, max_len_slow = 0;  // reads for the generic kernels, the longest of them
    uint32_t t8_lane = 0;                   // read group with the most reads (k_long: its per-cycle tiles go through the scratch array)
    bool multi_lane = false;
    raw_vector<CovEntry> cov;               // [n] anchors {win, pos} / BQC_COV_NONE
    raw_vector<uint32_t> order;             // [n] reads grouped by read group (only when multi_lane)
    std::vector<SuperWindow> sws;
    std::vector<Stretch> stretches;
    std::vector<CovTile> tiles;
    std::vector<uint8_t> lane_mask, started_after;
    std::vector<uint64_t> add_idx, add_val; // host-computed additions (zero-depth windows)
    uint32_t n_pending = 0;                 // shard mode: reads marked BQC_COV_PENDING in `cov` (their log: ShardCtx::batches.back())
    // scratch kept from batch to batch
    std::vector<std::vector<uint32_t>> lane_first; // per read group: first read index whose window is >= k
    std::vector<uint64_t> lane_count;
};

// A batch in device memory: one allocation, carved into the uploaded columns, the host pass's tables and the outputs of the
// device pre-pass.
struct BatchMem {
    void* dmem = nullptr;
    size_t dcap = 0;
    DevBatch d{};
    PrepArgs prep{};
    uint8_t* d_lane_mask = nullptr;
    uint8_t* d_started_after = nullptr;
    uint64_t* d_add_idx = nullptr;
    uint64_t* d_add_val = nullptr;
    uint32_t n_add = 0;
    uint32_t* d_rsum = nullptr;          // [n_reads][3] per-read sums of the long-read kernel
    ErrRec* d_err = nullptr;
    uint64_t algo_bytes = 0;
    uint32_t n_slow = 0, max_len_slow = 0, n_chunks_slow_ub = 0, t8_lane = 0;
    uint64_t lane_bits[4] = {0, 0, 0, 0}; // read groups the batch holds (their 8-mer counts may be in the scratch rows after its kernels)
    bool processed = false;
    // where the pieces of the host-side image go (offsets into dmem; the staged image has the same layout from h2d_begin on)
    size_t h2d_begin = 0, h2d_end = 0;   // the contiguous part that is copied from the staging image
    size_t o_col[13] = {0};              // flag mapq lane rid pos tlen nm as l_seq n_cigar seq qual cigar
    size_t col_bytes[13] = {0};
    size_t o_xr = 0, o_xv = 0, o_cov_in = 0, o_order = 0, o_sws = 0, o_stretch = 0, o_tiles = 0, o_mask = 0, o_started = 0, o_aidx = 0, o_aval = 0;
};

// Shard mode (bqc_options.shard_tail): the coverage reads set aside until the predecessor shard's final state is known.
struct PendBatch {                 // those of one batch, in stream order
    std::vector<uint8_t> lane;
    std::vector<int32_t> rid;
    std::vector<uint32_t> bp;
    void* dmem = nullptr;          // device: PendRun[n] | counter | PendExtra[extra_cap]
    uint32_t extra_cap = 0;
    uint32_t n() const { return (uint32_t)lane.size(); }
};
struct ShardCtx {
    bool tail = false, resolved = false, exported = false;
    std::vector<uint8_t> pending;  // [lane] 1: the read group's reads are still being set aside
    std::vector<uint8_t> has_prev; // [lane] the read before (of those that enter coverage): for the test "resets whatever the state"
    std::vector<int32_t> prev_rid;
    std::vector<uint32_t> prev_bp;
    std::deque<PendBatch> batches;
};

struct bqc_dbatch {
    BatchMem m;
    // host stream state after this batch (restored by bqc_process after a bqc_reset, see there)
    std::vector<LaneCov> cov_after;
    uint64_t seq = 0;
};

struct Slot { // one batch in flight through bqc_submit / bqc_submit_async
    BatchMem m;
    char* hmem = nullptr;     // page-locked staging image (host pass tables; the columns too on the staging path)
    size_t hcap = 0;
    ErrRec* h_err = nullptr;  // page-locked copy of the batch's error record (written by a D2H copy behind its last kernel)
    hipEvent_t ev_h2d = nullptr, ev_done = nullptr;
    bool busy = false;
    bool in_place = false;    // the batch's columns are read where the caller has them (bqc_submit_anchored): bqc_batch_uploaded waits for ev_done
    uint64_t ticket = 0;
};

// Anchors made on the card (anchor.h, k_anchor.hip) for batches whose columns live in device memory: bqc_anchor_enqueue /
// bqc_anchor_complete run in the thread that decodes the batches, bqc_submit_anchored in the one that submits them.
struct bqc_anchored {
    AnchorSummary* h_sum = nullptr;   // page-locked: the batch's summary ...
    uint32_t* h_bound = nullptr;      // ... and the first kInline entries of first_of[] (anchor.h)
    std::vector<uint32_t> rest;       // the entries behind them (sparse data), fetched by bqc_anchor_complete
    std::vector<int32_t> pend_rid;    // a shard_tail context: the reads set aside (the batch's first n_pending candidates) ...
    std::vector<uint32_t> pend_bp;    // ... chromosome and beginPos, for the pending log (bqc_shard_resolve)
    const CovEntry* d_cov = nullptr;  // the caller's device buffer with the anchors of the batch's reads
    uint32_t n = 0;
    bool completed = false;
    static const uint32_t kInline = 1u << 17;
};
struct AnchorEngine {
    std::atomic<int> mode{0};         // 0: not used yet, 1: the card keeps the state, 2: off for the rest of the stream (the host keeps it)
    AnchorState* d_state = nullptr;
    AnchorSummary* d_sum = nullptr;
    uint32_t* d_bound = nullptr;
    void* d_scratch = nullptr;
    size_t cap_n = 0;                 // reads the scratch buffers are sized for
    std::mutex m;                     // the free list (handles come back from the submitting thread)
    std::vector<bqc_anchored*> free_list, all;
    std::string err;
};

struct bqc_ctx {
    bqc_options opt{};
    std::vector<uint8_t> main_chrom;
    std::vector<int32_t> fasta_index;
    StateLayout sl{};
    int device = 0;
    hipStream_t stream = nullptr;      // compute: every kernel of the context, in batch order
    hipStream_t copy_stream = nullptr; // host-to-device copies of the batches in flight
    uint32_t n_cu = 256;
    uint64_t* d_state = nullptr;
    ErrRec* d_err0 = nullptr;          // error record of work outside a batch (final flush)
    int32_t* d_cursor = nullptr;       // FASTA cursor of the stream (TripletCounting.hpp:254-259), device resident
    int32_t* d_fasta_index = nullptr;
    // 8-mer scratch rows of k_short: every workgroup of a launch owns a slot of BQC_T8_SPW rows (64 KiB images of its packed LDS
    // counters, written with plain stores); d_t8used[slot] = rows written.  The slots of up to kT8Launches launches pile up
    // and are summed into d_state by fold_t8: before the state is read, when the table is full, or when another read group
    // needs it.  (kT8Slots * BQC_T8_SPW * 64 KiB = 512 MiB of the 288 GB.)
    uint32_t* d_t8rows = nullptr;
    uint32_t* d_t8used = nullptr;
    // per-cycle counter tiles of k_long's workgroups (48 KiB each; d_kl_cyc_used[wg] = written), summed by k_long_cyc_fold in the same launch
    uint32_t* d_kl_cyc = nullptr;
    uint32_t* d_kl_cyc_used = nullptr;
    uint32_t t8_slots_used = 0, t8_slots_cap = 0;
    uint64_t t8_lanes[4] = {0, 0, 0, 0}; // read groups (bit per lane) of the batches whose rows are in the table
    std::vector<std::pair<void*, size_t>> pool; // device buffers of freed batches, reused by bqc_upload (hipMalloc / hipFree cost milliseconds)
    uint32_t* d_carry = nullptr;  // [lane][2][2000]
    uint32_t* d_parity = nullptr; // [lane], then the count of finished workgroups of the running k_cov
    uint8_t* d_started = nullptr; // [lane]
    // references
    std::vector<uint8_t*> d_ref;
    // bqc_reserve_references: one allocation the contigs are carved from (an allocation behind a running kernel waits for it: a
    // caller that uploads contigs while batches run reserves first); pointers inside it are not released one by one
    uint8_t* ref_arena = nullptr;
    size_t ref_arena_cap = 0, ref_arena_used = 0;
    std::vector<uint32_t*> d_refn; // one-hot nibble copy for the short-read fast path
    std::vector<uint64_t> ref_len;
    uint8_t** d_ref_ptrs = nullptr;
    uint32_t** d_refn_ptrs = nullptr;
    bool no_fast = false;          // BQC_NO_FAST=1: every read takes the generic kernel
    uint64_t* d_ref_len = nullptr;
    uint8_t* d_main = nullptr;
    // coverage host state
    std::vector<LaneCov> cov;
    bool flushed = false;
    bool poisoned = false;
    int poison_code = 0;
    uint64_t upload_counter = 0; // number of batches passed so far
    uint64_t state_seq = 0;      // sequence number of the batch the host stream state (cov) reflects
    HostPass hp;                 // host pass of the batch being submitted (its vectors are reused)
    ShardCtx shard;
    // batches in flight
    static const int kSlots = 3;
    Slot slots[kSlots];
    uint64_t next_ticket = 1;    // ticket of the next submitted batch; slot = ticket % kSlots
    uint64_t checked_ticket = 0; // every batch up to here has been waited for and its error record read
    AnchorEngine anchor;
    // sketch (N1)
    SketchDevice* sketch = nullptr;
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev;
    std::vector<const char*> tnames;
    std::vector<float> tms;
    int n_timed = 0;
    // finalize output
    std::vector<uint64_t> h_state;
    std::vector<std::vector<uint64_t>> arrays;
    std::vector<bqc_lane_counts> lanes;
    std::vector<std::vector<bqc_sketch_counts>> sk_out;
    bqc_counts counts{};
    std::string err;
};

int bqc_fail(bqc_ctx* c, int code, const char* fmt, ...);
#define HIPCHK(c, call)                                                                                         \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess) return bqc_fail(c, BQC_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// bqc_api.cpp: a non-blocking stream on `device` — one of those bqc_warmup has made ahead, or a new one (nullptr: creation failed)
hipStream_t bqc_pool_stream(int device, int rank);
// the context's copy stream, taken when the first batch needs it (0: ok)
int bqc_copy_stream(bqc_ctx* c);
// bqc_pipeline.cpp
int bqc_report_errors(bqc_ctx* c, const ErrRec& e); // what the device found wrong with a batch -> error code + message (0: nothing)
int bqc_drain(bqc_ctx* c);                 // wait for every batch in flight; returns the first error of the stream (context poisoned)
void bqc_pipeline_destroy(bqc_ctx* c);     // frees slots and pooled buffers
void bqc_anchor_destroy(bqc_ctx* c);       // frees the anchor engine's buffers and handles
// Every reader of d_state goes through this: the packed 8-mer rows of k_short are summed into the state vector first.
void bqc_state_ready(bqc_ctx* c);
