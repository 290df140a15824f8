// bqc_api.cpp — host side of libbamqc_gpu.so: context, host pre-pass, device batches, launches,
// finalisation.  Implements include/bamqc.h.  There is NO CPU fallback in this library: without a
// working HIP device bqc_create fails with BQC_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <vector>

#include "../../include/bamqc.h"
#include "device_types.h"
#include "../host/parallel.h"
#include "../host/raw_vector.h"
#include "sketch.h"

extern "C" {
void bqc_launch_reads_chunks(const DevBatch&, const StateLayout&, uint64_t*, const DevRefs&, uint32_t*, uint32_t n_cu, hipStream_t);
void bqc_launch_nm_extra(const DevBatch&, const StateLayout&, uint64_t*, const DevRefs&, uint32_t*, hipStream_t);
void bqc_launch_long(const DevBatch&, const StateLayout&, uint64_t*, const DevRefs&, uint32_t*, uint32_t* rsum, uint32_t max_len, uint32_t n_cu, hipStream_t);
void bqc_launch_cov(const DevBatch&, const StateLayout&, uint64_t*, uint32_t* carry, uint32_t* parity, const uint8_t* lane_mask, uint8_t* started,
                    const uint8_t* started_after, uint32_t n_lanes, hipStream_t);
void bqc_launch_cov_final(const StateLayout&, uint64_t*, const uint32_t* carry, const uint32_t* parity, const uint8_t* started, hipStream_t);
void bqc_launch_add_words(uint64_t* state, const uint64_t* idx, const uint64_t* val, uint32_t n, hipStream_t);
void bqc_launch_or_bytes(uint8_t* dst, const uint8_t* src, uint32_t n, hipStream_t);
void bqc_launch_short(const DevBatch&, const StateLayout&, uint64_t*, const DevRefs&, uint32_t*, uint32_t grid, uint32_t* t8rows, uint32_t* t8_used, uint32_t t8_lane, hipStream_t);
void bqc_launch_ref_nibbles(const uint8_t* dna5, uint64_t len, uint32_t* out, uint64_t n_dwords, hipStream_t);
hipError_t bqc_long_init();
hipError_t bqc_short_init();
uint32_t bqc_short_parts();
void bqc_launch_t8_fold(const uint32_t* t8rows, const uint32_t* t8_used, uint32_t n_slots, const StateLayout&, uint64_t* state, uint32_t lane, hipStream_t);
}

static thread_local char g_create_err[512];

struct LaneCov { // host side of OverallNumbers' window state machine (OverallNumbers.hpp:84-110)
    bool first = true;
    int32_t id = 0;
    int32_t shift = 0;
    uint64_t win = 0;        // absolute index (flush order) of the window currently held in v1
    uint64_t batch_base = 0; // absolute window index that is batch-relative window 0 (= carry windows 0,1)
};

struct bqc_dbatch {
    void* dmem = nullptr;
    size_t dbytes = 0, dcap = 0;
    DevBatch d{};
    uint8_t* d_lane_mask = nullptr; // [n_lanes] lanes that own coverage tiles in this batch
    uint64_t algo_bytes = 0;
    std::vector<uint64_t> add_idx, add_val; // host-computed additions (zero-depth windows)
    uint64_t* d_add_idx = nullptr;
    uint64_t* d_add_val = nullptr;
    // host stream state after this batch (restored by bqc_process after a bqc_reset, see there)
    uint8_t* d_started_after = nullptr;
    std::vector<LaneCov> cov_after;
    int32_t fasta_cursor_after = -1;
    uint64_t seq = 0;
    uint32_t* d_rsum = nullptr; // [n_reads][3] per-read sums of the long-read kernel (present when the batch has generic chunks)
    uint32_t long_max_len = 0;
    uint32_t t8_lane = 0; // read group with the most fast chunks: its 8-mer counts go through the scratch rows
};

struct bqc_ctx {
    bqc_options opt{};
    std::vector<uint8_t> main_chrom;
    std::vector<int32_t> fasta_index;
    StateLayout sl{};
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t n_cu = 256;
    uint64_t* d_state = nullptr;
    uint32_t* d_err = nullptr;
    // 8-mer scratch rows of k_short: every workgroup of a launch owns a slot of BQC_T8_SPW rows (64 KiB images of its packed LDS
    // counters, written with plain stores); d_t8used[slot] = rows written.  The slots of up to kT8Launches launches pile up
    // and are summed into d_state by fold_t8: before the state is read, when the table is full, or when another read group
    // needs it.  (kT8Slots * BQC_T8_SPW * 64 KiB = 512 MiB of the 288 GB.)
    uint32_t* d_t8rows = nullptr;
    uint32_t* d_t8used = nullptr;
    uint32_t t8_slots_used = 0, t8_slots_cap = 0;
    uint32_t t8_rows_lane = 0;
    std::vector<std::pair<void*, size_t>> pool; // device buffers of freed batches, reused by bqc_upload (hipMalloc / hipFree cost milliseconds)
    uint32_t* d_carry = nullptr;  // [lane][2][2000]
    void* prep_cache = nullptr;   // Prep of the last upload: its vectors are reused (bqc_upload)
    uint32_t* d_parity = nullptr; // [lane], then the count of finished workgroups of the running k_cov
    uint8_t* d_started = nullptr; // [lane]
    // references
    std::vector<uint8_t*> d_ref;
    std::vector<uint32_t*> d_refn; // one-hot nibble copy for the short-read fast path
    std::vector<uint64_t> ref_len;
    uint8_t** d_ref_ptrs = nullptr;
    uint32_t** d_refn_ptrs = nullptr;
    bool no_fast = false;          // BQC_NO_FAST=1: every read takes the generic kernel
    uint64_t* d_ref_len = nullptr;
    uint8_t* d_main = nullptr;
    // coverage / genome host state
    std::vector<LaneCov> cov;
    int32_t fasta_cursor = -1;
    bool flushed = false;
    bool poisoned = false;
    uint64_t upload_counter = 0; // number of batches pre-passed so far
    uint64_t state_seq = 0;      // sequence number of the batch the host stream state (cov, fasta_cursor) reflects
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

static int fail(bqc_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else snprintf(g_create_err, sizeof g_create_err, "%s", buf);
    return code;
}
#define HIPCHK(c, call)                                                                                     \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) return fail(c, BQC_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

extern "C" int bqc_abi_version(void) { return BQC_ABI_VERSION; }
extern "C" const char* bqc_last_error(const bqc_ctx* c) { return c ? c->err.c_str() : g_create_err; }

static int upload_ref_tables(bqc_ctx* c)
{
    HIPCHK(c, hipMemcpyAsync(c->d_ref_ptrs, c->d_ref.data(), sizeof(uint8_t*) * c->d_ref.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_ref_len, c->ref_len.data(), sizeof(uint64_t) * c->ref_len.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_refn_ptrs, c->d_refn.data(), sizeof(uint32_t*) * c->d_refn.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int bqc_create(const bqc_options* opt, bqc_ctx** out)
{
    if (!opt || !out) return fail(nullptr, BQC_ERR_ARG, "bqc_create: null argument");
    if (opt->struct_size != sizeof(bqc_options)) return fail(nullptr, BQC_ERR_ARG, "bqc_create: struct_size mismatch");
    if (opt->n_lanes == 0 || opt->n_lanes > 256) return fail(nullptr, BQC_ERR_ARG, "bqc_create: n_lanes must be 1..256");
    if (opt->isize < 0) return fail(nullptr, BQC_ERR_ARG, "bqc_create: negative insert size");
    if (opt->max_read_len == 0 || opt->hist_cap == 0) return fail(nullptr, BQC_ERR_ARG, "bqc_create: zero capacity");
    if (opt->n_refs && !opt->main_chrom) return fail(nullptr, BQC_ERR_ARG, "bqc_create: main_chrom missing");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, BQC_ERR_DEVICE, "bqc_create: no HIP device available (this library has no CPU fallback)");
    if (opt->device < 0 || opt->device >= ndev) return fail(nullptr, BQC_ERR_DEVICE, "bqc_create: device %d out of range", opt->device);
    bqc_ctx* c = new bqc_ctx();
    c->opt = *opt;
    c->device = opt->device;
    uint32_t nr = opt->n_refs ? opt->n_refs : 1;
    c->main_chrom.assign(nr, 0);
    if (opt->n_refs) memcpy(c->main_chrom.data(), opt->main_chrom, opt->n_refs);
    if (opt->fasta_index) c->fasta_index.assign(opt->fasta_index, opt->fasta_index + opt->n_refs);
    c->opt.main_chrom = c->main_chrom.data();
    c->opt.fasta_index = c->fasta_index.empty() ? nullptr : c->fasta_index.data();
    c->sl = make_state_layout(opt->n_lanes, opt->max_read_len, opt->hist_cap, (uint32_t)opt->isize + 1);
    c->cov.assign(opt->n_lanes, LaneCov());
    c->d_ref.assign(nr, nullptr);
    c->d_refn.assign(nr, nullptr);
    c->ref_len.assign(nr, 0);
    const char* v = getenv("BQC_NO_FAST");
    c->no_fast = v && v[0] == '1';
#define CCHK(call)                                                                                            \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            fail(nullptr, BQC_ERR_DEVICE, "bqc_create: %s failed: %s", #call, hipGetErrorString(e_));         \
            bqc_destroy(c);                                                                                   \
            return BQC_ERR_DEVICE;                                                                            \
        }                                                                                                     \
    } while (0)
    CCHK(hipSetDevice(c->device));
    hipDeviceProp_t prop;
    CCHK(hipGetDeviceProperties(&prop, c->device));
    c->n_cu = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256;
    CCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CCHK(bqc_long_init());
    CCHK(bqc_short_init());
    CCHK(hipMalloc(&c->d_state, c->sl.words * 8));
    CCHK(hipMalloc(&c->d_err, 64));
    c->t8_slots_cap = std::max(1024u, 4u * c->n_cu);
    CCHK(hipMalloc(&c->d_t8rows, (size_t)c->t8_slots_cap * BQC_T8_SPW * 65536));
    CCHK(hipMalloc(&c->d_t8used, (size_t)c->t8_slots_cap * 4));
    CCHK(hipMalloc(&c->d_carry, (size_t)opt->n_lanes * 2 * 2000 * 4));
    CCHK(hipMalloc(&c->d_parity, ((size_t)opt->n_lanes + 1) * 4));
    CCHK(hipMalloc(&c->d_started, opt->n_lanes));
    CCHK(hipMalloc(&c->d_ref_ptrs, sizeof(uint8_t*) * nr));
    CCHK(hipMalloc(&c->d_ref_len, sizeof(uint64_t) * nr));
    CCHK(hipMalloc(&c->d_refn_ptrs, sizeof(uint32_t*) * nr));
    CCHK(hipMalloc(&c->d_main, nr));
    CCHK(hipMemcpy(c->d_main, c->main_chrom.data(), nr, hipMemcpyHostToDevice));
    CCHK(hipMemsetAsync(c->d_state, 0, c->sl.words * 8, c->stream));
    CCHK(hipMemsetAsync(c->d_err, 0, 64, c->stream));
    CCHK(hipMemsetAsync(c->d_carry, 0, (size_t)opt->n_lanes * 2 * 2000 * 4, c->stream));
    CCHK(hipMemsetAsync(c->d_parity, 0, ((size_t)opt->n_lanes + 1) * 4, c->stream));
    CCHK(hipMemsetAsync(c->d_started, 0, opt->n_lanes, c->stream));
    CCHK(hipStreamSynchronize(c->stream));
    if (upload_ref_tables(c)) { snprintf(g_create_err, sizeof g_create_err, "%s", c->err.c_str()); bqc_destroy(c); return BQC_ERR_DEVICE; }
    if (opt->sketch.n_k && opt->sketch.n_q) {
        std::string e;
        c->sketch = sketch_create(opt->sketch, opt->n_lanes, c->stream, e);
        if (!c->sketch) { fail(nullptr, BQC_ERR_ARG, "bqc_create: sketch: %s", e.c_str()); bqc_destroy(c); return BQC_ERR_ARG; }
    }
    for (int i = 0; i < 16; ++i) {
        hipEvent_t e;
        CCHK(hipEventCreate(&e));
        c->ev.push_back(e);
    }
    *out = c;
    return 0;
}

static void free_prep_cache(void* p);
extern "C" void bqc_destroy(bqc_ctx* c)
{
    if (c) { free_prep_cache(c->prep_cache); c->prep_cache = nullptr; }
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto p : c->d_ref) if (p) (void)hipFree(p);
    for (auto p : c->d_refn) if (p) (void)hipFree(p);
    (void)hipFree(c->d_refn_ptrs);
    if (c->sketch) sketch_destroy(c->sketch);
    for (auto& pb : c->pool) (void)hipFree(pb.first);
    (void)hipFree(c->d_state); (void)hipFree(c->d_err); (void)hipFree(c->d_t8rows); (void)hipFree(c->d_t8used); (void)hipFree(c->d_carry); (void)hipFree(c->d_parity);
    (void)hipFree(c->d_started); (void)hipFree(c->d_ref_ptrs); (void)hipFree(c->d_ref_len); (void)hipFree(c->d_main);
    for (auto e : c->ev) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int bqc_set_reference(bqc_ctx* c, int32_t rid, const uint8_t* dna5, uint64_t len)
{
    if (!c || rid < 0 || (uint32_t)rid >= c->opt.n_refs || (!dna5 && len)) return fail(c, BQC_ERR_ARG, "bqc_set_reference: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->d_ref[rid]) { HIPCHK(c, hipFree(c->d_ref[rid])); c->d_ref[rid] = nullptr; }
    if (c->d_refn[rid]) { HIPCHK(c, hipFree(c->d_refn[rid])); c->d_refn[rid] = nullptr; }
    uint8_t* p = nullptr;
    HIPCHK(c, hipMalloc(&p, len ? len : 1));
    HIPCHK(c, hipMemcpy(p, dna5, len, hipMemcpyHostToDevice));
    const uint64_t nd8 = (len + 7) / 8; // nibble table of the fast path: BQC_FAST_NH pad dwords + nd8 + BQC_FAST_NH pad dwords
    uint32_t* pn = nullptr;
    HIPCHK(c, hipMalloc(&pn, (nd8 + 2 * BQC_FAST_NH) * 4));
    bqc_launch_ref_nibbles(p, len, pn, nd8, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->d_ref[rid] = p;
    c->d_refn[rid] = pn;
    c->ref_len[rid] = len;
    return upload_ref_tables(c);
}

// ---------------------------------------------------------------------------------------------------
// host pre-pass
// ---------------------------------------------------------------------------------------------------
namespace {
struct Prep {
    std::vector<uint16_t> flag;
    std::vector<uint32_t> seq_off, qual_off, cigar_off, perm;
    std::vector<CovEntry> cov_list;
    std::vector<Chunk> chunks, chunks_fast;
    std::vector<TripSeg> segs;
    uint32_t fast_w = 10;
    std::vector<CovTile> tiles;
    std::vector<uint8_t> lane_mask;
    std::vector<uint64_t> add_idx, add_val;
    uint64_t seq_bytes = 0, qual_bytes = 0, cigar_words = 0;
    bool identity = true;
    uint32_t long_max_len = 0;
    // scratch of the pre-pass (kept with the context from batch to batch: ~100 MB that would otherwise be mapped, faulted in
    // and unmapped for every batch)
    std::vector<int8_t> elig;
    std::vector<uint8_t> cand;
    std::vector<uint32_t> run_c, np;
    std::vector<uint64_t> run_len;
    std::vector<std::vector<CovEntry>> lane_list;
    std::vector<std::vector<uint32_t>> lane_win, lane_ewin;
    void reset()
    {
        flag.clear(); seq_off.clear(); qual_off.clear(); cigar_off.clear(); perm.clear(); cov_list.clear(); chunks.clear(); chunks_fast.clear();
        segs.clear(); tiles.clear(); lane_mask.clear(); add_idx.clear(); add_val.clear();
        fast_w = 10; seq_bytes = qual_bytes = cigar_words = 0; identity = true; long_max_len = 0;
        np.clear();
        for (auto& v : lane_list) v.clear();
        for (auto& v : lane_win) v.clear();
        for (auto& v : lane_ewin) v.clear();
    }
};
}
static void free_prep_cache(void* p) { delete (Prep*)p; }

// checkFlagsAndQuality (TripletCounting.hpp:136-168): 1 eligible, 0 not, -1 fatal
static int triplet_eligible(uint32_t flag, uint32_t mapq, int32_t as, const uint32_t* cg, uint32_t ncig)
{
    if (!(flag & 0x1) || !(flag & 0x2) || (flag & 0x4) || (flag & 0x8) || (flag & 0x100)) return 0;
    if (mapq < 60) return 0;
    if (as == BQC_AS_ABSENT || as < 0) return -1;
    if (as < 50) return 0;
    uint32_t clipped = 0;
    for (uint32_t k = 0; k < ncig; ++k) {
        uint32_t op = cg[k] & 15u;
        if (op == 4u || op == 5u) clipped += cg[k] >> 4;
    }
    return clipped > 0 ? 0 : 1;
}

static int prepass(bqc_ctx* c, const bqc_batch* b, Prep& P)
{
    const uint32_t n = b->n_reads;
    const uint32_t nl = c->opt.n_lanes;
    const bool timing = getenv("BQC_TIMING") && getenv("BQC_TIMING")[0] == '2';
    const auto tp0 = std::chrono::steady_clock::now();
    // (these vectors are kept with the context: when one has to grow it asks for huge pages before it is touched)
    auto grow = [](auto& v, size_t k) { const size_t cap = v.capacity(); v.resize(k); if (v.capacity() != cap) advise_huge(v); };
    auto room = [](auto& v, size_t k) { const size_t cap = v.capacity(); v.reserve(k); if (v.capacity() != cap) advise_huge(v); };
    grow(P.flag, n);
    grow(P.seq_off, n); grow(P.qual_off, n); grow(P.cigar_off, n);
    P.lane_mask.assign(nl, 0);
    uint64_t so = 0, qo = 0, co = 0;
    auto& lane_list = P.lane_list; // covered intervals, in stream order (windows never decrease)
    auto& lane_win = P.lane_win;   // first live window per coverage read
    auto& lane_ewin = P.lane_ewin; // ... per interval
    lane_list.resize(nl); lane_win.resize(nl); lane_ewin.resize(nl);
    std::vector<uint8_t> started_before(nl);
    for (uint32_t l = 0; l < nl; ++l) {
        started_before[l] = !c->cov[l].first;
        c->cov[l].batch_base = c->cov[l].win;
    }
    // Three passes so that the per-read work can use the host's cores: (0) payload sizes per thread range -> offsets,
    // (1) everything that depends on the read alone, in parallel: device flag, triplet eligibility, the shape of the covered
    // interval; (2) the order-dependent rules in read order, a few operations per read: FASTA cursor, coverage anchors.
    // Errors are reported for the first failing read, with the reference's order of checks within a read.
    struct PErr { uint32_t index = 0xFFFFFFFFu; int order = 0; int code = 0; uint32_t a = 0, b2 = 0; }; // order: 1 length, 2 lane, 4 AS, 6 mate flag
    const unsigned nt_max = std::min(16u, bqc_host_threads());
    std::vector<uint64_t> tso(nt_max + 1, 0), tqo(nt_max + 1, 0), tco(nt_max + 1, 0);
    std::vector<uint8_t> tmulti(nt_max, 0);
    const uint32_t lane0 = n ? b->lane[0] : 0;
    const unsigned nt = parallel_ranges(n, nt_max, 65536, [&](unsigned t, size_t lo, size_t hi) {
        uint64_t s1 = 0, s2 = 0, s3 = 0;
        uint8_t ml = 0;
        for (size_t i = lo; i < hi; ++i) {
            const uint32_t L = b->l_seq[i];
            s1 += (L + 1) / 2; s2 += L; s3 += b->n_cigar[i];
            ml |= b->lane[i] != lane0;
        }
        tso[t + 1] = s1; tqo[t + 1] = s2; tco[t + 1] = s3; tmulti[t] = ml;
    });
    for (unsigned t = 0; t < nt; ++t) { tso[t + 1] += tso[t]; tqo[t + 1] += tqo[t]; tco[t + 1] += tco[t]; }
    so = tso[nt]; qo = tqo[nt]; co = tco[nt];
    bool multi_lane = false;
    for (unsigned t = 0; t < nt; ++t) multi_lane |= tmulti[t] != 0;
    const bool offsets_fit = so <= 0xFFFFFFFFull && qo <= 0xFFFFFFFFull && co <= 0xFFFFFFFFull; // (else: found in read order below)
    auto& elig = P.elig;       // triplet_eligible (every element is written by pass 1)
    auto& cand = P.cand;       // 1: enters coverage, one covered run; 2: several runs (walked again in pass 2)
    auto& run_c = P.run_c;     // value of `c` at the start of the covered run (valid where cand != 0)
    auto& run_len = P.run_len; // its length
    grow(elig, n); grow(cand, n); grow(run_c, n); grow(run_len, n);
    for (uint32_t l = 0; l < nl && nl <= 4; ++l) { room(lane_list[l], n); room(lane_win[l], n); room(lane_ewin[l], n); } // (a few read groups: no regrowth in pass 2)
    std::vector<PErr> perr(nt);
    parallel_ranges(n, nt, 1, [&](unsigned t, size_t lo, size_t hi) { // (same ranges as pass 0: nt threads, n items)
        uint64_t so_ = tso[t], qo_ = tqo[t], co_ = tco[t];
        PErr& E = perr[t];
        for (size_t i = lo; i < hi; ++i) {
            const uint32_t L = b->l_seq[i], nc = b->n_cigar[i], lane = b->lane[i];
            if (E.index == 0xFFFFFFFFu) {
                if (L > c->opt.max_read_len) { E.index = (uint32_t)i; E.order = 1; E.a = L; }
                else if (lane >= nl) { E.index = (uint32_t)i; E.order = 2; E.a = lane; }
            }
            P.seq_off[i] = (uint32_t)so_; P.qual_off[i] = (uint32_t)qo_; P.cigar_off[i] = (uint32_t)co_;
            uint32_t flag = b->flag[i] & (0x0FFFu | BQC_FLAG_MATE_MAIN | BQC_FLAG_NO_QUAL);
            if (L > 0 && b->qual[qo_] == 0xFF) flag |= BQC_FLAG_NO_QUAL;
            const uint32_t* cg = b->cigar + co_;
            so_ += (L + 1) / 2; qo_ += L; co_ += nc;
            int8_t e = 0;
            uint8_t cd = 0;
            if (!(flag & 0x900)) { // primary record: bamqualcheck.cpp:318-327
                const bool dup = flag & 0x400, qcf = flag & 0x200;
                if (!dup && !qcf) { // tripletCounting, :338-342
                    e = (int8_t)triplet_eligible(flag, b->mapq[i], b->as[i], cg, nc);
                    if (e < 0 && E.index == 0xFFFFFFFFu) { E.index = (uint32_t)i; E.order = 4; }
                }
                if (!(flag & 0xC0)) { if (E.index == 0xFFFFFFFFu) { E.index = (uint32_t)i; E.order = 6; } }
                else {
                    const int32_t rid = b->rid[i];
                    const bool in_main = rid >= 0 && (uint32_t)rid < c->opt.n_refs && c->main_chrom[rid];
                    if (in_main && !(flag & 0x4) && !dup) { // all.coverage(record), :430-433: the covered run(s) relative to pos
                        const bool rc = flag & 0x10;
                        uint32_t cc = 0; // `int c` in the reference; wraps identically
                        uint32_t runs = 0, c0 = 0;
                        uint64_t len = 0, next = 0; // (next: value of c right behind the current run, as a 64-bit position)
                        for (uint32_t k = 0; k < nc; ++k) {
                            const uint32_t w = cg[rc ? nc - 1 - k : k], op = w & 15u, nn = w >> 4;
                            if (op == 4u) cc += nn;
                            if (op == 0u || op == 2u) {
                                if (runs && next == (uint64_t)cc) len += nn;
                                else { if (++runs == 1) { c0 = cc; len = nn; } }
                                next = (uint64_t)cc + nn;
                                cc += nn;
                            }
                        }
                        cd = runs <= 1 ? 1 : 2;
                        run_c[i] = c0; run_len[i] = runs ? len : 0;
                    }
                }
            }
            elig[i] = e; cand[i] = cd;
            P.flag[i] = (uint16_t)flag;
        }
    });
    // first error of the parallel passes (the lowest read index; within a read the order of the checks)
    PErr first;
    for (auto& E : perr) if (E.index < first.index) first = E;
    if (!offsets_fit) { // rare: find the first read whose offsets do not fit, in read order
        uint64_t a1 = 0, a2 = 0, a3 = 0;
        for (uint32_t i = 0; i < n; ++i) {
            if (a1 > 0xFFFFFFFFull || a2 > 0xFFFFFFFFull || a3 > 0xFFFFFFFFull) { if (i <= first.index && !(i == first.index && first.order <= 2)) { first = PErr(); first.index = i; first.order = 3; } break; }
            a1 += (b->l_seq[i] + 1) / 2; a2 += b->l_seq[i]; a3 += b->n_cigar[i];
        }
    }
    auto report = [&](const PErr& E) {
        switch (E.order) {
        case 1: return fail(c, BQC_ERR_RANGE, "read %u is %u bases long; max_read_len is %u", E.index, E.a, c->opt.max_read_len);
        case 2: return fail(c, BQC_ERR_ARG, "read %u: lane %u out of range", E.index, E.a);
        case 3: return fail(c, BQC_ERR_ARG, "batch too large: payload offsets exceed 32 bits (split the batch)");
        case 4: return fail(c, BQC_ERR_AS_TAG, "ERROR: read %u has no usable AS tag.", E.index);
        default: return fail(c, BQC_ERR_NO_MATE_FLAG, "ERROR: No first or second flag in read %u", E.index);
        }
    };
    // pass 2: order-dependent rules, in read order (up to the first failing read)
    const uint32_t n_ok = std::min<uint32_t>(n, first.index == 0xFFFFFFFFu ? n : first.index + 1);
    for (uint32_t i = 0; i < n_ok; ++i) {
        const bool failing = i == first.index;
        if (failing && first.order <= 4) return report(first); // these checks come before the FASTA rule of the same read
        if (elig[i] > 0) { // Genome: forward-only FASTA scan (TripletCounting.hpp:254-259)
            const int32_t rid = b->rid[i];
            int32_t target = -1;
            if (rid >= 0 && (uint32_t)rid < c->opt.n_refs) target = c->fasta_index.empty() ? rid : c->fasta_index[rid];
            if (target < 0 || target < c->fasta_cursor || !c->d_ref[rid])
                return fail(c, BQC_ERR_FASTA, "ERROR: Could not read fasta record for reference id %d (read %u)", rid, i);
            c->fasta_cursor = target;
            P.flag[i] |= BQC_FLAG_TRIPLET;
        }
        if (failing) return report(first); // (missing mate flag)
        if (!cand[i]) continue;
        const uint32_t lane = b->lane[i];
        const int32_t rid = b->rid[i];
        LaneCov& s = c->cov[lane];
        const uint32_t beginpos = (uint32_t)b->pos[i];
        if (s.first) { s.first = false; s.id = rid; s.shift = (int32_t)beginpos; }
        if (s.id != rid || (uint32_t)(beginpos - (uint32_t)s.shift) > 2u * BQC_VSIZE) { // reset: two windows flushed
            s.id = rid; s.win += 2; s.shift = (int32_t)beginpos;
        }
        uint32_t pos = beginpos - (uint32_t)s.shift;
        if (pos > BQC_VSIZE && pos < 2u * BQC_VSIZE) { // slide: one window flushed
            s.win += 1; s.shift += BQC_VSIZE; pos = beginpos - (uint32_t)s.shift;
        }
        const uint64_t rel = s.win - s.batch_base;
        if (rel > 0xFFFFFFF0ull) return fail(c, BQC_ERR_ARG, "batch spans too many coverage windows (split the batch)");
        P.flag[i] |= BQC_FLAG_COV;
        lane_win[lane].push_back((uint32_t)rel);
        // The read's covered interval(s) relative to its first live window (OverallNumbers.hpp:112-131): `c` runs over the
        // seq-oriented CIGAR (reversed for reverse reads, bamqualcheck.cpp:349) and advances on S, M and D; M and D add
        // coverage.  DEFINED: increments at window offset >= 2000 are dropped.  One interval unless a clip sits between
        // two match operations.
        auto emit = [&](int64_t a, int64_t z) {
            z = std::min<int64_t>(z, 2 * BQC_VSIZE);
            if (a >= 0 && a < z) {
                lane_list[lane].push_back(CovEntry{(uint32_t)rel, (uint32_t)a | ((uint32_t)(z - a) << 16)});
                lane_ewin[lane].push_back((uint32_t)rel);
            }
        };
        if (cand[i] == 1) {
            if (run_len[i]) { const int64_t a = (int64_t)pos + run_c[i]; emit(a, a + (int64_t)std::min<uint64_t>(run_len[i], 1ull << 40)); }
        } else { // several runs: walk the CIGAR again
            const uint32_t nc = b->n_cigar[i];
            const uint32_t* cg = b->cigar + P.cigar_off[i];
            const bool rc = P.flag[i] & 0x10;
            uint32_t cc = 0;
            int64_t run_a = -1, run_z = -1;
            for (uint32_t k = 0; k < nc; ++k) {
                const uint32_t w = cg[rc ? nc - 1 - k : k], op = w & 15u, nn = w >> 4;
                if (op == 4u) cc += nn;
                if (op == 0u || op == 2u) {
                    const int64_t a = (int64_t)pos + cc, z = a + nn;
                    if (run_z == a) run_z = z;
                    else { if (run_a >= 0) emit(run_a, run_z); run_a = a; run_z = z; }
                    cc += nn;
                }
            }
            if (run_a >= 0) emit(run_a, run_z);
        }
    }
    if (first.index != 0xFFFFFFFFu) return report(first); // (unreachable: reported inside the loop)
    P.seq_bytes = so; P.qual_bytes = qo; P.cigar_words = co;
    // extras must reference valid reads
    for (uint32_t e = 0; e < b->n_nm_extra; ++e)
        if (b->nm_extra_read[e] >= n) return fail(c, BQC_ERR_ARG, "nm_extra_read out of range");

    const auto tp1 = std::chrono::steady_clock::now();
    // ---- lane grouping (stable) and chunk table
    P.identity = !multi_lane;
    if (multi_lane) {
        std::vector<uint32_t> cnt(nl + 1, 0);
        for (uint32_t i = 0; i < n; ++i) cnt[b->lane[i] + 1]++;
        for (uint32_t l = 0; l < nl; ++l) cnt[l + 1] += cnt[l];
        P.perm.resize(n);
        for (uint32_t i = 0; i < n; ++i) P.perm[cnt[b->lane[i]]++] = i;
    }
    { // chunk tables: reads of up to BQC_FAST_MAXLEN bases -> k_short (chunks_fast), everything else -> k_long (chunks).
      // A fast chunk is a sequence of groups of rpw entries (what a wave of k_short handles at once): the first h0 slots of a
      // group hold first-mate reads, the other h1 second-mate reads, both taken in stream order, so that a group covers one
      // short stretch of the stream (its 128-byte lines hold reads of both mates) while every lane of k_short still sees reads
      // of one mate only and can keep that mate's per-cycle counters in registers.  Missing reads of a mate are null entries
      // (0xFFFFFFFF).  Behind the read groups: the triplet segments of those reads, as groups of their own.
        uint32_t maxfast = 0;
        if (!c->no_fast)
            for (uint32_t i = 0; i < n; ++i) if (b->l_seq[i] <= BQC_FAST_MAXLEN) maxfast = std::max(maxfast, b->l_seq[i]);
        P.fast_w = std::max(1u, (maxfast + 8 * BQC_FAST_NH - 1) / (8 * BQC_FAST_NH)); // lanes per read: 8 * BQC_FAST_NH sequencing cycles each
        const uint32_t rpw = 64u / P.fast_w;           // reads a wave handles at once
        const uint32_t h0 = (rpw + 1) / 2, h1 = rpw / 2; // slots per mate
        const uint32_t groups_cap = BQC_FAST_WAVES * (64u / rpw); // groups per chunk: one tile of whole groups per wave of k_short
        auto& np = P.np;
        room(np, n + n / 4); room(P.perm, n + n / 4); // (the two swap roles at the end)
        std::vector<uint32_t> q[2];    // reads of the current fast chunk per mate, in stream order
        std::vector<uint32_t> win_seg; // their triplet segments (indices into P.segs)
        uint32_t wlane = 0;
        auto groups_of = [&](size_t n0, size_t n1) { return (uint32_t)std::max((n0 + h0 - 1) / h0, h1 ? (n1 + h1 - 1) / h1 : (n1 ? (size_t)1 << 30 : 0)); };
        auto flush_window = [&]() {
            if (q[0].empty() && q[1].empty()) return;
            const uint32_t first = (uint32_t)np.size();
            const uint32_t ng = groups_of(q[0].size(), q[1].size());
            for (uint32_t g = 0; g < ng; ++g) {
                for (uint32_t k = 0; k < h0; ++k) { const size_t i = (size_t)g * h0 + k; np.push_back(i < q[0].size() ? q[0][i] : 0xFFFFFFFFu); }
                for (uint32_t k = 0; k < h1; ++k) { const size_t i = (size_t)g * h1 + k; np.push_back(i < q[1].size() ? q[1][i] : 0xFFFFFFFFu); }
            }
            const uint32_t n_read_entries = (uint32_t)np.size() - first;
            uint32_t cnt = 0;
            for (uint32_t k : win_seg) { np.push_back(BQC_ENTRY_SEG | k); ++cnt; }
            while (cnt % rpw) { np.push_back(0xFFFFFFFFu); ++cnt; }
            P.chunks_fast.push_back(Chunk{first, (uint32_t)np.size() - first, wlane, n_read_entries, 0, 0, 0, 0});
            q[0].clear(); q[1].clear(); win_seg.clear();
        };
        uint32_t start = 0, count = 0, cl = 0;
        uint64_t bases = 0;
        auto close_slow = [&]() {
            if (count) P.chunks.push_back(Chunk{start, count, cl, 0, 0, 0, 0, 0});
            count = 0; bases = 0;
        };
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t r = P.identity ? k : P.perm[k];
            const uint32_t L = b->l_seq[r], lane = b->lane[r];
            const bool fast = !c->no_fast && L <= BQC_FAST_MAXLEN;
            if (fast) {
                close_slow();
                if (lane != wlane) flush_window();
                // k_short evaluates triplets with chromPos = pos + i inside the first CIGAR operation (assumed match-like,
                // TripletCounting.hpp:203); every further match-like operation becomes a segment entry with its own offset
                const uint32_t seg0 = (uint32_t)P.segs.size();
                if ((P.flag[r] & BQC_FLAG_TRIPLET) && b->n_cigar[r] > 1 && L >= 3) {
                    const uint32_t* cg = b->cigar + P.cigar_off[r];
                    const uint32_t n0 = cg[0] >> 4;
                    if (n0 != 0) { // (n0 == 0: every position counts as inside the first operation, no walk)
                        uint64_t rp = n0;
                        int64_t cpos = (int64_t)b->pos[r] + n0;
                        for (uint32_t k2 = 1; k2 < b->n_cigar[r] && rp < L; ++k2) {
                            const uint32_t op = cg[k2] & 15u, nn = cg[k2] >> 4;
                            if (op == 2u || op == 3u || op == 5u || op == 6u) cpos += nn;   // D N H P
                            else if (op == 4u || op == 1u) rp += nn;                          // S I
                            else {                                                            // M = X (and unknown)
                                const uint64_t ia = std::max<uint64_t>(rp, 1), ib = std::min<uint64_t>(rp + nn, (uint64_t)L - 1);
                                const int64_t posv = cpos - (int64_t)rp;
                                if (ia < ib && posv > INT32_MIN / 2 && posv < INT32_MAX / 2)
                                    P.segs.push_back(TripSeg{r, (int32_t)posv, (uint32_t)ia | ((uint32_t)ib << 8), 0});
                                rp += nn; cpos += nn;
                            }
                        }
                    }
                }
                const uint32_t m = (P.flag[r] & 0x40u) ? 0u : 1u;
                if (m == 1 && h1 == 0) return fail(c, BQC_ERR_RANGE, "internal: no slot for second-mate reads"); // (rpw >= 4 always)
                if (groups_of(q[0].size() + (m == 0), q[1].size() + (m == 1)) > groups_cap) flush_window(); // this read opens the next chunk
                wlane = lane;
                q[m].push_back(r);
                for (uint32_t k = seg0; k < P.segs.size(); ++k) win_seg.push_back(k);
                continue;
            }
            flush_window();
            if (count && (lane != cl || count == (uint32_t)BQC_CHUNK_READS || bases + (uint64_t)L > BQC_CHUNK_BASES)) close_slow();
            P.long_max_len = std::max(P.long_max_len, L);
            if (!count) { start = (uint32_t)np.size(); cl = lane; }
            np.push_back(r);
            ++count; bases += L;
        }
        close_slow();
        flush_window();
        P.perm.swap(np);
        P.identity = false;
    }
    const auto tp2 = std::chrono::steady_clock::now();
    // ---- coverage tiles
    for (uint32_t l = 0; l < nl; ++l) {
        const auto& list = lane_list[l];
        const auto& win = lane_win[l];
        const auto& ewin = lane_ewin[l];
        if (win.empty()) continue;
        P.lane_mask[l] = 1;
        const uint32_t W1 = win.back(); // windows < W1 are complete after this batch
        const uint32_t base_off = (uint32_t)P.cov_list.size();
        P.cov_list.insert(P.cov_list.end(), list.begin(), list.end());
        std::vector<uint32_t> need; // tile ids, ascending
        auto push_tile = [&](uint32_t w) {
            uint32_t t = w / BQC_COV_TILE_WINDOWS;
            if (need.empty() || need.back() < t) need.push_back(t);
        };
        // tiles that hold a live window of some read (its first and the next one), the two windows carried in from the previous
        // batch and the two carried out (W1, W1 + 1 = the last read's).  `win` never decreases, so one ordered pass suffices.
        if (started_before[l]) push_tile(0), push_tile(1);
        for (size_t k = 0; k < win.size(); ++k) { push_tile(win[k]); push_tile(win[k] + 1); }
        uint64_t covered_final = 0;
        for (uint32_t t : need) {
            const uint32_t wlo = t * BQC_COV_TILE_WINDOWS;
            const uint32_t lo_key = wlo == 0 ? 0 : wlo - 1;
            const uint32_t b0 = (uint32_t)(std::lower_bound(ewin.begin(), ewin.end(), lo_key) - ewin.begin());
            const uint32_t b1 = (uint32_t)(std::lower_bound(ewin.begin(), ewin.end(), wlo + BQC_COV_TILE_WINDOWS) - ewin.begin());
            CovTile ct{};
            ct.lane = l; ct.win_lo = wlo; ct.list_begin = base_off + b0; ct.list_end = base_off + b1; ct.win_final = W1;
            P.tiles.push_back(ct);
            const uint64_t hi = std::min<uint64_t>((uint64_t)wlo + BQC_COV_TILE_WINDOWS, W1);
            if (hi > wlo) covered_final += hi - wlo;
        }
        if (W1 > covered_final) { // complete windows nobody touched: depth 0 everywhere
            P.add_idx.push_back(c->sl.lane_base(l) + c->sl.o_poscov + 0);
            P.add_val.push_back((uint64_t)(W1 - covered_final) * BQC_VSIZE);
        }
        // the next batch numbers its windows from this batch's last live window
        c->cov[l].batch_base = c->cov[l].win;
    }
    if (timing) {
        const auto tp3 = std::chrono::steady_clock::now();
        auto d = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point z) { return std::chrono::duration<double>(z - a).count(); };
        fprintf(stderr, "[timing] pre-pass of %u reads: per-read annotations + coverage %.3f s, chunk tables %.3f s, coverage tiles %.3f s\n", n, d(tp0, tp1),
                d(tp1, tp2), d(tp2, tp3));
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// upload / process
// ---------------------------------------------------------------------------------------------------
namespace {
struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; }
};
}

extern "C" void bqc_dbatch_free(bqc_ctx* c, bqc_dbatch* db)
{
    if (!db) return;
    if (c) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
    if (c && db->dmem && c->pool.size() < 3) c->pool.emplace_back(db->dmem, db->dcap);
    else (void)hipFree(db->dmem);
    delete db;
}
extern "C" uint64_t bqc_dbatch_bytes(const bqc_dbatch* db) { return db ? db->algo_bytes : 0; }

extern "C" int bqc_upload(bqc_ctx* c, const bqc_batch* b, bqc_dbatch** out)
{
    if (!c || !b || !out) return fail(c, BQC_ERR_ARG, "bqc_upload: null argument");
    if (c->poisoned) return fail(c, BQC_ERR_STATE, "context is in an error state: %s", c->err.c_str());
    if (c->flushed) return fail(c, BQC_ERR_STATE, "bqc_upload after bqc_flush/bqc_finalize (call bqc_reset first)");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->prep_cache) c->prep_cache = new Prep();
    Prep& P = *(Prep*)c->prep_cache;
    P.reset();
    const bool timing = getenv("BQC_TIMING") && getenv("BQC_TIMING")[0] == '2';
    const auto t0 = std::chrono::steady_clock::now();
    int rc = prepass(c, b, P);
    if (rc) { c->poisoned = true; return rc; }
    const auto t1 = std::chrono::steady_clock::now();
    const uint32_t n = b->n_reads;
    bqc_dbatch* db = new bqc_dbatch();
    Carver cv;
    const size_t o_flag = cv.take(2ull * n), o_mapq = cv.take(n), o_lane = cv.take(n), o_rid = cv.take(4ull * n), o_pos = cv.take(4ull * n),
                 o_tlen = cv.take(4ull * n), o_nm = cv.take(4ull * n), o_as = cv.take(4ull * n), o_lseq = cv.take(4ull * n),
                 o_ncig = cv.take(2ull * n), o_soff = cv.take(4ull * n),
                 o_qoff = cv.take(4ull * n), o_cgoff = cv.take(4ull * n), o_seq = cv.take(P.seq_bytes + 512), o_qual = cv.take(P.qual_bytes + 512),
                 o_cig = cv.take(4 * P.cigar_words + 16), o_perm = cv.take(P.identity ? 0 : 4ull * P.perm.size()),
                 o_chunks = cv.take(sizeof(Chunk) * P.chunks.size()), o_chf = cv.take(sizeof(Chunk) * P.chunks_fast.size()),
                 o_segs = cv.take(sizeof(TripSeg) * P.segs.size()), o_xr = cv.take(4ull * b->n_nm_extra), o_xv = cv.take(4ull * b->n_nm_extra),
                 o_clist = cv.take(sizeof(CovEntry) * P.cov_list.size()), o_tiles = cv.take(sizeof(CovTile) * P.tiles.size()),
                 o_rsum = cv.take(P.chunks.empty() ? 0 : 12ull * n), o_mask = cv.take(c->opt.n_lanes), o_started = cv.take(c->opt.n_lanes), o_aidx = cv.take(8ull * P.add_idx.size()), o_aval = cv.take(8ull * P.add_val.size());
    db->dbytes = cv.off + 256;
    hipError_t he = hipSuccess;
    for (size_t k = 0; k < c->pool.size(); ++k)
        if (c->pool[k].second >= db->dbytes && c->pool[k].second <= 2 * db->dbytes + (64u << 20)) { // a freed buffer of a similar size
            db->dmem = c->pool[k].first; db->dcap = c->pool[k].second;
            c->pool.erase(c->pool.begin() + k);
            break;
        }
    if (!db->dmem) {
        if (c->pool.size() >= 3) { (void)hipFree(c->pool.front().first); c->pool.erase(c->pool.begin()); }
        db->dcap = db->dbytes + db->dbytes / 16; // a little slack, so that the next batch of about this size fits as well
        he = hipMalloc(&db->dmem, db->dcap);
    }
    if (he != hipSuccess) { delete db; c->poisoned = true; return fail(c, BQC_ERR_DEVICE, "hipMalloc(%zu) failed: %s", db->dbytes, hipGetErrorString(he)); }
    char* base = (char*)db->dmem;
#define UP(off, src, bytes)                                                                                     \
    do {                                                                                                        \
        if ((bytes) > 0) {                                                                                      \
            hipError_t e_ = hipMemcpyAsync(base + (off), (src), (bytes), hipMemcpyHostToDevice, c->stream);     \
            if (e_ != hipSuccess) { bqc_dbatch_free(c, db); c->poisoned = true; return fail(c, BQC_ERR_DEVICE, "upload failed: %s", hipGetErrorString(e_)); } \
        }                                                                                                       \
    } while (0)
    UP(o_flag, P.flag.data(), 2ull * n); UP(o_mapq, b->mapq, n); UP(o_lane, b->lane, n); UP(o_rid, b->rid, 4ull * n);
    UP(o_pos, b->pos, 4ull * n); UP(o_tlen, b->tlen, 4ull * n); UP(o_nm, b->nm, 4ull * n); UP(o_as, b->as, 4ull * n);
    UP(o_lseq, b->l_seq, 4ull * n); UP(o_ncig, b->n_cigar, 2ull * n);
    UP(o_soff, P.seq_off.data(), 4ull * n); UP(o_qoff, P.qual_off.data(), 4ull * n);
    UP(o_cgoff, P.cigar_off.data(), 4ull * n); UP(o_seq, b->seq, P.seq_bytes); UP(o_qual, b->qual, P.qual_bytes);
    UP(o_cig, b->cigar, 4 * P.cigar_words);
    if (!P.identity) UP(o_perm, P.perm.data(), 4ull * P.perm.size());
    UP(o_chunks, P.chunks.data(), sizeof(Chunk) * P.chunks.size());
    UP(o_chf, P.chunks_fast.data(), sizeof(Chunk) * P.chunks_fast.size());
    UP(o_segs, P.segs.data(), sizeof(TripSeg) * P.segs.size());
    UP(o_xr, b->nm_extra_read, 4ull * b->n_nm_extra); UP(o_xv, b->nm_extra_val, 4ull * b->n_nm_extra);
    UP(o_clist, P.cov_list.data(), sizeof(CovEntry) * P.cov_list.size()); UP(o_tiles, P.tiles.data(), sizeof(CovTile) * P.tiles.size());
    UP(o_mask, P.lane_mask.data(), c->opt.n_lanes);
    std::vector<uint8_t> st(c->opt.n_lanes);
    for (uint32_t l = 0; l < c->opt.n_lanes; ++l) st[l] = !c->cov[l].first; // lanes that have seen a coverage read so far
    UP(o_started, st.data(), c->opt.n_lanes);
    UP(o_aidx, P.add_idx.data(), 8ull * P.add_idx.size()); UP(o_aval, P.add_val.data(), 8ull * P.add_val.size());
    he = hipStreamSynchronize(c->stream); // buffers may be reused by the caller on return
    if (timing) {
        const auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[timing] upload of %u reads: pre-pass %.3f s, H2D %.3f s (%.1f MB)\n", b->n_reads, std::chrono::duration<double>(t1 - t0).count(),
                std::chrono::duration<double>(t2 - t1).count(), db->dbytes / 1e6);
    }
    if (he != hipSuccess) { bqc_dbatch_free(c, db); c->poisoned = true; return fail(c, BQC_ERR_DEVICE, "upload sync failed: %s", hipGetErrorString(he)); }
    DevBatch& d = db->d;
    d.n_reads = n;
    d.flag = (const uint16_t*)(base + o_flag); d.mapq = (const uint8_t*)(base + o_mapq); d.lane = (const uint8_t*)(base + o_lane);
    d.rid = (const int32_t*)(base + o_rid); d.pos = (const int32_t*)(base + o_pos); d.tlen = (const int32_t*)(base + o_tlen);
    d.nm = (const int32_t*)(base + o_nm); d.as_ = (const int32_t*)(base + o_as); d.l_seq = (const uint32_t*)(base + o_lseq);
    d.n_cigar = (const uint16_t*)(base + o_ncig);
    d.seq_off = (const uint32_t*)(base + o_soff); d.qual_off = (const uint32_t*)(base + o_qoff); d.cigar_off = (const uint32_t*)(base + o_cgoff);
    d.seq = (const uint8_t*)(base + o_seq); d.qual = (const uint8_t*)(base + o_qual); d.cigar = (const uint32_t*)(base + o_cig);
    d.perm = P.identity ? nullptr : (const uint32_t*)(base + o_perm);
    d.n_perm = P.identity ? n : (uint32_t)P.perm.size();
    d.chunks = (const Chunk*)(base + o_chunks); d.n_chunks = (uint32_t)P.chunks.size();
    d.chunks_fast = (const Chunk*)(base + o_chf); d.n_chunks_fast = (uint32_t)P.chunks_fast.size(); d.fast_w = P.fast_w;
    d.segs = (const TripSeg*)(base + o_segs);
    d.nm_extra_read = (const uint32_t*)(base + o_xr); d.nm_extra_val = (const int32_t*)(base + o_xv); d.n_nm_extra = b->n_nm_extra;
    d.cov_list = (const CovEntry*)(base + o_clist); d.cov_tiles = (const CovTile*)(base + o_tiles); d.n_cov_tiles = (uint32_t)P.tiles.size();
    db->d_lane_mask = (uint8_t*)(base + o_mask);
    db->d_rsum = (uint32_t*)(base + o_rsum); db->long_max_len = P.long_max_len;
    { // the read group with the most fast chunks
        std::vector<uint32_t> cnt(c->opt.n_lanes, 0);
        for (const Chunk& fc : P.chunks_fast) if (++cnt[fc.lane] > cnt[db->t8_lane]) db->t8_lane = fc.lane;
    }
    db->d_add_idx = (uint64_t*)(base + o_aidx); db->d_add_val = (uint64_t*)(base + o_aval);
    db->add_idx = P.add_idx; db->add_val = P.add_val;
    db->algo_bytes = 48ull * n + P.seq_bytes + P.qual_bytes + 4 * P.cigar_words; // A(L,n) of SURVEY.md §8d summed over the batch
    db->d_started_after = (uint8_t*)(base + o_started);
    db->cov_after = c->cov;
    db->fasta_cursor_after = c->fasta_cursor;
    db->seq = ++c->upload_counter;
    c->state_seq = db->seq;
    *out = db;
    return 0;
}

static int check_device_error(bqc_ctx* c)
{
    uint32_t e = 0;
    HIPCHK(c, hipMemcpyAsync(&e, c->d_err, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!e) return 0;
    c->poisoned = true;
    if (e & BQC_DEVERR_INTERNAL) return fail(c, BQC_ERR_DEVICE, "internal error: kernel layout assumption violated");
    if (e & BQC_DEVERR_MATE) return fail(c, BQC_ERR_NO_MATE_FLAG, "ERROR: No first or second flag in read");
    if (e & BQC_DEVERR_RANGE) return fail(c, BQC_ERR_RANGE, "mismatch/deletion/insertion count exceeds hist_cap (or NM < D+I)");
    return fail(c, BQC_ERR_RANGE, "base quality above 222 cannot be represented by the reference (q+33 wraps)");
}

static void tick(bqc_ctx* c, const char* name)
{
    if (!c->timing || c->n_timed + 1 >= (int)c->ev.size()) return;
    (void)hipEventRecord(c->ev[c->n_timed + 1], c->stream);
    c->tnames.push_back(name);
    c->n_timed++;
}

static void fold_t8(bqc_ctx* c)
{
    if (!c->t8_slots_used) return;
    bqc_launch_t8_fold(c->d_t8rows, c->d_t8used, c->t8_slots_used, c->sl, c->d_state, c->t8_rows_lane, c->stream);
    c->t8_slots_used = 0;
}

extern "C" int bqc_process(bqc_ctx* c, bqc_dbatch* db)
{
    if (!c || !db) return fail(c, BQC_ERR_ARG, "bqc_process: null argument");
    if (c->poisoned) return fail(c, BQC_ERR_STATE, "context is in an error state: %s", c->err.c_str());
    if (c->flushed) return fail(c, BQC_ERR_STATE, "bqc_process after bqc_flush (call bqc_reset first)");
    HIPCHK(c, hipSetDevice(c->device));
    DevRefs refs{(const uint8_t* const*)c->d_ref_ptrs, c->d_ref_len, c->d_main, c->opt.n_refs, (const uint32_t* const*)c->d_refn_ptrs};
    if (db->seq > c->state_seq) { // re-processing after bqc_reset: this batch (uploaded on a fresh context) defines the stream state again
        c->cov = db->cov_after;
        c->fasta_cursor = db->fasta_cursor_after;
        c->state_seq = db->seq;
    }
    if (!db->d.n_cov_tiles) bqc_launch_or_bytes(c->d_started, db->d_started_after, c->opt.n_lanes, c->stream); // (else: k_cov's epilogue)
    if (c->timing) { c->n_timed = 0; c->tnames.clear(); (void)hipEventRecord(c->ev[0], c->stream); }
    DevBatch slow = db->d; // generic kernels see only the reads that are not on the fast path
    if (db->d.n_chunks_fast) {
        if (!(bqc_short_parts() & 8u)) { // (profiling only: per-read statistics of the fast chunks as a separate kernel)
            DevBatch fr = db->d;
            fr.chunks = db->d.chunks_fast; fr.n_chunks = db->d.n_chunks_fast;
            bqc_launch_reads_chunks(fr, c->sl, c->d_state, refs, c->d_err, c->n_cu, c->stream);
            tick(c, "k_reads");
        }
        const uint32_t grid = std::min(c->n_cu, db->d.n_chunks_fast); // one workgroup per CU; every workgroup owns a slot of scratch rows
        if (c->t8_rows_lane != db->t8_lane || c->t8_slots_used + grid > c->t8_slots_cap) { fold_t8(c); c->t8_rows_lane = db->t8_lane; }
        bqc_launch_short(db->d, c->sl, c->d_state, refs, c->d_err, grid, c->d_t8rows + (size_t)c->t8_slots_used * BQC_T8_SPW * 16384u,
                         c->d_t8used + c->t8_slots_used, db->t8_lane, c->stream);
        c->t8_slots_used += grid;
        tick(c, "k_short");
    }
    if (slow.n_chunks) {
        bqc_launch_reads_chunks(slow, c->sl, c->d_state, refs, c->d_err, c->n_cu, c->stream);
        tick(c, "k_reads(generic)");
        HIPCHK(c, hipMemsetAsync(db->d_rsum, 0, 12ull * db->d.n_reads, c->stream));
        bqc_launch_long(slow, c->sl, c->d_state, refs, c->d_err, db->d_rsum, db->long_max_len, c->n_cu, c->stream);
        tick(c, "k_long");
    }
    if (db->d.n_nm_extra) bqc_launch_nm_extra(db->d, c->sl, c->d_state, refs, c->d_err, c->stream);
    if (db->d.n_cov_tiles) {
        bqc_launch_cov(db->d, c->sl, c->d_state, c->d_carry, c->d_parity, db->d_lane_mask, c->d_started, db->d_started_after, c->opt.n_lanes, c->stream);
    }
    bqc_launch_add_words(c->d_state, db->d_add_idx, db->d_add_val, (uint32_t)db->add_idx.size(), c->stream);
    tick(c, "k_cov");
    if (c->sketch) { sketch_process(c->sketch, db->d, c->stream); tick(c, "k_sketch"); }
    HIPCHK(c, hipGetLastError());
    return 0;
}

extern "C" int bqc_submit(bqc_ctx* c, const bqc_batch* b)
{
    bqc_dbatch* db = nullptr;
    int rc = bqc_upload(c, b, &db);
    if (rc) return rc;
    rc = bqc_process(c, db);
    if (!rc) rc = check_device_error(c); // also drains the stream so the batch can be freed
    bqc_dbatch_free(c, db);
    return rc;
}

extern "C" int bqc_sync(bqc_ctx* c)
{
    if (!c) return BQC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_device_error(c);
}

extern "C" int bqc_reset(bqc_ctx* c)
{
    if (!c) return BQC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    fold_t8(c);
    HIPCHK(c, hipMemsetAsync(c->d_state, 0, c->sl.words * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_err, 0, 64, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_carry, 0, (size_t)c->opt.n_lanes * 2 * 2000 * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_parity, 0, ((size_t)c->opt.n_lanes + 1) * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_started, 0, c->opt.n_lanes, c->stream));
    if (c->sketch) sketch_reset(c->sketch, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->cov.assign(c->opt.n_lanes, LaneCov());
    c->fasta_cursor = -1;
    c->state_seq = 0;
    c->flushed = false;
    c->poisoned = false;
    return 0;
}

extern "C" int bqc_set_timing(bqc_ctx* c, int enable)
{
    if (!c) return BQC_ERR_ARG;
    c->timing = enable != 0;
    return 0;
}

extern "C" int bqc_last_timing(bqc_ctx* c, uint32_t* n, const char* const** names, const float** ms)
{
    if (!c || !n || !names || !ms) return BQC_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->tms.assign(c->n_timed, 0.f);
    for (int i = 0; i < c->n_timed; ++i) (void)hipEventElapsedTime(&c->tms[i], c->ev[i], c->ev[i + 1]);
    *n = (uint32_t)c->n_timed;
    *names = c->tnames.data();
    *ms = c->tms.data();
    return 0;
}

extern "C" int bqc_flush(bqc_ctx* c)
{
    if (!c) return BQC_ERR_ARG;
    if (c->poisoned) return fail(c, BQC_ERR_STATE, "context is in an error state: %s", c->err.c_str());
    if (c->flushed) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    fold_t8(c);
    bqc_launch_cov_final(c->sl, c->d_state, c->d_carry, c->d_parity, c->d_started, c->stream);
    HIPCHK(c, hipGetLastError());
    int rc = check_device_error(c);
    if (rc) return rc;
    c->flushed = true;
    return 0;
}

extern "C" uint64_t bqc_state_words(const bqc_ctx* c) { return c ? c->sl.words + (c->sketch ? sketch_state_words(c->sketch) : 0) : 0; }

extern "C" int bqc_state_export(bqc_ctx* c, void* dst)
{
    if (!c || !dst) return BQC_ERR_ARG;
    int rc = bqc_flush(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(dst, c->d_state, c->sl.words * 8, hipMemcpyDeviceToDevice, c->stream));
    if (c->sketch) sketch_state_export(c->sketch, (uint64_t*)dst + c->sl.words, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int bqc_state_import(bqc_ctx* c, const void* src)
{
    if (!c || !src) return BQC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    fold_t8(c);
    HIPCHK(c, hipMemcpyAsync(c->d_state, src, c->sl.words * 8, hipMemcpyDeviceToDevice, c->stream));
    if (c->sketch) sketch_state_import(c->sketch, (const uint64_t*)src + c->sl.words, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->flushed = true; // an imported vector is already flushed
    return 0;
}
extern "C" int bqc_state_export_host(bqc_ctx* c, uint64_t* dst)
{
    if (!c || !dst) return BQC_ERR_ARG;
    int rc = bqc_flush(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(dst, c->d_state, c->sl.words * 8, hipMemcpyDeviceToHost));
    if (c->sketch) {
        uint64_t* tmp = nullptr;
        const uint64_t w = sketch_state_words(c->sketch);
        HIPCHK(c, hipMalloc(&tmp, w * 8));
        sketch_state_export(c->sketch, tmp, c->stream);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(dst + c->sl.words, tmp, w * 8, hipMemcpyDeviceToHost));
        HIPCHK(c, hipFree(tmp));
    }
    return 0;
}
extern "C" int bqc_state_import_host(bqc_ctx* c, const uint64_t* src)
{
    if (!c || !src) return BQC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    fold_t8(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(c->d_state, src, c->sl.words * 8, hipMemcpyHostToDevice));
    if (c->sketch) {
        uint64_t* tmp = nullptr;
        const uint64_t w = sketch_state_words(c->sketch);
        HIPCHK(c, hipMalloc(&tmp, w * 8));
        HIPCHK(c, hipMemcpy(tmp, src + c->sl.words, w * 8, hipMemcpyHostToDevice));
        sketch_state_import(c->sketch, tmp, c->stream);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(tmp));
    }
    c->flushed = true;
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// finalize
// ---------------------------------------------------------------------------------------------------
static uint32_t last_nonzero_len(const uint64_t* p, uint32_t n)
{
    while (n > 0 && p[n - 1] == 0) --n;
    return n;
}

extern "C" int bqc_finalize(bqc_ctx* c, const bqc_counts** out)
{
    if (!c || !out) return BQC_ERR_ARG;
    int rc = bqc_flush(c);
    if (rc) return rc;
    const StateLayout& sl = c->sl;
    c->h_state.resize(sl.words);
    HIPCHK(c, hipMemcpy(c->h_state.data(), c->d_state, sl.words * 8, hipMemcpyDeviceToHost));
    c->arrays.clear();
    c->lanes.assign(sl.n_lanes, bqc_lane_counts{});
    c->sk_out.assign(sl.n_lanes, {});
    auto keep = [&](std::vector<uint64_t>&& v) -> const uint64_t* {
        c->arrays.push_back(std::move(v));
        return c->arrays.back().data();
    };
    auto u32copy = [&](const uint64_t* p, uint32_t n) { // reference type `unsigned`: value mod 2^32
        std::vector<uint64_t> v(n ? n : 1);
        for (uint32_t i = 0; i < n; ++i) v[i] = p[i] & 0xFFFFFFFFull;
        return v;
    };
    c->arrays.reserve(sl.n_lanes * 64);
    for (uint32_t l = 0; l < sl.n_lanes; ++l) {
        const uint64_t* S = c->h_state.data() + sl.lane_base(l);
        bqc_lane_counts& L = c->lanes[l];
        for (int i = 0; i < BQC_N_SCALARS; ++i)
            L.scalars[i] = i == BQC_S_TOTALBPS ? S[sl.o_scalars + i] : (S[sl.o_scalars + i] & 0xFFFFFFFFull);
        for (int i = 0; i <= BQC_COVSIZE; ++i) L.poscov[i] = S[sl.o_poscov + i];
        if (S[sl.o_covstart] == 0) L.poscov[0] += 2 * BQC_VSIZE; // lane never saw coverage(): final flush of two empty windows
        for (int i = 0; i <= BQC_COVSIZE; ++i) L.poscov[i] &= 0xFFFFFFFFull;
        L.eightmer = S + sl.o_eightmer;
        L.triplet = S + sl.o_triplet;
        for (uint32_t m = 0; m < 2; ++m) {
            const uint64_t* Mq = S + sl.o_mate[m];
            bqc_mate_counts& mc = L.mate[m];
            const uint32_t n_rl = last_nonzero_len(Mq + sl.m_readlen, sl.lcap + 1); // maxL + 1, or 0 when no read
            const uint32_t ncyc = n_rl ? n_rl - 1 : 0;
            mc.n_cycles = ncyc;
            for (int j = 0; j < 5; ++j) mc.dnacount[j] = Mq + sl.m_dnacount + (uint64_t)j * sl.lcap;
            mc.qualcount = Mq + sl.m_qualcount;
            mc.qualcount_readnr = Mq[sl.m_readnr] & 0xFFFFFFFFull;
            { // sc5[j] = #reads whose leading clip exceeds j (suffix sum of the clip-length histogram)
                std::vector<uint64_t> v(ncyc ? ncyc : 1, 0);
                uint64_t run = 0;
                for (uint32_t j = sl.lcap + 1; j-- > 0;) {
                    if (j < ncyc) v[j] = run & 0xFFFFFFFFull; // run = sum_{n > j} H[n]
                    run += Mq[sl.m_sc5hist + j];
                }
                // v[j] = sum_{n > j} H[n]
                mc.sc5 = keep(std::move(v));
            }
            { // sc3 = prefix sum of the difference array
                std::vector<uint64_t> v(ncyc ? ncyc : 1, 0);
                uint64_t run = 0;
                for (uint32_t j = 0; j < ncyc; ++j) { run += Mq[sl.m_sc3diff + j]; v[j] = run & 0xFFFFFFFFull; }
                mc.sc3 = keep(std::move(v));
            }
            mc.n_Ncount = n_rl; mc.Ncount = keep(u32copy(Mq + sl.m_ncount, n_rl));
            mc.n_GCcount = n_rl; mc.GCcount = Mq + sl.m_gccount;
            mc.n_averageQual = last_nonzero_len(Mq + sl.m_avgceil, 256);
            mc.averageQual = keep(u32copy(Mq + sl.m_avgqual, mc.n_averageQual));
            mc.n_insertSize = sl.icap; mc.insertSize = keep(u32copy(Mq + sl.m_insert, sl.icap));
            mc.n_mapQ = last_nonzero_len(Mq + sl.m_mapq, 256); mc.mapQ = keep(u32copy(Mq + sl.m_mapq, mc.n_mapQ));
            mc.n_readLength = n_rl; mc.readLength = keep(u32copy(Mq + sl.m_readlen, n_rl));
            mc.n_mismatch = last_nonzero_len(Mq + sl.m_mismatch, sl.hcap); mc.mismatch = keep(u32copy(Mq + sl.m_mismatch, mc.n_mismatch));
            mc.n_delhist = last_nonzero_len(Mq + sl.m_delhist, sl.hcap); mc.delhist = keep(u32copy(Mq + sl.m_delhist, mc.n_delhist));
            mc.n_inshist = last_nonzero_len(Mq + sl.m_inshist, sl.hcap); mc.inshist = keep(u32copy(Mq + sl.m_inshist, mc.n_inshist));
        }
        if (c->sketch) {
            std::string e;
            if (!sketch_finalize(c->sketch, l, c->sk_out[l], c->stream, e)) return fail(c, BQC_ERR_DEVICE, "sketch finalize: %s", e.c_str());
            L.n_sketch = (uint32_t)c->sk_out[l].size();
            L.sketch = c->sk_out[l].data();
        }
    }
    c->counts.n_lanes = sl.n_lanes;
    c->counts.lanes = c->lanes.data();
    *out = &c->counts;
    return 0;
}
