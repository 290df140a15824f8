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
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "bqc_ctx.h"
#include "../host/parallel.h"

extern "C" {
void bqc_launch_cov_final(const StateLayout&, uint64_t*, const uint32_t* carry, const uint32_t* parity, const uint8_t* started, const uint8_t* sel, uint32_t count_start, hipStream_t);
void bqc_launch_ref_nibbles(const uint8_t* dna5, uint64_t len, uint32_t* out, uint64_t n_dwords, hipStream_t);
hipError_t bqc_long_init();
hipError_t bqc_short_init();
}

static thread_local char g_create_err[512];

int bqc_fail(bqc_ctx* c, int code, const char* fmt, ...)
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
#define fail bqc_fail

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

// the records that follow the stream of batches: merged error record (none), FASTA cursor (-1: before the first contig)
static int reset_stream_records(bqc_ctx* c)
{
    ErrRec e{};
    e.first_key = BQC_ERRKEY_NONE;
    const int32_t cur[2] = {-1, -1}; // last / first FASTA position of the stream's triplet-eligible reads
    HIPCHK(c, hipMemcpyAsync(c->d_err0, &e, sizeof e, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_cursor, cur, 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream)); // (the sources are on this function's stack)
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
    c->shard.tail = opt->shard_tail != 0;
    c->shard.pending.assign(opt->n_lanes, c->shard.tail ? 1 : 0);
    c->shard.has_prev.assign(opt->n_lanes, 0);
    c->shard.prev_rid.assign(opt->n_lanes, 0);
    c->shard.prev_bp.assign(opt->n_lanes, 0);
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
    const bool ctiming = getenv("BQC_TIMING") && getenv("BQC_TIMING")[0] == '1';
    const auto ct0 = std::chrono::steady_clock::now();
    double cts[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto cstamp = [&](int k) { cts[k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - ct0).count() * 1e3; };
    CCHK(hipSetDevice(c->device));
    hipDeviceProp_t prop;
    CCHK(hipGetDeviceProperties(&prop, c->device));
    c->n_cu = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256;
    // (the kernels' attributes — hipFuncSetAttribute for k_short's and k_long's LDS — are set at their first launch: the first such call
    // loads the code object, 40-60 ms that sat between the runtime's start and the first copy of a run here)
    cstamp(0);
    c->stream = bqc_pool_stream(c->device, 0); // (made ahead by bqc_warmup, or now; the copy stream: at the first submit, bqc_copy_stream)
    if (!c->stream) { fail(nullptr, BQC_ERR_DEVICE, "bqc_create: hipStreamCreate failed"); bqc_destroy(c); return BQC_ERR_DEVICE; }
    cstamp(1);
    CCHK(hipMalloc(&c->d_state, c->sl.words * 8));
    CCHK(hipMalloc(&c->d_err0, sizeof(ErrRec)));
    CCHK(hipMalloc(&c->d_cursor, 8));
    cstamp(6);
    if (!c->fasta_index.empty()) {
        CCHK(hipMalloc(&c->d_fasta_index, 4 * c->fasta_index.size()));
        CCHK(hipMemcpy(c->d_fasta_index, c->fasta_index.data(), 4 * c->fasta_index.size(), hipMemcpyHostToDevice));
    }
    cstamp(7);
    c->t8_slots_cap = std::max(1024u, 4u * c->n_cu);
    CCHK(hipMalloc(&c->d_t8rows, (size_t)c->t8_slots_cap * BQC_T8_SPW * 65536));
    CCHK(hipMalloc(&c->d_t8used, (size_t)c->t8_slots_cap * BQC_T8_USED * 4));
    cstamp(8);
    CCHK(hipMalloc(&c->d_kl_cyc, (size_t)c->n_cu * 2 * 6 * 1024 * 4));
    CCHK(hipMalloc(&c->d_kl_cyc_used, (size_t)c->n_cu * 4));
    CCHK(hipMalloc(&c->d_carry, (size_t)opt->n_lanes * 2 * 2000 * 4));
    CCHK(hipMalloc(&c->d_parity, ((size_t)opt->n_lanes + 1) * 4));
    CCHK(hipMalloc(&c->d_started, opt->n_lanes));
    CCHK(hipMalloc(&c->d_ref_ptrs, sizeof(uint8_t*) * nr));
    CCHK(hipMalloc(&c->d_ref_len, sizeof(uint64_t) * nr));
    CCHK(hipMalloc(&c->d_refn_ptrs, sizeof(uint32_t*) * nr));
    CCHK(hipMalloc(&c->d_main, nr));
    cstamp(9);
    CCHK(hipMemcpy(c->d_main, c->main_chrom.data(), nr, hipMemcpyHostToDevice));
    cstamp(2);
    CCHK(hipMemsetAsync(c->d_state, 0, c->sl.words * 8, c->stream));
    if (reset_stream_records(c)) { snprintf(g_create_err, sizeof g_create_err, "%s", c->err.c_str()); bqc_destroy(c); return BQC_ERR_DEVICE; }
    CCHK(hipMemsetAsync(c->d_carry, 0, (size_t)opt->n_lanes * 2 * 2000 * 4, c->stream));
    CCHK(hipMemsetAsync(c->d_parity, 0, ((size_t)opt->n_lanes + 1) * 4, c->stream));
    CCHK(hipMemsetAsync(c->d_started, 0, opt->n_lanes, c->stream));
    CCHK(hipStreamSynchronize(c->stream));
    if (upload_ref_tables(c)) { snprintf(g_create_err, sizeof g_create_err, "%s", c->err.c_str()); bqc_destroy(c); return BQC_ERR_DEVICE; }
    cstamp(3);
    if (opt->sketch.n_k && opt->sketch.n_q) {
        std::string e;
        c->sketch = sketch_create(opt->sketch, opt->n_lanes, c->stream, e);
        if (!c->sketch) { fail(nullptr, BQC_ERR_ARG, "bqc_create: sketch: %s", e.c_str()); bqc_destroy(c); return BQC_ERR_ARG; }
    }
    cstamp(4);
    for (int i = 0; i < 16; ++i) {
        hipEvent_t e;
        CCHK(hipEventCreate(&e));
        c->ev.push_back(e);
    }
    cstamp(5);
    if (ctiming) fprintf(stderr, "[timing] bqc_create: kernel attributes (code object load) at %.1f ms, compute stream %.1f, allocations %.1f (first three %.1f, FASTA index copied %.1f, 8-mer rows %.1f, the rest %.1f, then a copy), memsets + tables %.1f, sketch %.1f, events %.1f\n", cts[0], cts[1], cts[2], cts[6], cts[7], cts[8], cts[9], cts[3], cts[4], cts[5]);
    *out = c;
    return 0;
}

// Streams made ahead.  On this card a process's first stream costs 20-25 ms and each of the next three 8-10 (a hardware queue each:
// tools/micro/startup_probe.cpp, profiles/r4_startup_probe.txt), one after the other whichever threads ask — 45-50 ms that the program's
// reader (two streams) and the context (two) used to pay in turn, between the runtime's start and the first kernel.  bqc_warmup makes
// them in ITS thread right behind the runtime's start, while the caller's other threads allocate, page-lock and parse; whoever needs
// a stream takes one from here (waiting for the warm-up if it is still at it), or creates it when there is none.
namespace {
struct StreamPool {
    std::mutex m;
    std::condition_variable cv;
    static const int kAhead = 4;
    hipStream_t made[kAhead] = {nullptr, nullptr, nullptr, nullptr};
    bool taken[kAhead] = {false, false, false, false};
    int device = -1;
    int n_made = 0;
    bool making = false; // the warm-up thread is still at it
};
StreamPool g_streams;
}
// `rank`: the order in which the program needs its streams — 0 the context's compute stream (its creation, the references), 1 the
// reader's producer (first copy, first inflate), 2 the reader's consumer (first walk), 3 the context's copy stream (first submit):
// a caller waits for the stream of ITS rank, not for whichever comes next.
hipStream_t bqc_pool_stream(int device, int rank)
{
    {
        std::unique_lock<std::mutex> lk(g_streams.m);
        if (g_streams.device == device && rank >= 0 && rank < StreamPool::kAhead) {
            g_streams.cv.wait(lk, [&] { return g_streams.n_made > rank || !g_streams.making; });
            if (g_streams.n_made > rank && !g_streams.taken[rank]) { g_streams.taken[rank] = true; return g_streams.made[rank]; }
        }
    }
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return s;
}

extern "C" int bqc_warmup(int32_t device)
{
    {
        std::lock_guard<std::mutex> lk(g_streams.m);
        if (g_streams.device != -1) return g_streams.device == device ? 0 : BQC_ERR_ARG; // (once per process)
        g_streams.device = device;
        g_streams.making = true;
    }
    auto done = [](int rc) { { std::lock_guard<std::mutex> lk(g_streams.m); g_streams.making = false; } g_streams.cv.notify_all(); return rc; };
    if (hipSetDevice(device) != hipSuccess || hipFree(nullptr) != hipSuccess) return done(BQC_ERR_DEVICE);
    // (Tried in round 4: the first copy from pageable memory — ~50 ms of runtime set-up that bqc_create pays behind its compute stream,
    // `[timing] bqc_create` — made here by a thread of its own beside the streams: the record loop starts at 0.19 s either way,
    // gpurun_out/r9d; the runtime serialises the two.)
    for (int k = 0; k < StreamPool::kAhead; ++k) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return done(BQC_ERR_DEVICE); }
        { std::lock_guard<std::mutex> lk(g_streams.m); g_streams.made[k] = s; g_streams.n_made = k + 1; }
        g_streams.cv.notify_all();
    }
    return done(0);
}

extern "C" int bqc_set_fasta_index(bqc_ctx* c, const int32_t* idx)
{
    if (!c || !idx) return BQC_ERR_ARG;
    if (c->upload_counter) return bqc_fail(c, BQC_ERR_STATE, "bqc_set_fasta_index after the first batch");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = c->opt.n_refs ? c->opt.n_refs : 1;
    c->fasta_index.assign(n, -1);
    for (uint32_t r = 0; r < c->opt.n_refs; ++r) c->fasta_index[r] = idx[r];
    c->opt.fasta_index = c->fasta_index.data();
    if (!c->d_fasta_index) HIPCHK(c, hipMalloc(&c->d_fasta_index, 4 * n));
    HIPCHK(c, hipMemcpy(c->d_fasta_index, c->fasta_index.data(), 4 * n, hipMemcpyHostToDevice));
    return 0;
}

extern "C" void bqc_destroy(bqc_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    bqc_pipeline_destroy(c);
    bqc_anchor_destroy(c);
    auto in_arena = [&](const void* p) { return c->ref_arena && (const uint8_t*)p >= c->ref_arena && (const uint8_t*)p < c->ref_arena + c->ref_arena_cap; };
    for (auto p : c->d_ref) if (p && !in_arena(p)) (void)hipFree(p);
    for (auto p : c->d_refn) if (p && !in_arena(p)) (void)hipFree(p);
    if (c->ref_arena) (void)hipFree(c->ref_arena);

    (void)hipFree(c->d_refn_ptrs);
    if (c->sketch) sketch_destroy(c->sketch);
    (void)hipFree(c->d_state); (void)hipFree(c->d_err0); (void)hipFree(c->d_cursor); (void)hipFree(c->d_fasta_index); (void)hipFree(c->d_t8rows); (void)hipFree(c->d_t8used); (void)hipFree(c->d_kl_cyc); (void)hipFree(c->d_kl_cyc_used); (void)hipFree(c->d_carry); (void)hipFree(c->d_parity);
    (void)hipFree(c->d_started); (void)hipFree(c->d_ref_ptrs); (void)hipFree(c->d_ref_len); (void)hipFree(c->d_main);
    for (auto e : c->ev) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    delete c;
}

extern "C" int bqc_reserve_references(bqc_ctx* c, uint64_t total_bases, uint32_t n_contigs)
{
    if (!c) return BQC_ERR_ARG;
    if (c->ref_arena) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    // per contig: len bytes of codes + (len / 8 + pads) dwords of nibbles, each rounded up to 256 bytes
    const size_t cap = (size_t)total_bases + (size_t)total_bases / 2 + (size_t)n_contigs * (8 * BQC_FAST_NH + 1024) + 4096;
    if (hipMalloc((void**)&c->ref_arena, cap) != hipSuccess) { (void)hipGetLastError(); c->ref_arena = nullptr; return 0; } // (not fatal: the contigs are then allocated one by one)
    c->ref_arena_cap = cap;
    c->ref_arena_used = 0;
    return 0;
}

extern "C" int bqc_set_reference(bqc_ctx* c, int32_t rid, const uint8_t* dna5, uint64_t len)
{
    if (!c || rid < 0 || (uint32_t)rid >= c->opt.n_refs || (!dna5 && len)) return fail(c, BQC_ERR_ARG, "bqc_set_reference: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    auto in_arena = [&](const void* p) { return c->ref_arena && (const uint8_t*)p >= c->ref_arena && (const uint8_t*)p < c->ref_arena + c->ref_arena_cap; };
    if (c->d_ref[rid]) { if (!in_arena(c->d_ref[rid])) HIPCHK(c, hipFree(c->d_ref[rid])); c->d_ref[rid] = nullptr; }
    if (c->d_refn[rid]) { if (!in_arena(c->d_refn[rid])) HIPCHK(c, hipFree(c->d_refn[rid])); c->d_refn[rid] = nullptr; }
    const uint64_t nd8 = (len + 7) / 8; // nibble table of the fast path: BQC_FAST_NH pad dwords + nd8 + BQC_FAST_NH pad dwords
    const size_t b1 = ((len ? len : 1) + 255) & ~(size_t)255, b2 = ((nd8 + 2 * BQC_FAST_NH) * 4 + 255) & ~(size_t)255;
    uint8_t* p = nullptr;
    uint32_t* pn = nullptr;
    if (c->ref_arena && c->ref_arena_used + b1 + b2 <= c->ref_arena_cap) {
        p = c->ref_arena + c->ref_arena_used;
        pn = (uint32_t*)(p + b1);
        c->ref_arena_used += b1 + b2;
    } else {
        HIPCHK(c, hipMalloc(&p, len ? len : 1));
        HIPCHK(c, hipMalloc(&pn, (nd8 + 2 * BQC_FAST_NH) * 4));
    }
    // (a plain synchronous copy from the caller's pageable memory: 8.9 GB/s for a human genome's 3.1 GB — measured in round 3 against
    // hipMemcpyAsync on a stream of its own, 4.8 GB/s, and against page-locking the contigs first, 4-5 GB/s all told)
    HIPCHK(c, hipMemcpy(p, dna5, len, hipMemcpyHostToDevice));
    bqc_launch_ref_nibbles(p, len, pn, nd8, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->d_ref[rid] = p;
    c->d_refn[rid] = pn;
    c->ref_len[rid] = len;
    return upload_ref_tables(c);
}

// Wait for everything submitted and report the first error of the stream (as the reference would have met it): batches of the
// submit pipeline through their own records, resident batches (bqc_process) through the context's merged record.
static int sync_and_check(bqc_ctx* c)
{
    HIPCHK(c, hipSetDevice(c->device));
    if (c->poisoned) return bqc_fail(c, BQC_ERR_STATE, "context is in an error state: %s", c->err.c_str());
    int rc = bqc_drain(c);
    if (rc) return rc;
    ErrRec e;
    HIPCHK(c, hipMemcpyAsync(&e, c->d_err0, sizeof e, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    rc = bqc_report_errors(c, e);
    if (rc) { c->poisoned = true; c->poison_code = rc; }
    return rc;
}

extern "C" int bqc_sync(bqc_ctx* c)
{
    if (!c) return BQC_ERR_ARG;
    return sync_and_check(c);
}

extern "C" int bqc_reset(bqc_ctx* c)
{
    if (!c) return BQC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    (void)bqc_drain(c); // (whatever the batches in flight found is forgotten with the counters)
    bqc_state_ready(c);
    HIPCHK(c, hipMemsetAsync(c->d_state, 0, c->sl.words * 8, c->stream));
    if (reset_stream_records(c)) return BQC_ERR_DEVICE;
    HIPCHK(c, hipMemsetAsync(c->d_carry, 0, (size_t)c->opt.n_lanes * 2 * 2000 * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_parity, 0, ((size_t)c->opt.n_lanes + 1) * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_started, 0, c->opt.n_lanes, c->stream));
    if (c->sketch) sketch_reset(c->sketch, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->cov.assign(c->opt.n_lanes, LaneCov());
    if (c->anchor.d_state) { // (anchors made on the card: the read group's state starts over as well)
        AnchorState s0{};
        s0.first = 1;
        s0.pending = c->shard.tail && !c->shard.resolved ? 1u : 0u;
        HIPCHK(c, hipMemcpy(c->anchor.d_state, &s0, sizeof s0, hipMemcpyHostToDevice));
    }
    c->anchor.mode = 0;
    c->state_seq = 0;
    c->flushed = false;
    c->poisoned = false;
    c->poison_code = 0;
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
    int rc = sync_and_check(c);
    if (rc) return rc;
    bqc_state_ready(c);
    bqc_launch_cov_final(c->sl, c->d_state, c->d_carry, c->d_parity, c->d_started, nullptr, 1, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->flushed = true;
    return 0;
}

extern "C" uint64_t bqc_state_words(const bqc_ctx* c) { return c ? c->sl.words + (c->sketch ? sketch_state_words(c->sketch) : 0) : 0; }

extern "C" int bqc_state_export(bqc_ctx* c, void* dst)
{
    if (!c || !dst) return BQC_ERR_ARG;
    int rc = bqc_flush(c);
    if (rc) return rc;
    bqc_state_ready(c);
    HIPCHK(c, hipMemcpyAsync(dst, c->d_state, c->sl.words * 8, hipMemcpyDeviceToDevice, c->stream));
    if (c->sketch) sketch_state_export(c->sketch, (uint64_t*)dst + c->sl.words, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int bqc_state_import(bqc_ctx* c, const void* src)
{
    if (!c || !src) return BQC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    (void)bqc_drain(c);
    bqc_state_ready(c);
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
    bqc_state_ready(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
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
    (void)bqc_drain(c);
    bqc_state_ready(c);
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
