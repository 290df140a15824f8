// multi_gpu.cpp — `bamqualcheck --gpus N`: the program over the N GPUs of one node as ONE binary, like the reference's single
// main() (src/bamqualcheck.cpp:239-457), without Python or torch in the product.
//
//   front end (no GPU call ever) ──fork──> worker 0 .. N-1, one per GPU, each = bqc_main_shard(i, N) over ITS byte range of the
//   BAM file (driver.cpp; the range is read, inflated and decoded on the worker's own card: csrc/gpu_bam.hip set_range)
//
// What needs the other workers happens in the shard hook, once per run, after the record loops:
//   * a few words per worker through the front end (socket pairs): exit status, split check (range_over of a shard ==
//     range_first of its successor), FASTA order, lane names — agreed on BEFORE any data moves, so that an error on one
//     worker ends all of them with status 1;
//   * the coverage state machine's state down the chain of workers (8 KB per read group: bqc_shard_export -> bqc_shard_resolve),
//     relayed by the front end;
//   * ONE sum of the flat state vectors onto worker 0: ncclReduce(uint64, sum) over RCCL / xGMI on device pointers
//     (bqc_state_export -> ncclReduce -> bqc_state_import).  librccl is loaded at run time (dlopen) by the workers only, and its
//     communicator is set up by a thread of its own while the record loop runs.  BQC_REDUCE=pipe (or workers that share one
//     card, BQC_GPUS_SHARE_DEVICE=1: RCCL refuses two ranks on a device) sums host vectors in the front end instead.
// Worker 0 then finalises and writes the output, as the single-GPU program does.
#include <dlfcn.h>
#include <poll.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h> // types only: the functions are looked up in librccl at run time

#include "../../include/bamqc_host.h"

namespace {
// ---- framed messages over a socket pair --------------------------------------------------------------------------------
bool send_all(int fd, const void* p, size_t n)
{
    const char* c = (const char*)p;
    while (n) {
        const ssize_t k = send(fd, c, n, MSG_NOSIGNAL);
        if (k < 0) { if (errno == EINTR) continue; return false; }
        c += k; n -= (size_t)k;
    }
    return true;
}
bool recv_all(int fd, void* p, size_t n)
{
    char* c = (char*)p;
    while (n) {
        const ssize_t k = recv(fd, c, n, 0);
        if (k == 0) return false;
        if (k < 0) { if (errno == EINTR) continue; return false; }
        c += k; n -= (size_t)k;
    }
    return true;
}
bool send_msg(int fd, const std::vector<uint8_t>& m)
{
    const uint64_t n = m.size();
    return send_all(fd, &n, 8) && (n == 0 || send_all(fd, m.data(), m.size()));
}
bool recv_msg(int fd, std::vector<uint8_t>& m)
{
    uint64_t n = 0;
    if (!recv_all(fd, &n, 8) || n > (1ull << 34)) return false;
    m.resize((size_t)n);
    return n == 0 || recv_all(fd, m.data(), m.size());
}
struct Writer {
    std::vector<uint8_t> b;
    template <typename T> void put(T v) { const uint8_t* p = (const uint8_t*)&v; b.insert(b.end(), p, p + sizeof v); }
    void str(const std::string& s) { put<uint32_t>((uint32_t)s.size()); b.insert(b.end(), s.begin(), s.end()); }
    void bytes(const void* p, size_t n) { put<uint64_t>(n); b.insert(b.end(), (const uint8_t*)p, (const uint8_t*)p + n); }
};
struct Reader {
    const std::vector<uint8_t>& b;
    size_t at = 0;
    bool ok = true;
    template <typename T> T get() { T v{}; if (at + sizeof v > b.size()) { ok = false; return v; } memcpy(&v, b.data() + at, sizeof v); at += sizeof v; return v; }
    std::string str() { const uint32_t n = get<uint32_t>(); if (!ok || at + n > b.size()) { ok = false; return {}; } std::string s((const char*)b.data() + at, n); at += n; return s; }
    std::vector<uint8_t> bytes() { const uint64_t n = get<uint64_t>(); if (!ok || at + n > b.size()) { ok = false; return {}; } std::vector<uint8_t> v(b.begin() + at, b.begin() + at + n); at += n; return v; }
};

enum Verdict : int32_t { V_FAIL = 0, V_DONE = 1, V_FALLBACK = 2, V_GO = 3 };

// ---- what a worker tells the front end after its record loop -----------------------------------------------------------
struct WorkerInfo {
    int32_t status = 1, has_ctx = 0;
    uint64_t b0 = 0, b1 = 0, first = 0, over = 0;
    int32_t span[2] = {-1, -1};
    std::vector<std::pair<std::string, uint32_t>> lanes;
};

// bamqc_amd/distributed.py: split_is_consistent / fasta_order_is_consistent, restated
bool split_is_consistent(const std::vector<WorkerInfo>& w, uint64_t file_size)
{
    if (file_size == 0) return false; // (a size that is not known verifies nothing: one process takes the file)
    const WorkerInfo* prev = nullptr;
    for (const WorkerInfo& e : w) {
        if (!(e.b0 < std::min(e.b1, file_size))) continue; // (a shard without blocks: its neighbours meet directly)
        if (prev && (prev->b1 != e.b0 || prev->over != e.first)) return false;
        prev = &e;
    }
    return true;
}
bool fasta_order_is_consistent(const std::vector<WorkerInfo>& w)
{
    int32_t last = -1;
    for (const WorkerInfo& e : w) {
        if (e.span[0] < 0) continue;
        if (e.span[0] < last) return false;
        last = std::max(last, e.span[1]);
    }
    return true;
}

// ---- RCCL, looked up at run time -----------------------------------------------------------------------------------------
struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& err)
    {
        for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
            so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (so) break;
        }
        if (!so) { err = std::string("librccl could not be loaded: ") + dlerror(); return false; }
        GetUniqueId = (decltype(GetUniqueId))dlsym(so, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(so, "ncclCommInitRank");
        Reduce = (decltype(Reduce))dlsym(so, "ncclReduce");
        GetErrorString = (decltype(GetErrorString))dlsym(so, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !Reduce || !GetErrorString) { err = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclReduce"; return false; }
        return true;
    }
};

// ---- a worker ------------------------------------------------------------------------------------------------------------
struct Worker {
    int fd = -1;
    uint32_t rank = 0, world = 1;
    int device = 0;
    bool use_rccl = false;
    Rccl rccl;
    ncclComm_t comm = nullptr;
    std::thread init;        // sets the communicator up while the record loop runs
    std::string init_err;    // (read after init has been joined)
    // what the hook hands back to run_program for BQC_SHARD_WRITE
    std::vector<std::string> names;
    std::vector<const char*> name_ptrs;
    std::vector<uint32_t> index;

    // BQC_TEST_WORKER_FAULT="<rank>:<what>" (tests/test_gpu_sharded.py): worker <rank> fails the way <what> names
    bool test_fault(const char* what) const
    {
        const char* e = getenv("BQC_TEST_WORKER_FAULT");
        if (!e) return false;
        const char* colon = strchr(e, ':');
        return colon && (uint32_t)atoi(e) == rank && strcmp(colon + 1, what) == 0;
    }

    void start_rccl()
    {
        init = std::thread([this] {
            // Whatever fails here, the exchange of the id below takes place (the front end and the other workers count on it)
            // and the failure is reported by the hook, where every worker's status is agreed on.
            bool ok = rccl.load(init_err);
            if (ok && hipSetDevice(device) != hipSuccess) { init_err = "hipSetDevice failed"; ok = false; }
            ncclUniqueId id;
            memset(&id, 0, sizeof id);
            // the id is made by worker 0 and handed round by the front end (this thread is the only user of the socket until it ends)
            if (rank == 0) {
                if (ok) {
                    const ncclResult_t r = rccl.GetUniqueId(&id);
                    if (r != ncclSuccess) { init_err = std::string("ncclGetUniqueId: ") + rccl.GetErrorString(r); ok = false; }
                }
                std::vector<uint8_t> m((const uint8_t*)&id, (const uint8_t*)&id + sizeof id);
                m.push_back(ok ? 1 : 0);
                if (!send_msg(fd, m)) { init_err = "the front end is gone"; return; }
            }
            std::vector<uint8_t> m;
            if (!recv_msg(fd, m) || m.size() != sizeof id + 1) { init_err = "the front end is gone"; return; }
            if (!m.back()) { if (init_err.empty()) init_err = "worker 0 could not create the RCCL id"; return; }
            // A worker that cannot join the communicator ends at once: the others wait for it inside ncclCommInitRank, where no
            // message reaches them — the front end sees this process go and ends them.
            auto die = [&]() { fprintf(stderr, "ERROR: worker %u: %s\n", rank, init_err.c_str()); fflush(stderr); _exit(1); };
            if (!ok) die();
            memcpy(&id, m.data(), sizeof id);
            const ncclResult_t r = rccl.CommInitRank(&comm, (int)world, id, (int)rank);
            if (r != ncclSuccess) { init_err = std::string("ncclCommInitRank: ") + rccl.GetErrorString(r); die(); }
        });
    }

    int hook(const bqc_shard_info* info, bqc_shard_result* out)
    {
        if (init.joinable()) init.join();
        bqc_ctx* ctx = info->ctx;
        int32_t status = info->status;
        if (use_rccl && !init_err.empty()) { fprintf(stderr, "ERROR: %s\n", init_err.c_str()); status = 1; }
        int32_t span[2] = {-1, -1};
        if (ctx && !status && bqc_shard_fasta_span(ctx, span)) { fprintf(stderr, "%s\n", bqc_last_error(ctx)); status = 1; }
        { // ---- everybody's status, ranges and lane names: agreed on before any data moves
            Writer w;
            w.put<int32_t>(status); w.put<int32_t>(ctx ? 1 : 0);
            w.put<uint64_t>(info->begin_block); w.put<uint64_t>(info->end_block); w.put<uint64_t>(info->first); w.put<uint64_t>(info->over);
            w.put<int32_t>(span[0]); w.put<int32_t>(span[1]);
            w.put<uint32_t>(info->n_lane_names);
            for (uint32_t i = 0; i < info->n_lane_names; ++i) { w.str(info->lane_names[i]); w.put<uint32_t>(info->lane_index[i]); }
            if (!send_msg(fd, w.b)) return BQC_SHARD_FAIL;
        }
        std::vector<uint8_t> m;
        auto verdict = [&]() -> int32_t { // the front end's decision after a phase
            if (!recv_msg(fd, m) || m.size() < 4) return V_FAIL;
            int32_t v;
            memcpy(&v, m.data(), 4);
            return v;
        };
        int32_t v = verdict();
        if (v == V_FAIL) return BQC_SHARD_FAIL;
        if (v == V_DONE) return BQC_SHARD_DONE;
        if (v == V_FALLBACK) return BQC_SHARD_FALLBACK;
        // ---- coverage state down the chain: predecessor's final state in, own final state out
        const uint64_t nbytes = bqc_shard_state_bytes(ctx);
        int rc = 0;
        if (rank > 0) {
            if (!recv_msg(fd, m) || m.size() != nbytes) return BQC_SHARD_FAIL;
            rc = bqc_shard_resolve(ctx, m.data());
        }
        std::vector<uint8_t> state((size_t)(rank + 1 < world ? nbytes : 0), 0);
        if (!rc) rc = rank + 1 < world ? bqc_shard_export(ctx, state.data()) : bqc_flush(ctx); // (the last shard ends as a whole stream does)
        if (rc) { fprintf(stderr, "%s\n", bqc_last_error(ctx)); std::fill(state.begin(), state.end(), 0); }
        {
            Writer w;
            w.put<int32_t>(rc ? 1 : 0);
            w.bytes(state.data(), state.size()); // (sent also after a failure: the successor is waiting)
            if (!send_msg(fd, w.b)) return BQC_SHARD_FAIL;
        }
        if (verdict() != V_GO) return BQC_SHARD_FAIL;
        // ---- ONE sum of the flat state vectors onto worker 0
        const uint64_t words = bqc_state_words(ctx);
        rc = 0;
        if (test_fault("wedge")) for (;;) pause(); // (test: a worker that never gets to the sum — the front end's deadline ends the run)
        if (use_rccl) {
            void* dvec = nullptr;
            hipStream_t s = nullptr;
            if (test_fault("nomem") || hipSetDevice(device) != hipSuccess || hipMalloc(&dvec, words * 8) != hipSuccess || hipMemset(dvec, 0, words * 8) != hipSuccess ||
                hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
                // A worker that cannot enter the collective ends at once, as in start_rccl: its peers are inside ncclReduce, where no
                // message reaches them — the front end sees this process go (SIGCHLD) and ends them.
                fprintf(stderr, "ERROR: worker %u: no device memory for the state vector\n", rank); fflush(stderr);
                _exit(1);
            }
            if ((rc = bqc_state_export(ctx, dvec))) fprintf(stderr, "%s\n", bqc_last_error(ctx));
            // (every worker that got here enters the collective, also after a failed export: the others are in it)
            {
                const ncclResult_t r = rccl.Reduce(dvec, dvec, (size_t)words, ncclUint64, ncclSum, 0, comm, s);
                if (r != ncclSuccess) { fprintf(stderr, "ERROR: ncclReduce: %s\n", rccl.GetErrorString(r)); rc = 1; }
                if (hipStreamSynchronize(s) != hipSuccess) { fprintf(stderr, "ERROR: the RCCL reduce failed on the device\n"); rc = 1; }
            }
            { Writer w; w.put<int32_t>(rc ? 1 : 0); if (!send_msg(fd, w.b)) return BQC_SHARD_FAIL; }
            if (verdict() != V_GO) return BQC_SHARD_FAIL; // (nobody imports or writes a sum that lacks a worker's part)
            if (rank == 0 && (rc = bqc_state_import(ctx, dvec))) fprintf(stderr, "%s\n", bqc_last_error(ctx));
            if (s) (void)hipStreamDestroy(s);
            if (dvec) (void)hipFree(dvec);
        } else {
            std::vector<uint64_t> host((size_t)words, 0);
            if ((rc = bqc_state_export_host(ctx, host.data()))) fprintf(stderr, "%s\n", bqc_last_error(ctx));
            {
                Writer w;
                w.put<int32_t>(rc ? 1 : 0);
                if (rank > 0) w.bytes(host.data(), host.size() * 8); else w.bytes(nullptr, 0);
                if (!send_msg(fd, w.b)) return BQC_SHARD_FAIL;
            }
            if (verdict() != V_GO) return BQC_SHARD_FAIL;
            if (rank == 0) { // the others' sum comes back with the verdict
                if (m.size() != 4 + words * 8) return BQC_SHARD_FAIL;
                const uint8_t* p = m.data() + 4;
                for (uint64_t i = 0; i < words; ++i) { uint64_t x; memcpy(&x, p + 8 * i, 8); host[i] += x; }
                if ((rc = bqc_state_import_host(ctx, host.data()))) fprintf(stderr, "%s\n", bqc_last_error(ctx));
            }
        }
        { Writer w; w.put<int32_t>(rc ? 1 : 0); if (!send_msg(fd, w.b)) return BQC_SHARD_FAIL; }
        if (verdict() != V_GO) return BQC_SHARD_FAIL;
        if (rank != 0) return BQC_SHARD_DONE;
        // ---- worker 0 writes: the merged lane names (output order) come with the last verdict
        Reader r{m};
        (void)r.get<int32_t>();
        const uint32_t n = r.get<uint32_t>();
        names.clear(); index.clear();
        for (uint32_t i = 0; i < n && r.ok; ++i) { names.push_back(r.str()); index.push_back(r.get<uint32_t>()); }
        if (!r.ok) return BQC_SHARD_FAIL;
        name_ptrs.clear();
        for (const std::string& s : names) name_ptrs.push_back(s.c_str());
        out->n_lane_names = n;
        out->lane_names = name_ptrs.data();
        out->lane_index = index.data();
        return BQC_SHARD_WRITE;
    }
};

int worker_hook(void* user, const bqc_shard_info* info, bqc_shard_result* out) { return ((Worker*)user)->hook(info, out); }

// ---- the front end's side ---------------------------------------------------------------------------------------------------
std::vector<pid_t> g_children;
std::vector<int> g_exit;                 // exit status of a worker that has ended (-1: still running)
volatile sig_atomic_t g_worker_failed = 0; // a worker ended with a status other than 0 (or by a signal)
void forward_signal(int sig)
{
    for (size_t i = 0; i < g_children.size(); ++i) if (g_children[i] > 0 && g_exit[i] < 0) kill(g_children[i], sig);
}
void reap(int)
{
    for (;;) {
        int ws = 0;
        const pid_t p = waitpid(-1, &ws, WNOHANG);
        if (p <= 0) return;
        for (size_t i = 0; i < g_children.size(); ++i)
            if (g_children[i] == p) {
                g_exit[i] = WIFEXITED(ws) ? WEXITSTATUS(ws) : 128 + (WIFSIGNALED(ws) ? WTERMSIG(ws) : 0);
                if (g_exit[i] != 0) g_worker_failed = 1;
            }
    }
}
double now_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
// a message from a worker — or false as soon as any worker has failed (one that waits inside an RCCL call cannot be told), or when
// `deadline` (CLOCK_MONOTONIC seconds; 0: none) has passed: a worker wedged on its card never exits and never answers
bool recv_watching(int fd, std::vector<uint8_t>& m, double deadline)
{
    for (;;) {
        if (g_worker_failed) return false;
        if (deadline > 0 && now_s() > deadline) { fprintf(stderr, "bamqualcheck: a worker did not answer in time (BQC_GPUS_TIMEOUT / BQC_GPUS_LOOP_TIMEOUT); ending the run\n"); return false; }
        struct pollfd pf{fd, POLLIN, 0};
        const int k = poll(&pf, 1, 200);
        if (k < 0 && errno != EINTR) return false;
        if (k > 0) return recv_msg(fd, m);
    }
}

struct Coordinator {
    std::vector<int> fds;
    uint32_t world;
    bool use_rccl;
    uint64_t file_size;
    // Deadlines per phase: the record loops (phase 1) are as long as the file is — no limit unless BQC_GPUS_LOOP_TIMEOUT (seconds) sets
    // one; every later phase moves kilobytes and one reduce: BQC_GPUS_TIMEOUT seconds (default 300) from the end of the phase before.
    double loop_timeout = 0, phase_timeout = 300;
    double until(double secs) const { return secs > 0 ? now_s() + secs : 0; }
    bool tell_all(int32_t v)
    {
        bool ok = true;
        Writer w; w.put<int32_t>(v);
        for (int fd : fds) ok = send_msg(fd, w.b) && ok;
        return ok;
    }
    // false: a worker is gone or the run failed (everybody has been told where that was still possible)
    bool run()
    {
        std::vector<uint8_t> m;
        if (use_rccl) { // the communicator's id: from worker 0 to all of them
            if (!recv_watching(fds[0], m, until(phase_timeout))) return false;
            for (uint32_t i = 0; i < world; ++i) if (!send_msg(fds[i], m)) return false;
        }
        // ---- phase 1: status, ranges, lane names
        std::vector<WorkerInfo> info(world);
        bool any_gone = false;
        double dl = until(loop_timeout);
        for (uint32_t i = 0; i < world; ++i) {
            if (!recv_watching(fds[i], m, dl)) { any_gone = true; continue; }
            Reader r{m};
            WorkerInfo& e = info[i];
            e.status = r.get<int32_t>(); e.has_ctx = r.get<int32_t>();
            e.b0 = r.get<uint64_t>(); e.b1 = r.get<uint64_t>(); e.first = r.get<uint64_t>(); e.over = r.get<uint64_t>();
            e.span[0] = r.get<int32_t>(); e.span[1] = r.get<int32_t>();
            const uint32_t n = r.get<uint32_t>();
            for (uint32_t k = 0; k < n && r.ok; ++k) { std::string s = r.str(); const uint32_t ix = r.get<uint32_t>(); e.lanes.emplace_back(s, ix); }
            if (!r.ok) e.status = 1;
        }
        bool bad = any_gone;
        for (const WorkerInfo& e : info) bad = bad || e.status;
        if (bad) { tell_all(V_FAIL); return false; }
        bool any_ctx = false;
        for (const WorkerInfo& e : info) any_ctx = any_ctx || e.has_ctx;
        if (!any_ctx) return tell_all(V_DONE); // (no @RG line: there are no counters)
        if (!fasta_order_is_consistent(info)) {
            printf("ERROR: Could not read fasta record (the BAM file's contig order runs backwards in the FASTA file)\n");
            fflush(stdout);
            tell_all(V_FAIL);
            return false;
        }
        if (!split_is_consistent(info, file_size)) {
            printf("bamqualcheck: the split of the file could not be verified; processing it in one process\n");
            fflush(stdout);
            bool ok = true;
            for (uint32_t i = 0; i < world; ++i) { Writer w; w.put<int32_t>(i == 0 ? V_FALLBACK : V_DONE); ok = send_msg(fds[i], w.b) && ok; }
            return ok;
        }
        if (!tell_all(V_GO)) return false;
        // ---- phase 2: the coverage state down the chain
        bad = false;
        dl = until(phase_timeout);
        for (uint32_t i = 0; i < world; ++i) {
            if (!recv_watching(fds[i], m, dl)) return false;
            Reader r{m};
            bad = r.get<int32_t>() || bad;
            std::vector<uint8_t> state = r.bytes();
            if (!r.ok) return false;
            if (i + 1 < world && !send_msg(fds[i + 1], state)) return false;
        }
        if (bad) { tell_all(V_FAIL); return false; }
        if (!tell_all(V_GO)) return false;
        // ---- phase 3: the sum
        bad = false;
        dl = until(phase_timeout);
        std::vector<uint64_t> sum;
        for (uint32_t i = 0; i < world; ++i) {
            if (!recv_watching(fds[i], m, dl)) return false;
            Reader r{m};
            bad = r.get<int32_t>() || bad;
            if (use_rccl) continue;
            const std::vector<uint8_t> v = r.bytes();
            if (!r.ok) return false;
            if (i == 0) continue;
            if (sum.empty()) sum.assign(v.size() / 8, 0);
            if (v.size() != sum.size() * 8) { bad = true; continue; }
            for (size_t k = 0; k < sum.size(); ++k) { uint64_t x; memcpy(&x, v.data() + 8 * k, 8); sum[k] += x; } // (uint64 sums wrap as the counters do)
        }
        if (bad) { tell_all(V_FAIL); return false; }
        for (uint32_t i = 0; i < world; ++i) {
            Writer w; w.put<int32_t>(V_GO);
            if (i == 0 && !use_rccl) w.b.insert(w.b.end(), (const uint8_t*)sum.data(), (const uint8_t*)sum.data() + sum.size() * 8);
            if (!send_msg(fds[i], w.b)) return false;
        }
        // ---- phase 4: import on worker 0; the merged lane names go to the worker that writes
        bad = false;
        dl = until(phase_timeout);
        for (uint32_t i = 0; i < world; ++i) {
            if (!recv_watching(fds[i], m, dl)) return false;
            Reader r{m};
            bad = r.get<int32_t>() || bad;
        }
        if (bad) { tell_all(V_FAIL); return false; }
        // union of the lane names, first come first kept (getLane inserts unknown @RG IDs with index 0, bamqualcheck.cpp:86); output
        // order: lexicographic (writeOutput iterates a std::map)
        std::map<std::string, uint32_t> merged;
        for (const WorkerInfo& e : info) for (const auto& kv : e.lanes) merged.emplace(kv.first, kv.second);
        for (uint32_t i = 0; i < world; ++i) {
            Writer w; w.put<int32_t>(V_GO);
            if (i == 0) { w.put<uint32_t>((uint32_t)merged.size()); for (const auto& kv : merged) { w.str(kv.first); w.put<uint32_t>(kv.second); } }
            if (!send_msg(fds[i], w.b)) return false;
        }
        return true;
    }
};
} // namespace

// argv: the program's arguments without `--gpus N`.  Returns the exit status.
extern "C" int bqc_main_multi(int argc, const char** argv, int n_gpus)
{
    if (n_gpus < 1 || n_gpus > 64) { fprintf(stderr, "bamqualcheck: --gpus wants a number between 1 and 64\n"); return 1; }
    const bool share = getenv("BQC_GPUS_SHARE_DEVICE") && getenv("BQC_GPUS_SHARE_DEVICE")[0] == '1'; // (tests: every worker on one card)
    const char* red = getenv("BQC_REDUCE");
    const bool use_rccl = !(red && strcmp(red, "pipe") == 0) && !(share && n_gpus > 1);
    // -s 0 (table seeded from the time, RepHash.cpp:5-7): every worker must use the same table
    std::vector<std::string> args(argv, argv + argc);
    for (size_t k = 1; k + 1 < args.size(); ++k)
        if ((args[k] == "-s" || args[k] == "--seed") && atoi(args[k + 1].c_str()) == 0) args[k + 1] = std::to_string((long long)std::max<time_t>(1, time(nullptr)));
    // the command line is checked ONCE, here: a usage error is printed once and no worker starts (RCCL included) for it; the input
    // path is the one the workers' own parser finds, not a guess from the file extension
    std::string bam;
    {
        char path[4096] = "";
        const int pr = bqc_program_args(argc, argv, path, sizeof path);
        if (pr == 2) return 0;
        if (pr != 0) return 1;
        bam = path;
    }
    fflush(stdout); fflush(stderr);
    std::vector<int> fds;
    g_children.assign((size_t)n_gpus, 0);
    g_exit.assign((size_t)n_gpus, -1);
    {
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_handler = reap;
        sa.sa_flags = SA_NOCLDSTOP;
        sigaction(SIGCHLD, &sa, nullptr);
        sa.sa_handler = forward_signal;
        sa.sa_flags = 0;
        sigaction(SIGINT, &sa, nullptr);
        sigaction(SIGTERM, &sa, nullptr);
    }
    sigset_t block, old_mask; // (the table of children is written with SIGCHLD held back)
    sigemptyset(&block);
    sigaddset(&block, SIGCHLD);
    for (int i = 0; i < n_gpus; ++i) {
        int sv[2];
        if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv) != 0) { perror("socketpair"); forward_signal(SIGKILL); return 1; }
        sigprocmask(SIG_BLOCK, &block, &old_mask);
        const pid_t pid = fork(); // (before anything touches a GPU)
        if (pid < 0) { perror("fork"); sigprocmask(SIG_SETMASK, &old_mask, nullptr); forward_signal(SIGKILL); return 1; }
        if (pid == 0) {
            signal(SIGCHLD, SIG_DFL); signal(SIGINT, SIG_DFL); signal(SIGTERM, SIG_DFL);
            sigprocmask(SIG_SETMASK, &old_mask, nullptr);
            (void)prctl(PR_SET_PDEATHSIG, SIGTERM); // (a front end that is killed takes its workers with it)
            close(sv[0]);
            for (int fd : fds) close(fd);
            Worker W;
            W.fd = sv[1]; W.rank = (uint32_t)i; W.world = (uint32_t)n_gpus; W.use_rccl = use_rccl;
            W.device = share ? 0 : i;
            std::vector<std::string> a = args;
            a.push_back("--device"); a.push_back(std::to_string(W.device));
            std::vector<const char*> av;
            for (const std::string& s : a) av.push_back(s.c_str());
            if (use_rccl) W.start_rccl();
            const int rc = bqc_main_shard((int)av.size(), av.data(), W.rank, W.world, worker_hook, &W);
            if (W.init.joinable()) W.init.join();
            fflush(stdout); fflush(stderr);
            _exit(rc); // (without the static destructors of the HIP runtime and of RCCL: the process is over)
        }
        close(sv[1]);
        fds.push_back(sv[0]);
        g_children[(size_t)i] = pid;
        sigprocmask(SIG_SETMASK, &old_mask, nullptr);
    }
    Coordinator C{fds, (uint32_t)n_gpus, use_rccl, bam == "-" ? 0 : bqc_file_size(bam.c_str())};
    if (const char* e = getenv("BQC_GPUS_TIMEOUT")) C.phase_timeout = atof(e);
    if (const char* e = getenv("BQC_GPUS_LOOP_TIMEOUT")) C.loop_timeout = atof(e);
    const bool ok = C.run();
    if (!ok) { // a worker that waits for a message ends when its socket closes; one that waits inside an RCCL call is ended
        for (int fd : fds) shutdown(fd, SHUT_RDWR);
        for (int t = 0; t < 30; ++t) {
            bool all = true;
            for (int e : g_exit) all = all && e >= 0;
            if (all) break;
            usleep(100000);
        }
        forward_signal(SIGKILL);
    }
    int status = ok ? 0 : 1;
    for (;;) { // (the SIGCHLD handler collects them)
        bool all = true;
        for (int e : g_exit) all = all && e >= 0;
        if (all) break;
        usleep(2000);
    }
    for (int e : g_exit) if (e != 0) status = 1;
    for (int fd : fds) close(fd);
    return status;
}
