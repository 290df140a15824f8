// parallel.h — contiguous-range fork/join over host threads (decode and pre-pass of a batch).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sched.h>
#include <thread>
#include <vector>

// CPUs this process may use: the smaller of the affinity mask and the cgroup CPU quota (cpu.max of cgroup v2,
// cfs_quota_us / cfs_period_us of v1).  A container limited to 16 of a host's 256 hardware threads is throttled by the
// scheduler when 64 workers burn through the quota early in every period.
inline unsigned bqc_cpu_limit()
{
    unsigned n = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min(n ? n : (unsigned)c, (unsigned)c); }
    long long quota = -1, period = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else {
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
    }
    if (quota > 0 && period > 0) n = std::min<unsigned>(n ? n : 1u, (unsigned)((quota + period - 1) / period));
    return std::max(1u, n);
}

inline unsigned bqc_host_threads() // BQC_IO_THREADS, else the CPUs available to the process (at most 64)
{
    static const unsigned cached = [] {
        unsigned n = bqc_cpu_limit();
        if (const char* e = getenv("BQC_IO_THREADS")) n = (unsigned)atoi(e);
        return std::max(1u, std::min(n, 64u));
    }();
    return cached;
}

// f(thread, lo, hi) over at most `threads` contiguous ranges of [0, n); ranges hold at least `grain` items
template <typename F>
unsigned parallel_ranges(size_t n, unsigned threads, size_t grain, F f)
{
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n / std::max<size_t>(grain, 1) + 1));
    if (nt == 1) { f(0u, (size_t)0, n); return 1; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back([=, &f]() { f(t, n * t / nt, n * (t + 1) / nt); });
    for (auto& x : th) x.join();
    return nt;
}
