// parallel.h — contiguous-range fork/join over host threads (decode and pre-pass of a batch).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <thread>
#include <vector>

inline unsigned bqc_host_threads() // BQC_IO_THREADS, else the hardware threads (at most 64)
{
    unsigned n = std::thread::hardware_concurrency();
    if (const char* e = getenv("BQC_IO_THREADS")) n = (unsigned)atoi(e);
    return std::max(1u, std::min(n, 64u));
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
