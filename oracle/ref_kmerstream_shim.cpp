// ref_kmerstream_shim.cpp — drives the REFERENCE's own RepHash / StreamCounter
// (compiled from /root/reference/src/kmerstream where they lie; see Makefile)
// so the restatement in sketch_oracle.c can be pinned against them and golden
// vectors can be generated (tests/golden/make_kmerstream_golden.py).
// TEST INFRASTRUCTURE ONLY.  Contains no reference code: it only calls it.
#include <cstring>   // StreamCounter.hpp uses memset without including it
#include <cstddef>
#define private public   // RepHash::hvals is private; the shim only reads it
#include "RepHash.hpp"
#undef private
#include "StreamCounter.hpp"

extern "C" {
void ref_rephash_table(int seed, uint64_t out[64])
{
    RepHash h;
    h.seed(seed);
    for (int i = 0; i < 32; ++i) { out[2 * i] = h.hvals[i].hi; out[2 * i + 1] = h.hvals[i].lo; }
}
uint32_t ref_rephash_sequence(int seed, int k, const char* s, uint32_t l, uint64_t* out)
{
    RepHash h;
    h.seed(seed);
    h.init(k);
    if (l < (uint32_t)k) return 0;
    h.init(s);
    uint32_t n = 0;
    out[n++] = h.hash();
    for (uint32_t j = k; j < l; ++j) { h.update(s[j - k], s[j]); out[n++] = h.hash(); }
    return n;
}
void ref_streamcounter_run(double e, const uint64_t* hashes, uint64_t n, uint64_t res[4])
{
    StreamCounter sc(e, 1);
    for (uint64_t i = 0; i < n; ++i) sc(hashes[i]);
    res[0] = sc.get_sumCount(); res[1] = sc.F0(); res[2] = sc.f1(); res[3] = sc.F2();
}
}
