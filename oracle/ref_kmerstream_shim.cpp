// ref_kmerstream_shim.cpp — drives the REFERENCE's own RepHash / StreamCounter
// (compiled from /root/reference/src/kmerstream where they lie; see Makefile)
// so the restatement in sketch_oracle.c can be pinned against them and golden
// vectors can be generated (tests/golden/make_kmerstream_golden.py).
// TEST INFRASTRUCTURE ONLY.  Contains no reference code: it only calls it.
#include <cstring>   // StreamCounter.hpp uses memset without including it
#include <cstddef>
#include "RepHash.hpp"
#include "StreamCounter.hpp"

// RepHash::hvals is private and no public member exposes it.  Access checks do not apply to the arguments of an explicit
// template instantiation, so the member pointer can be handed out this way without touching the reference's headers
// (standard C++, no macro redefinition of keywords).
namespace {
struct HvalsTag { typedef state_t (RepHash::*type)[32]; friend type hvals_member(HvalsTag); };
template <typename Tag, typename Tag::type M> struct Expose { friend typename Tag::type hvals_member(Tag) { return M; } };
template struct Expose<HvalsTag, &RepHash::hvals>;
}

extern "C" {
void ref_rephash_table(int seed, uint64_t out[64])
{
    RepHash h;
    h.seed(seed);
    const state_t (&hv)[32] = h.*hvals_member(HvalsTag());
    for (int i = 0; i < 32; ++i) { out[2 * i] = hv[i].hi; out[2 * i + 1] = hv[i].lo; }
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
