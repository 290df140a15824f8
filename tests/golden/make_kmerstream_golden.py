"""Generates tests/golden/kmerstream_golden.json from the REFERENCE's own kmerstream sources compiled
into oracle/_ref/libref_kmerstream.so (build container only).  Inputs are defined here; outputs come
from the reference code."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_kmerstream.so"))
u64p = C.POINTER(C.c_uint64)


def splitmix_stream(seed, n):
    out = np.zeros(n, np.uint64)
    x = seed & (2 ** 64 - 1)
    for i in range(n):
        x = (x + 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
        out[i] = z ^ (z >> 31)
    return out


gold = {"source": "DecodeGenetics/BamQC src/kmerstream (RepHash.cpp, RepHash.hpp, StreamCounter.hpp, lsb.cpp) compiled with g++ -std=c++0x"}
tables = {}
for seed in (1, 2, 12345):
    t = (C.c_uint64 * 64)()
    ref.ref_rephash_table(seed, t)
    tables[str(seed)] = ["%016x" % v for v in t]
gold["rephash_table"] = tables
seqs = {"survey": "ACGTACGTTTGACCAGTACGATCGATCGGGCTAACGTTAGC",
        "mixed": "TTGACCAGTANNGTACGATCGATCGGGCTAACGTTAGCACGTACGTTTGACCAGTACGATCGMRATCGGGCTAACGTTAGCAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA"}
hs = []
for name, s in seqs.items():
    for k in (1, 5, 31, 32, 33, 63):
        if k > len(s):
            continue
        out = (C.c_uint64 * len(s))()
        n = ref.ref_rephash_sequence(1, k, s.encode(), len(s), out)
        hs.append({"seq": s, "k": k, "seed": 1, "hashes": ["%016x" % out[i] for i in range(n)]})
gold["rephash_sequences"] = hs
scs = []
for seed, n, mod in ((1, 0, 0), (2, 10, 0), (3, 5000, 0), (4, 200000, 0), (5, 300000, 1000), (6, 2000000, 0)):
    h = splitmix_stream(seed, n)
    if mod:
        h = h[np.arange(n) % mod]  # heavy repetition
    res = (C.c_uint64 * 4)()
    ref.ref_streamcounter_run(C.c_double(0.01), h.ctypes.data_as(u64p), n, res)
    scs.append({"stream_seed": seed, "n": n, "repeat_mod": mod, "e": 0.01, "sumCount": int(res[0]), "F0": int(res[1]), "f1": int(res[2]), "F2": int(res[3])})
gold["streamcounter"] = scs
json.dump(gold, open(os.path.join(HERE, "kmerstream_golden.json"), "w"), indent=0)
print("wrote", len(hs), "hash sequences,", len(scs), "stream counter cases;", tables["1"][:2], scs)
