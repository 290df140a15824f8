"""Device time of the long-read kernels on config-5-shaped reads (10 kb, 20-60 CIGAR operations, 10 % clipped): n reads resident
in HBM, HIP events of the library's stream.  usage: python tools/k_long_time.py [n_reads] [read_len]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from bamqc_amd import Aggregator, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
lens = [250_000_000]
refs = [synth.reference(1005, 0, lens[0])]
cols = synth.batch(1005, n, lens, refs, read_len=L, isize=30_000, long_reads=True)
agg = Aggregator(n_refs=1, isize=30_000, max_read_len=max(16_384, L), hist_cap=16_384)
agg.set_reference(0, refs[0])
db = agg.upload(cols)
agg.set_timing(True)
kt = {}
for it in range(8):
    agg.process(db)
    for k, v in agg.last_timing().items():
        if it >= 2:
            kt.setdefault(k, []).append(v)
agg.sync()
# the same without per-kernel events: everything a batch costs the stream, the deferred fold of the 8-mer scratch rows included
import time
agg.set_timing(False)
t0 = time.perf_counter()
for it in range(16):
    agg.process(db)
agg.sync()
print("16 batches back to back, synchronised at the end: %.3f ms per batch (all kernels of a batch and the folds)" % ((time.perf_counter() - t0) * 1e3 / 16))
agg.set_timing(True)
ab = db.algorithmic_bytes
print("reads %d x %d bases, CIGAR operations per read %.1f, algorithmic bytes %.1f MB" % (n, L, len(cols["cigar"]) / n, ab / 1e6))
for k, v in kt.items():
    print("%-18s %.3f ms" % (k, float(np.mean(v))))
kl = float(np.mean(kt["k_long"]))
print("k_long: %.1f G bases/s, %.0f GB/s algorithmic = %.3f of 8 TB/s" % (n * L / kl / 1e6, ab / kl / 1e6, ab / kl / 1e6 / 8000))
