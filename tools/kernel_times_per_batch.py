"""Per-kernel HIP-event times (bqc_last_timing) of consecutive 1 M-read batches with the k-mer sketch enabled.
usage: python tools/kernel_times_per_batch.py"""
import sys
sys.path.insert(0, ".")
from bamqc_amd import Aggregator, synth
lens = [25_000_000] * 4
refs = [synth.reference(3, i, n) for i, n in enumerate(lens)]
agg = Aggregator(n_refs=4, klist=[32], qlist=[17])
for i, r in enumerate(refs): agg.set_reference(i, r)
agg.set_timing(True)
from tests import synth as tsynth
n = 1_000_000
all_cols = synth.batch(3, 10 * n, lens, refs)
for k in range(3):
    cols = tsynth.slice_batch(all_cols, k * n, (k + 1) * n)
    db = agg.upload(cols); agg.process(db); agg.sync()
    t = agg.last_timing(); db.free()
    print(k, {a: round(b, 3) for a, b in t.items()})
