"""Device time of the hot path on config-2-shaped reads of SEVERAL read groups (the 8-mer counters of all but the read group with the
most reads leave the workgroups by global atomics, not through the scratch rows).  usage: python tools/lanes_time.py [n_reads] [n_lanes]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from bamqc_amd import Aggregator, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lens = [25_000_000] * 4
refs = [synth.reference(1002, i, x) for i, x in enumerate(lens)]
cols = synth.batch(1002, n, lens, refs, n_lanes=n_lanes)
agg = Aggregator(n_refs=4, n_lanes=n_lanes, klist=(), qlist=())
for i, r in enumerate(refs):
    agg.set_reference(i, r)
db = agg.upload(cols)
agg.set_timing(True)
kt = {}
for it in range(8):
    agg.process(db)
    for k, v in agg.last_timing().items():
        if it >= 3:
            kt.setdefault(k, []).append(v)
agg.sync()
print("%d reads, %d read groups" % (n, n_lanes))
for k, v in kt.items():
    print("%-12s %.3f ms" % (k, float(np.mean(v))))
