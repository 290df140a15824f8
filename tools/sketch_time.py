"""Device time of the hot path WITH the k-mer sketch (default options of the program: -k 32 -q 17) on config-2-shaped reads
resident in HBM.  usage: python tools/sketch_time.py [n_reads]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from bamqc_amd import Aggregator, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
lens = [25_000_000] * 4
refs = [synth.reference(1002, i, x) for i, x in enumerate(lens)]
cols = synth.batch(1002, n, lens, refs)
agg = Aggregator(n_refs=4, klist=(32,), qlist=(17,))
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
for k, v in kt.items():
    print("%-12s %.3f ms = %.3f ms per 10 M reads" % (k, float(np.mean(v)), float(np.mean(v)) * 1e7 / n))
