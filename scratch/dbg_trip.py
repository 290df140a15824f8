import sys, numpy as np
sys.path.insert(0, ".")
from tests import synth
from tests.parity import run_oracle, run_gpu
cols, refs = synth.synth(seed=1, n_reads=6000, n_refs=2, ref_len=120_000)
for r in (701, 1274, 1278, 2351):
    sub = synth.slice_batch(cols, r, r + 1)
    _, co, _ = run_oracle([sub], refs, n_refs=2); _, cg, _ = run_gpu([sub], refs, n_refs=2)
    to = np.array(co[0]["triplet"], dtype=np.int64); tg = np.array(cg[0]["triplet"], dtype=np.int64)
    for i in np.nonzero(to != tg)[0]:
        print(r, "bin", i, "ctx", i >> 4, "grp", (i >> 2) & 3, "base", i & 3, "oracle", to[i], "gpu", tg[i])
