timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -5
python - <<'PY'
import time, numpy as np
from bamqc_amd import Aggregator, synth as csynth
lens=[50_000_000]
refs=[csynth.reference(1005,0,lens[0])]
cols=csynth.batch(1005, 50_000, lens, refs, read_len=10_000, isize=30_000, long_reads=True)
a=Aggregator(n_refs=1, isize=30000, max_read_len=16384, hist_cap=16384)
a.set_reference(0, refs[0])
db=a.upload(cols); a.set_timing(True)
for _ in range(2): a.process(db)
a.sync(); print("long reads 50k x 10kb:", a.last_timing(), "algo bytes", db.algorithmic_bytes)
PY
