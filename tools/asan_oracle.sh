#!/bin/bash
# The CPU restatement (oracle/) under ASan + UBSan: its known-answer tests, the sketch tests and the wild records of
# tests/test_gpu_fuzz.py (oracle side only).  The oracle defines what the HIP path is compared with, so an out-of-bounds
# read in it is a wrong expectation.  Usage: tools/asan_oracle.sh
set -e
cd "$(dirname "$0")/.."
make -C oracle liboracle_asan.so > /dev/null
export BQC_ORACLE_ASAN=1 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
python -m pytest tests/test_oracle_kats.py tests/test_sketch_oracle.py tests/test_golden_config1.py -x -q 2>&1 | tail -3
python - <<'PY'
import sys
sys.path.insert(0, ".")
import numpy as np
from tests.test_gpu_fuzz import wild_batch
from tests.parity import run_oracle, split
n_ok = 0
for seed in range(60):
    rng = np.random.default_rng(seed)
    nl = int(rng.integers(1, 5))
    cols, refs = wild_batch(5000 + seed, int(rng.choice([1, 50, 2000])), n_lanes=nl, max_len=int(rng.choice([40, 255, 600])), ref_len=int(rng.choice([300, 5000, 30000])))
    rc, co, o = run_oracle([cols], refs, n_refs=3, n_lanes=nl, max_read_len=1024, isize=1000)
    n_ok += rc == 0
print("wild records through the sanitized oracle: %d of 60 batches without an error code, no sanitizer report" % n_ok)
PY
