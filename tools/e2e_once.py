"""One run of bin/bamqualcheck on a synthetic BAM with all its stderr lines. usage: python tools/e2e_once.py reads level [ENV=VALUE,...]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bamqc_amd import hostio  # noqa: E402

reads, level = int(sys.argv[1]), int(sys.argv[2])
env = dict(kv.split("=") for kv in sys.argv[3].split(",")) if len(sys.argv) > 3 else {}
names, lens = ["chr1", "chr2", "chr3", "chr4"], [25_000_000] * 4
with tempfile.TemporaryDirectory(prefix="bqc_e2e_") as tmp:
    bam, fa = os.path.join(tmp, "c2.bam"), os.path.join(tmp, "c2.fa")
    hostio.synth_stream(bam, fa, 1002, reads, names, lens, level=level)
    for rep in range(2):
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(ROOT, "bin", "bamqualcheck"), "-r", fa, "-o", os.path.join(tmp, "o.bamqc"), "-c", ",".join(names), bam],
                           capture_output=True, text=True, env=dict(os.environ, BQC_TIMING=os.environ.get("E2E_TIMING", "1"), BQC_T0="%.6f" % time.monotonic(), **env))
        print("run %d: %.3f s rc %d" % (rep, time.perf_counter() - t0, r.returncode))
    print(r.stderr[-12000:])
