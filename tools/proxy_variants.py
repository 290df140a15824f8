"""One worker's eighth of config 3 (bench.py: e2e_shard_proxy) written once, then bin/bamqualcheck on it under several environments, three runs
each: wall time, when the record loop starts, its length.  usage: python tools/proxy_variants.py [ENV=VALUE[,ENV=VALUE] ...]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bamqc_amd import hostio  # noqa: E402
from bamqc_amd.synth import GRCH38  # noqa: E402

names = ["chr4", "chr5", "chr6"]
lens = [GRCH38[3], GRCH38[4], GRCH38[5]]
reads = 618_000_000 // 8
exe = os.path.join(ROOT, "bin", "bamqualcheck")
tmp = tempfile.mkdtemp(prefix="bqc_proxy_")
try:
    bam, fa = os.path.join(tmp, "p.bam"), os.path.join(tmp, "p.fa")
    hostio.synth_stream(bam, fa, 1013, reads, names, lens, level=1)
    ref = None
    for spec in [""] + sys.argv[1:]:
        env = dict(kv.split("=") for kv in spec.split(",")) if spec else {}
        rows = []
        for k in range(3):
            time.sleep(1.5)
            out = os.path.join(tmp, "o.bamqc")
            t0 = time.perf_counter()
            r = subprocess.run([exe, "-r", fa, "-o", out, "-c", ",".join(names), bam], capture_output=True, text=True,
                               env=dict(os.environ, BQC_TIMING="1", BQC_T0="%.6f" % time.monotonic(), **env))
            dt = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr[-2000:]
            got = open(out, "rb").read()
            ref = ref or got
            assert got == ref
            ls = re.search(r"record loop starts: ([0-9.]+) s", r.stderr)
            lp = re.search(r"record loop ([0-9.]+) s", r.stderr)
            rows.append((round(dt, 3), float(ls.group(1)) if ls else None, float(lp.group(1)) if lp else None))
        print(spec or "default", rows, flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
