"""Ten 10 M-read files back to back through bin/bamqualcheck under several environments, alternating, on ONE box: what a knob does to
the time between two runs of a pipeline (the worker of the run before is still handing its memory back when the next one starts).
usage: python tools/b2b_ab.py [out.json] -- NAME=K=V,K=V ... (NAME= alone: no change)"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bamqc_amd import hostio  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "--" else None
    variants = []
    for a in sys.argv[sys.argv.index("--") + 1:]:
        name, _, rest = a.partition("=")
        variants.append((name, dict(kv.split("=", 1) for kv in rest.split(",") if kv)))
    exe = os.path.join(ROOT, "bin", "bamqualcheck")
    tmp = tempfile.mkdtemp(prefix="bqc_b2b_")
    names, lens = ["chr1", "chr2", "chr3", "chr4"], [25_000_000] * 4
    bam, fa = os.path.join(tmp, "c2.bam"), os.path.join(tmp, "c2.fa")
    hostio.synth_stream(bam, fa, 1002, 10_000_000, names, lens, read_len=150, level=1)
    rows = []
    for rnd in range(3):
        for name, env in variants:
            time.sleep(1.5)
            singles = []
            t0 = time.perf_counter()
            for k in range(10):
                t1 = time.perf_counter()
                r = subprocess.run([exe, "-r", fa, "-o", os.path.join(tmp, "b.bamqc"), "-c", ",".join(names), bam], capture_output=True, text=True, env=dict(os.environ, **env))
                assert r.returncode == 0, r.stderr[-2000:]
                singles.append(round(time.perf_counter() - t1, 3))
            dt = time.perf_counter() - t0
            rows.append({"variant": name, "round": rnd, "s_per_file": dt / 10, "each": singles})
            print(rows[-1], flush=True)
    if out:
        json.dump(rows, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
