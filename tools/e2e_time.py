"""Wall time of bin/bamqualcheck on a synthetic config-2 BAM file, with the [timing] lines of the program.
usage: python tools/e2e_time.py [reads] [level] [ENV=VALUE ...]   (each ENV=VALUE set is run as its own variant after the default)"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bamqc_amd import hostio  # noqa: E402


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    variants = [{}] + [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[3:]]
    names, lens = ["chr1", "chr2", "chr3", "chr4"], [25_000_000] * 4
    exe = os.path.join(ROOT, "bin", "bamqualcheck")
    with tempfile.TemporaryDirectory(prefix="bqc_e2e_") as tmp:
        bam, fa = os.path.join(tmp, "c2.bam"), os.path.join(tmp, "c2.fa")
        hostio.synth_stream(bam, fa, 1002, reads, names, lens, level=level)
        print("input: %d reads, %.0f MB" % (reads, os.path.getsize(bam) / 1e6), flush=True)
        outs = []
        for v in variants:
            for rep in range(3):
                out = os.path.join(tmp, "o%d.bamqc" % len(outs))
                t0 = time.perf_counter()
                r = subprocess.run([exe, "-r", fa, "-o", out, "-c", ",".join(names), bam], capture_output=True, text=True,
                                   env=dict(os.environ, BQC_TIMING="1", BQC_T0="%.6f" % time.monotonic(), **v))
                dt = time.perf_counter() - t0
                assert r.returncode == 0, r.stderr[-3000:]
                print("%s run %d: %.3f s = %.1f M reads/s" % (v or "default", rep, dt, reads / dt / 1e6), flush=True)
                if rep == 2:
                    print("\n".join(ln for ln in r.stderr.splitlines() if ln.startswith(("[timing]", "[gpu reader] open"))), flush=True)
                outs.append(out)
        ref = open(outs[0], "rb").read()
        for o in outs[1:]:
            assert open(o, "rb").read() == ref, "outputs differ: %s" % o
        print("all outputs identical")


main()
