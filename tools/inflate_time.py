"""Times the BGZF reader's runs (BQC_GI_TIMING lines) on a synthetic BAM with the GPU inflater at several workgroup widths and on the CPU.
usage: python tools/inflate_time.py [reads] [level]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = """
import sys
sys.path.insert(0, %r)
from bamqc_amd import _lib, hostio
lib = _lib.load()
dev = int(sys.argv[2])
if dev >= 0:
    lib.bqc_gpu_inflate_device(dev)
b = hostio.BamFile(sys.argv[1])
n = 0
for batch in b.batches(1 << 20):
    n += len(batch["flag"])
b.close()
print("reads", n, "blocks on the gpu", lib.bqc_gpu_inflated_blocks())
""" % ROOT


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from bamqc_amd import hostio
    with tempfile.TemporaryDirectory(prefix="bqc_gi_") as tmp:
        bam = os.path.join(tmp, "x.bam")
        hostio.synth_stream(bam, None, 1002, reads, ["chr1", "chr2", "chr3", "chr4"], [25_000_000] * 4, level=level)
        for lanes in (None, 64, 32, 16, 8):
            env = dict(os.environ, BQC_GI_TIMING="1")
            if lanes:
                env["BQC_GI_LANES"] = str(lanes)
            r = subprocess.run([sys.executable, "-c", CHILD, bam, "0" if lanes else "-1"], env=env, capture_output=True, text=True)
            print("==== lanes per workgroup:", lanes or "cpu decoder", flush=True)
            lines = [ln for ln in r.stderr.splitlines() if ln.startswith("[")]
            print("\n".join(lines[:3] + lines[-8:]))
            print(r.stdout.strip(), r.returncode, flush=True)


main()
