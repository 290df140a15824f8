"""The program on reads whose qualities are NOISE (uniform Phred 2..41 per base instead of the generator's smooth profiles): the BGZF
blocks are then mostly literals (qualities do not compress), the case the synthetic files of the benchmarks do not cover.  Output
against the oracle (the whole file), wall times, the reader's inflate launches.
usage: python tools/noisy_quals.py [reads] [out.json]"""
import filecmp
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bamqc_amd import hostio, synth  # noqa: E402
from tests.cli_oracle import oracle_bamqualcheck  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    out = sys.argv[2] if len(sys.argv) > 2 else None
    tmp = tempfile.mkdtemp(prefix="bqc_noisy_")
    names, lens = ["chr1", "chr2", "chr3", "chr4"], [25_000_000] * 4
    refs = [synth.reference(1002, i, m) for i, m in enumerate(lens)]
    cols = hostio.synth_slice(1002, n, 0, n, lens, refs=refs)
    rng = np.random.default_rng(7)
    cols["qual"] = rng.integers(2, 42, len(cols["qual"]), dtype=np.uint8)
    bam, fa = os.path.join(tmp, "n.bam"), os.path.join(tmp, "n.fa")
    hostio.write_bam(bam, cols, names, lens)
    hostio.write_fasta(fa, names, refs)
    exe = os.path.join(ROOT, "bin", "bamqualcheck")
    res = {"reads": n, "bam_MB": os.path.getsize(bam) / 1e6, "what": "uniform random qualities 2..41: mostly literals in the BGZF blocks"}
    walls = []
    for k in range(3):
        time.sleep(1.0)
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-r", fa, "-o", os.path.join(tmp, "g.bamqc"), "-c", ",".join(names), bam], capture_output=True, text=True,
                           env=dict(os.environ, BQC_TIMING="1", BQC_GPU_DECODE="1", BQC_GB_TIMING="2" if k == 2 else "0"))
        walls.append(time.perf_counter() - t0)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "records decoded on the GPU" in r.stderr
    res["wall_s"] = walls
    res["inflate_launches"] = [{"blocks": int(m.group(1)), "compressed_MB": float(m.group(2)), "inflated_MB": float(m.group(3)), "ms": float(m.group(4))}
                               for m in re.finditer(r"run of (\d+) blocks, ([0-9.]+) MB -> ([0-9.]+) MB: inflate \+ crc ([0-9.]+) ms", r.stderr)]
    t0 = time.perf_counter()
    assert oracle_bamqualcheck(bam, fa, os.path.join(tmp, "o.bamqc"), chroms=",".join(names)) == 0
    res["oracle_s"] = time.perf_counter() - t0
    res["identical_to_oracle"] = filecmp.cmp(os.path.join(tmp, "g.bamqc"), os.path.join(tmp, "o.bamqc"), shallow=False)
    print(json.dumps(res, indent=1))
    if out:
        json.dump(res, open(out, "w"), indent=1)
    assert res["identical_to_oracle"]


if __name__ == "__main__":
    main()
