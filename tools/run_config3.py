"""BASELINE.json config 3 at full size on one MI355X: ~618 M synthetic 150 bp PE reads (30x of 3.088 Gb) over chr1..chr22, chrX,
chrY with GRCh38 lengths, default -c, default k-mer sketch, streamed generator -> FIFO -> bin/bamqualcheck (the input never
exists as a file: it would be ~185 GB of BAM records).  Checked by the size-independent properties of tests/bamqc_text.py.
usage: python tools/run_config3.py [n_reads] [bgzf_level] [out.json]   (under gpurun; see profiles/r2_config3*.json)"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bamqc_amd.synth import GRCH38  # noqa: E402
from tests import bamqc_text  # noqa: E402
from tests.test_gpu_stream import NAMES24, stream_through_cli  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 618_000_000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 0
out_json = sys.argv[3] if len(sys.argv) > 3 else None
tmp = tempfile.mkdtemp(prefix="bqc_c3_")
t0 = time.time()
out, stderr, wall = stream_through_cli(tmp, 1003, n, NAMES24, GRCH38, [], level=level)
total = time.time() - t0
lanes = bamqc_text.parse(out)
info = bamqc_text.check_invariants(lanes["L1"], n_records=n, read_len=150)
g = lanes["L1"]
cov = g["genome_coverage_histogram"]
res = {"config": "3: 30x WGS-scale synthetic, 24 GRCh38-length contigs, default -c / -k 32 -q 17", "reads": n, "bgzf_level": level,
       "program_wall_s": wall, "reads_per_s": n / wall, "total_s_incl_reference_generation": total, "primary_reads": info["primary"],
       "triplets": info["triplets"], "eightmers": info["eightmers"], "coverage_positions": int(cov.sum()),
       "mean_depth_main": float((cov * range(101)).sum() / max(1, cov.sum())), "invariants": "ok",
       "timing": [ln for ln in stderr.splitlines() if ln.startswith("[timing]")]}
print(json.dumps(res, indent=1))
if out_json:
    json.dump(res, open(out_json, "w"), indent=1)
