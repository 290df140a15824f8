"""GPU: the chromosome-sharded multi-process path (2 ranks sharing the one GPU of the test box,
gloo for the host-side reduce; on a multi-GPU node the same code runs with RCCL) must write the
same bytes as the single-process program."""
import filecmp
import os
import socket
import subprocess
import sys

import pytest

from bamqc_amd import hostio

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from bamqc_amd import distributed as D
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[3]
dist.init_process_group("gloo", rank=rank, world_size=world)
rc = D.run_sharded(sys.argv[4], sys.argv[5], sys.argv[6], chroms="chr1,chr2,chr3", isize=800, klist=(), qlist=(), device=0, batch_reads=5000)
dist.barrier(); dist.destroy_process_group(); sys.exit(rc)
"""


def test_two_ranks_equal_single_process(tmp_path):
    bam, fa = str(tmp_path / "s.bam"), str(tmp_path / "s.fa")
    names = ["chr1", "chr2", "chr3", "chrM"]
    hostio.synth_write(bam, fa, seed=77, n_reads=40_000, ref_names=names, ref_lens=[500_000, 300_000, 250_000, 20_000], n_lanes=2)
    single, sharded = str(tmp_path / "single.bamqc"), str(tmp_path / "sharded.bamqc")
    r = subprocess.run([os.path.join(ROOT, "bin", "bamqualcheck"), "-r", fa, "-o", single, "-c", "chr1,chr2,chr3", "-i", "800",
                        "--no-sketch", bam], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = str(tmp_path / "worker.py")
    open(script, "w").write(WORKER % {"root": ROOT})
    procs = [subprocess.Popen([sys.executable, script, str(rank), "2", str(port), bam, fa, sharded]) for rank in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    assert filecmp.cmp(single, sharded, shallow=False)
