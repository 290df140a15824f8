"""GPU: the multi-process path (ranks sharing the one GPU of the test box, gloo for the collectives; on a multi-GPU node the
same code runs with RCCL) must write the same bytes as the single-process program.  The BAM file's byte stream is split
between the ranks INSIDE chromosomes; the coverage state machine's state is handed down the chain of ranks."""
import filecmp
import os
import socket
import subprocess
import sys

import pytest

from bamqc_amd import hostio

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "bin", "bamqualcheck")


def same(a, b):
    """byte comparison; on a mismatch the names of the differing lines go into the assertion message"""
    if filecmp.cmp(a, b, shallow=False):
        return True
    la, lb = open(a).read().splitlines(), open(b).read().splitlines()
    diff = []
    for k in range(max(len(la), len(lb))):
        x, y = (la[k] if k < len(la) else ""), (lb[k] if k < len(lb) else "")
        if x != y:
            tx, ty = x.split(), y.split()
            where = [j for j in range(min(len(tx), len(ty))) if tx[j] != ty[j]][:6]
            diff.append("%s: %s" % (tx[0] if tx else "?", [(j, tx[j], ty[j]) for j in where]))
    raise AssertionError("outputs differ in %d lines: %s" % (len(diff), diff[:12]))


def run_ranks(world, args, env_extra=None, timeout=600):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=ROOT)
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, "-m", "bamqc_amd.dist_cli", "--backend", "gloo"] + list(args), env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=timeout)[0] for p in procs]
    return [p.returncode for p in procs], outs


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    d = tmp_path_factory.mktemp("shard")
    bam, fa = str(d / "s.bam"), str(d / "s.fa")
    # three contigs of very different size: with 2 or 3 ranks the cuts fall inside chr1; two read groups interleaved
    hostio.synth_write(bam, fa, seed=77, n_reads=300_000, ref_names=["chr1", "chr2", "chrM"], ref_lens=[2_500_000, 300_000, 20_000], n_lanes=2)
    single = str(d / "single.bamqc")
    r = subprocess.run([EXE, "-r", fa, "-o", single, "-c", "chr1,chr2", "-i", "800", bam], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return bam, fa, single, d


@pytest.mark.parametrize("world", [2, 3])
def test_byte_range_shards_equal_single_process(inputs, world):
    bam, fa, single, d = inputs
    out = str(d / ("sharded%d.bamqc" % world))
    rcs, outs = run_ranks(world, ["-r", fa, "-o", out, "-c", "chr1,chr2", "-i", "800", bam])
    assert rcs == [0] * world, outs
    assert same(single, out)


def test_sparse_coverage_and_small_batches(tmp_path):
    """Few reads over a long contig: gaps of more than 2000 positions reset the coverage state machine all the time, so most
    of a shard runs on its own trajectory and only a short prefix waits for the predecessor; batches of 3001 reads."""
    bam, fa = str(tmp_path / "g.bam"), str(tmp_path / "g.fa")
    hostio.synth_write(bam, fa, seed=5, n_reads=60_000, ref_names=["chr1", "chr2"], ref_lens=[40_000_000, 30_000_000])
    single, out = str(tmp_path / "single.bamqc"), str(tmp_path / "sharded.bamqc")
    r = subprocess.run([EXE, "-r", fa, "-o", single, "-c", "chr1,chr2", "--no-sketch", bam], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rcs, outs = run_ranks(3, ["-r", fa, "-o", out, "-c", "chr1,chr2", "--no-sketch", "--batch-reads", "3001", bam])
    assert rcs == [0, 0, 0], outs
    assert same(single, out)


def test_unverifiable_split_falls_back_to_one_process(inputs):
    """BQC_TEST_SHARD_SKEW makes every middle shard guess its first record wrong: the ranks notice (the predecessor's last
    record does not end where the successor began), rank 0 processes the file alone, the output is still right."""
    bam, fa, single, d = inputs
    out = str(d / "fallback.bamqc")
    rcs, outs = run_ranks(2, ["-r", fa, "-o", out, "-c", "chr1,chr2", "-i", "800", bam], env_extra={"BQC_TEST_SHARD_SKEW": "1"})
    assert rcs == [0, 0], outs
    assert "could not be verified" in outs[0]
    assert same(single, out)


def test_an_error_on_one_rank_ends_all_ranks(inputs):
    """The FASTA file lacks a contig that only the LAST rank's reads need: that rank reports it, every rank exits 1 (no rank is
    left waiting in a collective)."""
    bam, fa, single, d = inputs
    fa1 = str(d / "chr1_only.fa")
    with open(fa) as f, open(fa1, "w") as g:
        keep = False
        for line in f:
            if line.startswith(">"):
                keep = line.startswith(">chr1")
            if keep:
                g.write(line)
    rcs, outs = run_ranks(2, ["-r", fa1, "-o", str(d / "err.bamqc"), "-c", "chr1,chr2", "-i", "800", bam])
    assert rcs == [1, 1], outs
    assert any("Could not read fasta record" in o for o in outs)
