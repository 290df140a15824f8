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


def run_ranks(world, args, env_extra=None, timeout=600, launcher="torch"):
    if launcher == "cxx":  # the one-binary form: bamqualcheck --gpus N forks its workers itself (bamqc_amd/host/multi_gpu.cpp)
        env = dict(os.environ, BQC_GPUS_SHARE_DEVICE="1")  # (the test box has one card: every worker uses device 0, sums go through pipes)
        env.update(env_extra or {})
        r = subprocess.run([EXE, "--gpus", str(world)] + list(args), env=env, capture_output=True, text=True, timeout=timeout)
        return [r.returncode] * world, [r.stdout + r.stderr] * world
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


@pytest.mark.parametrize("launcher", ["torch", "cxx"])
@pytest.mark.parametrize("decode", ["0", "1"])
@pytest.mark.parametrize("world", [2, 3])
def test_byte_range_shards_equal_single_process(inputs, world, decode, launcher):
    """decode 1: every rank inflates and decodes ITS byte range on the card (csrc/gpu_bam.hip: set_range); 0: the host reader."""
    bam, fa, single, d = inputs
    out = str(d / ("sharded%d_%s_%s.bamqc" % (world, decode, launcher)))
    rcs, outs = run_ranks(world, ["-r", fa, "-o", out, "-c", "chr1,chr2", "-i", "800", bam], env_extra={"BQC_GPU_DECODE": decode, "BQC_TIMING": "1"}, launcher=launcher)
    assert rcs == [0] * world, outs
    assert same(single, out)
    if decode == "1":
        assert all("records decoded on the GPU" in o for o in outs), outs
    if world == 2 and launcher == "torch":  # against the ORACLE's file too, not only the single-process GPU run
        from tests.cli_oracle import oracle_bamqualcheck
        want = str(d / "oracle.bamqc")
        if not os.path.exists(want):
            assert oracle_bamqualcheck(bam, fa, want, chroms="chr1,chr2", isize=800) == 0
        assert same(want, out)


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


@pytest.mark.parametrize("launcher,decode", [("torch", "0"), ("torch", "1"), ("cxx", "1")])
def test_unverifiable_split_falls_back_to_one_process(inputs, launcher, decode):
    """BQC_TEST_SHARD_SKEW makes every middle shard guess its first record wrong (either reader): the ranks notice (the
    predecessor's last record does not end where the successor began), rank 0 processes the file alone, the output is still right."""
    bam, fa, single, d = inputs
    out = str(d / ("fallback_%s_%s.bamqc" % (launcher, decode)))
    rcs, outs = run_ranks(2, ["-r", fa, "-o", out, "-c", "chr1,chr2", "-i", "800", bam], env_extra={"BQC_TEST_SHARD_SKEW": "1", "BQC_GPU_DECODE": decode}, launcher=launcher)
    assert rcs == [0, 0], outs
    assert "could not be verified" in outs[0]
    assert same(single, out)


@pytest.mark.parametrize("launcher", ["torch", "cxx"])
def test_an_error_on_one_rank_ends_all_ranks(inputs, launcher):
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
    rcs, outs = run_ranks(2, ["-r", fa1, "-o", str(d / "err.bamqc"), "-c", "chr1,chr2", "-i", "800", bam], launcher=launcher)
    assert rcs == [1, 1], outs
    assert any("Could not read fasta record" in o for o in outs)


@pytest.mark.parametrize("launcher", ["cxx", "torch"])
def test_rccl_set_up_and_reduce_run_on_one_card(inputs, launcher):
    """One process through the sharded path with the REAL backend (the other tests of this module sum through gloo / pipes, the
    box has one card): communicator set-up and the reduce of the state vector on a device pointer — ncclCommInitRank + ncclReduce
    from C++ (`bamqualcheck --gpus 1`, BQC_GPUS_FORCE=1), init_process_group("nccl") + dist.reduce from the torch launcher."""
    bam, fa, single, d = inputs
    out = str(d / ("rccl1_%s.bamqc" % launcher))
    args = ["-r", fa, "-o", out, "-c", "chr1,chr2", "-i", "800", bam]
    if launcher == "cxx":
        r = subprocess.run([EXE, "--gpus", "1"] + args, env=dict(os.environ, BQC_GPUS_FORCE="1", BQC_GPU_DECODE="1"), capture_output=True, text=True, timeout=600)
    else:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ, BQC_FORCE_SHARD_PATH="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, "-m", "bamqc_amd.dist_cli"] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert same(single, out)


def test_a_worker_that_cannot_enter_the_reduce_ends_the_run(inputs):
    """BQC_REDUCE unset: the RCCL path (`--gpus 1` forced through it — RCCL refuses two ranks on the one card of the test box).  A worker
    whose device allocation for the state vector fails must not leave its peers inside ncclReduce for ever: it exits at once
    (multi_gpu.cpp, as start_rccl's die()), the front end sees it go and reports failure — exit status 1, well inside the time limit."""
    bam, fa, single, d = inputs
    out = str(d / "fault_nomem.bamqc")
    env = dict(os.environ, BQC_GPUS_FORCE="1", BQC_TEST_WORKER_FAULT="0:nomem")
    env.pop("BQC_REDUCE", None)
    r = subprocess.run([EXE, "--gpus", "1", "-r", fa, "-o", out, "-c", "chr1,chr2", "-i", "800", bam], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1, r.stdout + r.stderr
    assert "no device memory for the state vector" in r.stderr


@pytest.mark.parametrize("reduce", ["rccl", "pipe"])
def test_a_wedged_worker_meets_the_front_ends_deadline(inputs, reduce):
    """A worker that never answers (wedged on its card) and never exits: the front end's per-phase deadline (BQC_GPUS_TIMEOUT) ends the
    run with status 1 and kills the workers."""
    import time
    bam, fa, single, d = inputs
    out = str(d / ("fault_wedge_%s.bamqc" % reduce))
    if reduce == "rccl":
        env = dict(os.environ, BQC_GPUS_FORCE="1", BQC_TEST_WORKER_FAULT="0:wedge", BQC_GPUS_TIMEOUT="3")
        env.pop("BQC_REDUCE", None)
        world = 1
    else:
        env = dict(os.environ, BQC_GPUS_SHARE_DEVICE="1", BQC_TEST_WORKER_FAULT="1:wedge", BQC_GPUS_TIMEOUT="3")
        world = 2
    t0 = time.time()
    r = subprocess.run([EXE, "--gpus", str(world), "-r", fa, "-o", out, "-c", "chr1,chr2", "-i", "800", bam], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1, r.stdout + r.stderr
    assert "did not answer in time" in r.stderr
    assert time.time() - t0 < 60


def test_usage_errors_are_reported_once_by_the_front_end(inputs):
    """`--gpus N` checks the command line before it forks: one message, no worker started; `-o x.bam` behind the input does not
    confuse the front end about which file is the input (it asks the program's own parser)."""
    bam, fa, single, d = inputs
    r = subprocess.run([EXE, "--gpus", "3", "-r", fa, "-o", str(d / "u.bamqc"), "--no-such-option", bam], capture_output=True, text=True, timeout=60,
                       env=dict(os.environ, BQC_GPUS_SHARE_DEVICE="1"))
    assert r.returncode == 1
    assert r.stderr.count("illegal option") == 1, r.stderr
    out = str(d / "looks_like_input.bam")
    rcs, outs = run_ranks(2, ["-r", fa, "-c", "chr1,chr2", "-i", "800", bam, "-o", out], launcher="cxx")
    assert rcs == [0, 0], outs
    assert same(single, out)


@pytest.mark.parametrize("world,batch", [(2, "1000000"), (3, "20011")])
def test_single_read_group_shards_set_aside_on_the_card(tmp_path, world, batch):
    """ONE read group and the reader on the card: the workers' coverage anchors are made on the card (csrc/k_anchor.hip), also by the workers
    that start inside the stream — their reads up to the first certain reset (here: the next chromosome, the data has no gaps) are set
    aside there exactly as the host's pass sets them aside; small batches: the pending phase spans many of them.  Same bytes as the
    single-process run and as the host's pass (BQC_DEVICE_ANCHORS=0)."""
    bam, fa = str(tmp_path / "one.bam"), str(tmp_path / "one.fa")
    hostio.synth_write(bam, fa, seed=78, n_reads=250_000, ref_names=["chr1", "chr2", "chrM"], ref_lens=[2_000_000, 400_000, 20_000], n_lanes=1)
    single = str(tmp_path / "single.bamqc")
    r = subprocess.run([EXE, "-r", fa, "-o", single, "-c", "chr1,chr2", bam], capture_output=True, text=True, env=dict(os.environ, BQC_GPU_DECODE="1", BQC_TIMING="1"))
    assert r.returncode == 0, r.stderr
    assert " batches anchored on the card" in r.stderr and "[timing] 0 batches anchored" not in r.stderr, r.stderr
    outs_seen = []
    for anchors in ("1", "0"):
        out = str(tmp_path / ("sharded_%s.bamqc" % anchors))
        rcs, outs = run_ranks(world, ["-r", fa, "-o", out, "-c", "chr1,chr2", "--batch-reads", batch, bam],
                              env_extra={"BQC_GPU_DECODE": "1", "BQC_TIMING": "1", "BQC_DEVICE_ANCHORS": anchors}, launcher="cxx")
        assert rcs == [0] * world, outs
        assert same(single, out)
        outs_seen.append(outs[0])
    import re
    counts = [int(x) for x in re.findall(r"\[timing\] (\d+) batches anchored on the card", outs_seen[0])]
    assert len(counts) == world and all(c > 0 for c in counts), outs_seen[0]  # every worker, the ones in the middle of the stream too
    assert all(int(x) == 0 for x in re.findall(r"\[timing\] (\d+) batches anchored on the card", outs_seen[1]))
