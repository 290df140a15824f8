"""CPU: the N>1 path with world_size 2 over gloo — shard planning, the state-vector reduce
(uint64 sums carried as int64) and the lane-name union.  No GPU compute here."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bamqc_amd import distributed as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    words = rng.integers(0, 2 ** 64, size=5000, dtype=np.uint64)
    words[0] = np.uint64(2 ** 64 - 1 - rank)  # force wrap-around
    vec = torch.from_numpy(words.view(np.int64).copy())
    D.reduce_state(vec, dst=0)
    names = D.gather_lane_names({"L1": 0, "L2": 1} if rank == 0 else {"L1": 0, "zz_unknown": 0})
    if rank == 0:
        q.put((vec.numpy().view(np.uint64).copy(), names))
    dist.barrier()
    dist.destroy_process_group()


def test_state_reduce_and_lane_union_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, names = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = np.zeros(5000, np.uint64)
    for r in range(world):
        rng = np.random.default_rng(100 + r)
        w = rng.integers(0, 2 ** 64, size=5000, dtype=np.uint64)
        w[0] = np.uint64(2 ** 64 - 1 - r)
        want = want + w  # numpy uint64 addition wraps mod 2^64
    assert np.array_equal(got, want)
    assert names == {"L1": 0, "L2": 1, "zz_unknown": 0}


def test_split_and_fasta_order_checks():
    ok = [(True, 0, 100, 0, 5), (True, 100, 200, 5, 7), (False, 200, 200, 0, 0), (True, 200, 2 ** 64 - 1, 7, 0)]
    assert D.split_is_consistent(ok)
    assert not D.split_is_consistent([(True, 0, 100, 0, 5), (True, 100, 200, 6, 7)])      # the successor guessed another record start
    assert not D.split_is_consistent([(True, 0, 100, 0, 5), (True, 104, 200, 5, 7)])      # ... or another block boundary
    assert D.fasta_order_is_consistent([(0, 3), (-1, -1), (3, 5)])
    assert not D.fasta_order_is_consistent([(2, 3), (1, 5)])
    assert D.merge_lane_names([{"L1": 0, "L2": 1}, {"L1": 0, "zz": 0}]) == {"L1": 0, "L2": 1, "zz": 0}


def test_byte_range_shards_partition_the_record_stream(tmp_path):
    """CPU: the shard reader.  For 2..16 shards of a BAM with two read groups every record is read by exactly one shard, in
    order, and the chain check holds (a shard's last record ends where its successor's first begins); with
    BQC_TEST_SHARD_SKEW (wrong guesses) the chain check fails — that is what sends the program to its unsharded fallback."""
    import subprocess
    from bamqc_amd import hostio
    bam = str(tmp_path / "r.bam")
    hostio.synth_stream(bam, None, 31, 400_000, ["chr1", "chr2"], [3_000_000, 1_000_000], n_lanes=2, level=1)
    size = os.path.getsize(bam)
    keys = ("flag", "pos", "rid", "l_seq", "nm", "seq", "qual", "cigar", "lane")

    def read(f):
        bl = list(f.batches(max_reads=77_777))
        return {k: np.concatenate([b[k] for b in bl]) if bl else np.zeros(0) for k in keys}

    whole = read(hostio.BamFile(bam))
    assert len(whole["flag"]) == 400_000
    for n in (2, 3, 7, 16):
        hints = [size * r // n for r in range(n + 1)]
        parts, infos = [], []
        for r in range(n):
            f = hostio.BamFile(bam, hints[r], None if r == n - 1 else hints[r + 1])
            parts.append(read(f))
            infos.append(f.range_info)
        for k in keys:
            assert np.array_equal(np.concatenate([p[k] for p in parts]), whole[k]), (n, k)
        assert D.split_is_consistent([(i[0] < min(i[1], size), i[0], i[1], i[2], i[3]) for i in infos])
    code = ("import sys; sys.path.insert(0, %r)\nfrom bamqc_amd import hostio\nf = hostio.BamFile(%r, %d, None)\n"
            "n = sum(len(b['flag']) for b in f.batches())\nprint(f.range_info[2])\n") % (ROOT, bam, size // 2)
    good = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    bad = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, BQC_TEST_SHARD_SKEW="1"))
    assert good.returncode == 0 and bad.returncode == 0 and good.stdout != bad.stdout


def test_cxx_launcher_without_a_gpu_fails_cleanly(tmp_path):
    """`bamqualcheck --gpus 2` on a box without a GPU: every worker fails at context creation, reports through the shard hook, the
    front end tells all of them to stop and returns 1 — no worker is left waiting for the others (the coordinator's error path;
    the success paths need a card: tests/test_gpu_sharded.py)."""
    import subprocess
    import torch
    from bamqc_amd import hostio
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box WITHOUT a GPU")
    bam, fa = str(tmp_path / "s.bam"), str(tmp_path / "s.fa")
    hostio.synth_write(bam, fa, seed=3, n_reads=5000, ref_names=["chr1"], ref_lens=[200_000])
    exe = os.path.join(ROOT, "bin", "bamqualcheck")
    for env in ({}, {"BQC_REDUCE": "pipe"}):
        r = subprocess.run([exe, "--gpus", "2", "-r", fa, "-o", str(tmp_path / "o.bamqc"), "-c", "chr1", bam], env=dict(os.environ, **env), capture_output=True, text=True, timeout=120)
        assert r.returncode == 1, (r.stdout, r.stderr)
        assert "no HIP device" in r.stderr or "hipSetDevice" in r.stderr
    r = subprocess.run([exe, "--gpus", "3", "--version"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.count("bamqualcheck version") == 1  # answered once, by the front end
    r = subprocess.run([exe, "--gpus", "0", "-r", fa, "-o", str(tmp_path / "o.bamqc"), bam], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "--gpus" in r.stderr
