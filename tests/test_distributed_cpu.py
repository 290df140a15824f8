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


def test_plan_shards_covers_every_contig_once_and_balances():
    from bamqc_amd.synth import GRCH38
    for world in (1, 2, 4, 8):
        owner = D.plan_shards(GRCH38, world)
        assert owner.shape == (24,) and set(owner.tolist()) == set(range(world))
        loads = np.bincount(owner, weights=np.array(GRCH38, float), minlength=world)
        assert loads.max() / loads.mean() < 1.08  # LPT over 24 contigs is within a few % of perfect
    assert D.plan_shards([5, 5, 5], 2).tolist() == [0, 1, 0]  # deterministic tie-breaks


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
