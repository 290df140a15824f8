"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The record stream shards by chromosome: every statistic of the hot path is a sum over reads except
the coverage-depth histogram, whose window state machine (reference src/OverallNumbers.hpp:84-110)
resets whenever the chromosome changes — and a reset flushes exactly the two windows the end-of-run
flush would (bamqualcheck.cpp:447-453).  So ranks that each own whole chromosomes of a
coordinate-sorted BAM produce state vectors that ADD to the single-process result.  The only
collective is one reduce (uint64 sum, carried as int64) of the flat state vector at the end.
"""
import os

import numpy as np


def plan_shards(ref_lens, world_size):
    """Longest-processing-time assignment of contigs to ranks. Returns owner[rid] (np.int32).
    Deterministic: ties break on the lower contig id / lower rank."""
    order = sorted(range(len(ref_lens)), key=lambda i: (-int(ref_lens[i]), i))
    load = [0] * world_size
    owner = np.zeros(len(ref_lens), np.int32)
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += int(ref_lens[i])
    return owner


def reduce_state(vec, dst=0):
    """Sum the flat state vector over ranks onto `dst`. `vec` is a torch int64 tensor (cuda for
    nccl, cpu for gloo) holding uint64 words; two's-complement addition wraps identically."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(vec, dst=dst, op=dist.ReduceOp.SUM)
    return vec


def gather_lane_names(names, dst=0):
    """Union of the lane-name maps (getLane inserts unknown @RG IDs with index 0,
    bamqualcheck.cpp:86) so the output lists the same blocks as a single process."""
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return dict(names)
    objs = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(dict(names), objs, dst=dst)
    if dist.get_rank() != dst:
        return None
    merged = {}
    for d in objs:
        for k, v in d.items():
            merged.setdefault(k, v)
    return merged


def run_sharded(bam, fasta, out, chroms=None, isize=1000, klist=(32,), qlist=(17,), backend=None, device=None,
                batch_reads=1 << 20, max_read_len=65536, hist_cap=65536):
    """bamqualcheck over a coordinate-sorted BAM, sharded by chromosome across the ranks of the
    initialised process group (or a single process). Rank 0 writes `out`. Returns 0 / error code."""
    import torch
    import torch.distributed as dist
    from . import Aggregator, hostio
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    backend = backend or (dist.get_backend() if dist.is_initialized() else "none")
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    f = hostio.BamFile(bam)
    chroms = chroms if chroms is not None else ",".join("chr%d" % i for i in range(1, 23))
    main = np.array([1 if n in chroms.split(",") else 0 for n in f.ref_names], np.uint8)
    f.set_main_chrom(main)
    owner = plan_shards(f.ref_lens, world)
    f.set_rid_filter((owner == rank).astype(np.uint8), keep_unplaced=(rank == world - 1))
    try:
        fa = hostio.load_fasta(fasta)
    except IOError:
        fa = []
    fidx = np.full(max(1, len(f.ref_names)), -1, np.int32)
    for r, name in enumerate(f.ref_names):
        for i, (n, _) in enumerate(fa):
            if n == name:
                fidx[r] = i
                break
    agg = Aggregator(n_lanes=max(1, f.lane_count), n_refs=len(f.ref_names), isize=isize, main_chrom=main, fasta_index=fidx,
                     max_read_len=max_read_len, hist_cap=hist_cap, klist=klist, qlist=qlist, device=device)
    for r in range(len(f.ref_names)):
        if fidx[r] >= 0 and owner[r] == rank:
            agg.set_reference(r, fa[fidx[r]][1])
    last_rid = -2
    for cols in f.batches(max_reads=batch_reads):
        rid = cols["rid"]
        placed = rid[rid >= 0]
        if len(placed) and (np.any(np.diff(placed) < 0) or placed[0] < last_rid):
            raise RuntimeError("multi-GPU sharding needs a coordinate-sorted BAM (reference ids went backwards)")
        if len(placed):
            last_rid = int(placed[-1])
        agg.submit(cols)
    names = gather_lane_names(dict(f.lanes()))
    if world > 1:
        if backend == "nccl":
            vec = torch.empty(agg.state_words, dtype=torch.int64, device=torch.device("cuda", device))
            agg.state_export_device(vec.data_ptr())
            reduce_state(vec)
            if rank == 0:
                agg.state_import_device(vec.data_ptr())
        else:  # gloo: host tensors
            vec = torch.from_numpy(agg.state_export_host().view(np.int64))
            reduce_state(vec)
            if rank == 0:
                agg.state_import_host(vec.numpy().view(np.uint64))
    if rank == 0:
        agg.finalize_raw()
        order = sorted(names.items())
        agg.write_bamqc(out, sample_id=f.sample_id, lane_names=[n for n, _ in order], lane_index=[i for _, i in order])
    agg.close()
    return 0
