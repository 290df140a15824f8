"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The record stream is split by BYTES of the compressed file: process i of n runs the program (`bqc_main_shard`: the same C++
reader, decode thread and asynchronous submit pipeline as the single-GPU program) over the records that start in its n-th of
the file — no process inflates or decodes another's part, and the split does not care about chromosomes or sort order.
Every statistic of the path is a sum over reads except the coverage-depth histogram, whose window state machine (reference
src/OverallNumbers.hpp:84-110) depends on the reads before: a process that does not start at the stream's first record sets
aside, per read group, its reads up to the first one at which the state machine resets whatever its state (another chromosome,
or more than 2000 positions from the read before) and runs them once its predecessor's final state has arrived
(include/bamqc.h: bqc_shard_resolve / bqc_shard_export).  Collectives: one all_gather of a few words per process (status, split
check, FASTA order), one point-to-point hand-over of the coverage state per neighbour pair (8 KB per read group), ONE reduce
(uint64 sum, carried as int64) of the flat state vector onto process 0, which writes the output.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import _abi, _lib


def reduce_state(vec, dst=0):
    """Sum the flat state vector over ranks onto `dst`. `vec` is a torch int64 tensor (cuda for
    nccl, cpu for gloo) holding uint64 words; two's-complement addition wraps identically."""
    import torch.distributed as dist
    if dist.is_initialized():  # (also with one process: the collective then runs on its own — a smoke test of the backend)
        dist.reduce(vec, dst=dst, op=dist.ReduceOp.SUM)
    return vec


def merge_lane_names(maps):
    """Union of the lane-name maps of the processes, first come first kept (getLane inserts unknown @RG IDs with index 0,
    bamqualcheck.cpp:86), so that the output lists the same blocks as a single process."""
    merged = {}
    for d in maps:
        for k, v in d.items():
            merged.setdefault(k, v)
    return merged


def gather_lane_names(names, dst=0):
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return dict(names)
    objs = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(dict(names), objs, dst=dst)
    return merge_lane_names(objs) if dist.get_rank() == dst else None


def split_is_consistent(ranges):
    """ranges[i] = (has_blocks, begin_block, end_block, first, over) of shard i.  Every shard that starts in the middle of the
    file GUESSED where its first record starts; the guess is right iff the predecessor's last record ends exactly there
    (`over` of the predecessor, relative to the block both call their boundary)."""
    prev = None
    for has, b0, b1, first, over in ranges:
        if not has:
            continue  # (a shard without blocks: its neighbours meet directly)
        if prev is not None:
            if prev[2] != b0 or prev[4] != first:
                return False
        prev = (has, b0, b1, first, over)
    return True


def fasta_order_is_consistent(spans):
    """spans[i] = (first, last) FASTA position of shard i's triplet-eligible reads (-1: none): the forward-only FASTA scan
    (TripletCounting.hpp:254-259) of the whole stream succeeds iff no shard starts before its predecessors ended."""
    last = -1
    for first, lst in spans:
        if first < 0:
            continue
        if first < last:
            return False
        last = max(last, lst)
    return True


def run_sharded(argv, backend=None, device=None):
    """`bamqualcheck` (argv as for the program, argv[0] = program name) over the initialised process group: returns the exit
    status of this process (0 on every process on success)."""
    import torch
    import torch.distributed as dist
    lib = _lib.load()
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    backend = backend or (dist.get_backend() if dist.is_initialized() else "none")
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    argv = list(argv) + ["--device", str(device)]
    cargs = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
    if world == 1 and not (dist.is_initialized() and os.environ.get("BQC_FORCE_SHARD_PATH") == "1"):
        return int(lib.bqc_main(len(argv), cargs))
    tdev = torch.device("cuda", device) if backend == "nccl" else torch.device("cpu")
    keep = {}

    def hook_body(info_p, out_p):
        info = info_p.contents
        ctx = info.ctx
        mine = dict(status=int(info.status), has_ctx=bool(ctx), b0=int(info.begin_block), b1=int(info.end_block), first=int(info.first),
                    over=int(info.over), span=(-1, -1),
                    lanes={info.lane_names[i].decode(): int(info.lane_index[i]) for i in range(info.n_lane_names)})
        if ctx and not mine["status"]:
            sp = (C.c_int32 * 2)()
            if lib.bqc_shard_fasta_span(ctx, sp):
                print((lib.bqc_last_error(ctx) or b"").decode(), flush=True)
                mine["status"] = 1
            mine["span"] = (int(sp[0]), int(sp[1]))
        every = [None] * world
        dist.all_gather_object(every, mine)  # ---- agree on the status before any data collective
        if any(e["status"] for e in every):
            return _abi.SHARD_FAIL
        if not any(e["has_ctx"] for e in every):
            return _abi.SHARD_DONE  # (no @RG line: there are no counters)
        if not fasta_order_is_consistent([e["span"] for e in every]):
            if rank == 0:
                print("ERROR: Could not read fasta record (the BAM file's contig order runs backwards in the FASTA file)", flush=True)
            return _abi.SHARD_FAIL
        size = int(lib.bqc_file_size([a for a in argv if a.endswith(".bam")][-1].encode()))
        ranges = [(e["b0"] < min(e["b1"], size), e["b0"], e["b1"], e["first"], e["over"]) for e in every]
        if not split_is_consistent(ranges):
            if rank == 0:
                print("bamqualcheck: the split of the file could not be verified; processing it in one process", flush=True)
            return _abi.SHARD_FALLBACK if rank == 0 else _abi.SHARD_DONE
        # ---- coverage state down the chain: predecessor's final state in, own final state out
        nbytes = int(lib.bqc_shard_state_bytes(ctx))
        if rank > 0:
            t = torch.empty(nbytes, dtype=torch.uint8, device=tdev)
            dist.recv(t, src=rank - 1)
            buf = np.ascontiguousarray(t.cpu().numpy())
            rc = lib.bqc_shard_resolve(ctx, buf.ctypes.data_as(C.c_void_p))
        else:
            rc = 0
        if not rc:
            if rank < world - 1:
                buf = np.zeros(nbytes, np.uint8)
                rc = lib.bqc_shard_export(ctx, buf.ctypes.data_as(C.c_void_p))
            else:
                rc = lib.bqc_flush(ctx)  # the last shard ends as a whole stream does: its two live windows per read group are flushed
        if rank < world - 1:  # (sent also after a failure: the successor is waiting)
            dist.send(torch.from_numpy(buf if not rc else np.zeros(nbytes, np.uint8)).to(tdev), dst=rank + 1)
        bad = torch.tensor([1 if rc else 0], dtype=torch.int32, device=tdev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()):
            if rc:
                print((lib.bqc_last_error(ctx) or b"").decode(), flush=True)
            return _abi.SHARD_FAIL
        # ---- ONE reduce of the flat state vector onto process 0
        words = int(lib.bqc_state_words(ctx))

        def all_ok(rc):  # every process learns whether every process's step worked, before the next collective or the output
            bad = torch.tensor([1 if rc else 0], dtype=torch.int32, device=tdev)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            return not int(bad.item())

        if backend == "nccl":
            vec = torch.zeros(words, dtype=torch.int64, device=tdev)
            rc = lib.bqc_state_export(ctx, C.c_void_p(vec.data_ptr()))
            if not all_ok(rc):
                if rc:
                    print((lib.bqc_last_error(ctx) or b"").decode(), flush=True)
                return _abi.SHARD_FAIL
            reduce_state(vec)
            rc = lib.bqc_state_import(ctx, C.c_void_p(vec.data_ptr())) if rank == 0 else 0
        else:  # gloo: host tensors
            host = np.zeros(words, np.uint64)
            rc = lib.bqc_state_export_host(ctx, host.ctypes.data_as(_abi.u64p))
            if not all_ok(rc):
                if rc:
                    print((lib.bqc_last_error(ctx) or b"").decode(), flush=True)
                return _abi.SHARD_FAIL
            vec = torch.from_numpy(host.view(np.int64))
            reduce_state(vec)
            rc = lib.bqc_state_import_host(ctx, host.ctypes.data_as(_abi.u64p)) if rank == 0 else 0
        if not all_ok(rc):
            if rc:
                print((lib.bqc_last_error(ctx) or b"").decode(), flush=True)
            return _abi.SHARD_FAIL
        if rank != 0:
            return _abi.SHARD_DONE
        merged = sorted(merge_lane_names([e["lanes"] for e in every]).items())  # writeOutput iterates a std::map: lexicographic
        names = (C.c_char_p * len(merged))(*[k.encode() for k, _ in merged])
        idx = np.ascontiguousarray([v for _, v in merged], np.uint32)
        keep["names"], keep["idx"] = names, idx
        out = out_p.contents
        out.n_lane_names = len(merged)
        out.lane_names = names
        out.lane_index = idx.ctypes.data_as(_abi.u32p)
        return _abi.SHARD_WRITE

    def hook(_user, info_p, out_p):
        # (a ctypes callback that raises returns 0 to its caller after printing the traceback: 0 is SHARD_FAIL, and the other
        # processes are told so that none of them is left waiting or writes a sum that lacks this process's part)
        try:
            return hook_body(info_p, out_p)
        except BaseException:  # noqa: BLE001 - whatever it was, this process's result is void
            import traceback
            traceback.print_exc()
            sys.stdout.flush()
            return _abi.SHARD_FAIL

    cb = _abi.SHARD_HOOK(hook)
    return int(lib.bqc_main_shard(len(argv), cargs, rank, world, cb, None))
