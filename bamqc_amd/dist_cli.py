"""`python -m torch.distributed.run --nproc-per-node N -m bamqc_amd.dist_cli -r FASTA -o OUT [options] BAM`

bamqualcheck over the GPUs of one node, one process per GPU: the BAM file's byte stream is split between the processes
(bamqc_amd/distributed.py), the state vectors are summed with one RCCL reduce.  Options and output as the single-process
program's (`bin/bamqualcheck`); `--backend gloo` rehearses the same path without RCCL."""
import os
import sys
import time


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    backend = "nccl"
    if "--backend" in argv:
        i = argv.index("--backend")
        backend = argv[i + 1]
        del argv[i:i + 2]
    import torch
    import torch.distributed as dist
    from . import distributed as D
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("BQC_FORCE_SHARD_PATH") == "1"  # (one process through the sharded path: backend set-up + reduce on one card)
    if force and world == 1:
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        # -s 0 (table seeded from the time, RepHash.cpp:5-7): every process must use the same table
        for k, a in enumerate(argv):
            if a in ("-s", "--seed") and k + 1 < len(argv) and argv[k + 1].lstrip("+-").isdigit() and int(argv[k + 1]) == 0:
                seed = [int(time.time()) or 1]
                dist.broadcast_object_list(seed, src=0)
                argv[k + 1] = str(seed[0])
    rc = D.run_sharded(["bamqualcheck"] + argv, backend=backend if (world > 1 or force) else "none", device=local if backend == "nccl" else 0)
    if world > 1 or force:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
