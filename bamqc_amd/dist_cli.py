"""`python -m torch.distributed.run --nproc-per-node N -m bamqc_amd.dist_cli -r FASTA -o OUT [-c ..] [-i ..] BAM`

bamqualcheck sharded by chromosome over the GPUs of one node (one process per GPU, RCCL reduce of
the state vector at the end).  Same options as the single-process program."""
import argparse
import os
import sys


def main(argv=None):
    ap = argparse.ArgumentParser(prog="bamqc_amd.dist_cli")
    ap.add_argument("bam")
    ap.add_argument("-r", "--reference", required=True)
    ap.add_argument("-o", "--output-file", required=True)
    ap.add_argument("-c", "--chromosomes", default=None)
    ap.add_argument("-i", "--insert-size", type=int, default=1000)
    ap.add_argument("-k", "--kmer-size", default="32")
    ap.add_argument("-q", "--quality-cutoff", default="17")
    ap.add_argument("--no-sketch", action="store_true")
    ap.add_argument("--backend", default="nccl")
    a = ap.parse_args(argv)
    import torch
    import torch.distributed as dist
    from . import distributed as D
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.backend)
    if int(os.environ.get("RANK", "0")) == 0:
        open(a.output_file, "wb").close()  # output is opened (truncated) before the scan
    ks = () if a.no_sketch else tuple(int(x) for x in a.kmer_size.split(",") if x)
    qs = () if a.no_sketch else tuple(int(x) for x in a.quality_cutoff.split(",") if x)
    try:
        rc = D.run_sharded(a.bam, a.reference, a.output_file, chroms=a.chromosomes, isize=a.insert_size, klist=ks, qlist=qs,
                           device=local)
    except Exception as e:  # message + exit 1, like the reference
        print("ERROR: %s" % e, file=sys.stderr)
        rc = 1
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
