#!/usr/bin/env python3
"""bench.py — BAM records/s of the per-read aggregation hot path on N MI355X (one process per GPU).

A "step" is one pass of the hot path (k_short + k_cov; k_reads + k_long for long reads) over one device-resident batch of
synthetic 150 bp paired-end reads (config 2 of BASELINE.json: 10 M reads over 4 x 25 Mb contigs).
Each rank owns its own batch (records shard by read batch: weak scaling, no data-path collective);
at the end of the job the flat uint64 state vectors are summed onto rank 0 with one RCCL reduce.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("config2", "config5", "config3"), default="config2",
                    help="config2 (default, the configuration the metric is quoted on): 10 M x 150 bp PE; config5: long reads, 10 kb, indel / soft-clip heavy; "
                         "config3: the 30x WGS-scale BAM FILE (618 M reads, ~54 GB at BGZF level 1: written first, ~3 min) through bin/bamqualcheck, live — "
                         "the line's value is then the program's reads/s, not a kernel rate")
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU per step (default: 10 M for config2, 100 k for config5)")
    ap.add_argument("--read-len", type=int, default=None)
    ap.add_argument("--sketch", action="store_true", help="steps include the k-mer sketch of the program's default options (-k 32 -q 17)")
    ap.add_argument("--cpu-sample", type=int, default=3_000_000, help="reads timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the BAM file -> bin/bamqualcheck -> .bamqc leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra device timings (default options with sketch; long reads)")
    ap.add_argument("--e2e-prefix", type=int, default=1_000_000, help="reads of the e2e input checked against the oracle")
    ap.add_argument("--no-large", action="store_true", help="skip e2e_large (100 M reads over 24 GRCh38-length contigs as a 10 GB file) and e2e_shard_proxy (one worker's eighth of config 3)")
    ap.add_argument("--large-reads", type=int, default=100_000_000)
    ap.add_argument("--only-large", action="store_true", help="(development) only e2e_large and e2e_shard_proxy, printed as one JSON object")
    args = ap.parse_args()
    if args.workload == "config3":
        return config3_line(args)
    if args.only_large:
        print(json.dumps(large_legs(args)), flush=True)
        return
    long_reads = args.workload == "config5"
    if args.reads is None:
        args.reads = 100_000 if long_reads else 10_000_000
    if args.read_len is None:
        args.read_len = 10_000 if long_reads else 150

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    one_gpu_rehearsal = os.environ.get("BQC_BENCH_ONE_GPU") == "1"  # rehearse the N > 1 code path on a 1-GPU box: gloo, every rank on cuda:0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu_rehearsal:
            local = 0
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    if world != args.gpus:
        sys.exit("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    dev = torch.device("cuda", local)
    sharded_flag = os.path.join(tempfile_dir(), "bqc_bench_sharded_%s.done" % os.environ.get("MASTER_PORT", "0"))
    if world > 1 and rank == 0 and os.path.exists(sharded_flag):
        os.remove(sharded_flag)

    from bamqc_amd import Aggregator, _abi, synth
    lens = [250_000_000] if long_reads else [25_000_000] * 4
    seed = 1005 if long_reads else 1002  # 1000 + config index
    opts = dict(n_refs=len(lens), n_lanes=1, isize=30_000 if long_reads else 1000, max_read_len=max(1024, 16_384 if long_reads else 0, args.read_len),
                hist_cap=16_384 if long_reads else 4096, device=local)
    if args.sketch:
        opts.update(klist=(32,), qlist=(17,))
    refs = [synth.reference(seed, i, n) for i, n in enumerate(lens)]
    e2e, e2e_big = None, None
    if world == 1 and rank == 0 and not args.no_e2e and not long_reads:
        # first, while this process has not touched the GPU: the program's start-up (HIP initialisation, context creation) was
        # measured to take 0.25-0.7 s instead of 0.1 s next to another process that holds a context on the card or is handing one back
        e2e = e2e_leg(args, refs, None)
        if not args.no_large:
            try:
                e2e_big = large_legs(args)
            except BaseException as e:  # noqa: BLE001 - the line is printed whatever happens in this leg
                e2e_big = {"e2e_large": {"error": repr(e)[:600]}}
    t0 = time.time()
    cols = synth.batch(seed, args.reads, lens, refs, read_len=args.read_len, first_read_index=rank * args.reads, isize=opts["isize"], long_reads=long_reads)
    t_gen = time.time() - t0
    agg = Aggregator(**opts)
    for i, r in enumerate(refs):
        agg.set_reference(i, r)
    t0 = time.time()
    db = agg.upload(cols)  # host pass + H2D; inputs are resident in HBM before the timed region
    t_up = time.time() - t0
    abytes = db.algorithmic_bytes
    agg.set_timing(True)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        agg.process(db)
    agg.sync()
    agg.reset()
    kt = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        agg.process(db)  # device pre-pass (k_prep) + hot kernels over the resident batch
        for k, v in agg.last_timing().items():  # HIP events on the library's own stream
            kt.setdefault(k, []).append(v)
    agg.sync()
    if world > 1:  # end of job: one RCCL reduce of the flat state vector onto rank 0
        vec = torch.empty(agg.state_words, dtype=torch.int64, device=dev)
        agg.state_export_device(vec.data_ptr())
        if one_gpu_rehearsal:
            host = vec.cpu()
            dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
            vec.copy_(host)
        else:
            dist.reduce(vec, dst=0, op=dist.ReduceOp.SUM)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_gpu_rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if rank == 0:
            agg.state_import_device(vec.data_ptr())
            total = agg.finalize()
            assert int(total[0]["scalars"][4]) == (args.reads * args.steps * world - int(total[0]["scalars"][0]) - int(total[0]["scalars"][3])) % 2 ** 32

    e2e_sharded = None
    if world > 1 and not args.no_e2e and not long_reads:
        # What N GPUs do for a user: the PROGRAM `bin/bamqualcheck --gpus N` (one worker per card on its byte range of the file, ONE RCCL
        # reduce from C++) on a BAM file, as child processes of rank 0 — a failure or a hang of that leg cannot take the line below with
        # it.  The other ranks wait for a file, not in a collective: an RCCL barrier would spin on their cards beside the workers.
        if rank == 0:
            try:
                e2e_sharded = e2e_sharded_leg(args, world, one_gpu_rehearsal)
            except BaseException as e:  # noqa: BLE001 - whatever went wrong there, the benchmark's line is still printed
                e2e_sharded = {"error": repr(e)[:400]}
            open(sharded_flag, "w").close()
        else:
            t_wait = time.time()
            while not os.path.exists(sharded_flag) and time.time() - t_wait < 900:
                time.sleep(0.2)

    if rank == 0:
        ms_step = elapsed * 1e3 / args.steps
        value = world * args.reads * args.steps / elapsed
        kavg = {k: float(np.mean(v)) for k, v in kt.items()}
        dom = max(kavg, key=kavg.get)
        peak = 8000.0  # GB/s HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling
        ach = abytes / (kavg[dom] * 1e-3) / 1e9
        # HBM traffic per launch and the issue-slot counters come from rocprofv3 PMC passes (profiles/collect_r2.sh): counters
        # cannot be read in-process
        traffic, limiter, tname = None, None, None
        try:
            tname = next(f for f in ("r4_traffic.json", "r3_traffic.json", "r2_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
            tj = json.load(open(os.path.join(ROOT, "profiles", tname)))
            if tj.get("kernel") == dom and args.reads == 10_000_000 and args.read_len == 150 and not args.sketch:
                traffic = tj["traffic_bytes_per_launch"]
                limiter = tj.get("limiter")
        except Exception:
            pass
        out = {
            "metric": "BAM records/sec (and GB/s vs HBM roofline), 150 bp PE, 1/2/4/8 MI355X",
            "value": value, "unit": "reads/s",
            "value_kind": "device-resident kernel step (inputs in HBM when the clock starts: the contract's `value`); BAM-file throughput of the program is e2e.reads_per_s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u32/u64 integer", "data": "synthetic",
            "config": {"workload": ("config 5: %d reads x %d bases per GPU, 20-60 CIGAR operations, 1 x 250 Mb contig, -i 30000" if long_reads else
                                    "config 2: %d reads x %d bp PE per GPU, 4 x 25 Mb contigs, 1 lane") % (args.reads, args.read_len) +
                                   ", device-resident raw SoA batch; a step = device pre-pass + hot kernels" + (" + k-mer sketch k32 q17" if args.sketch else ""),
                       "reads_per_gpu_per_step": args.reads, "parallelism": "shard by read batch; RCCL reduce of state vector at end"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": peak, "unit": "GB/s", "frac": ach / peak,
                         "frac_vs_measured_copy_6290": ach / 6290.0, "traffic": traffic,
                         "traffic_source": ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x2 calibrated; limiter against the MEASURED issue ceiling of profiles/r3_valu_issue.json)" % tname) if traffic else None,
                         "limiter": limiter or {"what": "valu+lds issue (integer SWAR and LDS atomics per base), not HBM: see DESIGN.md 4.2"},
                         "algorithmic_bytes_per_launch": abytes, "bytes_per_read": abytes / args.reads,
                         "kernel_ms": kavg},
            "host": {"generate_s": t_gen, "host_pass_upload_s": t_up,
                     "pcie_inclusive_reads_per_s": args.reads / (t_up + ms_step * 1e-3)},
        }
        if world == 1 and not args.no_cpu and not long_reads:
            out["cpu_baseline"] = cpu_baseline(cols, refs, args, agg, db)
        db.free()
        agg.close()
        del cols
        if world == 1 and not args.no_extra and not long_reads and not args.sketch:
            out["extra"] = extra_timings(refs)
        try:  # the headline WGS configuration as a FILE: measured by tools/run_config_file.py under gpurun (generating its 54 GB input takes 3 minutes;
            # `--workload config3` measures it live), copied here with its source named
            c3name = next(f for f in ("r4_config3_file.json", "r3_config3_file.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
            c3 = json.load(open(os.path.join(ROOT, "profiles", c3name)))
            out["e2e_config3"] = {"source": "profiles/%s (tools/run_config_file.py --config 3; not re-measured in this run)" % c3name, "reads": c3["reads"],
                                  "bam_GB": c3["bam_bytes"] / 1e9, "wall_s": c3["program_wall_s"], "reads_per_s": c3["reads_per_s"], "compressed_GB_per_s": c3["compressed_GB_per_s"],
                                  "reader": c3["reader"], "which": c3["which"], "host_reader_reads_per_s": c3["host_reader"]["reads_per_s"],
                                  "prefix_matches_oracle": c3.get("prefix_matches_oracle")}
        except Exception:
            pass
        if e2e_big:
            out.update(e2e_big)
            px, c3 = e2e_big.get("e2e_shard_proxy"), out.get("e2e_config3")
            if px and c3 and "wall_s" in px:  # the 1 -> 8 projection: config 3 on one card over one worker's eighth of it
                px["projected_speedup_8"] = c3["wall_s"] / px["wall_s"]
                px["projection"] = ("config 3 on ONE card (%.2f s, %s) / this leg's wall time: what `bamqualcheck --gpus 8` gains when its slowest worker takes as long as this run "
                                    "(not in it: the hook's hand-over and one RCCL reduce of ~3 MB, a middle shard's search for its first record, eight workers sharing the host's cores and page cache)"
                                    % (c3["wall_s"], c3["source"]))
        if e2e_sharded is not None:
            out["e2e_sharded"] = e2e_sharded
        if e2e is not None:
            cv = out.get("cpu_baseline", {}).get("value")
            e2e["speedup_vs_cpu_port"] = (e2e["reads_per_s"] / cv) if cv else None
            out["e2e"] = e2e
        print(json.dumps(out), flush=True)
    else:
        db.free()
        agg.close()
    if world > 1:
        dist.destroy_process_group()


def launch_ranks(args):
    """`python bench.py --gpus N` started like `--gpus 1`: this process has touched no GPU (torch is not even imported), so it starts the N ranks as
    CHILD processes through torch.distributed.run (never an exec), relays their output — rank 0's JSON line — and leaves with their exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd).returncode)


def tempfile_dir():
    import tempfile
    return tempfile.gettempdir()


def e2e_sharded_leg(args, world, one_gpu_rehearsal):
    """`bin/bamqualcheck --gpus N` on a BAM file of 25 M reads per GPU (BGZF level 1, 2.2 GB per GPU: every worker's byte range is large
    enough for the reader on the card), three runs, beside one single-GPU run of the same file: wall times, speed-up, byte identity of the outputs,
    the size-independent properties of the output.  Child processes with a time limit each."""
    import filecmp
    import shutil
    import signal
    import statistics
    import subprocess
    import tempfile
    from bamqc_amd import hostio
    from tests import bamqc_text
    exe = os.path.join(ROOT, "bin", "bamqualcheck")
    reads = int(os.environ.get("BQC_BENCH_SHARDED_READS", str(min(25_000_000 * world, 200_000_000))))  # (2.2 GB of BAM per GPU: a worker's start-up of ~0.3 s is not all there is to see)
    tmp = tempfile.mkdtemp(prefix="bqc_e2e_sharded_")
    try:
        names, lens = ["chr1", "chr2", "chr3", "chr4"], [25_000_000] * 4
        bam, fa = os.path.join(tmp, "s.bam"), os.path.join(tmp, "s.fa")
        t0 = time.time()
        hostio.synth_stream(bam, fa, 1002, reads, names, lens, read_len=args.read_len, level=1)
        t_write = time.time() - t0
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                                                                  "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
        env["BQC_TIMING"] = "1"
        if one_gpu_rehearsal:
            env["BQC_GPUS_SHARE_DEVICE"] = "1"  # (the workers share the one card; sums through pipes)

        def run(extra, out):
            t1 = time.perf_counter()
            p = subprocess.Popen([exe] + extra + ["-r", fa, "-o", out, "-c", ",".join(names), bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env,
                                 start_new_session=True)
            try:
                _, err = p.communicate(timeout=240)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                p.communicate()
                raise RuntimeError("timed out: bamqualcheck " + " ".join(extra))
            dt = time.perf_counter() - t1
            if p.returncode != 0:
                raise RuntimeError("bamqualcheck %s failed (%d): %s" % (" ".join(extra), p.returncode, err[-600:]))
            return dt, err

        multi = []
        for k in range(3):
            time.sleep(1.0)
            dt, err = run(["--gpus", str(world)], os.path.join(tmp, "m%d.bamqc" % k))
            multi.append(dt)
        loops = [float(x) for x in __import__("re").findall(r"record loop ([0-9.]+) s", err)]
        time.sleep(1.0)
        one, _ = run([], os.path.join(tmp, "one.bamqc"))
        lanes = bamqc_text.parse(os.path.join(tmp, "m0.bamqc"))
        bamqc_text.check_invariants(lanes["L1"], n_records=reads, read_len=args.read_len)
        wall = statistics.median(multi)
        return {"what": "bin/bamqualcheck --gpus %d (one worker per GPU on its byte range of the file, one RCCL reduce from C++) on a BAM file of %d reads (BGZF level 1, %.1f GB), "
                        "default options; median of three runs; beside ONE run of the single-GPU program on the same file%s" %
                        (world, reads, os.path.getsize(bam) / 1e9, "; REHEARSAL: all workers on one card" if one_gpu_rehearsal else ""),
                "gpus": world, "reads": reads, "wall_s": wall, "runs_s": multi, "reads_per_s": reads / wall, "record_loops_s_last_run": loops,
                "one_gpu_wall_s": one, "one_gpu_reads_per_s": reads / one, "speedup_vs_one_gpu": one / wall,
                "identical_to_one_gpu": all(filecmp.cmp(os.path.join(tmp, "one.bamqc"), os.path.join(tmp, "m%d.bamqc" % k), shallow=False) for k in range(3)),
                "invariants": "ok", "write_input_s": t_write}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def config3_line(args):
    """--workload config3: BASELINE.json config 3 at full size as a FILE through the program, live (tools/run_config_file.py)."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory(prefix="bqc_c3line_") as tmp:
        out = os.path.join(tmp, "c3.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_config_file.py"), "--config", "3", "--runs", str(max(1, min(args.steps, 5))), "--out", out] +
                           (["--reads", str(args.reads)] if args.reads else []), capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        c3 = json.load(open(out))
    line = {"metric": "BAM records/sec (and GB/s vs HBM roofline), 150 bp PE, 1/2/4/8 MI355X", "value": c3["reads_per_s"], "unit": "reads/s",
            "value_kind": "the PROGRAM: BAM file (BGZF level 1) -> bin/bamqualcheck, default options -> .bamqc; wall time of the whole process, median of the runs",
            "n_gpus": 1, "steps": len(c3["runs"]), "warmup": 0, "ms_per_step": c3["program_wall_s"] * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u32/u64 integer", "data": "synthetic",
            "config": {"workload": "config 3: %d reads x 150 bp PE, 24 GRCh38-length contigs, as a %.1f GB BAM file" % (c3["reads"], c3["bam_bytes"] / 1e9)},
            "e2e_config3": c3}
    print(json.dumps(line), flush=True)


def extra_timings(refs):
    """Device times that the headline step leaves out, measured the same way (resident batch, HIP events): the program's
    default options (k-mer sketch k32 q17) on config-2-shaped reads, and the long-read kernels on config-5-shaped reads."""
    from bamqc_amd import Aggregator, synth
    out = {}
    n = 4_000_000
    lens = [25_000_000] * 4
    cols = synth.batch(1002, n, lens, refs)
    agg = Aggregator(n_refs=4, klist=(32,), qlist=(17,))
    for i, r in enumerate(refs):
        agg.set_reference(i, r)
    db = agg.upload(cols)
    agg.set_timing(True)
    kt = {}
    for it in range(8):
        agg.process(db)
        if it >= 3:
            for k, v in agg.last_timing().items():
                kt.setdefault(k, []).append(v)
    agg.sync()
    per10 = {k: float(np.mean(v)) * 1e7 / n for k, v in kt.items()}
    out["default_options_with_sketch"] = {"reads": n, "ms_per_10M_reads": per10, "total_ms_per_10M_reads": sum(per10.values()),
                                          "what": "config-2-shaped reads, -k 32 -q 17 (k_sketch is VALU bound: two 128-bit rolling hashes per base)"}
    db.free()
    agg.close()
    del cols
    n, L = 50_000, 10_000
    ref5 = [synth.reference(1005, 0, 250_000_000)]
    cols = synth.batch(1005, n, [250_000_000], ref5, read_len=L, isize=30_000, long_reads=True)
    agg = Aggregator(n_refs=1, isize=30_000, max_read_len=16_384, hist_cap=16_384)
    agg.set_reference(0, ref5[0])
    db = agg.upload(cols)
    ab = db.algorithmic_bytes
    agg.set_timing(True)
    kt = {}
    for it in range(8):
        agg.process(db)
        if it >= 3:
            for k, v in agg.last_timing().items():
                kt.setdefault(k, []).append(v)
    agg.sync()
    km = {k: float(np.mean(v)) for k, v in kt.items()}
    out["long_reads_config5_shape"] = {"reads": n, "read_len": L, "kernel_ms": km, "algorithmic_bytes": ab,
                                       "k_long_GBps": ab / km["k_long"] / 1e6, "k_long_frac_of_8TBps": ab / km["k_long"] / 1e6 / 8000.0,
                                       "reads_per_s": n / (sum(km.values()) * 1e-3)}
    db.free()
    agg.close()
    return out


GRCH38_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]


def program_runs(exe, cli, reads, n_runs, tmp, tag):
    """bin/bamqualcheck `cli` n_runs times, a second apart: (median run, all runs); outputs tmp/<tag><k>.bamqc must be identical"""
    import filecmp
    import re
    import subprocess
    runs = []
    for k in range(n_runs):
        out = os.path.join(tmp, "%s%d.bamqc" % (tag, k))
        time.sleep(1.0)
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-o", out] + cli, capture_output=True, text=True, env=dict(os.environ, BQC_TIMING="1", BQC_GB_TIMING="1", BQC_T0="%.6f" % time.monotonic()))
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr[-2000:]
        m = re.search(r"record loop ([0-9.]+) s", r.stderr)
        ls = re.search(r"record loop starts: ([0-9.]+) s after launch", r.stderr)
        runs.append({"wall_s": dt, "reads_per_s": reads / dt, "record_loop_s": float(m.group(1)) if m else None, "loop_start_s": float(ls.group(1)) if ls else None,
                     "reader": "gpu" if "records decoded on the GPU" in r.stderr else "host", "timing": [ln for ln in r.stderr.splitlines() if ln.startswith(("[timing]", "[gpu reader] open"))]})
    assert all(filecmp.cmp(os.path.join(tmp, tag + "0.bamqc"), os.path.join(tmp, "%s%d.bamqc" % (tag, k)), shallow=False) for k in range(1, n_runs))
    return sorted(runs, key=lambda x: x["wall_s"])[n_runs // 2], runs


def large_legs(args):
    """Two BAM files through bin/bamqualcheck with default options, both driver-timed (inside this benchmark's own run):
    e2e_large — 100 M reads over the 24 GRCh38-length contigs (~10 GB at BGZF level 1): a start-up-amortised BAM throughput, median of three runs, the whole
    output through the invariants of tests/bamqc_text.py, the first million reads of the same plan byte for byte against the oracle;
    e2e_shard_proxy — what ONE of the eight workers of `bamqualcheck --gpus 8` has to do on config 3 (618 M reads): 77.25 M reads over the contigs the third
    eighth of the genome touches (chr4, chr5, chr6), as a file of their own through the single-GPU program: wall time and the time the record loop starts."""
    import shutil
    import tempfile
    from bamqc_amd import hostio
    from bamqc_amd.synth import GRCH38
    from tests import bamqc_text
    from tests.test_gpu_stream import prefix_parity
    exe = os.path.join(ROOT, "bin", "bamqualcheck")
    res = {}
    tmp = tempfile.mkdtemp(prefix="bqc_e2e_large_")
    try:
        bam, fa = os.path.join(tmp, "l.bam"), os.path.join(tmp, "l.fa")
        t0 = time.time()
        hostio.synth_stream(bam, fa, 1003, args.large_reads, GRCH38_NAMES, GRCH38, level=1)
        t_write = time.time() - t0
        size = os.path.getsize(bam)
        med, runs = program_runs(exe, ["-r", fa, bam], args.large_reads, 3, tmp, "l")
        lanes = bamqc_text.parse(os.path.join(tmp, "l0.bamqc"))
        bamqc_text.check_invariants(lanes["L1"], n_records=args.large_reads, read_len=150)
        npre = min(args.large_reads, 1_000_000)
        prefix_parity(tmp, 1003, args.large_reads, npre, GRCH38_NAMES, GRCH38, ["-c", "chr1"], dict(chroms="chr1"))
        res["e2e_large"] = {"what": "BAM file of %d reads x 150 bp PE over 24 GRCh38-length contigs (BGZF level 1, %.1f GB) -> bin/bamqualcheck, default options -> .bamqc; wall time of "
                                    "the whole process, median of 3 runs, timed live in this run" % (args.large_reads, size / 1e9),
                            "reads": args.large_reads, "bam_GB": size / 1e9, "wall_s": med["wall_s"], "reads_per_s": med["reads_per_s"], "compressed_GB_per_s": size / med["wall_s"] / 1e9,
                            "frac_of_hbm": med["reads_per_s"] * 277.0 / 8e12, "frac_of_hbm_what": "reads/s x 277 algorithmic bytes per read / 8 TB/s",
                            "record_loop_s": med["record_loop_s"], "loop_start_s": med["loop_start_s"], "reader": med["reader"], "runs": runs, "invariants": "ok",
                            "prefix_matches_oracle": {"reads": npre, "identical": True}, "write_input_s": t_write}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    tmp = tempfile.mkdtemp(prefix="bqc_e2e_proxy_")
    try:
        which = [3, 4, 5]  # chr4, chr5, chr6: where the third of eight equal parts of the genome (772 - 1158 Mb) lies
        names, lens = [GRCH38_NAMES[k] for k in which], [GRCH38[k] for k in which]
        reads = 618_000_000 // 8
        bam, fa = os.path.join(tmp, "p.bam"), os.path.join(tmp, "p.fa")
        t0 = time.time()
        hostio.synth_stream(bam, fa, 1013, reads, names, lens, level=1)
        t_write = time.time() - t0
        size = os.path.getsize(bam)
        med, runs = program_runs(exe, ["-r", fa, "-c", ",".join(names), bam], reads, 3, tmp, "p")
        lanes = bamqc_text.parse(os.path.join(tmp, "p0.bamqc"))
        bamqc_text.check_invariants(lanes["L1"], n_records=reads, read_len=150)
        res["e2e_shard_proxy"] = {"what": "one worker's eighth of config 3 as a file of its own: %d reads x 150 bp PE over %s (BGZF level 1, %.1f GB) -> bin/bamqualcheck, default sketch; "
                                          "median of 3 runs, timed live in this run" % (reads, "+".join(names), size / 1e9),
                                  "reads": reads, "bam_GB": size / 1e9, "wall_s": med["wall_s"], "loop_start_s": med["loop_start_s"], "record_loop_s": med["record_loop_s"],
                                  "reads_per_s": med["reads_per_s"], "reader": med["reader"], "runs": runs, "invariants": "ok", "write_input_s": t_write}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res


def cpu_baseline(cols, refs, args, agg, db):
    """Oracle (single-thread CPU restatement of the reference; the reference itself cannot be compiled
    here: SeqAn 1.4.2 is absent) timed on a bounded prefix of the same workload, and used to check the
    GPU result for that prefix bit-exactly."""
    from bamqc_amd import Aggregator, _abi
    from tests import synth as tsynth
    from tests.oracle_lib import Oracle
    n = min(args.cpu_sample, len(cols["flag"]))
    sub = tsynth.slice_batch(cols, 0, n)
    o = Oracle(n_refs=4, max_read_len=max(1024, args.read_len))
    for i, r in enumerate(refs):
        o.reference(i, r)
    t0 = time.perf_counter()
    rc = o.process(sub)
    dt = time.perf_counter() - t0
    assert rc == 0
    want = o.finalize()
    g = Aggregator(n_refs=4, max_read_len=max(1024, args.read_len), device=int(os.environ.get("LOCAL_RANK", "0")))
    for i, r in enumerate(refs):
        g.set_reference(i, r)
    g.submit(sub)
    diffs = _abi.diff_counts(want, g.finalize())
    g.close()
    return {"value": n / dt, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": "first %d reads of the same batch, oracle/liboracle.so (C restatement), 1 thread" % n,
            "gpu_matches_oracle_on_sample": not diffs, "diffs": diffs[:3]}


def e2e_leg(args, refs, cpu_kernel_value):
    """What a user gets: the same workload as a BAM FILE through the program bin/bamqualcheck (default options: k-mer sketch
    -k 32 -q 17 included) to the `.bamqc` text — BGZF inflate, record decode, pre-pass, H2D, kernels, finalisation, process
    start and HIP initialisation all inside the wall time.  Checked: the whole output through the size-independent properties
    of tests/bamqc_text.py; a prefix of the same plan byte for byte against the oracle's `.bamqc`."""
    import filecmp
    import re
    import shutil
    import subprocess
    import tempfile
    from bamqc_amd import hostio
    from bamqc_amd.host_info import cpu_limit
    from tests import bamqc_text
    from tests.cli_oracle import oracle_bamqualcheck
    exe = os.path.join(ROOT, "bin", "bamqualcheck")
    tmp = tempfile.mkdtemp(prefix="bqc_e2e_")
    try:
        names, lens = ["chr1", "chr2", "chr3", "chr4"], [25_000_000] * 4
        bam, fa = os.path.join(tmp, "c2.bam"), os.path.join(tmp, "c2.fa")
        t0 = time.time()
        hostio.synth_stream(bam, fa, 1002, args.reads, names, lens, read_len=args.read_len, level=1)
        t_write = time.time() - t0
        runs = []
        for extra in ([], [], [], [], [], ["--no-sketch"]):
            out = os.path.join(tmp, "o%d.bamqc" % len(runs))
            time.sleep(1.0)  # (the worker process of the run before hands its memory and its context back after its front end has left)
            t0 = time.perf_counter()
            r = subprocess.run([exe, "-r", fa, "-o", out, "-c", ",".join(names)] + extra + [bam], capture_output=True, text=True,
                               env=dict(os.environ, BQC_TIMING="1", BQC_T0="%.6f" % time.monotonic()))
            dt = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr[-2000:]
            m = re.search(r"record loop ([0-9.]+) s", r.stderr)
            runs.append({"args": extra, "wall_s": dt, "reads_per_s": args.reads / dt, "record_loop_s": float(m.group(1)) if m else None,
                         "reader": "gpu (inflate, CRC, record walk and column decode on the card)" if "records decoded on the GPU" in r.stderr else "host",
                         "timing": [ln for ln in r.stderr.splitlines() if ln.startswith("[timing]")]})
        assert all(filecmp.cmp(os.path.join(tmp, "o0.bamqc"), os.path.join(tmp, "o%d.bamqc" % k), shallow=False) for k in (1, 2))
        lanes = bamqc_text.parse(os.path.join(tmp, "o1.bamqc"))
        bamqc_text.check_invariants(lanes["L1"], n_records=args.reads, read_len=args.read_len)
        # prefix of the same plan: program vs oracle, byte for byte; the oracle run is also the CPU end-to-end baseline
        npre = min(args.e2e_prefix, args.reads)
        pbam, pfa = os.path.join(tmp, "p.bam"), fa
        hostio.write_bam(pbam, hostio.synth_slice(1002, args.reads, 0, npre, lens, refs=refs, read_len=args.read_len), names, lens)
        got, want = os.path.join(tmp, "p_gpu.bamqc"), os.path.join(tmp, "p_cpu.bamqc")
        r = subprocess.run([exe, "-r", pfa, "-o", got, "-c", ",".join(names), pbam], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        t0 = time.perf_counter()
        assert oracle_bamqualcheck(pbam, pfa, want, chroms=",".join(names)) == 0
        t_cpu = time.perf_counter() - t0
        timed = sorted(runs[:5], key=lambda x: x["wall_s"])
        best = timed[len(timed) // 2]  # the MEDIAN of five runs (each started a second after the one before has left)
        # the reader's inflate launches on the same file, each waited for (BQC_GB_TIMING=2): the largest kernels of a whole-file run
        r = subprocess.run([exe, "-r", fa, "-o", os.path.join(tmp, "t.bamqc"), "-c", ",".join(names), bam], capture_output=True, text=True,
                           env=dict(os.environ, BQC_GB_TIMING="2"))
        inflate_runs = []
        for m in re.finditer(r"run of (\d+) blocks, ([0-9.]+) MB -> ([0-9.]+) MB: inflate \+ crc ([0-9.]+) ms", r.stderr):
            nb, mb_in, mb_out, ms = int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4))
            inflate_runs.append({"blocks": nb, "compressed_MB": mb_in, "inflated_MB": mb_out, "ms_inflate_resolve_crc": ms,
                                 "GB_per_s_in_plus_out": (mb_in + mb_out) / ms, "frac_of_8TBps": (mb_in + mb_out) / ms / 8000.0})
        # Ten files back to back, as a QC pipeline runs them: every front end starts the moment the one before has left, i.e. while
        # that run's worker process is still handing its page-locked buffers and its GPU context back — the price of the early exit
        # of tools/bamqualcheck.cpp is inside this wall time.
        n_b2b = 10
        time.sleep(1.0)
        t0 = time.perf_counter()
        for k in range(n_b2b):
            r = subprocess.run([exe, "-r", fa, "-o", os.path.join(tmp, "b%d.bamqc" % k), "-c", ",".join(names), bam], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
        t_b2b = time.perf_counter() - t0
        assert all(filecmp.cmp(os.path.join(tmp, "o0.bamqc"), os.path.join(tmp, "b%d.bamqc" % k), shallow=False) for k in range(n_b2b))
        return {"frac_of_hbm": best["reads_per_s"] * 277.0 / 8e12,
                "what": "BAM file (BGZF level 1, %.0f MB) -> bin/bamqualcheck (default options, sketch k32 q17) -> .bamqc; wall time of the whole process "
                        "(median of 5 runs; the front end leaves when the output is complete, its worker's teardown of 0.2-0.5 s is outside a single "
                        "run's wall time and inside back_to_back's)" % (os.path.getsize(bam) / 1e6),
                "reads": args.reads, "wall_s": best["wall_s"], "reads_per_s": best["reads_per_s"], "wall_s_min": timed[0]["wall_s"], "wall_s_max": timed[-1]["wall_s"],
                "back_to_back": {"files": n_b2b, "wall_s": t_b2b, "s_per_file": t_b2b / n_b2b, "reads_per_s": n_b2b * args.reads / t_b2b},
                "host_cpus": cpu_limit(),
                "inflate_launches": {"what": "k_inflate_wave + k_inflate_resolve + k_gi_crc per run of BGZF blocks of this file (each launch waited for; not bound by HBM: DESIGN.md 4.5)",
                                     "runs": inflate_runs},
                "runs": runs, "write_input_s": t_write,
                "matches_oracle": filecmp.cmp(got, want, shallow=False), "oracle_prefix_reads": npre,
                "cpu_port_e2e_reads_per_s": npre / t_cpu,
                "cpu_port_e2e_what": "same program flow on the CPU oracle (own reader on all host threads + oracle incl. sketch, 1 thread) on the prefix",
                "speedup_vs_cpu_port_e2e": best["reads_per_s"] / (npre / t_cpu),
                "speedup_vs_cpu_port": (best["reads_per_s"] / cpu_kernel_value) if cpu_kernel_value else None}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
