"""BASELINE.json configs 3 and 5 at full size as FILES on one MI355X (--config).  Config 3: ~618 M synthetic 150 bp PE reads (30x of 3.088 Gb) over chr1..chr22,
chrX, chrY with GRCh38 lengths, written once at BGZF level 1 (~51 GB), then `bin/bamqualcheck` (default -c, default k-mer sketch)
timed on it: reads/s, compressed GB/s in, the reader named, device memory in use.  Checked by the size-independent properties of
tests/bamqc_text.py and by the first million reads of the same plan against the oracle, byte for byte.
With --gpus-shared N also: the same file through `bamqualcheck --gpus N` with N workers sharing the one card (each inflates and
decodes its byte range there; what a node with N cards does, minus the N cards).
usage: python tools/run_config3_file.py [--reads N] [--level L] [--runs K] [--dir D] [--out J] [--keep] [--gpus-shared N]"""
import argparse
import filecmp
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bamqc_amd import hostio  # noqa: E402
from bamqc_amd.synth import GRCH38  # noqa: E402
from tests import bamqc_text  # noqa: E402
from tests.test_gpu_stream import NAMES24, prefix_parity  # noqa: E402

EXE = os.path.join(ROOT, "bin", "bamqualcheck")


def timed_run(args, env_extra=None):
    env = dict(os.environ, BQC_TIMING="1", BQC_T0="%.6f" % time.monotonic())
    env.update(env_extra or {})
    t0 = time.perf_counter()
    r = subprocess.run([EXE] + args, capture_output=True, text=True, env=env)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    return dt, r.stderr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3, choices=(3, 5), help="3: 30x WGS 150 bp PE over 24 GRCh38 contigs; 5: 5 M reads x 10 kb, indel / soft-clip heavy, -i 30000")
    ap.add_argument("--reads", type=int, default=None)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--gpus-shared", type=int, default=0)
    ap.add_argument("--proxy", action="store_true", help="config 3: also one worker's eighth of it as a file of its own (77.25 M reads over chr4-chr6) through the program, on the same box: "
                                                         "shard_proxy and projected_speedup_8 in the result")
    ap.add_argument("--variant", action="append", default=[], help="ENV=VALUE[,ENV=VALUE]: the timed runs once more with these variables set (reported under `variants`)")
    a = ap.parse_args()
    if a.config == 3:
        names, lens, seed, cli, synth_kw, read_len = NAMES24, GRCH38, 1003, [], {}, 150
        what = "3: 30x WGS-scale synthetic, 24 GRCh38-length contigs, default -c / -k 32 -q 17, as a FILE"
        pre_n, pre_cli, pre_okw, pre_kw = 1_000_000, ["-c", "chr1"], dict(chroms="chr1"), {}
        a.reads = a.reads or 618_000_000
    else:
        names, lens, seed, read_len = ["chr1"], [250_000_000], 1005, 10_000
        cli = ["-c", "chr1", "-i", "30000", "--no-sketch", "--max-read-len", "16384"]
        synth_kw = dict(read_len=10_000, isize=30_000, long_reads=True)
        what = "5: long-read stress, 10 kb reads with 20-60 CIGAR operations, 1 x 250 Mb contig, -i 30000 --no-sketch, as a FILE"
        pre_n, pre_cli, pre_kw = 20_000, cli, synth_kw
        pre_okw = dict(chroms="chr1", isize=30000, klist=(), qlist=(), max_read_len=16384, hist_cap=65536)
        a.reads = a.reads or 5_000_000
    tmp = tempfile.mkdtemp(prefix="bqc_c%df_" % a.config, dir=a.dir)
    res = {"config": what, "reads": a.reads, "bgzf_level": a.level}
    try:
        bam, fa = os.path.join(tmp, "c3.bam"), os.path.join(tmp, "c3.fa")
        t0 = time.time()
        hostio.synth_stream(bam, fa, seed, a.reads, names, lens, level=a.level, **synth_kw)
        res["write_input_s"] = time.time() - t0
        size = os.path.getsize(bam)
        res["bam_bytes"] = size
        print("input written: %.1f GB in %.0f s" % (size / 1e9, res["write_input_s"]), flush=True)
        runs = []
        for k in range(a.runs):
            out = os.path.join(tmp, "o%d.bamqc" % k)
            time.sleep(2.0)
            dt, err = timed_run(["-r", fa, "-o", out] + cli + [bam])
            m = re.search(r"record loop ([0-9.]+) s", err)
            runs.append({"wall_s": dt, "reads_per_s": a.reads / dt, "compressed_GB_per_s": size / dt / 1e9, "record_loop_s": float(m.group(1)) if m else None,
                         "reader": "gpu (inflate, CRC, record walk and column decode on the card)" if "records decoded on the GPU" in err else "host",
                         "timing": [ln for ln in err.splitlines() if ln.startswith(("[timing]", "[gpu reader]"))]})
            print("run %d: %.2f s = %.1f M reads/s, %.2f GB/s of compressed input" % (k, dt, a.reads / dt / 1e6, size / dt / 1e9), flush=True)
        assert all(filecmp.cmp(os.path.join(tmp, "o0.bamqc"), os.path.join(tmp, "o%d.bamqc" % k), shallow=False) for k in range(1, a.runs))
        lanes = bamqc_text.parse(os.path.join(tmp, "o0.bamqc"))
        info = bamqc_text.check_invariants(lanes["L1"], n_records=a.reads, read_len=read_len)
        cov = lanes["L1"]["genome_coverage_histogram"]
        med = sorted(runs, key=lambda x: x["wall_s"])[len(runs) // 2]
        res.update({"program_wall_s": med["wall_s"], "reads_per_s": med["reads_per_s"], "compressed_GB_per_s": med["compressed_GB_per_s"], "which": "median of %d runs" % a.runs,
                    "reader": med["reader"], "runs": runs, "primary_reads": info["primary"], "triplets": info["triplets"], "eightmers": info["eightmers"],
                    "coverage_positions": int(cov.sum()), "mean_depth_main": float((cov * range(101)).sum() / max(1, cov.sum())), "invariants": "ok"})
        res["variants"] = []
        for spec in a.variant:
            env = dict(kv.split("=") for kv in spec.split(","))
            vr = []
            for k in range(a.runs):
                time.sleep(2.0)
                dt, err = timed_run(["-r", fa, "-o", os.path.join(tmp, "v.bamqc")] + cli + [bam], env)
                vr.append({"wall_s": dt, "timing": [ln for ln in err.splitlines() if ln.startswith("[timing]")]})
                assert filecmp.cmp(os.path.join(tmp, "o0.bamqc"), os.path.join(tmp, "v.bamqc"), shallow=False)
                print("%s run %d: %.2f s" % (spec, k, dt), flush=True)
            res["variants"].append({"env": env, "runs": vr, "median_wall_s": sorted(x["wall_s"] for x in vr)[len(vr) // 2]})
        # the host reader on the same file, once (what the GPU reader replaced)
        dt, err = timed_run(["-r", fa, "-o", os.path.join(tmp, "h.bamqc")] + cli + [bam], {"BQC_GPU_DECODE": "0"})
        res["host_reader"] = {"wall_s": dt, "reads_per_s": a.reads / dt, "identical_output": filecmp.cmp(os.path.join(tmp, "o0.bamqc"), os.path.join(tmp, "h.bamqc"), shallow=False)}
        print("host reader: %.2f s = %.1f M reads/s" % (dt, a.reads / dt / 1e6), flush=True)
        if a.gpus_shared > 1:
            shared = []
            for n in sorted({2, a.gpus_shared}):
                out = os.path.join(tmp, "g%d.bamqc" % n)
                time.sleep(2.0)
                dt, err = timed_run(["--gpus", str(n), "-r", fa, "-o", out] + cli + [bam], {"BQC_GPUS_SHARE_DEVICE": "1"})
                loops = [float(x) for x in re.findall(r"record loop ([0-9.]+) s", err)]
                shared.append({"workers": n, "wall_s": dt, "reads_per_s": a.reads / dt, "record_loops_s": loops, "identical_output": filecmp.cmp(os.path.join(tmp, "o0.bamqc"), out, shallow=False),
                               "what": "%d workers SHARING one card (BQC_GPUS_SHARE_DEVICE=1): each reads, inflates and decodes its byte range; sums through pipes" % n})
                print("--gpus %d on one card: %.2f s, record loops %s" % (n, dt, loops), flush=True)
            res["workers_sharing_one_card"] = shared
        if a.proxy and a.config == 3:
            os.remove(bam)  # (room for the second file)
            which = [3, 4, 5]
            pn, pl = [names[k] for k in which], [lens[k] for k in which]
            preads = 618_000_000 // 8
            pbam, pfa = os.path.join(tmp, "p.bam"), os.path.join(tmp, "p.fa")
            hostio.synth_stream(pbam, pfa, 1013, preads, pn, pl, level=a.level)
            pr = []
            for k in range(a.runs):
                time.sleep(2.0)
                dt, err = timed_run(["-r", pfa, "-o", os.path.join(tmp, "p%d.bamqc" % k), "-c", ",".join(pn), pbam])
                m = re.search(r"record loop starts: ([0-9.]+) s", err)
                pr.append({"wall_s": dt, "loop_start_s": float(m.group(1)) if m else None, "timing": [ln for ln in err.splitlines() if ln.startswith("[timing]")]})
                print("proxy run %d: %.3f s" % (k, dt), flush=True)
            pmed = sorted(x["wall_s"] for x in pr)[len(pr) // 2]
            bamqc_text.check_invariants(bamqc_text.parse(os.path.join(tmp, "p0.bamqc"))["L1"], n_records=preads, read_len=150)
            res["shard_proxy"] = {"what": "one worker's eighth of config 3 as a file of its own (%d reads over %s, %.1f GB) through the single-GPU program on the SAME box, median of %d runs"
                                          % (preads, "+".join(pn), os.path.getsize(pbam) / 1e9, a.runs), "reads": preads, "wall_s": pmed, "runs": pr}
            res["projected_speedup_8"] = res["program_wall_s"] / pmed
            res["projected_speedup_8_what"] = "program_wall_s / shard_proxy.wall_s: what `bamqualcheck --gpus 8` gains over one card when its slowest worker takes as long as the proxy run (not in it: the hook's hand-over, one RCCL reduce of ~3 MB, eight workers sharing the host)"
            print("projected 1 -> 8: %.2f x" % res["projected_speedup_8"], flush=True)
        # the first million reads of the same plan against the oracle, byte for byte
        prefix_parity(tmp, seed, a.reads, min(a.reads, pre_n), names, lens, pre_cli, pre_okw, **pre_kw)
        res["prefix_matches_oracle"] = {"reads": min(a.reads, pre_n), "identical": True}
        print(json.dumps({k: v for k, v in res.items() if k != "runs"}, indent=1), flush=True)
        if a.out:
            json.dump(res, open(a.out, "w"), indent=1)
    finally:
        if not a.keep:
            shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
