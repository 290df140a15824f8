#!/bin/bash
# Collects the round-1 rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of bench.py, FETCH_SIZE / WRITE_SIZE in separate --pmc passes, and the same two
#   counters for k_calib_read4 (1 GiB, 4-byte-per-lane loads) to calibrate FETCH_SIZE for this access width.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r1d; rm -rf $O; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --steps 10 --warmup 2 --no-cpu > $O/bench_under_rocprof.json 2> $O/kt.err
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 3 --warmup 0 --no-cpu > /dev/null 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --steps 3 --warmup 0 --no-cpu > /dev/null 2>&1
timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -- python bench.py --steps 3 --warmup 0 --no-cpu > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum --output-format csv -d $O/cal_dram -- python -c "from bamqc_amd import _lib; _lib.load().bqc_calib_read4(1<<30, 3)" > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -- python -c "from bamqc_amd import _lib; _lib.load().bqc_calib_read4(1<<30, 3)" > /dev/null 2>&1
# SQ counters of k_short (instruction mix, VALU / LDS utilisation), three passes of at most 8 counters
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/sq_a -- python bench.py --steps 3 --warmup 0 --no-cpu > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq_b -- python bench.py --steps 3 --warmup 0 --no-cpu > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_IFETCH --output-format csv -d $O/sq_c -- python bench.py --steps 3 --warmup 0 --no-cpu > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, json
O = "gpurun_out/prof_r1d"
out = {}
for name in ("pmc_fetch", "pmc_write", "pmc_tcc", "cal_fetch", "cal_dram", "sq_a", "sq_b", "sq_c"):
    for f in glob.glob(O + "/" + name + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            agg[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in agg.items():
            if k.startswith("k_") or "k_" in k:
                out["%s:%s:%s" % (name, k, c)] = {"dispatches": len(v), "avg": sum(v) / len(v)}
json.dump(out, open(O + "/pmc_summary.json", "w"), indent=1)
for k, v in out.items():
    print(k, v)
PY
cat $O/kt/*/*kernel_stats.csv | head -12
