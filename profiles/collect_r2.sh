#!/bin/bash
# Collects the round-2 rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of bench.py (config 2, device-resident batch: k_prep_* + k_short + k_cov per step), of the long-read
#   kernels (tools/k_long_time.py, config-5-shaped reads) and of the default options with the k-mer sketch (tools/sketch_time.py);
#   FETCH_SIZE / WRITE_SIZE in separate --pmc passes with the 1-GiB calibration kernel; SQ counter passes (<= 8 counters each).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r2; rm -rf $O; mkdir -p $O
B="python bench.py --steps 10 --warmup 2 --no-cpu --no-e2e --no-extra"
B3="python bench.py --steps 3 --warmup 0 --no-cpu --no-e2e --no-extra"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/bench_under_rocprof.json 2> $O/kt.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_long -- python tools/k_long_time.py > $O/k_long_time.txt 2> $O/kt_long.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_sketch -- python tools/sketch_time.py > $O/sketch_time.txt 2> $O/kt_sketch.err
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B3 > /dev/null 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B3 > /dev/null 2>&1
timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -- $B3 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum --output-format csv -d $O/cal_dram -- python -c "from bamqc_amd import _lib; _lib.load().bqc_calib_read4(1<<30, 3)" > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -- python -c "from bamqc_amd import _lib; _lib.load().bqc_calib_read4(1<<30, 3)" > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/sq_a -- $B3 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq_b -- $B3 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_IFETCH --output-format csv -d $O/sq_c -- $B3 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_IDX_ACTIVE --output-format csv -d $O/long_a -- python tools/k_long_time.py > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $O/long_b -- python tools/k_long_time.py > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/long_fetch -- python tools/k_long_time.py > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sketch_a -- python tools/sketch_time.py > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $O/sketch_b -- python tools/sketch_time.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, json
O = "gpurun_out/prof_r2"
out = {}
for name in ("pmc_fetch", "pmc_write", "pmc_tcc", "cal_fetch", "cal_dram", "sq_a", "sq_b", "sq_c", "long_a", "long_b", "long_fetch", "sketch_a", "sketch_b"):
    for f in glob.glob(O + "/" + name + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            agg[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in agg.items():
            if k.startswith("k_") or "k_" in k:
                out["%s:%s:%s" % (name, k, c)] = {"dispatches": len(v), "avg": sum(v) / len(v)}
json.dump(out, open(O + "/pmc_summary.json", "w"), indent=1)
print(len(out), "counter averages")
PY
for d in kt kt_long kt_sketch; do cp $O/$d/*/*kernel_stats.csv $O/${d}_kernel_stats.csv 2>/dev/null; done
head -12 $O/kt_kernel_stats.csv; head -6 $O/kt_long_kernel_stats.csv; head -6 $O/kt_sketch_kernel_stats.csv
