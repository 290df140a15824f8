#!/bin/bash
# Collects the round-4 rocprofv3 evidence on the GPU box (run through gpurun from the repo root; rocprofv3 always gets the program
# itself behind `--`, --pmc passes carry no trace options):
#   kt             kernel-trace stats of bench.py's step (config 2, device-resident batch: k_prep_* + k_short + k_cov)
#   kt_e2e_gpu     kernel-trace stats of the whole program on the benchmark's 10 M-read BAM file, records inflated and decoded on the card
#   pmc_*          FETCH_SIZE / WRITE_SIZE / TCC in separate passes over bench.py, with the 1-GiB calibration kernel; SQ counter passes
#   pmc_e2e_*      SQ counters and FETCH_SIZE / WRITE_SIZE of the reader's kernels (k_inflate*, k_gi_crc, k_gb_*) over the program
#   kt_long / kt_sketch   the long-read kernels and the default options with the k-mer sketch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r4; rm -rf $O; mkdir -p $O
IN=/tmp/bqc_prof_in
python tools/make_e2e_input.py $IN 10000000 1 > $O/input_bytes.txt
B="python bench.py --steps 10 --warmup 2 --no-cpu --no-e2e --no-extra"
B3="python bench.py --steps 3 --warmup 0 --no-cpu --no-e2e --no-extra"
P="bin/bamqualcheck -r $IN/c2.fa -o $IN/o.bamqc -c chr1,chr2,chr3,chr4 $IN/c2.bam"
export BQC_NO_FORK=1 BQC_FAST_EXIT=0
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/bench_under_rocprof.json 2> $O/kt.err
BQC_TIMING=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_e2e_gpu -- $P > $O/e2e_gpu.txt 2>&1
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
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_e2e_a -- $P > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_e2e_b -- $P > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_e2e_fetch -- $P > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_e2e_write -- $P > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, json
O = "gpurun_out/prof_r4"
out, e2e = {}, {}
for name in ("pmc_fetch", "pmc_write", "pmc_tcc", "cal_fetch", "cal_dram", "sq_a", "sq_b", "sq_c", "pmc_e2e_a", "pmc_e2e_b", "pmc_e2e_fetch", "pmc_e2e_write"):
    for f in glob.glob(O + "/" + name + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            agg[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in agg.items():
            if name.startswith("pmc_e2e"):
                if "inflate" in k or "k_g" in k or "k_an" in k:
                    e2e["%s:%s" % (k, c)] = {"dispatches": len(v), "avg": sum(v) / len(v), "sum": sum(v)}
            elif k.startswith("k_") or "k_" in k:
                out["%s:%s:%s" % (name, k, c)] = {"dispatches": len(v), "avg": sum(v) / len(v)}
json.dump(out, open(O + "/pmc_summary.json", "w"), indent=1)
json.dump(e2e, open(O + "/pmc_e2e_summary.json", "w"), indent=1)
print(len(out), "+", len(e2e), "counter averages")
PY
for d in kt kt_e2e_gpu kt_long kt_sketch; do cp $O/$d/*/*kernel_stats.csv $O/${d}_kernel_stats.csv 2>/dev/null; done
for d in kt kt_e2e_gpu kt_long kt_sketch pmc_fetch pmc_write pmc_tcc cal_dram cal_fetch sq_a sq_b sq_c pmc_e2e_a pmc_e2e_b pmc_e2e_fetch pmc_e2e_write; do rm -rf $O/$d; done
rm -rf $IN
head -12 $O/kt_kernel_stats.csv; head -16 $O/kt_e2e_gpu_kernel_stats.csv; grep timing $O/e2e_gpu.txt | tail -5
