#!/bin/bash
# Round 2, second collection (run through gpurun from the repo root): kernel-trace stats of bench.py's step after the last change to
# the pre-pass (config 2), and of the whole program on the benchmark's 10 M-read BAM file with the records inflated and decoded
# on the GPU (BQC_FAST_EXIT=0: the program leaves through exit() so that the profiler gets to write its files; k_inflate, k_gi_crc, k_gb_walk / _decode / _copy next to the aggregation kernels); the same with the host reader.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r2b; rm -rf $O; mkdir -p $O
IN=/tmp/bqc_prof_in
python tools/make_e2e_input.py $IN 10000000 1 > $O/input_bytes.txt
B="python bench.py --steps 10 --warmup 2 --no-cpu --no-e2e --no-extra"
[ "$1" = "e2e" ] || timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/bench_under_rocprof.json 2> $O/kt.err
BQC_NO_FORK=1 BQC_FAST_EXIT=0 BQC_TIMING=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_e2e_gpu -- bin/bamqualcheck -r $IN/c2.fa -o $IN/o_gpu.bamqc -c chr1,chr2,chr3,chr4 $IN/c2.bam > $O/e2e_gpu.txt 2>&1
BQC_NO_FORK=1 BQC_FAST_EXIT=0 BQC_TIMING=1 BQC_GPU_DECODE=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_e2e_host -- bin/bamqualcheck -r $IN/c2.fa -o $IN/o_host.bamqc -c chr1,chr2,chr3,chr4 $IN/c2.bam > $O/e2e_host.txt 2>&1
cmp $IN/o_gpu.bamqc $IN/o_host.bamqc && echo "outputs identical" > $O/e2e_cmp.txt
# SQ counters of the reader's kernels (one pass; the program once more)
BQC_NO_FORK=1 BQC_FAST_EXIT=0 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_e2e_a -- bin/bamqualcheck -r $IN/c2.fa -o $IN/o_pmc.bamqc -c chr1,chr2,chr3,chr4 $IN/c2.bam > /dev/null 2>&1
BQC_NO_FORK=1 BQC_FAST_EXIT=0 timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_e2e_b -- bin/bamqualcheck -r $IN/c2.fa -o $IN/o_pmc.bamqc -c chr1,chr2,chr3,chr4 $IN/c2.bam > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, json
O = "gpurun_out/prof_r2b"
out = {}
for name in ("pmc_e2e_a", "pmc_e2e_b"):
    for f in glob.glob(O + "/" + name + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            agg[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in agg.items():
            if "inflate" in k or "k_g" in k:
                out["%s:%s" % (k, c)] = {"dispatches": len(v), "avg": sum(v) / len(v), "sum": sum(v)}
json.dump(out, open(O + "/pmc_e2e_summary.json", "w"), indent=1)
print(len(out), "counter averages")
PY
rm -rf $O/pmc_e2e_a $O/pmc_e2e_b
for d in kt kt_e2e_gpu kt_e2e_host; do cp $O/$d/*/*kernel_stats.csv $O/${d}_kernel_stats.csv 2>/dev/null; done
rm -rf $O/kt $O/kt_e2e_gpu $O/kt_e2e_host $IN
head -14 $O/kt_kernel_stats.csv 2>/dev/null; head -24 $O/kt_e2e_gpu_kernel_stats.csv; grep timing $O/e2e_gpu.txt | tail -4; cat $O/e2e_cmp.txt
