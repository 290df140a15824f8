cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_a -- python bench.py --steps 2 --warmup 0 --no-cpu --reads 4000000 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_b -- python bench.py --steps 2 --warmup 0 --no-cpu --reads 4000000 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_ADD_F16 SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_c -- python bench.py --steps 2 --warmup 0 --no-cpu --reads 4000000 > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections
for d in ("pmc_a","pmc_b","pmc_c"):
    for f in glob.glob("gpurun_out/%s/*/*counter_collection.csv"%d):
        agg=collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "k_short" in row["Kernel_Name"]: agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k,v in agg.items(): print(d,k,len(v),sum(v)/len(v))
PY
