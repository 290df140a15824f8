for p in 15 0 1 2 4; do
  BQC_SHORT_PARTS=$p timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu --reads 4000000 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('parts=$p', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['kernel_ms'].items()})"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_a
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_a -- python bench.py --steps 2 --warmup 0 --no-cpu --reads 4000000 > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections
for f in glob.glob("gpurun_out/pmc_a/*/*counter_collection.csv"):
    agg=collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "k_short" in row["Kernel_Name"]: agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in agg.items(): print(k,len(v),sum(v)/len(v))
PY
