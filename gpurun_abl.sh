timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_generic_path.py tests/test_gpu_cli.py tests/test_gpu_sketch.py -m gpu -q -x 2>&1 | tail -2
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['kernel_ms'].items()})"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_t
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_32B_sum --output-format csv -d gpurun_out/pmc_t -- python bench.py --steps 2 --warmup 0 --no-cpu > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections
for f in glob.glob("gpurun_out/pmc_t/*/*counter_collection.csv"):
    agg=collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].split("(")[0]
        if k.startswith("k_short"): agg[(k,row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k,v in sorted(agg.items()): print(k,len(v),sum(v)/len(v))
PY
