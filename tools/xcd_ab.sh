cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3u
for x in 0 1; do
  BQC_SHORT_XCD=$x python bench.py --steps 20 --warmup 3 --no-cpu --no-e2e --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('xcd=$x', d['ms_per_step'], d['roofline']['kernel_ms'])"
  BQC_SHORT_XCD=$x timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r3u/f$x -- python bench.py --steps 3 --warmup 0 --no-cpu --no-e2e --no-extra > /dev/null 2>&1
  python - <<PY
import csv, glob
v=[float(r["Counter_Value"]) for f in glob.glob("gpurun_out/r3u/f$x/*/*counter_collection.csv") for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("k_short")]
print("xcd=$x FETCH_SIZE KB avg", sum(v)/max(1,len(v)), "x2 =", 2*sum(v)/max(1,len(v))*1024/1e9, "GB")
PY
  rm -rf gpurun_out/r3u/f$x
done
BQC_SHORT_XCD=1 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
