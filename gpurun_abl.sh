timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_generic_path.py tests/test_gpu_cli.py -m gpu -q -x 2>&1 | tail -3
for p in 15 0 1 2 4; do
  BQC_SHORT_PARTS=$p timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu --reads 4000000 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('parts=$p', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['kernel_ms'].items()})"
done
