set -x
python -m pytest tests/test_gpu_fx.py -x -q 2>&1 | tail -25
for w in kms2_n1024 kms2party cggi; do
  for impl in 0 1; do
    MKT_EXACT_IMPL=$impl python bench.py --workload $w --arith exact --no-cpu-baseline --no-secondary --no-roofline --steps 5 --warmup 2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('RESULT', '$w', 'impl=$impl', round(d['value']), 'gates/s', d.get('kernels_ms_per_step'), d['roofline'].get('kernel'), 'dec_err', d.get('decrypt_errors'))
"
  done
done
