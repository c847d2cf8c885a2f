timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "exact_mode" 2>&1 | tail -4
for w in kms2_n1024 kms2party kms2partyblock; do python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --workload $w --arith exact 2>&1 | grep '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$w exact', 'gates/s %.0f'%d['value'], 'rot %.2f'%d['kernels_ms_per_step']['blindrotate'], 'errs', d['decrypt_errors'])
"; done
