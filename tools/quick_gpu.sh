# quick GPU check: parity subset + the three bench workloads (no CPU baseline)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for w in kms2_n1024 kms2party cggi; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w 2>&1 | grep '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d.get('roofline',{})
    print(d['config']['params'], 'gates/s %.0f'%d['value'], 'ms', {k:round(v,2) for k,v in d['kernels_ms_per_step'].items()}, 'rot GF %.0f'%d.get('blindrotate',{}).get('f64_gflops',0), 'fft GB/s %.0f'%r.get('achieved',0), 'ok', d['decrypt_ok'])
"
done
