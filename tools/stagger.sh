for st in 0 1 2 4 8 16 64; do for w in kms2_n1024 kms2party cggi; do
MKT_ROT_STAGGER=$st python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload $w 2>/dev/null | grep '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('stagger $st', d['config']['params'], 'rot ms %.2f'%d['kernels_ms_per_step']['blindrotate'], 'gates/s %.0f'%d['value'])"
done; done
