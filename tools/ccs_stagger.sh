for st in 0 5 10 20 40 100 400; do
  echo "stagger $st: $(MKT_CCS_STAGGER=$st python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --workload ccs2party 2>&1 | grep -a '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('rot ms %.2f'%d['kernels_ms_per_step']['blindrotate'], 'gates/s %.0f'%d['value'])")"
done
