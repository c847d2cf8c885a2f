for rep in 1 2; do
for st in 0 8 16 32 64; do for sp in 0 1024; do
 MKT_ROT_STAGGER=$st MKT_ROT_SPLIT=$sp python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stagger $st split $sp rot %.3f value %.0f'%(d['kernels_ms_per_step']['blindrotate'], d['value']))"
done; done; done
