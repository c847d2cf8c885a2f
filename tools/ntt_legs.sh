# usage: bash tools/ntt_legs.sh  -- the transform legs of the bench line (Float64 and EXACT) and the exact gate rate
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | grep -a '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    for r in d['roofline_transform']: print(r.get('kernel'), r.get('N'), r.get('direction'), 'TB/s %.3f'%(r['achieved']/1e3), 'frac %.3f'%r['frac'])
"
python3 tools/exact_rate.py 2>&1 | grep EXACT
