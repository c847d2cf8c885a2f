for g in ${GRIDS:-0}; do
MKT_FFT_GRID=$g python bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload kms2_n1024 2>&1 | grep '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('grid $g N1024 fft GB/s %.0f frac %.3f ms %.3f'%(r['achieved'], r['frac'], r['avg_launch_ms']))
"
MKT_FFT_GRID=$g python bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload kms2party 2>&1 | grep '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('grid $g N2048 fft GB/s %.0f frac %.3f ms %.3f'%(r['achieved'], r['frac'], r['avg_launch_ms']))
"
done
