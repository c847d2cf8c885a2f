for G in 16 32; do for BL in 512 1024 2048 4096; do
 for w in kms2party kms2_n1024; do
  MKT_KS_G=$G MKT_KS_BLOCKS=$BL python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload $w 2>&1 | grep '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    print('G $G blocks $BL', d['config']['params'], 'ks ms %.2f'%d['kernels_ms_per_step']['keyswitch'])
"
 done; done; done
