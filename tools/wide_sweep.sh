# usage: bash tools/wide_sweep.sh  -- plain (MKT_ROT_WIDE=1) against latency kernel (MKT_ROT_WIDE=2) around one compute unit's worth of rotations
for w in ${WORKLOADS:-kms2_n1024 cggi}; do for b in ${BATCHES:-96 128 170 256 300 340 400 512}; do for m in 1 2; do
  MKT_ROT_WIDE=$m python3 bench.py --steps 5 --warmup 1 --workload $w --batch $b --no-roofline --no-cpu-baseline --no-secondary 2>/dev/null | grep -a '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$w batch $b', 'WIDE=$m', 'rot ms %.3f'%d['kernels_ms_per_step']['blindrotate'], 'gates/s %.0f'%d['value'], d['decrypt_ok'])
"; done; done; done
