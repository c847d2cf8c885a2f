# usage: bash tools/ks_blocks_sweep.sh  -- key-switch time against the launch's target workgroup count (MKT_KS_BLOCKS) at full batches
for w in "kms2_n1024 1024" "cggi 1024" "lmss_k2 8192" "kms2party 1024" "ccs2party 1024"; do set -- $w
 for kb in 1024 2048 4096 8192 16384; do
  MKT_KS_BLOCKS=$kb python3 bench.py --steps 3 --warmup 1 --workload $1 --batch $2 --no-roofline --no-cpu-baseline --no-secondary 2>/dev/null | grep -a '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config']['params'], 'batch $2 KS_BLOCKS $kb', 'ks ms %.3f'%d['kernels_ms_per_step']['keyswitch'], 'gates/s %.0f'%d['value'], 'ok', d['decrypt_ok'])
"; done; done
