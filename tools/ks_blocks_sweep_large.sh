for w in "kms2_n1024 8192" "cggi 8192" "lmss 16384"; do set -- $w
 for kb in 1024 4096 8192 16384 32768; do
  MKT_KS_BLOCKS=$kb python3 bench.py --steps 2 --warmup 1 --workload $1 --batch $2 --no-roofline --no-cpu-baseline --no-secondary 2>/dev/null | grep -a '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config']['params'], 'batch $2 KS_BLOCKS $kb', 'ks ms %.3f'%d['kernels_ms_per_step']['keyswitch'], 'gates/s %.0f'%d['value'], 'ok', d['decrypt_ok'])
"; done; done
