# usage: bash tools/ks_pair_ab.sh  -- key switch with one table read per digit (MKT_KS_PAIR=0) against one per digit pair (1), same box
for w in "kms2_n1024 1024" "cggi 1024" "lmss 1024" "lmss 16384" "kms2partyblock 1024" "kms2party 1024" "ccs2party 1024"; do set -- $w
 for kp in 0 1; do
  MKT_KS_PAIR=$kp python3 bench.py --steps 5 --warmup 2 --workload $1 --batch $2 --no-roofline --no-cpu-baseline --no-secondary 2>/dev/null | grep -a '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config']['params'], 'batch $2 KS_PAIR $kp', 'ks ms %.3f'%d['kernels_ms_per_step']['keyswitch'], 'gates/s %.0f'%d['value'], 'ok', d['decrypt_ok'], d.get('oracle_bitexact'))
"; done; done
