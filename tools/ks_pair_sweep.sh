# usage: [WAVES="4 8"] [BLOCKS="512 1024 ..."] bash tools/ks_pair_sweep.sh  -- digit-pair key switch: time against the target workgroup count
# and the waves sharing one table (8 = sixteen ciphertexts per wave)
for w in "kms2_n1024 1024" "cggi 1024" "lmss 1024" "lmss 16384" "kms2partyblock 1024"; do set -- $w
 for kw in ${WAVES:-1 2 4 8}; do for kb in ${BLOCKS:-512 1024 2048 4096 8192}; do
  MKT_KS_WAVES=$kw MKT_KS_BLOCKS=$kb python3 bench.py --steps 3 --warmup 1 --workload $1 --batch $2 --no-roofline --no-cpu-baseline --no-secondary 2>/dev/null | grep -a '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config']['params'], 'batch $2 waves $kw blocks $kb', 'ks ms %.3f'%d['kernels_ms_per_step']['keyswitch'], 'ok', d['decrypt_ok'])
"; done; done; done
