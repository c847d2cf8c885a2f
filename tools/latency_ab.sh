# usage: LIBS="old base" bash tools/latency_ab.sh  -- small-batch gate latency of alternative builds on one device (the latency kernel: MKT_ROT_WIDE default rules)
cp mktfhe_amd/lib/libmktfhe_hip.so /tmp/orig.so
for sfx in ${LIBS:-base}; do
 if [ "$sfx" != base ]; then cp mktfhe_amd/lib/libmktfhe_hip_$sfx.so mktfhe_amd/lib/libmktfhe_hip.so; else cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so; fi
 for w in ${WORKLOADS:-cggi cggi_l2 kms2_n1024 kms2party}; do for b in ${BATCHES:-1 16 64}; do
  python3 bench.py --steps 10 --warmup 2 --workload $w --batch $b --no-roofline --no-cpu-baseline --no-secondary 2>/dev/null | grep -a '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$sfx', d['config']['params'], 'batch $b', 'ms/batch %.3f'%d['ms_per_step'], 'rot %.3f'%d['kernels_ms_per_step']['blindrotate'], 'ok', d['decrypt_ok'], d.get('oracle_bitexact'))
"; done; done
done
cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so
