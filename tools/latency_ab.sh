# usage: LIBS="base x" bash tools/latency_ab.sh  -- single-gate / small-batch step time of alternative builds (latency variant automatic)
cp mktfhe_amd/lib/libmktfhe_hip.so /tmp/orig.so
for sfx in ${LIBS:-base}; do
 if [ "$sfx" != base ]; then cp mktfhe_amd/lib/libmktfhe_hip_$sfx.so mktfhe_amd/lib/libmktfhe_hip.so; else cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so; fi
 for w in ${WORKLOADS:-cggi cggi_l2 kms2_n1024}; do for b in ${BATCHES:-1 64}; do
  python3 bench.py --workload $w --batch $b --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$sfx $w batch $b', 'ms/step %.3f'%d['ms_per_step'], 'rot %.3f'%d['kernels_ms_per_step']['blindrotate'], 'ks %.3f'%d['kernels_ms_per_step']['keyswitch'], 'errs', d['decrypt_errors'], 'bitexact', d.get('oracle_bitexact'))
"
 done; done
done
cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so
