# usage: LIBS="base x y" WORKLOADS="..." bash tools/ab_bench.sh  -- bench.py A/B of alternative builds libmktfhe_hip_<sfx>.so
cp mktfhe_amd/lib/libmktfhe_hip.so /tmp/orig.so
for sfx in ${LIBS:-base}; do
 if [ "$sfx" != base ]; then cp mktfhe_amd/lib/libmktfhe_hip_$sfx.so mktfhe_amd/lib/libmktfhe_hip.so; else cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so; fi
 for w in ${WORKLOADS:-kms2_n1024 kms2party cggi}; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload $w ${ARGS:-} 2>&1 | grep '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    print('$sfx', d['config']['params'], 'gates/s %.0f'%d['value'], 'rot ms %.2f'%d['kernels_ms_per_step']['blindrotate'], 'ks ms %.3f'%d['kernels_ms_per_step']['keyswitch'], 'ok', d['decrypt_ok'])
"
 done
done
cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so
