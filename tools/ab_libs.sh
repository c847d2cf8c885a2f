# usage: ab_libs.sh <suffix list>   -- A/B of alternative builds mktfhe_amd/lib/libmktfhe_hip_<suffix>.so ("base" = default)
mkdir -p gpurun_out
cp mktfhe_amd/lib/libmktfhe_hip.so /tmp/orig.so
for sfx in "$@"; do
 if [ "$sfx" != base ]; then cp mktfhe_amd/lib/libmktfhe_hip_$sfx.so mktfhe_amd/lib/libmktfhe_hip.so; else cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so; fi
 for v in ${VARIANTS:-21 22}; do
 for w in ${WORKLOADS:-kms2_n1024 kms2party cggi}; do
  MKT_ROT_VARIANT=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload $w 2>&1 | grep '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    print('$sfx variant $v', d['config']['params'], 'gates/s %.0f'%d['value'], 'rot ms %.2f'%d['kernels_ms_per_step']['blindrotate'], 'ks ms %.2f'%d['kernels_ms_per_step']['keyswitch'], 'ok', d['decrypt_ok'])
"
 done; done
done
cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so
