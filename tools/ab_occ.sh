cp mktfhe_amd/lib/libmktfhe_hip.so /tmp/orig.so
for rep in 1 2; do
for cfg in "base 0" "base 21" "occ3 21"; do set -- $cfg
 if [ $1 != base ]; then cp mktfhe_amd/lib/libmktfhe_hip_$1.so mktfhe_amd/lib/libmktfhe_hip.so; else cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so; fi
 for w in kms2_n1024 cggi; do
 MKT_ROT_VARIANT=$2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload $w 2>&1 | grep '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$1 variant $2', d['config']['params'], 'gates/s %.0f'%d['value'], 'rot ms %.2f'%d['kernels_ms_per_step']['blindrotate'])"
 done
done; done
cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so
