mkdir -p gpurun_out
cp mktfhe_amd/lib/libmktfhe_hip.so /tmp/orig.so
for ab in 0 1 2 4 8 3 15; do
 if [ $ab != 0 ]; then cp mktfhe_amd/lib/libmktfhe_hip_ab$ab.so mktfhe_amd/lib/libmktfhe_hip.so; fi
 for v in 21 22; do
 for w in kms2_n1024 kms2party; do
  MKT_ROT_VARIANT=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload $w 2>&1 | grep '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    print('ablate $ab variant $v', d['config']['params'], 'rot ms %.2f'%d['kernels_ms_per_step']['blindrotate'])
"
 done; done
done
cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so
