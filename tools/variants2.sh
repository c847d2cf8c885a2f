mkdir -p gpurun_out
cp mktfhe_amd/lib/libmktfhe_hip.so /tmp/orig.so
for mw in 3 4; do
 cp mktfhe_amd/lib/libmktfhe_hip_mw$mw.so mktfhe_amd/lib/libmktfhe_hip.so
 for v in 21 22; do
 for w in kms2_n1024 kms2party cggi; do
  MKT_ROT_VARIANT=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload $w 2>&1 | grep '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    print('minw $mw variant $v', d['config']['params'], 'gates/s %.0f'%d['value'], 'rot ms %.2f'%d['kernels_ms_per_step']['blindrotate'], 'rot GF %.0f'%d.get('blindrotate',{}).get('f64_gflops',0), 'ok', d['decrypt_ok'])
"
 done; done
done
cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so
