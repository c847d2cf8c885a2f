# usage: LIBS="ccsprobe ccsabl1" bash tools/ccs_probe.sh -- phase times of the CCS kernel (builds with -DMKT_CCS_PROBE=1) + its launch time
cp mktfhe_amd/lib/libmktfhe_hip.so /tmp/orig.so
for sfx in ${LIBS:-ccsprobe}; do
 cp mktfhe_amd/lib/libmktfhe_hip_$sfx.so mktfhe_amd/lib/libmktfhe_hip.so
 for w in ${WORKLOADS:-ccs2party}; do
  python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary ${ARGS:-} --workload $w > /tmp/o.txt 2>&1
  echo "$sfx $w: $(grep -a 'ccs probe' /tmp/o.txt | tail -1)"
  grep -a '"metric"' /tmp/o.txt | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   rot ms %.2f'%d['kernels_ms_per_step']['blindrotate'], 'gates/s %.0f'%d['value'])"
 done
done
cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so
