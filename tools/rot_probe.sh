# usage: bash tools/rot_probe.sh  -- phase times of the rotation kernel (builds with -DMKT_ROT_PROBE=1: libmktfhe_hip_rotprobe.so), plain and block
cp mktfhe_amd/lib/libmktfhe_hip.so /tmp/orig.so; cp mktfhe_amd/lib/libmktfhe_hip_rotprobe.so mktfhe_amd/lib/libmktfhe_hip.so
for w in ${WORKLOADS:-cggi lmss}; do for v in 21 22; do
  MKT_ROT_VARIANT=$v python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary --workload $w > /tmp/o.txt 2>&1
  echo "$w variant $v: $(grep -a 'rot probe' /tmp/o.txt | tail -1)"
done; done
cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so
