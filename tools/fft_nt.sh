cp mktfhe_amd/lib/libmktfhe_hip.so /tmp/orig.so
for sfx in ${LIBS:-nt3}; do
 if [ $sfx != base ]; then cp mktfhe_amd/lib/libmktfhe_hip_$sfx.so mktfhe_amd/lib/libmktfhe_hip.so; else cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so; fi
 TAG=$sfx NBS=${NBS:--} python tools/fft_sweep.py ${GRIDS:-0 20480 65536} 2>&1 | grep GB/s
done
cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so
