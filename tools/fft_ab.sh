cp mktfhe_amd/lib/libmktfhe_hip.so /tmp/orig.so
for sfx in ${LIBS:-base}; do
 if [ $sfx != base ]; then cp mktfhe_amd/lib/libmktfhe_hip_$sfx.so mktfhe_amd/lib/libmktfhe_hip.so; fi
 echo "== $sfx"; GRIDS="0 2560 2304 2048 5120" bash tools/fft_bench.sh
done
cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so
