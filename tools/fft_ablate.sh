cp mktfhe_amd/lib/libmktfhe_hip.so /tmp/orig.so
for ab in 0 2 4 6 16; do
 if [ $ab != 0 ]; then cp mktfhe_amd/lib/libmktfhe_hip_ab$ab.so mktfhe_amd/lib/libmktfhe_hip.so; fi
 echo "ablate $ab"; bash tools/fft_bench.sh
done
cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so
