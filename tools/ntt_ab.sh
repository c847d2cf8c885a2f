# usage: LIBS="old base" bash tools/ntt_ab.sh  -- transform legs of alternative builds on one device
cp mktfhe_amd/lib/libmktfhe_hip.so /tmp/orig.so
for sfx in ${LIBS:-base}; do
 if [ "$sfx" != base ]; then cp mktfhe_amd/lib/libmktfhe_hip_$sfx.so mktfhe_amd/lib/libmktfhe_hip.so; else cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so; fi
 echo "== $sfx"
 timeout 300 bash tools/ntt_legs.sh 2>&1 | grep -a -i -E "ntt|EXACT"
done
cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so
