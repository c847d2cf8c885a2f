# parity + perf of an alternative build: ab_parity.sh <suffix>
cp mktfhe_amd/lib/libmktfhe_hip.so /tmp/orig.so
cp mktfhe_amd/lib/libmktfhe_hip_$1.so mktfhe_amd/lib/libmktfhe_hip.so
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
cp /tmp/orig.so mktfhe_amd/lib/libmktfhe_hip.so
