for l in u4 u2 u1; do
export MKT_LIB_PATH=$PWD/mktfhe_amd/lib/libmktfhe_hip_$l.so
echo == $l
python tools/fx_shape_time.py KMS2party --impl=1
python tools/fx_shape_time.py KMS2party N=1024 --impl=1
python tools/fx_shape_time.py KMS2party N=512 --impl=1
done 2>&1 | grep -v amdgpu.ids
