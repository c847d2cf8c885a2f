for l in e0 f1 f2; do
export MKT_LIB_PATH=$PWD/mktfhe_amd/lib/libmktfhe_hip_$l.so
echo == $l
python tools/fx_shape_time.py KMS2party_N1024_l2 --impl=1
python tools/fx_shape_time.py CGGIparam --impl=1
python tools/fx_shape_time.py CGGI_N1024_l2 --impl=1
python tools/fx_shape_time.py CGGIparam N=2048 --impl=1
done 2>&1 | grep -v amdgpu.ids
