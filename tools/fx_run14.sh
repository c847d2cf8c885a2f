python -m pytest tests/test_gpu_fx.py -x -q 2>&1 | tail -3
for s in "KMS2party" "KMS2party N=1024" "KMS2party_N1024_l2" "KMS2party_N1024_l2 N=2048 logB_gsw=14" "KMS2party_N1024_l2 N=512" "CGGIparam" "CGGI_N1024_l2" "CGGIparam N=2048" "CGGIparam N=4096" "CGGIparam N=512"; do python tools/fx_shape_time.py $s; done 2>&1 | grep -v amdgpu.ids
