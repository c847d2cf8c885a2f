python tools/fx_shape_time.py KMS2party
python tools/fx_shape_time.py KMS2party l_gsw=2 logB_gsw=14
python tools/fx_shape_time.py KMS2party N=1024
python tools/fx_shape_time.py KMS2party_N1024_l2
python tools/fx_shape_time.py KMS2party_N1024_l2 N=2048 logB_gsw=14
python tools/fx_shape_time.py CGGIparam N=2048
python tools/fx_shape_time.py CGGIparam N=2048 l_gsw=2 logB_gsw=10
