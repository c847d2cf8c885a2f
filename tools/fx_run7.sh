python -m pytest tests/test_gpu_fx.py -x -q 2>&1 | tail -3
bash tools/sweep.sh --libs "orig w0 wpf wpf1 wrp" --workloads "cggi kms2_n1024 cggi_l2" -- --arith exact
