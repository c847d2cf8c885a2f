python -m pytest tests/test_gpu_fx.py -x -q 2>&1 | tail -3
bash tools/sweep.sh --libs "base pb pc pd pe" --workloads "kms2_n1024 cggi kms2party" -- --arith exact
