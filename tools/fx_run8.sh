bash tools/sweep.sh --libs "x0 xtr xilp xmc x0" --workloads "cggi kms2_n1024 cggi_l2" -- --arith exact
