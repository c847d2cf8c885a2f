bash tools/sweep.sh --libs "base orig base orig" --workloads "cggi kms2_n1024" -- --arith exact
