bash tools/sweep.sh --libs "base r0p0" --workloads "kms2_n1024 cggi cggi_l2 kms2party" -- --arith exact
