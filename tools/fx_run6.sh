bash tools/sweep.sh --libs "orig v0 vpf vtv vrp vall" --workloads "cggi kms2_n1024 cggi_l2" -- --arith exact
