export MKT_LIB_PATH=$PWD/mktfhe_amd/lib/libmktfhe_hip_dbg.so
MKT_FX_DEBUG=1 MKT_EXACT_IMPL=1 python bench.py --workload kms2party --arith exact --no-cpu-baseline --no-secondary --no-roofline --steps 1 --warmup 0 2>&1 | grep fx_blind | sort | uniq -c
MKT_FX_DEBUG=1 MKT_EXACT_IMPL=1 python bench.py --workload kms2_n1024 --arith exact --no-cpu-baseline --no-secondary --no-roofline --steps 1 --warmup 0 2>&1 | grep fx_blind | sort | uniq -c
