bash tools/sweep.sh MKT_ROT_SPLIT=0,-1,512,2048 --workloads "kms2_n1024 cggi kms2party" -- --arith exact
KERN=fx_blindrotate PMC_GROUPS="sq1 clk fetch" bash tools/pmc.sh fxpmc2 -- bench.py --workload kms2_n1024 --arith exact --steps 2 --warmup 0 --no-cpu-baseline --no-secondary --no-roofline
