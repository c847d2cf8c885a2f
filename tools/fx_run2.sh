bash tools/sweep.sh --libs "base mo0 mo1 rt0" --workloads kms2_n1024 -- --arith exact
bash tools/sweep.sh MKT_ROT_STAGGER=0,8,32,64 --workloads kms2_n1024 -- --arith exact
bash tools/sweep.sh MKT_ROT_MAP=0 --workloads kms2_n1024 -- --arith exact
bash tools/sweep.sh MKT_ROT_STAGGER=0,16 --workloads "kms2party cggi cggi_l2" -- --arith exact
KERN=fx_blindrotate PMC_GROUPS="sq1 sq2 clk tcp" bash tools/pmc.sh fxpmc -- bench.py --workload kms2_n1024 --arith exact --steps 2 --warmup 0 --no-cpu-baseline --no-secondary --no-roofline
