KERN=ccs_blindrotate PMC_GROUPS="sq1 sq2 clk" bash tools/pmc.sh r06ccs2 -- bench.py --workload ccs2party --steps 2 --warmup 0 --no-cpu-baseline --no-secondary --no-roofline
KERN=ccs_blindrotate PMC_GROUPS="sq1 sq2 clk" bash tools/pmc.sh r06ccs8 -- bench.py --workload ccs8party --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-roofline
KERN=blindrotate PMC_GROUPS="sq1 sq2 clk" bash tools/pmc.sh r06blk -- bench.py --workload kms2partyblock --steps 2 --warmup 0 --no-cpu-baseline --no-secondary --no-roofline
KERN=blindrotate PMC_GROUPS="sq1 sq2 clk" bash tools/pmc.sh r06k1 -- bench.py --workload kms2_n1024 --steps 2 --warmup 0 --no-cpu-baseline --no-secondary --no-roofline
