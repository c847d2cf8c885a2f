KERN=fx_blindrotate PMC_GROUPS="sq1 sq2 clk fetch" bash tools/pmc.sh fxpmc3 -- bench.py --workload kms2party --arith exact --steps 2 --warmup 0 --no-cpu-baseline --no-secondary --no-roofline
KERN=fx_blindrotate PMC_GROUPS="sq1 clk" bash tools/pmc.sh fxpmc4 -- bench.py --workload cggi --arith exact --steps 2 --warmup 0 --no-cpu-baseline --no-secondary --no-roofline
