# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# writes rocprofv3 kernel-trace stats + HBM PMC passes for the default bench workload under gpurun_out/<tag>/
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/$TAG; cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 5 --warmup 1 > $R/gpurun_out/$TAG/bench_kms2_n1024.json 2> /dev/null
python3 $R/bench.py --steps 3 --warmup 1 --workload kms2party > $R/gpurun_out/$TAG/bench_kms2party.json 2> /dev/null
python3 $R/bench.py --steps 5 --warmup 1 --workload cggi > $R/gpurun_out/$TAG/bench_cggi.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$TAG/trace_bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/$TAG/pmc_sq1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/$TAG/pmc_sq2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1
grep -h '"metric"' $R/gpurun_out/$TAG/bench_*.json | cut -c1-200
