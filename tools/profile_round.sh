# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag> [workload]
# writes the bench line, rocprofv3 kernel-trace stats and separate PMC passes of the SAME command under
# gpurun_out/<tag>/ ; tools/collect_profiles.py <tag> [workload] copies the summaries into profiles/
TAG=${1:-r02}; WL=${2:-kms2_n1024}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
ARGS="--workload $WL --no-cpu-baseline --no-secondary"
python3 $R/bench.py --steps 20 --warmup 5 --workload $WL > $O/bench_$WL.json 2> $O/bench_$WL.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$WL -- python3 $R/bench.py --steps 5 --warmup 1 $ARGS > $O/trace_$WL.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${WL}_fetch -- python3 $R/bench.py --steps 2 --warmup 0 $ARGS > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${WL}_write -- python3 $R/bench.py --steps 2 --warmup 0 $ARGS > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/pmc_${WL}_sq1 -- python3 $R/bench.py --steps 2 --warmup 0 $ARGS --no-roofline > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_${WL}_sq2 -- python3 $R/bench.py --steps 2 --warmup 0 $ARGS --no-roofline > /dev/null 2>&1
grep -h '"metric"' $O/bench_$WL.json | cut -c1-400
