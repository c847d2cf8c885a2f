# usage (GPU box, repo root): bash tools/pmc_pass.sh <tag> <workload> <name> [full] [-- extra bench.py args]
# rocprofv3 passes of ONE bench.py command (--steps 2 --warmup 0, no CPU baseline / secondary legs), each counter group in its own
# run as MI355X_MICROARCH.md prescribes (--pmc only with --kernel-trace): FETCH_SIZE, WRITE_SIZE, clock + hit rates; `full` adds
# the kernel-trace --stats run and the two SQ groups.  Output: gpurun_out/<tag>/{trace,pmc}_<name>_*; tools/collect_profiles.py
# <tag> <name> turns them into profiles/<tag>_bench_<name>_{kernel_stats.csv,pmc.txt}.
TAG=$1; WL=$2; NAME=$3; shift 3
FULL=0; if [ "$1" = full ]; then FULL=1; shift; fi
if [ "$1" = -- ]; then shift; fi
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
ARGS="--workload $WL --no-cpu-baseline --no-secondary ${NOROOF---no-roofline} $*"     # NOROOF= (set, empty) keeps the transform legs in the profiled command
run() { timeout 400 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/pmc_${NAME}_$1 -- python3 $R/bench.py --steps 2 --warmup 0 $ARGS > /dev/null 2>&1; }
# which library and device these passes measured (tools/collect_profiles.py copies it into the summary's header; bench.py quotes the summary
# only when the library it has loaded carries the same build id)
python3 -c "
import sys, hashlib; sys.path.insert(0, '$R')
import mktfhe_amd as mk, torch
print('build_id', mk.build_id()); print('so_sha256', hashlib.sha256(open(mk.LIB_PATH, 'rb').read()).hexdigest()); print('device', torch.cuda.get_device_name(0))
" > $O/pmc_${NAME}_meta.txt 2>/dev/null
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run clk "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"
if [ $FULL = 1 ]; then
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$NAME -- python3 $R/bench.py --steps 20 --warmup 5 $ARGS > $O/trace_$NAME.log 2>&1     # the driver's step counts: the average launch duration here is the one the bench line's avg_launch_ms must agree with
  run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
  run sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_WAIT_INST_LDS"
fi
ls $O | grep "_${NAME}_" | tr '\n' ' '; echo
