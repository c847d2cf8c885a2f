# GPU test suite + bench + profile in one gpurun call: bash tools/quick_gpu.sh <tag>
TAG=${1:-r02a}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/$TAG
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$TAG/pytest_gpu.log
tail -5 gpurun_out/$TAG/pytest_gpu.log
bash tools/profile_round.sh $TAG kms2_n1024
