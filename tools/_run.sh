bash tools/profile_round.sh r03 kms2_n1024 2>&1 | tail -2
TAG=r03 bash tools/all_workloads.sh 2>&1 | tail -40
