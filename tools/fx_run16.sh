bash tools/sweep.sh --libs "base sameat base sameat" --workloads "kms2partyblock"
