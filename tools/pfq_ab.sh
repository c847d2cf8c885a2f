for v in 21 22; do echo "== variant $v"; MKT_ROT_VARIANT=$v LIBS="base pfq1 pfq2" WORKLOADS="lmss kms2partyblock" bash tools/ab_bench.sh 2>&1 | grep -v amdgpu.ids; done
