for v in 22 21; do echo "== MKT_ROT_VARIANT=$v"; MKT_ROT_VARIANT=$v LIBS="base k1lt0 k1plain" WORKLOADS="cggi" bash tools/ab_bench.sh 2>&1 | grep -v amdgpu.ids; done
