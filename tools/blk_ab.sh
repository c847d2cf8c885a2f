# usage: LIBS="base x" bash tools/blk_ab.sh  -- the block-binary workloads under both transform pairings (MKT_ROT_VARIANT 21 / 22)
for v in 21 22; do echo "== variant $v"; MKT_ROT_VARIANT=$v WORKLOADS="lmss kms2partyblock" bash tools/ab_bench.sh 2>&1 | grep -v amdgpu.ids; done
