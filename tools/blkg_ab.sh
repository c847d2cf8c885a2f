# usage (GPU box, repo root): [LIBS="base x"] [BATCH=1024] bash tools/blkg_ab.sh  -- block-binary workloads under each rotation grouping (MKT_ROT_BLKG 1 / 2 / 4)
for g in ${GS:-1 2 4}; do echo "== MKT_ROT_BLKG=$g"; MKT_ROT_BLKG=$g WORKLOADS="${WORKLOADS:-lmss kms2partyblock}" ARGS="--batch ${BATCH:-1024}" bash tools/ab_bench.sh 2>&1 | grep -v amdgpu.ids; done
