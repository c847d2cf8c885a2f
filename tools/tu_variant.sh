# usage: bash tools/tu_variant.sh <sfx> <tu[,tu...]> "<-D flags>"  -- libmktfhe_hip_<sfx>.so = the default build with the named
# translation units of kernels.hip recompiled with extra flags (development A/B builds; the default build must be current)
SFX=$1; TUS=$(echo $2 | tr ',' ' '); EXTRA="$3"
cd $(dirname $0)/../mktfhe_amd/csrc
mkdir -p /tmp/mkt_tuv
OBJ=$(ls build/*.o); NEW=""
for TU in $TUS; do
  TF=$(grep "^TUFLAGS_$TU *=" Makefile | sed 's/^[^=]*= *//' | sed 's/$(OPT_TRACKERS)/-mllvm -amdgpu-use-amdgpu-trackers/; s/$(OPT_MAXILP)/-mllvm -amdgpu-sched-strategy=max-ilp/; s/$(OPT_MEMCLAUSE)/-mllvm -amdgpu-sched-strategy=max-memory-clause/')        # the unit's own flags (Makefile)
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-cuda-compat -Wno-pass-failed -Wno-unused-function \
      $TF $EXTRA -DMKT_TU=$TU -c kernels.hip -o /tmp/mkt_tuv/kernels_tu${TU}_$SFX.o || touch /tmp/mkt_tuv/failed_$SFX ) &
  OBJ=$(echo "$OBJ" | grep -v kernels_tu$TU.o); NEW="$NEW /tmp/mkt_tuv/kernels_tu${TU}_$SFX.o"
done; wait
[ -e /tmp/mkt_tuv/failed_$SFX ] && { rm -f /tmp/mkt_tuv/failed_$SFX; echo "compile failed"; exit 1; }
# the variant names itself: context.o rebuilt with a build id over the sources AND the extra flags (bench.py quotes committed PMC traffic only for the build it was taken from)
BID=$( (cat *.hip *.h *.cpp Makefile ../../include/mktfhe.h; echo "variant tu $2 $EXTRA"; /opt/rocm/bin/hipcc --version) | sha256sum | cut -c1-16)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-cuda-compat -Wno-pass-failed -Wno-unused-function -DMKT_BUILD_ID="\"$BID\"" -x hip -c context.cpp -o /tmp/mkt_tuv/context_$SFX.o || exit 1
OBJ=$(echo "$OBJ" | grep -v "/context.o"); NEW="$NEW /tmp/mkt_tuv/context_$SFX.o"
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libmktfhe_hip_$SFX.so $OBJ $NEW -lpthread && echo built libmktfhe_hip_$SFX.so
