# usage: bash tools/blk_variant.sh <sfx> "<-D flags>"  -- libmktfhe_hip_<sfx>.so = the default build with rot_block.hip (both word
# sizes) recompiled with extra flags (development A/B builds for tools/sweep.sh --libs; the default build must be current)
SFX=$1; EXTRA="$2"
cd $(dirname $0)/../mktfhe_amd/csrc
mkdir -p /tmp/mkt_tuv
TF=$(grep "^TUFLAGS_BLK *=" Makefile | sed 's/^[^=]*= *//' | sed 's/$(OPT_TRACKERS)/-mllvm -amdgpu-use-amdgpu-trackers/; s/$(OPT_MAXILP)/-mllvm -amdgpu-sched-strategy=max-ilp/; s/$(OPT_MEMCLAUSE)/-mllvm -amdgpu-sched-strategy=max-memory-clause/')
for W in 32 64; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-cuda-compat -Wno-pass-failed -Wno-unused-function \
      $TF $EXTRA -DMKT_BLK_WORD=$W -c rot_block.hip -o /tmp/mkt_tuv/rot_block_${W}_$SFX.o || touch /tmp/mkt_tuv/failed_$SFX ) &
done; wait
[ -e /tmp/mkt_tuv/failed_$SFX ] && { rm -f /tmp/mkt_tuv/failed_$SFX; echo "compile failed"; exit 1; }
OBJ=$(ls build/*.o | grep -v rot_block_)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libmktfhe_hip_$SFX.so $OBJ /tmp/mkt_tuv/rot_block_32_$SFX.o /tmp/mkt_tuv/rot_block_64_$SFX.o -lpthread && echo built libmktfhe_hip_$SFX.so
