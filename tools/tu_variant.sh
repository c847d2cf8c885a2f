# usage: bash tools/tu_variant.sh <sfx> <tu> "<-D flags>"  -- libmktfhe_hip_<sfx>.so = the default build with ONE translation
# unit of kernels.hip recompiled with extra flags (development A/B builds; the default build must be current)
SFX=$1; TU=$2; EXTRA="$3"
cd $(dirname $0)/../mktfhe_amd/csrc
mkdir -p /tmp/mkt_tuv
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-cuda-compat -Wno-pass-failed -Wno-unused-function \
  $EXTRA -DMKT_TU=$TU -c kernels.hip -o /tmp/mkt_tuv/kernels_tu${TU}_$SFX.o || exit 1
OBJ=$(ls build/*.o | grep -v kernels_tu$TU.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libmktfhe_hip_$SFX.so $OBJ /tmp/mkt_tuv/kernels_tu${TU}_$SFX.o -lpthread && echo built libmktfhe_hip_$SFX.so
