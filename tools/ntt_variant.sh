# usage: bash tools/ntt_variant.sh <sfx> "<-D flags>"  -- libmktfhe_hip_<sfx>.so = the default build with ntt_exact.hip recompiled with extra flags
# (development A/B builds of the EXACT kernels; the default build must be current; run them with tools/sweep.sh --libs "base <sfx>" -- --arith exact)
SFX=$1; EXTRA="$2"
cd $(dirname $0)/../mktfhe_amd/csrc
mkdir -p /tmp/mkt_tuv
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-cuda-compat -Wno-pass-failed -Wno-unused-function \
  $EXTRA -c ntt_exact.hip -o /tmp/mkt_tuv/ntt_exact_$SFX.o || { echo "compile failed"; exit 1; }
OBJ=$(ls build/*.o | grep -v ntt_exact.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libmktfhe_hip_$SFX.so $OBJ /tmp/mkt_tuv/ntt_exact_$SFX.o -lpthread && echo built libmktfhe_hip_$SFX.so
