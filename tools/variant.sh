# usage: bash tools/variant.sh <source: fx_exact | ntt_exact_0..3 | rot_block_32 | rot_block_64 | ccs_pipe | keygen> <sfx> "<-D flags>"
# libmktfhe_hip_<sfx>.so = the default build with ONE translation unit recompiled with extra flags and context.o rebuilt with a build id
# that names the variant (so that bench.py never quotes the default build's committed PMC traffic for it).  Development A/B builds; run
# them with tools/sweep.sh --libs "base <sfx>" (selected through MKT_LIB_PATH).  The default build must be current.
SRC=$1; SFX=$2; EXTRA="$3"
cd $(dirname $0)/../mktfhe_amd/csrc
mkdir -p /tmp/mkt_tuv
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-cuda-compat -Wno-pass-failed -Wno-unused-function"
case $SRC in
  rot_block_32|rot_block_64) FILE=rot_block.hip; EXTRA="$EXTRA -DMKT_BLK_WORD=${SRC#rot_block_}";;
  ntt_exact_[0-3]) FILE=ntt_exact.hip; EXTRA="$EXTRA -DMKT_NTT_TU=${SRC#ntt_exact_}";;      # one kernel family of ntt_exact.hip (see its header)
  *) FILE=$SRC.hip;;
esac
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c $FILE -o /tmp/mkt_tuv/${SRC}_$SFX.o || { echo "compile failed"; exit 1; }
BID=$( (cat *.hip *.h *.cpp Makefile ../../include/mktfhe.h; echo "variant $SRC $EXTRA"; /opt/rocm/bin/hipcc --version) | sha256sum | cut -c1-16)
/opt/rocm/bin/hipcc $FLAGS -DMKT_BUILD_ID="\"$BID\"" -x hip -c context.cpp -o /tmp/mkt_tuv/context_$SFX.o || exit 1
OBJ=$(ls build/*.o | grep -v "/$SRC.o" | grep -v "/context.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libmktfhe_hip_$SFX.so $OBJ /tmp/mkt_tuv/${SRC}_$SFX.o /tmp/mkt_tuv/context_$SFX.o -lpthread && echo built libmktfhe_hip_$SFX.so build_id $BID
