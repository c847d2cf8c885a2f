# usage: bash tools/isa.sh "<extra -D flags>" [TU=2] [LOGM=9] [name-filter]  -- device ISA of one translation unit of
# kernels.hip (development build restricted to one transform size), main-loop instruction mix and register use
EXTRA="$1"; TU=${2:-2}; LM=${3:-9}; FILT=${4:-blindrotate_k1}
OUT=/tmp/isa_tu${TU}_$(echo "$EXTRA" | md5sum | cut -c1-6).s
cd $(dirname $0)/../mktfhe_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-cuda-compat -Wno-pass-failed -Wno-unused-function \
  -DMKT_TU=$TU -DMKT_ONLY_LOGM=$LM $EXTRA --cuda-device-only -S kernels.hip -o $OUT 2>&1 | grep -i error
python3 ../../tools/isa_loop_count.py $OUT $FILT
grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size|name):" $OUT | paste - - - - - - | grep $FILT | sed 's/ \+/ /g' | cut -c1-300
echo $OUT
