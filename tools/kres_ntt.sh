# usage: bash tools/kres_ntt.sh [LOGN=10] ["<extra -D flags>"] [name-filter]  -- registers, spills, scratch, LDS and occupancy of the
# EXACT-mode kernels (csrc/ntt_exact.hip) at ONE transform size (development build, seconds)
LN=${1:-10}; EXTRA="$2"; FILT=${3:-.}
cd $(dirname $0)/../mktfhe_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-cuda-compat -Wno-pass-failed -Wno-unused-function \
  -DMKT_NTT_ONLY_LOGN=$LN $EXTRA --cuda-device-only -S ntt_exact.hip -o /tmp/ntt_exact_$LN.s 2>&1 | grep -iE "error|warning: v"
grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size|name):" /tmp/ntt_exact_$LN.s | paste - - - - - - | grep -E "$FILT" | sed 's/ \+/ /g; s/_ZN4mktd12_GLOBAL__N_1//' | cut -c1-260
echo /tmp/ntt_exact_$LN.s
