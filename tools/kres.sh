# usage: bash tools/kres.sh <file.hip> [extra hipcc flags]  -- VGPRs / scratch / occupancy of every kernel of one source file (run from mktfhe_amd/csrc)
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-pass-failed "$@" -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kres.o 2>&1 \
 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*(Function Name|VGPRs|ScratchSize \[bytes\/lane\]|Occupancy \[waves\/SIMD\]): ([^ ]*).*/\2/' | paste - - - - | while read n v s o; do echo "$(echo $n | c++filt | sed 's/(mktd::RotArgs.*//; s/void mktd:://') vgpr=$v scratch=$s occ=$o"; done
