# usage: bash tools/isa_scan.sh [LOGM=9]  -- every kernel of kernels.hip at one transform size: VGPRs, spills, stack bytes, instruction count
LM=${1:-9}
cd $(dirname $0)/../mktfhe_amd/csrc
for TU in 0 1 2 3 4 5 6 7; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-cuda-compat -Wno-pass-failed -Wno-unused-function \
      -DMKT_TU=$TU -DMKT_ONLY_LOGM=$LM --cuda-device-only -S kernels.hip -o /tmp/isa_scan_$TU.s 2>/dev/null ) &
done; wait
python3 - <<'PY'
import re, glob
for f in sorted(glob.glob('/tmp/isa_scan_*.s')):
    src = open(f).read()
    meta = {}
    for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', src):
        meta[m.group(1)] = (int(m.group(2)), int(m.group(3)), int(m.group(4)))
    for n in re.findall(r'^(_Z\S+):', src, re.M):
        s = src.index('\n' + n + ':'); e = src.find('s_endpgm', s)
        if e < 0: continue
        cnt = sum(1 for l in src[s:e].split('\n') if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':'))
        pv, vg, sp = meta.get(n, (-1, -1, -1))
        import subprocess
        dn = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
        print(f'{cnt:6d} instr  vgpr {vg:3d}  spill {sp:3d}  stack {pv:4d} B  {dn[:130]}')
PY
