# usage: bash tools/fft_w32_sweep.sh  -- forward / inverse batched transforms on the 32-bit ring (12 N bytes per transform) against
# the launcher knobs (read once per process: one process per setting)
for nb in 1 2; do for g in 0 2048 4096 8192 16384; do
MKT_FFT_NB=$nb MKT_FFT_GRID=$g MKT_FFT_IGRID=$g python3 - <<PY
import sys, os
sys.path.insert(0, '.')
import torch
import mktfhe_amd as mk
for N in (1024, 2048):
    p = mk.CGGIparam.scaled(N=N)
    sch = mk.Scheme(p, device=0)
    nb = (4 << 30) // (N * 12)
    dev = torch.device("cuda", 0)
    pv = torch.randint(-2**31, 2**31 - 1, (nb, N), dtype=torch.int32, device=dev)
    tout = torch.empty((nb, N // 2), dtype=torch.complex128, device=dev)
    back = torch.empty_like(pv)
    res = []
    for fn, a, o in ((sch.transform_fwd, pv, tout), (sch.transform_inv, tout, back)):
        fn(a, out=o); torch.cuda.synchronize()
        sch.enable_timing(True)
        for _ in range(5): fn(a, out=o)
        ms, cnt = sch.kernel_ms(3)
        sch.enable_timing(False)
        res.append(nb * N * 12 / (ms / cnt * 1e-3) / 8e12)
    print(f"W=32 N={N} nb $nb grid $g: fwd {res[0]:.3f}  inv {res[1]:.3f} of 8 TB/s", flush=True)
    sch.close(); del pv, tout, back
PY
done; done
