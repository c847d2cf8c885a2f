# usage: bash tools/ntt_grid_sweep.sh  -- forward / inverse batched integer NTTs (MKT_ARITH_EXACT transform entry points, 64-bit ring: 16 N bytes per
# transform) against the cap on the workgroups of a launch (MKT_NTT_GRID, read once per process)
for g in 0 1024 2048 4096 8192 16384; do
MKT_NTT_GRID=$g python3 - <<PY
import sys
sys.path.insert(0, '.')
import torch
import mktfhe_amd as mk
for p in (mk.KMS2party_N1024_l2, mk.KMS2party):
    N = p.N
    sch = mk.Scheme(p, device=0, arith=mk.ARITH_EXACT)
    nb = (4 << 30) // (N * 16)
    dev = torch.device("cuda", 0)
    pv = torch.randint(-2**62, 2**62, (nb, N), dtype=torch.int64, device=dev)
    tout = torch.empty((nb, N // 2), dtype=torch.complex128, device=dev)
    back = torch.empty_like(pv)
    res = []
    for fn, a, o in ((sch.transform_fwd, pv, tout), (sch.transform_inv, tout, back)):
        fn(a, out=o); torch.cuda.synchronize()
        sch.enable_timing(True)
        for _ in range(5): fn(a, out=o)
        ms, cnt = sch.kernel_ms(3)
        sch.enable_timing(False)
        res.append(nb * N * 16 / (ms / cnt * 1e-3) / 8e12)
    print(f"NTT W=64 N={N} grid $g: fwd {res[0]:.3f}  inv {res[1]:.3f} of 8 TB/s", flush=True)
    sch.close(); del pv, tout, back
PY
done
