#!/usr/bin/env python3
"""usage (GPU box): [MKT_FFT_GRID=.. MKT_FFT_IGRID=.. MKT_FFT_NB=.. MKT_NTT_GRID=..] python3 tools/legs.py [--arith f64ref exact] [--N 1024 2048] [--W 64 32] [--reps 5]
The batched transform legs (BASELINE metric 2) alone: forward and inverse, 4 GiB per launch, GB/s of algorithmic bytes N (W/8 + 8) and the
fraction of 8 TB/s.  The launcher knobs are read once per process, so a sweep is one process per setting: tools/sweep.sh-style loops
(`for g in 0 2048 8192; do MKT_FFT_GRID=$g python3 tools/legs.py --N 2048 --W 32; done`).  Replaces fft_sweep.py, fft_bench.sh,
fft_w32/w64_sweep.sh, ntt_grid_sweep.sh, ntt_only.py, ntt_legs.sh."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import mktfhe_amd as mk
ap = argparse.ArgumentParser()
ap.add_argument("--arith", nargs="+", default=["f64ref", "exact"])
ap.add_argument("--N", type=int, nargs="+", default=[1024, 2048])
ap.add_argument("--W", type=int, nargs="+", default=[64, 32])
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--warm", type=int, default=30, help="untimed launches first: the part needs tens of milliseconds of back-to-back work to reach the clock it then holds (5 timed launches after 2 warm-ups read 15-25 %% low)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
knobs = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith(("MKT_FFT_", "MKT_NTT_")))
for W in a.W:
    for N in a.N:
        p = (mk.KMS2party if W == 64 else mk.CGGIparam).scaled(n=8, N=N)
        per = N * (W // 8 + 8)
        nb = (4 << 30) // per
        polys = torch.randint(-2**31, 2**31 - 1, (nb, N * (2 if W == 64 else 1)), dtype=torch.int32, device=dev)
        pv = polys.view(torch.int64) if W == 64 else polys
        tr = torch.empty((nb, N // 2), dtype=torch.complex128, device=dev)
        back = torch.empty_like(pv)
        for ar in a.arith:
            s = mk.Scheme(p, device=0, arith=mk.ARITH_EXACT if ar == "exact" else mk.ARITH_F64REF)
            out = []
            for fn in ((lambda: s.transform_fwd(pv, out=tr)), (lambda: s.transform_inv(tr, out=back))):
                for _ in range(a.warm):
                    fn()
                torch.cuda.synchronize()
                s.enable_timing(True)
                for _ in range(a.reps):
                    fn()
                ms, cnt = s.kernel_ms(3)
                s.enable_timing(False)
                out.append(nb * per / (ms / cnt * 1e-3) / 1e9)
            print(f"{ar:6s} N {N} W {W}: forward {out[0]:6.0f} GB/s ({out[0] / 8000:.3f})  inverse {out[1]:6.0f} GB/s ({out[1] / 8000:.3f})  {knobs}", flush=True)
            s.close()
        del polys, tr, back
        torch.cuda.empty_cache()
