#!/usr/bin/env python3
"""Transform-kernel bandwidth sweep (diagnostic): MKT_FFT_GRID / MKT_FFT_NB knobs, forward and inverse, N in {1024, 2048}.
usage: python tools/fft_sweep.py [grid ...]   (0 = library default)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mktfhe_amd as mk

grids = [int(g) for g in sys.argv[1:]] or [0]
tag = os.environ.get("TAG", "")
for pname in ("KMS2party_N1024_l2", "KMS2party", "CGGIparam"):
    p = getattr(mk, pname)
    sch = mk.Scheme(p, device=0)
    N = p.N
    nb = (4 << 30) // (N * (p.W // 8 + 8))
    dev = torch.device("cuda", 0)
    polys = torch.randint(-2**31, 2**31 - 1, (nb, N * (2 if p.W == 64 else 1)), dtype=torch.int32, device=dev)
    pv = polys.view(torch.int64) if p.W == 64 else polys
    tout = torch.empty((nb, N // 2), dtype=torch.complex128, device=dev)
    back = torch.empty_like(pv)
    for g in grids + grids:
        for nbt in os.environ.get("NBS", "-").split(","):
            if g: os.environ["MKT_FFT_GRID"] = str(g); os.environ["MKT_FFT_IGRID"] = str(g)
            else: os.environ.pop("MKT_FFT_GRID", None); os.environ.pop("MKT_FFT_IGRID", None)
            if nbt != "-": os.environ["MKT_FFT_NB"] = nbt
            res = []
            for fn, a, o in ((sch.transform_fwd, pv, tout), (sch.transform_inv, tout, back)):
                fn(a, out=o); torch.cuda.synchronize()
                sch.enable_timing(True)
                for _ in range(5): fn(a, out=o)
                ms, cnt = sch.kernel_ms(3)
                sch.enable_timing(False)
                res.append(nb * N * (p.W // 8 + 8) / (ms / cnt * 1e-3) / 1e9)
            print(f"{tag} {pname:20s} grid {g:7d} nb {nbt}: fwd {res[0]:6.0f} GB/s  inv {res[1]:6.0f} GB/s", flush=True)
    sch.close(); del polys, tout, back
