#!/usr/bin/env python3
"""Blind-rotation throughput per transform for parameter variants (diagnostic): which (W, l, n, M) shapes run below the
KMS N=1024 l=2 rate?  usage: python tools/rot_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mktfhe_amd as mk

B = 1024
cases = [
    ("CGGI l=3 (shipped)", mk.CGGIparam),
    ("CGGI l=2", mk.CGGIparam.scaled(l_gsw=2, logB_gsw=12)),
    ("CGGI l=4", mk.CGGIparam.scaled(l_gsw=4, logB_gsw=7)),
    ("CGGI l=3 n=560", mk.CGGIparam.scaled(n=560)),
    ("KMS N1024 l=2", mk.KMS2party_N1024_l2),
    ("KMS N1024 l=3", mk.KMS2party_N1024_l2.scaled(l_gsw=3, logB_gsw=12)),
    ("KMS N2048 l=3 (shipped)", mk.KMS2party),
    ("KMS N2048 l=2", mk.KMS2party.scaled(l_gsw=2, logB_gsw=16)),
]
for name, p in cases:
    if p.multikey:
        crs = mk.CRS(p, 1); keys = [mk.party_keygen(crs, p, deterministic_seed=1, party=i) for i in range(p.k)]
        sch = mk.setup(p, keys=keys, a=crs, device=0)
    else:
        keys = mk.PartyKeys(p, deterministic_seed=1); sch = mk.setup(p, keys=keys, device=0)[1]
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.integers(0, 2**32, (B, p.lwe_len), dtype=np.uint64).astype(np.uint32).view(np.int32)).cuda()
    y = torch.from_numpy(rng.integers(0, 2**32, (B, p.lwe_len), dtype=np.uint64).astype(np.uint32).view(np.int32)).cuda()
    out = torch.empty_like(x)
    for v in os.environ.get("VARIANTS", "0").split(","):
        os.environ["MKT_ROT_VARIANT"] = v
        mk.NAND(x, y, sch, out=out); torch.cuda.synchronize()
        sch.enable_timing(True)
        for _ in range(3): mk.NAND(x, y, sch, out=out)
        torch.cuda.synchronize()
        ms, cnt = sch.kernel_ms(1); sch.enable_timing(False)
        rows = (1 + (p.k - 1) * p.l_lev) if p.scheme in (mk.KMS, mk.KMS_BLOCK) else 1
        ntr = rows * B * p.n * (2 * p.l_gsw + 2)
        print(f"{name:26s} variant {v}: rot {ms/3:7.2f} ms  {ntr/(ms/3*1e-3)/1e6:6.0f} M transforms/s (M={p.N//2}, W={p.W}, l={p.l_gsw})", flush=True)
    sch.close()
