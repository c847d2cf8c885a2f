#!/usr/bin/env python3
"""Does the blind-rotation time depend on the input ciphertexts?  random words / fresh encryptions / gate outputs (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mktfhe_amd as mk
B = 1024
for p in (mk.KMS2party_N1024_l2, mk.KMS2party):
    crs = mk.CRS(p, 1); keys = [mk.party_keygen(crs, p, deterministic_seed=1, party=i) for i in range(p.k)]
    sch = mk.setup(p, keys=keys, a=crs, device=0)
    rng = np.random.default_rng(5)
    def fresh(n, s0):
        b = rng.integers(0, 2, n).astype(bool); ct = np.empty((n, p.lwe_len), dtype=np.uint32)
        for j in range(n): ct[j] = mk.lwe_ith_encrypt(int(b[j]), j % p.k, keys[j % p.k], p, deterministic_seed=s0 + j)
        return torch.from_numpy(ct.view(np.int32)).cuda()
    f = fresh(4 * B, 1000)
    sets = {
        "random": (torch.from_numpy(rng.integers(0, 2**32, (B, p.lwe_len), dtype=np.uint64).astype(np.uint32).view(np.int32)).cuda(),
                   torch.from_numpy(rng.integers(0, 2**32, (B, p.lwe_len), dtype=np.uint64).astype(np.uint32).view(np.int32)).cuda()),
        "fresh same-party": (f[:B].clone(), f[B:2*B].clone()),
        "fresh cross-party": (f[:B].clone(), f[B+1:2*B+1].clone()),
        "dense": (mk.NAND(f[:B], f[B:2*B], sch), mk.NAND(f[2*B:3*B], f[3*B:], sch)),
    }
    for name, (x, y) in sets.items():
        xz = (x.cpu().numpy().view(np.uint32)[:, :-1].reshape(B, p.k, p.n) == 0).mean(axis=(0, 2))
        out = torch.empty_like(x)
        mk.NAND(x, y, sch, out=out); torch.cuda.synchronize()
        sch.enable_timing(True)
        for _ in range(3): mk.NAND(x, y, sch, out=out)
        torch.cuda.synchronize(); ms, cnt = sch.kernel_ms(1); sch.enable_timing(False)
        print(f"{p.name:20s} {name:18s} rot {ms/3:7.2f} ms ({cnt} launches)  zero fraction of x per party block {xz}", flush=True)
    sch.close()
