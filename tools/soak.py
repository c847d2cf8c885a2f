"""Determinism soak: repeated batches of every scheme must reproduce their first result word for word (an intermittent
race in an LDS exchange or a missing barrier would show up as a rare mismatch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, 'tests')
import numpy as np, torch
from helpers import *
REPS = int(os.environ.get("REPS", "40"))
ARITH = mk.ARITH_EXACT if os.environ.get("ARITH") == "exact" else mk.ARITH_F64REF   # ARITH=exact: the integer-NTT gate paths (all five schemes; CGGI / LMSS up to RLWE length 3)
sets = [mk.CGGIparam, mk.KMS2party_N1024_l2, mk.KMS2party, mk.Blockparam, mk.Blockparam_k2, mk.KMS2partyblock, mk.CCS2party,
        mk.CGGIparam.scaled(n=64, N=256), mk.KMS2party.scaled(n=64, N=512), mk.KMS4party.scaled(n=32, N=4096), mk.CGGIparam.scaled(n=64, N=2048, k=2)]
bad = 0
SMALL = {mk.CGGIparam.name, mk.KMS2party_N1024_l2.name, mk.Blockparam_k2.name}      # also at 64 gates: the latency variant of the rotation / ragged groups
if os.environ.get("ONLY"):
    sets = [p for p in sets if p.name in os.environ["ONLY"].split(",")]
for p, Bs in [(p, None) for p in sets] + [(p, 64) for p in sets if p.name in SMALL and p.n > 100]:
    crs, keys = keygen(p, 3)
    sg = gpu_scheme(p, crs, keys, arith=ARITH)
    B = Bs or (1024 if p.N <= 2048 and p.n > 100 else 512)
    bits = np.random.default_rng(4).integers(0, 2, 2 * B + 1).astype(bool)
    c = encrypt_bits(p, keys, bits, seed=40)
    x = torch.from_numpy(c[:B].view(np.int32)).cuda(); y = torch.from_numpy(c[B + 1:].view(np.int32)).cuda()
    ref = mk.NAND(x, y, sg).clone()
    ref2 = mk.NAND(ref, mk.NAND(y, x, sg), sg).clone()          # dense second level
    t0 = time.time(); mism = 0
    for r in range(REPS):
        o = mk.NAND(x, y, sg)
        mism += int(not torch.equal(o, ref))
        o2 = mk.NAND(o, mk.NAND(y, x, sg), sg)
        mism += int(not torch.equal(o2, ref2))
    torch.cuda.synchronize()
    print(f"{p.name:20s} n={p.n:4d} N={p.N:5d} B={B}: {2*REPS} batches, mismatches {mism}, {time.time()-t0:.1f}s", flush=True)
    bad += mism
    sg.close()
print("SOAK", "FAILED" if bad else "OK")
