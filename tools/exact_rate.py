"""Gate throughput of the MKT_ARITH_EXACT path (integer-NTT blind rotation, CGGI, 32-bit ring) beside the shipped Float64
path on the same keys and inputs: the measured cost of exact products on this part.
    python tools/exact_rate.py [--batch 1024] [--steps 3]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mktfhe_amd as mk

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--steps", type=int, default=3)
args = ap.parse_args()
B = args.batch
for p in (mk.CGGIparam, mk.CGGI_N1024_l2, mk.Blockparam):
    keys = mk.PartyKeys(p, deterministic_seed=1)
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2, 2 * B).astype(bool)
    ct = np.stack([mk.lwe_ith_encrypt(int(b), 0, keys, p, deterministic_seed=100 + j) for j, b in enumerate(bits)])
    x, y = ct[:B].copy(), ct[B:].copy()
    rates = {}
    for name, arith in (("F64REF", mk.ARITH_F64REF), ("EXACT", mk.ARITH_EXACT)):
        s = mk.Scheme(p, arith=arith)
        s.load_party(0, keys)
        out = s.gate(0, x, y)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = s.gate(0, x, y)
        dt = (time.perf_counter() - t0) / args.steps
        ok = np.array_equal(mk.lwe_decrypt(out, keys, p), ~(bits[:B] & bits[B:]))
        rates[name] = B / dt
        print(f"{p.name:16s} {name:7s} batch {B}: {dt * 1e3:8.2f} ms/step  {B / dt:10.0f} gates/s  decrypt_ok={ok}", flush=True)
        s.close()
    print(f"{p.name:16s} F64REF / EXACT = {rates['F64REF'] / rates['EXACT']:.2f}x")
