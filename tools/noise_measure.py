"""Measured output noise of a gate bootstrap for parameter sets (run on the GPU box; the engine is bit-identical to the
oracle, so this is the noise of the reference's algorithm).  For every set: the phase error of NAND outputs whose inputs
involve every party (a NAND fold over one fresh encryption per party, as test/KMS.jl:29-34), its standard deviation
against the 1/8 decryption margin, and the same with the key noise switched off (beta = 0: ring keys noiseless; alpha = 0:
LWE side noiseless; both: only the algorithm's own rounding -- gadget rounding, Float64 transform error, truncations --
is left).  tools/noise_theory.py predicts the same figures from the papers' variance formulas.
  python tools/noise_measure.py [--batch 512] [--variants] NAME [NAME ...]   ->  one JSON line per (set, variant)"""
import argparse
import json
import sys

import numpy as np

sys.path.insert(0, 'tests')
from helpers import *   # noqa


def phase_err(p, keys, ct, bits):
    ph = ct[:, -1].astype(np.int64)
    for i, kk in enumerate(keys):
        ph = (ph + (ct[:, i * p.n:(i + 1) * p.n].astype(np.int64) * kk.lwekey.astype(np.int64)).sum(1)) % (1 << 32)
    ph = np.where(ph >= 1 << 31, ph - (1 << 32), ph) / 2.0**32
    return ph - np.where(bits, 0.125, -0.125)


def measure(p, B, arith=0):
    crs, keys = keygen(p, 12)
    sg = mk.setup(p, keys=keys, a=crs, arith=arith) if p.multikey else mk.setup(p, keys=keys[0], arith=arith)[1]
    k = p.nparty
    rng = np.random.default_rng(13)
    bits = rng.integers(0, 2, 2 * B * k).astype(bool)
    c = encrypt_bits(p, keys, bits, seed=7000)
    acc, ab = c[0::k].copy(), bits[0::k].copy()
    per_party = []
    for i in range(1, k):
        acc = sg.gate(0, acc, c[i::k]); ab = ~(ab & bits[i::k])
        e = phase_err(p, keys, acc, ab)
        per_party.append(float(e.std()))
    x, y, bx, by = acc[:B], acc[B:], ab[:B], ab[B:]
    out = sg.gate(0, x, y)
    want = ~(bx & by)
    e = phase_err(p, keys, out, want)
    dk = keys if p.multikey else keys[0]
    fails = int((mk.lwe_decrypt(out, dk, p) != want).sum())
    sg.close()
    return dict(sigma=float(e.std()), mean=float(e.mean()), max=float(np.abs(e).max()), fails=fails, gates=B, sigma_after_parties=per_party)


ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="+")
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--variants", action="store_true")
ap.add_argument("--arith", default="f64ref", choices=["f64ref", "exact"])
args = ap.parse_args()
for name in args.names:
    p0 = eval(name, {"mk": mk}) if "." in name or "(" in name else getattr(mk, name)
    variants = [("as shipped", {})]
    if args.variants:
        variants += [("beta=0", dict(beta=0.0)), ("alpha=0", dict(alpha=0.0)), ("alpha=beta=0", dict(alpha=0.0, beta=0.0))]
    for vn, kw in variants:
        p = p0.scaled(**kw) if kw else p0
        B = args.batch if p.nparty * p.N <= 8192 else max(64, args.batch // 4)
        r = measure(p, B, mk.ARITH_EXACT if args.arith == "exact" else mk.ARITH_F64REF)
        r.update(set=name, variant=vn, arith=args.arith)
        print(json.dumps(r), flush=True)
