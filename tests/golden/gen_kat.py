#!/usr/bin/env python3
"""Known-answer fixture WITH its inputs: tiny parameter sets of all five schemes -- evaluation keys in integer form,
CRS, input ciphertexts, the LWE secrets, and the oracle's outputs (all six gates; for NAND also the accumulator after
blindrotate! (bootstrapping.jl:25) and the mod-switched input).  Unlike e2e_hashes.json it does not depend on the
client's random generator, and the GPU test replays it WITHOUT the oracle library.

The reference holds no vectors of its own and cannot run here; these were produced by the C oracle after
it passed the unit fixtures.  Regenerate from the repo root:  python tests/golden/gen_kat.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import O, encrypt_bits, keygen, mk, oracle_scheme  # noqa: E402

CASES = {
    "CGGI": mk.CGGIparam.scaled(n=6, N=64),
    "LMSS": mk.Blockparam.scaled(n=6, N=64, blk_d=2),
    "CCS": mk.CCS2party.scaled(n=4, N=64),
    "KMS": mk.KMS2party.scaled(n=4, N=64),
    "KMS_block": mk.KMS2partyblock.scaled(n=6, N=64, blk_d=2),
}
FIELDS = ("scheme", "n", "N", "k", "W", "l_gsw", "logB_gsw", "l_lev", "logB_lev", "l_uni", "logB_uni", "f", "logD", "blk_len", "blk_d")


def params_of(arr, name="kat"):
    kw = dict(zip(FIELDS, (int(v) for v in arr)))
    return mk.Params(name=name, alpha=2.0**17, beta=1.0, **kw)


def make(name):
    p = CASES[name]
    crs, keys = keygen(p, 91)
    so = oracle_scheme(p, crs, keys)
    k = p.nparty
    bits = np.array([0, 1, 1, 1, 0, 0, 1, 0] * k, dtype=bool)[:8 * k]
    c = encrypt_bits(p, keys, bits, seed=9100)                    # ciphertext j under party j mod k
    if k > 1:                                                     # every party's mask block populated: NAND folds
        acc, ab = c[0::k].copy(), bits[0::k].copy()
        for i in range(1, k):
            acc = np.stack([so.gate(0, acc[j], c[i::k][j]) for j in range(8)])
            ab = ~(ab & bits[i::k])
        c, bits = acc, ab
    x, y = c[:4], c[4:]
    d = {"params": np.array([getattr(p, f) for f in FIELDS], dtype=np.int64), "x": x, "y": y, "bits": bits,
         "lwekeys": np.stack([kk.lwekey for kk in keys])}
    if crs is not None:
        d["crs"] = crs
    for i, kk in enumerate(keys):
        d[f"brk{i}"], d[f"ksk{i}"] = kk.brk, kk.ksk
        if p.multikey:
            d[f"pub{i}"] = kk.pubkey
        if p.scheme in (mk.KMS, mk.KMS_BLOCK):
            d[f"rlkd{i}"], d[f"rlkf{i}"] = kk.rlk_d, kk.rlk_f
    d["out"] = np.stack([np.stack([so.gate(op, x[j], y[j]) for j in range(4)]) for op in range(6)])
    lin = np.stack([O.gate_linear(0, x[j], y[j]) for j in range(4)])
    ms = [so.modswitch(lin[j]) for j in range(4)]
    d["atilde"] = np.stack([m[0] for m in ms]); d["btilde"] = np.array([m[1] for m in ms], dtype=np.uint32)
    d["acc"] = np.stack([so.blindrotate(d["atilde"][j], so.testvector(d["btilde"][j])) for j in range(4)])
    return d


if __name__ == "__main__":
    out = {}
    for name in CASES:
        for key, v in make(name).items():
            out[f"{name}/{key}"] = v
    np.savez_compressed(os.path.join(HERE, "kat_tiny.npz"), **out)
    print({k: v.shape for k, v in out.items() if k.endswith("/out")}, os.path.getsize(os.path.join(HERE, "kat_tiny.npz")))
