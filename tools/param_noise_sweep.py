"""Noise study of the synthetic BASELINE shape (KMS k=2, N=1024, l_gsw=2): output phase error over 1024 NAND gates
for a few gadget choices (run on the GPU box; the engine is bit-identical to the oracle)."""
import sys, numpy as np
sys.path.insert(0, 'tests')
from helpers import *
base = mk.KMS2party_N1024_l2
cands = [dict(), dict(logB_gsw=14), dict(logB_gsw=18), dict(logB_gsw=20), dict(l_lev=3, logB_lev=6), dict(l_lev=2, logB_lev=9),
         dict(l_uni=4, logB_uni=8), dict(l_uni=3, logB_uni=12), dict(logB_gsw=18, l_uni=4, logB_uni=9), dict(logB_gsw=18, l_lev=2, logB_lev=9, l_uni=4, logB_uni=9),
         dict(n=500), dict(logB_gsw=18, l_lev=3, logB_lev=6, l_uni=4, logB_uni=9)]
for kw in cands:
    p = base.scaled(**kw)
    crs, keys = keygen(p, 12)
    sg = gpu_scheme(p, crs, keys)
    B = 1024
    rng = np.random.default_rng(13)
    bits = rng.integers(0, 2, 256).astype(bool)
    uniq = encrypt_bits(p, keys, bits, seed=7000)
    idx = rng.integers(0, 256, 2 * B)
    c = uniq[idx]; bb = bits[idx]
    out = sg.gate(0, c[:B], c[B:])
    got = mk.lwe_decrypt(out, keys, p)
    want = ~(bb[:B] & bb[B:])
    ph = out[:, -1].astype(np.int64)
    for i, kk in enumerate(keys):
        ph = (ph + (out[:, i*p.n:(i+1)*p.n].astype(np.int64) * kk.lwekey.astype(np.int64)).sum(1)) % (1 << 32)
    ph = np.where(ph >= 1 << 31, ph - (1 << 32), ph) / 2.0**32
    err = np.abs(ph) - 0.125
    print(kw, 'fails', int((got != want).sum()), 'err std %.4f max %.4f' % (err.std(), np.abs(err).max()), flush=True)
    sg.close()
