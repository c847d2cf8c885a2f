"""Coordinate-descent noise study of the synthetic BASELINE shape (KMS k=2, N=1024, l_gsw=2) on CROSS-PARTY gates (run on the
GPU box; the engine is bit-identical to the oracle): output phase error std of NAND(x under party 0, y under party 1)."""
import sys, numpy as np
sys.path.insert(0, 'tests')
from helpers import *
base = mk.KMS2party_N1024_l2

def measure(p):
    crs, keys = keygen(p, 12)
    sg = gpu_scheme(p, crs, keys)
    B = 1024
    rng = np.random.default_rng(13)
    bits = rng.integers(0, 2, 2 * B).astype(bool)
    c = encrypt_bits(p, keys, bits, seed=7000)        # ciphertext j under party j mod k
    x, y = c[0::2][:B // 1], c[1::2][:B // 1]
    bx, by = bits[0::2], bits[1::2]
    out = sg.gate(0, x, y)
    got = mk.lwe_decrypt(out, keys, p)
    want = ~(bx & by)
    ph = out[:, -1].astype(np.int64)
    for i, kk in enumerate(keys):
        ph = (ph + (out[:, i*p.n:(i+1)*p.n].astype(np.int64) * kk.lwekey.astype(np.int64)).sum(1)) % (1 << 32)
    ph = np.where(ph >= 1 << 31, ph - (1 << 32), ph) / 2.0**32
    err = np.abs(ph) - 0.125
    sg.close()
    return int((got != want).sum()), err.std(), np.abs(err).max()

best = dict()
stages = [
    [dict(logB_gsw=b) for b in (13, 14, 15, 16, 17, 18, 20)],
    [dict(l_lev=l, logB_lev=b) for l, b in ((2, 6), (2, 7), (2, 8), (2, 9), (2, 10), (3, 5), (3, 6), (3, 7), (4, 5))],
    [dict(l_uni=l, logB_uni=b) for l, b in ((3, 8), (3, 10), (3, 12), (3, 14), (4, 8), (4, 10), (4, 12), (5, 8), (6, 8))],
    [dict(logB_gsw=b) for b in (14, 15, 16, 17, 18)],
]
for st in stages:
    res = []
    for kw in st:
        cur = dict(best); cur.update(kw)
        f, s, m = measure(base.scaled(**cur))
        print(cur, 'fails', f, 'err std %.4f max %.4f' % (s, m), flush=True)
        res.append((s, kw))
    best.update(min(res, key=lambda r: r[0])[1])
    print('-> best so far', best, flush=True)
