"""Output-noise probe (run on the GPU box): phase error of NAND outputs on inputs that involve every party, and of a
second bootstrap level, for parameter-set expressions given on the command line, e.g.
  python tools/noise_probe.py "mk.CGGI_N1024_l2" "mk.CGGI_N1024_l2.scaled(logB_gsw=9)" """
import sys
import numpy as np
sys.path.insert(0, 'tests')
from helpers import *   # noqa


def phase_err(p, keys, ct):
    ph = ct[:, -1].astype(np.int64)
    for i, kk in enumerate(keys):
        ph = (ph + (ct[:, i * p.n:(i + 1) * p.n].astype(np.int64) * kk.lwekey.astype(np.int64)).sum(1)) % (1 << 32)
    ph = np.where(ph >= 1 << 31, ph - (1 << 32), ph) / 2.0**32
    return np.abs(ph) - 0.125


def measure(p, B=1024):
    crs, keys = keygen(p, 12)
    sg = gpu_scheme(p, crs, keys)
    k = p.nparty
    rng = np.random.default_rng(13)
    bits = rng.integers(0, 2, 2 * B * k).astype(bool)
    c = encrypt_bits(p, keys, bits, seed=7000)
    acc, ab = c[0::k].copy(), bits[0::k].copy()
    for i in range(1, k):
        acc = sg.gate(0, acc, c[i::k]); ab = ~(ab & bits[i::k])
    x, y, bx, by = acc[:B], acc[B:], ab[:B], ab[B:]
    out = sg.gate(0, x, y)
    want = ~(bx & by)
    dk = keys if p.multikey else keys[0]
    f1 = int((mk.lwe_decrypt(out, dk, p) != want).sum())
    e1 = phase_err(p, keys, out)
    out2 = sg.gate(0, out, np.roll(out, 1, axis=0))
    f2 = int((mk.lwe_decrypt(out2, dk, p) != ~(want & np.roll(want, 1))).sum())
    e2 = phase_err(p, keys, out2)
    sg.close()
    return f1, e1.std(), np.abs(e1).max(), f2, e2.std(), np.abs(e2).max()


for expr in sys.argv[1:]:
    p = eval(expr)
    r = measure(p)
    print(expr, 'level1: fails %d err std %.4f max %.4f | level2: fails %d std %.4f max %.4f (margin 0.125)' % r, flush=True)
