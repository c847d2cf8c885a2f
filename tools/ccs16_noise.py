import sys, numpy as np
sys.path.insert(0, 'tests')
from helpers import *
p = mk.CCS16party
crs, keys = keygen(p, 3)
sg = gpu_scheme(p, crs, keys)
B = 16
rng = np.random.default_rng(5)
bits = rng.integers(0, 2, (p.k, B)).astype(bool)
cts = [np.stack([mk.lwe_ith_encrypt(int(bits[i, j]), i, keys[i], p, deterministic_seed=1000 * i + j) for j in range(B)]) for i in range(p.k)]
def phase_err(c, m):
    ph = c[:, -1].astype(np.int64)
    for i, kk in enumerate(keys):
        ph = (ph + (c[:, i*p.n:(i+1)*p.n].astype(np.int64) * kk.lwekey.astype(np.int64)).sum(1)) % (1 << 32)
    ph = np.where(ph >= 1 << 31, ph - (1 << 32), ph) / 2.0**32
    return ph - np.where(m, 0.125, -0.125)
res, mres = cts[0], bits[0].copy()
for i in range(1, p.k):
    res = sg.gate(0, res, cts[i]); mres = ~(mres & bits[i])
    e = phase_err(res, mres)
    print('parties involved %2d  phase error std %.4f max %.4f  wrong %d/%d' % (i + 1, e.std(), np.abs(e).max(), (np.abs(e) > 0.125).sum(), B), flush=True)
