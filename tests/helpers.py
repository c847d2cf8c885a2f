"""Shared test helpers: build the ORACLE scheme object from the product's client-side keys."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
import mktfhe_amd as mk  # noqa: E402


def ora_params(p: mk.Params):
    return O.OraParams(p.scheme, p.n, p.N, p.k, p.W, p.l_gsw, p.logB_gsw, p.l_lev, p.logB_lev,
                       p.l_uni, p.logB_uni, p.f, p.logD, p.blk_len, p.blk_d)


def keygen(p: mk.Params, seed=1):
    """-> (crs or None, [PartyKeys])"""
    if p.multikey:
        a = mk.CRS(p, seed)
        return a, [mk.party_keygen(a, p, deterministic_seed=seed, party=i) for i in range(p.k)]
    return None, [mk.PartyKeys(p, deterministic_seed=seed, party=0)]


def oracle_scheme(p: mk.Params, crs, keys):
    s = O.Scheme(ora_params(p))
    if crs is not None:
        s.set_crs(crs.astype(np.uint64))
    for i, k in enumerate(keys):
        s.set_brk(i, k.brk.astype(np.uint64))
        s.set_ksk(i, k.ksk)
        if p.multikey:
            s.set_pubkey(i, k.pubkey.astype(np.uint64))
        if p.scheme in (mk.KMS, mk.KMS_BLOCK):
            s.set_rlk(i, k.rlk_d.astype(np.uint64), k.rlk_f.astype(np.uint64))
    return s


def gpu_scheme(p: mk.Params, crs, keys, device=0):
    if p.multikey:
        return mk.setup(p, keys=keys, a=crs, device=device)
    return mk.setup(p, keys=keys[0], device=device)[1]


def encrypt_bits(p: mk.Params, keys, bits, seed=100):
    """bit j is encrypted under party (j mod nparty) (lwe_ith_encrypt layout, scheme.jl:379-386)"""
    out = np.empty((len(bits), p.lwe_len), dtype=np.uint32)
    for j, b in enumerate(bits):
        i = j % p.nparty
        out[j] = mk.lwe_ith_encrypt(int(b), i, keys[i], p, deterministic_seed=seed + j)
    return out


GATE_FUNCS = {
    0: lambda x, y: ~(x & y), 1: lambda x, y: x & y, 2: lambda x, y: x | y,
    3: lambda x, y: x ^ y, 4: lambda x, y: ~(x ^ y), 5: lambda x, y: ~(x | y),
}
