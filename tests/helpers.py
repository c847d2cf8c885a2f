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


def gpu_scheme(p: mk.Params, crs, keys, device=0, arith=0):
    if p.multikey:
        return mk.setup(p, keys=keys, a=crs, device=device, arith=arith)
    return mk.setup(p, keys=keys[0], device=device, arith=arith)[1]


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


# ---- tests/golden/kat_tiny.npz (gen_kat.py): fixture with inputs AND expected outputs ----
def kat_cases():
    import types
    from golden.gen_kat import params_of
    g = np.load(os.path.join(ROOT, "tests", "golden", "kat_tiny.npz"))
    names = sorted({k.split("/")[0] for k in g.files})
    for name in names:
        d = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith(name + "/")}
        p = params_of(d["params"], name="kat_" + name)
        keys = []
        for i in range(p.nparty):
            keys.append(types.SimpleNamespace(brk=d[f"brk{i}"], ksk=d[f"ksk{i}"], pubkey=d.get(f"pub{i}"),
                                              rlk_d=d.get(f"rlkd{i}"), rlk_f=d.get(f"rlkf{i}"), lwekey=d["lwekeys"][i]))
        yield name, p, d, keys


def kat_decrypt(p, d, ct):
    """scheme.jl:388-407 with the fixture's LWE secrets, in numpy"""
    ph = ct[..., -1].astype(np.uint64)
    for i in range(p.nparty):
        ph = ph + (ct[..., i * p.n:(i + 1) * p.n].astype(np.uint64) * d["lwekeys"][i].astype(np.uint64)).sum(-1)
    ph = ph & np.uint64(0xFFFFFFFF)
    if p.multikey:
        return ph < (1 << 31)
    return ((ph >> np.uint64(29)) + ((ph >> np.uint64(28)) & np.uint64(1))) == 1
