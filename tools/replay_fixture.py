#!/usr/bin/env python3
"""Replay a fixture written by tools/dump_fixture.jl (the Julia reference) -- or by --make (this repo's oracle, same
format, used to test the replay path) -- on the MI355X engine and compare the NAND outputs bit for bit.

  python tools/replay_fixture.py <dir>            # needs a GPU
  python tools/replay_fixture.py --make <dir>     # CPU: write a fixture from the oracle (reduced KMS parameters)
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
DT = {"f64": np.float64, "u32": np.uint32, "u8": np.uint8}


def load(d):
    man = json.load(open(os.path.join(d, "manifest.json")))
    assert man["format"] == "mktfhe-fixture-1"
    arr = {f["name"]: np.fromfile(os.path.join(d, f["name"] + ".bin"), dtype=DT[f["dtype"]]) for f in man["files"]}
    return man, arr


def replay(d):
    import mktfhe_amd as mk
    from mktfhe_amd import _lib
    man, a = load(d)
    pd = man["params"]
    p = mk.Params("fixture", pd["scheme"], pd["n"], pd["N"], pd["k"], pd["W"], 0.0, 0.0, l_gsw=pd["l_gsw"], logB_gsw=pd["logB_gsw"],
                  l_lev=pd["l_lev"], logB_lev=pd["logB_lev"], l_uni=pd["l_uni"], logB_uni=pd["logB_uni"], f=pd["f"], logD=pd["logD"],
                  blk_len=pd["blk_len"], blk_d=pd["blk_d"])
    s = mk.Scheme(p)
    tabs = [np.ascontiguousarray(a[n]) for n in ("psi", "psiinv", "roots", "rootsinv")]
    _lib.check(_lib.lib().mkt_set_twiddles(s.h, *[t.ctypes.data_as(C.c_void_p) for t in tabs]), s.h)
    cx = lambda v: np.ascontiguousarray(v).view(np.complex128)
    if p.multikey:
        s.load_crs(cx(a["crs"]), fmt=mk.FMT_F64_FFT)
    for i in range(p.nparty):
        kw = dict(brk=cx(a[f"brk{i}"]), ksk=a[f"ksk{i}"], fmt=mk.FMT_F64_FFT)
        if p.multikey:
            kw.update(rlk_d=cx(a[f"rlk_d{i}"]), rlk_f=cx(a[f"rlk_f{i}"]), pubkey=cx(a[f"pubkey{i}"]))
        s.load_party(i, **kw)
    B = man["batch"]
    x, y, ref = (a[n].reshape(B, p.lwe_len) for n in ("x", "y", "nand"))
    out = s.gate(0, x, y)
    same = np.array_equal(out, ref)
    print(f"{man['producer']}: {B} NAND gates, engine == fixture bit for bit: {same}")
    s.close()
    return same


def make(d):
    """same file format, produced by this repo's oracle at reduced KMS parameters"""
    from helpers import encrypt_bits, keygen, mk, oracle_scheme
    os.makedirs(d, exist_ok=True)
    p = mk.KMS2party.scaled(n=12, N=256)
    crs, keys = keygen(p, 77)
    so = oracle_scheme(p, crs, keys)
    f = so.ffter
    files = []

    def put(name, arr, dt):
        arr = np.ascontiguousarray(arr)
        arr.tofile(os.path.join(d, name + ".bin"))
        files.append({"name": name, "dtype": dt, "shape": list(arr.shape)})
    for w, n in enumerate(("psi", "psiinv", "roots", "rootsinv")):
        put(n, f.table(w).view(np.float64), "f64")
    tr = lambda v: f.fwd(v.astype(np.uint64).reshape(-1, p.N)).view(np.float64)
    put("crs", tr(crs), "f64")
    for i, kk in enumerate(keys):
        put(f"brk{i}", tr(kk.brk), "f64"); put(f"ksk{i}", kk.ksk, "u32"); put(f"rlk_d{i}", tr(kk.rlk_d), "f64")
        put(f"rlk_f{i}", tr(kk.rlk_f), "f64"); put(f"pubkey{i}", tr(kk.pubkey), "f64"); put(f"lwekey{i}", kk.lwekey, "u32")
    B = 8
    bits = (np.arange(2 * B) % 3 == 0)
    c = encrypt_bits(p, keys, bits, seed=7700)
    put("x", c[:B], "u32"); put("y", c[B:], "u32"); put("nand", so.gate_batch(0, c[:B], c[B:], threads=4), "u32"); put("bits", bits.astype(np.uint8), "u8")
    pd = dict(scheme=p.scheme, n=p.n, N=p.N, k=p.k, W=p.W, l_gsw=p.l_gsw, logB_gsw=p.logB_gsw, l_lev=p.l_lev, logB_lev=p.logB_lev,
              l_uni=p.l_uni, logB_uni=p.logB_uni, f=p.f, logD=p.logD, blk_len=0, blk_d=0)
    json.dump({"format": "mktfhe-fixture-1", "producer": "oracle", "params": pd, "batch": B, "files": files}, open(os.path.join(d, "manifest.json"), "w"))
    print("wrote", d)


if __name__ == "__main__":
    if sys.argv[1] == "--make":
        make(sys.argv[2])
    else:
        sys.exit(0 if replay(sys.argv[1]) else 1)
