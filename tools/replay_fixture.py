#!/usr/bin/env python3
"""Replay a fixture written by tools/dump_fixture.jl (the Julia reference; all five scheme kinds) -- or by --make
(this repo's oracle, same format, used to test the replay path) -- on the MI355X engine and compare bit for bit: the
mod-switched inputs, the accumulator after blindrotate! (through mkt_blindrotate_batch), the key switch of the
fixture's accumulator (mkt_keyswitch_batch) and the NAND outputs, so a mismatch localises to a stage.

  python tools/replay_fixture.py <dir>                 # needs a GPU
  python tools/replay_fixture.py --make <dir> [KIND]   # CPU: write a fixture from the oracle (reduced parameters);
                                                       # KIND = KMS | CGGI | LMSS | CCS | KMSblock
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
DT = {"f64": np.float64, "u32": np.uint32, "u64": np.uint64, "u8": np.uint8}


def load(d):
    man = json.load(open(os.path.join(d, "manifest.json")))
    assert man["format"] in ("mktfhe-fixture-1", "mktfhe-fixture-2")
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
            kw.update(pubkey=cx(a[f"pubkey{i}"]))
        if f"rlk_d{i}" in a:
            kw.update(rlk_d=cx(a[f"rlk_d{i}"]), rlk_f=cx(a[f"rlk_f{i}"]))
        s.load_party(i, **kw)
    B = man["batch"]
    x, y, ref = (a[n].reshape(B, p.lwe_len) for n in ("x", "y", "nand"))
    same = True
    if "acc" in a:                                         # format 2: intermediates of bootstrapping! (bootstrapping.jl:4-27)
        lin = ((np.arange(p.lwe_len) == p.lwe_len - 1).astype(np.uint32) * np.uint32(1 << 29) - x - y).astype(np.uint32)   # gate.jl:1-8
        at, bt = s.modswitch(lin)
        ok_ms = np.array_equal(at, a["atilde"].reshape(B, -1)) and np.array_equal(bt, a["btilde"])
        N = p.N
        acc0 = np.zeros((B, 1 + p.k, N), dtype=p.ring_dtype)
        E = p.ring_dtype(1 << (p.W - 3))
        for j in range(B):
            b = int(a["btilde"][j]); lo, hi = (E, -E) if b <= N else (-E, E); b = b if b <= N else b - N
            acc0[j, 0] = np.where(np.arange(N) < b, lo, hi).astype(p.ring_dtype)
        acc_ref = a["acc"].astype(p.ring_dtype).reshape(B, -1)
        acc = s.blindrotate_(a["atilde"].reshape(B, -1), acc0.reshape(B, -1).copy())
        ok_rot = np.array_equal(acc, acc_ref)
        ok_ks = np.array_equal(s.keyswitch(acc_ref.reshape(B, 1 + p.k, N)), ref)
        print(f"  mod-switch == fixture: {ok_ms}; accumulator after blindrotate! == fixture: {ok_rot}; keyswitch!(fixture accumulator) == fixture output: {ok_ks}")
        same = ok_ms and ok_rot and ok_ks
    out = s.gate(0, x, y)
    same = same and np.array_equal(out, ref)
    print(f"{man['producer']}: {B} NAND gates ({p.name}, scheme kind {p.scheme}), engine == fixture bit for bit: {same}")
    s.close()
    return same


def make(d, kind="KMS"):
    """same file format, produced by this repo's oracle at reduced parameters"""
    from helpers import O, encrypt_bits, keygen, mk, oracle_scheme
    os.makedirs(d, exist_ok=True)
    p = {"KMS": mk.KMS2party.scaled(n=12, N=256), "CGGI": mk.CGGIparam.scaled(n=16, N=256), "LMSS": mk.Blockparam.scaled(n=18, N=256, blk_d=6),
         "CCS": mk.CCS2party.scaled(n=8, N=256), "KMSblock": mk.KMS2partyblock.scaled(n=12, N=256, blk_d=4)}[kind]
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
    if p.multikey:
        put("crs", tr(crs), "f64")
    for i, kk in enumerate(keys):
        put(f"brk{i}", tr(kk.brk), "f64"); put(f"ksk{i}", kk.ksk, "u32"); put(f"lwekey{i}", kk.lwekey, "u32")
        if p.multikey:
            put(f"pubkey{i}", tr(kk.pubkey), "f64")
        if kk.rlk_d is not None:
            put(f"rlk_d{i}", tr(kk.rlk_d), "f64"); put(f"rlk_f{i}", tr(kk.rlk_f), "f64")
    B = 8
    bits = (np.arange(2 * B) % 3 == 0)
    c = encrypt_bits(p, keys, bits, seed=7700)
    x, y = c[:B], c[B:]
    put("x", x, "u32"); put("y", y, "u32"); put("nand", so.gate_batch(0, x, y, threads=4), "u32"); put("bits", bits.astype(np.uint8), "u8")
    ms = [so.modswitch(O.gate_linear(0, x[j], y[j])) for j in range(B)]
    put("atilde", np.stack([m[0] for m in ms]), "u32"); put("btilde", np.array([m[1] for m in ms], dtype=np.uint32), "u32")
    put("acc", np.stack([so.blindrotate(m[0], so.testvector(m[1])) for m in ms]).astype(p.ring_dtype), "u64" if p.W == 64 else "u32")
    pd = dict(scheme=p.scheme, n=p.n, N=p.N, k=p.k, W=p.W, l_gsw=p.l_gsw, logB_gsw=p.logB_gsw, l_lev=p.l_lev, logB_lev=p.logB_lev,
              l_uni=p.l_uni, logB_uni=p.logB_uni, f=p.f, logD=p.logD, blk_len=p.blk_len, blk_d=p.blk_d)
    json.dump({"format": "mktfhe-fixture-2", "producer": "oracle", "params": pd, "batch": B, "files": files}, open(os.path.join(d, "manifest.json"), "w"))
    print("wrote", d)


if __name__ == "__main__":
    if sys.argv[1] == "--make":
        make(sys.argv[2], *(sys.argv[3:4]))
    else:
        sys.exit(0 if replay(sys.argv[1]) else 1)
