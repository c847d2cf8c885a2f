"""Flat, versioned key-blob format for evaluation keys.

The reference never serialises keys (every run regenerates them: test/KMS.jl:5-12).  A blob carries ONE
party's evaluation key (or the common reference string) in the integer layouts of include/mktfhe.h, so that
a client -- this package's seeded keygen, or the Julia reference through a 30-line writer -- can hand keys
to an evaluator process.  Secret keys are never written.

  offset  size  field
  0       8     magic  b"MKTKEY\\0\\1"  (last byte = format version 1)
  8       60    mkt_params: 15 little-endian int32 (scheme, n, N, k, W, l_gsw, logB_gsw, l_lev, logB_lev,
                l_uni, logB_uni, f, logD, blk_len, blk_d)
  68      4     party index (int32, -1 for a CRS blob)
  72      4     number of sections S (int32)
  76      S*32  section table: name[16] (ASCII, NUL padded), dtype code int32 (4 = uint32, 8 = uint64),
                reserved int32, byte length int64
  ...           section payloads, each 16-byte aligned, in table order; sections: "crs" | "brk", "ksk",
                "rlk_d", "rlk_f", "pubkey" (those that exist for the scheme)
  end     32    SHA-256 of everything before it
"""
import hashlib
import struct

import numpy as np

from .params import Params

MAGIC = b"MKTKEY\x00\x01"
_PFIELDS = ("scheme", "n", "N", "k", "W", "l_gsw", "logB_gsw", "l_lev", "logB_lev", "l_uni", "logB_uni", "f", "logD", "blk_len", "blk_d")


def _pack(params: Params, party, sections):
    head = bytearray(MAGIC)
    head += struct.pack("<15i", *[getattr(params, f) for f in _PFIELDS])
    head += struct.pack("<ii", party, len(sections))
    table, payload = bytearray(), bytearray()
    for name, arr in sections:
        a = np.ascontiguousarray(arr)
        assert a.dtype in (np.uint32, np.uint64)
        table += struct.pack("<16siiq", name.encode(), a.dtype.itemsize, 0, a.nbytes)
    base = len(head) + len(table)
    for name, arr in sections:
        pad = (-(base + len(payload))) % 16
        payload += b"\0" * pad + np.ascontiguousarray(arr).tobytes()
    body = bytes(head + table + payload)
    return body + hashlib.sha256(body).digest()


def dump_party(keys) -> bytes:
    """serialise a PartyKeys' EVALUATION key (never the secret key)"""
    p = keys.params
    secs = [("brk", keys.brk), ("ksk", keys.ksk)]
    for name in ("rlk_d", "rlk_f", "pubkey"):
        v = getattr(keys, name)
        if v is not None:
            secs.append((name, v))
    return _pack(p, keys.party, secs)


def dump_arrays(params: Params, party, brk, ksk, rlk_d=None, rlk_f=None, pubkey=None) -> bytes:
    """serialise evaluation-key arrays (e.g. the export of Scheme.keygen_device on the party's own GPU)"""
    secs = [("brk", np.ascontiguousarray(brk, dtype=params.ring_dtype).reshape(-1)), ("ksk", np.ascontiguousarray(ksk, dtype=np.uint32).reshape(-1))]
    for name, v in (("rlk_d", rlk_d), ("rlk_f", rlk_f), ("pubkey", pubkey)):
        if v is not None:
            secs.append((name, np.ascontiguousarray(v, dtype=params.ring_dtype).reshape(-1)))
    return _pack(params, party, secs)


def dump_crs(params: Params, crs) -> bytes:
    return _pack(params, -1, [("crs", np.ascontiguousarray(crs, dtype=params.ring_dtype).reshape(-1))])


def load(blob: bytes):
    """-> (params dict, party, {section: ndarray}); raises ValueError on a corrupt or foreign blob"""
    if len(blob) < 76 + 32 or blob[:8] != MAGIC:
        raise ValueError("not a version-1 MKTKEY blob")
    if hashlib.sha256(blob[:-32]).digest() != blob[-32:]:
        raise ValueError("key blob checksum mismatch")
    pv = struct.unpack_from("<15i", blob, 8)
    party, nsec = struct.unpack_from("<ii", blob, 68)
    off = 76 + 32 * nsec
    out = {}
    for s in range(nsec):
        name, isz, _, nbytes = struct.unpack_from("<16siiq", blob, 76 + 32 * s)
        off += (-off) % 16
        dt = {4: np.uint32, 8: np.uint64}[isz]
        out[name.rstrip(b"\0").decode()] = np.frombuffer(blob, dtype=dt, count=nbytes // isz, offset=off)
        off += nbytes
    return dict(zip(_PFIELDS, pv)), party, out


def load_into(scheme, blob: bytes):
    """upload a blob's keys into a Scheme (params must match the context)"""
    pd, party, secs = load(blob)
    mine = {f: getattr(scheme.params, f) for f in _PFIELDS}
    if pd != mine:
        raise ValueError("key blob was generated for different parameters")
    if party < 0:
        scheme.load_crs(secs["crs"])
    else:
        scheme.load_party(party, brk=secs.get("brk"), ksk=secs.get("ksk"), rlk_d=secs.get("rlk_d"), rlk_f=secs.get("rlk_f"), pubkey=secs.get("pubkey"))
    return party
