#!/usr/bin/env python3
"""Regression fixture: SHA-256 of the oracle's output ciphertext words for seeded keys and inputs.

The reference holds no golden vectors; these hashes pin the oracle's bits so that a
later edit cannot silently change them.  They were produced by this script from the oracle AFTER it
passed the unit fixtures (twiddles, decomposition, exact products, decrypt-correctness).
Regenerate from the repo root:  python tests/golden/gen_e2e.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import encrypt_bits, keygen, mk, oracle_scheme  # noqa: E402

CASES = {
    "CGGI_n24_N256": mk.CGGIparam.scaled(n=24, N=256),
    "LMSS_n30_N256": mk.Blockparam.scaled(n=30, N=256, blk_d=10),
    "CCS2_n16_N256": mk.CCS2party.scaled(n=16, N=256),
    "KMS2_n16_N256": mk.KMS2party.scaled(n=16, N=256),
    "KMS2block_n24_N256": mk.KMS2partyblock.scaled(n=24, N=256, blk_d=8),
    "KMS2_n12_N1024_l2": mk.KMS2party_N1024_l2.scaled(n=12),
    "CGGI_n12_N1024": mk.CGGIparam.scaled(n=12, N=1024),
}


def run_case(name):
    p = CASES[name]
    crs, keys = keygen(p, 41)
    s = oracle_scheme(p, crs, keys)
    bits = np.array([0, 1, 1, 1, 0, 0, 1, 0], dtype=bool)
    c = encrypt_bits(p, keys, bits, seed=4100)
    h = hashlib.sha256()
    for op in range(6):
        for j in range(4):
            h.update(s.gate(op, c[j], c[4 + j]).tobytes())
    return h.hexdigest()


def run_mixed(name):
    """multi-key sets on ciphertexts that involve every party: NAND folds over one fresh encryption per party
    (test/KMS.jl:29-34), then all six gates on two such ciphertexts (fresh same-party pairs, as in run_case, leave
    the other parties' mask blocks zero and their rotations are all skips)"""
    p = CASES[name]
    crs, keys = keygen(p, 41)
    s = oracle_scheme(p, crs, keys)
    k = p.nparty
    bits = np.array([1, 0, 0, 1, 1, 1, 0, 1] * k, dtype=bool)[:4 * k]
    c = encrypt_bits(p, keys, bits, seed=4200)
    acc = [c[j * k] for j in range(4)]
    for i in range(1, k):
        acc = [s.gate(0, acc[j], c[j * k + i]) for j in range(4)]
    h = hashlib.sha256()
    for op in range(6):
        for j in range(2):
            h.update(s.gate(op, acc[j], acc[2 + j]).tobytes())
    return h.hexdigest()


MIXED = [name for name, p in CASES.items() if p.multikey]

if __name__ == "__main__":
    out = {name: run_case(name) for name in CASES}
    out.update({name + "/mixed": run_mixed(name) for name in MIXED})
    json.dump(out, open(os.path.join(HERE, "e2e_hashes.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
