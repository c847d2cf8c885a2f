"""CPU tests (-m "not gpu"): the oracle against this repo's pinned fixtures and independent
restatements, the host-side logic, and the C-ABI surface.  No GPU compute."""
import ctypes as C
import sys
import hashlib
import json
import os
import re

import numpy as np
import pytest

from helpers import GATE_FUNCS, O, ROOT, encrypt_bits, keygen, mk, oracle_scheme

GOLD = os.path.join(ROOT, "tests", "golden")


# ---------------------------------------------------------------- tables (fft.jl:18-45)
@pytest.mark.parametrize("N", [16, 256, 1024, 2048, 4096])
def test_twiddle_tables_match_mpmath_fixture(N):
    g = np.load(os.path.join(GOLD, "twiddles.npz"))
    f = O.Ffter(N, 64)
    from mktfhe_amd import _lib
    for w, name in enumerate(("psi", "psiinv", "roots", "rootsinv")):
        gold = g[f"{name}_{N}"].view(np.uint64)
        assert np.array_equal(f.table(w).view(np.uint64), gold), ("oracle", name)
        out = np.empty(N // 2, dtype=np.complex128)
        assert _lib.lib().mkt_make_twiddles(N, w, out.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(out.view(np.uint64), gold), ("engine host generator", name)


def test_twiddle_special_entries():
    f = O.Ffter(1024, 32)
    psi = f.table(0)
    assert psi[0] == 1.0 and np.signbit(psi[0].imag)            # exp(0 - 0im): -0.0 kept
    assert psi[1].imag == -1.0 and 0 < psi[1].real < 1e-70       # cos(RN256(pi)/2), not 0
    assert float(psi[1].real).hex() == "0x1.452821e638d01p-257"


# ---------------------------------------------------------------- integer semantics
def py_divbits(a, bit, W):                       # arithmetic.jl:23-27
    m = (1 << W) - 1
    a &= m
    if bit == 0:
        return a
    return ((a >> bit) + (((a << (W - bit)) & m) >> (W - 1))) & m


def py_decomp(a, l, logB, W):                    # gsw.jl:42-52, written out step by step
    m = (1 << W) - 1
    mask, half = (1 << logB) - 1, 1 << (logB - 1)
    ai = py_divbits(a, W - l * logB, W)
    out = [0] * l
    for i in range(l - 1, 0, -1):
        d = ai & mask
        ai >>= logB
        ai = (ai + (d >> (logB - 1))) & m
        out[i] = (d - ((d & half) << 1)) & m
    d = ai & mask
    out[0] = (d - ((d & half) << 1)) & m
    return out


def test_divbits_kats():
    for W in (32, 64):
        m = (1 << W) - 1
        for a in (0, 1, m, m - 1, 1 << (W - 1), (1 << (W - 1)) - 1, 0x12345678 & m, (1 << 20) + (1 << 19)):
            for bit in (0, 1, 2, 20, 21, W - 3, W - 1):
                assert O.divbits(a, bit, W) == py_divbits(a, bit, W)
    # rounds half up and can return 2^(w-bit): the 2N case of bootstrapping.jl:8-9 (N = 1024)
    assert O.divbits(0xFFFFFFFF, 21, 32) == 2048
    assert O.divbits(0xFFF00000 - 1, 21, 32) == 2047
    assert O.divbits((1 << 20), 21, 32) == 1 and O.divbits((1 << 20) - 1, 21, 32) == 0


GADGETS = [(32, 3, 9), (32, 3, 8), (32, 4, 8), (32, 5, 6), (32, 12, 2), (64, 3, 12), (64, 2, 7), (64, 3, 10), (64, 5, 8),
           (64, 2, 8), (64, 7, 6), (64, 4, 9), (64, 3, 6), (64, 8, 4), (64, 9, 4), (64, 6, 7), (64, 3, 7), (64, 16, 2), (64, 2, 16)]


@pytest.mark.parametrize("W,l,logB", GADGETS)      # every (l, logB) of params.jl
def test_decomposition_kats(W, l, logB):
    rng = np.random.default_rng(l * 64 + logB)
    m = (1 << W) - 1
    vals = [0, 1, m, m - 1, 1 << (W - 1), (1 << (W - 1)) - 1, (1 << (W - 1)) + 1, 1 << (W - l * logB), (1 << (W - l * logB)) - 1,
            (1 << (W - l * logB - 1)) if W - l * logB > 0 else 3, m - (1 << (W - l * logB)) + 1]
    vals += [int(x) for x in rng.integers(0, 1 << 63, 200, dtype=np.uint64)]
    B = 1 << logB
    for a in vals:
        a &= m
        got = [int(x) for x in O.decomp_word(a, l, logB, W)]
        assert got == py_decomp(a, l, logB, W)
        # digits are balanced and recompose to the rounded value mod 2^W
        sd = [x - (1 << W) if x >> (W - 1) else x for x in got]
        assert all(-B // 2 <= d < B // 2 for d in sd)
        recomposed = sum(d << (W - (j + 1) * logB) for j, d in enumerate(sd)) & m
        rounded = (py_divbits(a, W - l * logB, W) << (W - l * logB)) & m
        assert recomposed == rounded
    # unbalanced (key switch, gsw.jl:34-40)
    for a in vals[:40]:
        got = [int(x) for x in O.unbalanced_decomp_word(a & 0xFFFFFFFF, 8, 2, 32)]
        t = py_divbits(a & 0xFFFFFFFF, 16, 32)
        assert got == [(t >> (2 * (7 - j))) & 3 for j in range(8)]


def test_gate_linear_constants():
    x = np.array([5, 7, 100], dtype=np.uint32); y = np.array([1, 2, 3], dtype=np.uint32)
    M = 1 << 32
    exp = {0: [(-5 - 1) % M, (-7 - 2) % M, ((1 << 29) - 103) % M], 1: [6, 9, ((7 << 29) + 103) % M], 2: [6, 9, ((1 << 29) + 103) % M],
           3: [12, 18, ((1 << 30) + 206) % M], 4: [(-12) % M, (-18) % M, ((3 << 30) - 206) % M], 5: [(-6) % M, (-9) % M, ((7 << 29) - 103) % M]}
    for op, e in exp.items():
        assert [int(v) for v in O.gate_linear(op, x, y)] == e


# ---------------------------------------------------------------- transform semantics
def py_negacyclic(a, b, W):
    N = len(a)
    out = [0] * N
    for i in range(N):
        for j in range(N):
            if i + j < N:
                out[i + j] += int(a[i]) * int(b[j])
            else:
                out[i + j - N] -= int(a[i]) * int(b[j])
    return [v % (1 << W) for v in out]


def test_schoolbook_helper_against_bigint():
    rng = np.random.default_rng(5)
    for W in (32, 64):
        a = rng.integers(0, 1 << 63, 32, dtype=np.uint64) & np.uint64((1 << W) - 1)
        b = rng.integers(0, 1 << 63, 32, dtype=np.uint64) & np.uint64((1 << W) - 1)
        assert [int(v) for v in O.negacyclic(a, b, W)] == py_negacyclic(a, b, W)


def signed_digits(rng, N, logB, W):
    d = rng.integers(-(1 << (logB - 1)), 1 << (logB - 1), N).astype(np.int64)
    return d.astype(np.uint64) & np.uint64((1 << W) - 1)


def test_f64ref_product_truncation_envelope_u32():
    """on the 32-bit ring native() truncates, so digit x key products come out exact or one
    below the exact negacyclic product -- never anything else."""
    rng = np.random.default_rng(9)
    N = 1024
    f = O.Ffter(N, 32)
    acc = np.zeros(N // 2, dtype=np.complex128)
    exact = np.zeros(N, dtype=np.uint64)
    for _ in range(6):
        d = signed_digits(rng, N, 9, 32)
        k = rng.integers(0, 1 << 32, N, dtype=np.uint64)
        td, tk = f.fwd(d), f.fwd(k)
        O.lib().ora_tp_muladd(acc.ctypes.data, td.ctypes.data, tk.ctypes.data, N // 2)
        exact = (exact + O.negacyclic(d, k, 32)) & np.uint64(0xFFFFFFFF)
    got = f.inv(acc)
    diff = (got.astype(np.int64) - exact.astype(np.int64))
    diff = np.where(diff > (1 << 31), diff - (1 << 32), diff)
    diff = np.where(diff < -(1 << 31), diff + (1 << 32), diff)
    assert set(np.unique(diff)).issubset({0, -1})
    assert (diff == 0).mean() > 0.3 and (diff == -1).mean() > 0.2


def test_f64ref_product_error_u64():
    """64-bit ring: the Float64 result is the exact product up to ~2^30 (53-bit mantissa at magnitude 2^81)."""
    rng = np.random.default_rng(10)
    N = 2048
    f = O.Ffter(N, 64)
    acc = np.zeros(N // 2, dtype=np.complex128)
    exact = np.zeros(N, dtype=np.uint64)
    for _ in range(6):
        d = signed_digits(rng, N, 12, 64)
        k = rng.integers(0, 1 << 63, N, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, N, dtype=np.uint64)
        td, tk = f.fwd(d), f.fwd(k)
        O.lib().ora_tp_muladd(acc.ctypes.data, td.ctypes.data, tk.ctypes.data, N // 2)
        exact = exact + O.negacyclic(d, k, 64)
    got = f.inv(acc)
    diff = (got - exact).astype(np.int64)
    assert np.abs(diff).max() < (1 << 33)
    assert np.abs(diff).max() > (1 << 20)          # it is NOT exact: parity is with the Float64 path


def test_forward_inverse_roundtrip_small_values():
    """ifft(fft(p)) is p or p-1: native() truncates (arithmetic.jl:1-9), it does not round"""
    rng = np.random.default_rng(11)
    for N, W in ((64, 32), (1024, 32), (2048, 64)):
        f = O.Ffter(N, W)
        p = signed_digits(rng, N, 10, W)
        d = (f.inv(f.fwd(p)) - p) & np.uint64((1 << W) - 1)
        if W == 32:
            assert set(int(v) for v in np.unique(d)).issubset({0, (1 << W) - 1})
        else:   # negative values sit at 2^64 - |d|, where a double's spacing is 2^11
            assert np.abs(d.astype(np.int64)).max() <= 2048


def test_forward_transform_edge_words_in_both_halves():
    """fft.jl:60: coefficient i + M enters as -signed(p[i + M]) with the negation done in the INTEGER type, so typemin
    (0x80..0) wraps to itself and converts to -2^(W-1), not +2^(W-1).  Special words at indices < M and >= M, the C oracle
    against the independent numpy transcription, raw bits."""
    import ref_numpy as R
    rng = np.random.default_rng(61)
    for N, W in ((256, 32), (256, 64), (1024, 32), (1024, 64), (2048, 64)):
        m = (1 << W) - 1
        special = [0, 1, 2, m, m - 1, 1 << (W - 1), (1 << (W - 1)) - 1, (1 << (W - 1)) + 1, 1 << (W - 3), m - (1 << (W - 3)) + 1]
        v = (rng.integers(0, 1 << 63, N, dtype=np.uint64) * 2 + rng.integers(0, 2, N, dtype=np.uint64)) & np.uint64(m)
        v[:10] = special; v[N // 2:N // 2 + 10] = special; v[N - 10:] = special[::-1]
        t_c = O.Ffter(N, W).fwd(v[None, :])[0]
        t_n = R.FFT(N, W).fwd(v.astype(np.uint64 if W == 64 else np.uint32))
        assert np.array_equal(t_c.real.view(np.uint64), t_n.re.view(np.uint64)), (N, W)
        assert np.array_equal(t_c.imag.view(np.uint64), t_n.im.view(np.uint64)), (N, W)
        # the typemin word in the upper half contributes -(-2^(W-1)) ... i.e. the imaginary input is -2^(W-1) exactly
        one = np.zeros(N, dtype=np.uint64); one[N // 2 + 3] = 1 << (W - 1)
        z = O.Ffter(N, W).fwd(one[None, :])[0]
        lone = np.zeros(N, dtype=np.uint64); lone[3] = 1 << (W - 1)
        zl = O.Ffter(N, W).fwd(lone[None, :])[0]
        # both are -2^(W-1) placed in the imaginary / real slot of point 3: z = i * zl
        assert np.array_equal(z.real.view(np.uint64), (-zl.imag).view(np.uint64)) and np.array_equal(z.imag.view(np.uint64), zl.real.view(np.uint64))


def test_exact_ntt_restatement_against_schoolbook():
    """tests/ref_ntt.py (the checker of the MKT_ARITH_EXACT transforms on the GPU) against the oracle's exact schoolbook
    product: NTT(a) . NTT(b) -> inverse == a (*) b mod (X^N + 1, 2^W) for digit polynomials a, and fwd -> inv = id"""
    import ref_ntt as R
    rng = np.random.default_rng(81)
    for N, W in ((32, 32), (128, 64), (256, 32)):
        a = rng.integers(-256, 256, N).astype(np.int64)
        aw = a.astype(np.uint64) & np.uint64((1 << W) - 1)
        b = rng.integers(0, 1 << 63, N, dtype=np.uint64) & np.uint64((1 << W) - 1)
        ref = O.negacyclic(aw, b, W)
        got = np.zeros(N, dtype=np.uint64)
        za = R.fwd(aw, W)
        for h in range(W // 32):                                  # 32-bit pieces of b keep every true coefficient below P / 2
            piece = [(int(x) >> (32 * h)) & 0xFFFFFFFF for x in b]
            zb = R.fwd(piece, 64)
            prod = R.inv(R.pmul(za, zb), 64)
            got = (got + (np.array(prod, dtype=np.uint64) << np.uint64(32 * h))) & np.uint64((1 << W) - 1)
        assert np.array_equal(got, ref), (N, W)
        assert R.inv(R.fwd(aw, W), W) == [int(x) for x in aw]


def test_every_shipped_set_fits_the_exact_modulus():
    """MKT_ARITH_EXACT is exact only while every transform-domain product sum stays below P / 2 (context.cpp exact_gate_ok, restated
    here): centered 32-bit words / pieces (magnitude <= 2^31) against balanced digits (<= 2^(logB-1)), N terms per product.  With the
    two 30-bit primes (P = 2^59.9998) every parameter set of params.jl must still fit -- the tightest are the headline shape
    KMS k=2 N=1024 l=2 base 2^16 (2^58, thanks to the coefficient-domain monomial of KMS phase 1) and KMS2partyblock (2^58.2)."""
    import ref_ntt as R
    half_P = R.P / 2
    assert all(q < 2**30 and (q - 1) % 8192 == 0 for q in R.PRIMES) and R.PRIMES[0] < R.PRIMES[1]
    worst = {}
    for name in dir(mk):
        p = getattr(mk, name)
        if not isinstance(p, mk.Params):
            continue
        n31 = p.N * 2.0**31
        if p.scheme in (mk.KMS, mk.KMS_BLOCK):
            ph1 = (2.0 * p.blk_len if p.scheme == mk.KMS_BLOCK else 1.0) * 2.0 * p.l_gsw * 2.0**(p.logB_gsw - 1) * n31
            acc = (p.l_lev * 2.0**(p.logB_lev - 1) + 2.0 * p.l_uni * 2.0**(p.logB_uni - 1)) * n31
            tv = p.k * p.l_uni * 2.0**(p.logB_uni - 1) * n31
            b = max(ph1, acc, tv)
        elif p.scheme == mk.CCS:
            b = 2.0 * (p.k + 2.0) * p.l_uni * 2.0**(p.logB_uni - 1) * n31
        else:
            # CGGI too carries the doubling: the kernel multiplies the product sum by X^a - 1 in the transform domain before its one lift
            b = 2.0 * (p.blk_len if p.scheme == mk.LMSS else 1.0) * (p.k + 1.0) * p.l_gsw * 2.0**(p.logB_gsw - 1) * n31   # (k + 1) l digit polynomials per key bit
        worst[name] = b
        assert b < half_P, (name, np.log2(b))
    assert len(worst) >= 19 and abs(np.log2(worst["KMS2party_N1024_l2"]) - 58.0) < 1e-9


@pytest.mark.parametrize("p", [mk.CGGIparam.scaled(n=9, N=128), mk.Blockparam.scaled(n=9, N=128, blk_d=3),
                               mk.CGGIparam.scaled(n=8, N=128, k=2), mk.Blockparam_k2.scaled(n=9, N=128, blk_d=3)], ids=lambda p: f"{p.name}-k{p.k}")
def test_exact_gate_restatement_decrypts(p):
    """tests/ref_exact.py (the checker of the MKT_ARITH_EXACT gate path on the GPU): gates bootstrapped with exact products
    decrypt to the plaintext gate, and one CMux step / block differs from the oracle's Float64 step by 0..2 per coefficient
    (the truncating native(), arithmetic.jl:1-9)"""
    import ref_exact as RX
    crs, keys = keygen(p, 33)
    so = oracle_scheme(p, crs, keys)
    bits = np.array([1, 0, 1, 1, 0, 1, 0, 0], dtype=bool)
    c = encrypt_bits(p, keys, bits, seed=3300)
    for op in (0, 3):
        out = np.stack([RX.gate(p, so, keys[0].brk, op, c[j], c[4 + j]) for j in range(4)])
        assert np.array_equal(mk.lwe_decrypt(out, keys[0], p), GATE_FUNCS[op](bits[:4], bits[4:]))
    lin = O.gate_linear(0, c[0], c[4])
    at, bt = so.modswitch(lin)
    L = max(p.blk_len, 1)
    one = np.zeros_like(at); one[:L] = at[:L]                      # the first step (block) only
    acc0 = so.testvector(bt)
    rot = RX.blindrotate_lmss if p.blk_len > 1 else RX.blindrotate
    ex = rot(p, keys[0].brk, one, acc0).astype(np.int64)
    fl = so.blindrotate(one, acc0.copy()).astype(np.int64).reshape(-1)
    d = (ex - fl + (1 << 31)) % (1 << 32) - (1 << 31)
    assert d.min() >= 0 and d.max() <= 2 * L * p.k                  # one truncation per product sum; (k + 1) l products per sum


def test_monomial_table_semantics():
    N = 64
    f = O.Ffter(N, 32)
    rng = np.random.default_rng(12)
    p = signed_digits(rng, N, 8, 32)
    tp = f.fwd(p)
    for e in (1, 5, N - 1, N, N + 1, 2 * N - 1, 2 * N):
        out = np.zeros(N // 2, dtype=np.complex128)
        mono = f.monomial(e)
        O.lib().ora_tp_mul(out.ctypes.data, mono.ctypes.data, tp.ctypes.data, N // 2)
        got = f.inv(out).astype(np.int64)
        # (X^e - 1) * p exactly
        coeffs = np.zeros(N, dtype=np.int64)
        ps = p.astype(np.int64); ps = np.where(ps >= 1 << 31, ps - (1 << 32), ps)
        for i in range(N):
            j = (i + e) % (2 * N)
            if j < N:
                coeffs[j] += ps[i]
            else:
                coeffs[j - N] -= ps[i]
        coeffs -= ps
        diff = (got - coeffs % (1 << 32)) % (1 << 32)       # exact or one below (truncating native())
        assert set(int(v) for v in np.unique(diff)).issubset({0, (1 << 32) - 1}), e
    assert not f.monomial(2 * N).any()


# ---------------------------------------------------------------- end to end (the reference's own test property)
SMALL = [
    mk.CGGIparam.scaled(n=24, N=256), mk.Blockparam.scaled(n=30, N=256, blk_d=10), mk.CCS2party.scaled(n=16, N=256),
    mk.KMS2party.scaled(n=16, N=256), mk.KMS2partyblock.scaled(n=24, N=256, blk_d=8), mk.KMS4party.scaled(n=8, N=256),
    mk.CCS4party.scaled(n=8, N=256),
]


@pytest.mark.parametrize("p", SMALL, ids=lambda p: p.name)
def test_random_gate_circuit_decrypts(p):
    """counterpart of test/{CGGI,LMSS,CCS,KMS,KMSblock}.jl: fold random gates, bootstrap once more, decrypt"""
    crs, keys = keygen(p, 21)
    s = oracle_scheme(p, crs, keys)
    rng = np.random.default_rng(22)
    dk = keys if p.multikey else keys[0]
    for trial in range(4):
        nb = max(p.nparty, 4)
        bits = rng.integers(0, 2, nb).astype(bool)
        c = encrypt_bits(p, keys, bits, seed=300 + 20 * trial)
        res, mres = c[0], bool(bits[0])
        for i in range(1, nb):
            op = int(rng.integers(0, 6))
            res = s.gate(op, res, c[i])
            mres = bool(GATE_FUNCS[op](np.bool_(mres), np.bool_(bits[i])))
        res = s.bootstrap(res)
        assert mk.lwe_decrypt(res, dk, p) == mres


def test_end_to_end_golden_hashes():
    """regression pin of the oracle's ciphertext bits (fixture written by tests/golden/gen_e2e.py)"""
    gold = json.load(open(os.path.join(GOLD, "e2e_hashes.json")))
    from golden.gen_e2e import CASES, MIXED, run_case, run_mixed
    for name in CASES:
        assert run_case(name) == gold[name], name
    for name in MIXED:      # ciphertexts that involve every party (no skipped rotations)
        assert run_mixed(name) == gold[name + "/mixed"], name


# ---------------------------------------------------------------- second, independent restatement (numpy)
def test_kat_fixture_reproduced_by_the_oracle():
    """tests/golden/kat_tiny.npz holds inputs (integer keys, CRS, ciphertexts) and expected outputs for all five
    schemes; the oracle reproduces every word, and the outputs decrypt to the gate truth tables"""
    from helpers import kat_cases, kat_decrypt
    seen = 0
    for name, p, d, keys in kat_cases():
        so = oracle_scheme(p, d.get("crs"), keys)
        x, y = d["x"], d["y"]
        for op in range(6):
            got = np.stack([so.gate(op, x[j], y[j]) for j in range(4)])
            assert np.array_equal(got, d["out"][op]), (name, op)
            assert np.array_equal(kat_decrypt(p, d, got), GATE_FUNCS[op](d["bits"][:4], d["bits"][4:])), (name, op, "decrypt")
        for j in range(4):
            at, bt = so.modswitch(O.gate_linear(0, x[j], y[j]))
            assert np.array_equal(at, d["atilde"][j]) and bt == d["btilde"][j]
            assert np.array_equal(so.blindrotate(at, so.testvector(bt)), d["acc"][j]), (name, "blindrotate")
        seen += 1
    assert seen == 5


@pytest.mark.parametrize("p", [
    mk.CGGIparam.scaled(n=10, N=256), mk.CGGIparam.scaled(n=6, N=256, k=2), mk.KMS2party.scaled(n=6, N=256),
    mk.KMS4party.scaled(n=4, N=256, k=3), mk.KMS2party_N1024_l2.scaled(n=4), mk.CGGIparam, mk.KMS2party.scaled(n=40),
    mk.Blockparam.scaled(n=12, N=256, blk_d=4), mk.Blockparam.scaled(n=30, N=256, blk_d=10, k=2),
    mk.Blockparam.scaled(n=300, N=256, blk_d=100, k=2),                 # n > N: the LMSS key switch copies a whole component
    mk.CCS2party.scaled(n=6, N=256), mk.CCS4party.scaled(n=3, N=256, k=3),
    mk.KMS2partyblock.scaled(n=12, N=256, blk_d=4), mk.KMS4partyblock.scaled(n=6, N=256, blk_d=2, k=3),
], ids=lambda p: f"{p.name}-n{p.n}-N{p.N}-k{p.k}")
def test_numpy_restatement_equals_the_c_oracle(p):
    """tests/ref_numpy.py transcribes bootstrapping.jl / fft.jl / gsw.jl a second time (numpy, one array op per
    reference operation, twiddles from the mpmath fixture); its NAND outputs equal the C oracle's word for word --
    on single-party and on mixed-party inputs, all five schemes (RLWE length 1-2, 2-3 parties), reduced and shipped sizes"""
    import ref_numpy as RN
    crs, keys = keygen(p, 9)
    so = oracle_scheme(p, crs, keys)
    rs = RN.Scheme(p, crs, keys)
    bits = np.array([1, 0, 1, 1, 0, 1, 0, 0], dtype=bool)
    c = encrypt_bits(p, keys, bits, seed=90)
    full = p.n >= 40
    pairs = [(0, 3)] if full else [(0, 3), (1, 4), (2, 6)]          # j, j+3: cross-party for k = 2, 3
    outs = []
    for j, q in pairs:
        ref = so.gate(0, c[j], c[q])
        assert np.array_equal(rs.nand(c[j], c[q]), ref), (j, q)
        outs.append(ref)
    if not full:                                                     # second level: inputs that involve every party
        ref = so.gate(0, outs[0], outs[1])
        assert np.array_equal(rs.nand(outs[0], outs[1]), ref)


# ---------------------------------------------------------------- client keys are what the layouts say
def test_client_key_layouts():
    p = mk.KMS2party.scaled(n=8, N=64)
    crs, keys = keygen(p, 31)
    k0 = keys[0]
    n, N, f, D1 = p.n, p.N, p.f, 3
    ksk = k0.ksk.reshape(N, D1, f, n + 1)
    s = k0.lwekey.astype(np.uint32)
    # every KSK row decrypts to (d+1) * z_j * 2^(32-(t+1)logD) up to noise alpha (keygen.jl:110-114)
    phase = (ksk[..., n] + (ksk[..., :n] * s).sum(-1, dtype=np.uint32)).astype(np.uint32)
    for t in range(4):      # deeper gadget levels sit below the noise alpha = 2^17 by design
        q = (phase[:, :, t].astype(np.int64) + (1 << (31 - 2 * (t + 1)))) >> (32 - 2 * (t + 1))
        zj = (q[:, 0] & 3)
        assert set(np.unique(zj)).issubset({0, 1})
        for d in range(D1):
            assert np.array_equal(q[:, d] & 3, ((d + 1) * zj) & 3)
    brk = k0.brk.reshape(n, 2 * p.l_gsw, 2, N)
    assert brk.dtype == np.uint64 and k0.rlk_d.size == p.l_uni * N and k0.rlk_f.size == 2 * p.l_uni * N and k0.pubkey.size == p.l_uni * N
    # seeded: same seed -> same keys, different party -> different keys
    again = mk.party_keygen(crs, p, deterministic_seed=31, party=0)
    assert np.array_equal(again.brk, k0.brk) and not np.array_equal(keys[1].brk, k0.brk)


# ---------------------------------------------------------------- the C ABI surface
def test_library_exports_every_declared_symbol():
    from mktfhe_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mktfhe.h")).read()
    declared = set(re.findall(r"\b(mkt_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mkt_status"}
    L = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/mktfhe.h but not exported"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    assert _lib.lib().mkt_abi_version() == 3
    # the reference-side shim (unexecuted: no Julia in the image) may only ccall symbols the header declares
    jl = open(os.path.join(ROOT, "integration", "MKTFHEHip.jl")).read()
    called = set(re.findall(r"ccall\(\(:(mkt_[a-z0-9_]+)", jl))
    assert called and called <= declared, called - declared
    # ... and so may the C examples
    for ex in ("kms_nand.c", "multi_nand.c"):
        used = set(re.findall(r"\b(mkt_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "examples", ex)).read()))
        assert used and used <= declared, (ex, used - declared)


def test_no_cpu_fallback_and_argument_errors():
    """without a GPU the product path must fail loudly, never compute on the CPU"""
    import torch
    p = mk.CGGIparam.scaled(n=8, N=64)
    if not torch.cuda.is_available():
        with pytest.raises(mk.MktError) as ei:
            mk.Scheme(p)
        assert ei.value.code == -3
    with pytest.raises(mk.MktError):
        mk.PartyKeys(p.scaled(N=100))            # not a power of two
    with pytest.raises(mk.MktError):
        mk.PartyKeys(mk.Blockparam.scaled(n=31, N=64, blk_d=10))   # n != blk_len * blk_d


def test_product_package_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mktfhe_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".h", ".hip")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in src.lower() or fn == "scheme.py" and False, f"{fn} mentions the oracle"


def test_keyblob_roundtrip_and_corruption():
    p = mk.KMS2party.scaled(n=8, N=64)
    crs, keys = keygen(p, 61)
    blob = mk.keyblob.dump_party(keys[1])
    pd, party, secs = mk.keyblob.load(blob)
    assert party == 1 and pd["N"] == 64 and pd["scheme"] == mk.KMS
    for name in ("brk", "ksk", "rlk_d", "rlk_f", "pubkey"):
        assert np.array_equal(secs[name], getattr(keys[1], name)), name
    assert "lwekey" not in secs                                   # secrets never leave
    cb = mk.keyblob.dump_crs(p, crs)
    assert np.array_equal(mk.keyblob.load(cb)[2]["crs"].reshape(crs.shape), crs) and mk.keyblob.load(cb)[1] == -1
    bad = bytearray(blob); bad[200] ^= 1
    with pytest.raises(ValueError):
        mk.keyblob.load(bytes(bad))
    with pytest.raises(ValueError):
        mk.keyblob.load(b"garbage" * 20)


def test_circuit_levelisation_with_oracle_backend():
    """the levelised scheduler (batched per level and gate type) against plaintext evaluation; CPU backend = oracle"""
    from mktfhe_amd import circuit as CI
    p = mk.CGGIparam.scaled(n=16, N=128)
    crs, keys = keygen(p, 71)
    so = oracle_scheme(p, crs, keys)
    circ = CI.ripple_adder(3)
    depth, sched = circ.levels()
    assert max(depth) == 5 and circ.n_inputs == 6          # xor/and, then 2 levels per further bit
    B = 4
    rng = np.random.default_rng(72)
    bits = rng.integers(0, 2, (6, B)).astype(bool)
    inputs = [np.stack([mk.lwe_encrypt(int(bits[i, j]), keys[0], p, deterministic_seed=7200 + 10 * i + j) for j in range(B)]) for i in range(6)]
    calls = []
    def gate_fn(op, x, y):
        calls.append((op, x.shape[0]))
        return so.gate_batch(op, x, y, threads=4)
    outs = CI.evaluate(circ, inputs, gate_fn, lambda x: (0 - x.astype(np.int64)).astype(np.uint32))
    want = circ.plain(bits)
    for o, w in zip(outs, want):
        assert np.array_equal(mk.lwe_decrypt(o, keys[0], p), w)
    a = sum(bits[i].astype(int) << i for i in range(3)); b = sum(bits[3 + i].astype(int) << i for i in range(3))
    got = sum(mk.lwe_decrypt(o, keys[0], p).astype(int) << i for i, o in enumerate(outs))
    assert np.array_equal(got, a + b)
    assert len(calls) == sum(len(v) for v in sched.values())   # one batched call per (level, gate type)
    # NOT is free and composes
    c2 = CI.Circuit(); x = c2.input(); y = c2.input(); c2.output(c2.NOT(c2.NAND(x, c2.NOT(y))))
    o2 = CI.evaluate(c2, inputs[:2], gate_fn, lambda v: (0 - v.astype(np.int64)).astype(np.uint32))
    assert np.array_equal(mk.lwe_decrypt(o2[0], keys[0], p), bits[0] & ~bits[1])
    # MUX = OR(AND(s, a), AND(NOT s, b)): two levels, the two ANDs in one batched call
    c3 = CI.Circuit(); s, a3, b3 = c3.input(), c3.input(), c3.input(); c3.output(c3.MUX(s, a3, b3))
    calls.clear()
    o3 = CI.evaluate(c3, inputs[:3], gate_fn, lambda v: (0 - v.astype(np.int64)).astype(np.uint32))
    assert np.array_equal(mk.lwe_decrypt(o3[0], keys[0], p), np.where(bits[0], bits[1], bits[2]))
    assert calls == [(1, 2 * B), (2, B)]


def test_circuit_plan_one_call_per_level_with_oracle_backend():
    """circuit.Plan / evaluate_on (one mkt_gate_batch_gather per LEVEL, NOTs folded into per-gate op codes) driven by a stand-in
    scheme whose gate_gather is the ORACLE: same ciphertext words as the per-(level, op) evaluator, on a circuit that mixes
    gate kinds in a level, chains NOTs and ends in a NOT"""
    from mktfhe_amd import circuit as CI
    p = mk.CGGIparam.scaled(n=16, N=128)
    crs, keys = keygen(p, 73)
    so = oracle_scheme(p, crs, keys)
    neg = lambda x: (0 - x.astype(np.int64)).astype(np.uint32)   # noqa: E731

    class OracleAsScheme:
        calls = []
        def gate_gather(self, ops, pool, ix, iy, out):
            self.calls.append(len(ops))
            for j in range(len(ops)):
                a = neg(pool[ix[j]]) if ops[j] & 8 else pool[ix[j]]
                b = neg(pool[iy[j]]) if ops[j] & 16 else pool[iy[j]]
                out[j] = so.gate(int(ops[j] & 7), a, b)
            return out
        def not_(self, x):
            x[...] = neg(x)
            return x
        def mux_gather(self, pool, js, ja, jb, out, not_ab=None):
            self.calls.append(len(js))
            for j in range(len(js)):
                a = neg(pool[ja[j]]) if not_ab is not None and not_ab[j] & 1 else pool[ja[j]]
                b = neg(pool[jb[j]]) if not_ab is not None and not_ab[j] & 2 else pool[jb[j]]
                out[j] = oracle_mux(pool[js[j]], a, b)
            return out

    def oracle_mux(sel, a, b):
        """the native MUX on the oracle's operators (no composite gate): two blindrotate! of the AND-linear parts, accumulators added,
        + 1/8 at X^0 of b, one keyswitch!"""
        accs = []
        for x, y in ((sel, a), (neg(sel), b)):
            at, bt = so.modswitch(O.gate_linear(1, x, y))
            accs.append(so.blindrotate(at, so.testvector(bt)).astype(np.uint64))
        acc = ((accs[0] + accs[1]) & np.uint64(0xFFFFFFFF)).reshape(-1, p.N)
        acc[0, 0] = np.uint64((int(acc[0, 0]) + (1 << 29)) & 0xFFFFFFFF)
        return so.keyswitch(acc)

    B = 3
    rng = np.random.default_rng(74)
    circ = CI.ripple_adder(2)
    c2 = CI.Circuit(); u, v, w = c2.input(), c2.input(), c2.input()
    c2.output(c2.NOT(c2.XOR(c2.NOT(c2.NOT(u)), c2.NOT(v)))); c2.output(c2.MUX(u, v, w))
    # native MUX nodes mixed with two-input gates in one level, a negated selector (= operands swapped) and a negated data operand
    c3 = CI.Circuit(); s3, a3, b3, d3 = (c3.input() for _ in range(4))
    m3 = c3.MUXN(c3.NOT(s3), a3, c3.NOT(b3)); x3 = c3.XOR(a3, d3)
    c3.output(c3.MUXN(m3, x3, d3)); c3.output(m3)
    mux_batch = lambda S, A, Bv: np.stack([oracle_mux(S[j], A[j], Bv[j]) for j in range(len(S))])   # noqa: E731
    for cc in (circ, c2, c3):
        bits = rng.integers(0, 2, (cc.n_inputs, B)).astype(bool)
        inputs = [np.stack([mk.lwe_encrypt(int(bits[i, j]), keys[0], p, deterministic_seed=7400 + 10 * i + j) for j in range(B)]) for i in range(cc.n_inputs)]
        fake = OracleAsScheme(); fake.calls = []
        plan = CI.Plan(cc, B)
        outs = CI.evaluate_on(cc, inputs, fake, plan)
        depth, sched = cc.levels()
        ncalls = sum((1 if l[1] else 0) + (1 if m else 0) for l, m in zip(plan.levels, plan.mux_levels))
        assert len(fake.calls) == ncalls and max(depth) <= ncalls <= 2 * max(depth) and sum(fake.calls) == plan.gates
        if cc is not c3:
            assert ncalls == max(depth)                     # no native MUX nodes: one call per level
        ref = CI.evaluate(cc, inputs, lambda op, x, y: so.gate_batch(op, x, y, threads=4), neg, mux_fn=mux_batch)
        for o, r, wv in zip(outs, ref, cc.plain(bits)):
            assert np.array_equal(o, r)
            assert np.array_equal(mk.lwe_decrypt(o, keys[0], p), wv)


def test_lds_staging_swizzle_is_conflict_free_in_the_bank_model():
    """the XOR swizzle shipped in csrc/fft_device.h (lds_pos) has zero bank conflicts for every exchange pattern of the
    4-points-per-thread schedule in the gfx950 b128 bank model (tools/lds_swizzle_search.py); PMC confirms on hardware"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lds_swizzle_search as S
    S.LOGR = 2
    swz = lambda i: i ^ ((i >> 1) & 8) ^ ((i >> 2) & 15)
    for LOGM in (7, 8, 9, 10, 11):
        M = 1 << LOGM
        assert sorted(swz(i) for i in range(M)) == list(range(M))          # a permutation of the staging slots
        rd, ird, wr, iwr = S.evaluate(LOGM, swz)
        assert (rd, wr) == (ird, iwr), LOGM
        rd, ird, wr, iwr = S.evaluate(LOGM, swz, S.schedule(LOGM))         # the shipped window sequence (odd LOGM: ..5,4,2,0)
        assert (rd, wr) == (ird, iwr), ("shipped schedule", LOGM)
        rd0, _, wr0, _ = S.evaluate(LOGM, lambda i: i)
        assert rd0 > ird and wr0 > iwr                                     # the identity layout does conflict
