"""MKT_ARITH_EXACT on the Float64 pipe (mktfhe_amd/csrc/fx_exact.hip; option exact_impl): exact products from FMA complex
transforms over centered 16-bit key limbs.  EXACT owes the reference no rounding sequence, only the exact product its transform
approximates (/root/reference/src/ring/polynomial.jl:99-113, README.md:9), so the checker is the big-integer restatement
(tests/ref_exact.py, oracle negacyclic products) -- and the integer-NTT kernels, which must give the SAME words."""
import numpy as np
import pytest

from helpers import O, mk, keygen, oracle_scheme, encrypt_bits, GATE_FUNCS
from test_gpu_parity import edge_words

pytestmark = pytest.mark.gpu


def _fx(p):
    sx = mk.Scheme(p, arith=mk.ARITH_EXACT)
    sx.set_option("exact_impl", 1)
    return sx


@pytest.mark.parametrize("N,W", [(128, 32), (256, 64), (512, 32), (1024, 32), (1024, 64), (2048, 64), (4096, 32), (4096, 64)])
def test_fx_polymul_is_exact(require_gpu, N, W):
    """digit polynomial (balanced digits of every gadget base up to 2^16) x ring polynomial (edge words in both halves) == the schoolbook
    product mod 2^W, word for word; the largest rounding distance met stays far below the 1/4 the call certifies itself with"""
    rng = np.random.default_rng(N + W)
    p = mk.CGGIparam.scaled(n=8, N=N, W=W)
    ex = _fx(p)
    B = 4
    polys = np.stack([edge_words(W, N, rng) for _ in range(B)]).astype(p.ring_dtype)
    worst = 0.0
    for logB in (2, 9, 12, 16):
        a = rng.integers(-(1 << (logB - 1)), 1 << (logB - 1), (B, N)).astype(np.int64)
        a[0, :4] = [-(1 << (logB - 1)), (1 << (logB - 1)) - 1, 0, -1]
        aw = a.astype(np.uint64).astype(p.ring_dtype) if W == 64 else (a & 0xFFFFFFFF).astype(np.uint32)
        got = ex.exact_polymul(aw, polys)
        worst = max(worst, ex.get_metric("fx_last_resid"))
        for b in range(B):
            ref = O.negacyclic(aw[b].astype(np.uint64) & np.uint64((1 << W) - 1), polys[b].astype(np.uint64), W)
            assert np.array_equal(got[b].astype(np.uint64), ref), (N, W, logB, b)
    assert worst < 2.0 ** -6, worst
    # the same call on the integer NTT: the same words
    ex.set_option("exact_impl", 0)
    assert np.array_equal(ex.exact_polymul(aw, polys), got)
    ex.close()


@pytest.mark.parametrize("N", [1024, 4096])
def test_fx_products_at_the_bound(require_gpu, N):
    """Adversarial operands: EVERY digit at +-2^15 (the largest gadget base, 2^16) and EVERY key limb at +-2^15 -- same signs throughout
    (the largest coefficient any product can have, N 2^30, and the largest 2-norm of what the inverse transforms), alternating signs,
    and random signs.  Exact, and the measured rounding distance is reported against the 1/4 certificate."""
    p = mk.CGGIparam.scaled(n=8, N=N, W=64)
    ex = _fx(p)
    rng = np.random.default_rng(N)
    lim = 0x8000800080008000                                   # every 16-bit field 0x8000: centered limbs -2^15 (with the carries: -2^15 + 1 above the lowest)
    pat = {"same": (np.full(N, -(1 << 15)), np.full(N, lim, dtype=np.uint64)),
           "alt": (np.where(np.arange(N) & 1, -(1 << 15), (1 << 15) - 1), np.where(np.arange(N) & 1, lim, 0x7FFF7FFF7FFF7FFF).astype(np.uint64)),
           "rnd": (rng.choice([-(1 << 15), (1 << 15) - 1], N), rng.choice(np.array([lim, 0x7FFF7FFF7FFF7FFF], dtype=np.uint64), N))}
    for name, (a, b) in pat.items():
        aw = a.astype(np.int64).astype(np.uint64)[None]
        bw = b.astype(np.uint64)[None]
        got = ex.exact_polymul(aw, bw)
        assert np.array_equal(got[0].astype(np.uint64), O.negacyclic(aw[0], bw[0], 64)), name
        assert ex.get_metric("fx_last_resid") < 2.0 ** -5, (name, ex.get_metric("fx_last_resid"))
    ex.close()


FX_SETS = [mk.CGGIparam.scaled(n=12, N=256), mk.CGGIparam.scaled(n=10, N=1024), mk.CGGI_N1024_l2.scaled(n=10), mk.CGGIparam.scaled(n=6, N=2048),
           mk.CGGIparam.scaled(n=6, N=4096), mk.CGGIparam.scaled(n=8, N=128, l_gsw=2, logB_gsw=10)]


@pytest.mark.parametrize("p", FX_SETS, ids=lambda p: f"{p.name}-n{p.n}-N{p.N}-l{p.l_gsw}")
def test_fx_cggi_gates(require_gpu, p):
    """CGGI (bootstrapping.jl:32-76) with exact products on the Float64 pipe: accumulators and gate outputs equal the big-integer
    restatement AND the integer-NTT kernel word for word; the keys are certified (proven bound below 1/2)."""
    import ref_exact as RX
    crs, keys = keygen(p, 171)
    so = oracle_scheme(p, crs, keys)
    sx = _fx(p)
    sx.load_party(0, keys[0])
    assert sx.get_metric("fx_available") == 1.0 and 0.0 < sx.get_metric("fx_bound") < 0.45
    B = 4
    bits = np.array([1, 0, 1, 1, 0, 1, 0, 0], dtype=bool)
    c = encrypt_bits(p, keys, bits, seed=17100)
    x, y = c[:B], c[B:]
    lin = np.stack([O.gate_linear(0, x[j], y[j]) for j in range(B)])
    at, bt = sx.modswitch(lin)
    at[0, :3] = [0, 2 * p.N, p.N]
    acc0 = np.stack([so.testvector(bt[j]) for j in range(B)])
    acc_x = sx.blindrotate_(at, acc0.astype(np.uint32).copy())
    assert sx.last_kernel_name() == "fx_blindrotate_kernel"
    for j in range(B):
        assert np.array_equal(acc_x[j].astype(np.uint64).reshape(-1), RX.blindrotate(p, keys[0].brk, at[j], acc0[j])), f"fx blindrotate {j}"
    sx.set_option("exact_impl", 0)
    assert np.array_equal(sx.blindrotate_(at, acc0.astype(np.uint32).copy()), acc_x) and sx.last_kernel_name() == "exact_blindrotate_kernel"
    sx.set_option("exact_impl", 1)
    for op in (0, 3, 5):
        out = sx.gate(op, x, y)
        assert np.array_equal(out, np.stack([RX.gate(p, so, keys[0].brk, op, x[j], y[j]) for j in range(B)])), f"fx gate {op}"
        assert np.array_equal(mk.lwe_decrypt(out, keys[0], p), GATE_FUNCS[op](bits[:B], bits[B:]))
    # keys generated on the device: the limb transforms are made from the generated words as well
    sd = _fx(p)
    sd.keygen_device(0, keys[0])
    assert np.array_equal(sd.gate(0, x, y), sx.gate(0, x, y)) and sd.last_kernel_name() == "fx_blindrotate_kernel"
    sd.close(); sx.close()


FX_KMS = [mk.KMS2party.scaled(n=8, N=256), mk.KMS2party_N1024_l2.scaled(n=8), mk.KMS2party.scaled(n=6), mk.KMS2party_N1024_l2.scaled(n=6, N=512),
          mk.KMS2party.scaled(n=4, N=4096), mk.KMS8party.scaled(n=3, N=256, k=3, l_gsw=3, logB_gsw=12)]


@pytest.mark.parametrize("p", FX_KMS, ids=lambda p: f"{p.name}-n{p.n}-N{p.N}-k{p.k}")
def test_fx_kms_gates(require_gpu, p):
    """KMS (bootstrapping.jl:369-594): phase 1 on the Float64 pipe (fx_blindrotate_kernel: one rotation per RLEV row, rows handed to the integer
    phase 2 as split residue tables).  Accumulators after the whole blind rotation and gate outputs equal the big-integer restatement and the
    integer-NTT phase 1 word for word, on inputs that involve every party; ragged batches (odd rotation counts)."""
    import ref_exact as RX
    crs, keys = keygen(p, 173)
    so = oracle_scheme(p, crs, keys)
    sx = _fx(p)
    sx.load_crs(crs)
    for i, kk in enumerate(keys):
        sx.load_party(i, kk)
    assert sx.get_metric("fx_available") == 1.0 and 0.0 < sx.get_metric("fx_bound") < 0.45
    k, B = p.k, 3
    rng = np.random.default_rng(174)
    bits = rng.integers(0, 2, 2 * B).astype(bool)

    def allp(j):
        ct = mk.lwe_ith_encrypt(int(bits[j]), 0, keys[0], p, deterministic_seed=18000 + 100 * j).astype(np.uint32)
        for i in range(1, k):
            for m in (0, 1):
                ct = ct + mk.lwe_ith_encrypt(m, i, keys[i], p, deterministic_seed=18000 + 100 * j + 2 * i + m).astype(np.uint32)
        return ct
    x = np.stack([allp(j) for j in range(B)]); y = np.stack([allp(B + j) for j in range(B)])
    lin = np.stack([O.gate_linear(0, x[j], y[j]) for j in range(B)])
    at, bt = sx.modswitch(lin)
    at[0, :2] = [0, 2 * p.N]
    acc0 = np.stack([so.testvector(bt[j]) for j in range(B)])
    acc_x = sx.blindrotate_(at, acc0.astype(np.uint64).copy())
    assert sx.last_kernel_name() == "fx_blindrotate_kernel"
    for j in range(B):
        assert np.array_equal(acc_x[j], RX.kms_blindrotate(p, keys, crs, at[j], acc0[j])), f"fx KMS blind rotation {j}"
    lev_fx = sx.kms_phase1(at[:, :])
    sx.set_option("exact_impl", 0)
    assert np.array_equal(sx.blindrotate_(at, acc0.astype(np.uint64).copy()), acc_x) and sx.last_kernel_name().startswith("exact_kms_phase1")
    assert np.array_equal(sx.kms_phase1(at[:, :]).view(np.uint64), lev_fx.view(np.uint64))      # the rows themselves, as split residue tables
    sx.set_option("exact_impl", 1)
    for op in (0, 3):
        out = sx.gate(op, x[:2], y[:2])
        assert np.array_equal(out, np.stack([RX.kms_gate(p, so, keys, crs, op, x[j], y[j]) for j in range(2)])), f"fx KMS gate {op}"
        assert np.array_equal(mk.lwe_decrypt(out, keys, p), GATE_FUNCS[op](bits[:2], bits[B:B + 2]))
    sx.close()


def test_fx_refuses_uncertified_keys(require_gpu):
    """The bound takes the loaded keys' largest transform-domain magnitude.  A key whose limbs all share one sign (nothing a key
    generator produces: magnitude N 2^15 in one transform point) fails it at the headline gadget (base 2^16): the context keeps
    serving -- on the integer NTT -- and the words are still the big-integer ones."""
    import types
    import ref_exact as RX
    p = mk.KMS2party_N1024_l2.scaled(n=4)
    crs, keys = keygen(p, 175)
    so = oracle_scheme(p, crs, keys)
    sy = _fx(p)                                                 # honest keys of this shape: certified
    sy.load_crs(crs)
    for i, kk in enumerate(keys):
        sy.load_party(i, kk)
    assert sy.get_metric("fx_available") == 1.0 and sy.get_metric("fx_kmax") < 64 * 32 * 32768
    sy.close()
    sx = _fx(p)
    sx.load_crs(crs)
    bad = np.full_like(np.asarray(keys[1].brk), 0x7FFF7FFF7FFF7FFF)
    sx.load_party(0, keys[0])
    sx.load_party(1, brk=bad, ksk=keys[1].ksk, rlk_d=keys[1].rlk_d, rlk_f=keys[1].rlk_f, pubkey=keys[1].pubkey)
    assert sx.get_metric("fx_kmax") > 0.6 * 1024 * 32767 and sx.get_metric("fx_bound") > 0.45 and sx.get_metric("fx_available") == 0.0
    bits = np.array([1, 0, 1, 1], dtype=bool)
    c = encrypt_bits(p, keys, bits, seed=17500)
    lin = np.stack([O.gate_linear(0, c[j], c[2 + j]) for j in range(2)])
    at, bt = sx.modswitch(lin)
    acc0 = np.stack([so.testvector(bt[j]) for j in range(2)])
    acc_x = sx.blindrotate_(at, acc0.astype(np.uint64).copy())
    assert sx.last_kernel_name().startswith("exact_kms_phase1")
    kb = [keys[0], types.SimpleNamespace(brk=bad, rlk_d=keys[1].rlk_d, rlk_f=keys[1].rlk_f, pubkey=keys[1].pubkey)]
    for j in range(2):
        assert np.array_equal(acc_x[j], RX.kms_blindrotate(p, kb, crs, at[j], acc0[j]))
    sx.close()
