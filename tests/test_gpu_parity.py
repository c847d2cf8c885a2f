"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Integer outputs (ciphertext words, accumulators, digits) and the Float64 transform-domain values
are compared as raw bits -- the F64REF mode reproduces the reference's operation sequence, so
equality is exact, tolerance 0.
"""
import os

import numpy as np
import pytest

from helpers import (GATE_FUNCS, O, encrypt_bits, gpu_scheme, keygen, mk, oracle_scheme)

pytestmark = pytest.mark.gpu


def bits_equal(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def edge_words(W, n, rng):
    m = (1 << W) - 1
    special = [0, 1, 2, m, m - 1, 1 << (W - 1), (1 << (W - 1)) - 1, (1 << (W - 1)) + 1, 1 << (W - 3), m - (1 << (W - 3)) + 1]
    v = rng.integers(0, 1 << 63, n, dtype=np.uint64) * 2 + rng.integers(0, 2, n, dtype=np.uint64)
    v &= np.uint64(m)
    v[: len(special)] = np.array(special, dtype=np.uint64)
    # the same words in the UPPER half of the polynomial: coefficient i + M feeds the imaginary slot through
    # -signed(p[i + M]) (fft.jl:60), where typemin wraps to itself
    if n >= 4 * len(special):
        v[n // 2: n // 2 + len(special)] = np.array(special, dtype=np.uint64)
        v[n - len(special):] = np.array(special[::-1], dtype=np.uint64)
    return v


@pytest.fixture(scope="module")
def golden_tw():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "twiddles.npz"))


@pytest.mark.parametrize("N,W", [(256, 32), (1024, 32), (1024, 64), (2048, 64), (4096, 64)])
def test_twiddles_and_monomial(require_gpu, golden_tw, N, W):
    p = mk.CGGIparam.scaled(n=8, N=N, W=W)
    s = mk.Scheme(p)
    for w, name in enumerate(("psi", "psiinv", "roots", "rootsinv")):
        assert bits_equal(s.twiddles(w), golden_tw[f"{name}_{N}"]), (name, N)
    f = O.Ffter(N, W)
    for e in (1, 2, N // 2, N - 1, N, N + 1, 2 * N - 1, 2 * N):
        assert bits_equal(s.monomial(e), f.monomial(e)), e
    s.close()


@pytest.mark.parametrize("N,W", [(64, 32), (256, 64), (512, 32), (1024, 32), (1024, 64), (2048, 64), (2048, 32), (4096, 64)])
def test_transform_bitexact(require_gpu, N, W):
    rng = np.random.default_rng(N + W)
    p = mk.CGGIparam.scaled(n=8, N=N, W=W)
    s = mk.Scheme(p)
    f = O.Ffter(N, W)
    B = 37
    polys = np.stack([edge_words(W, N, rng) for _ in range(B)])
    # small balanced digits too (the in-loop distribution)
    polys[1] = (rng.integers(-256, 256, N).astype(np.int64).astype(np.uint64)) & np.uint64((1 << W) - 1)
    polys[2] = 0
    t_gpu = s.transform_fwd(polys.astype(p.ring_dtype))
    t_ref = f.fwd(polys)
    assert bits_equal(t_gpu, t_ref)
    # inverse on products of transforms (realistic magnitudes) and on raw transforms
    prod = t_ref * 0  # built with oracle arithmetic to stay bit-defined
    for b in range(B):
        acc = np.zeros(N // 2, dtype=np.complex128)
        O.lib().ora_tp_muladd(acc.ctypes.data, t_ref[b].ctypes.data, t_ref[(b + 1) % B].ctypes.data, N // 2)
        prod[b] = acc
    for src in (t_ref, prod):
        p_gpu = s.transform_inv(src)
        p_ref = f.inv(src)
        assert np.array_equal(p_gpu.astype(np.uint64), p_ref)
    s.close()


@pytest.mark.parametrize("W,l,logB", [(32, 3, 9), (64, 3, 12), (64, 2, 7), (64, 7, 6), (32, 12, 2), (64, 16, 2), (64, 8, 8), (32, 4, 8)])
def test_decompose(require_gpu, W, l, logB):
    rng = np.random.default_rng(l * 100 + logB)
    N = 256
    p = mk.CGGIparam.scaled(n=8, N=N, W=W)
    s = mk.Scheme(p)
    polys = np.stack([edge_words(W, N, rng) for _ in range(5)])
    d_gpu = s.decompose(polys.astype(p.ring_dtype), l, logB)
    for b in range(5):
        assert np.array_equal(d_gpu[b].astype(np.uint64), O.decomp_poly(polys[b], l, logB, W))
    s.close()


SMALL = [
    mk.CGGIparam.scaled(n=20, N=256),
    mk.CGGIparam.scaled(n=12, N=1024, l_gsw=2, logB_gsw=10),
    mk.Blockparam.scaled(n=30, N=256, blk_d=10),
    mk.KMS2party.scaled(n=16, N=256),
    mk.KMS4party.scaled(n=10, N=256),
    mk.KMS2partyblock.scaled(n=24, N=256, blk_d=8),
    mk.KMS2party_N1024_l2.scaled(n=12),
    mk.CCS2party.scaled(n=12, N=256),
    mk.CCS4party.scaled(n=6, N=256),
    mk.CCS8party.scaled(n=4, N=512),
    mk.CGGIparam.scaled(n=20, N=256, f=5, logD=3),          # other key-switch gadgets
    mk.Blockparam.scaled(n=30, N=256, blk_d=10, f=4, logD=3),
    mk.KMS8party.scaled(n=4, N=256),                        # larger party counts / gadget shapes (params.jl:63-85)
    mk.CCS16party.scaled(n=2, N=256),
    mk.CGGIparam.scaled(n=16, N=256, k=2),                  # RLWE length > 1 (TFHEparams_bin.k, scheme.jl:6-20)
    mk.CGGIparam.scaled(n=12, N=512, k=3, l_gsw=2, logB_gsw=10),
    mk.Blockparam.scaled(n=30, N=256, blk_d=10, k=2),       # LMSS with RLWE length > 1 (TFHEparams_block.k)
    mk.Blockparam.scaled(n=300, N=128, blk_d=100, k=3),     # n > N: the key switch copies whole components (:179-186)
    mk.Blockparam.scaled(n=150, N=128, blk_d=50, k=2, blk_len=3),
    mk.Blockparam_k2.scaled(n=24, blk_d=8),                 # BASELINE configs[4] shape (N = 1024, k = 2, block length 3) at reduced n
    mk.Blockparam_k2.scaled(n=24, blk_d=12, blk_len=2),     # other block lengths take the generic path
    mk.CGGIparam.scaled(n=10, N=256, k=4),                  # RLWE length > 3: the run-time-k kernel (accumulators in memory)
    mk.CGGIparam.scaled(n=8, N=128, k=6, l_gsw=2, logB_gsw=10),
    mk.Blockparam.scaled(n=24, N=256, blk_d=8, k=4),
    mk.Blockparam.scaled(n=20, N=128, blk_d=10, k=5, blk_len=2),
    # N = 4096 (M = 2048: the largest transform the header admits; the reference leaves N free, scheme.jl:6-20): every stage and gate, all five schemes
    mk.CGGIparam.scaled(n=6, N=4096),
    mk.KMS2party.scaled(n=4, N=4096),
    mk.Blockparam.scaled(n=12, N=4096, blk_d=4),
    mk.CCS2party.scaled(n=4, N=4096),
    mk.KMS2partyblock.scaled(n=6, N=4096, blk_d=2),
]


def _stage_check(p, B=6, seed=1):
    crs, keys = keygen(p, seed)
    so = oracle_scheme(p, crs, keys)
    sg = gpu_scheme(p, crs, keys)
    rng = np.random.default_rng(seed + 7)
    bits = rng.integers(0, 2, 2 * B).astype(bool)
    c = encrypt_bits(p, keys, bits, seed=500)
    x, y = c[:B], c[B:]
    for op in range(6):
        lin = np.stack([O.gate_linear(op, x[j], y[j]) for j in range(B)])
        # stage by stage
        at_g, bt_g = sg.modswitch(lin)
        for j in range(B):
            at_o, bt_o = so.modswitch(lin[j])
            assert np.array_equal(at_g[j], at_o) and bt_g[j] == bt_o
        if op == 0:
            acc0 = np.stack([so.testvector(bt_g[j]) for j in range(B)])
            acc_o = np.stack([so.blindrotate(at_g[j], acc0[j]) for j in range(B)])
            acc_g = sg.blindrotate_(at_g, acc0.astype(p.ring_dtype).copy())
            assert np.array_equal(acc_g.astype(np.uint64), acc_o), "blindrotate"
            if p.scheme in (mk.KMS, mk.KMS_BLOCK):
                lev_g = sg.kms_phase1(at_g)
                for j in range(2):
                    row = 0
                    for party in range(p.k):
                        lev_o = so.kms_phase1(party, at_g[j, party * p.n:(party + 1) * p.n])
                        assert bits_equal(lev_g[j, row:row + lev_o.shape[0]], lev_o), ("phase1", j, party)
                        row += lev_o.shape[0]
            ks_o = np.stack([so.keyswitch(acc_o[j]) for j in range(B)])
            ks_g = sg.keyswitch(acc_o.astype(p.ring_dtype))
            assert np.array_equal(ks_g, ks_o), "keyswitch"
        out_g = sg.gate(op, x, y)
        out_o = np.stack([so.gate(op, x[j], y[j]) for j in range(B)])
        assert np.array_equal(out_g, out_o), f"gate {op}"
        want = GATE_FUNCS[op](bits[:B], bits[B:])
        got = mk.lwe_decrypt(out_g, keys if p.multikey else keys[0], p)
        assert np.array_equal(got, want), f"decrypt {op}"
    # in-place bootstrap
    z = x.copy()
    sg.bootstrapping_(z)
    assert np.array_equal(z, np.stack([so.bootstrap(x[j]) for j in range(B)]))
    if p.multikey:
        _mixed_party_check(p, keys, so, sg, rng)
    sg.close()


def _mixed_party_check(p, keys, so, sg, rng, B=3):
    """Fresh encryptions populate one party's mask block and same-party gates keep it so (the other parties' rotations
    are all skips, bootstrapping.jl:413 / :261).  Here every ciphertext involves ALL k parties: NAND folds over one
    fresh encryption per party (as test/KMS.jl:29-34), then every stage and gate on those dense ciphertexts."""
    k = p.nparty
    bits = rng.integers(0, 2, 2 * B * k).astype(bool)
    c = encrypt_bits(p, keys, bits, seed=7700)                      # ciphertext j under party j mod k
    acc_g, acc_b = c[0::k].copy(), bits[0::k].copy()
    for i in range(1, k):
        nxt = c[i::k]
        ref = np.stack([so.gate(0, acc_g[j], nxt[j]) for j in range(2 * B)])
        acc_g = sg.gate(0, acc_g, nxt)
        assert np.array_equal(acc_g, ref), f"fold step {i}"
        acc_b = ~(acc_b & bits[i::k])
    assert (acc_g[:, :-1].reshape(2 * B, k, p.n) != 0).any(axis=2).all(), "every party block populated"
    x, y = acc_g[:B], acc_g[B:]
    lin = np.stack([O.gate_linear(0, x[j], y[j]) for j in range(B)])
    at_g, bt_g = sg.modswitch(lin)
    acc0 = np.stack([so.testvector(bt_g[j]) for j in range(B)])
    acc_o = np.stack([so.blindrotate(at_g[j], acc0[j]) for j in range(B)])
    assert np.array_equal(sg.blindrotate_(at_g, acc0.astype(p.ring_dtype).copy()).astype(np.uint64), acc_o), "blindrotate (mixed)"
    assert np.array_equal(sg.keyswitch(acc_o.astype(p.ring_dtype)), np.stack([so.keyswitch(acc_o[j]) for j in range(B)])), "keyswitch (mixed)"
    for op in range(6):
        out_g = sg.gate(op, x, y)
        assert np.array_equal(out_g, np.stack([so.gate(op, x[j], y[j]) for j in range(B)])), f"gate {op} (mixed)"
        if p.name not in NOISY:
            got = mk.lwe_decrypt(out_g, keys, p)
            assert np.array_equal(got, GATE_FUNCS[op](acc_b[:B], acc_b[B:])), f"decrypt {op} (mixed)"


# parameter sets whose own noise makes gates on many-party ciphertexts decrypt wrongly now and then (the oracle
# produces the identical words): bit parity is asserted for them, decryption is not
NOISY = {"CCS16party", "CCS8party", "CCS4party", "KMS8party"}


@pytest.mark.parametrize("p", SMALL, ids=lambda p: f"{p.name}-n{p.n}-N{p.N}")
def test_stages_small(require_gpu, p):
    _stage_check(p)


# The block-binary rotation has two kernels: one rotation per workgroup (blindrotate_k1_kernel<LB>) and G = 2 / 4
# rotations of one slot per workgroup (rot_block.hip: digit transforms shared through LDS, each key element loaded once
# per G rotations).  MKT_ROT_BLKG forces the grouping; every stage and gate must give the oracle's words under each, at
# batch sizes that leave ragged workgroups (B = 5, and the 3- / 6-ciphertext batches inside _stage_check).
BLOCK_SETS = [
    mk.Blockparam.scaled(n=30, N=256, blk_d=10),
    mk.Blockparam.scaled(n=24, N=1024, blk_d=8),                       # the shipped shape (M = 512, l = 3, 2^9) at reduced n
    mk.Blockparam.scaled(n=24, N=64, blk_d=12, blk_len=2),             # thread groups smaller than a wave
    mk.Blockparam.scaled(n=24, N=512, blk_d=6, blk_len=4),
    mk.KMS2partyblock.scaled(n=24, N=256, blk_d=8),
    mk.KMS2partyblock.scaled(n=12, N=2048, blk_d=4),                   # the shipped 64-bit shape (M = 1024, l = 3, 2^12)
    mk.KMS4partyblock.scaled(n=6, N=512, blk_d=2),
    mk.Blockparam_k2.scaled(n=12, N=1024, blk_d=4),                     # RLWE length 2 (BASELINE configs[4]): three accumulator polynomials, G = 4 only
    mk.Blockparam_k2.scaled(n=18, N=128, blk_d=6),
    mk.CGGIparam.scaled(n=16, N=256, k=2),                             # the plain CMux with RLWE length 2 / 3 on the same kernel (one key bit per block)
    mk.CGGIparam.scaled(n=8, N=1024, k=2),
    mk.CGGIparam.scaled(n=12, N=512, k=3, l_gsw=2, logB_gsw=10),
    mk.Blockparam.scaled(n=12, N=4096, blk_d=4),                       # M = 2048 under every forced grouping (the grouped kernels fall back where LDS does not admit them)
    mk.KMS2partyblock.scaled(n=6, N=4096, blk_d=2),
    mk.Blockparam_k2.scaled(n=6, N=4096, blk_d=2),
]


@pytest.mark.parametrize("G", ["1", "2", "4"])
@pytest.mark.parametrize("p", BLOCK_SETS, ids=lambda p: f"{p.name}-n{p.n}-N{p.N}-b{p.blk_len}")
def test_block_rotation_groupings_are_bit_identical(require_gpu, p, G, monkeypatch):
    monkeypatch.setenv("MKT_ROT_BLKG", G)
    _stage_check(p, B=5, seed=3)


# CCS has two blind-rotation kernels: one thread group per ciphertext, and -- for batches that leave compute units idle -- two
# thread groups per ciphertext as a two-stage pipeline over a step's input polynomials (ccs_pipe.hip: the ordered Float64 sums
# into tacc.b / tacc.a[idx] stay one chain in one group).  MKT_CCS_PIPE forces either; every stage and gate must give the oracle's words.
CCS_SETS = [mk.CCS2party.scaled(n=12, N=256), mk.CCS2party.scaled(n=8, N=1024), mk.CCS4party.scaled(n=6, N=512), mk.CCS8party.scaled(n=4, N=512),
            mk.CCS8party.scaled(n=3, N=2048, k=3), mk.CCS16party.scaled(n=2, N=128, k=5), mk.CCS2party.scaled(n=4, N=4096), mk.CCS4party.scaled(n=2, N=4096, k=3)]


@pytest.mark.parametrize("pipe", ["0", "1"])
@pytest.mark.parametrize("p", CCS_SETS, ids=lambda p: f"{p.name}-n{p.n}-N{p.N}-k{p.k}")
def test_ccs_rotation_kernels_are_bit_identical(require_gpu, p, pipe, monkeypatch):
    monkeypatch.setenv("MKT_CCS_PIPE", pipe)
    _stage_check(p, B=5, seed=4)


FULL = [mk.CGGIparam, mk.CGGI_N1024_l2, mk.Blockparam, mk.Blockparam_k2, mk.KMS2party, mk.KMS2partyblock, mk.KMS2party_N1024_l2, mk.CCS2party]


@pytest.mark.parametrize("p", FULL, ids=lambda p: p.name)
def test_gate_full_size(require_gpu, p):
    """reference parameter sets: GPU NAND / XOR batch == oracle, and decrypts"""
    crs, keys = keygen(p, 2)
    so = oracle_scheme(p, crs, keys)
    sg = gpu_scheme(p, crs, keys)
    B = 4
    rng = np.random.default_rng(11)
    bits = rng.integers(0, 2, 2 * B + 1).astype(bool)
    c = encrypt_bits(p, keys, bits, seed=900)
    x, y, bx, by = c[:B], c[B + 1:], bits[:B], bits[B + 1:]           # y_j under party (j + B + 1) mod k: cross-party pairs
    for op in (0, 3):
        out_g = sg.gate(op, x, y)
        out_o = so.gate_batch(op, x, y, threads=4)
        assert np.array_equal(out_g, out_o)
        got = mk.lwe_decrypt(out_g, keys if p.multikey else keys[0], p)
        assert np.array_equal(got, GATE_FUNCS[op](bx, by))
    # second level: gate outputs (every party block populated) as inputs
    lvl1 = sg.gate(0, x, y)
    out2 = sg.gate(0, lvl1[:2], lvl1[2:])
    assert np.array_equal(out2, so.gate_batch(0, lvl1[:2], lvl1[2:], threads=2))
    b1 = ~(bx & by)
    assert np.array_equal(mk.lwe_decrypt(out2, keys if p.multikey else keys[0], p), ~(b1[:2] & b1[2:]))
    sg.close()


def test_kat_fixture_replayed_without_the_oracle(require_gpu):
    """tests/golden/kat_tiny.npz (inputs + expected outputs, all five schemes): the engine, keyed from the fixture's
    integer keys alone, reproduces every gate output, the mod-switch and the accumulator after the blind rotation.
    Nothing of oracle/ is called here."""
    from helpers import kat_cases, kat_decrypt
    for name, p, d, keys in kat_cases():
        sg = mk.Scheme(p)
        if p.multikey:
            sg.load_crs(d["crs"])
        for i, kk in enumerate(keys):
            sg.load_party(i, brk=kk.brk, ksk=kk.ksk, rlk_d=kk.rlk_d, rlk_f=kk.rlk_f, pubkey=kk.pubkey)
        for op in range(6):
            out = sg.gate(op, d["x"], d["y"])
            assert np.array_equal(out, d["out"][op]), (name, op)
            assert np.array_equal(kat_decrypt(p, d, out), GATE_FUNCS[op](d["bits"][:4], d["bits"][4:]))
        lin = (np.uint32(1 << 29) * (np.arange(p.lwe_len) == p.lwe_len - 1).astype(np.uint32) - d["x"] - d["y"]).astype(np.uint32)   # gate.jl:1-8
        at, bt = sg.modswitch(lin)
        assert np.array_equal(at, d["atilde"]) and np.array_equal(bt, d["btilde"])
        N = p.N
        acc0 = np.zeros((4, 1 + p.k, N), dtype=p.ring_dtype)                     # bootstrapping.jl:11-23
        E = p.ring_dtype(1 << (p.W - 3))
        for j in range(4):
            b = int(bt[j]); lo, hi = (E, -E) if b <= N else (-E, E)
            b = b if b <= N else b - N
            acc0[j, 0] = np.where(np.arange(N) < b, lo, hi).astype(p.ring_dtype)
        acc = sg.blindrotate_(at, acc0.reshape(4, -1).copy())
        assert np.array_equal(acc.astype(np.uint64), d["acc"].reshape(4, -1)), (name, "blindrotate")
        sg.close()


def test_device_tensors_and_not(require_gpu):
    import torch
    p = mk.CGGIparam.scaled(n=20, N=256)
    crs, keys = keygen(p, 3)
    so = oracle_scheme(p, crs, keys)
    sg = gpu_scheme(p, crs, keys)
    c = encrypt_bits(p, keys, [1, 0, 1, 1, 0, 0], seed=40)
    xd = torch.from_numpy(c[:3].view(np.int32)).cuda()
    yd = torch.from_numpy(c[3:].view(np.int32)).cuda()
    out = mk.NAND(xd, yd, sg)
    torch.cuda.synchronize()
    sg.synchronize()
    ref = np.stack([so.gate(0, c[j], c[3 + j]) for j in range(3)])
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref)
    n1 = c[:2].copy()
    mk.NOT_(n1, sg)
    assert np.array_equal(n1, (0 - c[:2].astype(np.int64)).astype(np.uint32))
    sg.close()


def test_errors(require_gpu):
    p = mk.CGGIparam.scaled(n=20, N=256)
    s = mk.Scheme(p)
    x = np.zeros((2, p.lwe_len), dtype=np.uint32)
    with pytest.raises(mk.MktError):      # keys not loaded
        s.gate(0, x, x)
    with pytest.raises(ValueError):       # wrong length (reference: @assert)
        s.gate(0, x[:, :-1], x[:, :-1])
    # EXACT: a gadget whose product sums would not fit the two-prime modulus (P / 2 = 2^58.9998) is refused instead of computing
    # something else (64-bit ring; 32-bit ring, where the bound grows with the RLWE length)
    for pk in (mk.KMS2party.scaled(n=8, N=2048, l_gsw=2, logB_gsw=20), mk.CGGIparam.scaled(n=8, N=2048, k=4, l_gsw=2, logB_gsw=16)):
        ex = mk.Scheme(pk, arith=mk.ARITH_EXACT)
        xk = np.zeros((2, pk.lwe_len), dtype=np.uint32)
        with pytest.raises(mk.MktError, match="MKT_ARITH_EXACT evaluates gates for CGGI"):
            ex.gate(0, xk, xk)
        ex.close()
    s.close()


def test_reference_tables_and_fft_form_keys(require_gpu):
    """a caller may install the reference's own ffter tables (mkt_set_twiddles) and hand over keys already in
    transform form (MKT_FMT_F64_FFT, the reference's Trans* values): results are identical"""
    import ctypes as C
    from mktfhe_amd import _lib
    p = mk.KMS2party.scaled(n=12, N=256)
    crs, keys = keygen(p, 8)
    so = oracle_scheme(p, crs, keys)
    f = so.ffter
    sg = mk.Scheme(p)
    tabs = [np.ascontiguousarray(f.table(w)) for w in range(4)]
    _lib.check(_lib.lib().mkt_set_twiddles(sg.h, *[t.ctypes.data_as(C.c_void_p) for t in tabs]), sg.h)
    bad = tabs[1].copy(); bad[5] += 1e-9        # Psiinv must be conj(Psi)
    assert _lib.lib().mkt_set_twiddles(sg.h, tabs[0].ctypes.data_as(C.c_void_p), bad.ctypes.data_as(C.c_void_p),
                                       tabs[2].ctypes.data_as(C.c_void_p), tabs[3].ctypes.data_as(C.c_void_p)) < 0
    # ... and have the reference's shape in the first entries, Psi[1] = (eps, -1), Psi[2] = (c, -c), Psi[3] = (-c, -c): the first
    # two butterfly stages are written for it (fft_device.h, MKT_FFT_SPECIAL)
    odd, oddi = tabs[0].copy(), tabs[1].copy()
    odd[2] = complex(odd[2].real, odd[2].imag * (1 + 2.0**-52)); oddi[2] = np.conj(odd[2])
    assert _lib.lib().mkt_set_twiddles(sg.h, odd.ctypes.data_as(C.c_void_p), oddi.ctypes.data_as(C.c_void_p),
                                       tabs[2].ctypes.data_as(C.c_void_p), tabs[3].ctypes.data_as(C.c_void_p)) < 0
    _lib.check(_lib.lib().mkt_set_twiddles(sg.h, *[t.ctypes.data_as(C.c_void_p) for t in tabs]), sg.h)
    sg.load_crs(f.fwd(crs.astype(np.uint64)), fmt=mk.FMT_F64_FFT)
    for i, kk in enumerate(keys):
        sg.load_party(i, brk=f.fwd(kk.brk.astype(np.uint64).reshape(-1, p.N)), ksk=kk.ksk,
                      rlk_d=f.fwd(kk.rlk_d.astype(np.uint64).reshape(-1, p.N)), rlk_f=f.fwd(kk.rlk_f.astype(np.uint64).reshape(-1, p.N)),
                      pubkey=f.fwd(kk.pubkey.astype(np.uint64).reshape(-1, p.N)), fmt=mk.FMT_F64_FFT)
    c = encrypt_bits(p, keys, [1, 1, 0, 1, 0, 0], seed=60)
    out = sg.gate(0, c[:3], c[3:])
    assert np.array_equal(out, np.stack([so.gate(0, c[j], c[3 + j]) for j in range(3)]))
    sg.close()


@pytest.mark.parametrize("p", [mk.KMS2party_N1024_l2, mk.KMS2party, mk.CGGIparam, mk.CGGI_N1024_l2], ids=lambda p: p.name)
def test_full_batch_properties(require_gpu, p):
    """BASELINE.json batch size (1024 gates): every output decrypts to NAND of its inputs, a sub-batch equals the
    oracle bit for bit, results do not depend on how the batch is split or repeated (independence + determinism)"""
    crs, keys = keygen(p, 12)
    sg = gpu_scheme(p, crs, keys)
    B = 1024
    rng = np.random.default_rng(13)
    bits = rng.integers(0, 2, 2 * B).astype(bool)
    uniq = encrypt_bits(p, keys, bits[:128], seed=7000)
    idx = rng.integers(0, 128, 2 * B)
    idx[:128] = np.arange(128)
    c = uniq[idx]; bits = bits[:128][idx]
    x, y = c[:B], c[B:]
    out = sg.gate(0, x, y)
    got = mk.lwe_decrypt(out, keys if p.multikey else keys[0], p)
    want = ~(bits[:B] & bits[B:])
    if p is mk.KMS2party_N1024_l2:
        # BASELINE.json's synthetic shape is not a parameter set of the reference; its noise margin is thinner than the
        # shipped sets' (output phase error std 0.025 against a 0.125 margin, tools/param_noise_sweep2.py): an isolated
        # gate may decrypt wrongly, in the oracle exactly as on the GPU (the bit-exact comparison below is the parity
        # statement)
        assert (got == want).mean() > 0.995
    else:
        assert np.array_equal(got, want)
    assert np.array_equal(out, sg.gate(0, x, y))                       # deterministic (atomics are integer adds)
    halves = np.concatenate([sg.gate(0, x[:300], y[:300]), sg.gate(0, x[300:], y[300:])])
    assert np.array_equal(out, halves)                                 # independent of batching
    so = oracle_scheme(p, crs, keys)
    assert np.array_equal(out[:8], so.gate_batch(0, x[:8], y[:8], threads=8))
    # NOT(NAND) == AND after one more bootstrap level, on device tensors end to end
    z = out.copy(); mk.NOT_(z, sg); sg.bootstrapping_(z)
    ok2 = mk.lwe_decrypt(z, keys if p.multikey else keys[0], p) == ~want
    assert ok2.mean() > 0.99 if p is mk.KMS2party_N1024_l2 else ok2.all()
    sg.close()


def test_empty_and_single(require_gpu):
    p = mk.CGGIparam.scaled(n=20, N=256)
    crs, keys = keygen(p, 3)
    sg = gpu_scheme(p, crs, keys)
    e = np.zeros((0, p.lwe_len), dtype=np.uint32)
    assert sg.gate(0, e, e).shape == (0, p.lwe_len)
    c = encrypt_bits(p, keys, [1, 0], seed=5)
    one = sg.gate(2, c[0], c[1])                                       # un-batched 1-D ciphertexts
    assert one.shape == (p.lwe_len,) and mk.lwe_decrypt(one, keys[0], p) is True
    sg.close()


def test_keyblob_upload(require_gpu):
    """evaluation keys travel as flat versioned blobs: same outputs as direct upload"""
    p = mk.KMS2party.scaled(n=12, N=256)
    crs, keys = keygen(p, 9)
    sg = mk.Scheme(p)
    mk.keyblob.load_into(sg, mk.keyblob.dump_crs(p, crs))
    for kk in keys:
        mk.keyblob.load_into(sg, mk.keyblob.dump_party(kk))
    so = oracle_scheme(p, crs, keys)
    c = encrypt_bits(p, keys, [1, 0, 0, 1], seed=90)
    assert np.array_equal(sg.gate(5, c[:2], c[2:]), np.stack([so.gate(5, c[j], c[2 + j]) for j in range(2)]))
    with pytest.raises(ValueError):
        mk.keyblob.load_into(mk.Scheme(p.scaled(n=10)), mk.keyblob.dump_party(keys[0]))
    sg.close()


def test_circuit_on_device(require_gpu):
    """4-bit adder, 64 independent input sets, ciphertexts resident in HBM between levels"""
    import torch
    from mktfhe_amd import circuit as CI
    p = mk.KMS2party
    crs, keys = keygen(p, 14)
    sg = gpu_scheme(p, crs, keys)
    circ = CI.ripple_adder(4)
    B = 64
    rng = np.random.default_rng(15)
    bits = rng.integers(0, 2, (8, B)).astype(bool)
    pool = {(i, v): mk.lwe_ith_encrypt(v, i % 2, keys[i % 2], p, deterministic_seed=1500 + 2 * i + v) for i in range(8) for v in (0, 1)}
    inputs = [torch.from_numpy(np.stack([pool[(i, int(bits[i, j]))] for j in range(B)]).view(np.int32)).cuda() for i in range(8)]
    outs = CI.evaluate_on(circ, inputs, sg)
    torch.cuda.synchronize()
    got = sum(mk.lwe_decrypt(o.cpu().numpy().view(np.uint32), keys, p).astype(int) << i for i, o in enumerate(outs))
    a = sum(bits[i].astype(int) << i for i in range(4)); b = sum(bits[4 + i].astype(int) << i for i in range(4))
    assert np.array_equal(got, a + b)
    # the same circuit through the ORACLE backend on the first input sets: identical ciphertext words at every output
    so = oracle_scheme(p, crs, keys)
    nb = 6
    o_in = [t[:nb].cpu().numpy().view(np.uint32) for t in inputs]
    o_out = CI.evaluate(circ, o_in, lambda op, x, y: so.gate_batch(op, x, y, threads=16), lambda x: (0 - x.astype(np.int64)).astype(np.uint32))
    for og, oo in zip(outs, o_out):
        assert np.array_equal(og[:nb].cpu().numpy().view(np.uint32), oo)
    # MUX (composite of the reference's gates): selector under party 0, data under parties 1 / 0
    sel, d1, d0 = inputs[0], inputs[1], inputs[2]
    m = mk.MUX_composite(sel, d1, d0, sg)
    assert np.array_equal(mk.lwe_decrypt(m.cpu().numpy().view(np.uint32), keys, p), np.where(bits[0], bits[1], bits[2]))
    cm = CI.Circuit(); s_, a_, b_ = cm.input(), cm.input(), cm.input(); cm.output(cm.MUX(s_, a_, b_))
    (mo,) = CI.evaluate_on(cm, [sel, d1, d0], sg)
    assert np.array_equal(mo.cpu().numpy(), m.cpu().numpy())          # same gates, same order -> same words
    (mo_o,) = CI.evaluate(cm, [o_in[0], o_in[1], o_in[2]], lambda op, x, y: so.gate_batch(op, x, y, threads=16), lambda x: (0 - x.astype(np.int64)).astype(np.uint32))
    assert np.array_equal(mo[:nb].cpu().numpy().view(np.uint32), mo_o)  # MUX on the engine == MUX of oracle gates
    sg.close()


def test_c_example_through_the_abi(require_gpu, tmp_path):
    """examples/kms_nand.c: a plain C caller (gcc, no Python) of include/mktfhe.h"""
    import os, subprocess
    from helpers import ROOT
    exe = str(tmp_path / "kms_nand")
    lib = os.path.join(ROOT, "mktfhe_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "kms_nand.c"),
                           "-o", exe, "-L" + lib, "-lmktfhe_hip", "-Wl,-rpath," + lib])
    out = subprocess.run([exe, "24", "256"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
    # examples/multi_nand.c: the multi-shard evaluator (mkt_multi_*), per-gate ops and the native MUX from plain C; three
    # logical shards of the one device, a ragged split of 13 gates
    exe2 = str(tmp_path / "multi_nand")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "multi_nand.c"),
                           "-o", exe2, "-L" + lib, "-lmktfhe_hip", "-Wl,-rpath," + lib])
    out = subprocess.run([exe2, "24", "256", "3", "1"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok") and "gates [5, 9)" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("p", [mk.CGGIparam.scaled(n=24, N=256), mk.KMS2party.scaled(n=12, N=256), mk.Blockparam.scaled(n=30, N=256, blk_d=10),
                               mk.CCS2party.scaled(n=10, N=256)], ids=lambda p: p.name)
def test_blindrotate_special_exponents(require_gpu, p):
    """atilde = 0 is skipped (bootstrapping.jl:48), atilde = 2N multiplies by the zero table entry (scheme.jl:125),
    atilde = N is X^N - 1 = -2; whole blocks of zeros for the block schemes; btilde at 0 / N / N+1 / 2N"""
    crs, keys = keygen(p, 4)
    so = oracle_scheme(p, crs, keys)
    sg = gpu_scheme(p, crs, keys)
    N, nat = p.N, p.lwe_len - 1
    rng = np.random.default_rng(5)
    rows = []
    for pattern in range(6):
        at = rng.integers(0, 2 * N + 1, nat).astype(np.uint32)
        if pattern == 0: at[:] = 0
        if pattern == 1: at[::2] = 0; at[1::4] = 2 * N
        if pattern == 2: at[:] = 2 * N
        if pattern == 3: at[: nat // 2] = 0; at[nat // 2:] = N
        if pattern == 4: at[:6] = [0, 0, 0, 1, 0, 2 * N][: min(6, nat)]
        rows.append(at)
    at = np.stack(rows)
    acc0 = np.stack([so.testvector(bt) for bt in (0, 1, N, N + 1, 2 * N, 2 * N - 1)])
    ref = np.stack([so.blindrotate(at[j], acc0[j]) for j in range(6)])
    got = sg.blindrotate_(at, acc0.astype(p.ring_dtype).copy())
    assert np.array_equal(got.astype(np.uint64), ref)
    sg.close()


@pytest.mark.parametrize("kind", ["KMS", "CGGI", "LMSS", "CCS", "KMSblock"])
def test_fixture_replay_path(require_gpu, tmp_path, kind):
    """tools/replay_fixture.py replays fixtures in the format tools/dump_fixture.jl (Julia reference, unexecuted here)
    writes -- all five scheme kinds, with the mod-switch, the accumulator after blindrotate! and the key switch checked
    separately; the path is exercised with fixtures written by the oracle in the same format"""
    import subprocess, sys
    from helpers import ROOT
    d = str(tmp_path / "fx")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "replay_fixture.py"), "--make", d, kind])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "replay_fixture.py"), d], capture_output=True, text=True)
    assert out.returncode == 0 and "False" not in out.stdout and out.stdout.count("True") >= 4, out.stdout + out.stderr


@pytest.mark.parametrize("p,B", [(mk.KMS4party, 8192 + 17), (mk.CCS8party, 1024), (mk.CCS8party_N2048, 1024), (mk.Blockparam, 16384), (mk.Blockparam_k2, 16384), (mk.KMS2partyblock, 2048)],
                         ids=lambda v: getattr(v, "name", str(v)))
def test_baseline_config_shares(require_gpu, p, B):
    """BASELINE.json configs[2..4] at one GPU's share of the batch (65536 / 8 KMS4party gates -- plus a ragged tail that
    crosses the engine's 8192-gate workspace chunk --, 8192 / 8 CCS8party gates, 16384 block-binary gates): cross-party
    input pairs, every output decrypts (where the set's own noise allows), sampled gates equal the oracle bit for
    bit, and the result does not depend on the position of a gate in the batch."""
    crs, keys = keygen(p, 21)
    sg = gpu_scheme(p, crs, keys)
    rng = np.random.default_rng(23)
    nu = 64 * p.nparty
    ubits = rng.integers(0, 2, nu).astype(bool)
    uniq = encrypt_bits(p, keys, ubits, seed=9100)                     # ciphertext j under party j mod k
    ix, iy = rng.integers(0, nu, B), rng.integers(0, nu, B)            # random pairs: same- and cross-party gates
    x, y = uniq[ix], uniq[iy]
    out = sg.gate(0, x, y)
    want = ~(ubits[ix] & ubits[iy])
    got = mk.lwe_decrypt(out, keys if p.multikey else keys[0], p)
    if p.name in NOISY:
        assert (got == want).mean() > 0.98
    else:
        assert np.array_equal(got, want)
    so = oracle_scheme(p, crs, keys)
    pick = np.concatenate([[0, B - 1], rng.integers(0, B, 6)])
    assert np.array_equal(out[pick], so.gate_batch(0, x[pick], y[pick], threads=8))
    perm = rng.permutation(B)[:512]                                    # same gates at other batch positions
    assert np.array_equal(sg.gate(0, x[perm], y[perm]), out[perm])
    sg.close()


def test_two_contexts_two_host_threads(require_gpu):
    """one context per host thread; contexts on one device are independent.  Two schemes (different
    parameter sets, own streams) evaluate batches concurrently from two threads (ctypes drops the GIL during the
    calls); every result equals the single-threaded one.  Also: output aliasing an input (in-place gate)."""
    import threading
    import torch
    ps = [mk.KMS2party.scaled(n=24, N=512), mk.CGGIparam.scaled(n=40, N=1024)]
    work = []
    for i, p in enumerate(ps):
        crs, keys = keygen(p, 50 + i)
        sg = gpu_scheme(p, crs, keys)
        bits = np.random.default_rng(60 + i).integers(0, 2, 513).astype(bool)
        c = encrypt_bits(p, keys, bits, seed=6000 + i)
        x, y = c[:256], c[257:513]
        ref = sg.gate(0, x, y)
        work.append((sg, x, y, ref))
    results = [None, None]

    def run(i):
        sg, x, y, _ = work[i]
        st = torch.cuda.Stream()
        sg.set_stream(st.cuda_stream)
        outs = [sg.gate(0, x, y) for _ in range(6)]
        sg.synchronize()
        results[i] = outs

    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    for i in range(2):
        assert all(np.array_equal(o, work[i][3]) for o in results[i]), f"context {i}"
    sg, x, y, ref = work[0]
    xd = torch.from_numpy(x.view(np.int32)).cuda(); yd = torch.from_numpy(y.view(np.int32)).cuda()
    sg.set_stream(torch.cuda.current_stream().cuda_stream)
    mk.NAND(xd, yd, sg, out=xd)                                        # out aliases the first input
    torch.cuda.synchronize()
    assert np.array_equal(xd.cpu().numpy().view(np.uint32), ref)
    for w in work:
        w[0].close()


KEYGEN_SETS = [
    mk.CGGIparam.scaled(n=20, N=256), mk.CGGIparam.scaled(n=16, N=256, k=2), mk.Blockparam.scaled(n=30, N=256, blk_d=10),
    mk.Blockparam.scaled(n=150, N=128, blk_d=50, k=2), mk.CGGIparam.scaled(n=10, N=256, k=4),
    mk.KMS2party.scaled(n=16, N=256), mk.KMS2partyblock.scaled(n=24, N=256, blk_d=8), mk.CCS2party.scaled(n=12, N=256),
    mk.KMS4party.scaled(n=10, N=512), mk.CCS4party.scaled(n=6, N=1024), mk.CGGIparam, mk.KMS2party,
]


@pytest.mark.parametrize("p", KEYGEN_SETS, ids=lambda p: f"{p.name}-n{p.n}-N{p.N}")
def test_device_keygen_matches_host_keygen(require_gpu, p):
    """the bootstrapping key and the key-switching key generated on the GPU (mkt_keygen_device) are
    the host generator's words exactly -- same seeded streams, exact integer arithmetic: the key-switching key is
    compared word for word, the bootstrapping key through its pre-transformed effect (every stage output and gate of
    a scheme keyed on the device equals the scheme loaded with host keys, hence the oracle)."""
    seed = 77
    crs = mk.CRS(p, seed) if p.multikey else None
    host = [mk.party_keygen(crs, p, deterministic_seed=seed, party=i) for i in range(p.nparty)]
    secr = [mk.party_keygen(crs, p, deterministic_seed=seed, party=i, secrets_only=True) for i in range(p.nparty)]
    assert all(s.brk is None and s.ksk is None for s in secr)
    sh = gpu_scheme(p, crs, host)
    sd = gpu_scheme(p, crs, secr)                                   # setup() generates the large keys on the device
    for i in range(p.nparty):
        assert np.array_equal(sd.get_ksk(i), sh.get_ksk(i)), f"ksk party {i}"
        assert np.array_equal(sh.get_ksk(i).ravel(), host[i].ksk), "read-back layout"
    B = 5
    bits = np.random.default_rng(78).integers(0, 2, 2 * B + 1).astype(bool)
    c = encrypt_bits(p, secr, bits, seed=7800)
    assert np.array_equal(c, encrypt_bits(p, host, bits, seed=7800))   # same secrets
    x, y = c[:B], c[B + 1:]
    lin = np.stack([O.gate_linear(0, x[j], y[j]) for j in range(B)])
    at, bt = sh.modswitch(lin)
    acc0 = np.zeros((B, (1 + (p.k)) * p.N), dtype=p.ring_dtype).reshape(B, -1)
    rng = np.random.default_rng(79)
    acc0 = rng.integers(0, 2**32, acc0.shape, dtype=np.uint64).astype(p.ring_dtype)     # arbitrary accumulator
    assert np.array_equal(sd.blindrotate_(at, acc0.copy()), sh.blindrotate_(at, acc0.copy())), "blind rotation (bootstrapping key)"
    for op in (0, 3):
        out_d = sd.gate(op, x, y)
        assert np.array_equal(out_d, sh.gate(op, x, y))
        assert np.array_equal(mk.lwe_decrypt(out_d, secr if p.multikey else secr[0], p), GATE_FUNCS[op](bits[:B], bits[B + 1:]))
    sh.close(); sd.close()


def _centered(x, W):
    x = x.astype(np.uint64) & np.uint64((1 << W) - 1)
    return np.where(x >= (1 << (W - 1)), x.astype(np.float64) - 2.0 ** W, x.astype(np.float64))


@pytest.mark.parametrize("p", [mk.CGGIparam.scaled(n=10, N=256), mk.CGGIparam.scaled(n=6, N=256, k=2), mk.Blockparam.scaled(n=12, N=256, blk_d=4),
                               mk.KMS2party.scaled(n=8, N=256), mk.KMS2partyblock.scaled(n=12, N=256, blk_d=4), mk.CCS2party.scaled(n=8, N=256)],
                         ids=lambda p: f"{p.name}-k{p.k}")
def test_device_keygen_is_valid_under_the_secrets(require_gpu, p):
    """SURVEY 8f rank 3, judged WITHOUT client.cpp's generator: the keys mkt_keygen_device_export produces on the GPU
    are checked for what they must BE -- every RGSW row an RLWE sample whose phase under the ring secret is s_i * g_j
    (times z_c for the mask rows) within 6 beta (gsw.jl:174-178, lev.jl:88-102, lwe.jl:78-105); every UniEnc (d, f) pair
    consistent with one ternary r (unienc.jl:36-55); every key-switching row an LWE sample of (d+1) z_j 2^(32-(t+1)logD)
    (keygen.jl:17-23) -- with the oracle's exact negacyclic product; and the exported keys, loaded into a fresh context
    as a key blob, evaluate gates exactly like the oracle keyed with them."""
    crs = mk.CRS(p, 5) if p.multikey else None
    secr = [mk.party_keygen(crs, p, party=i, secrets_only=True, deterministic_seed=55) for i in range(p.nparty)]
    sd = mk.Scheme(p)
    if p.multikey:
        sd.load_crs(crs)
    N, W, n = p.N, p.W, p.n
    exported = []
    for i, kk in enumerate(secr):
        brk, ksk = sd.keygen_device(i, kk, export=True)
        exported.append((brk.copy(), ksk.copy()))
        s = kk.lwekey.astype(np.uint64)
        if p.scheme != mk.CCS:
            kr = 1 if p.multikey else p.k
            l, logB = p.l_gsw, p.logB_gsw
            z = [kk.ringkey(c).astype(np.uint64) for c in range(kr)]                # KMS: index 0 = the gsw key
            rows = brk.reshape(n, (kr + 1) * l, kr + 1, N).astype(np.uint64)
            worst = 0.0
            for i_s in range(n):
                for c in range(kr + 1):
                    for j in range(l):
                        r = rows[i_s, c * l + j]
                        ph = r[0].copy()
                        for q in range(kr):
                            ph = ph + O.negacyclic(r[1 + q], z[q], W)
                        msg = np.zeros(N, dtype=np.uint64)
                        g = np.uint64(1 << (W - (j + 1) * logB)) * s[i_s]
                        if c == 0:
                            msg[0] = g
                        else:
                            msg = (z[c - 1] * g)
                        worst = max(worst, np.abs(_centered(ph - msg, W)).max())
            assert worst <= 6 * p.beta + 1, ("RGSW phase", worst)
        else:
            l, logB = p.l_uni, p.logB_uni
            z = kk.ringkey(0).astype(np.uint64)
            rows = brk.reshape(n, 3 * l, N).astype(np.uint64)
            ca = crs.astype(np.uint64)
            for i_s in range(n):
                f0 = _centered(rows[i_s, l] + O.negacyclic(rows[i_s, l + 1], z, W), W)      # phase of f[0] = e + g_0 r
                rr = np.rint(f0 / 2.0 ** (W - logB))
                assert set(np.unique(rr)).issubset({-1.0, 0.0, 1.0})
                rw = rr.astype(np.int64).astype(np.uint64)
                for j in range(l):
                    g = np.uint64(1 << (W - (j + 1) * logB))
                    fj = rows[i_s, l + 2 * j] + O.negacyclic(rows[i_s, l + 2 * j + 1], z, W) - g * rw
                    assert np.abs(_centered(fj, W)).max() <= 6 * p.beta + 1, ("f", i_s, j)
                    dj = rows[i_s, j] - O.negacyclic(ca[j], rw, W)
                    with np.errstate(over="ignore"):                    # words mod 2^W: the wrap is the arithmetic
                        dj[0] -= g * s[i_s]
                    assert np.abs(_centered(dj, W)).max() <= 6 * p.beta + 1, ("d", i_s, j)
        # key-switching key: phase of row (c, j, d, t) = (d + 1) * z_c[j] * 2^(32 - (t+1) logD) + noise(alpha)
        D = 1 << p.logD
        drows = D // 2 if p.scheme in (mk.LMSS, mk.KMS_BLOCK) else D - 1
        kk_r = 1 if p.multikey else p.k
        K = ksk.reshape(kk_r, N, drows, p.f, n + 1)
        phase = (K[..., n] + (K[..., :n] * kk.lwekey.astype(np.uint32)).sum(-1, dtype=np.uint32)).astype(np.uint32)
        zoff = 1 if p.scheme in (mk.KMS, mk.KMS_BLOCK) else 0
        for c in range(kk_r):
            zc = kk.ringkey(zoff + c).astype(np.int64)
            for t in range(4):                                          # deeper levels sit below alpha = 2^17 by design
                sh = 32 - (t + 1) * p.logD
                want = ((np.arange(1, drows + 1)[None, :] * zc[:, None]) << sh) & 0xFFFFFFFF
                err = _centered(phase[c, :, :, t].astype(np.int64) - want, 32)
                if p.scheme in (mk.LMSS, mk.KMS_BLOCK):                 # rows of the embedded LWE key are not generated (keygen.jl:46,:147)
                    live = (c * N + np.arange(N)) >= n
                    err = err[live]
                assert np.abs(err).max() <= 6 * p.alpha, ("ksk", c, t)
    # the exported keys travel as blobs to an evaluator that never sees a secret
    ev = mk.Scheme(p)
    if p.multikey:
        mk.keyblob.load_into(ev, mk.keyblob.dump_crs(p, crs))
    for i, kk in enumerate(secr):
        mk.keyblob.load_into(ev, mk.keyblob.dump_arrays(p, i, exported[i][0], exported[i][1], rlk_d=kk.rlk_d, rlk_f=kk.rlk_f, pubkey=kk.pubkey))
    import types
    so = oracle_scheme(p, crs, [types.SimpleNamespace(brk=exported[i][0], ksk=exported[i][1], pubkey=kk.pubkey, rlk_d=kk.rlk_d, rlk_f=kk.rlk_f)
                                for i, kk in enumerate(secr)])
    bits = np.array([1, 0, 1, 1, 0, 1, 0, 0], dtype=bool)
    c = encrypt_bits(p, secr, bits, seed=5600)
    for op in (0, 2, 3):
        out = ev.gate(op, c[:4], c[4:])
        assert np.array_equal(out, sd.gate(op, c[:4], c[4:]))
        assert np.array_equal(out, np.stack([so.gate(op, c[j], c[4 + j]) for j in range(4)]))
        assert np.array_equal(mk.lwe_decrypt(out, secr if p.multikey else secr[0], p), GATE_FUNCS[op](bits[:4], bits[4:]))
    sd.close(); ev.close()


def test_bench_multi_rank_launch_on_a_shared_gpu(require_gpu):
    """`python bench.py --gpus 2` (no torchrun): the parent starts two rank processes before it touches the GPU, the
    ranks rendezvous (gloo here, both on this box's one GPU: MKT_BENCH_SHARE_GPU=1), each times its own shard, and
    rank 0's line reports the whole job: n_gpus = 2, ranks_seen = 2 (all-reduce census), value over both shards."""
    import json, subprocess, sys
    from helpers import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MKT_BENCH_SHARE_GPU="1", MKT_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "128",
                        "--workload", "cggi", "--no-cpu-baseline", "--no-roofline", "--no-secondary"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["decrypt_errors"] == 0
    assert abs(j["value"] - 2 * 128 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]
    # 128 rotations per rank leave compute units idle: the engine picks the latency variant and the line names THAT kernel
    assert j["roofline"]["kernel"] == "blindrotate_wide_kernel" and 0 < j["roofline"]["frac"] < 1


@pytest.mark.parametrize("workload,batch", [("kms4party", 24), ("ccs8_n2048", 6)])
def test_bench_strong_scaling_shards_a_fixed_batch(require_gpu, workload, batch):
    """BASELINE.json configs[2] / [3] are FIXED total batches over the GPUs of a node: `--scaling strong` shards --batch
    over the ranks (contiguous slices, keys replicated, no data-path collective).  Two ranks on this box's one GPU: the
    line reports the whole batch once, both ranks' step times, and every gate of both shards decrypts."""
    import json, subprocess, sys
    from helpers import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MKT_BENCH_SHARE_GPU="1", MKT_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", str(batch),
                        "--scaling", "strong", "--workload", workload, "--no-cpu-baseline", "--no-roofline", "--no-secondary"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["scaling"] == "strong"
    assert j["config"]["batch_total"] == batch and j["config"]["batch_per_gpu"] == batch // 2
    assert len(j["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in j["per_rank_ms_per_step"])
    assert j["decrypt_checked"] == batch and j["decrypt_ok"]
    assert abs(j["value"] - batch / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]


WIDE_SETS = [
    mk.CGGIparam.scaled(n=24, N=512), mk.CGGIparam.scaled(n=20, N=1024), mk.CGGI_N1024_l2.scaled(n=16), mk.CGGIparam.scaled(n=10, N=2048),
    mk.CGGIparam.scaled(n=12, N=1024, l_gsw=4, logB_gsw=7),
    mk.KMS2party_N1024_l2.scaled(n=12), mk.KMS2party.scaled(n=10), mk.KMS2party.scaled(n=10, N=512), mk.KMS8party.scaled(n=3, N=1024, k=3),
]


@pytest.mark.parametrize("p", WIDE_SETS, ids=lambda p: f"{p.name}-n{p.n}-N{p.N}")
def test_latency_variant_of_the_rotation_is_bit_identical(require_gpu, p, monkeypatch):
    """Small batches run the blind rotation on the latency variant (one rotation spread over 2l thread groups, products
    summed through LDS in the reference's order, bootstrapping.jl:63-68).  Forced on (MKT_ROT_WIDE=2) and off (=1):
    accumulators, KMS phase-1 transforms (raw f64 bits) and gate outputs equal the oracle and each other, including
    skipped (zero) and full-turn (2N) mask words."""
    crs, keys = keygen(p, 31)
    so = oracle_scheme(p, crs, keys)
    sg = gpu_scheme(p, crs, keys)
    B = 5
    rng = np.random.default_rng(32)
    bits = rng.integers(0, 2, 2 * B).astype(bool)
    c = encrypt_bits(p, keys, bits, seed=3200)
    x, y = c[:B], c[B:]
    lin = np.stack([O.gate_linear(0, x[j], y[j]) for j in range(B)])
    at, bt = sg.modswitch(lin)
    at[0, :3] = [0, 2 * p.N, p.N]; at[1, :] = 0; at[2, ::2] = 0
    acc0 = np.stack([so.testvector(bt[j]) for j in range(B)])
    acc_o = np.stack([so.blindrotate(at[j], acc0[j]) for j in range(B)])
    res = {}
    for mode in ("1", "2"):
        sg.set_option("rot_wide", int(mode))      # mkt_set_option: the environment is only read at context creation
        acc_g = sg.blindrotate_(at, acc0.astype(p.ring_dtype).copy())
        assert np.array_equal(acc_g.astype(np.uint64), acc_o), f"blindrotate, MKT_ROT_WIDE={mode}"
        lev = sg.kms_phase1(at) if p.scheme == mk.KMS else None
        out = np.stack([sg.gate(op, x, y) for op in (0, 3, 5)])
        res[mode] = (lev, out)
        for i, op in enumerate((0, 3, 5)):
            assert np.array_equal(out[i], np.stack([so.gate(op, x[j], y[j]) for j in range(B)])), f"gate {op}, MKT_ROT_WIDE={mode}"
            assert np.array_equal(mk.lwe_decrypt(out[i], keys if p.multikey else keys[0], p), GATE_FUNCS[op](bits[:B], bits[B:]))
    if res["1"][0] is not None:
        assert bits_equal(res["1"][0], res["2"][0])
        row = 0
        for party in range(p.k):
            lev_o = so.kms_phase1(party, at[3, party * p.n:(party + 1) * p.n])
            assert bits_equal(res["2"][0][3, row:row + lev_o.shape[0]], lev_o)
            row += lev_o.shape[0]
    sg.close()


@pytest.mark.parametrize("p", [mk.CGGIparam, mk.Blockparam_k2, mk.KMS2partyblock, mk.CCS2party], ids=lambda p: p.name)
def test_keyswitch_at_full_key_length_over_batch_sizes(require_gpu, p):
    """keyswitch! (bootstrapping.jl:81-109, :170-229, :333-364, :664-695) alone, on random accumulators, at the shipped key lengths and at
    batch sizes that leave ragged groups of 32 ciphertexts, one or many slabs and partly filled column chunks in the digit-pair kernel
    (digit words -> per-slab partial sums -> reduce): first, last and random ciphertexts of every batch equal the oracle's output."""
    crs, keys = keygen(p, 81)
    so = oracle_scheme(p, crs, keys)
    sg = gpu_scheme(p, crs, keys)
    rng = np.random.default_rng(82)
    for B in (1, 33, 257, 1000, 2049):
        acc = rng.integers(0, 2**p.W, (B, 1 + p.k, p.N), dtype=np.uint64)
        out = sg.keyswitch(acc.astype(p.ring_dtype))
        for j in sorted({0, B - 1, *rng.integers(0, B, 6).tolist()}):
            assert np.array_equal(out[j], so.keyswitch(acc[j])), (B, j)
    sg.close()


@pytest.mark.parametrize("name,B", [("CGGIparam", 1024 + 100), ("CGGIparam", 1024 + 400), ("CGGIparam", 2048 + 300),
                                     ("KMS2party_N1024_l2", 375), ("KMS2party_N1024_l2", 475), ("KMS2party_N1024_l2", 1125)])
def test_rotation_launch_plans_agree(require_gpu, name, B):
    """A batch past one chip-fill (1024 rotations at N = 1024) is cut into launches by its remainder (launch_blindrotate_k1):
    up to 256 rotations go to the latency variant, 257..512 make two equal launches with the last fill, the rest one launch.
    Each plan against the single launch (rot_split past the batch) and against the oracle, word for word, at reduced n."""
    p = getattr(mk, name).scaled(n=8)
    crs, keys = keygen(p, 71)
    so = oracle_scheme(p, crs, keys)
    sg = gpu_scheme(p, crs, keys)
    rng = np.random.default_rng(72)
    bits = rng.integers(0, 2, 2 * B).astype(bool)
    c = encrypt_bits(p, keys, bits, seed=7200)
    x, y = c[:B], c[B:]
    planned = sg.gate(0, x, y)
    sg.set_option("rot_split", 1 << 30)
    single = sg.gate(0, x, y)
    assert np.array_equal(planned, single)
    assert np.array_equal(planned, so.gate_batch(0, x, y, threads=16))
    sg.close()


def test_forked_contexts_share_one_key_set(require_gpu):
    """mkt_ctx_fork: the reference's contract is ONE read-only scheme shared by concurrent bootstrapping! calls (all
    scratch per call, bootstrapping.jl:38-45).  Forks share the resident keys (no second copy in HBM), each with its own
    stream and workspace: four host threads evaluate different batches concurrently and reproduce the single-threaded
    words; the shared key set is immutable; it outlives the context it was loaded through."""
    import threading
    import torch
    p = mk.KMS2party.scaled(n=40, N=1024)
    crs, keys = keygen(p, 61)
    base = gpu_scheme(p, crs, keys)
    so = oracle_scheme(p, crs, keys)
    rng = np.random.default_rng(62)
    bits = rng.integers(0, 2, 2 * 96).astype(bool)
    c = encrypt_bits(p, keys, bits, seed=6200)
    x, y = c[:96], c[96:]
    want = base.gate(0, x, y)
    assert np.array_equal(want[:4], so.gate_batch(0, x[:4], y[:4], threads=4))
    free0 = torch.cuda.mem_get_info()[0]
    forks = [base.fork() for _ in range(4)]
    key_bytes = p.k * (p.n * 2 * p.l_gsw * 2 * p.N // 2 * 16 + p.N * 3 * p.f * (p.n + 4) * 4)
    assert free0 - torch.cuda.mem_get_info()[0] < key_bytes               # four forks (own streams included) cost less than ONE key copy
    with pytest.raises(mk.MktError):                                       # immutable once shared
        base.load_party(0, keys[0])
    with pytest.raises(mk.MktError):
        forks[0].load_crs(crs)
    base.close()                                                           # the forks keep the key set alive
    outs = [None] * 4

    def run(i):
        if i >= 2:                           # forks 0 and 1 run on the non-blocking stream mkt_ctx_fork gave them
            st = torch.cuda.Stream()
            forks[i].set_stream(st.cuda_stream)
        sl = slice(i * 24, (i + 1) * 24)
        res = [forks[i].gate(0, x[sl], y[sl]) for _ in range(5)]
        forks[i].synchronize()
        outs[i] = res

    th = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
    for i in range(4):
        assert all(np.array_equal(o, want[i * 24:(i + 1) * 24]) for o in outs[i]), f"fork {i}"
    for f in forks:
        f.close()


@pytest.mark.parametrize("p,nfold", [(mk.KMS8party, 2), (mk.KMS4party, 3)], ids=lambda v: getattr(v, "name", str(v)))
def test_many_party_sets_at_full_size(require_gpu, p, nfold):
    """params.jl:55-69 at full size (N = 2048, n = 560, k = 4 / 8 parties, gadget lengths 5 / 4): NAND folds over one fresh
    encryption per party -- the reference's own test shape (test/KMS.jl:23-37) -- every gate output equal to the oracle's
    words, the final ciphertexts decrypt."""
    crs, keys = keygen(p, 91)
    so = oracle_scheme(p, crs, keys)
    sg = gpu_scheme(p, crs, keys)
    k = p.k
    bits = np.random.default_rng(92).integers(0, 2, nfold * k).astype(bool)
    c = encrypt_bits(p, keys, bits, seed=9200)                       # ciphertext j under party j mod k
    acc, ab = c[0::k].copy(), bits[0::k].copy()
    for i in range(1, k):
        nxt = c[i::k]
        out = sg.gate(0, acc, nxt)
        assert np.array_equal(out, so.gate_batch(0, acc, nxt, threads=16)), f"fold step {i}"
        acc, ab = out, ~(ab & bits[i::k])
    assert np.array_equal(mk.lwe_decrypt(acc, keys, p), ab)
    sg.close()


@pytest.mark.parametrize("name,decrypts", [("KMS16party", True), ("KMS32party", True), ("CCS16party", False)])
def test_largest_party_counts_at_full_size(require_gpu, name, decrypts):
    """params.jl:39-45, :71-85 at FULL size: KMS16party, KMS32party (N = 2048, l_uni = 9 / 16) and CCS16party (l = 12, base 2^2).
    Inputs that involve EVERY party without the k - 1 gates of a fold: party 0's encryption of the bit plus, for every other
    party, an encryption of 1 and one of 0 (+1/8 - 1/8: the messages cancel, the mask block stays populated) -- LWE
    ciphertexts add.  Two gate levels on such inputs (every party's rotation, every merge / hybrid product), word for word
    the oracle's.  Decryption is asserted where the set's own noise leaves the margin (profiles/r03_noise_theory_vs_measured.md):
    CCS16party's predicted output sigma is 0.08 against the 0.125 margin (base-4 digits are far from zero-mean: tools/noise_theory.py),
    measured 0.098 -- it does not decrypt reliably on the oracle either, and the reference's tests never run it (test/CCS.jl: CCS2party)."""
    p = getattr(mk, name)
    crs, keys = keygen(p, 3)
    sg = gpu_scheme(p, crs, keys)
    so = oracle_scheme(p, crs, keys)
    B = 4
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2, 2 * B).astype(bool)

    def all_party(bit, seed):
        ct = mk.lwe_ith_encrypt(int(bit), 0, keys[0], p, deterministic_seed=seed).astype(np.uint32)
        for i in range(1, p.k):
            for m in (0, 1):
                ct = ct + mk.lwe_ith_encrypt(m, i, keys[i], p, deterministic_seed=seed + 2 * i + m).astype(np.uint32)
        return ct

    c = np.stack([all_party(bits[j], 100000 * (j + 1)) for j in range(2 * B)])
    assert (c[:, :-1].reshape(2 * B, p.k, p.n) != 0).any(axis=2).all() and np.array_equal(mk.lwe_decrypt(c, keys, p), bits)
    x, y = c[:B], c[B:]
    lvl1 = sg.gate(0, x, y)
    assert np.array_equal(lvl1, so.gate_batch(0, x, y, threads=B)), "level 1"
    lvl2 = sg.gate(3, lvl1, np.roll(lvl1, 1, axis=0))
    assert np.array_equal(lvl2, so.gate_batch(3, lvl1, np.roll(lvl1, 1, axis=0), threads=B)), "level 2"
    if decrypts:
        b1 = ~(bits[:B] & bits[B:])
        assert np.array_equal(mk.lwe_decrypt(lvl1, keys, p), b1)
        assert np.array_equal(mk.lwe_decrypt(lvl2, keys, p), b1 ^ np.roll(b1, 1))
    sg.close()


@pytest.mark.parametrize("N,W", [(32, 32), (64, 64), (256, 32), (1024, 32), (1024, 64), (2048, 64), (4096, 32)])
def test_exact_mode_integer_ntt(require_gpu, N, W):
    """MKT_ARITH_EXACT, transform level: the negacyclic NTT over Z_P[X]/(X^N+1), P = p1 p2 (two 30-bit primes, residue pairs).  Forward transforms
    equal a pure-Python restatement residue for residue (same network and table as the reference's FFT, fft.jl:105-155),
    forward -> inverse is the identity on edge words, and the product of a gadget-digit polynomial with a ring polynomial
    equals the exact schoolbook product mod 2^W (what polynomial.jl:99-113 approximates in Float64) -- bit-exact, also
    where the Float64 path is one below (32-bit ring) or 2^33 off (64-bit ring)."""
    import ref_ntt as R
    rng = np.random.default_rng(N + W)
    p = mk.CGGIparam.scaled(n=8, N=N, W=W)
    ex = mk.Scheme(p, arith=mk.ARITH_EXACT)
    B = 5
    polys = np.stack([edge_words(W, N, rng) for _ in range(B)]).astype(p.ring_dtype)
    t = ex.transform_fwd(polys).view(np.uint64)                       # N residues per polynomial
    assert t.shape == (B, N)
    if N <= 1024:
        for b in range(2):
            assert [int(v) for v in t[b]] == R.fwd(polys[b], W), (N, W, b)
    back = ex.transform_inv(t.view(np.complex128))
    if N <= 1024:
        assert [int(v) for v in back[0]] == R.inv([int(v) for v in t[0]], W)
    # the inverse returns the integer of least magnitude: the identity wherever |signed word| < P / 2 = 2^58.9998
    small = np.abs(polys.astype(np.int64 if W == 64 else np.int32).astype(np.float64)) < 2.0**58.99
    assert np.array_equal(back[small], polys[small]) and small.mean() > 0.05
    if W == 32:
        assert small.all()
    # digit polynomial (balanced digits of every shipped gadget base) times key-like polynomial
    for logB in (2, 9, 16):
        a = rng.integers(-(1 << (logB - 1)), 1 << (logB - 1), (B, N)).astype(np.int64)
        a[0, :4] = [-(1 << (logB - 1)), (1 << (logB - 1)) - 1, 0, -1]
        aw = a.astype(np.uint64).astype(p.ring_dtype) if W == 64 else (a & 0xFFFFFFFF).astype(np.uint32)
        bw = polys
        got = ex.exact_polymul(aw, bw)
        for b in range(B):
            ref = O.negacyclic(aw[b].astype(np.uint64) & np.uint64((1 << W) - 1), bw[b].astype(np.uint64), W)
            assert np.array_equal(got[b].astype(np.uint64), ref), (N, W, logB, b)
    ex.close()


def test_exact_products_at_the_modulus_edge(require_gpu):
    """The largest product the two-prime modulus admits: N = 4096, every digit -2^15, every centered 32-bit piece -2^31, so that
    coefficient N - 1 of a piece product is 4096 * 2^46 = 2^58 (P / 2 = 2^58.9998) -- and the mirrored signs.  Exact, i.e. equal to
    the oracle's schoolbook product mod 2^64 (the lazy ranges of the transforms and the Garner lift at their far end)."""
    p = mk.CGGIparam.scaled(n=8, N=4096, W=64)
    ex = mk.Scheme(p, arith=mk.ARITH_EXACT)
    for sa, sb in ((-(1 << 15), 0x8000000080000000), ((1 << 15) - 1, 0x8000000080000000), (-(1 << 15), 0x7FFFFFFF7FFFFFFF)):
        aw = np.full((1, 4096), sa, dtype=np.int64).astype(np.uint64)
        bw = np.full((1, 4096), sb, dtype=np.uint64)
        assert np.array_equal(ex.exact_polymul(aw, bw)[0].astype(np.uint64), O.negacyclic(aw[0], bw[0], 64)), (sa, hex(sb))
    ex.close()


@pytest.mark.parametrize("p", [mk.KMS2party.scaled(n=8, N=256), mk.KMS2party_N1024_l2.scaled(n=8), mk.KMS2party.scaled(n=6), mk.KMS4party.scaled(n=4, N=512), mk.KMS2party.scaled(n=3, N=4096),
                               mk.KMS8party.scaled(n=3, N=256, k=3), mk.KMS2partyblock.scaled(n=12, N=256, blk_d=4), mk.KMS2partyblock.scaled(n=6, blk_d=2)],
                         ids=lambda p: f"{p.name}-n{p.n}-N{p.N}-k{p.k}")
def test_exact_mode_kms_gates(require_gpu, p):
    """MKT_ARITH_EXACT on the 64-bit ring (KMS, bootstrapping.jl:369-594): every resident 64-bit table kept as the transforms
    of its low and high 32-bit halves, every product sum as a (low, high) accumulator pair with two inverse transforms.
    Accumulators after the whole blind rotation (phase 1 + phase 2) and gate outputs equal the exact-arithmetic restatement
    (tests/ref_exact.py: the oracle's integer steps + exact schoolbook products mod 2^64) word for word, on inputs that
    involve every party; they decrypt; and their noise is BELOW the Float64 path's (whose transform error dominates the KMS
    output noise: profiles/r03_kms_stage_noise.txt)."""
    import ref_exact as RX
    crs, keys = keygen(p, 73)
    so = oracle_scheme(p, crs, keys)
    sx = mk.Scheme(p, arith=mk.ARITH_EXACT)
    sx.set_option("exact_impl", 0)                                    # this test is the integer-NTT path's; tests/test_gpu_fx.py runs the same shapes on the Float64 pipe
    sx.load_crs(crs)
    for i, kk in enumerate(keys):
        sx.load_party(i, kk)
    k, B = p.k, 3
    rng = np.random.default_rng(74)
    bits = rng.integers(0, 2, 2 * B * k).astype(bool)
    c = encrypt_bits(p, keys, bits, seed=7400)                       # ciphertext j under party j mod k
    # all-party inputs by ciphertext addition: party 0's bit + (enc 1 + enc 0) of every other party
    def allp(j):
        ct = mk.lwe_ith_encrypt(int(bits[j]), 0, keys[0], p, deterministic_seed=8000 + 100 * j).astype(np.uint32)
        for i in range(1, k):
            for m in (0, 1):
                ct = ct + mk.lwe_ith_encrypt(m, i, keys[i], p, deterministic_seed=8000 + 100 * j + 2 * i + m).astype(np.uint32)
        return ct
    x = np.stack([allp(j) for j in range(B)]); y = np.stack([allp(B + j) for j in range(B)])
    lin = np.stack([O.gate_linear(0, x[j], y[j]) for j in range(B)])
    at, bt = sx.modswitch(lin)
    at[0, :2] = [0, 2 * p.N]
    acc0 = np.stack([so.testvector(bt[j]) for j in range(B)])
    acc_x = sx.blindrotate_(at, acc0.astype(np.uint64).copy())
    for j in range(B):
        assert np.array_equal(acc_x[j], RX.kms_blindrotate(p, keys, crs, at[j], acc0[j])), f"exact KMS blind rotation {j}"
    if p.blk_len > 1:
        # KMS_block has two phase-1 kernels: one set of digit transforms per block, each multiplied by a key bit's monomial row before the
        # key rows (exact integers re-associate), and one that transforms the digits again for every key bit (exact_wide = 0) -- the same words
        assert sx.last_kernel_name() == "exact_kms_block_phase1_kernel"
        sx.set_option("exact_wide", 0)
        assert np.array_equal(sx.blindrotate_(at, acc0.astype(np.uint64).copy()), acc_x) and sx.last_kernel_name() == "exact_kms_phase1_kernel"
        sx.set_option("exact_wide", 1)
    for op in (0, 3):
        out = sx.gate(op, x, y)
        assert np.array_equal(out, np.stack([RX.kms_gate(p, so, keys, crs, op, x[j], y[j]) for j in range(B)])), f"exact KMS gate {op}"
        assert np.array_equal(mk.lwe_decrypt(out, keys, p), GATE_FUNCS[op](bits[:B], bits[B:2 * B]))
    z = x.copy()
    sx.bootstrapping_(z)
    assert np.array_equal(mk.lwe_decrypt(z, keys, p), bits[:B])
    sx.close()


# The integer-NTT KMS phase 1 at l_gsw = 2 has two kernels (exact_wide 0: one transform at a time, one product chain per term -- the reference's
# loop order; 1, the default: paired transforms, products gathered in 64 bits, the first sum's key rows requested ahead): both must give the
# big-integer restatement's words, at three ring sizes and ragged batches.
@pytest.mark.parametrize("wide", [0, 1])
@pytest.mark.parametrize("p", [mk.KMS2party_N1024_l2.scaled(n=10), mk.KMS2party_N1024_l2.scaled(n=8, N=256), mk.KMS2party_N1024_l2.scaled(n=12, N=512)], ids=lambda p: f"N{p.N}-n{p.n}")     # (N = 2048 with this gadget exceeds the two-prime modulus: refused)
def test_exact_kms_phase1_kernels_are_word_identical(require_gpu, p, wide):
    import ref_exact as RX
    crs, keys = keygen(p, 83)
    so = oracle_scheme(p, crs, keys)
    sx = mk.Scheme(p, arith=mk.ARITH_EXACT)
    sx.load_crs(crs)
    for i, kk in enumerate(keys):
        sx.load_party(i, kk)
    sx.set_option("exact_impl", 0)                                    # the integer-NTT kernels (the Float64-pipe phase 1 has tests/test_gpu_fx.py)
    sx.set_option("exact_wide", wide)
    rng = np.random.default_rng(84)
    for B in (3, 2):                                                  # 3 gates x 3 RLEV rows = 9 rotations: odd
        bits = rng.integers(0, 2, 2 * B).astype(bool)
        c = encrypt_bits(p, keys, bits, seed=8400 + B)
        x, y = c[:B], c[B:]
        out = sx.gate(0, x, y)
        assert np.array_equal(out, np.stack([RX.kms_gate(p, so, keys, crs, 0, x[j], y[j]) for j in range(B)])), (wide, B)
        assert np.array_equal(mk.lwe_decrypt(out, keys, p), ~(bits[:B] & bits[B:]))
    lin = np.stack([O.gate_linear(0, x[j], y[j]) for j in range(B)])
    at, bt = sx.modswitch(lin)
    at[0, :3] = [0, 2 * p.N, 0]                                       # zero mask words: the reference skips them, kernel 2 multiplies by X^0 - 1 = 0
    acc0 = np.stack([so.testvector(bt[j]) for j in range(B)])
    acc_x = sx.blindrotate_(at, acc0.astype(np.uint64).copy())
    for j in range(B):
        assert np.array_equal(acc_x[j], RX.kms_blindrotate(p, keys, crs, at[j], acc0[j])), (wide, j)
    sx.close()


@pytest.mark.parametrize("p", [mk.CCS2party.scaled(n=8, N=256), mk.CCS2party.scaled(n=6), mk.CCS4party.scaled(n=4, N=512), mk.CCS8party.scaled(n=3, N=256, k=3), mk.CCS2party.scaled(n=3, N=4096),
                               mk.CCS16party.scaled(n=2, N=256, k=4)], ids=lambda p: f"{p.name}-n{p.n}-N{p.N}-k{p.k}")
def test_exact_mode_ccs_gates(require_gpu, p):
    """MKT_ARITH_EXACT for CCS (bootstrapping.jl:234-364, 32-bit ring): the hybrid products over Z_P.  Accumulators after the
    blind rotation and gate outputs equal the exact-arithmetic restatement (tests/ref_exact.py ccs_*) word for word on inputs
    that involve every party, and decrypt."""
    import ref_exact as RX
    crs, keys = keygen(p, 75)
    so = oracle_scheme(p, crs, keys)
    sx = mk.Scheme(p, arith=mk.ARITH_EXACT)
    sx.load_crs(crs)
    for i, kk in enumerate(keys):
        sx.load_party(i, kk)
    k, B = p.k, 3
    bits = np.random.default_rng(76).integers(0, 2, 2 * B).astype(bool)

    def allp(j):
        ct = mk.lwe_ith_encrypt(int(bits[j]), 0, keys[0], p, deterministic_seed=9000 + 100 * j).astype(np.uint32)
        for i in range(1, k):
            for m in (0, 1):
                ct = ct + mk.lwe_ith_encrypt(m, i, keys[i], p, deterministic_seed=9000 + 100 * j + 2 * i + m).astype(np.uint32)
        return ct
    x = np.stack([allp(j) for j in range(B)]); y = np.stack([allp(B + j) for j in range(B)])
    lin = np.stack([O.gate_linear(0, x[j], y[j]) for j in range(B)])
    at, bt = sx.modswitch(lin)
    at[0, :2] = [0, 2 * p.N]
    acc0 = np.stack([so.testvector(bt[j]) for j in range(B)])
    acc_x = sx.blindrotate_(at, acc0.astype(np.uint32).copy())
    for j in range(B):
        assert np.array_equal(acc_x[j].astype(np.uint64), RX.ccs_blindrotate(p, keys, crs, at[j], acc0[j])), f"exact CCS blind rotation {j}"
    for op in (0, 3):
        out = sx.gate(op, x, y)
        assert np.array_equal(out, np.stack([RX.ccs_gate(p, so, keys, crs, op, x[j], y[j]) for j in range(B)])), f"exact CCS gate {op}"
        assert np.array_equal(mk.lwe_decrypt(out, keys, p), GATE_FUNCS[op](bits[:B], bits[B:]))
    sx.close()


@pytest.mark.parametrize("p", [mk.CGGIparam.scaled(n=12, N=256), mk.CGGIparam.scaled(n=10, N=1024), mk.CGGI_N1024_l2.scaled(n=10),
                               mk.CGGIparam.scaled(n=6, N=2048, l_gsw=4, logB_gsw=7), mk.CGGIparam.scaled(n=4, N=4096), mk.Blockparam.scaled(n=6, N=4096, blk_d=2),
                               mk.Blockparam.scaled(n=12, N=256, blk_d=4), mk.Blockparam.scaled(n=9, N=1024, blk_d=3),
                               # RLWE length 2 / 3 and other block lengths: exact_blindrotate_kr_kernel (BASELINE configs[4] = LMSS, k = 2, in the integer arithmetic)
                               mk.Blockparam_k2.scaled(n=9, blk_d=3), mk.Blockparam_k2.scaled(n=12, N=256, blk_d=4), mk.CGGIparam.scaled(n=8, N=256, k=2),
                               mk.CGGIparam.scaled(n=6, N=512, k=3, l_gsw=2, logB_gsw=10), mk.Blockparam.scaled(n=8, N=256, blk_d=4, blk_len=2),
                               mk.Blockparam.scaled(n=8, N=128, blk_d=2, blk_len=4, k=2),
                               # RLWE length beyond 3 (scheme.jl:6-36 leaves k free): exact_blindrotate_kany_kernel, sums in memory
                               mk.CGGIparam.scaled(n=6, N=256, k=4), mk.Blockparam.scaled(n=8, N=128, blk_d=4, blk_len=2, k=5),
                               mk.Blockparam.scaled(n=9, N=512, blk_d=3, k=4)],
                         ids=lambda p: f"{p.name}-n{p.n}-N{p.N}-l{p.l_gsw}-k{p.k}-b{p.blk_len}")
def test_exact_mode_cggi_gates(require_gpu, p):
    """MKT_ARITH_EXACT gate path (CGGI and LMSS, 32-bit ring): blind rotation with integer-NTT products.  Accumulators and
    gate outputs equal the exact-arithmetic restatement (tests/ref_exact.py: the oracle's integer steps + exact schoolbook
    products) word for word, and decrypt."""
    import ref_exact as RX
    crs, keys = keygen(p, 71)
    so = oracle_scheme(p, crs, keys)
    sx = mk.Scheme(p, arith=mk.ARITH_EXACT)
    sx.set_option("exact_impl", 0)                                    # the integer-NTT kernels (tests/test_gpu_fx.py: the Float64 pipe on the k = 1 shapes)
    sx.load_party(0, keys[0])
    B = 4
    bits = np.array([1, 0, 1, 1, 0, 1, 0, 0], dtype=bool)
    c = encrypt_bits(p, keys, bits, seed=7100)
    x, y = c[:B], c[B:]
    lin = np.stack([O.gate_linear(0, x[j], y[j]) for j in range(B)])
    at, bt = sx.modswitch(lin)
    at[0, :3] = [0, 2 * p.N, p.N]
    acc0 = np.stack([so.testvector(bt[j]) for j in range(B)])
    acc_x = sx.blindrotate_(at, acc0.astype(np.uint32).copy())
    for j in range(B):
        rot = RX.blindrotate_lmss if p.blk_len > 1 else RX.blindrotate
        assert np.array_equal(acc_x[j].astype(np.uint64).reshape(-1), rot(p, keys[0].brk, at[j], acc0[j])), f"exact blindrotate {j}"
    # the run-time-RLWE-length kernel (the only one beyond k = 3) forced where the register kernels serve: the same words
    assert ("kany" in sx.last_kernel_name()) == (p.k > 3)
    sx.set_option("exact_kany", 1)
    assert np.array_equal(sx.blindrotate_(at, acc0.astype(np.uint32).copy()), acc_x) and "kany" in sx.last_kernel_name()
    sx.set_option("exact_kany", 0)
    for op in (0, 3, 5):
        out = sx.gate(op, x, y)
        assert np.array_equal(out, np.stack([RX.gate(p, so, keys[0].brk, op, x[j], y[j]) for j in range(B)])), f"exact gate {op}"
        assert np.array_equal(mk.lwe_decrypt(out, keys[0], p), GATE_FUNCS[op](bits[:B], bits[B:]))
    # keys generated on the device from the secrets (mkt_keygen_device on an EXACT context: the same words, uploaded as residues)
    sd = mk.Scheme(p, arith=mk.ARITH_EXACT)
    sd.keygen_device(0, keys[0])
    assert np.array_equal(sd.gate(0, x, y), sx.gate(0, x, y))
    sd.close()
    # one CMux step from the same accumulator: the Float64 path is the exact value or one below it per coefficient
    # (truncating native(), arithmetic.jl:1-9); over many steps the two diverge in the words, not in the phase
    sf = gpu_scheme(p, crs, keys)
    one = np.zeros_like(at); one[:, 0] = at[:, 1]
    e1 = sx.blindrotate_(one, acc0.astype(np.uint32).copy()).astype(np.int64)
    f1 = sf.blindrotate_(one, acc0.astype(np.uint32).copy()).astype(np.int64)
    d = (e1 - f1 + (1 << 31)) % (1 << 32) - (1 << 31)
    assert d.min() >= 0 and d.max() <= 2
    sx.close(); sf.close()
