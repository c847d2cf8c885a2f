"""CPU tests of the Float64-pipe EXACT product's algorithm (tests/ref_fx.py restates mktfhe_amd/csrc/fx_exact.hip in numpy): the engine's own
transform pair, the centered 16-bit limb split, rounding to the exact integer, and the host-side error bound with its measured side.  The GPU side of
the same statements is tests/test_gpu_fx.py; what both compute is the exact negacyclic product the reference's Float64 transform approximates
(/root/reference/src/ring/polynomial.jl:99-113)."""
import numpy as np
import pytest

from helpers import O
import ref_fx as F


def rand_words(rng, n, W):
    return (rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n).astype(np.uint64)) & np.uint64((1 << W) - 1)


def schoolbook_sum(digits, keys, W):
    acc = np.zeros(len(digits[0]), dtype=np.uint64)
    for d, k in zip(digits, keys):
        acc = (acc + O.negacyclic(np.asarray(d, dtype=np.int64).astype(np.uint64) & np.uint64((1 << W) - 1), np.asarray(k, dtype=np.uint64), W)) & np.uint64((1 << W) - 1)
    return [int(x) for x in acc]


@pytest.mark.parametrize("N", [16, 64, 256])
def test_transform_pair_is_a_negacyclic_convolution(N):
    """forward x forward -> pointwise -> inverse, untwisted and scaled, is the negacyclic product (the transform-domain points of a resident key
    and of a digit polynomial are in the same bit-reversed order, so no permutation sits between them)"""
    rng = np.random.default_rng(N)
    om, tw = F.tables(N)
    a, b = rng.integers(-50, 50, N), rng.integers(-50, 50, N)
    c = F.inverse(F.fx_transform(a.astype(float), om, tw) * F.fx_transform(b.astype(float), om, tw) / (N // 2)) * np.conj(tw)
    got = np.rint(np.concatenate([c.real, -c.imag])).astype(np.int64)
    ref = np.array(schoolbook_sum([a], [b.astype(np.int64).astype(np.uint64)], 64), dtype=np.uint64).astype(np.int64)
    assert np.array_equal(got, ref)
    # position twiddles of the inverse are conjugates of forward-table entries: exp(+i pi j / h) = conj(om[h + rev_b(j)])
    for b_ in range(1, (N // 2).bit_length() - 1):
        h = 1 << b_
        assert np.allclose([np.exp(1j * np.pi * j / h) for j in range(h)], [np.conj(om[h + F.bitrev(j, b_)]) for j in range(h)], atol=1e-15)


@pytest.mark.parametrize("W", [32, 64])
def test_limbs_recombine_mod_2W(W):
    rng = np.random.default_rng(W)
    words = [0, 1, (1 << W) - 1, 1 << (W - 1), (1 << (W - 1)) - 1, (1 << (W - 1)) + 1, 0x7FFF, 0x8000, 0xFFFF8000 % (1 << W)] + [int(x) for x in rand_words(rng, 200, W)]
    lb = F.limbs_of(words, W)
    assert len(lb) == W // 16 and all(int(np.abs(x).max()) <= 1 << 15 for x in lb)
    for i, w in enumerate(words):
        assert sum(int(lb[h][i]) << (16 * h) for h in range(W // 16)) % (1 << W) == w


@pytest.mark.parametrize("N,W,l,logB", [(64, 32, 3, 9), (256, 64, 2, 16), (256, 64, 3, 12), (1024, 64, 2, 16)])
def test_rounded_sums_are_the_exact_product(N, W, l, logB):
    """random and adversarial operands (every digit at +-2^(logB-1), every limb at +-2^15, one sign / alternating): the rounded limb sums recombine to the
    schoolbook product mod 2^W; the largest pre-rounding distance from an integer stays under the PROVEN bound (fx_bound with this key's measured
    transform magnitude), which itself is under 1/2 for the keys a generator makes"""
    rng = np.random.default_rng(N + W + l)
    G, half = 2 * l, 1 << (logB - 1)
    cases = {"random": ([rng.integers(-half, half, N) for _ in range(G)], [rand_words(rng, N, W) for _ in range(G)])}
    lim = sum(0x8000 << (16 * h) for h in range(W // 16))
    cases["one sign"] = ([np.full(N, -half) for _ in range(G)], [np.full(N, lim, dtype=np.uint64) for _ in range(G)])
    cases["alternating"] = ([np.where(np.arange(N) & 1, -half, half - 1) for _ in range(G)], [np.where(np.arange(N) & 1, lim, lim >> 1).astype(np.uint64) for _ in range(G)])
    for name, (d, k) in cases.items():
        got, worst = F.exact_product_sum(d, [[int(x) for x in kk] for kk in k], W)
        assert got == schoolbook_sum(d, k, W), name
        kmax = F.key_max([[int(x) for x in kk] for kk in k], W)
        bound = F.fx_bound(N, l, logB, kmax)
        assert worst <= bound, (name, worst, bound)
        if name == "random":
            assert bound < 0.45 and kmax < 6 * np.sqrt(N) * 32768 / np.sqrt(3), (bound, kmax)


def test_bound_matches_the_documented_figures():
    """headline 0.10 for a generated key, 0.50 for the worst-case key (hence the measured side), KMS2party 0.02, CGGIparam 1.2e-3"""
    gen = lambda N: 4 * np.sqrt(N) * 32768 / np.sqrt(3)
    assert abs(F.fx_bound(1024, 2, 16, gen(1024)) - 0.101) < 0.002
    assert F.fx_bound(1024, 2, 16, 0.65 * 1024 * 32768) > 0.45
    assert abs(F.fx_bound(2048, 3, 12, gen(2048)) - 0.0216) < 0.001
    assert F.fx_bound(1024, 3, 9, gen(1024)) < 2e-3
