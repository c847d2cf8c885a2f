"""Pure-numpy restatement of the Float64-pipe EXACT product (mktfhe_amd/csrc/fx_exact.hip) -- test infrastructure, CPU only.

What it restates: the engine's OWN transform (not the reference's network) and the limb algebra around it:
    fold a_j = p_j - i p_{j+M};  twist by rho^j, rho = exp(-i pi / N);  cyclic Cooley-Tukey forward, natural -> bit-reversed, twiddles by block
    om[m + i] = exp(-i pi rev_s(i) / m);  decimation-in-time inverse, bit-reversed -> natural, twiddles by position exp(+i pi j / h);  untwist by conj(rho^j) / M;
    key word K = sum_h limb_h 2^(16 h) mod 2^W with centered 16-bit limbs;  sum_g d_g (*) K_g = sum_h 2^(16 h) round( sum_g d_g (*) limb_{g,h} ) mod 2^W.
numpy has no fused multiply-add, so the roundings differ from the kernel's; the ROUNDED integers cannot (that is the point of the bound), and the
tests compare them with big-integer schoolbook products.  Also the host-side error bound of context.cpp (fx_bound), restated, with the measured side of it."""
import math

import numpy as np


def bitrev(i, bits):
    r = 0
    for b in range(bits):
        r |= ((i >> b) & 1) << (bits - 1 - b)
    return r


def tables(N):
    M = N // 2
    logM = M.bit_length() - 1
    om = np.ones(M, dtype=np.complex128)
    for s in range(1, logM):
        m = 1 << s
        om[m:2 * m] = [np.exp(-1j * np.pi * bitrev(i, s) / m) for i in range(m)]
    j = np.arange(M)
    return om, np.exp(-1j * np.pi * j / N)


def forward(b, om):
    a = np.array(b, dtype=np.complex128)
    M, m, t = len(a), 1, len(a) // 2
    while m < M:
        for i in range(m):
            lo = slice(2 * i * t, 2 * i * t + t); hi = slice(2 * i * t + t, 2 * i * t + 2 * t)
            u, v = a[lo].copy(), a[hi] * om[m + i]
            a[lo], a[hi] = u + v, u - v
        m *= 2; t //= 2
    return a                      # bit-reversed frequency order


def inverse(A):
    a = np.array(A, dtype=np.complex128)
    M, h = len(a), 1
    while h < M:
        tau = np.exp(1j * np.pi * np.arange(h) / h)
        blk = a.reshape(-1, 2 * h)
        x, y = blk[:, :h].copy(), blk[:, h:] * tau
        blk[:, :h], blk[:, h:] = x + y, x - y
        h *= 2
    return a                      # natural order, unscaled


def fx_transform(p, om, tw):
    M = len(p) // 2
    return forward((p[:M] - 1j * p[M:]) * tw, om)


def limbs_of(words, W):
    """centered 16-bit limbs of ring words (python ints mod 2^W): list over limb index of int64 arrays, every step exact mod 2^W"""
    out, v = [], [int(x) % (1 << W) for x in words]
    for _ in range(W // 16):
        r = [((x & 0xFFFF) ^ 0x8000) - 0x8000 for x in v]
        v = [(((x - y) % (1 << W)) >> 16) - ((1 << (W - 16)) if ((x - y) % (1 << W)) >> (W - 1) else 0) for x, y in zip(v, r)]
        v = [x % (1 << W) for x in v]
        out.append(np.array(r, dtype=np.int64))
    return out


def exact_product_sum(digits, keys, W):
    """sum_g digits[g] (*) keys[g] mod (X^N + 1, 2^W) on the Float64 pipe; digits: signed small ints [G][N], keys: ring words [G][N].
    -> (words as python ints, largest distance of a pre-rounding value from the nearest integer)"""
    N = len(digits[0]); M = N // 2
    om, tw = tables(N)
    D = [fx_transform(np.asarray(d, dtype=np.float64), om, tw) for d in digits]
    L = [limbs_of(k, W) for k in keys]
    res, worst = [0] * N, 0.0
    for h in range(W // 16):
        S = sum(D[g] * (fx_transform(L[g][h].astype(np.float64), om, tw) / M) for g in range(len(digits)))
        c = inverse(S) * np.conj(tw)
        q = np.concatenate([c.real, -c.imag])
        r = np.rint(q)
        worst = max(worst, float(np.abs(q - r).max()))
        for i in range(N):
            res[i] = (res[i] + (int(r[i]) << (16 * h))) % (1 << W)
    return res, worst


def fx_bound(N, l, logB, kmax, blk_len=1):
    """context.cpp fx_bound, restated: proven bound on |computed - exact| of one rounded sum"""
    u, logM, g2 = 2.0 ** -53, int(math.log2(N)) - 1, 2.0 * l
    gt = (3.0 + 5.5 * (logM - 2) + 1.5 * 2) * u
    gm = (2.0 + 3.0 * g2) * u
    dn, kn = math.sqrt(N) * 2.0 ** (logB - 1), math.sqrt(N) * 32768.0
    return (gt + (gt + u) + gm) * g2 * dn * kn + gt * g2 * dn * kmax


def key_max(keys, W):
    """largest transform-domain magnitude over the limbs of the given key polynomials (what fx_key_fwd_kernel measures)"""
    N = len(keys[0])
    om, tw = tables(N)
    return max(float(np.abs(fx_transform(lb.astype(np.float64), om, tw)).max()) for k in keys for lb in limbs_of(k, W))
