"""Exact-arithmetic restatement of the CGGI and LMSS gate bootstraps (test infrastructure; checker of the MKT_ARITH_EXACT
gate path): bootstrapping.jl:4-76 / :114-165 with every transform-domain product replaced by the exact negacyclic product
mod 2^W (the oracle's schoolbook), i.e. what the reference's Float64 pipeline approximates.  Mod-switch, test vector, gadget
decomposition and key switch are the integer steps of the oracle itself."""
import numpy as np

from helpers import O


def monomial_minus_one(x, a, N, W):
    """(X^a - 1) * x in Z_{2^W}[X]/(X^N + 1), a in [1, 2N]  (scheme.jl:121-146: table entry a)"""
    mask = np.uint64((1 << W) - 1)
    x = x.astype(np.uint64)
    s = a % (2 * N)
    r = np.roll(x, s % N)
    neg = np.zeros(N, dtype=bool)
    neg[: s % N] = True                       # coefficients that wrapped around X^N = -1
    if s >= N:
        neg = ~neg
    r = np.where(neg, (np.uint64(0) - r) & mask, r)
    return (r - x) & mask


def blindrotate(p, brk, atilde, acc):
    """brk: integer bootstrapping key [n][2l][2][N]; acc: [2][N] ring words (b, a)"""
    N, W, l = p.N, p.W, p.l_gsw
    mask = np.uint64((1 << W) - 1)
    acc = acc.reshape(2, N).astype(np.uint64).copy()
    brk = brk.reshape(p.n, 2 * l, 2, N).astype(np.uint64)
    for i in range(p.n):
        a = int(atilde[i])
        if a == 0:
            continue                                                   # bootstrapping.jl:48
        dig = [O.decomp_poly(acc[c], l, p.logB_gsw, W) for c in range(2)]   # :50-51, [l][N] wrapped signed digits
        for pp in range(2):
            t = np.zeros(N, dtype=np.uint64)
            for c in range(2):
                for j in range(l):                                     # :63-68, exactly
                    t = (t + O.negacyclic(dig[c][j], brk[i, c * l + j, pp], W)) & mask
            acc[pp] = (acc[pp] + monomial_minus_one(t, a, N, W)) & mask       # :71-73
    return acc.reshape(-1)


def blindrotate_lmss(p, brk, atilde, acc):
    """LMSS (bootstrapping.jl:114-165): one decomposition per block of blk_len key bits, every key bit of the block multiplies
    the SAME digits into its own rows, the block adds sum_q (X^a_q - 1) * product_q"""
    N, W, l, L = p.N, p.W, p.l_gsw, p.blk_len
    mask = np.uint64((1 << W) - 1)
    acc = acc.reshape(2, N).astype(np.uint64).copy()
    brk = brk.reshape(p.n, 2 * l, 2, N).astype(np.uint64)
    for blk in range(p.n // L):
        dig = [O.decomp_poly(acc[c], l, p.logB_gsw, W) for c in range(2)]   # :131-132
        add = np.zeros((2, N), dtype=np.uint64)
        for q in range(L):
            i = blk * L + q
            a = int(atilde[i])
            if a == 0:
                continue                                                   # :145
            for pp in range(2):
                t = np.zeros(N, dtype=np.uint64)
                for c in range(2):
                    for j in range(l):                                     # :146-154
                        t = (t + O.negacyclic(dig[c][j], brk[i, c * l + j, pp], W)) & mask
                add[pp] = (add[pp] + monomial_minus_one(t, a, N, W)) & mask       # :157
        acc = (acc + add) & mask                                           # :162-163
    return acc.reshape(-1)


def gate(p, so, brk, op, x, y):
    lin = O.gate_linear(op, x, y)
    at, bt = so.modswitch(lin)
    rot = blindrotate_lmss if p.blk_len > 1 else blindrotate
    acc = rot(p, brk, at, so.testvector(bt))
    return so.keyswitch(acc)
