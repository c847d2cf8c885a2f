"""Exact-arithmetic restatement of the CGGI, LMSS and KMS gate bootstraps (test infrastructure; checker of the MKT_ARITH_EXACT
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
    """brk: integer bootstrapping key [n][(kr+1) l][kr+1][N], kr = p.k the RLWE length; acc: [kr+1][N] ring words (b, a_0 ..)"""
    N, W, l, kr = p.N, p.W, p.l_gsw, p.k
    mask = np.uint64((1 << W) - 1)
    acc = acc.reshape(kr + 1, N).astype(np.uint64).copy()
    brk = brk.reshape(p.n, (kr + 1) * l, kr + 1, N).astype(np.uint64)
    for i in range(p.n):
        a = int(atilde[i])
        if a == 0:
            continue                                                   # bootstrapping.jl:48
        dig = [O.decomp_poly(acc[c], l, p.logB_gsw, W) for c in range(kr + 1)]   # :50-51, [l][N] wrapped signed digits
        new = acc.copy()
        for pp in range(kr + 1):
            t = np.zeros(N, dtype=np.uint64)
            for c in range(kr + 1):
                for j in range(l):                                     # :63-68, exactly
                    t = (t + O.negacyclic(dig[c][j], brk[i, c * l + j, pp], W)) & mask
            new[pp] = (acc[pp] + monomial_minus_one(t, a, N, W)) & mask          # :71-73
        acc = new
    return acc.reshape(-1)


def blindrotate_lmss(p, brk, atilde, acc):
    """LMSS (bootstrapping.jl:114-165): one decomposition per block of blk_len key bits, every key bit of the block multiplies
    the SAME digits into its own rows, the block adds sum_q (X^a_q - 1) * product_q; RLWE length kr = p.k"""
    N, W, l, L, kr = p.N, p.W, p.l_gsw, p.blk_len, p.k
    mask = np.uint64((1 << W) - 1)
    acc = acc.reshape(kr + 1, N).astype(np.uint64).copy()
    brk = brk.reshape(p.n, (kr + 1) * l, kr + 1, N).astype(np.uint64)
    for blk in range(p.n // L):
        dig = [O.decomp_poly(acc[c], l, p.logB_gsw, W) for c in range(kr + 1)]   # :131-132
        add = np.zeros((kr + 1, N), dtype=np.uint64)
        for q in range(L):
            i = blk * L + q
            a = int(atilde[i])
            if a == 0:
                continue                                                   # :145
            for pp in range(kr + 1):
                t = np.zeros(N, dtype=np.uint64)
                for c in range(kr + 1):
                    for j in range(l):                                     # :146-154
                        t = (t + O.negacyclic(dig[c][j], brk[i, c * l + j, pp], W)) & mask
                add[pp] = (add[pp] + monomial_minus_one(t, a, N, W)) & mask       # :157
        acc = (acc + add) & mask                                           # :162-163
    return acc.reshape(-1)


def kms_phase1(p, brk, at_party, party):
    """bootstrapping.jl:389-443 with exact products: the RLEV rows [(b, a)] of one party (1 row for party 0, l_lev otherwise)"""
    N, W, l = p.N, p.W, p.l_gsw
    mask = np.uint64((1 << W) - 1)
    brk = brk.reshape(p.n, 2 * l, 2, N).astype(np.uint64)
    rows = []
    for r in range(1 if party == 0 else p.l_lev):
        acc = np.zeros((2, N), dtype=np.uint64)
        acc[0, 0] = np.uint64(1) << np.uint64(W - (r + 1) * p.logB_lev)             # :403-406
        L = p.blk_len if p.blk_len > 1 else 1                                  # KMS_block (:599-659): one decomposition per block
        for blk in range(p.n // L):
            ats = [int(at_party[blk * L + q]) for q in range(L)]
            if not any(ats):
                continue                                                       # :413 / :638
            dig = [O.decomp_poly(acc[c], l, p.logB_gsw, W) for c in range(2)]
            add = np.zeros((2, N), dtype=np.uint64)
            for q, a in enumerate(ats):
                if a == 0:
                    continue
                i = blk * L + q
                for pp in range(2):
                    t = np.zeros(N, dtype=np.uint64)
                    for c in range(2):
                        for j in range(l):
                            t = (t + O.negacyclic(dig[c][j], brk[i, c * l + j, pp], W)) & mask
                    add[pp] = (add[pp] + monomial_minus_one(t, a, N, W)) & mask      # :435 / :648
            acc = (acc + add) & mask                                           # :437 / :654
        rows.append(acc)
    return rows


def kms_phase2(p, lev, keys, crs, acc):
    """bootstrapping.jl:448-558 with exact products.  lev[party] = rows of kms_phase1; keys: PartyKeys (integer rlk_d, rlk_f,
    pubkey); crs: [l_uni][N]; acc: [k+1][N] (test vector in polynomial 0)"""
    N, W, k = p.N, p.W, p.k
    ll, lu = p.l_lev, p.l_uni
    mask = np.uint64((1 << W) - 1)
    mul = lambda d, t: O.negacyclic(d, t, W)
    acc = [np.asarray(a, dtype=np.uint64).copy() for a in np.asarray(acc).reshape(k + 1, N)]
    crs = np.asarray(crs).reshape(lu, N).astype(np.uint64)
    for idx in range(k):
        rows = lev[idx]
        it = 1 if idx == 0 else ll                                             # :481
        rd = keys[idx].rlk_d.reshape(lu, N).astype(np.uint64)
        rf = keys[idx].rlk_f.reshape(lu, 2, N).astype(np.uint64)
        dig = [O.decomp_poly(acc[q], ll, p.logB_lev, W) for q in range(idx + 1)]    # :470-479
        tx = [np.zeros(N, dtype=np.uint64) for _ in range(k + 1)]
        ty = [np.zeros(N, dtype=np.uint64) for _ in range(k + 1)]
        for q in range(idx + 1):
            for j in range(it):                                                # :485-499
                tx[q] = (tx[q] + mul(dig[q][j], rows[j][0])) & mask
                ty[q] = (ty[q] + mul(dig[q][j], rows[j][1])) & mask
        ydig = [O.decomp_poly(ty[q], lu, p.logB_uni, W) for q in range(idx + 1)]    # :501-517
        ty = [np.zeros(N, dtype=np.uint64) for _ in range(k + 1)]
        tv = np.zeros(N, dtype=np.uint64)
        for q in range(idx + 1):
            vk = crs if q == 0 else keys[q - 1].pubkey.reshape(lu, N).astype(np.uint64)
            for j in range(lu):                                                # :521-535
                ty[q] = (ty[q] + mul(ydig[q][j], rd[j])) & mask
                pr = mul(ydig[q][j], vk[j])
                tv = (tv - pr) & mask if q == 0 else (tv + pr) & mask
        vdig = O.decomp_poly(tv, lu, p.logB_uni, W)                              # :538-544
        for i in range(lu):                                                    # :547-550
            ty[0] = (ty[0] + mul(vdig[i], rf[i, 0])) & mask
            ty[1 + idx] = (ty[1 + idx] + mul(vdig[i], rf[i, 1])) & mask
        acc = [(tx[q] + ty[q]) & mask for q in range(k + 1)]                   # :553-556
    return np.stack(acc)


def kms_blindrotate(p, keys, crs, atilde, acc):
    lev = [kms_phase1(p, keys[i].brk, atilde[i * p.n:(i + 1) * p.n], i) for i in range(p.k)]
    return kms_phase2(p, lev, keys, crs, acc)


def kms_gate(p, so, keys, crs, op, x, y):
    lin = O.gate_linear(op, x, y)
    at, bt = so.modswitch(lin)
    return so.keyswitch(kms_blindrotate(p, keys, crs, at, so.testvector(bt)))


def ccs_blindrotate(p, keys, crs, atilde, acc):
    """bootstrapping.jl:234-328 with exact products (32-bit ring): UniEnc brk [n][3l][N] = d[l], then (f[j].b, f[j].a)"""
    N, W, k, l, n = p.N, p.W, p.k, p.l_uni, p.n
    mask = np.uint64((1 << W) - 1)
    mul = lambda d, t: O.negacyclic(d, t, W)
    acc = [np.asarray(a, dtype=np.uint64).copy() for a in np.asarray(acc).reshape(k + 1, N)]
    crs = np.asarray(crs).reshape(l, N).astype(np.uint64)
    for idx in range(k):
        npm = idx + 1
        uni_all = keys[idx].brk.reshape(n, 3 * l, N).astype(np.uint64)
        for i in range(n):
            a = int(atilde[idx * n + i])
            if a == 0:
                continue                                                       # :261
            ud, uf = uni_all[i, :l], uni_all[i, l:].reshape(l, 2, N)
            dig = [O.decomp_poly(acc[q], l, p.logB_uni, W) for q in range(npm + 1)]     # :264-275
            tacc = [np.zeros(N, dtype=np.uint64) for _ in range(k + 1)]
            for q in range(npm + 1):                                           # :279-284 u
                for j in range(l):
                    tacc[q] = (tacc[q] + mul(dig[q][j], ud[j])) & mask
            for q in range(npm + 1):                                           # :287-300 v, :303-320 w
                vk = crs if q == 0 else keys[q - 1].pubkey.reshape(l, N).astype(np.uint64)
                v = np.zeros(N, dtype=np.uint64)
                for j in range(l):
                    pr = mul(dig[q][j], vk[j])
                    v = (v - pr) & mask if q == 0 else (v + pr) & mask
                vd = O.decomp_poly(v, l, p.logB_uni, W)
                for j in range(l):
                    tacc[0] = (tacc[0] + mul(vd[j], uf[j, 0])) & mask
                    tacc[1 + idx] = (tacc[1 + idx] + mul(vd[j], uf[j, 1])) & mask
            for q in range(npm + 1):                                           # :322-324
                acc[q] = (acc[q] + monomial_minus_one(tacc[q], a, N, W)) & mask
    return np.stack(acc)


def ccs_gate(p, so, keys, crs, op, x, y):
    lin = O.gate_linear(op, x, y)
    at, bt = so.modswitch(lin)
    return so.keyswitch(ccs_blindrotate(p, keys, crs, at, so.testvector(bt)))


def gate(p, so, brk, op, x, y):
    lin = O.gate_linear(op, x, y)
    at, bt = so.modswitch(lin)
    rot = blindrotate_lmss if p.blk_len > 1 else blindrotate
    acc = rot(p, brk, at, so.testvector(bt))
    return so.keyswitch(acc)
