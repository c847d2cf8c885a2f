"""Second, independent restatement of the reference's gate-bootstrapping path, in numpy (test infrastructure).

Purpose: the C oracle (oracle/mkt_oracle.c) is the root of trust of every parity test, and the reference cannot be run
here (no Julia).  This module transcribes the same Julia source a second time -- different language, different data
layout (one array per polynomial, one numpy operation per reference operation), written from the .jl files, not from
the C oracle -- so that `tests/test_oracle_cpu.py::test_numpy_restatement_*` can compare the two bit for bit.  A slip
in either restatement (operand order, rounding point, index off-by-one) shows up as a mismatch; a shared misreading of
the Julia source would not.  All five schemes are covered: CGGI (bootstrapping.jl:4-109), LMSS (:114-229), CCS
(:234-364), KMS (:369-594) and KMS_block (:599-695).

numpy's element-wise float64 add / multiply are single IEEE operations (no fused multiply-add), which is the reference's
arithmetic (Julia Base `*`, `+` on Complex{Float64}; no @fastmath, no muladd on this path).
Twiddle tables come from tests/golden/twiddles.npz (mpmath, see gen_twiddles.py).
"""
import os

import numpy as np

np.seterr(over="ignore")       # ring words wrap modulo 2^W by design (unsigned numpy scalars warn otherwise)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class C:
    """a vector of Complex{Float64} kept as two float64 arrays so that every real operation is explicit"""
    __slots__ = ("re", "im")

    def __init__(self, re, im):
        self.re, self.im = re, im

    @staticmethod
    def zeros(m):
        return C(np.zeros(m), np.zeros(m))

    def copy(self):
        return C(self.re.copy(), self.im.copy())

    def __mul__(self, o):      # Julia Base: (a+bi)(c+di) = (ac - bd) + (ad + bc)i
        return C(self.re * o.re - self.im * o.im, self.re * o.im + self.im * o.re)

    def __add__(self, o):
        return C(self.re + o.re, self.im + o.im)

    def __sub__(self, o):
        return C(self.re - o.re, self.im - o.im)


class FFT:
    """src/ring/fft.jl: FFTransformer, fftto! (:57-63), ifftto! (:74-81), fft! (:105-155), ifft! (:159-209)"""

    def __init__(self, N, W):
        g = np.load(os.path.join(GOLD, "twiddles.npz"))
        self.N, self.M, self.W = N, N // 2, W
        t = lambda name: C(g[f"{name}_{N}"].real.copy(), g[f"{name}_{N}"].imag.copy())
        self.psi, self.psiinv, self.roots, self.rootsinv = t("psi"), t("psiinv"), t("roots"), t("rootsinv")
        self.udt = np.uint64 if W == 64 else np.uint32
        self.sdt = np.int64 if W == 64 else np.int32

    def fwd(self, p):
        """fftto!: t[i] = (signed(p[i]) - im*signed(p[halfN+i])) * roots[i]; fft!(t, Psi)"""
        M = self.M
        s = p.view(self.sdt)
        a = C(s[:M].astype(np.float64), (-s[M:]).astype(np.float64)) * self.roots    # integer negation, then convert
        # Cooley-Tukey (:105-155): m groups of 2k points, butterfly (j, j+k) with Psi[m + i] (0-based table index)
        m, k = 1, M >> 1
        while k >= 1:
            re, im = a.re.reshape(m, 2, k), a.im.reshape(m, 2, k)
            w = C(self.psi.re[m:2 * m, None], self.psi.im[m:2 * m, None])
            t = C(re[:, 0, :], im[:, 0, :])
            u = C(re[:, 1, :], im[:, 1, :]) * w                                       # a[j+k] * Psi[m+i+1]
            hi, lo = t + u, t - u
            a = C(np.stack([hi.re, lo.re], 1).reshape(M), np.stack([hi.im, lo.im], 1).reshape(M))
            m, k = m << 1, k >> 1
        return a

    def native(self, x):
        """arithmetic.jl:1-9"""
        if self.W == 32:
            y = x - np.floor(x * 2.3283064365386963e-10) * 4.294967296e9
            return np.where(y == 4.294967296e9, 0, np.trunc(y)).astype(np.uint64).astype(np.uint32)
        y = x - np.floor(x * 5.421010862427522e-20) * 1.8446744073709552e19
        y = np.where(y == 1.8446744073709552e19, 0.0, np.trunc(y))
        hi = np.floor(y * 2.0 ** -32)                     # exact split of an integer-valued double < 2^64
        lo = y - hi * 2.0 ** 32
        return (hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)

    def inv(self, a):
        """ifftto!: ifft!(t, Psiinv); t .*= rootsinv; p[1:halfN] = native(real), p[halfN+1:] = native(-imag)"""
        M = self.M
        a = a.copy()
        m, k = M >> 1, 1
        while m >= 1:                                                                  # Gentleman-Sande (:159-209)
            re, im = a.re.reshape(m, 2, k), a.im.reshape(m, 2, k)
            w = C(self.psiinv.re[m:2 * m, None], self.psiinv.im[m:2 * m, None])
            t, u = C(re[:, 0, :], im[:, 0, :]), C(re[:, 1, :], im[:, 1, :])
            hi, lo = t + u, (t - u) * w                                                # (t - u) * Psiinv[m+i+1]
            a = C(np.stack([hi.re, lo.re], 1).reshape(M), np.stack([hi.im, lo.im], 1).reshape(M))
            m, k = m >> 1, k << 1
        a = a * self.rootsinv
        return np.concatenate([self.native(a.re), self.native(-a.im)])


def divbits(a, bit, W):                                   # arithmetic.jl:23-27 on an array of W-bit words
    if bit == 0:
        return a.copy()
    dt = a.dtype.type
    carry = (a << dt(W - bit)) >> dt(W - 1)
    return (a >> dt(bit)) + carry


def decomp_poly(a, l, logB, W):
    """gsw.jl:86-96 decompto!(avec, a::NativePoly, params): -> l digit polynomials (index 0 = most significant)"""
    dt = a.dtype.type
    mask, half = dt((1 << logB) - 1), dt(1 << (logB - 1))
    out = [None] * l
    run = divbits(a, W - l * logB, W)                     # avec[1] doubles as the running value
    for j in range(l - 1, 0, -1):
        d = run & mask
        run = run >> dt(logB)
        run = run + (d >> dt(logB - 1))
        out[j] = d - ((d & half) << dt(1))
    d = run & mask
    out[0] = d - ((d & half) << dt(1))
    return out


def unbalanced_word(a, l, logB):
    """lev.jl unbalanceddecompto!(avec, a::UInt32, kskpar): python ints, index 0 = most significant"""
    bit = 32 - l * logB
    ai = (a >> bit) + (((a << (32 - bit)) & 0xFFFFFFFF) >> 31) if bit else a
    out = [0] * l
    for i in range(l - 1, -1, -1):
        out[i] = ai & ((1 << logB) - 1)
        ai >>= logB
    return out


class Scheme:
    """CGGI (scheme.jl:107-146) or KMS (scheme.jl:256-297) evaluator state built from integer-form keys"""

    def __init__(self, p, crs, keys):
        import mktfhe_amd as mk
        self.mk, self.p = mk, p
        self.N, self.n, self.k, self.W = p.N, p.n, p.k, p.W
        self.f = FFT(p.N, p.W)
        self.udt = self.f.udt
        self.kms = p.scheme in (mk.KMS, mk.KMS_BLOCK)
        self.block = p.scheme in (mk.LMSS, mk.KMS_BLOCK)
        self.ccs = p.scheme == mk.CCS
        self.multikey = self.kms or self.ccs
        N, T = p.N, self.udt
        # scheme.jl:121-146 getmonomial: entry e (1-based) = fft(X^e - 1) for e < N, fft(-2) at N, fft(-1 - X^(e-N)) above, 0 at 2N
        self.monomial = [None] * (2 * N + 1)
        tmp = np.zeros(N, dtype=T); tmp[0] = T(0) - T(1)
        for i in range(1, N):
            tmp[i] = 1; self.monomial[i] = self.f.fwd(tmp); tmp[i] = 0
        tmp[0] = T(0) - T(2); self.monomial[N] = self.f.fwd(tmp); tmp[0] = T(0) - T(1)
        for i in range(1, N):
            tmp[i] = T(0) - T(1); self.monomial[N + i] = self.f.fwd(tmp); tmp[i] = 0
        self.monomial[2 * N] = C.zeros(N // 2)
        tr = lambda arr: [self.f.fwd(np.ascontiguousarray(x)) for x in arr]
        self.parties = []
        for kk in keys:
            d = {}
            if self.ccs:                                                          # UniEnc: d[l], then (f[j].b, f[j].a)
                u = kk.brk.reshape(p.n, 3 * p.l_uni, N)
                d["uni_d"] = [tr(e[:p.l_uni]) for e in u]
                d["uni_f"] = [[tr(e[p.l_uni + 2 * j:p.l_uni + 2 * j + 2]) for j in range(p.l_uni)] for e in u]
            else:
                if self.kms:
                    brk = kk.brk.reshape(p.n, 2 * p.l_gsw, 2, N)
                else:
                    brk = kk.brk.reshape(p.n, (p.k + 1) * p.l_gsw, p.k + 1, N)
                d["brk"] = [[tr(row) for row in e] for e in brk]                  # [n][rows][polys] TransPolys
            D1 = (1 << p.logD) // 2 if self.block else (1 << p.logD) - 1
            d["ksk"] = kk.ksk.reshape(-1, N, D1, p.f, p.n + 1)                     # [component][coef][digit-1][level][a..., b]
            if self.multikey:
                d["pub"] = tr(kk.pubkey.reshape(p.l_uni, N))
            if self.kms:
                d["rlk_d"] = tr(kk.rlk_d.reshape(p.l_uni, N))
                d["rlk_f"] = [tr(x) for x in kk.rlk_f.reshape(p.l_uni, 2, N)]
            self.parties.append(d)
        if self.multikey:
            self.crs = tr(np.asarray(crs).reshape(p.l_uni, N))

    # ---- bootstrapping.jl:4-27
    def bootstrap(self, ct):
        p, N = self.p, self.N
        logN = N.bit_length() - 1
        ct = ct.astype(np.uint32)
        tilde = divbits(ct, 32 - logN - 1, 32)
        ta, tb = tilde[:-1], int(tilde[-1])
        T = self.udt
        eighth = T(1) << T(self.W - 3)
        i1 = np.arange(1, N + 1)
        if tb <= N:
            b = np.where(i1 <= tb, eighth, T(0) - eighth).astype(T)
        else:
            tb -= N
            b = np.where(i1 <= tb, T(0) - eighth, eighth).astype(T)
        acc = [b] + [np.zeros(N, dtype=T) for _ in range(p.k)]
        mk = self.mk
        if self.kms:
            acc = self.blindrotate_kms(ta, acc)
        elif self.ccs:
            acc = self.blindrotate_ccs(ta, acc)
        elif p.scheme == mk.LMSS:
            acc = self.blindrotate_lmss(ta, acc)
        else:
            acc = self.blindrotate_cggi(ta, acc)
        if p.scheme == mk.LMSS:
            return self.keyswitch_lmss(acc)
        if p.scheme == mk.KMS_BLOCK:
            return self.keyswitch_kms_block(acc)
        return self.keyswitch(acc)

    # ---- bootstrapping.jl:32-76
    def blindrotate_cggi(self, ta, acc):
        p, f = self.p, self.f
        l, k = p.l_gsw, p.k
        brk = self.parties[0]["brk"]
        for idx in range(p.n):
            if ta[idx] == 0:
                continue
            tb = [f.fwd(d) for d in decomp_poly(acc[0], l, p.logB_gsw, self.W)]
            tav = [[f.fwd(d) for d in decomp_poly(acc[1 + i], l, p.logB_gsw, self.W)] for i in range(k)]
            tacc = [C.zeros(self.N // 2) for _ in range(k + 1)]
            for i in range(l):                                          # basketb.stack[i]
                for q in range(k + 1):
                    tacc[q] = tacc[q] + tb[i] * brk[idx][i][q]
            for i in range(k):                                          # basketa[i].stack[j]
                for j in range(l):
                    for q in range(k + 1):
                        tacc[q] = tacc[q] + tav[i][j] * brk[idx][(1 + i) * l + j][q]
            mono = self.monomial[int(ta[idx])]
            for q in range(k + 1):
                acc[q] = acc[q] + f.inv(mono * tacc[q])
        return acc

    # ---- bootstrapping.jl:389-443
    def phase1(self, party, ta):
        p, f, N = self.p, self.f, self.N
        l = p.l_gsw
        brk = self.parties[party]["brk"]
        it = 1 if party == 0 else p.l_lev
        T = self.udt
        stack = []
        for i in range(it):
            b = np.zeros(N, dtype=T); b[0] = T(1) << T(self.W - (i + 1) * p.logB_lev)
            stack.append([b, np.zeros(N, dtype=T)])
        if self.block:                                                  # bootstrapping.jl:599-659
            for idx1 in range(p.blk_d):
                new = []
                for i in range(it):
                    tb = [f.fwd(d) for d in decomp_poly(stack[i][0], l, p.logB_gsw, self.W)]
                    tav = [f.fwd(d) for d in decomp_poly(stack[i][1], l, p.logB_gsw, self.W)]
                    tacc2 = [C.zeros(N // 2), C.zeros(N // 2)]
                    for idx2 in range(p.blk_len):
                        idx = idx1 * p.blk_len + idx2
                        if ta[idx] == 0:
                            continue
                        tacc = [C.zeros(N // 2), C.zeros(N // 2)]
                        for j in range(l):
                            for q in range(2):
                                tacc[q] = tacc[q] + tb[j] * brk[idx][j][q]
                        for j in range(l):
                            for q in range(2):
                                tacc[q] = tacc[q] + tav[j] * brk[idx][l + j][q]
                        mono = self.monomial[int(ta[idx])]
                        for q in range(2):
                            tacc2[q] = tacc2[q] + mono * tacc[q]
                    new.append([stack[i][q] + f.inv(tacc2[q]) for q in range(2)])
                stack = new
            return [[f.fwd(r[0]), f.fwd(r[1])] for r in stack]
        for idx in range(p.n):
            if ta[idx] == 0:
                continue
            mono = self.monomial[int(ta[idx])]
            new = []
            for i in range(it):
                tb = [f.fwd(d) for d in decomp_poly(stack[i][0], l, p.logB_gsw, self.W)]
                tav = [f.fwd(d) for d in decomp_poly(stack[i][1], l, p.logB_gsw, self.W)]
                tacc = [C.zeros(N // 2), C.zeros(N // 2)]
                for j in range(l):
                    for q in range(2):
                        tacc[q] = tacc[q] + tb[j] * brk[idx][j][q]
                for j in range(l):
                    for q in range(2):
                        tacc[q] = tacc[q] + tav[j] * brk[idx][l + j][q]
                new.append([stack[i][q] + f.inv(mono * tacc[q]) for q in range(2)])
            stack = new
        return [[f.fwd(r[0]), f.fwd(r[1])] for r in stack]

    # ---- bootstrapping.jl:369-385, :448-558
    def blindrotate_kms(self, ta, acc):
        p, f, N, k = self.p, self.f, self.N, self.k
        M = N // 2
        lev = [self.phase1(i, ta[i * p.n:(i + 1) * p.n]) for i in range(k)]
        ll, lu = p.l_lev, p.l_uni
        for idx in range(k):                                            # idx = 0-based party; idx mask polys active before
            tb = [f.fwd(d) for d in decomp_poly(acc[0], ll, p.logB_lev, self.W)]
            tav = [[f.fwd(d) for d in decomp_poly(acc[1 + i], ll, p.logB_lev, self.W)] for i in range(idx)]
            it = 1 if idx == 0 else ll
            tx = [C.zeros(M) for _ in range(k + 1)]
            ty = [C.zeros(M) for _ in range(k + 1)]
            for i in range(it):
                tx[0] = tx[0] + tb[i] * lev[idx][i][0]
            for i in range(idx):
                for j in range(it):
                    tx[1 + i] = tx[1 + i] + tav[i][j] * lev[idx][j][0]
            for i in range(it):
                ty[0] = ty[0] + tb[i] * lev[idx][i][1]
            for i in range(idx):
                for j in range(it):
                    ty[1 + i] = ty[1 + i] + tav[i][j] * lev[idx][j][1]
            yb = f.inv(ty[0])
            ya = [f.inv(ty[1 + i]) for i in range(idx)]
            tb = [f.fwd(d) for d in decomp_poly(yb, lu, p.logB_uni, self.W)]
            tav = [[f.fwd(d) for d in decomp_poly(ya[i], lu, p.logB_uni, self.W)] for i in range(idx)]
            P = self.parties[idx]
            ty = [C.zeros(M) for _ in range(k + 1)]
            for i in range(lu):
                ty[0] = ty[0] + tb[i] * P["rlk_d"][i]
            for i in range(idx):
                for j in range(lu):
                    ty[1 + i] = ty[1 + i] + tav[i][j] * P["rlk_d"][j]
            tv = C.zeros(M)
            for i in range(lu):
                tv = tv - tb[i] * self.crs[i]                            # mulsubto!
            for i in range(idx):
                for j in range(lu):
                    tv = tv + tav[i][j] * self.parties[i]["pub"][j]
            v = f.inv(tv)
            tvv = [f.fwd(d) for d in decomp_poly(v, lu, p.logB_uni, self.W)]
            for i in range(lu):
                ty[0] = ty[0] + tvv[i] * P["rlk_f"][i][0]
                ty[1 + idx] = ty[1 + idx] + tvv[i] * P["rlk_f"][i][1]
            acc = [f.inv(tx[q] + ty[q]) for q in range(k + 1)]
        return acc

    # ---- bootstrapping.jl:81-109 (CGGI), :564-594 (KMS)
    def keyswitch(self, acc):
        p, N, n = self.p, self.N, self.n
        bitdiff = self.W - 32
        nparty = self.k if self.multikey else 1
        res = np.zeros(nparty * n + 1, dtype=np.uint32)
        res[-1] = np.uint32(int(acc[0][0]) >> bitdiff)
        for i in range(p.k):
            a = acc[1 + i]
            ksk = self.parties[i]["ksk"][0] if self.multikey else self.parties[0]["ksk"][i]
            off = i * n if self.multikey else 0
            for j in range(N):                                          # extracted coefficient j (0-based)
                w = int(a[0]) >> bitdiff if j == 0 else (-(int(a[N - j]) >> bitdiff)) & 0xFFFFFFFF
                w &= 0xFFFFFFFF
                for t, d in enumerate(unbalanced_word(w, p.f, p.logD)):
                    if d > 0:
                        row = ksk[j, d - 1, t]
                        res[off:off + n] += row[:n]
                        res[-1] += row[n]
        return res

    # ---- bootstrapping.jl:114-165
    def blindrotate_lmss(self, ta, acc):
        p, f = self.p, self.f
        l, k = p.l_gsw, p.k
        brk = self.parties[0]["brk"]
        for idx1 in range(p.blk_d):
            tb = [f.fwd(d) for d in decomp_poly(acc[0], l, p.logB_gsw, self.W)]
            tav = [[f.fwd(d) for d in decomp_poly(acc[1 + i], l, p.logB_gsw, self.W)] for i in range(k)]
            tacc2 = [C.zeros(self.N // 2) for _ in range(k + 1)]
            for idx2 in range(p.blk_len):
                idx = idx1 * p.blk_len + idx2
                if ta[idx] == 0:
                    continue
                tacc = [C.zeros(self.N // 2) for _ in range(k + 1)]
                for i in range(l):
                    for q in range(k + 1):
                        tacc[q] = tacc[q] + tb[i] * brk[idx][i][q]
                for i in range(k):
                    for j in range(l):
                        for q in range(k + 1):
                            tacc[q] = tacc[q] + tav[i][j] * brk[idx][(1 + i) * l + j][q]
                mono = self.monomial[int(ta[idx])]
                for q in range(k + 1):
                    tacc2[q] = tacc2[q] + mono * tacc[q]                 # muladdto!(tacc2, monomial, tacc)
            acc = [acc[q] + f.inv(tacc2[q]) for q in range(k + 1)]
        return acc

    # ---- bootstrapping.jl:234-328
    def blindrotate_ccs(self, ta, acc):
        p, f, N, k, n = self.p, self.f, self.N, self.k, self.n
        M, l, logB = N // 2, p.l_uni, p.logB_uni
        for idx in range(k):                                            # party; mask polys 0..idx are live
            P = self.parties[idx]
            for i in range(n):
                at = int(ta[idx * n + i])
                if at == 0:
                    continue
                tb = [f.fwd(d) for d in decomp_poly(acc[0], l, logB, self.W)]
                tav = [[f.fwd(d) for d in decomp_poly(acc[1 + j1], l, logB, self.W)] for j1 in range(idx + 1)]
                ud, uf = P["uni_d"][i], P["uni_f"][i]
                tacc = [C.zeros(M) for _ in range(k + 1)]
                for j in range(l):                                      # u
                    tacc[0] = tacc[0] + tb[j] * ud[j]
                for j1 in range(idx + 1):
                    for j2 in range(l):
                        tacc[1 + j1] = tacc[1 + j1] + tav[j1][j2] * ud[j2]
                tv0 = C.zeros(M)                                         # v
                tv = [C.zeros(M) for _ in range(idx + 1)]
                for j in range(l):
                    tv0 = tv0 - tb[j] * self.crs[j]
                for j1 in range(idx + 1):
                    for j2 in range(l):
                        tv[j1] = tv[j1] + tav[j1][j2] * self.parties[j1]["pub"][j2]
                v0 = f.inv(tv0)
                v = [f.inv(x) for x in tv]
                tv0v = [f.fwd(d) for d in decomp_poly(v0, l, logB, self.W)]
                tvv = [[f.fwd(d) for d in decomp_poly(v[j1], l, logB, self.W)] for j1 in range(idx + 1)]
                for j in range(l):                                      # w
                    tacc[0] = tacc[0] + tv0v[j] * uf[j][0]
                    tacc[1 + idx] = tacc[1 + idx] + tv0v[j] * uf[j][1]
                for j1 in range(idx + 1):
                    for j2 in range(l):
                        tacc[0] = tacc[0] + tvv[j1][j2] * uf[j2][0]
                        tacc[1 + idx] = tacc[1 + idx] + tvv[j1][j2] * uf[j2][1]
                mono = self.monomial[at]
                acc = [acc[q] + f.inv(mono * tacc[q]) for q in range(k + 1)]
        return acc

    def _balanced_word(self, w):
        """gsw.jl:42-52 decompto!(avec, a::UInt32, kskpar): signed digits, index 0 = most significant"""
        l, logB = self.p.f, self.p.logD
        mask, half = (1 << logB) - 1, 1 << (logB - 1)
        bit = 32 - l * logB
        ai = ((w >> bit) + (((w << (32 - bit)) & 0xFFFFFFFF) >> 31)) & 0xFFFFFFFF if bit else w
        out = [0] * l
        for i in range(l - 1, 0, -1):
            d = ai & mask
            ai >>= logB
            ai = (ai + (d >> (logB - 1))) & 0xFFFFFFFF
            out[i] = d - ((d & half) << 1)
        d = ai & mask
        out[0] = d - ((d & half) << 1)
        return out

    def _ks_balanced(self, res, off, ksk, j, w):
        n = self.n
        for t, d in enumerate(self._balanced_word(w)):
            if d > 0:
                row = ksk[j, d - 1, t]; res[off:off + n] += row[:n]; res[-1] += row[n]
            elif d != 0:
                row = ksk[j, -d - 1, t]; res[off:off + n] -= row[:n]; res[-1] -= row[n]

    def _extract(self, a, j, bitdiff):
        N = self.N
        return (int(a[0]) >> bitdiff) & 0xFFFFFFFF if j == 0 else (-(int(a[N - j]) >> bitdiff)) & 0xFFFFFFFF

    # ---- bootstrapping.jl:170-229
    def keyswitch_lmss(self, acc):
        p, N, n = self.p, self.N, self.n
        res = np.zeros(n + 1, dtype=np.uint32)
        res[-1] = np.uint32(int(acc[0][0]) & 0xFFFFFFFF)
        current = 1                                                     # 1-based, as in the source
        for i in range(p.k):
            a, ksk = acc[1 + i], self.parties[0]["ksk"][i]
            if current + N <= n:
                for j in range(N):
                    res[current - 1 + j] = self._extract(a, j, 0)
                current += N
            elif current <= n:
                cnt = n - current + 1                                   # words copied
                for j in range(cnt):
                    res[current - 1 + j] = self._extract(a, j, 0)
                for j in range(cnt, N):
                    self._ks_balanced(res, 0, ksk, j, self._extract(a, j, 0))
                current = n + 1
            else:
                for j in range(N):
                    self._ks_balanced(res, 0, ksk, j, self._extract(a, j, 0))
        return res

    # ---- bootstrapping.jl:664-695
    def keyswitch_kms_block(self, acc):
        p, N, n, k = self.p, self.N, self.n, self.k
        bitdiff = self.W - 32
        res = np.zeros(k * n + 1, dtype=np.uint32)
        res[-1] = np.uint32((int(acc[0][0]) >> bitdiff) & 0xFFFFFFFF)
        for i in range(k):
            a, ksk = acc[1 + i], self.parties[i]["ksk"][0]
            for j in range(n):
                res[i * n + j] = self._extract(a, j, bitdiff)
            for j in range(n, N):
                self._ks_balanced(res, i * n, ksk, j, self._extract(a, j, bitdiff))
        return res

    # ---- gate.jl:1-8
    def nand(self, c1, c2):
        lin = (np.uint32(0) - c1.astype(np.uint32) - c2.astype(np.uint32)).astype(np.uint32)
        lin[-1] = np.uint32((1 << 29) - int(c1[-1]) - int(c2[-1]) & 0xFFFFFFFF)
        return self.bootstrap(lin)
