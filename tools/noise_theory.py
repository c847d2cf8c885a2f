"""Predicted output noise of a gate bootstrap for every shipped parameter set, from the schemes' own correctness
identities -- NOT from the oracle, the engine or the Julia source's arithmetic.  CPU only.

What is modelled (torus units; sigma_r = beta / 2^W ring-key noise, sigma_l = alpha / 2^32 LWE noise):
  * gadget digits of the reference's decomposition (gsw.jl:42-52) are uniform on [-B/2, B/2): E[d] = -1/2, E[d^2] = (B^2 + 2) / 12;
    the rounding error of a decomposition to l digits is uniform on +-2^-(l logB + 1);
  * binary keys have E[z] = E[z^2] = 1/2, block-binary LWE keys 1/(len + 1), the ternary r of UniEnc E[r^2] = 2/3;
  * CGGI16 / LMSS23: one external product adds (k+1) l N E[d^2] sigma_r^2 + s (1 + sum z^2) eps^2 to the phase variance, and the
    reference forms acc += (X^a - 1) (BRK [.] acc) (bootstrapping.jl:71-73, :157), which doubles it;
  * CCS19 hybrid product (unienc.jl:36-90, bootstrapping.jl:262-320): u, v, w terms as derived in DESIGN.md 8; because the digits and
    the keys are NOT zero-mean, products  digits (*) key  carry a coherent ramp  (-1/4)(2c + 2 - N)  on top of their random part:
    sum_c E[(D (*) z)_c^2] = N^2 (E[d^2]/2 - 1/16) + N^3 / 48, and the ramps of the np mask polynomials of a step ADD (np^2 N^3 / 48).
    At base 2^2 (CCS16party) the ramp is 30x the random part: this, not a defect, is why CCS16party does not decrypt;
  * KMS (eprint 2022/1460; bootstrapping.jl:389-558): the same coherent terms act on the phase-1 rows' error, which is itself long-range
    correlated (delta (*) z' with E[z'] = 1/2), so no closed form is attempted: the error recursion of phase 1 and the error identity of
    phase 2 are SIMULATED on random digits, keys and rounding errors (a linear noise model: no ciphertexts, no transforms); the one
    empirical input is the error of a single Float64 product of a digit polynomial with a 64-bit polynomial, measured here against
    the exact integer product (the 64-bit ring in Float64 is inexact: README.md:9 of the reference offers MultiFloats for that reason);
  * key switch: every non-zero digit of every extracted coefficient adds one LWE row's noise.

  python tools/noise_theory.py [--measured gpurun_out/r03b/noise_all.jsonl] [NAME ...]  ->  table: predicted sigma, measured sigma, ratio"""
import argparse
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mktfhe_amd as mk  # noqa: E402  (parameter sets only)

V = lambda logB: ((1 << (2 * logB)) + 2) / 12.0                       # E[d^2] of a balanced base-2^logB digit
eps2 = lambda l, logB, W: (2.0 ** (-2 * l * logB)) / 12.0 if l * logB < W else 0.0


def ks_var(p, coeffs, zsq, balanced=False):
    """key switch: coeffs extracted coefficients, sum of z^2 over them = zsq"""
    D = 1 << p.logD
    sl2 = (p.alpha / 2.0**32) ** 2
    return coeffs * p.f * (1.0 - 1.0 / D) * sl2 + zsq * eps2(p.f, p.logD, 32)


def cggi(p):
    N, kr, l = p.N, p.k, p.l_gsw
    sr2 = (p.beta / 2.0**p.W) ** 2
    ep = (kr + 1) * l * N * V(p.logB_gsw) * sr2 + 0.5 * (1 + kr * N / 2.0) * eps2(l, p.logB_gsw, p.W)
    br = p.n * 2.0 * ep
    return br, ks_var(p, kr * N, kr * N / 2.0)


def lmss(p):
    N, kr, l, LB, d = p.N, p.k, p.l_gsw, p.blk_len, p.blk_d
    sr2 = (p.beta / 2.0**p.W) ** 2
    zsq = p.n / (LB + 1.0) + (kr * N - p.n) / 2.0                     # ring key: the first n coefficients embed the block-binary LWE key (key.jl:52-69)
    per_block = LB * 2.0 * (kr + 1) * l * N * V(p.logB_gsw) * sr2 + (LB / (LB + 1.0)) * 2.0 * (1 + zsq) * eps2(l, p.logB_gsw, p.W)
    rest = kr * N - p.n
    return d * per_block, ks_var(p, rest, rest / 2.0, balanced=True)


def ccs(p, parties=None):
    N, k, l, n = p.N, p.k, p.l_uni, p.n
    sr2 = (p.beta / 2.0**p.W) ** 2
    v, e2 = V(p.logB_uni), eps2(l, p.logB_uni, p.W)
    ramp = N**3 / 48.0
    br = 0.0
    for idx in range(parties or k):
        npm = idx + 1
        n1 = 0.5 * (1 + npm * N / 2.0) * e2                                             # s * sum_q delta_q ztilde_q
        n2 = l * sr2 * (N * v + npm * N * N * (v / 2.0 - 1.0 / 16) + npm * npm * ramp)      # sum_q sum_j D_j(c_q) e1_j ztilde_q  (e1_j shared by the q)
        n3 = l * sr2 * npm * N * N * (2.0 / 3) * v                                      # r * sum D_j(c_q) e2
        n4 = (2.0 * N / 3) * (npm + 1) * e2                                             # r * sum_q delta^v_q
        n5 = l * sr2 * N * ((npm + 1) * (v - 0.25) + (npm + 1) ** 2 / 4.0)              # sum_q sum_j D_j(v_q) e3_j  (e3_j shared)
        br += n * 2.0 * (n1 + n2 + n3 + n4 + n5)
    return br, ks_var(p, k * N, k * N / 2.0)


# ---- KMS: linear noise model, simulated -----------------------------------------------------------------------------
def negconv(a, b):
    """negacyclic product of two real coefficient vectors (float64 FFT: a noise model, not ciphertext arithmetic)"""
    N = len(a)
    tw = np.exp(1j * np.pi * np.arange(N) / N)
    return np.real(np.fft.ifft(np.fft.fft(a * tw) * np.fft.fft(b * tw)) * np.conj(tw))


def rot(x, t):
    N = len(x)
    t %= 2 * N
    y = np.roll(x, t % N)
    y[:t % N] *= -1
    return -y if t >= N else y


def digits(rng, logB, N):
    return rng.integers(-(1 << (logB - 1)), 1 << (logB - 1), N).astype(np.float64)


_fft_err_cache = {}


def float64_product_error(N, logB, W, ndig):
    """std (torus units) of the error of ONE Float64 negacyclic sum of ndig products digit polynomial x W-bit polynomial, against the
    exact integer result (tests/ref_numpy.py FFT = the reference's transform; Kronecker substitution = exact)"""
    key = (N, logB, W, ndig)
    if key in _fft_err_cache:
        return _fft_err_cache[key]
    import ref_numpy as R
    f = R.FFT(N, W)
    rng = np.random.default_rng(99)
    T = np.uint64 if W == 64 else np.uint32
    acc_t, acc_e = R.C.zeros(N // 2), [0] * N
    for _ in range(ndig):
        d = rng.integers(-(1 << (logB - 1)), 1 << (logB - 1), N)
        kpoly = rng.integers(0, 1 << 63, N, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, N, dtype=np.uint64) if W == 64 else rng.integers(0, 1 << 32, N, dtype=np.uint64)
        kw = kpoly.astype(T)
        acc_t = acc_t + f.fwd(d.astype(np.int64).astype(T)) * f.fwd(kw)
        ks = [int(x) for x in kw.astype(np.int64 if W == 64 else np.int32)]
        SH = 160
        A = sum(int(x) << (SH * i) for i, x in enumerate(d)); B = sum(x << (SH * i) for i, x in enumerate(ks))
        P = A * B
        half, mask = 1 << (SH - 1), (1 << SH) - 1
        out = [0] * (2 * N)
        for i in range(2 * N - 1):
            c = ((P + half) & mask) - half; out[i] = c; P = (P - c) >> SH
        acc_e = [acc_e[i] + out[i] - out[i + N] for i in range(N)]
    got = f.inv(acc_t).astype(np.uint64)
    want = np.array([v % (1 << W) for v in acc_e], dtype=np.uint64)
    diff = (got - want).astype(T).astype(np.int64 if W == 64 else np.int32).astype(np.float64) / 2.0**W
    _fft_err_cache[key] = float(diff.std())
    return _fft_err_cache[key]


def kms(p, trials=24, seed=1, block=False):
    N, k, n, W = p.N, p.k, p.n, p.W
    lg, bg, ll, bl, lu, bu = p.l_gsw, p.logB_gsw, p.l_lev, p.logB_lev, p.l_uni, p.logB_uni
    sr2 = (p.beta / 2.0**W) ** 2
    rng = np.random.default_rng(seed)
    u = lambda bits, size=N: (rng.random(size) - 0.5) * 2.0 ** (-bits)             # rounding error of a decomposition to `bits` bits
    LB = p.blk_len if block else 1
    # one CMux / block: 2 l LB products summed in the transform domain, times the monomial (|X^a - 1|, rms sqrt 2), ONE inverse
    s_fft1 = float64_product_error(N, bg, W, 2 * lg * LB) * math.sqrt(2.0)
    s_fft2 = float64_product_error(N, bl, W, ll)
    s_fft3 = float64_product_error(N, bu, W, lu)
    samples = []
    for _ in range(trials):
        zg = [rng.integers(0, 2, N).astype(np.float64) for _ in range(k)]           # gsw keys z'
        zu = [rng.integers(0, 2, N).astype(np.float64) for _ in range(k)]           # uni keys z
        total = np.zeros(N)
        for idx in range(k):
            rows = 1 if idx == 0 else ll
            # phase 1 (bootstrapping.jl:389-443 / :599-659): error of each RLEV row under z'
            if block:
                skey = np.zeros(n, dtype=int)
                for b in range(n // LB):
                    j = rng.integers(0, LB + 1)
                    if j: skey[b * LB + j - 1] = 1
            else:
                skey = rng.integers(0, 2, n)
            e_rows = []
            for r in range(rows):
                err = np.zeros(N)
                nst = n // LB
                for st in range(nst):
                    Delta = u(lg * bg) + negconv(u(lg * bg), zg[idx])                  # one decomposition per step / block
                    for q in range(LB):
                        a = int(rng.integers(1, 2 * N))
                        nu = rng.normal(0.0, math.sqrt(2 * lg * N * V(bg) * sr2), N) if sr2 > 0 else 0.0
                        if skey[st * LB + q]:
                            err = rot(err, a) - (rot(Delta, a) - Delta)
                        if sr2 > 0:
                            err = err + rot(nu, a) - nu
                    err = err + rng.normal(0.0, s_fft1, N) + negconv(rng.normal(0.0, s_fft1, N), zg[idx])   # Float64 error of the step's two output polynomials: b + a z'
                e_rows.append(err)
            # phase 2 merge (bootstrapping.jl:448-558)
            if idx == 0:
                d0 = np.full(N, float(1 << (bl - 3)))                                 # digit 0 of the test vector +-1/8
                contrib = negconv(d0, e_rows[0])
            else:
                contrib = np.zeros(N)
                for q in range(idx + 1):
                    Eq = -u(ll * bl) + rng.normal(0.0, s_fft2, N) + negconv(rng.normal(0.0, s_fft2, N), zg[idx])   # rounding of c_q; Float64 error of x_q + y_q z'
                    for j in range(ll):
                        Eq = Eq + negconv(digits(rng, bl, N), e_rows[j])
                    contrib += Eq if q == 0 else negconv(Eq, zu[q - 1])
            # relinearisation (hybrid product with rlk, crs, public keys): rounding terms (+ key noise, white approximation)
            for q in range(idx + 1):
                t1 = negconv(u(lu * bu), zg[idx])
                contrib -= t1 if q == 0 else negconv(t1, zu[q - 1])
            r3 = rng.integers(-1, 2, N).astype(np.float64)
            contrib -= negconv(r3, u(lu * bu))
            for q in range(idx + 2):                                                  # Float64 error of the relinearised polynomials (lu digit products each)
                w3 = rng.normal(0.0, s_fft3, N)
                contrib += w3 if q == 0 else negconv(w3, zu[min(q - 1, k - 1)])
            if sr2 > 0:
                var_key = lu * N * V(bu) * sr2 * (1 + idx * N / 2.0 + idx * 2.0 * N / 3 + 1)
                contrib += rng.normal(0.0, math.sqrt(var_key), N)
            total += contrib
        samples.append(total)
    e = np.concatenate(samples)
    rest = (N - n) if block else N
    return float(e.var() + e.mean() ** 2), ks_var(p, k * rest, k * rest / 2.0, balanced=block)


def predict(p):
    if p.scheme == mk.CGGI: br, ks = cggi(p)
    elif p.scheme == mk.LMSS: br, ks = lmss(p)
    elif p.scheme == mk.CCS: br, ks = ccs(p)
    elif p.scheme == mk.KMS: br, ks = kms(p)
    else: br, ks = kms(p, block=True)
    return math.sqrt(br), math.sqrt(ks), math.sqrt(br + ks)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*")
    ap.add_argument("--measured", default=None, help="JSON lines of tools/noise_measure.py")
    args = ap.parse_args()
    meas = {}
    if args.measured and os.path.exists(args.measured):
        for ln in open(args.measured):
            d = json.loads(ln)
            if d["variant"] == "as shipped":
                meas[d["set"]] = d
    names = args.names or ["CGGIparam", "CGGI_N1024_l2", "Blockparam", "Blockparam_k2", "CCS2party", "CCS4party", "CCS8party", "CCS16party",
                           "CCS8party_N2048", "KMS2party", "KMS2party_N1024_l2", "KMS4party", "KMS8party", "KMS2partyblock"]
    print("Output phase error of a NAND whose inputs involve every party: predicted (tools/noise_theory.py) beside measured on the engine")
    print("(tools/noise_measure.py, MI355X; the engine is bit-identical to the oracle).  `measured` = the k-party fold's last level where the")
    print("final level has wrong gates (their wrapped phases inflate a standard deviation), else the final level.  Margin = 1/8.\n")
    print("| set | predicted sigma (blind rotation / key switch / total) | measured sigma | measured / predicted | margin / predicted sigma | wrong gates measured (final level) |")
    print("|---|---|---|---|---|---|")
    for nm in names:
        p = getattr(mk, nm)
        br, ks, tot = predict(p)
        m = meas.get(nm)
        mv = None
        if m:
            mv = m["sigma_after_parties"][-1] if (m["fails"] > 0 and m["sigma_after_parties"]) else m["sigma"]
        ms = f"{mv:.4f}" if m else "-"
        ratio = f"{mv / tot:.2f}" if m else "-"
        wrong = f"{m['fails']} / {m['gates']}" if m else "-"
        print(f"| {nm} | {br:.4f} / {ks:.4f} / **{tot:.4f}** | {ms} | {ratio} | {0.125 / tot:.1f} | {wrong} |", flush=True)
