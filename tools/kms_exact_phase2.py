"""Is the KMS blind rotation's output noise the algorithm's or the Float64 transform's?  (CPU only.)
Runs bootstrapping.jl:369-558 twice on the same noise-free keys and the same phase-1 rows: phase 2 (:448-558) once with the
reference's Float64 transforms (tests/ref_numpy.py, bit-identical to the oracle and the engine) and once with EXACT integer
negacyclic products (Kronecker substitution on Python integers) -- same decompositions, same operation order.  Prints the phase
error of the rotated test vector (every coefficient is a sample) for both.
  python tools/kms_exact_phase2.py [NAME] [n]"""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import mktfhe_amd as mk
import ref_numpy as R

name = sys.argv[1] if len(sys.argv) > 1 else "KMS2party_N1024_l2"
p = getattr(mk, name).scaled(alpha=0.0, beta=0.0)
if len(sys.argv) > 2:
    p = p.scaled(n=int(sys.argv[2]))
N, k, W, n = p.N, p.k, p.W, p.n
crs = mk.CRS(p, 12)
keys = [mk.party_keygen(crs, p, deterministic_seed=12, party=i) for i in range(k)]
S = R.Scheme(p, crs, keys)
f = S.f
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 5)
ta = rng.integers(0, 2 * N + 1, k * n).astype(np.uint32)
T = np.uint64
e8 = T(1) << T(61)
tv = np.full(N, e8, dtype=T)
zu = [np.asarray(kk.ringkey(1)).astype(np.int64) for kk in keys]
MOD = 1 << W


def to_signed_list(w):
    return [int(x) for x in w.astype(np.int64)]


def negacyclic(a, b):                      # exact product of two signed integer coefficient lists mod X^N + 1
    SH = 200
    A = sum(int(x) << (SH * i) for i, x in enumerate(a))
    B = sum(int(x) << (SH * i) for i, x in enumerate(b))
    P = A * B
    half, mask = 1 << (SH - 1), (1 << SH) - 1
    out = [0] * (2 * N)
    for i in range(2 * N - 1):
        c = ((P + half) & mask) - half
        out[i] = c
        P = (P - c) >> SH
    return [out[i] - out[i + N] for i in range(N)]


def phase_err(acc):
    ph = acc[0].copy()
    for i in range(k):
        a, z = acc[1 + i], zu[i]
        out = np.zeros(N, dtype=T)
        for j in np.nonzero(z)[0]:
            r = np.roll(a, j).copy(); r[:j] = (~r[:j]) + T(1); out += r
        ph += out
    fr = ph.astype(np.int64).astype(np.float64) / 2.0**64
    return np.abs(fr) - 0.125


lev = [S.phase1(i, ta[i * n:(i + 1) * n]) for i in range(k)]

# ---- Float64 phase 2 (the reference's arithmetic): ref_numpy's own code path with these rows
S.phase1 = lambda party, t, _lev=lev: _lev[party]
acc_f = S.blindrotate_kms(ta, [tv.copy()] + [np.zeros(N, dtype=T) for _ in range(k)])
e = phase_err(acc_f)
print(f"{name} n={n}: Float64 phase 2: rotated test vector error std {e.std():.5f} max {np.abs(e).max():.4f}", flush=True)

# ---- exact phase 2: the same steps on integer polynomials
levw = [[[to_signed_list(f.inv(r[q].copy())) for q in range(2)] for r in lev[i]] for i in range(k)]
rk = lambda arr: [to_signed_list(np.ascontiguousarray(x).astype(np.uint64)) for x in arr]
crs_i = rk(np.asarray(crs).reshape(p.l_uni, N))
P = [dict(pub=rk(kk.pubkey.reshape(p.l_uni, N)), rlk_d=rk(kk.rlk_d.reshape(p.l_uni, N)),
          rlk_f=[rk(x) for x in kk.rlk_f.reshape(p.l_uni, 2, N)]) for kk in keys]
dig = lambda w, l, logB: [to_signed_list(d) for d in R.decomp_poly(w, l, logB, W)]
add = lambda x, y: [a + b for a, b in zip(x, y)]
sub = lambda x, y: [a - b for a, b in zip(x, y)]
words = lambda x: np.array([v % MOD for v in x], dtype=np.uint64)
zero = [0] * N
zg = [[int(v) for v in np.asarray(kk.ringkey(0))] for kk in keys]
zui = [[int(v) for v in np.asarray(kk.ringkey(1))] for kk in keys]
sk = [kk.lwekey.astype(np.int64) for kk in keys]
def frac(x):        # signed integer list -> torus fractions of the value mod 2^W
    return np.array([((v + (MOD >> 1)) % MOD) - (MOD >> 1) for v in x], dtype=np.float64) / float(MOD)
def rot(x, t):      # x * X^t, t in [0, 2N)
    out = [0] * N
    for i, v in enumerate(x):
        j = i + t
        sgn = 1
        while j >= N: j -= N; sgn = -sgn
        out[j] = sgn * v
    return out
acc = [tv.copy()] + [np.zeros(N, dtype=T) for _ in range(k)]
ll, lu = p.l_lev, p.l_uni
for idx in range(k):
    tb = dig(acc[0], ll, p.logB_lev)
    tav = [dig(acc[1 + i], ll, p.logB_lev) for i in range(idx)]
    it = 1 if idx == 0 else ll
    tx = [list(zero) for _ in range(k + 1)]; ty = [list(zero) for _ in range(k + 1)]
    for i in range(it):
        tx[0] = add(tx[0], negacyclic(tb[i], levw[idx][i][0])); ty[0] = add(ty[0], negacyclic(tb[i], levw[idx][i][1]))
    for i in range(idx):
        for j in range(it):
            tx[1 + i] = add(tx[1 + i], negacyclic(tav[i][j], levw[idx][j][0])); ty[1 + i] = add(ty[1 + i], negacyclic(tav[i][j], levw[idx][j][1]))
    t_idx = int((ta[idx * n:(idx + 1) * n].astype(np.int64) * sk[idx]).sum() % (2 * N))
    for q in range(idx + 1):                      # LEV multiplication error: x_q + y_q z' - c_q X^t
        cq = to_signed_list(acc[q])
        Eq = sub(add(tx[q], negacyclic(ty[q], zg[idx])), rot(cq, t_idx))
        amp = Eq if q == 0 else negacyclic(Eq, zui[q - 1])
        print(f"  merge {idx}: LEV-multiplication error of polynomial {q}: std {frac(Eq).std():.3e}; as it enters the phase (times z_{q-1} for q>0): std {frac(amp).std():.3e} mean {frac(amp).mean():.3e}", flush=True)
    x_keep, y_keep = [list(v) for v in tx], [list(v) for v in ty]
    yb = words(ty[0]); ya = [words(ty[1 + i]) for i in range(idx)]
    tb = dig(yb, lu, p.logB_uni); tav = [dig(ya[i], lu, p.logB_uni) for i in range(idx)]
    ty = [list(zero) for _ in range(k + 1)]
    for i in range(lu):
        ty[0] = add(ty[0], negacyclic(tb[i], P[idx]["rlk_d"][i]))
    for i in range(idx):
        for j in range(lu):
            ty[1 + i] = add(ty[1 + i], negacyclic(tav[i][j], P[idx]["rlk_d"][j]))
    tvv = list(zero)
    for i in range(lu):
        tvv = sub(tvv, negacyclic(tb[i], crs_i[i]))
    for i in range(idx):
        for j in range(lu):
            tvv = add(tvv, negacyclic(tav[i][j], P[i]["pub"][j]))
    v = words(tvv)
    dv = dig(v, lu, p.logB_uni)
    for i in range(lu):
        ty[0] = add(ty[0], negacyclic(dv[i], P[idx]["rlk_f"][i][0]))
        ty[1 + idx] = add(ty[1 + idx], negacyclic(dv[i], P[idx]["rlk_f"][i][1]))
    acc = [words(add(tx[q], ty[q])) for q in range(k + 1)]
    # hybrid-product error: phase(new acc) - sum_q (x_q + y_q z') ztilde_q
    ph = to_signed_list(acc[0])
    for i in range(idx + 1):
        ph = add(ph, negacyclic(to_signed_list(acc[1 + i]), zui[i]))
    want = add(x_keep[0], negacyclic(y_keep[0], zg[idx]))
    for q in range(1, idx + 1):
        want = add(want, negacyclic(add(x_keep[q], negacyclic(y_keep[q], zg[idx])), zui[q - 1]))
    hp = frac(sub(ph, want))
    print(f"  merge {idx}: hybrid-product (relinearisation) error: std {hp.std():.3e} mean {hp.mean():.3e}", flush=True)
e = phase_err(acc)
print(f"{name} n={n}: EXACT   phase 2: rotated test vector error std {e.std():.5f} max {np.abs(e).max():.4f}", flush=True)
