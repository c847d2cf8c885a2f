"""Where the KMS output noise comes from (run on the GPU box): phase error of the phase-1 RLEV rows under the party's gsw key,
of the accumulator after the whole blind rotation under the uni keys (every coefficient of the rotated test vector is a
sample), and of the gate output after key switching.  python tools/kms_stage_noise.py [NAME] [alpha beta]"""
import sys
import numpy as np
sys.path.insert(0, 'tests')
from helpers import *   # noqa

name = sys.argv[1] if len(sys.argv) > 1 else "KMS2party_N1024_l2"
p = getattr(mk, name)
if len(sys.argv) > 3:
    p = p.scaled(alpha=float(sys.argv[2]), beta=float(sys.argv[3]))
crs, keys = keygen(p, 12)
sg = gpu_scheme(p, crs, keys)
k, N, n, B = p.nparty, p.N, p.n, 8
rng = np.random.default_rng(13)
bits = rng.integers(0, 2, 2 * B * k).astype(bool)
c = encrypt_bits(p, keys, bits, seed=7000)
acc, ab = c[0::k].copy(), bits[0::k].copy()
for i in range(1, k):
    acc = sg.gate(0, acc, c[i::k]); ab = ~(ab & bits[i::k])
x, y = acc[:B], acc[B:]
lin = np.stack([O.gate_linear(0, x[j], y[j]) for j in range(B)])
at, bt = sg.modswitch(lin)


def negacyclic_by_binary(a, z):           # a: uint64 [N], z: 0/1 [N] -> a * z mod (X^N + 1, 2^64)
    out = np.zeros(N, dtype=np.uint64)
    for i in np.nonzero(z)[0]:
        r = np.roll(a, i).copy()
        r[:i] = (~r[:i]) + np.uint64(1)
        out += r
    return out


def signed_frac(w):                        # uint64 -> torus fraction in [-1/2, 1/2)
    return w.astype(np.int64).astype(np.float64) / 2.0**64


zg = [np.asarray(kk.ringkey(0)).astype(np.int64) for kk in keys]     # gsw keys z'
zu = [np.asarray(kk.ringkey(1)).astype(np.int64) for kk in keys]     # uni keys z
s = [kk.lwekey.astype(np.int64) for kk in keys]
lev = sg.kms_phase1(at)                                              # [B][rtot][2][M]
rows = sg.transform_inv(lev.reshape(-1, N // 2)).reshape(B, -1, 2, N)
errs = {}
r0 = 0
for party in range(k):
    nrow = 1 if party == 0 else p.l_lev
    for r in range(nrow):
        e_all = []
        for j in range(B):
            b, a = rows[j, r0 + r, 0], rows[j, r0 + r, 1]
            ph = b + negacyclic_by_binary(a, zg[party])
            t = int((at[j, party * n:(party + 1) * n].astype(np.int64) * s[party]).sum() % (2 * N))
            want = np.zeros(N, dtype=np.uint64)
            g = np.uint64(1) << np.uint64(64 - (r + 1) * p.logB_lev)
            if t < N: want[t] = g
            else: want[t - N] = (~g) + np.uint64(1)
            e_all.append(signed_frac(ph - want))
        e = np.concatenate(e_all)
        print(f"phase 1, party {party} row {r}: error std {e.std():.3e} = 2^{np.log2(e.std() + 1e-300):.1f}  max {np.abs(e).max():.3e}", flush=True)
    r0 += nrow
# accumulator after the whole blind rotation
acc0 = np.stack([oracle_scheme(p, crs, keys).testvector(bt[j]) for j in range(1)]) if False else None
tv = np.zeros((B, k + 1, N), dtype=np.uint64)
for j in range(B):
    tb = int(bt[j]); e8 = np.uint64(1) << np.uint64(61); me = (~e8) + np.uint64(1)
    lo, hi = e8, me
    if tb > N: tb -= N; lo, hi = me, e8
    tv[j, 0, :tb] = lo; tv[j, 0, tb:] = hi
accr = sg.blindrotate_(at, tv.copy())
e_all = []
for j in range(B):
    ph = accr[j, 0].copy()
    for i in range(k):
        ph += negacyclic_by_binary(accr[j, 1 + i], zu[i])
    f = signed_frac(ph)
    e_all.append(np.abs(f) - 0.125)
e = np.concatenate(e_all)
print(f"after the blind rotation (phase 1 + phase 2), every coefficient: error std {e.std():.5f}  max {np.abs(e).max():.4f}", flush=True)
out = sg.keyswitch(accr)
ph = out[:, -1].astype(np.int64)
for i, kk in enumerate(keys):
    ph = (ph + (out[:, i * n:(i + 1) * n].astype(np.int64) * s[i]).sum(1)) % (1 << 32)
ph = np.where(ph >= 1 << 31, ph - (1 << 32), ph) / 2.0**32
print(f"after key switching ({B} samples): error std {(np.abs(ph) - 0.125).std():.5f}")
sg.close()
