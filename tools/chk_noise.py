import sys, numpy as np
sys.path.insert(0,'tests')
from helpers import *
for p in (mk.KMS2party_N1024_l2, mk.KMS2party):
    crs, keys = keygen(p, 12)
    sg = gpu_scheme(p, crs, keys)
    B = 1024
    rng = np.random.default_rng(13)
    bits = rng.integers(0, 2, 2 * B).astype(bool)
    uniq = encrypt_bits(p, keys, bits[:128], seed=7000)
    idx = rng.integers(0, 128, 2 * B); idx[:128] = np.arange(128)
    c = uniq[idx]; bb = bits[:128][idx]
    out = sg.gate(0, c[:B], c[B:])
    got = mk.lwe_decrypt(out, keys, p)
    want = ~(bb[:B] & bb[B:])
    bad = np.nonzero(got != want)[0]
    print(p.name, 'mismatches', len(bad), bad[:10])
    so = oracle_scheme(p, crs, keys)
    if len(bad):
        ref = so.gate_batch(0, c[bad[:8]], c[B + bad[:8]], threads=8)
        print('  oracle equals gpu on failing gates:', np.array_equal(ref, out[bad[:8]]))
    # phase distance of outputs
    ph = out[:, -1].astype(np.int64)
    for i, kk in enumerate(keys):
        ph = (ph + (out[:, i*p.n:(i+1)*p.n].astype(np.int64) * kk.lwekey.astype(np.int64)).sum(1)) % (1 << 32)
    ph = np.where(ph >= 1 << 31, ph - (1 << 32), ph) / 2.0**32
    err = np.abs(np.abs(ph) - 0.125)
    print('  output phase error: max %.4f  mean %.4f' % (err.max(), err.mean()))
