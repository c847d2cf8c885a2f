"""One-off full-size parity runs for the large reference parameter sets (params.jl:23-125): GPU batch vs the oracle."""
import sys, time, json
import numpy as np
sys.path.insert(0, 'tests')
from helpers import *
names = sys.argv[1:] or ["CCS4party", "CCS16party", "KMS8party", "KMS16party"]
res = {}
for name in names:
    p = getattr(mk, name)
    t0 = time.time(); crs, keys = keygen(p, 3); t1 = time.time()
    sg = gpu_scheme(p, crs, keys); t2 = time.time()
    so = oracle_scheme(p, crs, keys); t3 = time.time()
    # the reference's own test shape (test/KMS.jl:23-37): one fresh bit per party, folded through gates, one more
    # bootstrap -- the later gates involve every party's rotation / hybrid product
    B = 4
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2, (p.k, B)).astype(bool)
    cts = [np.stack([mk.lwe_ith_encrypt(int(bits[i, j]), i, keys[i], p, deterministic_seed=1000 * i + j) for j in range(B)]) for i in range(p.k)]
    res_g, mres, ok, t_gpu, t_ora = cts[0], bits[0].copy(), True, 0.0, 0.0
    for i in range(1, p.k):
        op = int(rng.integers(0, 6))
        t4 = time.time(); nxt = sg.gate(op, res_g, cts[i]); t5 = time.time()
        ref = so.gate_batch(op, res_g, cts[i], threads=B); t6 = time.time()
        ok &= bool(np.array_equal(nxt, ref)); t_gpu += t5 - t4; t_ora += t6 - t5
        res_g = nxt; mres = GATE_FUNCS[op](mres, bits[i])
    fin = res_g.copy(); sg.bootstrapping_(fin)
    ok &= bool(np.array_equal(fin, np.stack([so.bootstrap(res_g[j]) for j in range(B)])))
    dec = mk.lwe_decrypt(fin, keys, p)
    res[name] = dict(bitexact=ok, decrypt_ok=bool(np.array_equal(dec, mres)), gates=p.k - 1,
                     keygen_s=round(t1 - t0, 2), gpu_load_s=round(t2 - t1, 2), oracle_load_s=round(t3 - t2, 2), gpu_fold_s=round(t_gpu, 3), oracle_fold_s=round(t_ora, 2))
    print(name, res[name], flush=True)
    sg.close(); del so, keys
json.dump(res, open('gpurun_out/fullsize_parity.json', 'w'), indent=1)
