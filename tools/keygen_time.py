"""Host vs device generation of a party's large keys (bootstrapping + key-switching key), wall time per party."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mktfhe_amd as mk
for p in (mk.CGGIparam, mk.KMS2party, mk.KMS4party, mk.CCS8party, mk.KMS16party):
    crs = mk.CRS(p, 1) if p.multikey else None
    t0 = time.perf_counter(); kh = mk.party_keygen(crs, p, deterministic_seed=1, party=0); th = time.perf_counter() - t0
    t0 = time.perf_counter(); ks = mk.party_keygen(crs, p, deterministic_seed=1, party=0, secrets_only=True); ts = time.perf_counter() - t0
    sch = mk.Scheme(p, device=0)
    if p.multikey: sch.load_crs(crs)
    t0 = time.perf_counter(); sch.load_party(0, kh); sch.synchronize(); tu = time.perf_counter() - t0
    sch.keygen_device(0, ks); sch.synchronize()
    t0 = time.perf_counter(); sch.keygen_device(0, ks); sch.synchronize(); td = time.perf_counter() - t0
    print(f"{p.name:12s} host keygen {th*1e3:8.1f} ms + upload/pre-transform {tu*1e3:7.1f} ms | secrets {ts*1e3:6.1f} ms + device keygen {td*1e3:7.1f} ms", flush=True)
    sch.close()
