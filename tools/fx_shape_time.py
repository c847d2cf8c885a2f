"""Blind-rotation time of the EXACT implementations (integer NTT / Float64 pipe) at an arbitrary shape derived from a named parameter set:
    python tools/fx_shape_time.py KMS2party [field=value ...] [--batch 1024] [--impl 0,1]
e.g.  python tools/fx_shape_time.py KMS2party l_gsw=2 logB_gsw=14        (which of N = 2048 and l = 3 costs the Float64 pipe its rate there)
Timing only: random accumulator-side inputs (mod-switched words drawn at random), keys generated on the device; nothing is decrypted."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mktfhe_amd as mk
import torch

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = dict(a.lstrip("-").split("=") for a in sys.argv[1:] if a.startswith("--") and "=" in a)
B = int(opts.get("batch", 1024))
impls = [int(v) for v in opts.get("impl", "0,1").split(",")]
p = getattr(mk, args[0])
over = {k: int(v) for k, v in (a.split("=") for a in args[1:])}
if over:
    p = p.scaled(**over)
if p.multikey:
    crs = mk.CRS(p, 1)
    keys = [mk.party_keygen(crs, p, party=i, secrets_only=True, deterministic_seed=1) for i in range(p.k)]
    sch = mk.setup(p, keys=keys, a=crs, device=0, arith=mk.ARITH_EXACT)
else:
    keys = [mk.PartyKeys(p, secrets_only=True, deterministic_seed=1)]
    sch = mk.setup(p, keys=keys[0], device=0, arith=mk.ARITH_EXACT)[1]
rng = np.random.default_rng(3)
at = torch.from_numpy(rng.integers(1, 2 * p.N, (B, p.lwe_len - 1), dtype=np.int64).astype(np.uint32).view(np.int32)).cuda()
acc = torch.zeros((B, 1 + (p.k if p.multikey else p.k), p.N), dtype=torch.int64 if p.W == 64 else torch.int32, device="cuda")
acc[:, 0, :] = 1 << (p.W - 3)
for impl in impls:
    sch.set_option("exact_impl", impl)
    a = acc.clone()
    sch.blindrotate_(at, a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        sch.blindrotate_(at, a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    print(f"{p.name} N={p.N} l={p.l_gsw} logB={p.logB_gsw} W={p.W} batch {B} exact_impl={impl}: {dt * 1e3:8.2f} ms per blind rotation of the batch  ({B / dt:9.0f} /s)  kernel {sch.last_kernel_name()}  fx_available {sch.get_metric('fx_available')}", flush=True)
