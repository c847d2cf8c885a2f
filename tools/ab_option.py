"""usage (GPU box): python3 tools/ab_option.py --workload kms2_n1024 [--arith exact] --option rot_map --values 0 1 [--batch 1024] [--rounds 4]
Same-process, same-device A/B of one kernel-selection switch (mkt_set_option): ONE context, ONE set of inputs, the values
alternated round by round (A B A B ...) so that clock / thermal drift hits both alike.  Prints per value the mean and the
minimum of the blind-rotation and whole-step device times, and checks that the output words do not depend on the switch."""
import argparse, os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np, torch
import mktfhe_amd as mk
import bench as BN

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="kms2_n1024")
ap.add_argument("--arith", default="f64ref")
ap.add_argument("--option", required=True)
ap.add_argument("--values", type=int, nargs="+", required=True)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--inputs", default="mixed")
a = ap.parse_args()
p = getattr(mk, BN.WORKLOADS[a.workload][0])
arith = mk.ARITH_EXACT if a.arith == "exact" else mk.ARITH_F64REF
dev = torch.device("cuda", 0)
crs, keys, sch = BN.make_scheme(mk, p, 0, False, arith)
bits, x, y = BN.make_inputs(mk, torch, p, keys, sch, a.batch, 0, dev, a.inputs)
out = torch.empty_like(x)
res, ref = {v: [] for v in a.values}, None
for r in range(a.rounds):
    for v in a.values:
        sch.set_option(a.option, v)
        mk.NAND(x, y, sch, out=out)                      # warm-up under this value
        torch.cuda.synchronize()
        sch.enable_timing(True)
        for _ in range(a.steps):
            mk.NAND(x, y, sch, out=out)
        rot, n = sch.kernel_ms(1); whole, nw = sch.kernel_ms(0)
        sch.enable_timing(False)
        res[v].append((rot / max(n, 1), whole / max(nw, 1)))
        o = out.cpu().numpy()
        if ref is None:
            ref = o.copy()
        assert np.array_equal(o, ref), f"output words depend on {a.option}={v}"
for v in a.values:
    rr = np.array(res[v])
    print(f"{a.workload} {a.arith} batch {a.batch} {a.option}={v}: rotation mean {rr[:,0].mean():.3f} min {rr[:,0].min():.3f} ms | step mean {rr[:,1].mean():.3f} min {rr[:,1].min():.3f} ms | kernel {sch.last_kernel_name()}", flush=True)
dec = mk.lwe_decrypt(ref.view(np.uint32), keys if p.multikey else keys[0], p)
print("wrong decryptions", int(np.count_nonzero(dec != ~(bits[:a.batch] & bits[a.batch:]))), "of", a.batch)
