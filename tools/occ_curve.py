"""usage (GPU box): python3 tools/occ_curve.py [--arith exact] [--workload kms2_n1024] [--batches 85 170 341 512 1024]
Blind-rotation time against the number of resident workgroups: batches chosen so that the rotation grid is 1, 2, 4, 6 ... workgroups
per CU (one round), which separates what a wave costs alone from what co-resident waves hide."""
import argparse, os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np, torch
import mktfhe_amd as mk
import bench as BN
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="kms2_n1024")
ap.add_argument("--arith", default="exact")
ap.add_argument("--batches", type=int, nargs="+", default=[85, 170, 341, 512, 1024])
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
p = getattr(mk, BN.WORKLOADS[a.workload][0])
arith = mk.ARITH_EXACT if a.arith == "exact" else mk.ARITH_F64REF
dev = torch.device("cuda", 0)
crs, keys, sch = BN.make_scheme(mk, p, 0, False, arith)
bmax = max(a.batches)
bits, x, y = BN.make_inputs(mk, torch, p, keys, sch, bmax, 0, dev, "mixed")
for b in a.batches:
    out = torch.empty_like(x[:b])
    mk.NAND(x[:b], y[:b], sch, out=out); torch.cuda.synchronize()
    sch.enable_timing(True)
    for _ in range(a.steps): mk.NAND(x[:b], y[:b], sch, out=out)
    rot, n = sch.kernel_ms(1)
    sch.enable_timing(False)
    print(f"{a.workload} {a.arith} batch {b}: rotation {rot / max(n, 1):.3f} ms  ({b / (rot / max(n, 1)) * 1e3:.0f} gates/s of rotation)  kernel {sch.last_kernel_name()}", flush=True)
