import sys, os, hashlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import mktfhe_amd as mk
import bench as BN
wl, B = sys.argv[1], int(sys.argv[2])
p = getattr(mk, BN.WORKLOADS[wl][0])
dev = torch.device("cuda", 0)
crs, keys, sch = BN.make_scheme(mk, p, 0, False, mk.ARITH_F64REF)
bits, x, y = BN.make_inputs(mk, torch, p, keys, sch, B, 0, dev, "mixed")
out = torch.empty_like(x)
mk.NAND(x, y, sch, out=out)
torch.cuda.synchronize()
o = out.cpu().numpy()
dec = mk.lwe_decrypt(o.view(np.uint32), keys if p.multikey else keys[0], p)
print(wl, B, "KS_PAIR", os.environ.get("MKT_KS_PAIR"), "sha", hashlib.sha256(o.tobytes()).hexdigest()[:16], "wrong", int(np.count_nonzero(dec != ~(bits[:B] & bits[B:]))))
