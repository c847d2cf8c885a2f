"""usage (GPU box): MKT_ROT_BLKG=1|4 python3 tools/kr_time.py [k] [batch]  -- blind-rotation time of CGGIparam with RLWE length k (no shipped set has k > 1)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np, torch
from helpers import *
k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
p = mk.CGGIparam.scaled(k=k)
keys = [mk.PartyKeys(p, secrets_only=True, deterministic_seed=1)]
sch = mk.setup(p, keys=keys[0], device=0)[1]
bits = np.random.default_rng(4).integers(0, 2, 2 * B).astype(bool)
c = encrypt_bits(p, keys, bits, seed=40)
x = torch.from_numpy(c[:B].view(np.int32)).cuda(); y = torch.from_numpy(c[B:].view(np.int32)).cuda()
o = mk.NAND(x, y, sch); torch.cuda.synchronize()
sch.enable_timing(True)
for _ in range(3):
    o = mk.NAND(x, y, sch)
torch.cuda.synchronize()
ms, cnt = sch.kernel_ms(1)
dec = mk.lwe_decrypt(o.cpu().numpy().view(np.uint32), keys[0], p)
print('BLKG', os.environ.get('MKT_ROT_BLKG', 'default'), 'k', k, 'batch', B, 'rot ms %.2f' % (ms / cnt), 'decrypt ok', bool(np.array_equal(dec, ~(bits[:B] & bits[B:]))))
sch.close()
