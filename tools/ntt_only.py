"""usage: python3 tools/ntt_only.py [N] [launches]  -- only the EXACT batched transforms (ntt_fwd_kernel / ntt_inv_kernel), 4 GiB per launch:
the target of tools/pmc_ntt.sh"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import mktfhe_amd as mk
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device('cuda:0')
p = mk.KMS2party.scaled(n=8, N=N)
sx = mk.Scheme(p, device=0, arith=mk.ARITH_EXACT)
nb = (4 << 30) // (16 * N)
polys = torch.randint(-2**31, 2**31 - 1, (nb, 2 * N), dtype=torch.int32, device=dev).view(torch.int64)
tr = torch.empty((nb, N // 2), dtype=torch.complex128, device=dev)
back = torch.empty((nb, N), dtype=torch.int64, device=dev)
for direction in ("forward", "inverse"):
    fn = (lambda: sx.transform_fwd(polys, out=tr)) if direction == "forward" else (lambda: sx.transform_inv(tr, out=back))
    fn(); torch.cuda.synchronize()
    sx.enable_timing(True)
    for _ in range(reps):
        fn()
    ms, cnt = sx.kernel_ms(3)
    sx.enable_timing(False)
    print(direction, N, 'TB/s %.3f' % (nb * 16 * N / (ms / cnt * 1e-3) / 1e12))
sx.close()
