"""usage (GPU box): python3 tools/soak_exact.py [iterations]  -- randomized soak of the MKT_ARITH_EXACT transform-level entry points against the
pure-Python restatement (tests/ref_ntt.py) and the oracle's schoolbook product: random and adversarial words (all residues p - 1, alternating
extremes, single spikes), N = 32 .. 256, both ring widths.  Exercises the lazy ranges of the butterflies ([0, 4p) / [0, 2p)), the input folding
and the CRT sign test at their edges."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
import mktfhe_amd as mk
import ref_ntt as R
from helpers import O

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(2026)
bad = 0
for N in (32, 64, 128, 256):
    for W in (32, 64):
        p = mk.CGGIparam.scaled(n=8, N=N, W=W)
        ex = mk.Scheme(p, arith=mk.ARITH_EXACT)
        dt = p.ring_dtype
        for it in range(iters):
            B = 6
            polys = rng.integers(0, 1 << 63, (B, N), dtype=np.uint64).astype(np.uint64)
            if W == 32:
                polys &= np.uint64(0xFFFFFFFF)
            polys[1] = (1 << (W - 1)) - 1 if it % 2 else (1 << (W - 1))                   # all extreme
            polys[2, ::2] = (1 << (W - 1)); polys[2, 1::2] = (1 << (W - 1)) - 1              # alternating extremes
            polys[3] = 0; polys[3, rng.integers(0, N)] = (1 << W) - 1                        # a spike of -1
            for q in R.PRIMES:                                                              # words congruent to p - 1 and p and 2p mod 2^W
                polys[4, rng.integers(0, N)] = (q - 1) % (1 << W)
                polys[4, rng.integers(0, N)] = (2 * q) % (1 << W)
            pw = polys.astype(dt)
            t = ex.transform_fwd(pw).view(np.uint64)
            for b in range(B):
                if [int(v) for v in t[b]] != R.fwd(pw[b], W):
                    bad += 1; print('fwd mismatch', N, W, it, b)
            # inverse of arbitrary canonical residues, incl. all p - 1 and the CRT boundary (P - 1) / 2, (P + 1) / 2
            res = np.zeros((B, N), dtype=np.uint64)
            for b in range(B):
                a1 = rng.integers(0, R.PRIMES[0], N, dtype=np.uint64); a2 = rng.integers(0, R.PRIMES[1], N, dtype=np.uint64)
                if b == 0:
                    a1[:] = R.PRIMES[0] - 1; a2[:] = R.PRIMES[1] - 1
                res[b] = a1 | (a2 << np.uint64(32))
            back = ex.transform_inv(res.view(np.complex128))
            for b in range(3):
                if [int(v) for v in back[b]] != R.inv([int(v) for v in res[b]], W):
                    bad += 1; print('inv mismatch', N, W, it, b)
            for logB in (2, 7, 16):
                a = rng.integers(-(1 << (logB - 1)), 1 << (logB - 1), (B, N)).astype(np.int64)
                a[1] = -(1 << (logB - 1))
                aw = a.astype(np.uint64).astype(dt) if W == 64 else (a & 0xFFFFFFFF).astype(np.uint32)
                got = ex.exact_polymul(aw, pw)
                for b in range(B):
                    ref = O.negacyclic(aw[b].astype(np.uint64) & np.uint64((1 << W) - 1), pw[b].astype(np.uint64), W)
                    if not np.array_equal(got[b].astype(np.uint64), ref):
                        bad += 1; print('polymul mismatch', N, W, it, logB, b)
        ex.close()
        print('N', N, 'W', W, 'done, mismatches so far', bad, flush=True)
# the worst case the modulus admits: N = 4096, every digit -2^15, every centered piece -2^31 -> coefficient N - 1 of each piece product
# is 4096 * 2^46 = 2^58 (P / 2 = 2^58.9998); also the mirrored signs
p = mk.CGGIparam.scaled(n=8, N=4096, W=64)
ex = mk.Scheme(p, arith=mk.ARITH_EXACT)
for sa, sb in ((-(1 << 15), 0x8000000080000000), ((1 << 15) - 1, 0x8000000080000000), (-(1 << 15), 0x7FFFFFFF7FFFFFFF)):
    aw = np.full((1, 4096), sa, dtype=np.int64).astype(np.uint64)
    bw = np.full((1, 4096), sb, dtype=np.uint64)
    got = ex.exact_polymul(aw, bw)
    ref = O.negacyclic(aw[0], bw[0], 64)
    if not np.array_equal(got[0].astype(np.uint64), ref):
        bad += 1; print('worst-case polymul mismatch', sa, hex(sb))
ex.close()
print('worst-case products at N = 4096 done, mismatches so far', bad)
print('soak_exact:', 'OK' if bad == 0 else f'{bad} MISMATCHES')
sys.exit(1 if bad else 0)
