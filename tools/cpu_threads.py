"""CPU-baseline thread scaling of the C oracle on this host (diagnostic): gates/s at 16..256 threads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, 'tests')
import numpy as np
from helpers import *
for p in (mk.KMS2party_N1024_l2, mk.CGGIparam):
    crs, keys = keygen(p, 1)
    so = oracle_scheme(p, crs, keys)
    bits = np.random.default_rng(1).integers(0, 2, 1025).astype(bool)
    c = encrypt_bits(p, keys, bits, seed=1)
    x, y = c[:512], c[513:1025]
    for th in (16, 32, 64, 96, 128, 192, 256):
        n = min(512, 4 * th)
        t0 = time.perf_counter(); so.gate_batch(0, x[:n], y[:n], threads=th); dt = time.perf_counter() - t0
        print(p.name, 'threads', th, 'gates', n, '%.1f gates/s' % (n / dt), flush=True)
