"""usage (GPU box): python3 tools/fuzz_fx.py [rounds]  -- random shapes through the two MKT_ARITH_EXACT implementations of the RLWE-length-1 blind rotation
(CGGI on the 32-bit ring, every phase-1 row of KMS on the 64-bit ring): the Float64 pipe (fx_exact.hip, exact_impl=1) and the integer NTT (ntt_exact.hip,
exact_impl=0) must return the same words for RANDOM accumulators (full-range words: every digit pattern, not only test vectors), ragged batches, mask words
0 / N / 2N, N = 128 .. 4096, gadget length 2 / 3 and every base the bound certifies; at N <= 256 also the big-integer restatement (tests/ref_exact.py).
A shape whose keys the bound does not certify must fall back to the integer NTT (kernel name), never run the Float64 pipe.  The parity tests pin fixed
shapes (tests/test_gpu_fx.py); this walks around them."""
import os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np
import mktfhe_amd as mk
from mktfhe_amd._lib import MktError
from helpers import keygen, gpu_scheme

GADGETS = {32: [(2, 6), (2, 8), (2, 10), (2, 12), (3, 6), (3, 9), (3, 10)], 64: [(2, 12), (2, 14), (2, 16), (2, 20), (3, 10), (3, 12), (3, 14), (3, 16)]}      # (the widest: beyond the bound at large N)


def run(rounds=12, seed=61, log=print):
    """-> (mismatches, rounds on the Float64 pipe, rounds that fell back); a fixed-seed slice runs as a -m gpu test (tests/test_gpu_fuzz.py)"""
    import ref_exact as RX
    rng = np.random.default_rng(seed)
    bad = on_fx = fell_back = 0
    for it in range(rounds):
        kms = bool(rng.integers(0, 2))
        W = 64 if kms else 32
        N = 1 << int(rng.integers(7, 13))
        l, logB = GADGETS[W][int(rng.integers(0, len(GADGETS[W])))]
        n = int(rng.integers(3, 9)) if N <= 1024 else int(rng.integers(3, 6))
        B = int(rng.integers(1, 10))
        if kms:
            k = int(rng.choice([2, 3])) if N <= 1024 else 2
            p = mk.KMS2party.scaled(n=n, N=N, k=k, l_gsw=l, logB_gsw=logB)
        else:
            p = mk.CGGIparam.scaled(n=n, N=N, l_gsw=l, logB_gsw=logB)
        tag = f"round {it} {p.name} N={N} n={n} k={p.k} l={l} logB={logB} B={B}"
        crs, keys = keygen(p, int(rng.integers(1, 10000)))
        try:
            sx = gpu_scheme(p, crs, keys, arith=mk.ARITH_EXACT)
        except MktError as e:
            log(f"{tag}: refused at setup ({str(e)[:60]})"); continue
        try:
            sx.set_option("exact_impl", 1)
            certified = sx.get_metric("fx_available") == 1.0
            at = rng.integers(0, 2 * N + 1, (B, p.lwe_len - 1), dtype=np.int64).astype(np.uint32)
            at[0, :3] = [0, 2 * N, N]
            shape = (B, p.k + 1, N) if kms else (B, 2, N)
            acc0 = rng.integers(0, 1 << 63, shape, dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, shape).astype(np.uint64)
            acc0 = acc0.astype(p.ring_dtype)                      # (the 32-bit ring keeps the low words: uniform again)
            a1 = sx.blindrotate_(at, acc0.copy()); k1 = sx.last_kernel_name()
            rows1 = sx.kms_phase1(at).view(np.uint64).copy() if kms else None
            sx.set_option("exact_impl", 0)
            a0 = sx.blindrotate_(at, acc0.copy()); k0 = sx.last_kernel_name()
            rows0 = sx.kms_phase1(at).view(np.uint64) if kms else None
            ok = np.array_equal(a0, a1) and (not kms or np.array_equal(rows0, rows1))
            ok = ok and ("fx_" in k1) == certified and "fx_" not in k0
            if ok and N <= 256:
                want = RX.kms_blindrotate(p, keys, crs, at[0], acc0[0]) if kms else RX.blindrotate(p, keys[0].brk, at[0], acc0[0])
                ok = np.array_equal(np.asarray(a1[0]).astype(np.uint64).reshape(-1), np.asarray(want).astype(np.uint64).reshape(-1))
            on_fx += certified; fell_back += not certified
            bad += not ok
            log(f"{tag}: {'ok' if ok else 'MISMATCH'}  [{k1} | {k0}]  bound {sx.get_metric('fx_bound'):.3f}" + ("" if certified else "  (not certified: integer NTT serves)"))
        except MktError as e:
            log(f"{tag}: refused ({str(e)[:80]})")
        finally:
            sx.close()
    return bad, on_fx, fell_back


if __name__ == "__main__":
    bad, on_fx, fb = run(int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(os.environ.get("SEED", "61")), lambda m: print(m, flush=True))
    print("fuzz_fx:", "OK" if bad == 0 else f"{bad} MISMATCHES", f"({on_fx} shapes on the Float64 pipe, {fb} fell back to the integer NTT)")
    sys.exit(1 if bad else 0)
