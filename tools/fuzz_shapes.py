"""usage (GPU box): python3 tools/fuzz_shapes.py [rounds]  -- random small shapes of the rotation kernels that have more than one implementation
(block-binary with RLWE length 1 / 2 / 3, plain CMux with RLWE length 2 / 3), every stage and gate against the oracle under each forced
grouping (MKT_ROT_BLKG 1 / 2 / 4) with ragged batch sizes.  The parity tests pin fixed shapes; this walks around them."""
import os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np
import mktfhe_amd as mk
import test_gpu_parity as T



def run(rounds=6, seed=77, log=print):
    """-> number of mismatches; a fixed-seed slice of this runs as a -m gpu test (tests/test_gpu_fuzz.py)"""
    rng = np.random.default_rng(seed)
    bad = 0
    saved = os.environ.get("MKT_ROT_BLKG")
    for it in range(rounds):
        logN = int(rng.integers(6, 13))       # N = 64 .. 4096
        N = 1 << logN
        kind = int(rng.integers(0, 4))
        if kind == 0:      # block-binary, RLWE length 1
            L = int(rng.choice([2, 3, 4])); d = int(rng.integers(2, 6)); p = mk.Blockparam.scaled(n=L * d, N=N, blk_d=d, blk_len=L)
        elif kind == 1:    # block-binary, RLWE length 2 (block length 3: the grouped three-polynomial kernel)
            d = int(rng.integers(2, 6)); p = mk.Blockparam_k2.scaled(n=3 * d, N=N, blk_d=d)
        elif kind == 2:    # plain CMux, RLWE length 2 / 3
            k = int(rng.choice([2, 3])); p = mk.CGGIparam.scaled(n=int(rng.integers(5, 14)), N=min(N, 512) if k == 3 else N, k=k, l_gsw=2, logB_gsw=10)
        else:              # KMS_block, two parties
            d = int(rng.integers(2, 4)); p = mk.KMS2partyblock.scaled(n=3 * d, N=max(N, 128), blk_d=d)
        B = int(rng.integers(2, 10))   # (_stage_check looks at two ciphertexts of the KMS phase-1 rows)
        for G in ("1", "2", "4"):
            os.environ["MKT_ROT_BLKG"] = G      # read at context creation (inside _stage_check)
            try:
                T._stage_check(p, B=B, seed=int(rng.integers(1, 1000)))
                st = "ok"
            except AssertionError as e:
                st = f"MISMATCH {e}"; bad += 1
            log(f"round {it} {p.name} N={p.N} n={p.n} k={p.k} blk_len={p.blk_len} B={B} BLKG={G}: {st}")
    if saved is None:
        os.environ.pop("MKT_ROT_BLKG", None)
    else:
        os.environ["MKT_ROT_BLKG"] = saved
    return bad


if __name__ == "__main__":
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(os.environ.get("SEED", "77")), lambda m: print(m, flush=True))
    print("fuzz_shapes:", "OK" if bad == 0 else f"{bad} MISMATCHES")
    sys.exit(1 if bad else 0)
