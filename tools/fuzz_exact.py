"""usage (GPU box): python3 tools/fuzz_exact.py [rounds]  -- random small shapes and gadgets through the MKT_ARITH_EXACT gate paths (KMS, KMS_block,
CCS, CGGI, LMSS) against the big-integer restatement (tests/ref_exact.py), by calling the parity tests' bodies on shapes they do not pin.
Gadgets beyond the two-prime modulus must be refused (MKT_ERR_UNSUPPORTED), never evaluated."""
import os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np
import mktfhe_amd as mk
from mktfhe_amd._lib import MktError
import test_gpu_parity as T



def run(rounds=8, seed=91, log=print):
    """-> (failures, refused); a fixed-seed slice of this runs as a -m gpu test (tests/test_gpu_fuzz.py)"""
    rng = np.random.default_rng(seed)
    bad = refused = 0
    for it in range(rounds):
        N = int(rng.choice([64, 128, 256]))
        kind = int(rng.integers(0, 5))
        l, logB = [(2, 16), (3, 12), (4, 9), (2, 22), (3, 10), (2, 27)][int(rng.integers(0, 6))]
        try:
            if kind == 0:
                p = mk.KMS2party.scaled(n=int(rng.integers(3, 8)), N=N, l_gsw=l, logB_gsw=logB); fn = T.test_exact_mode_kms_gates
            elif kind == 1:
                d = int(rng.integers(1, 4)); p = mk.KMS2partyblock.scaled(n=3 * d, N=N, blk_d=d, l_gsw=l, logB_gsw=logB); fn = T.test_exact_mode_kms_gates
            elif kind == 2:
                p = mk.CCS2party.scaled(n=int(rng.integers(3, 8)), N=N, k=int(rng.integers(2, 4))); fn = T.test_exact_mode_ccs_gates
            elif kind == 3:
                p = mk.CGGIparam.scaled(n=int(rng.integers(4, 12)), N=N, l_gsw=min(l, 3), logB_gsw=min(logB, 10)); fn = T.test_exact_mode_cggi_gates
            else:
                d = int(rng.integers(2, 5)); L = int(rng.choice([2, 3, 4])); kk = int(rng.choice([1, 2]))      # any block length, RLWE length 1 / 2 (exact_blindrotate_kr_kernel)
                p = mk.Blockparam.scaled(n=L * d, N=N, blk_d=d, blk_len=L, k=kk, logB_gsw=7 if kk == 2 else 9); fn = T.test_exact_mode_cggi_gates
            fn(None, p)
            st = "ok"
        except MktError as e:
            st = "refused (gadget beyond the modulus)" if "EXACT" in str(e) else f"ERROR {e}"
            refused += "refused" in st; bad += "ERROR" in st
        except AssertionError as e:
            st = f"MISMATCH {str(e)[:120]}"; bad += 1
        log(f"round {it} {p.name} N={p.N} n={p.n} k={p.k} l={p.l_gsw} logB={p.logB_gsw}: {st}")
    return bad, refused


if __name__ == "__main__":
    bad, refused = run(int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(os.environ.get("SEED", "91")), lambda m: print(m, flush=True))
    print("fuzz_exact:", "OK" if bad == 0 else f"{bad} FAILURES", f"({refused} refused)")
    sys.exit(1 if bad else 0)
