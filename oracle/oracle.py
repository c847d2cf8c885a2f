"""ctypes binding of the CPU ORACLE (oracle/libmkt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py -- never by the product package (mktfhe_amd/).  See oracle/mkt_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

CGGI, LMSS, CCS, KMS, KMS_BLOCK = range(5)
NAND, AND, OR, XOR, XNOR, NOR = range(6)


class OraParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "scheme", "n", "N", "k", "W", "l_gsw", "logB_gsw", "l_lev", "logB_lev",
        "l_uni", "logB_uni", "f", "logD", "blk_len", "blk_d")]


def build(force=False):
    so = os.path.join(_HERE, "libmkt_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("mkt_oracle.c", "mkt_oracle.h")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmkt_oracle.so"], stdout=subprocess.DEVNULL)
    return so


_NATIVE = None


def use_native_build():
    """bench.py's cpu_baseline leg: compile the oracle for THIS host's cores (-O3 -march=native; still -ffp-contract=off, no
    fast-math: the same IEEE operation sequence, so the same bits) into a temporary directory and load that instead of the
    portable in-tree build, which has to run on whatever CPU the GPU box has.  Must be called before the first oracle call;
    returns the flags actually used (the portable build's if the native compile fails)."""
    global _NATIVE
    if _LIB is not None:
        return "already loaded"
    import tempfile
    out = os.path.join(tempfile.mkdtemp(prefix="mkt_oracle_"), "libmkt_oracle_native.so")
    flags = ["-O3", "-march=native", "-std=gnu11", "-fPIC", "-ffp-contract=off", "-fno-fast-math"]
    try:
        subprocess.check_call(["gcc"] + flags + ["-shared", "-o", out, os.path.join(_HERE, "mkt_oracle.c"), "-lquadmath", "-lpthread", "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _NATIVE = out
        return " ".join(flags)
    except (OSError, subprocess.CalledProcessError):
        return "-O3 -ffp-contract=off -fno-fast-math (portable in-tree build; native compile failed)"


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(_NATIVE or build())
        vp, i32, u32, u64, dbl = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_double
        L.ora_ffter_create.restype = vp; L.ora_ffter_create.argtypes = [i32, i32]
        L.ora_ffter_destroy.argtypes = [vp]
        L.ora_ffter_table.restype = C.POINTER(dbl); L.ora_ffter_table.argtypes = [vp, i32]
        L.ora_fft_fwd.argtypes = [vp, vp, vp]
        L.ora_fft_inv.argtypes = [vp, vp, vp]
        L.ora_fft_fwd_batch.argtypes = [vp, vp, vp, C.c_size_t]
        L.ora_fft_inv_batch.argtypes = [vp, vp, vp, C.c_size_t]
        L.ora_native.restype = u64; L.ora_native.argtypes = [dbl, i32]
        L.ora_divbits.restype = u64; L.ora_divbits.argtypes = [u64, i32, i32]
        L.ora_tp_muladd.argtypes = [vp, vp, vp, i32]
        L.ora_tp_mulsub.argtypes = [vp, vp, vp, i32]
        L.ora_tp_mul.argtypes = [vp, vp, vp, i32]
        L.ora_monomial.argtypes = [vp, i32, vp]
        L.ora_negacyclic_schoolbook.argtypes = [vp, vp, vp, i32, i32]
        L.ora_decomp_word.argtypes = [u64, i32, i32, i32, vp]
        L.ora_unbalanced_decomp_word.argtypes = [u64, i32, i32, i32, vp]
        L.ora_decomp_poly.argtypes = [vp, i32, i32, i32, i32, vp]
        L.ora_scheme_create.restype = vp; L.ora_scheme_create.argtypes = [C.POINTER(OraParams)]
        L.ora_scheme_destroy.argtypes = [vp]
        L.ora_scheme_ffter.restype = vp; L.ora_scheme_ffter.argtypes = [vp]
        L.ora_set_brk.argtypes = [vp, i32, vp]
        L.ora_set_ksk.argtypes = [vp, i32, vp]
        L.ora_set_rlk.argtypes = [vp, i32, vp, vp]
        L.ora_set_pubkey.argtypes = [vp, i32, vp]
        L.ora_set_crs.argtypes = [vp, vp]
        L.ora_gate_linear.argtypes = [i32, vp, vp, vp, i32]
        L.ora_not.argtypes = [vp, i32]
        L.ora_modswitch.argtypes = [vp, vp, vp, vp]
        L.ora_testvector.argtypes = [vp, u32, vp]
        L.ora_blindrotate.argtypes = [vp, vp, vp]
        L.ora_keyswitch.argtypes = [vp, vp, vp]
        L.ora_bootstrap.argtypes = [vp, vp]
        L.ora_gate.argtypes = [vp, i32, vp, vp, vp]
        L.ora_kms_phase1.restype = i32; L.ora_kms_phase1.argtypes = [vp, i32, vp, vp]
        L.ora_kms_phase2.argtypes = [vp, vp, vp]
        L.ora_acc_polys.restype = i32; L.ora_acc_polys.argtypes = [vp]
        L.ora_gate_batch.argtypes = [vp, i32, vp, vp, vp, C.c_size_t, i32]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a


class Ffter:
    """fft.jl:18-45 FFTransformer{Float64}(N, bits)"""

    def __init__(self, N, W, handle=None):
        self.N, self.M, self.W = N, N // 2, W
        self._own = handle is None
        self.h = lib().ora_ffter_create(N, W) if handle is None else handle

    def __del__(self):
        if getattr(self, "_own", False) and self.h:
            lib().ora_ffter_destroy(self.h)
            self.h = None

    def table(self, which):
        ptr = lib().ora_ffter_table(self.h, which)
        return np.ctypeslib.as_array(ptr, shape=(2 * self.M,)).copy().view(np.complex128)

    def fwd(self, p):
        """p: (..., N) uint64 ring words -> (..., M) complex128"""
        p = _c(p, np.uint64)
        B = p.size // self.N
        t = np.empty((B, self.M), dtype=np.complex128)
        lib().ora_fft_fwd_batch(self.h, _p(p), _p(t), B)
        return t.reshape(p.shape[:-1] + (self.M,))

    def inv(self, t):
        """t: (..., M) complex128 -> (..., N) uint64 (t is not modified)"""
        t = np.array(t, dtype=np.complex128, order="C", copy=True)
        B = t.size // self.M
        p = np.empty((B, self.N), dtype=np.uint64)
        lib().ora_fft_inv_batch(self.h, _p(t), _p(p), B)
        return p.reshape(t.shape[:-1] + (self.N,))

    def monomial(self, e):
        out = np.empty(self.M, dtype=np.complex128)
        lib().ora_monomial(self.h, int(e), _p(out))
        return out


def native(x, W):
    return int(lib().ora_native(float(x), W))


def divbits(a, bit, W):
    return int(lib().ora_divbits(int(a) & ((1 << 64) - 1), bit, W))


def decomp_word(a, l, logB, W):
    out = np.zeros(l, dtype=np.uint64)
    lib().ora_decomp_word(int(a), l, logB, W, _p(out))
    return out


def unbalanced_decomp_word(a, l, logB, W):
    out = np.zeros(l, dtype=np.uint64)
    lib().ora_unbalanced_decomp_word(int(a), l, logB, W, _p(out))
    return out


def decomp_poly(a, l, logB, W):
    a = _c(a, np.uint64)
    out = np.zeros((l, a.size), dtype=np.uint64)
    lib().ora_decomp_poly(_p(a), a.size, l, logB, W, _p(out))
    return out


def negacyclic(a, b, W):
    a, b = _c(a, np.uint64), _c(b, np.uint64)
    out = np.zeros_like(a)
    lib().ora_negacyclic_schoolbook(_p(a), _p(b), _p(out), a.size, W)
    return out


def gate_linear(op, x, y):
    x, y = _c(x, np.uint32), _c(y, np.uint32)
    out = np.empty_like(x)
    lib().ora_gate_linear(op, _p(x), _p(y), _p(out), x.size)
    return out


class Scheme:
    """The reference's scheme objects (scheme.jl:107-116, :168-179, :209-219, :256-265, :301-312)."""

    def __init__(self, params: OraParams):
        self.p = params
        self.h = lib().ora_scheme_create(C.byref(params))
        self.mk = params.scheme in (CCS, KMS, KMS_BLOCK)
        self.lwe_len = (params.k if self.mk else 1) * params.n + 1
        self.kacc = lib().ora_acc_polys(self.h)
        self.ffter = Ffter(params.N, params.W, handle=lib().ora_scheme_ffter(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_scheme_destroy(self.h)
            self.h = None

    def set_brk(self, party, brk_int):
        a = _c(brk_int, np.uint64); assert lib().ora_set_brk(self.h, party, _p(a)) == 0

    def set_ksk(self, party, ksk):
        a = _c(ksk, np.uint32); assert lib().ora_set_ksk(self.h, party, _p(a)) == 0

    def set_rlk(self, party, d_int, f_int):
        d, f = _c(d_int, np.uint64), _c(f_int, np.uint64)
        assert lib().ora_set_rlk(self.h, party, _p(d), _p(f)) == 0

    def set_pubkey(self, party, b_int):
        a = _c(b_int, np.uint64); assert lib().ora_set_pubkey(self.h, party, _p(a)) == 0

    def set_crs(self, a_int):
        a = _c(a_int, np.uint64); assert lib().ora_set_crs(self.h, _p(a)) == 0

    def modswitch(self, lwe):
        lwe = _c(lwe, np.uint32)
        at = np.empty(self.lwe_len - 1, dtype=np.uint32)
        bt = C.c_uint32(0)
        lib().ora_modswitch(self.h, _p(lwe), _p(at), C.byref(bt))
        return at, bt.value

    def testvector(self, btilde):
        acc = np.empty((self.kacc + 1, self.p.N), dtype=np.uint64)
        lib().ora_testvector(self.h, int(btilde), _p(acc))
        return acc

    def blindrotate(self, atilde, acc):
        at = _c(atilde, np.uint32)
        acc = np.array(acc, dtype=np.uint64, order="C", copy=True)
        lib().ora_blindrotate(self.h, _p(at), _p(acc))
        return acc

    def keyswitch(self, acc):
        acc = _c(acc, np.uint64)
        out = np.empty(self.lwe_len, dtype=np.uint32)
        lib().ora_keyswitch(self.h, _p(acc), _p(out))
        return out

    def bootstrap(self, lwe):
        out = np.array(lwe, dtype=np.uint32, order="C", copy=True)
        lib().ora_bootstrap(self.h, _p(out))
        return out

    def gate(self, op, x, y):
        x, y = _c(x, np.uint32), _c(y, np.uint32)
        out = np.empty_like(x)
        lib().ora_gate(self.h, op, _p(x), _p(y), _p(out))
        return out

    def gate_batch(self, op, x, y, threads=1):
        x, y = _c(x, np.uint32), _c(y, np.uint32)
        out = np.empty_like(x)
        lib().ora_gate_batch(self.h, op, _p(x), _p(y), _p(out), x.shape[0], threads)
        return out

    def kms_phase1(self, party, atilde_party):
        at = _c(atilde_party, np.uint32)
        rows = 1 if party == 0 else self.p.l_lev
        out = np.empty((rows, 2, self.p.N // 2), dtype=np.complex128)
        r = lib().ora_kms_phase1(self.h, party, _p(at), _p(out))
        assert r == rows
        return out

    def kms_phase2(self, levkeys, acc):
        lev = [np.ascontiguousarray(x, dtype=np.complex128) for x in levkeys]
        arr = (C.c_void_p * len(lev))(*[x.ctypes.data for x in lev])
        acc = np.array(acc, dtype=np.uint64, order="C", copy=True)
        lib().ora_kms_phase2(self.h, arr, _p(acc))
        return acc
