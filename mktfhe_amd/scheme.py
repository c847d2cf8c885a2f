"""Host-side mirror of the reference's operator surface for the gate-bootstrapping path.

Reference (Julia, /root/reference/src): exports at MKTFHE.jl:21-35 --
    setup, party_keygen, CRS, lwe_encrypt, lwe_ith_encrypt, lwe_decrypt      tfhe/scheme.jl
    bootstrapping!                                                            tfhe/bootstrapping.jl:4
    NAND, AND, OR, XOR, XNOR, NOR, NOT!                                       tfhe/gate.jl
plus the internal blindrotate! / keyswitch! named by the north star.  Same names and argument
meaning (`!` dropped: in-place functions carry a trailing underscore); every ciphertext argument
is a BATCH: a numpy uint32 array (..., k*n+1) in host memory or a torch CUDA int32/uint32 tensor
(device memory, zero copy).  All compute goes through the C ABI (include/mktfhe.h); this module
holds no arithmetic of its own and raises if the HIP library or a gfx950 GPU is missing.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MktError, check
from .params import CGGI, LMSS, CCS, KMS, KMS_BLOCK, Params

MEM_DEVICE, MEM_HOST = 0, 1
FMT_INT_COEFF, FMT_F64_FFT = 0, 1
ARITH_F64REF, ARITH_EXACT = 0, 1


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _arg(x, dtype, writable=False, scheme=None):
    """-> (pointer, mem kind, keepalive).  GPU tensors must live on the scheme's device; unless the caller pinned a stream
    with Scheme.set_stream, the engine enqueues on torch's CURRENT stream of that device, so its kernels are ordered
    with the producer and the consumer of the tensor like any torch op."""
    if _is_torch(x):
        if not x.is_cuda:
            raise ValueError("torch tensors must live on the GPU; pass numpy arrays for host memory")
        if scheme is not None:
            scheme._follow_torch(x)
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        if x.element_size() != np.dtype(dtype).itemsize:
            raise ValueError(f"tensor element size {x.element_size()} does not match {np.dtype(dtype)}")
        return C.c_void_p(x.data_ptr()), MEM_DEVICE, x
    a = np.ascontiguousarray(x, dtype=dtype)
    if writable and a is not x:
        raise ValueError("output / in-place argument must be a contiguous numpy array of dtype %s" % np.dtype(dtype))
    return _np_ptr(a), MEM_HOST, a


# ------------------------------------------------------------------------------------------------
# client side: CRS, party_keygen / setup keys, encrypt, decrypt  (CPU, exact integer arithmetic)
# ------------------------------------------------------------------------------------------------
def _seed_arg(deterministic_seed):
    """-> (pointer or None, keepalive).  None (the default everywhere) = the library draws a fresh 256-bit seed from the
    OS for this call, like the reference's per-call ChaCha20 entropy (sampler.jl:1-34).  An int (tests / benchmarks
    ONLY: the result is reproducible, hence public) is expanded by mkt_client_test_seed; 32 bytes are used as they are."""
    if deterministic_seed is None:
        return None, None
    buf = (C.c_uint8 * 32)()
    if isinstance(deterministic_seed, (bytes, bytearray)):
        if len(deterministic_seed) != 32:
            raise ValueError("a seed is 32 bytes")
        buf[:] = deterministic_seed
    else:
        check(_lib.lib().mkt_client_test_seed(int(deterministic_seed) & (2**64 - 1), buf))
    return C.cast(buf, C.c_void_p), buf


def CRS(params: Params, deterministic_seed=None):
    """scheme.jl:409-410 CRS(params): l_uni uniform ring polynomials -> (l_uni, N) ring words.
    Fresh OS randomness unless `deterministic_seed` (tests / benchmarks only) pins it."""
    out = np.empty((params.l_uni, params.N), dtype=params.ring_dtype)
    sp, _keep = _seed_arg(deterministic_seed)
    check(_lib.lib().mkt_client_crs(C.byref(params.c()), sp, _np_ptr(out)))
    return out


class PartyKeys:
    """One party's secret and evaluation keys (party_keygen, scheme.jl:227,:273,:324; setup for the
    single-key schemes, scheme.jl:151,:190).  Evaluation keys are in integer (coefficient) form."""

    def __init__(self, params: Params, party=0, crs=None, secrets_only=False, deterministic_seed=None):
        """Keys are drawn from fresh OS randomness; `deterministic_seed` (an int, or 32 bytes) pins the streams for
        tests and benchmarks ONLY -- such keys are reproducible by anyone.
        secrets_only: leave out the two large keys (bootstrapping key, key-switching key); they are then generated
        on the GPU by Scheme.keygen_device from the same streams (identical words).  That hands this party's secrets
        to that GPU: a party-local step (own machine), not something an evaluator does for every party."""
        self.params, self.party, self.secrets_only = params, party, secrets_only
        h = C.c_void_p()
        self._crs = np.ascontiguousarray(crs, dtype=params.ring_dtype) if crs is not None else None
        crs_p = _np_ptr(self._crs) if crs is not None else None
        fn = _lib.lib().mkt_client_party_secrets if secrets_only else _lib.lib().mkt_client_party_keygen
        sp, _keep = _seed_arg(deterministic_seed)
        check(fn(C.byref(params.c()), sp, party, crs_p, params.alpha, params.beta, C.byref(h)))
        self.h = h

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            try:
                _lib.lib().mkt_client_party_destroy(self.h)
            except Exception:       # interpreter shutdown: the library may already be gone
                pass
            self.h = None

    def _view(self, addr, nbytes, dtype):
        """zero-copy numpy view of key material owned by the C object; the view keeps this PartyKeys (and with it the
        memory) alive"""
        raw = (C.c_uint8 * nbytes).from_address(addr)
        raw._owner = self
        return np.frombuffer(raw, dtype=dtype)

    def _buf(self, fn, dtype):
        n = C.c_size_t(0)
        p = fn(self.h, C.byref(n))
        if not n.value:
            return None
        return self._view(p, n.value, dtype)

    @property
    def lwekey(self):
        return self._view(_lib.lib().mkt_client_lwekey(self.h), 4 * self.params.n, np.uint32)

    def ringkey(self, idx=0):
        """ring secret polynomial idx as int8 [N] (SK schemes: idx < k; CCS: 0; KMS: 0 = gsw key, 1 = uni key)"""
        n = C.c_size_t(0)
        p = _lib.lib().mkt_client_ringkey(self.h, idx, C.byref(n))
        return self._view(p, n.value, np.int8) if n.value else None

    @property
    def brk(self):
        return self._buf(_lib.lib().mkt_client_brk, self.params.ring_dtype)

    @property
    def ksk(self):
        return self._buf(_lib.lib().mkt_client_ksk, np.uint32)

    @property
    def rlk_d(self):
        return self._buf(_lib.lib().mkt_client_rlk_d, self.params.ring_dtype)

    @property
    def rlk_f(self):
        return self._buf(_lib.lib().mkt_client_rlk_f, self.params.ring_dtype)

    @property
    def pubkey(self):
        return self._buf(_lib.lib().mkt_client_pubkey, self.params.ring_dtype)


def party_keygen(a, params: Params, party=0, secrets_only=False, deterministic_seed=None):
    """scheme.jl:227/:273/:324 party_keygen(a, params) -> PartyKeys (lwekey + bootstrapping key); fresh randomness
    per call unless `deterministic_seed` (tests / benchmarks only) is given"""
    return PartyKeys(params, party=party, crs=a, secrets_only=secrets_only, deterministic_seed=deterministic_seed)


def lwe_encrypt(m, key: PartyKeys, params: Params, deterministic_seed=None):
    """scheme.jl:352-368 lwe_encrypt(m, key, params) (single-key schemes); fresh mask and noise per call"""
    return lwe_ith_encrypt(m, 0, key, params, deterministic_seed)


def lwe_ith_encrypt(m, i, key: PartyKeys, params: Params, deterministic_seed=None):
    """scheme.jl:370-386 lwe_ith_encrypt(m, i, key, params); i is the 0-based party index.  Mask and noise come from
    fresh OS randomness on every call; `deterministic_seed` (tests / benchmarks only) pins them"""
    out = np.empty(params.lwe_len, dtype=np.uint32)
    sp, _keep = _seed_arg(deterministic_seed)
    check(_lib.lib().mkt_client_lwe_encrypt(C.byref(params.c()), key.h, i, int(bool(m)), params.alpha, sp, _np_ptr(out)))
    return out


def lwe_decrypt(ctxt, keys, params: Params):
    """scheme.jl:388-407 lwe_decrypt(lwe, key(s)[, params]) -> bool (or array of bool for a batch)"""
    keys = [keys] if isinstance(keys, PartyKeys) else list(keys)
    arr = (C.c_void_p * len(keys))(*[k.h for k in keys])
    c = np.ascontiguousarray(ctxt, dtype=np.uint32)
    flat = c.reshape(-1, params.lwe_len)
    res = np.empty(flat.shape[0], dtype=bool)
    for j in range(flat.shape[0]):
        res[j] = bool(check(_lib.lib().mkt_client_lwe_decrypt(C.byref(params.c()), arr, len(keys), _np_ptr(flat[j]))))
    return res.reshape(c.shape[:-1]) if c.ndim > 1 else bool(res[0])


# ------------------------------------------------------------------------------------------------
# evaluator: the scheme object lives on one MI355X
# ------------------------------------------------------------------------------------------------
class Scheme:
    """The reference's CGGI / LMSS / CCS / KMS / KMS_block scheme object (scheme.jl:107-116, :168-179,
    :209-219, :256-265, :301-312) as a per-device engine context: twiddle tables (fft.jl:18-45),
    monomial table (scheme.jl:121-146) and the pre-transformed evaluation keys, all resident in HBM."""

    def __init__(self, params: Params, device=0, arith=ARITH_F64REF):
        self.params = params
        self.device = device
        h = C.c_void_p()
        check(_lib.lib().mkt_ctx_create(C.byref(params.c()), arith, device, C.byref(h)))
        self.h = h
        self.arith = arith
        self._user_stream = False     # True once set_stream pinned a stream; else torch's current stream is followed

    def _follow_torch(self, t):
        if t.device.index != self.device:
            raise ValueError(f"tensor lives on cuda:{t.device.index}, this scheme on device {self.device}")
        if not self._user_stream:
            import torch
            self._ck(_lib.lib().mkt_set_stream(self.h, C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)))

    def fork(self):
        """a second Scheme over the SAME resident keys (mkt_ctx_fork: no copy) with its own stream and workspace -- one
        per concurrent caller, as concurrent bootstrapping! calls share one read-only scheme object in the reference.
        From then on the keys are immutable on every sharer."""
        f = object.__new__(Scheme)
        f.params, f.device, f.arith, f._user_stream = self.params, self.device, self.arith, False
        h = C.c_void_p()
        self._ck(_lib.lib().mkt_ctx_fork(self.h, C.byref(h)))
        f.h = h
        return f

    def close(self):
        if getattr(self, "h", None):
            if getattr(self, "_owned", True):        # a borrowed view (MultiScheme.shard) never destroys the context
                try:
                    _lib.lib().mkt_ctx_destroy(self.h)
                except Exception:                    # interpreter shutdown: the module globals may already be gone
                    pass
            self.h = None

    __del__ = close

    def _ck(self, code):
        return check(code, self.h)

    # -- keys
    def load_party(self, party, keys: PartyKeys = None, *, brk=None, ksk=None, rlk_d=None, rlk_f=None, pubkey=None, fmt=FMT_INT_COEFF):
        L, p = _lib.lib(), self.params
        if keys is not None:
            brk, ksk, rlk_d, rlk_f, pubkey = keys.brk, keys.ksk, keys.rlk_d, keys.rlk_f, keys.pubkey
        kd = np.complex128 if fmt == FMT_F64_FFT else p.ring_dtype
        if brk is not None:
            self._ck(L.mkt_load_brk(self.h, party, _np_ptr(np.ascontiguousarray(brk, dtype=kd)), fmt))
        if ksk is not None:
            self._ck(L.mkt_load_ksk(self.h, party, _np_ptr(np.ascontiguousarray(ksk, dtype=np.uint32))))
        if rlk_d is not None:
            self._ck(L.mkt_load_rlk(self.h, party, _np_ptr(np.ascontiguousarray(rlk_d, dtype=kd)),
                                    _np_ptr(np.ascontiguousarray(rlk_f, dtype=kd)), fmt))
        if pubkey is not None:
            self._ck(L.mkt_load_pubkey(self.h, party, _np_ptr(np.ascontiguousarray(pubkey, dtype=kd)), fmt))

    def keygen_device(self, party, keys: PartyKeys, export=False):
        """keygen.jl:13-23 etc. on the GPU: bootstrapping + key-switching key of `party` from its secrets (the small
        keys -- public key, relinearisation key -- are uploaded from `keys`).  This hands the party's SECRETS to this
        context's GPU: it is the party's own step.  export=True also returns (brk, ksk) in the integer layouts of
        load_party, i.e. what the party ships to the evaluator (keyblob.dump_arrays)."""
        L = _lib.lib()
        crs_p = _np_ptr(keys._crs) if (self.params.scheme == CCS and keys._crs is not None) else None
        out = None
        if export:
            p = self.params
            brk = np.empty(self.brk_words(), dtype=p.ring_dtype)
            ksk = np.empty(self.get_ksk_shape(), dtype=np.uint32)
            self._ck(L.mkt_keygen_device_export(self.h, party, keys.h, crs_p, _np_ptr(brk), _np_ptr(ksk)))
            out = (brk, ksk.ravel())
        else:
            self._ck(L.mkt_keygen_device(self.h, party, keys.h, crs_p))
        self.load_party(party, rlk_d=keys.rlk_d, rlk_f=keys.rlk_f, pubkey=keys.pubkey)
        return out

    def brk_words(self):
        """ring words of one party's bootstrapping key (include/mktfhe.h layouts)"""
        p = self.params
        if p.scheme == CCS:
            return p.n * 3 * p.l_uni * p.N
        kr = 1 if p.scheme in (KMS, KMS_BLOCK) else p.k
        return p.n * (kr + 1) * p.l_gsw * (kr + 1) * p.N

    def get_ksk_shape(self):
        p = self.params
        D = 1 << p.logD
        rows = (1 if p.multikey else p.k) * p.N * (D // 2 if p.scheme in (LMSS, KMS_BLOCK) else D - 1) * p.f
        return (rows, p.n + 1)

    def get_ksk(self, party):
        out = np.empty(self.get_ksk_shape(), dtype=np.uint32)
        self._ck(_lib.lib().mkt_get_ksk(self.h, party, _np_ptr(out)))
        return out

    def load_crs(self, a, fmt=FMT_INT_COEFF):
        kd = np.complex128 if fmt == FMT_F64_FFT else self.params.ring_dtype
        self._ck(_lib.lib().mkt_load_crs(self.h, _np_ptr(np.ascontiguousarray(a, dtype=kd)), fmt))

    def set_stream(self, stream_handle):
        """pin the HIP stream (hipStream_t handle) every later call is enqueued on; None = back to the default:
        follow torch's current stream for GPU tensors (the NULL stream for host arrays)"""
        self._user_stream = stream_handle is not None
        self._ck(_lib.lib().mkt_set_stream(self.h, C.c_void_p(stream_handle or 0)))

    def synchronize(self):
        self._ck(_lib.lib().mkt_synchronize(self.h))

    def get_stream(self):
        """the hipStream_t handle (int) this context enqueues on right now (a fork: its own non-blocking stream)"""
        st = C.c_void_p()
        self._ck(_lib.lib().mkt_get_stream(self.h, C.byref(st)))
        return st.value or 0

    def set_option(self, name, value):
        """kernel-selection switch (mkt_set_option): "rot_wide", "rot_blkg", "ccs_pipe", ... -- results never depend on them;
        the parity tests force every kernel variant through this"""
        self._ck(_lib.lib().mkt_set_option(self.h, name.encode(), int(value)))

    def get_metric(self, name):
        """diagnostics of the Float64-pipe EXACT implementation (mkt_get_metric): "fx_available", "fx_bound", "fx_kmax", "fx_last_resid" """
        v = C.c_double(0.0)
        self._ck(_lib.lib().mkt_get_metric(self.h, name.encode(), C.byref(v)))
        return v.value

    def last_kernel_name(self):
        """base name of the blind-rotation kernel the last batch call launched"""
        return (_lib.lib().mkt_last_kernel_name(self.h) or b"").decode()

    # -- tables (tests)
    def twiddles(self, which):
        out = np.empty(self.params.N // 2, dtype=np.complex128)
        self._ck(_lib.lib().mkt_get_twiddles(self.h, which, _np_ptr(out)))
        return out

    def monomial(self, e):
        out = np.empty(self.params.N // 2, dtype=np.complex128)
        self._ck(_lib.lib().mkt_get_monomial(self.h, e, _np_ptr(out)))
        return out

    # -- timing
    def enable_timing(self, on=True):
        self._ck(_lib.lib().mkt_enable_timing(self.h, int(on)))

    def kernel_ms(self, which):
        """(total ms, launches) of kernel class `which` since enable_timing: 0 whole call, 1 blind
        rotation, 2 key switch, 3 transform, 4 KMS phase 2"""
        ms = C.c_double(0)
        n = self._ck(_lib.lib().mkt_last_kernel_ms(self.h, which, C.byref(ms)))
        return ms.value, n

    # -- hot path
    def _batch(self, x):
        return int(np.prod(x.shape[:-1])) if len(x.shape) > 1 else 1

    def gate(self, op, x, y, out=None):
        px, mem, kx = _arg(x, np.uint32, scheme=self)
        py, mem2, ky = _arg(y, np.uint32, scheme=self)
        if mem != mem2:
            raise ValueError("x and y must both be host arrays or both be GPU tensors")
        if out is None:
            out = kx.new_empty(kx.shape) if mem == MEM_DEVICE else np.empty_like(kx)
        po, mem3, ko = _arg(out, np.uint32, writable=True, scheme=self)
        if mem3 != mem:
            raise ValueError("out must live where the inputs live")
        if tuple(kx.shape) != tuple(ky.shape) or kx.shape[-1] != self.params.lwe_len:
            raise ValueError("ciphertext shape mismatch")       # reference: @assert length checks
        self._ck(_lib.lib().mkt_gate_batch(self.h, op, px, py, po, self._batch(kx), mem))
        return ko

    def gate_ops(self, ops, x, y, out=None):
        """a different gate per ciphertext pair (mkt_gate_batch_ops; the reference's tests draw a random gate per step,
        test/KMS.jl:29-34): ops[j] in 0..5 (NAND..NOR), optionally | OP_NOT_X / OP_NOT_Y (that input negated first, NOT!);
        a uint8 array living where x and y live"""
        px, mem, kx = _arg(x, np.uint32, scheme=self)
        py, mem2, ky = _arg(y, np.uint32, scheme=self)
        po_, mem4, kops = _arg(ops, np.uint8, scheme=self)
        if not (mem == mem2 == mem4):
            raise ValueError("ops, x and y must all be host arrays or all be GPU tensors")
        if out is None:
            out = kx.new_empty(kx.shape) if mem == MEM_DEVICE else np.empty_like(kx)
        po, mem3, ko = _arg(out, np.uint32, writable=True, scheme=self)
        if mem3 != mem:
            raise ValueError("out must live where the inputs live")
        B = self._batch(kx)
        if tuple(kx.shape) != tuple(ky.shape) or kx.shape[-1] != self.params.lwe_len or int(np.prod(kops.shape)) != B:
            raise ValueError("ciphertext / ops shape mismatch")
        self._ck(_lib.lib().mkt_gate_batch_ops(self.h, po_, px, py, po, B, mem))
        return ko

    def gate_gather(self, ops, pool, ix, iy, out):
        """one circuit level (mkt_gate_batch_gather): gate j = ops[j](pool[ix[j]], pool[iy[j]]) -> out[j]; pool (rows, k*n+1),
        ix / iy uint32 (int32 tensors) row indices; out may be a later region of the pool"""
        pp, mem, kp = _arg(pool, np.uint32, scheme=self)
        pops, m1, kops = _arg(ops, np.uint8, scheme=self)
        pix, m2, kix = _arg(ix, np.uint32, scheme=self)
        piy, m3, kiy = _arg(iy, np.uint32, scheme=self)
        po, m4, ko = _arg(out, np.uint32, writable=True, scheme=self)
        if not (mem == m1 == m2 == m3 == m4):
            raise ValueError("all arguments must live in the same memory")
        B = int(np.prod(kops.shape))
        if kp.shape[-1] != self.params.lwe_len or int(np.prod(kix.shape)) != B or int(np.prod(kiy.shape)) != B or self._batch(ko) != B:
            raise ValueError("shape mismatch")
        self._ck(_lib.lib().mkt_gate_batch_gather(self.h, pops, pp, self._batch(kp), pix, piy, po, B, mem))
        return ko

    def mux(self, s, a, b, out=None):
        """MUX(s, a, b) = s ? a : b with two blind rotations and one key switch (mkt_mux_batch; the reference has no MUX gate)"""
        ps, mem, ks = _arg(s, np.uint32, scheme=self)
        pa, m1, ka = _arg(a, np.uint32, scheme=self)
        pb, m2, kb = _arg(b, np.uint32, scheme=self)
        if out is None:
            out = ks.new_empty(ks.shape) if mem == MEM_DEVICE else np.empty_like(ks)
        po, m3, ko = _arg(out, np.uint32, writable=True, scheme=self)
        if not (mem == m1 == m2 == m3) or not (tuple(ks.shape) == tuple(ka.shape) == tuple(kb.shape)) or ks.shape[-1] != self.params.lwe_len:
            raise ValueError("ciphertext shape / memory mismatch")
        self._ck(_lib.lib().mkt_mux_batch(self.h, ps, pa, pb, po, self._batch(ks), mem))
        return ko

    def mux_gather(self, pool, i_s, i_a, i_b, out, not_ab=None):
        """a circuit level of native MUX gates (mkt_mux_batch_gather): out[j] = MUX(pool[i_s[j]], a', b'), a' = pool[i_a[j]] or its
        negation if bit 0 of not_ab[j] is set (b' likewise, bit 1)"""
        pp, mem, kp = _arg(pool, np.uint32, scheme=self)
        ps, m1, ks = _arg(i_s, np.uint32, scheme=self)
        pa, m2, ka = _arg(i_a, np.uint32, scheme=self)
        pb, m3, kb = _arg(i_b, np.uint32, scheme=self)
        po, m4, ko = _arg(out, np.uint32, writable=True, scheme=self)
        pf, m5, kf = (None, mem, None) if not_ab is None else _arg(not_ab, np.uint8, scheme=self)
        if not (mem == m1 == m2 == m3 == m4 == m5):
            raise ValueError("all arguments must live in the same memory")
        B = int(np.prod(ks.shape))
        if kp.shape[-1] != self.params.lwe_len or int(np.prod(ka.shape)) != B or int(np.prod(kb.shape)) != B or self._batch(ko) != B:
            raise ValueError("shape mismatch")
        self._ck(_lib.lib().mkt_mux_batch_gather(self.h, pp, self._batch(kp), ps, pa, pb, pf, po, B, mem))
        return ko

    def bootstrapping_(self, ctxt):
        p, mem, k = _arg(ctxt, np.uint32, writable=True, scheme=self)
        if k.shape[-1] != self.params.lwe_len:
            raise ValueError("ciphertext shape mismatch")
        self._ck(_lib.lib().mkt_bootstrap_batch(self.h, p, self._batch(k), mem))
        return k

    def not_(self, ctxt):
        p, mem, k = _arg(ctxt, np.uint32, writable=True, scheme=self)
        self._ck(_lib.lib().mkt_not_batch(self.h, p, self._batch(k), mem))
        return k

    def modswitch(self, ctxt):
        c = np.ascontiguousarray(ctxt, dtype=np.uint32)
        B = self._batch(c)
        at = np.empty(c.shape[:-1] + (self.params.lwe_len - 1,), dtype=np.uint32)
        bt = np.empty(c.shape[:-1] if c.ndim > 1 else (1,), dtype=np.uint32)
        self._ck(_lib.lib().mkt_modswitch_batch(self.h, _np_ptr(c), _np_ptr(at), _np_ptr(bt), B, MEM_HOST))
        return at, bt

    def blindrotate_(self, atilde, acc):
        pa, mem, ka = _arg(atilde, np.uint32, scheme=self)
        pc, mem2, kc = _arg(acc, self.params.ring_dtype, writable=True, scheme=self)
        if mem != mem2:
            raise ValueError("atilde and acc must live in the same memory")
        self._ck(_lib.lib().mkt_blindrotate_batch(self.h, pa, pc, self._batch(ka), mem))
        return kc

    def keyswitch(self, acc):
        a = np.ascontiguousarray(acc, dtype=self.params.ring_dtype)
        B = int(np.prod(a.shape[:-2])) if a.ndim > 2 else 1
        out = np.empty(a.shape[:-2] + (self.params.lwe_len,), dtype=np.uint32)
        self._ck(_lib.lib().mkt_keyswitch_batch(self.h, _np_ptr(a), _np_ptr(out), B, MEM_HOST))
        return out

    def kms_phase1(self, atilde):
        a = np.ascontiguousarray(atilde, dtype=np.uint32)
        B = self._batch(a)
        p = self.params
        rtot = 1 + (p.k - 1) * p.l_lev
        if self.arith == ARITH_EXACT:    # split residue tables: [rows][polynomial b / a][low / high 32-bit half][N] residue pairs, Montgomery form
            out = np.empty(a.shape[:-1] + (rtot, 2, 2, p.N), dtype=np.uint64)
        else:
            out = np.empty(a.shape[:-1] + (rtot, 2, p.N // 2), dtype=np.complex128)
        self._ck(_lib.lib().mkt_kms_phase1_batch(self.h, _np_ptr(a), _np_ptr(out), B, MEM_HOST))
        return out

    def transform_fwd(self, p, out=None):
        pp, mem, kp = _arg(p, self.params.ring_dtype, scheme=self)
        if out is None:
            if mem == MEM_DEVICE:
                import torch
                out = torch.empty(tuple(kp.shape[:-1]) + (self.params.N // 2,), dtype=torch.complex128, device=kp.device)
            else:
                out = np.empty(kp.shape[:-1] + (self.params.N // 2,), dtype=np.complex128)
        po, _, ko = _arg(out, np.complex128, writable=True, scheme=self)
        self._ck(_lib.lib().mkt_transform_fwd_batch(self.h, pp, po, self._batch(kp), mem))
        return ko

    def transform_inv(self, t, out=None):
        pt, mem, kt = _arg(t, np.complex128, scheme=self)
        if out is None:
            if mem == MEM_DEVICE:
                import torch
                tdt = torch.int64 if self.params.W == 64 else torch.int32
                out = torch.empty(tuple(kt.shape[:-1]) + (self.params.N,), dtype=tdt, device=kt.device)
            else:
                out = np.empty(kt.shape[:-1] + (self.params.N,), dtype=self.params.ring_dtype)
        po, _, ko = _arg(out, self.params.ring_dtype, writable=True, scheme=self)
        self._ck(_lib.lib().mkt_transform_inv_batch(self.h, pt, po, self._batch(kt), mem))
        return ko

    def exact_polymul(self, a, b):
        """MKT_ARITH_EXACT contexts: a (*) b mod (X^N + 1, 2^W), exactly, for digit polynomials a (signed, N * max|a| < 2^28) and ring
        polynomials b; host arrays (..., N)"""
        aa = np.ascontiguousarray(a, dtype=self.params.ring_dtype)
        bb = np.ascontiguousarray(b, dtype=self.params.ring_dtype)
        out = np.empty_like(aa)
        self._ck(_lib.lib().mkt_exact_polymul_batch(self.h, _np_ptr(aa), _np_ptr(bb), _np_ptr(out), self._batch(aa), MEM_HOST))
        return out

    def decompose(self, p, l, logB):
        a = np.ascontiguousarray(p, dtype=self.params.ring_dtype)
        B = self._batch(a)
        out = np.empty(a.shape[:-1] + (l, self.params.N), dtype=self.params.ring_dtype)
        self._ck(_lib.lib().mkt_decompose_batch(self.h, _np_ptr(a), _np_ptr(out), l, logB, B, MEM_HOST))
        return out


OP_NOT_X, OP_NOT_Y = 8, 16      # mktfhe.h MKT_OP_NOT_X / _Y: per-gate code bits of gate_ops / gate_gather


class MultiScheme:
    """ONE scheme object over several MI355X (mkt_multi_*): the reference's caller is one process whose threads
    share one read-only scheme (README.md:38-44, bootstrapping.jl:38-45); here the batch is cut into contiguous shards, one per
    entry of `devices` (a device named twice = two logical shards over that device's one key set), keys uploaded once and
    replicated device-to-device, every shard writing its slice of the caller's one output array.  No collective.  Same batch
    methods as Scheme; arrays are host numpy arrays or GPU tensors on ANY of the devices."""

    def __init__(self, params: Params, devices, arith=ARITH_F64REF, private_keys=False, stage_always=False, no_peer=False):
        """private_keys: shards that share a device each get their own replicated key copy (MKT_MULTI_PRIVATE_KEYS: the
        device-to-device replication path, testable on one GPU) instead of sharing that device's one key set"""
        self.params, self.devices, self.arith = params, list(devices), arith
        h = C.c_void_p()
        arr = (C.c_int * len(self.devices))(*self.devices)
        code = _lib.lib().mkt_multi_create(C.byref(params.c()), arith, arr, len(self.devices), (1 if private_keys else 0) | (2 if stage_always else 0) | (4 if no_peer else 0), C.byref(h))
        if code < 0:
            raise MktError(code, (_lib.lib().mkt_multi_last_error(None) or b"").decode())
        self.h = h
        self._sealed = False

    def _ck(self, code):
        if code < 0:
            raise MktError(code, (_lib.lib().mkt_multi_last_error(self.h) or b"").decode())
        return code

    def close(self):
        if getattr(self, "h", None):
            try:
                _lib.lib().mkt_multi_destroy(self.h)
            except Exception:                        # interpreter shutdown
                pass
            self.h = None

    __del__ = close

    @property
    def nshards(self):
        return len(self.devices)

    def shard_range(self, B, shard):
        lo, hi = C.c_size_t(), C.c_size_t()
        self._ck(_lib.lib().mkt_multi_shard_range(self.h, B, shard, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def shard(self, i):
        """borrowed Scheme view of shard i's context (timing, kernel names); valid while this object lives"""
        s = object.__new__(Scheme)
        s.params, s.device, s.arith, s._user_stream = self.params, self.devices[i], self.arith, True
        s.h = C.c_void_p(_lib.lib().mkt_multi_ctx(self.h, i))
        s._owned = False
        return s

    # -- keys: once, on the first device; replicate() copies them to the others
    def load_party(self, party, keys: "PartyKeys" = None, *, brk=None, ksk=None, rlk_d=None, rlk_f=None, pubkey=None, fmt=FMT_INT_COEFF):
        L, p = _lib.lib(), self.params
        if keys is not None:
            brk, ksk, rlk_d, rlk_f, pubkey = keys.brk, keys.ksk, keys.rlk_d, keys.rlk_f, keys.pubkey
        kd = np.complex128 if fmt == FMT_F64_FFT else p.ring_dtype
        if brk is not None:
            self._ck(L.mkt_multi_load_brk(self.h, party, _np_ptr(np.ascontiguousarray(brk, dtype=kd)), fmt))
        if ksk is not None:
            self._ck(L.mkt_multi_load_ksk(self.h, party, _np_ptr(np.ascontiguousarray(ksk, dtype=np.uint32))))
        if rlk_d is not None:
            self._ck(L.mkt_multi_load_rlk(self.h, party, _np_ptr(np.ascontiguousarray(rlk_d, dtype=kd)),
                                          _np_ptr(np.ascontiguousarray(rlk_f, dtype=kd)), fmt))
        if pubkey is not None:
            self._ck(L.mkt_multi_load_pubkey(self.h, party, _np_ptr(np.ascontiguousarray(pubkey, dtype=kd)), fmt))

    def keygen_device(self, party, keys: "PartyKeys"):
        crs_p = _np_ptr(keys._crs) if (self.params.scheme == CCS and keys._crs is not None) else None
        self._ck(_lib.lib().mkt_multi_keygen_device(self.h, party, keys.h, crs_p))
        self.load_party(party, rlk_d=keys.rlk_d, rlk_f=keys.rlk_f, pubkey=keys.pubkey)

    def load_crs(self, a, fmt=FMT_INT_COEFF):
        kd = np.complex128 if fmt == FMT_F64_FFT else self.params.ring_dtype
        self._ck(_lib.lib().mkt_multi_load_crs(self.h, _np_ptr(np.ascontiguousarray(a, dtype=kd)), fmt))

    def replicate(self):
        self._ck(_lib.lib().mkt_multi_replicate(self.h))
        self._sealed = True

    def set_option(self, name, value):
        self._ck(_lib.lib().mkt_multi_set_option(self.h, name.encode(), int(value)))

    # -- hot path (no stream following: calls are synchronous, inputs are settled by the library)
    @staticmethod
    def _batch(x):
        return int(np.prod(x.shape[:-1])) if len(x.shape) > 1 else 1

    def gate(self, op, x, y, out=None):
        px, mem, kx = _arg(x, np.uint32)
        py, mem2, ky = _arg(y, np.uint32)
        if mem != mem2:
            raise ValueError("x and y must both be host arrays or both be GPU tensors")
        if out is None:
            out = kx.new_empty(kx.shape) if mem == MEM_DEVICE else np.empty_like(kx)
        po, mem3, ko = _arg(out, np.uint32, writable=True)
        if mem3 != mem or tuple(kx.shape) != tuple(ky.shape) or kx.shape[-1] != self.params.lwe_len:
            raise ValueError("ciphertext shape / memory mismatch")
        self._ck(_lib.lib().mkt_multi_gate_batch(self.h, op, px, py, po, self._batch(kx), mem))
        return ko

    def gate_ops(self, ops, x, y, out=None):
        px, mem, kx = _arg(x, np.uint32)
        py, mem2, ky = _arg(y, np.uint32)
        pops, mem4, kops = _arg(ops, np.uint8)
        if out is None:
            out = kx.new_empty(kx.shape) if mem == MEM_DEVICE else np.empty_like(kx)
        po, mem3, ko = _arg(out, np.uint32, writable=True)
        if not (mem == mem2 == mem3 == mem4) or tuple(kx.shape) != tuple(ky.shape) or kx.shape[-1] != self.params.lwe_len:
            raise ValueError("ciphertext shape / memory mismatch")
        self._ck(_lib.lib().mkt_multi_gate_batch_ops(self.h, pops, px, py, po, self._batch(kx), mem))
        return ko

    def mux(self, s, a, b, out=None):
        ps, mem, ks = _arg(s, np.uint32)
        pa, m1, ka = _arg(a, np.uint32)
        pb, m2, kb = _arg(b, np.uint32)
        if out is None:
            out = ks.new_empty(ks.shape) if mem == MEM_DEVICE else np.empty_like(ks)
        po, m3, ko = _arg(out, np.uint32, writable=True)
        if not (mem == m1 == m2 == m3) or not (tuple(ks.shape) == tuple(ka.shape) == tuple(kb.shape)) or ks.shape[-1] != self.params.lwe_len:
            raise ValueError("ciphertext shape / memory mismatch")
        self._ck(_lib.lib().mkt_multi_mux_batch(self.h, ps, pa, pb, po, self._batch(ks), mem))
        return ko

    def bootstrapping_(self, ctxt):
        p, mem, k = _arg(ctxt, np.uint32, writable=True)
        self._ck(_lib.lib().mkt_multi_bootstrap_batch(self.h, p, self._batch(k), mem))
        return k

    def not_(self, ctxt):
        p, mem, k = _arg(ctxt, np.uint32, writable=True)
        self._ck(_lib.lib().mkt_multi_not_batch(self.h, p, self._batch(k), mem))
        return k

    def blindrotate_(self, atilde, acc):
        pa, mem, ka = _arg(atilde, np.uint32)
        pc, mem2, kc = _arg(acc, self.params.ring_dtype, writable=True)
        if mem != mem2:
            raise ValueError("atilde and acc must live in the same memory")
        self._ck(_lib.lib().mkt_multi_blindrotate_batch(self.h, pa, pc, self._batch(ka), mem))
        return kc

    def keyswitch(self, acc):
        a = np.ascontiguousarray(acc, dtype=self.params.ring_dtype)
        B = int(np.prod(a.shape[:-2])) if a.ndim > 2 else 1
        out = np.empty(a.shape[:-2] + (self.params.lwe_len,), dtype=np.uint32)
        self._ck(_lib.lib().mkt_multi_keyswitch_batch(self.h, _np_ptr(a), _np_ptr(out), B, MEM_HOST))
        return out


def setup_multi(params: Params, devices, keys=None, a=None, arith=ARITH_F64REF, private_keys=False, stage_always=False, no_peer=False):
    """setup (scheme.jl:151 / :190 / :244 / :292 / :343) for a MultiScheme: evaluation keys uploaded (or, for keys made with
    secrets_only=True, generated) once on devices[0], pre-transformed there and replicated to the other devices"""
    sch = MultiScheme(params, devices, arith=arith, private_keys=private_keys, stage_always=stage_always, no_peer=no_peer)
    if params.multikey:
        sch.load_crs(a)
        klist = list(keys)
    else:
        klist = [keys if isinstance(keys, PartyKeys) else keys[0]]
    for i, kk in enumerate(klist):
        (sch.keygen_device if kk.secrets_only else sch.load_party)(i, kk)
    sch.replicate()
    return sch


def setup(params: Params, keys=None, a=None, device=0, deterministic_seed=None, arith=ARITH_F64REF):
    """scheme.jl:151 / :190 setup(params) -> (keys, scheme) for the single-key schemes, and
    scheme.jl:244 / :292 / :343 setup(a, btk, params) -> scheme for the multi-key ones
    (keys = list of PartyKeys, a = CRS).  The evaluation keys are uploaded and pre-transformed
    on `device`."""
    def install(sch, i, kk):      # keys made with secrets_only=True get their large keys generated on the device
        (sch.keygen_device if kk.secrets_only else sch.load_party)(i, kk)
    if not params.multikey:
        ks = keys if keys is not None else PartyKeys(params, party=0, deterministic_seed=deterministic_seed)
        sch = Scheme(params, device=device, arith=arith)
        install(sch, 0, ks)
        return ks, sch
    sch = Scheme(params, device=device, arith=arith)
    sch.load_crs(a)
    for i, kk in enumerate(keys):
        install(sch, i, kk)
    return sch


def bootstrapping_(ctxt, scheme: Scheme):
    """bootstrapping.jl:4 bootstrapping!(ctxt, scheme): in place on the batch"""
    return scheme.bootstrapping_(ctxt)


def blindrotate_(atilde, acc, scheme: Scheme):
    """bootstrapping.jl:32/:114/:234/:369 blindrotate!(atilde, acc, scheme)"""
    return scheme.blindrotate_(atilde, acc)


def keyswitch(acc, scheme: Scheme):
    """bootstrapping.jl:81/:170/:333/:564/:664 keyswitch!(res, acc, scheme) -> res"""
    return scheme.keyswitch(acc)


def NAND(c1, c2, scheme: Scheme, out=None):
    """gate.jl:1-8"""
    return scheme.gate(0, c1, c2, out)


def AND(c1, c2, scheme: Scheme, out=None):
    """gate.jl:10-17"""
    return scheme.gate(1, c1, c2, out)


def OR(c1, c2, scheme: Scheme, out=None):
    """gate.jl:19-26"""
    return scheme.gate(2, c1, c2, out)


def XOR(c1, c2, scheme: Scheme, out=None):
    """gate.jl:28-35"""
    return scheme.gate(3, c1, c2, out)


def XNOR(c1, c2, scheme: Scheme, out=None):
    """gate.jl:37-44"""
    return scheme.gate(4, c1, c2, out)


def MUX(s, c1, c2, scheme: Scheme, out=None):
    """s ? c1 : c2 -- not a reference operator (gate.jl has none; the north star names it): two blind rotations and one key
    switch, blindrotate!(AND-linear(s, c1)) + blindrotate!(AND-linear(NOT s, c2)) + 1/8, then keyswitch! (mkt_mux_batch)"""
    return scheme.mux(s, c1, c2, out)


def MUX_composite(s, c1, c2, scheme: Scheme):
    """the same function as the composite OR(AND(s, c1), AND(NOT s, c2)) of the reference's gates (three bootstraps), the two
    ANDs evaluated as one batch"""
    ns = s.clone() if hasattr(s, "clone") else np.array(s, copy=True)
    NOT_(ns, scheme)
    if hasattr(s, "clone"):
        import torch
        both = scheme.gate(1, torch.cat([s, ns]), torch.cat([c1, c2]))
    else:
        both = scheme.gate(1, np.concatenate([np.atleast_2d(s), np.atleast_2d(ns)]), np.concatenate([np.atleast_2d(c1), np.atleast_2d(c2)]))
    h = both.shape[0] // 2
    return scheme.gate(2, both[:h], both[h:])      # 1 = AND, 2 = OR (params.py)


def NOR(c1, c2, scheme: Scheme, out=None):
    """gate.jl:46-53"""
    return scheme.gate(5, c1, c2, out)


def NOT_(ctxt, scheme: Scheme):
    """gate.jl:55-58 NOT!(ctxt): negation, no bootstrap"""
    return scheme.not_(ctxt)
