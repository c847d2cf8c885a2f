"""ctypes loader of the engine's C-ABI library (include/mktfhe.h).

The library is built in-tree (mktfhe_amd/lib/libmktfhe_hip.so) by `make -C mktfhe_amd/csrc`
(see __graft_entry__.build).  There is no Python or CPU fallback for any compute entry point:
a missing library or a missing gfx950 device raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MKT_LIB_PATH: an alternative build of the same library (development A/B runs: tools/variant.sh, tools/sweep.sh) -- never a fallback
LIB_PATH = os.environ.get("MKT_LIB_PATH") or os.path.join(_HERE, "lib", "libmktfhe_hip.so")


class MktParams(C.Structure):
    """mkt_params (include/mktfhe.h) -- scheme.jl:6-101 flattened."""
    _fields_ = [(n, C.c_int32) for n in (
        "scheme", "n", "N", "k", "W", "l_gsw", "logB_gsw", "l_lev", "logB_lev",
        "l_uni", "logB_uni", "f", "logD", "blk_len", "blk_d")]


class MktError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mktfhe error {code}: {msg}")
        self.code = code


# every symbol include/mktfhe.h declares: name -> (restype, argtypes)
_vp, _i, _sz, _u64, _dbl = C.c_void_p, C.c_int, C.c_size_t, C.c_uint64, C.c_double
_pp = C.POINTER(MktParams)
SYMBOLS = {
    "mkt_abi_version": (_i, []),
    "mkt_build_id": (C.c_char_p, []),
    "mkt_ctx_create": (_i, [_pp, _i, _i, C.POINTER(_vp)]),
    "mkt_ctx_destroy": (_i, [_vp]),
    "mkt_ctx_fork": (_i, [_vp, C.POINTER(_vp)]),
    "mkt_last_error": (C.c_char_p, [_vp]),
    "mkt_set_stream": (_i, [_vp, _vp]),
    "mkt_get_stream": (_i, [_vp, C.POINTER(_vp)]),
    "mkt_synchronize": (_i, [_vp]),
    "mkt_set_option": (_i, [_vp, C.c_char_p, _i]),
    "mkt_last_kernel_name": (C.c_char_p, [_vp]),
    "mkt_get_metric": (_i, [_vp, C.c_char_p, C.POINTER(C.c_double)]),
    "mkt_get_twiddles": (_i, [_vp, _i, _vp]),
    "mkt_set_twiddles": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "mkt_make_twiddles": (_i, [_i, _i, _vp]),
    "mkt_load_brk": (_i, [_vp, _i, _vp, _i]),
    "mkt_load_ksk": (_i, [_vp, _i, _vp]),
    "mkt_load_rlk": (_i, [_vp, _i, _vp, _vp, _i]),
    "mkt_load_pubkey": (_i, [_vp, _i, _vp, _i]),
    "mkt_load_crs": (_i, [_vp, _vp, _i]),
    "mkt_keygen_device": (_i, [_vp, _i, _vp, _vp]),
    "mkt_keygen_device_export": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "mkt_get_ksk": (_i, [_vp, _i, _vp]),
    "mkt_gate_batch": (_i, [_vp, _i, _vp, _vp, _vp, _sz, _i]),
    "mkt_gate_batch_ops": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i]),
    "mkt_gate_batch_gather": (_i, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _sz, _i]),
    "mkt_mux_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i]),
    "mkt_mux_batch_gather": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _i]),
    "mkt_not_batch": (_i, [_vp, _vp, _sz, _i]),
    "mkt_bootstrap_batch": (_i, [_vp, _vp, _sz, _i]),
    "mkt_modswitch_batch": (_i, [_vp, _vp, _vp, _vp, _sz, _i]),
    "mkt_blindrotate_batch": (_i, [_vp, _vp, _vp, _sz, _i]),
    "mkt_keyswitch_batch": (_i, [_vp, _vp, _vp, _sz, _i]),
    "mkt_kms_phase1_batch": (_i, [_vp, _vp, _vp, _sz, _i]),
    "mkt_transform_fwd_batch": (_i, [_vp, _vp, _vp, _sz, _i]),
    "mkt_transform_inv_batch": (_i, [_vp, _vp, _vp, _sz, _i]),
    "mkt_decompose_batch": (_i, [_vp, _vp, _vp, _i, _i, _sz, _i]),
    "mkt_exact_polymul_batch": (_i, [_vp, _vp, _vp, _vp, _sz, _i]),
    "mkt_get_monomial": (_i, [_vp, _i, _vp]),
    "mkt_multi_create": (_i, [_pp, _i, C.POINTER(_i), _i, _i, C.POINTER(_vp)]),
    "mkt_multi_destroy": (_i, [_vp]),
    "mkt_multi_last_error": (C.c_char_p, [_vp]),
    "mkt_multi_nshards": (_i, [_vp]),
    "mkt_multi_device": (_i, [_vp, _i]),
    "mkt_multi_ctx": (_vp, [_vp, _i]),
    "mkt_multi_shard_range": (_i, [_vp, _sz, _i, C.POINTER(_sz), C.POINTER(_sz)]),
    "mkt_multi_load_brk": (_i, [_vp, _i, _vp, _i]),
    "mkt_multi_load_ksk": (_i, [_vp, _i, _vp]),
    "mkt_multi_load_rlk": (_i, [_vp, _i, _vp, _vp, _i]),
    "mkt_multi_load_pubkey": (_i, [_vp, _i, _vp, _i]),
    "mkt_multi_load_crs": (_i, [_vp, _vp, _i]),
    "mkt_multi_keygen_device": (_i, [_vp, _i, _vp, _vp]),
    "mkt_multi_replicate": (_i, [_vp]),
    "mkt_multi_set_option": (_i, [_vp, C.c_char_p, _i]),
    "mkt_multi_gate_batch": (_i, [_vp, _i, _vp, _vp, _vp, _sz, _i]),
    "mkt_multi_gate_batch_ops": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i]),
    "mkt_multi_mux_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i]),
    "mkt_multi_bootstrap_batch": (_i, [_vp, _vp, _sz, _i]),
    "mkt_multi_not_batch": (_i, [_vp, _vp, _sz, _i]),
    "mkt_multi_blindrotate_batch": (_i, [_vp, _vp, _vp, _sz, _i]),
    "mkt_multi_keyswitch_batch": (_i, [_vp, _vp, _vp, _sz, _i]),
    "mkt_enable_timing": (_i, [_vp, _i]),
    "mkt_last_kernel_ms": (_i, [_vp, _i, C.POINTER(_dbl)]),
    "mkt_client_random_seed": (_i, [_vp]),
    "mkt_client_test_seed": (_i, [_u64, _vp]),
    "mkt_client_crs": (_i, [_pp, _vp, _vp]),
    "mkt_client_party_keygen": (_i, [_pp, _vp, _i, _vp, _dbl, _dbl, C.POINTER(_vp)]),
    "mkt_client_party_secrets": (_i, [_pp, _vp, _i, _vp, _dbl, _dbl, C.POINTER(_vp)]),
    "mkt_client_party_destroy": (_i, [_vp]),
    "mkt_client_lwekey": (_vp, [_vp]),
    "mkt_client_ringkey": (_vp, [_vp, _i, C.POINTER(_sz)]),
    "mkt_client_brk": (_vp, [_vp, C.POINTER(_sz)]),
    "mkt_client_ksk": (_vp, [_vp, C.POINTER(_sz)]),
    "mkt_client_rlk_d": (_vp, [_vp, C.POINTER(_sz)]),
    "mkt_client_rlk_f": (_vp, [_vp, C.POINTER(_sz)]),
    "mkt_client_pubkey": (_vp, [_vp, C.POINTER(_sz)]),
    "mkt_client_lwe_encrypt": (_i, [_pp, _vp, _i, _i, _dbl, _vp, _vp]),
    "mkt_client_lwe_decrypt": (_i, [_pp, C.POINTER(_vp), _i, _vp]),
}

_lib = None


def lib():
    """Load the shared library; raises (never falls back) if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
                "or make -C mktfhe_amd/csrc). mktfhe_amd has no CPU fallback.")
        # When PyTorch is part of the process it brings its own copy of the HIP runtime (same SONAME).  Whichever copy is
        # mapped first serves both; mapped in the other order (this library, then torch) the runtime this library was
        # bound to sees no device.  So torch, if installed, goes first.
        # A torch install that cannot load (missing ROCm libraries: OSError / RuntimeError) must not take the host-only
        # client API (CRS, keygen, encrypt, decrypt) down with it.
        try:
            import torch  # noqa: F401
        except Exception:  # noqa: BLE001
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def build_id():
    """which source tree the loaded library was built from (csrc/Makefile BUILD_ID): what profiles are matched against"""
    return lib().mkt_build_id().decode()


def check(code, ctx=None):
    if code < 0:
        msg = lib().mkt_last_error(ctx)
        raise MktError(code, msg.decode() if msg else "")
    return code
