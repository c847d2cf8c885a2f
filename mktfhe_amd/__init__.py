"""mktfhe_amd -- MI355X-native batched multi-key TFHE gate-bootstrapping engine.

Keeps the operator surface of the Julia reference SNUCP/MKTFHE for the gate-bootstrapping hot
path (bootstrapping!, blindrotate!, keyswitch!, NAND/AND/OR/XOR/XNOR/NOR/NOT!) behind a C ABI
(include/mktfhe.h) implemented with hand-written HIP kernels for gfx950.  No CPU fallback.
"""
from .params import *  # noqa: F401,F403
from .params import Params  # noqa: F401
from .scheme import (  # noqa: F401
    CRS, PartyKeys, Scheme, MultiScheme, party_keygen, setup, setup_multi, OP_NOT_X, OP_NOT_Y, lwe_encrypt, lwe_ith_encrypt, lwe_decrypt,
    bootstrapping_, blindrotate_, keyswitch, NAND, AND, OR, XOR, XNOR, NOR, NOT_, MUX, MUX_composite,
    MEM_DEVICE, MEM_HOST, FMT_INT_COEFF, FMT_F64_FFT, ARITH_F64REF, ARITH_EXACT,
)
from ._lib import MktError, LIB_PATH, build_id  # noqa: F401
from . import keyblob  # noqa: F401,E402
from . import circuit  # noqa: F401,E402
