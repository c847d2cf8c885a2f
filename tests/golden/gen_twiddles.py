#!/usr/bin/env python3
"""Generate the twiddle-table fixture that pins the F64REF transform tables.

Follows the reference's FFTransformer constructor (src/ring/fft.jl:31-41) with mpmath
standing in for Julia's 256-bit BigFloat (MPFR):

    idx      = 0 : N/2-1
    Psi      = Complex{Float64}.(exp.((-im * big(pi) / halfN) .* idx));  bit_reverse!(Psi)
    Psiinv   = Complex{Float64}.(exp.(( im * big(pi) / halfN) .* idx));  bit_reverse!(Psiinv)
    roots    = Complex{Float64}.(exp.(( im * big(pi) / N) .* idx))
    rootsinv = Complex{Float64}.(exp.((-im * big(pi) / N) .* idx) / halfN)

Every BigFloat operation is one correctly rounded 256-bit operation (emulated by evaluating
sin/cos at 640 bits and rounding to 256 bits, nearest-even), then one rounding to Float64
(nearest-even; NOTE float(mpf) truncates, so to_float(..., rnd='n') is used).
Signed zeros of entry 0 follow Julia's exp(::Complex): exp(0 -/+ 0im) keeps the sign of the
zero imaginary part.

Output: tests/golden/twiddles.npz with arrays  psi_N, psiinv_N, roots_N, rootsinv_N
(complex128, length N/2) for N in SIZES.  Run from the repo root:  python tests/golden/gen_twiddles.py
"""
import os
import numpy as np
import mpmath as mp
from mpmath.libmp import libmpf

SIZES = (16, 256, 1024, 2048, 4096)


def rn(x, prec):
    """round an mpf to `prec` bits, nearest-even"""
    s, m, e, bc = x._mpf_
    return mp.mpf(libmpf.normalize(s, m, e, bc, prec, 'n'))


def f64(x):
    return libmpf.to_float(x._mpf_, rnd='n')


def sincos256(theta):
    """correctly rounded (256-bit) sin and cos of a 256-bit theta"""
    with mp.workprec(640):
        s, c = mp.sin(theta), mp.cos(theta)
    return rn(s, 256), rn(c, 256)


def bit_reverse(v):
    # src/ring/fft.jl:1-15
    v = list(v)
    n = len(v)
    j = 0
    for i in range(1, n):
        bit = n >> 1
        while j >= bit:
            j -= bit
            bit >>= 1
        j += bit
        if i < j:
            v[i], v[j] = v[j], v[i]
    return v


def tables(N):
    mp.mp.prec = 256
    pi = +mp.pi                      # big(pi): RN256(pi)
    M = N // 2
    step_M = pi / M                  # exact (power of two)
    step_N = pi / N
    psi, psiinv, roots, rootsinv = [], [], [], []
    for j in range(M):
        th = step_M * j              # RN256
        if j == 0:
            psi.append(complex(1.0, -0.0)); psiinv.append(complex(1.0, 0.0))
        else:
            s, c = sincos256(th)
            psi.append(complex(f64(c), -f64(s))); psiinv.append(complex(f64(c), f64(s)))
        ph = step_N * j
        if j == 0:
            roots.append(complex(1.0, 0.0)); rootsinv.append(complex(1.0 / M, -0.0))
        else:
            s, c = sincos256(ph)
            roots.append(complex(f64(c), f64(s)))
            rootsinv.append(complex(f64(c / M), -f64(s / M)))
    return (np.array(bit_reverse(psi)), np.array(bit_reverse(psiinv)),
            np.array(roots), np.array(rootsinv))


def main():
    out = {}
    for N in SIZES:
        p, pi_, r, ri = tables(N)
        out[f"psi_{N}"], out[f"psiinv_{N}"], out[f"roots_{N}"], out[f"rootsinv_{N}"] = p, pi_, r, ri
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "twiddles.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
