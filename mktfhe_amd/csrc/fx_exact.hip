// MKT_ARITH_EXACT on the Float64 pipe ("FX"): exact negacyclic products  digit polynomial x ring polynomial  mod 2^W from
// Float64 complex transforms WITH fused multiply-adds, the ring polynomial (a resident key) split into centered 16-bit limbs.
//
//   key word  K = sum_h limb_h 2^(16 h)  mod 2^W,  limb_h in [-2^15, 2^15)          (NL = W / 16 limbs)
//   sum_g d_g (*) K_g  =  sum_h 2^(16 h) [ sum_g d_g (*) limb_{g,h} ]  mod 2^W        (d_g the 2l gadget digit polynomials)
//
// and every bracket is an integer polynomial of magnitude <= 2l N 2^(logB-1) 2^15 -- far inside the 53-bit significand -- that
// a Float64 transform product reproduces with an absolute error proven below 1/2 (checked on the host for
// the context's gadget AND the loaded key's largest transform-domain magnitude before this path is used: fx_bound, context.cpp),
// so rounding to nearest gives the exact integer and the result is word-identical to the integer-NTT kernels (ntt_exact.hip)
// and to the big-integer restatement (tests/ref_exact.py).  The value it computes is the product the reference's transform
// approximates (src/ring/polynomial.jl:99-113 on src/ring/fft.jl:57-81) and its MultiFloat option aims at (README.md:9,
// src/ring/arithmetic.jl:11-17); the operation SEQUENCE owes the reference nothing, so this is the engine's own transform:
//
//   fold    a_j = p_j - i p_{j+M}                      Z[X]/(X^N + 1) -> C[X]/(X^M + i), M = N / 2
//   twist   b_j = a_j rho^j, rho = exp(-i pi / N)      rho^M = -i: now cyclic, C[Y]/(Y^M - 1)
//   forward cyclic DFT, Cooley-Tukey, natural in -> bit-reversed out, butterflies (x + w y, x - w y) in 6 fused operations
//           (x' = x + w y by four, y' = 2 x - x' by two); twiddles by block, fx_om; the first two stages need no product
//   inverse decimation in time, bit-reversed in -> natural out, the SAME 6-operation butterfly (the Gentleman-Sande form the
//           reference's inverse has costs 8), twiddles by position = a function of the thread alone, held in registers;
//           its first two stages need no product; untwist by conj(rho^j), the 1 / M rides on the resident key
//
// Thread mapping, exchanges and LDS staging are those of the Float64-reference transform (fft_device.h Plan / exchange):
// 4 points per thread, passes of two stages, NB transforms side by side.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernel_common.h"

namespace mktd {

namespace {

__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// (x, y) -> (x + w y, x - w y)
__device__ __forceinline__ void fx_bfly(cplx &x, cplx &y, const cplx w) {
    const double ar = fma_(-w.im, y.im, fma_(w.re, y.re, x.re));
    const double ai = fma_(w.im, y.re, fma_(w.re, y.im, x.im));
    y.re = fma_(2.0, x.re, -ar); y.im = fma_(2.0, x.im, -ai);
    x.re = ar; x.im = ai;
}
// the same with the twiddle -i w:  (-i w) y = (w y).im - i (w y).re
__device__ __forceinline__ void fx_bfly_mi(cplx &x, cplx &y, const cplx w) {
    const double ar = fma_(w.im, y.re, fma_(w.re, y.im, x.re));
    const double ai = fma_(w.im, y.im, fma_(-w.re, y.re, x.im));
    y.re = fma_(2.0, x.re, -ar); y.im = fma_(2.0, x.im, -ai);
    x.re = ar; x.im = ai;
}
// the same with the twiddle +i w:  (i w) y = -(w y).im + i (w y).re
__device__ __forceinline__ void fx_bfly_pi(cplx &x, cplx &y, const cplx w) {
    const double ar = fma_(-w.im, y.re, fma_(-w.re, y.im, x.re));
    const double ai = fma_(-w.im, y.im, fma_(w.re, y.re, x.im));
    y.re = fma_(2.0, x.re, -ar); y.im = fma_(2.0, x.im, -ai);
    x.re = ar; x.im = ai;
}
// twiddles 1, -i, +i: additions only
__device__ __forceinline__ void fx_bfly_1(cplx &x, cplx &y) { const cplx a = x, b = y; x.re = a.re + b.re; x.im = a.im + b.im; y.re = a.re - b.re; y.im = a.im - b.im; }
__device__ __forceinline__ void fx_bfly_1mi(cplx &x, cplx &y) { const cplx a = x, b = y; x.re = a.re + b.im; x.im = a.im - b.re; y.re = a.re - b.im; y.im = a.im + b.re; }
__device__ __forceinline__ void fx_bfly_1pi(cplx &x, cplx &y) { const cplx a = x, b = y; x.re = a.re - b.im; x.im = a.im + b.re; y.re = a.re + b.im; y.im = a.im - b.re; }

constexpr int FLR = 2;   // points per thread = 4 (the pass code below writes out the two-stage radix-4 shape)
static_assert(MKT_LOGR == 2, "fx_exact.hip shares the device point order of the 4-points-per-thread schedule");

// ---- forward: stage bits descend inside a pass; slot pairs of stage bit sb differ in slot bit sb ----
// twiddle of the pair group g of stage (b, sb): fx_om[2^(LOGM-1-b) + ((t >> lo) << (1 - sb)) + g]; entry 2q + 1 = -i * entry 2q; the
// stages b = LOGM-1, LOGM-2 (pass 0) have the twiddles 1 and (1, -i)
template <int LOGM, int NB, int PASS>
__device__ __forceinline__ void fx_tw_fwd(const cplx *__restrict__ om, int t, cplx (&w)[2]) {
    using P = Plan<LOGM, FLR, NB>;
    constexpr int lo = P::lo(PASS);
    if constexpr (PASS == 0) { w[0].re = w[1].re = 1.0; w[0].im = w[1].im = 0.0; }
    else if constexpr (P::nst(PASS) == 2) {
        w[0] = om[(1 << (LOGM - 2 - lo)) + (t >> lo)];            // stage bit lo + 1
        w[1] = om[(1 << (LOGM - 1 - lo)) + ((t >> lo) << 1)];     // stage bit lo
    } else {
        constexpr int b = P::hib(PASS);
        w[0] = om[(1 << (LOGM - 1 - b)) + ((t >> lo) << 1)];
        w[1] = w[0];
    }
}
template <int LOGM, int NB, int PASS, int MO>
__device__ __forceinline__ void fx_forward_pass(cplx (&z)[NB][4], const cplx *__restrict__ om, cplx *lds, int t, const LaneX &lx, const cplx (&w)[2]) {
    using P = Plan<LOGM, FLR, NB>;
    constexpr int p = PASS;
    static_assert(P::hib(0) == LOGM - 1 && P::nst(0) == 2, "pass 0 holds the two product-free stages");
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        if constexpr (p == 0) {
            fx_bfly_1(z[nb][0], z[nb][2]); fx_bfly_1(z[nb][1], z[nb][3]);
            fx_bfly_1(z[nb][0], z[nb][1]); fx_bfly_1mi(z[nb][2], z[nb][3]);
        } else if constexpr (P::nst(p) == 2) {
            fx_bfly(z[nb][0], z[nb][2], w[0]); fx_bfly(z[nb][1], z[nb][3], w[0]);
            fx_bfly(z[nb][0], z[nb][1], w[1]); fx_bfly_mi(z[nb][2], z[nb][3], w[1]);
        } else {
            fx_bfly(z[nb][0], z[nb][1], w[0]); fx_bfly_mi(z[nb][2], z[nb][3], w[0]);
        }
    }
    if constexpr (p < P::NPASS - 1) {
        cplx wn[2];
        fx_tw_fwd<LOGM, NB, PASS + 1>(om, t, wn);                // read ahead of the exchange (as fft_device.h MO bit 10)
        exchange<LOGM, FLR, NB, P::lo(p), P::lo(p + 1), true, PASS, MO>(z, lds, t, lx);
        fx_forward_pass<LOGM, NB, PASS + 1, MO>(z, om, lds, t, lx, wn);
    }
}
// In: slot e = point e*NT + t (twisted).  Out: slot e = point 4t + e of the bit-reversed frequency order.
template <int LOGM, int NB, int MO = -1>
__device__ __forceinline__ void fx_forward(cplx (&z)[NB][4], const cplx *__restrict__ om, cplx *lds, int t, const LaneX &lx) {
    cplx w0[2];
    fx_tw_fwd<LOGM, NB, 0>(om, t, w0);
    if (Route<LOGM, FLR, MO>::guard_fwd) __syncthreads();
    fx_forward_pass<LOGM, NB, 0, MO>(z, om, lds, t, lx, w0);
}

// ---- inverse (decimation in time): stage bits ascend; the pair (e, e | 1 << sb) of stage b = lo + sb multiplies its second point by
// exp(i pi j / 2^b), j = idx mod 2^b = (t mod 2^lo) | (low slot bits << lo)  =  itw[pass][sb] * i^(low slot bit) ----
template <int LOGM> struct FxItw {
    using P = Plan<LOGM, FLR>;
    cplx v[P::NPASS][2];
    __device__ __forceinline__ void load(const cplx *__restrict__ nat, int t) {
#pragma unroll
        for (int p = 0; p < P::NPASS; p++) {
            const int lo = P::lo(p), tl = t & ((1 << lo) - 1);
            if (lo == 0) { v[p][0].re = v[p][1].re = 1.0; v[p][0].im = v[p][1].im = 0.0; continue; }
            if (P::nst(p) == 2) { v[p][0] = nat[(1 << lo) + tl]; v[p][1] = nat[(2 << lo) + tl]; }
            else { v[p][0] = nat[(1 << P::hib(p)) + tl]; v[p][1] = v[p][0]; }
        }
    }
};
template <int LOGM, int NB, int PASS, int MO>
__device__ __forceinline__ void fx_inverse_pass(cplx (&z)[NB][4], const FxItw<LOGM> &c, cplx *lds, int t, const LaneX &lx) {
    using P = Plan<LOGM, FLR, NB>;
    constexpr int p = PASS, lo = P::lo(p);
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        if constexpr (lo == 0 && P::nst(p) == 2) {
            fx_bfly_1(z[nb][0], z[nb][1]); fx_bfly_1(z[nb][2], z[nb][3]);
            fx_bfly_1(z[nb][0], z[nb][2]); fx_bfly_1pi(z[nb][1], z[nb][3]);
        } else if constexpr (lo == 0) {
            fx_bfly_1(z[nb][0], z[nb][1]); fx_bfly_1(z[nb][2], z[nb][3]);
        } else if constexpr (P::nst(p) == 2) {
            fx_bfly(z[nb][0], z[nb][1], c.v[p][0]); fx_bfly(z[nb][2], z[nb][3], c.v[p][0]);
            fx_bfly(z[nb][0], z[nb][2], c.v[p][1]); fx_bfly_pi(z[nb][1], z[nb][3], c.v[p][1]);
        } else {
            fx_bfly(z[nb][0], z[nb][1], c.v[p][0]); fx_bfly(z[nb][2], z[nb][3], c.v[p][0]);
        }
    }
    if constexpr (p > 0) {
        exchange<LOGM, FLR, NB, P::lo(p), P::lo(p - 1), false, PASS, MO>(z, lds, t, lx);
        fx_inverse_pass<LOGM, NB, PASS - 1, MO>(z, c, lds, t, lx);
    }
}
// In: slot e = point 4t + e (bit-reversed frequency order).  Out: slot e = coefficient point e*NT + t, unscaled, twisted.
template <int LOGM, int NB, int MO = -1>
__device__ __forceinline__ void fx_inverse(cplx (&z)[NB][4], const FxItw<LOGM> &c, cplx *lds, int t, const LaneX &lx) {
    if (Route<LOGM, FLR, MO>::guard_inv) __syncthreads();
    fx_inverse_pass<LOGM, NB, Plan<LOGM, FLR, NB>::NPASS - 1, MO>(z, c, lds, t, lx);
}

// centered 16-bit limb h of a ring word: w = sum_h limb_h 2^(16 h) mod 2^W, limb_h in [-2^15, 2^15)
template <typename WORD>
__device__ __forceinline__ int limb_of(WORD w, int h) {
    typedef typename WordTraits<WORD>::S SW;
    WORD v = w;                                                   // every step exact mod 2^W (a word near 2^(W-1) wraps into the top limb)
    int r = 0;
    for (int q = 0; q <= h; q++) { r = (int)(int16_t)(uint16_t)v; v = (WORD)((SW)(WORD)(v - (WORD)(SW)r) >> 16); }
    return r;
}
// nearest integer of q (|q| < 2^51) as a two's-complement 64-bit pattern PLUS FX_MAGIC_BITS: the significand of q + 1.5 * 2^52
constexpr double FX_MAGIC = 6755399441055744.0;                  // 1.5 * 2^52
constexpr uint32_t FX_MAGIC_HI = 0x43380000u;                    // high dword of its bit pattern; the low dword is 0
__device__ __forceinline__ uint64_t round_bits(double q) { return (uint64_t)__double_as_longlong(q + FX_MAGIC); }

// -------------------------------------------------------------------------------------------------------------------
// Key pre-transform: NP coefficient-form ring polynomials -> NL limb transforms each, scaled by 1 / M, device point order
// (fft_device.h dev_pos order 1).  kmax: the largest |transform value|^2 of the launch (a positive double as its bit pattern,
// atomicMax): what the host's error bound takes for the key (context.cpp fx_bound).
// -------------------------------------------------------------------------------------------------------------------
template <int LOGM, typename WORD>
__global__ __launch_bounds__((Plan<LOGM, FLR>::NT)) void fx_key_fwd_kernel(const cplx *__restrict__ om, const cplx *__restrict__ twist, const WORD *__restrict__ p,
                                                                        cplx *__restrict__ out, size_t NP, unsigned long long *__restrict__ kmax) {
    using P = Plan<LOGM, FLR>;
    constexpr int NT = P::NT, M = P::M, N = 2 * M, NL = WordTraits<WORD>::W / 16;
    cplx *lds = reinterpret_cast<cplx *>(mkt_smem);
    const int t = threadIdx.x;
    const LaneX lx = make_lanex();
    double mx = 0.0;
    for (size_t b = blockIdx.x; b < NP; b += gridDim.x) {
#pragma unroll 1
        for (int h = 0; h < NL; h++) {
            cplx z[1][4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int j = e * NT + t;
                const double d0 = (double)limb_of<WORD>(p[b * N + j], h), d1 = (double)limb_of<WORD>(p[b * N + M + j], h);
                const cplx r = twist[j];
                z[0][e].re = fma_(d0, r.re, d1 * r.im);           // (d0 - i d1) (r.re + i r.im)
                z[0][e].im = fma_(d0, r.im, -(d1 * r.re));
            }
            __syncthreads();
            fx_forward<LOGM, 1>(z, om, lds, t, lx);
            cplx *o = out + (b * NL + h) * (size_t)M;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                cplx v; v.re = z[0][e].re * (1.0 / M); v.im = z[0][e].im * (1.0 / M);
                o[dev_pos(1, t * 4 + e, NT)] = v;
                mx = fmax(mx, fma_(z[0][e].re, z[0][e].re, z[0][e].im * z[0][e].im));   // unscaled: |K_r|^2
            }
        }
    }
    if (kmax) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mx = fmax(mx, __shfl_xor(mx, d));
        if ((t & 63) == 0) atomicMax(kmax, (unsigned long long)__double_as_longlong(mx));
    }
}

// -------------------------------------------------------------------------------------------------------------------
// Exact negacyclic product out = a (*) b mod 2^W of a digit polynomial a (|a_i| <= amax) and a ring polynomial b, transform level
// (tests; the FX implementation of mkt_exact_polymul_batch).  resid: the largest distance |q - round(q)| met (bit pattern,
// atomicMax) -- the measured counterpart of the proven bound.
// -------------------------------------------------------------------------------------------------------------------
template <int LOGM, typename WORD>
__global__ __launch_bounds__((Plan<LOGM, FLR>::NT)) void fx_polymul_kernel(const cplx *__restrict__ om, const cplx *__restrict__ twist, const cplx *__restrict__ nat,
                                                                        const WORD *__restrict__ a, const WORD *__restrict__ bp, WORD *__restrict__ out, size_t B,
                                                                        unsigned long long *__restrict__ resid) {
    using P = Plan<LOGM, FLR>;
    typedef typename WordTraits<WORD>::S SW;
    constexpr int NT = P::NT, M = P::M, N = 2 * M, W = WordTraits<WORD>::W, NL = W / 16;
    cplx *lds = reinterpret_cast<cplx *>(mkt_smem);
    const int t = threadIdx.x;
    const LaneX lx = make_lanex();
    FxItw<LOGM> itw; itw.load(nat, t);
    double rmax = 0.0;
    for (size_t b = blockIdx.x; b < B; b += gridDim.x) {
        cplx za[1][4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int j = e * NT + t;
            const double d0 = (double)(SW)a[b * N + j], d1 = (double)(SW)a[b * N + M + j];
            const cplx r = twist[j];
            za[0][e].re = fma_(d0, r.re, d1 * r.im); za[0][e].im = fma_(d0, r.im, -(d1 * r.re));
        }
        __syncthreads();
        fx_forward<LOGM, 1>(za, om, lds, t, lx);
        WORD acc[4][2];
#pragma unroll
        for (int e = 0; e < 4; e++) acc[e][0] = acc[e][1] = 0;
#pragma unroll 1
        for (int h = 0; h < NL; h++) {
            cplx zb[1][4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int j = e * NT + t;
                const double d0 = (double)limb_of<WORD>(bp[b * N + j], h), d1 = (double)limb_of<WORD>(bp[b * N + M + j], h);
                const cplx r = twist[j];
                zb[0][e].re = fma_(d0, r.re, d1 * r.im); zb[0][e].im = fma_(d0, r.im, -(d1 * r.re));
            }
            __syncthreads();
            fx_forward<LOGM, 1>(zb, om, lds, t, lx);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const cplx x = za[0][e], y = zb[0][e];
                cplx v; v.re = fma_(x.re, y.re, -(x.im * y.im)) * (1.0 / M); v.im = fma_(x.re, y.im, x.im * y.re) * (1.0 / M);
                zb[0][e] = v;
            }
            __syncthreads();
            fx_inverse<LOGM, 1>(zb, itw, lds, t, lx);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const cplx r = twist[e * NT + t], v = zb[0][e];   // untwist: v * conj(r)
                const double q0 = fma_(v.re, r.re, v.im * r.im), q1 = -fma_(v.im, r.re, -(v.re * r.im));
                const uint64_t b0 = round_bits(q0), b1 = round_bits(q1);
                rmax = fmax(rmax, fmax(fabs(q0 - ((q0 + FX_MAGIC) - FX_MAGIC)), fabs(q1 - ((q1 + FX_MAGIC) - FX_MAGIC))));
                acc[e][0] = (WORD)(acc[e][0] + (WORD)((b0 - ((uint64_t)FX_MAGIC_HI << 32)) << (16 * h)));
                acc[e][1] = (WORD)(acc[e][1] + (WORD)((b1 - ((uint64_t)FX_MAGIC_HI << 32)) << (16 * h)));
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) { out[b * N + e * NT + t] = acc[e][0]; out[b * N + M + e * NT + t] = acc[e][1]; }
    }
    if (resid) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, d));
        if ((t & 63) == 0) atomicMax(resid, (unsigned long long)__double_as_longlong(rmax));
    }
}

// -------------------------------------------------------------------------------------------------------------------
// Blind rotation with exact products on the Float64 pipe, RLWE length 1: CGGI (bootstrapping.jl:32-76) and every row of KMS
// phase 1 (:389-443).  One workgroup per rotation; the accumulator (b, a), the 2l digit transforms of a step and the
// inverse's twiddles live in registers; LDS stages the exchanges, holds fx_om, and turns the lifted sum by X^at.
// Per step (:411-438):  S_c = sum_g D_g (.) K[i][g][c][limb]  per output polynomial c and limb (NB = 2 limbs side by side: :427-432),
// inverse, round, recombine  w_c = sum_h round(S_{c,h}) 2^(16 h) mod 2^W,  then  acc_c += X^at w_c - w_c  on the integers (:435-437:
// the same words as the transform-domain monomial product, and the rounded integer is half as large).
// brk: [party][n][2l][2][NL][M], device point order, scaled by 1 / M.
// -------------------------------------------------------------------------------------------------------------------
// Key rows are requested in explicit UNITS -- the KBU points of one digit, both limbs of the group (2 KBU loads into their own registers) -- a fixed number of units
// (KBLA) ahead of the multiply-adds that use them, fenced with sched_barrier.  Left to itself the compiler keeps ~10 loads in flight at some instantiations and collapses
// to ONE register quad for all loads of a group at others (64-bit ring at gadget length 3 or N >= 2048: load, wait, four multiply-adds, load ... KMS2party 284 ms against
// 102 ms).  Units and depth by gadget length (the 2l digit transforms hold 64 / 96 registers): measured in profiles/r06_experiments.txt items 3, 4, where also what was
// tried around it and not kept (rows requested ahead of the digit transforms or of every inverse, the turn per output polynomial, inverse twiddles from the LDS table,
// other exchange routes and LLVM schedulers).  Speed only: tools/variant.sh fx_exact <sfx> "-DMKT_FX_KBU3=2 ...".
#ifndef MKT_FX_KBU3
#define MKT_FX_KBU3 1
#endif
#ifndef MKT_FX_KBLA3
#define MKT_FX_KBLA3 3
#endif
#ifndef MKT_FX_KBU2
#define MKT_FX_KBU2 1
#endif
#ifndef MKT_FX_KBLA2
#define MKT_FX_KBLA2 4
#endif
template <int LOGM, typename WORD, int LT>
__global__ __launch_bounds__((Plan<LOGM, FLR>::NT)) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fx_blindrotate_kernel(const FxRotArgs a) {
    constexpr int NB = 2;
    using P = Plan<LOGM, FLR, NB>;
    constexpr int R = 4, NT = P::NT, M = P::M, N = 2 * M, W = WordTraits<WORD>::W, NL = W / 16, G2 = 2 * LT;
    constexpr int MO = -1;
    constexpr int KBU = LT >= 3 ? MKT_FX_KBU3 : MKT_FX_KBU2, UPG = R / KBU, NU = G2 * UPG, LA = LT >= 3 ? MKT_FX_KBLA3 : MKT_FX_KBLA2;
    cplx *lds = reinterpret_cast<cplx *>(mkt_smem);
    const int t = threadIdx.x;
    const LaneX lx = make_lanex();
    const unsigned bid = blockIdx.x + a.block0;
    if (a.stagger > 0 && ((bid >> 8) & 1)) {
        for (int s = 0; s < a.stagger; s++) __builtin_amdgcn_s_sleep(8);
    }
    cplx *om_l = lds + P::LDS_CPLX;
    for (int i = t; i < M; i += NT) om_l[i] = a.om[i];
    __syncthreads();
    size_t gate; int slot;
    {
        RotArgs ra{}; ra.map_mode = a.map_mode; ra.ngates = a.ngates; ra.rows_per_gate = a.rows_per_gate; ra.slot_party = a.slot_party;
        rot_decode(ra, bid, gate, slot);
    }
    const size_t rot = gate * (size_t)a.rows_per_gate + slot;
    const int party = __builtin_amdgcn_readfirstlane(a.slot_party[slot]), row = __builtin_amdgcn_readfirstlane(a.slot_row[slot]);
    const uint32_t *at_src = a.lwe + gate * (size_t)a.lwe_stride + (size_t)party * a.n;
    const cplx *brk = a.brk + (size_t)party * a.brk_party_stride;
    const __amdgpu_buffer_rsrc_t rs_brk = table_rsrc(brk, (size_t)a.brk_party_stride * sizeof(cplx));
    unsigned vo_dev[R];
#pragma unroll
    for (int e = 0; e < R; e++) vo_dev[e] = (unsigned)dev_pos(1, t * R + e, NT) * 16u;
    const Gadget<WORD> gd(LT, a.logB);
    cplx rt[R];                                   // rho^j of this thread's points j = e*NT + t (twist; the untwist conjugates it)
#pragma unroll
    for (int e = 0; e < R; e++) rt[e] = a.twist[e * NT + t];
    FxItw<LOGM> itw; itw.load(a.nat, t);

    WORD acc[2][R][2];
    WORD *accg = reinterpret_cast<WORD *>(a.acc_io) + rot * 2 * N;
    if (a.init_mode == 0) {
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < R; e++) { acc[c][e][0] = accg[c * N + e * NT + t]; acc[c][e][1] = accg[c * N + M + e * NT + t]; }
    } else {   // bootstrapping.jl:403-406
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < R; e++) { acc[c][e][0] = 0; acc[c][e][1] = 0; }
        if (t == 0) acc[0][0][0] = (WORD)1 << (W - (row + 1) * a.logB_lev);
    }

    const int msbit = 32 - a.logN - 1;
    uint32_t at_raw = at_src[0];
    for (int i = 0; i < a.n; i++) {
        const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.pre_switched ? at_raw : divbits<uint32_t>(at_raw, msbit)));
        at_raw = at_src[i + 1 < a.n ? i + 1 : i];
        if (at == 0) continue;                                          // :48 / :413

        // the 2l digit transforms (b digits, then a digits), two at a time, all kept
        cplx D[G2][R];
#pragma unroll
        for (int g0 = 0; g0 < G2; g0 += NB) {
            cplx z[NB][R];
#pragma unroll
            for (int h2 = 0; h2 < NB; h2++) {
                const int g = g0 + h2;
                const bool isa = g >= LT;
                const int j = isa ? g - LT : g;
#pragma unroll
                for (int e = 0; e < R; e++) {
                    const WORD w0 = isa ? acc[1][e][0] : acc[0][e][0], w1 = isa ? acc[1][e][1] : acc[0][e][1];
                    const double d0 = (double)gd.digit(gd.prep(w0), j), d1 = (double)gd.digit(gd.prep(w1), j);
                    z[h2][e].re = fma_(d0, rt[e].re, d1 * rt[e].im);
                    z[h2][e].im = fma_(d0, rt[e].im, -(d1 * rt[e].re));
                }
            }
            fx_forward<LOGM, NB, MO>(z, om_l, lds, t, lx);
#pragma unroll
            for (int h2 = 0; h2 < NB; h2++)
#pragma unroll
                for (int e = 0; e < R; e++) D[g0 + h2][e] = z[h2][e];
        }

        const unsigned so_bit = (unsigned)((size_t)i * G2 * 2 * NL * M * sizeof(cplx));
        WORD wsum[2][R][2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
#pragma unroll
            for (int h0 = 0; h0 < NL; h0 += NB) {
                cplx S[NB][R];
#pragma unroll
                for (int h2 = 0; h2 < NB; h2++)
#pragma unroll
                    for (int e = 0; e < R; e++) { S[h2][e].re = 0.0; S[h2][e].im = 0.0; }
                cplx kb[LA + 1][KBU][NB];
                auto req = [&](int u) {
                    const int g = u / UPG, e0 = (u % UPG) * KBU;
                    const unsigned so_row = so_bit + (unsigned)((((size_t)g * 2 + c) * NL + h0) * M * sizeof(cplx));
#pragma unroll
                    for (int e = 0; e < KBU; e++)
#pragma unroll
                        for (int h2 = 0; h2 < NB; h2++) kb[u % (LA + 1)][e][h2] = table_load(rs_brk, vo_dev[e0 + e], so_row + (unsigned)(h2 * M * sizeof(cplx)));
                };
#pragma unroll
                for (int u = 0; u < LA && u < NU; u++) req(u);
#pragma unroll
                for (int u = 0; u < NU; u++) {                          // :427-432 S += D_g (.) K[i][g][c][limb]
                    if (u + LA < NU) req(u + LA);
                    __builtin_amdgcn_sched_barrier(0);
                    const int g = u / UPG, e0 = (u % UPG) * KBU;
#pragma unroll
                    for (int e = 0; e < KBU; e++)
#pragma unroll
                        for (int h2 = 0; h2 < NB; h2++) {
                            const cplx k = kb[u % (LA + 1)][e][h2], d = D[g][e0 + e];
                            S[h2][e0 + e].re = fma_(-d.im, k.im, fma_(d.re, k.re, S[h2][e0 + e].re));
                            S[h2][e0 + e].im = fma_(d.im, k.re, fma_(d.re, k.im, S[h2][e0 + e].im));
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                fx_inverse<LOGM, NB, MO>(S, itw, lds, t, lx);
#pragma unroll
                for (int e = 0; e < R; e++) {
                    uint64_t b0[NB], b1[NB];
#pragma unroll
                    for (int h2 = 0; h2 < NB; h2++) {                   // untwist by conj(rho^j), nearest integer
                        const cplx v = S[h2][e];
                        b0[h2] = round_bits(fma_(v.re, rt[e].re, v.im * rt[e].im));
                        b1[h2] = round_bits(fma_(v.re, rt[e].im, -(v.im * rt[e].re)));     // -(Im)
                    }
                    // sum_h (bits_h - MAGIC) 2^(16 h): MAGIC's pattern has 48 zero low bits, so only limb 0 carries it
                    if constexpr (W == 32) {
                        wsum[c][e][0] = (WORD)((uint32_t)b0[0] + ((uint32_t)b0[1] << 16));
                        wsum[c][e][1] = (WORD)((uint32_t)b1[0] + ((uint32_t)b1[1] << 16));
                    } else if (h0 == 0) {
                        wsum[c][e][0] = (WORD)((b0[0] - ((uint64_t)FX_MAGIC_HI << 32)) + (b0[1] << 16));
                        wsum[c][e][1] = (WORD)((b1[0] - ((uint64_t)FX_MAGIC_HI << 32)) + (b1[1] << 16));
                    } else {
                        wsum[c][e][0] = (WORD)(wsum[c][e][0] + ((uint64_t)((uint32_t)b0[0] + ((uint32_t)b0[1] << 16)) << 32));
                        wsum[c][e][1] = (WORD)(wsum[c][e][1] + ((uint64_t)((uint32_t)b1[0] + ((uint32_t)b1[1] << 16)) << 32));
                    }
                }
            }
        }
        // :435-437 acc += X^at w - w, the turn through LDS: (X^at w)[i] = +-w[i - at mod N]
        WORD *wl = reinterpret_cast<WORD *>(lds);
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < R; e++) { wl[c * N + e * NT + t] = wsum[c][e][0]; wl[c * N + M + e * NT + t] = wsum[c][e][1]; }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < R; e++)
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    const uint32_t src = (uint32_t)(hf * M + e * NT + t - (int)at) & (2u * N - 1u);
                    const WORD v = wl[c * N + (src & (N - 1))];
                    acc[c][e][hf] = (WORD)(acc[c][e][hf] + (src >= (uint32_t)N ? (WORD)0 - v : v) - wsum[c][e][hf]);
                }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int e = 0; e < R; e++) { accg[c * N + e * NT + t] = acc[c][e][0]; accg[c * N + M + e * NT + t] = acc[c][e][1]; }
}

template <int LM, typename WORD, int LT>
hipError_t fx_rot_launch(const FxRotArgs &a, size_t nrot, hipStream_t s) {
    using P = Plan<LM, FLR, 2>;
    constexpr size_t LB = P::LDS_BYTES + (size_t)P::M * sizeof(cplx);
    static_assert(P::LDS_BYTES >= (size_t)4 * P::M * sizeof(WORD), "the staging buffers hold the turned sums of both polynomials");
    hipError_t e = set_lds(fx_blindrotate_kernel<LM, WORD, LT>, LB);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((fx_blindrotate_kernel<LM, WORD, LT>), dim3((unsigned)nrot), dim3(P::NT), LB, s, a);
    return hipGetLastError();
}

}  // namespace

#define MKT_FX_DISPATCH(logM, ...)                   \
    switch (logM) {                                  \
    case 6:  { constexpr int LM = 6;  __VA_ARGS__; } break; \
    case 7:  { constexpr int LM = 7;  __VA_ARGS__; } break; \
    case 8:  { constexpr int LM = 8;  __VA_ARGS__; } break; \
    case 9:  { constexpr int LM = 9;  __VA_ARGS__; } break; \
    case 10: { constexpr int LM = 10; __VA_ARGS__; } break; \
    case 11: { constexpr int LM = 11; __VA_ARGS__; } break; \
    default: return hipErrorInvalidValue;            \
    }

bool fx_supported(int logM, int W, int l) { return logM >= 6 && logM <= 11 && (W == 32 || W == 64) && (l == 2 || l == 3); }

hipError_t launch_fx_key_fwd(int logM, int W, const cplx *om, const cplx *twist, const void *p, cplx *out, size_t np, unsigned long long *kmax, hipStream_t s) {
    if (!np) return hipSuccess;
    const int grid = (int)(np < 16384 ? np : 16384);
    MKT_FX_DISPATCH(logM, {
        using P = Plan<LM, FLR, 1>;
        if (W == 64) { hipError_t e = set_lds(fx_key_fwd_kernel<LM, uint64_t>, P::LDS_BYTES); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fx_key_fwd_kernel<LM, uint64_t>), dim3(grid), dim3(P::NT), P::LDS_BYTES, s, om, twist, (const uint64_t *)p, out, np, kmax); }
        else { hipError_t e = set_lds(fx_key_fwd_kernel<LM, uint32_t>, P::LDS_BYTES); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fx_key_fwd_kernel<LM, uint32_t>), dim3(grid), dim3(P::NT), P::LDS_BYTES, s, om, twist, (const uint32_t *)p, out, np, kmax); }
    });
    return hipGetLastError();
}

hipError_t launch_fx_polymul(int logM, int W, const cplx *om, const cplx *twist, const cplx *nat, const void *a, const void *b, void *out, size_t B,
                             unsigned long long *resid, hipStream_t s) {
    if (!B) return hipSuccess;
    const int grid = (int)(B < 16384 ? B : 16384);
    MKT_FX_DISPATCH(logM, {
        using P = Plan<LM, FLR, 1>;
        if (W == 64) { hipError_t e = set_lds(fx_polymul_kernel<LM, uint64_t>, P::LDS_BYTES); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fx_polymul_kernel<LM, uint64_t>), dim3(grid), dim3(P::NT), P::LDS_BYTES, s, om, twist, nat, (const uint64_t *)a, (const uint64_t *)b, (uint64_t *)out, B, resid); }
        else { hipError_t e = set_lds(fx_polymul_kernel<LM, uint32_t>, P::LDS_BYTES); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fx_polymul_kernel<LM, uint32_t>), dim3(grid), dim3(P::NT), P::LDS_BYTES, s, om, twist, nat, (const uint32_t *)a, (const uint32_t *)b, (uint32_t *)out, B, resid); }
    });
    return hipGetLastError();
}

hipError_t launch_fx_blindrotate(int logM, int W, const FxRotArgs &a, size_t nrot, hipStream_t s) {
    if (!nrot) return hipSuccess;
    if (!fx_supported(logM, W, a.l)) return hipErrorInvalidValue;
    last_rot_kernel = "fx_blindrotate_kernel";
    // One launch per chip-fill on the 64-bit ring up to N = 1024.  Every rotation of a party walks the same key rows, step by step, and the workgroups of ONE fill run in
    // lock-step, so a fill reads each row from the fabric once per L2 and then hits; in a single launch of several fills the later workgroups start whenever a slot frees,
    // the resident ones spread over all key bits and the 4 x larger key of this arithmetic no longer fits the L2s (headline: hit rate 0.66 -> 0.96, 34.3 -> 31.6 ms).  At
    // N = 2048 (512 workgroups per fill, hit rate 0.98 either way) the six launch tails cost more than the locality gives (KMS2party 105.2 vs 101.7 ms in one launch), on the
    // 32-bit ring it makes no difference (profiles/r06_experiments.txt).  a.split: 0 = this rule, -1 = one launch, > 0 = that many workgroups per launch.
    size_t chunk = nrot;
    if (a.split == 0 && W == 64 && logM <= 9) chunk = (size_t)256 * 4;
    else if (a.split > 0) chunk = (size_t)a.split;
    for (size_t b0 = 0; b0 < nrot; b0 += chunk) {
        FxRotArgs b = a;
        b.block0 = a.block0 + (unsigned)b0;
        const size_t n = nrot - b0 < chunk ? nrot - b0 : chunk;
        hipError_t e = hipSuccess;
        MKT_FX_DISPATCH(logM, {
            if (W == 64) e = a.l == 2 ? fx_rot_launch<LM, uint64_t, 2>(b, n, s) : fx_rot_launch<LM, uint64_t, 3>(b, n, s);
            else e = a.l == 2 ? fx_rot_launch<LM, uint32_t, 2>(b, n, s) : fx_rot_launch<LM, uint32_t, 3>(b, n, s);
        });
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace mktd
