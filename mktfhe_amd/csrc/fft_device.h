// Device primitives of the F64REF negacyclic transform for gfx950 (wave64, LDS-staged).
//
// Reference semantics (operation for operation, IEEE double, no FMA contraction):
//   forward  fftto!  src/ring/fft.jl:57-63  + fft!  :105-155 (Cooley-Tukey, bit-reversed output)
//   inverse  ifftto! src/ring/fft.jl:74-81  + ifft! :159-209 (Gentleman-Sande) + native arithmetic.jl:1-9
//   complex product  (xr*yr - xi*yi, xr*yi + xi*yr)  (Julia Base complex `*`)
//
// Mapping.  One transform of M = 2^LOGM complex points is owned by NT = M / R threads, each holding
// R = 2^LOGR points in registers.  The radix-2 butterfly network of the reference is executed in
// "passes" of LOGR consecutive stages that are local to a thread; between passes the points are
// re-distributed through LDS.  In a pass with window low bit `lo`, thread t holds in slot e the point
//      idx(t, e, lo) = ((t >> lo) << (lo + LOGR)) | (e << lo) | (t & (2^lo - 1)).
// The first forward pass (window at the top bits) holds idx = e*NT + t, the last holds idx = t*R + e;
// the inverse runs the same windows backwards, so a (forward -> pointwise -> inverse) chain needs no
// re-distribution besides the in-transform exchanges, and the 2 coefficients (idx, idx + M) folded
// into point idx stay with one thread for the whole blind rotation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

#ifndef MKT_ABLATE
#define MKT_ABLATE 0   // timing-only experiments: 1 no key loads, 2 no twiddle loads, 4 no LDS exchange, 8 no barrier
#endif

namespace mktd {

struct __attribute__((aligned(16))) cplx { double re, im; };

__device__ __forceinline__ cplx cmul(const cplx x, const cplx y) {
    const double a = x.re * y.re, b = x.im * y.im, c = x.re * y.im, d = x.im * y.re;
    cplx r; r.re = a - b; r.im = c + d; return r;
}
// x * conj(w), bit-identical to cmul(x, (w.re, -w.im)): a - (-b) == a + b and (-c) + d == d - c in IEEE-754
__device__ __forceinline__ cplx cmul_conj(const cplx x, const cplx w) {
    const double a = x.re * w.re, b = x.im * w.im, c = x.re * w.im, d = x.im * w.re;
    cplx r; r.re = a + b; r.im = d - c; return r;
}
__device__ __forceinline__ cplx cadd(const cplx x, const cplx y) { cplx r; r.re = x.re + y.re; r.im = x.im + y.im; return r; }
__device__ __forceinline__ cplx csub(const cplx x, const cplx y) { cplx r; r.re = x.re - y.re; r.im = x.im - y.im; return r; }

template <int LOGM, int LOGR, int NB = 1>
struct Plan {
    static constexpr int M = 1 << LOGM, R = 1 << LOGR, NT = M / R;
    static constexpr int NPASS = (LOGM + LOGR - 1) / LOGR;
    static_assert(LOGM >= LOGR, "transform smaller than a thread's share");
    __host__ __device__ static constexpr int lo(int p) { return (LOGM - (p + 1) * LOGR) < 0 ? 0 : (LOGM - (p + 1) * LOGR); }
    __host__ __device__ static constexpr int nst(int p) { return p < NPASS - 1 ? LOGR : LOGM - (NPASS - 1) * LOGR; }
    // highest stage bit of pass p
    __host__ __device__ static constexpr int hib(int p) { return p < NPASS - 1 ? lo(p) + LOGR - 1 : nst(p) - 1; }
    // LDS staging: NB transforms side by side, two buffers (one barrier per exchange)
    static constexpr int BUF = NB * M;
    static constexpr int LDS_CPLX = 2 * BUF;
    static constexpr size_t LDS_BYTES = (size_t)LDS_CPLX * sizeof(cplx);
};

template <int LOGR>
__device__ __forceinline__ int pt_index(int t, int e, int lo) {
    return ((t >> lo) << (lo + LOGR)) | (e << lo) | (t & ((1 << lo) - 1));
}
// XOR swizzle of the staging slot: conflict-free ds_write_b128 / ds_read_b128 for every exchange pattern of
// the 4- and 8-points-per-thread schedules at M = 128 .. 2048 (found and verified by tools/lds_swizzle_search.py,
// which simulates the gfx950 b128 lane groups and bank widths)
template <int LOGR>
__device__ __forceinline__ int lds_pos(int idx) {
    if (LOGR == 2) return idx ^ ((idx >> 1) & 8) ^ ((idx >> 2) & 15);
    return idx ^ ((idx >> 3) & 15);   // 8 points per thread
}

template <int LOGM, int LOGR, int NB>
__device__ __forceinline__ void exchange(cplx (&z)[NB][1 << LOGR], cplx *buf, int t, int lo_from, int lo_to) {
    constexpr int R = 1 << LOGR, M = 1 << LOGM;
    if (MKT_ABLATE & 4) return;
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int e = 0; e < R; e++) buf[b * M + lds_pos<LOGR>(pt_index<LOGR>(t, e, lo_from))] = z[b][e];
    if (!(MKT_ABLATE & 8)) __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int e = 0; e < R; e++) z[b][e] = buf[b * M + lds_pos<LOGR>(pt_index<LOGR>(t, e, lo_to))];
}

// fft.jl:105-155: for stage bit b (stride k = 2^b, m = 2^(LOGM-1-b)), butterfly on (j, j+k):
//   u = a[j+k] * Psi[m + (j >> (b+1))];  a[j], a[j+k] = a[j] + u, a[j] - u
// In: slot e = point e*NT + t.  Out: slot e = point t*R + e.  NB independent transforms share the
// twiddle loads and the barriers.
template <int LOGM, int LOGR, int NB>
__device__ __forceinline__ void fft_forward(cplx (&z)[NB][1 << LOGR], const cplx *__restrict__ psi, cplx *lds, int t) {
    using P = Plan<LOGM, LOGR, NB>;
    // with an even pass count two back-to-back transforms of the same direction would start writing
    // the staging buffer the previous one may still be reading
    if (P::NPASS > 1 && (P::NPASS & 1) == 0) __syncthreads();
#pragma unroll
    for (int p = 0; p < P::NPASS; p++) {
        const int lo = P::lo(p);
#pragma unroll
        for (int s = 0; s < P::nst(p); s++) {
            const int b = P::hib(p) - s, sb = b - lo;
            const int twbase = (1 << (LOGM - 1 - b)) + ((t >> lo) << (LOGR - 1 - sb));
#pragma unroll
            for (int g = 0; g < (1 << (LOGR - 1 - sb)); g++) {
                cplx w; if (MKT_ABLATE & 2) { w.re = 0.5 + twbase; w.im = 0.25 * g; } else w = psi[twbase + g];
#pragma unroll
                for (int q = 0; q < (1 << sb); q++) {
                    const int e = (g << (sb + 1)) | q, e2 = e | (1 << sb);
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        const cplx u = cmul(z[nb][e2], w);
                        const cplx a = z[nb][e];
                        z[nb][e] = cadd(a, u); z[nb][e2] = csub(a, u);
                    }
                }
            }
        }
        if (p < P::NPASS - 1) exchange<LOGM, LOGR, NB>(z, lds + (p & 1) * P::BUF, t, lo, P::lo(p + 1));
    }
}

// fft.jl:159-209: t, u = a[j], a[j+k];  a[j] = t + u;  a[j+k] = (t - u) * Psiinv[m + (j >> (b+1))]
// In: slot e = point t*R + e.  Out: slot e = point e*NT + t.
// CONJ: `psiinv` points at the FORWARD table Psi and the butterflies multiply by its conjugate (Psiinv == conj(Psi)
// entry for entry, fft.jl:33-34), so one table -- e.g. a copy resident in LDS -- serves both directions.
template <int LOGM, int LOGR, int NB, bool CONJ = false>
__device__ __forceinline__ void fft_inverse(cplx (&z)[NB][1 << LOGR], const cplx *__restrict__ psiinv, cplx *lds, int t) {
    using P = Plan<LOGM, LOGR, NB>;
    if (P::NPASS > 1 && (P::NPASS & 1) == 0) __syncthreads();
#pragma unroll
    for (int p = P::NPASS - 1; p >= 0; p--) {
        const int lo = P::lo(p);
#pragma unroll
        for (int s = P::nst(p) - 1; s >= 0; s--) {
            const int b = P::hib(p) - s, sb = b - lo;
            const int twbase = (1 << (LOGM - 1 - b)) + ((t >> lo) << (LOGR - 1 - sb));
#pragma unroll
            for (int g = 0; g < (1 << (LOGR - 1 - sb)); g++) {
                cplx w; if (MKT_ABLATE & 2) { w.re = 0.5 + twbase; w.im = 0.25 * g; } else w = psiinv[twbase + g];
#pragma unroll
                for (int q = 0; q < (1 << sb); q++) {
                    const int e = (g << (sb + 1)) | q, e2 = e | (1 << sb);
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        const cplx a = z[nb][e], u = z[nb][e2];
                        z[nb][e] = cadd(a, u);
                        z[nb][e2] = CONJ ? cmul_conj(csub(a, u), w) : cmul(csub(a, u), w);
                    }
                }
            }
        }
        if (p > 0) exchange<LOGM, LOGR, NB>(z, lds + (p & 1) * P::BUF, t, lo, P::lo(p - 1));
    }
}

// Device point order of the resident TransPolys (keys, monomial table, phase-1 output): where the
// reference-order point x = 4t+e (slot e of thread t after the forward transform) is stored.
#ifndef MKT_DEVORDER
#define MKT_DEVORDER 2
#endif
#ifndef MKT_LOGR
#define MKT_LOGR 2   // points per thread = 2^MKT_LOGR in every transform schedule (and in dev_pos)
#endif
__host__ __device__ __forceinline__ int dev_pos(int x, int NT) {
    constexpr int LR = MKT_LOGR, RM = (1 << LR) - 1;
#if MKT_DEVORDER == 0
    (void)NT; return x;                                            // reference order: 16*R B per lane
#elif MKT_DEVORDER == 1
    return (x & RM) * NT + (x >> LR);                              // slot-major: 16 B per lane, wave-contiguous
#else
    return ((x & RM) >> 1) * (2 * NT) + ((x >> LR) << 1) + (x & 1); // slot pairs: 32 B per lane
#endif
}

// ---- ring words ----
template <typename WORD> struct WordTraits;
template <> struct WordTraits<uint32_t> { static constexpr int W = 32; typedef int32_t S; };
template <> struct WordTraits<uint64_t> { static constexpr int W = 64; typedef int64_t S; };

// signed(x) converted to Float64 (fft.jl:60); round-to-nearest for |x| > 2^53
template <typename WORD>
__device__ __forceinline__ double word_to_f64(WORD x) { return (double)(typename WordTraits<WORD>::S)x; }

// arithmetic.jl:1-9 native(x, mask)
template <typename WORD> __device__ __forceinline__ WORD native(double x);
template <> __device__ __forceinline__ uint32_t native<uint32_t>(double x) {
    x -= floor(x * 2.3283064365386963e-10) * 4.294967296e9;
    return x == 4.294967296e9 ? 0u : (uint32_t)x;
}
template <> __device__ __forceinline__ uint64_t native<uint64_t>(double x) {
    x -= floor(x * 5.421010862427522e-20) * 1.8446744073709552e19;
    return x == 1.8446744073709552e19 ? (uint64_t)0 : (uint64_t)x;
}

// arithmetic.jl:23-27 divbits (bit may be 0: Julia shifts by >= width give 0)
template <typename WORD>
__device__ __forceinline__ WORD divbits(WORD a, int bit) {
    constexpr int W = WordTraits<WORD>::W;
    if (bit <= 0) return a;
    const WORD carry = (WORD)(a << (W - bit)) >> (W - 1);
    return (WORD)((a >> bit) + carry);
}

// Balanced gadget decomposition (gsw.jl:42-52 / :86-96, unienc.jl:4-18), closed form:
//   t' = divbits(x, W - l*logB) + sum_j (B/2) * B^j ;  digit_j = ((t' >> logB*(l-1-j)) & (B-1)) - B/2
// which reproduces the reference's carry chain digit for digit (j = 0 most significant).
template <typename WORD>
struct Gadget {
    int l, logB;
    WORD offset, mask, half;
    int bit;
    __host__ __device__ Gadget() {}
    __host__ __device__ Gadget(int l_, int logB_) : l(l_), logB(logB_) {
        constexpr int W = WordTraits<WORD>::W;
        mask = (WORD)(((WORD)1 << logB) - 1);
        half = (WORD)((WORD)1 << (logB - 1));
        offset = 0;
        for (int j = 0; j < l; j++) offset = (WORD)(offset + (WORD)(half << (logB * j)));
        bit = W - l * logB;
    }
    __device__ __forceinline__ WORD prep(WORD x) const { return (WORD)(divbits<WORD>(x, bit) + offset); }
    // signed digit j of a prepared word, as a (small) int
    __device__ __forceinline__ int digit(WORD tp, int j) const {
        return (int)((tp >> (logB * (l - 1 - j))) & mask) - (int)half;
    }
};

}  // namespace mktd
