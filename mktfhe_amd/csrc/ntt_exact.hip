// MKT_ARITH_EXACT at the transform level: the negacyclic number-theoretic transform over Z_p[X]/(X^N + 1), p the
// Goldilocks prime 2^64 - 2^32 + 1, batched HBM -> HBM, and the exact negacyclic product of a gadget-digit polynomial
// with a ring polynomial mod 2^W built on it (the operation the reference's Float64 transform approximates:
// src/ring/fft.jl:57-81 + polynomial.jl:99-113; the MultiFloat option of README.md:9 aims at the same exact value).
//
// Same butterfly network as the Float64 transform (fft_device.h), so the same pass / window / staging machinery: stage
// with stride 2^b multiplies by psi_rev[m + i] (Cooley-Tukey, bit-reversed output), the inverse runs Gentleman-Sande
// with the inverse table and a final N^-1.  8 points per thread, N / 8 threads per polynomial, passes of 3 stages local
// to a thread, LDS exchanges between passes.
//
// The gate path does not use this mode (~15x the cost of the Float64 path on gfx950, and its ciphertexts
// are not the reference's bits); it is the exact yardstick and the unit-level transform of the mode.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_api.h"
#include "fft_device.h"

namespace mktd {

namespace {

constexpr uint64_t GL_P = 0xFFFFFFFF00000001ull, GL_EPS = 0xFFFFFFFFull;
constexpr int NLR = 3;   // points per thread = 8

__device__ __forceinline__ uint64_t gl_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    if (s < a) s += GL_EPS;                  // wrapped past 2^64 = EPS mod p
    return s >= GL_P ? s - GL_P : s;
}
__device__ __forceinline__ uint64_t gl_sub(uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    if (a < b) d -= GL_EPS;                  // borrowed 2^64 = EPS mod p
    return d;
}
__device__ __forceinline__ uint64_t gl_mul(uint64_t a, uint64_t b) {
    const uint64_t lo = a * b, hi = __umul64hi(a, b);
    const uint64_t hh = hi >> 32, hl = hi & GL_EPS;      // 2^96 = -1, 2^64 = 2^32 - 1 (mod p)
    uint64_t t0 = lo - hh;
    if (lo < hh) t0 -= GL_EPS;
    const uint64_t t1 = hl * GL_EPS;
    uint64_t r = t0 + t1;
    if (r < t0) r += GL_EPS;
    return r >= GL_P ? r - GL_P : r;
}

extern __shared__ __attribute__((aligned(16))) unsigned char ntt_smem[];

template <int LOGN>
__device__ __forceinline__ void ntt_exchange(uint64_t (&z)[8], uint64_t *lds, int t, int lo_from, int lo_to) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; e++) lds[lds_pos<NLR>(pt_index<NLR>(t, e, lo_from))] = z[e];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; e++) z[e] = lds[lds_pos<NLR>(pt_index<NLR>(t, e, lo_to))];
}

// In: slot e = point e*NT + t.  Out: slot e = point 8t + e (bit-reversed order of the transform, as the reference's)
template <int LOGN, int PASS = 0>
__device__ __forceinline__ void ntt_forward(uint64_t (&z)[8], const uint64_t *__restrict__ psi, uint64_t *lds, int t) {
    using P = Plan<LOGN, NLR>;
    constexpr int p = PASS, lo = P::lo(p);
#pragma unroll
    for (int s = 0; s < P::nst(p); s++) {
        const int b = P::hib(p) - s, sb = b - lo;
        const int twbase = (1 << (LOGN - 1 - b)) + ((t >> lo) << (NLR - 1 - sb));
#pragma unroll
        for (int g = 0; g < (1 << (NLR - 1 - sb)); g++) {
            const uint64_t w = psi[twbase + g];
#pragma unroll
            for (int q = 0; q < (1 << sb); q++) {
                const int e = (g << (sb + 1)) | q, e2 = e | (1 << sb);
                const uint64_t u = gl_mul(z[e2], w), a = z[e];
                z[e] = gl_add(a, u); z[e2] = gl_sub(a, u);
            }
        }
    }
    if constexpr (p < P::NPASS - 1) {
        ntt_exchange<LOGN>(z, lds, t, P::lo(p), P::lo(p + 1));
        ntt_forward<LOGN, PASS + 1>(z, psi, lds, t);
    }
}
// In: slot e = point 8t + e.  Out: slot e = point e*NT + t, NOT yet scaled by N^-1
template <int LOGN, int PASS>
__device__ __forceinline__ void ntt_inverse(uint64_t (&z)[8], const uint64_t *__restrict__ psiinv, uint64_t *lds, int t) {
    using P = Plan<LOGN, NLR>;
    constexpr int p = PASS, lo = P::lo(p);
#pragma unroll
    for (int s = P::nst(p) - 1; s >= 0; s--) {
        const int b = P::hib(p) - s, sb = b - lo;
        const int twbase = (1 << (LOGN - 1 - b)) + ((t >> lo) << (NLR - 1 - sb));
#pragma unroll
        for (int g = 0; g < (1 << (NLR - 1 - sb)); g++) {
            const uint64_t w = psiinv[twbase + g];
#pragma unroll
            for (int q = 0; q < (1 << sb); q++) {
                const int e = (g << (sb + 1)) | q, e2 = e | (1 << sb);
                const uint64_t a = z[e], u = z[e2];
                z[e] = gl_add(a, u); z[e2] = gl_mul(gl_sub(a, u), w);
            }
        }
    }
    if constexpr (p > 0) {
        ntt_exchange<LOGN>(z, lds, t, P::lo(p), P::lo(p - 1));
        ntt_inverse<LOGN, PASS - 1>(z, psiinv, lds, t);
    }
}

// signed W-bit ring word -> residue mod p
template <typename WORD> __device__ __forceinline__ uint64_t to_residue(WORD x) {
    const int64_t s = (int64_t)(typename WordTraits<WORD>::S)x;
    return s >= 0 ? (uint64_t)s : GL_P - ((uint64_t)0 - (uint64_t)s);      // |s| <= 2^63 < p
}
// residue -> the integer of least magnitude it stands for, as a W-bit ring word
template <typename WORD> __device__ __forceinline__ WORD from_residue(uint64_t r) {
    const uint64_t half = GL_P >> 1;
    return r > half ? (WORD)((uint64_t)0 - (GL_P - r)) : (WORD)r;
}

// tables: psi_rev[N] | psiinv_rev[N] | ninv (1 word)
template <int LOGN, typename WORD>
__global__ __launch_bounds__((1 << (LOGN - NLR))) void ntt_fwd_kernel(const uint64_t *__restrict__ tab, const WORD *__restrict__ p,
                                                                      uint64_t *__restrict__ out, size_t B) {
    constexpr int N = 1 << LOGN, NT = N >> NLR;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem);
    uint64_t *psi_l = lds + N;
    const int t = threadIdx.x;
    for (int i = t; i < N; i += NT) psi_l[i] = tab[i];
    __syncthreads();
    for (size_t b = blockIdx.x; b < B; b += gridDim.x) {
        uint64_t z[8];
#pragma unroll
        for (int e = 0; e < 8; e++) z[e] = to_residue<WORD>(__builtin_nontemporal_load(&p[b * N + e * NT + t]));
        ntt_forward<LOGN>(z, psi_l, lds, t);
        ntt_exchange<LOGN>(z, lds, t, 0, Plan<LOGN, NLR>::lo(0));        // thread-contiguous stores: point e*NT + t of the output order
#pragma unroll
        for (int e = 0; e < 8; e++) __builtin_nontemporal_store(z[e], &out[b * N + e * NT + t]);
    }
}
template <int LOGN, typename WORD>
__global__ __launch_bounds__((1 << (LOGN - NLR))) void ntt_inv_kernel(const uint64_t *__restrict__ tab, const uint64_t *__restrict__ in,
                                                                      WORD *__restrict__ p, size_t B) {
    constexpr int N = 1 << LOGN, NT = N >> NLR;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem);
    uint64_t *psi_l = lds + N;
    const int t = threadIdx.x;
    for (int i = t; i < N; i += NT) psi_l[i] = tab[N + i];
    __syncthreads();
    const uint64_t ninv = tab[2 * N];
    for (size_t b = blockIdx.x; b < B; b += gridDim.x) {
        uint64_t z[8];
#pragma unroll
        for (int e = 0; e < 8; e++) z[e] = __builtin_nontemporal_load(&in[b * N + e * NT + t]);
        ntt_exchange<LOGN>(z, lds, t, Plan<LOGN, NLR>::lo(0), 0);
        ntt_inverse<LOGN, Plan<LOGN, NLR>::NPASS - 1>(z, psi_l, lds, t);
#pragma unroll
        for (int e = 0; e < 8; e++) __builtin_nontemporal_store(from_residue<WORD>(gl_mul(z[e], ninv)), &p[b * N + e * NT + t]);
    }
}

// exact negacyclic product mod 2^W of a digit polynomial a (signed, small) and a ring polynomial b: the 32-bit halves of b
// go through separate transforms so that every true coefficient stays below p / 2 (N * max|a| * 2^32 < 2^63)
template <int LOGN, typename WORD>
__global__ __launch_bounds__((1 << (LOGN - NLR))) void exact_polymul_kernel(const uint64_t *__restrict__ tab, const WORD *__restrict__ a,
                                                                            const WORD *__restrict__ bp, WORD *__restrict__ out, size_t B) {
    constexpr int N = 1 << LOGN, NT = N >> NLR, W = WordTraits<WORD>::W, H = W == 64 ? 2 : 1;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem);
    uint64_t *psi_l = lds + N, *psii_l = psi_l + N;
    const int t = threadIdx.x;
    for (int i = t; i < N; i += NT) { psi_l[i] = tab[i]; psii_l[i] = tab[N + i]; }
    __syncthreads();
    const uint64_t ninv = tab[2 * N];
    for (size_t b = blockIdx.x; b < B; b += gridDim.x) {
        uint64_t za[8];
#pragma unroll
        for (int e = 0; e < 8; e++) za[e] = to_residue<WORD>(a[b * N + e * NT + t]);
        ntt_forward<LOGN>(za, psi_l, lds, t);
        WORD acc[8];
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = 0;
#pragma unroll
        for (int h = 0; h < H; h++) {
            uint64_t zb[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint64_t w = (uint64_t)bp[b * N + e * NT + t];
                zb[e] = W == 64 ? ((w >> (32 * h)) & GL_EPS) : w;          // unsigned 32-bit pieces
            }
            ntt_forward<LOGN>(zb, psi_l, lds, t);
#pragma unroll
            for (int e = 0; e < 8; e++) zb[e] = gl_mul(zb[e], za[e]);
            ntt_inverse<LOGN, Plan<LOGN, NLR>::NPASS - 1>(zb, psii_l, lds, t);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint64_t r = gl_mul(zb[e], ninv), half = GL_P >> 1;
                const uint64_t v = r > half ? (uint64_t)0 - (GL_P - r) : r;      // the exact integer, two's complement mod 2^64
                acc[e] = (WORD)(acc[e] + (WORD)(W == 64 ? v << (32 * h) : v));
            }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) out[b * N + e * NT + t] = acc[e];
    }
}


// ------------------------------------------------------------------------------------------------
// Blind rotation with EXACT products (CGGI, RLWE length 1, 32-bit ring): bootstrapping.jl:32-76 with every
// transform-domain product replaced by the exact negacyclic product mod 2^32 -- digit transforms, row MACs (:63-68),
// monomial multiply (:71) and inverse (:72) all over Z_p, one exact lift per CMux step.  True coefficients stay below
// 2 * 2l * N * 2^(logB-1) * 2^31 < p / 2 for every shipped gadget.  One workgroup of N / 8 threads per rotation; the
// accumulator lives in registers (slot e = coefficient e*NT + t).  Tables are in the transform's natural order.
// ------------------------------------------------------------------------------------------------
template <int LOGN>
__global__ __launch_bounds__((1 << (LOGN - NLR))) void exact_blindrotate_kernel(const uint64_t *__restrict__ tab, const uint64_t *__restrict__ brk,
                                                                              const uint64_t *__restrict__ mono, const uint32_t *__restrict__ lwe,
                                                                              int lwe_stride, int pre_switched, int n, int l, int logB, uint32_t *__restrict__ acc_io) {
    constexpr int N = 1 << LOGN, NT = N >> NLR;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem);
    uint64_t *psi_l = lds + N, *psii_l = psi_l + N;
    const int t = threadIdx.x;
    for (int i = t; i < N; i += NT) { psi_l[i] = tab[i]; psii_l[i] = tab[N + i]; }
    __syncthreads();
    const uint64_t ninv = tab[2 * N];
    const size_t rot = blockIdx.x;
    const uint32_t *at_src = lwe + rot * (size_t)lwe_stride;
    uint32_t *accg = acc_io + rot * 2 * (size_t)N;
    const Gadget<uint32_t> gd(l, logB);
    uint32_t acc[2][8];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int e = 0; e < 8; e++) acc[c][e] = accg[c * N + e * NT + t];
    const int msbit = 32 - LOGN - 1;
    for (int i = 0; i < n; i++) {
        const uint32_t v0 = at_src[i];
        const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pre_switched ? v0 : divbits<uint32_t>(v0, msbit)));
        if (at == 0) continue;                                           // :48
        uint64_t tacc[2][8];
#pragma unroll
        for (int pp = 0; pp < 2; pp++)
#pragma unroll
            for (int e = 0; e < 8; e++) tacc[pp][e] = 0;
        for (int c = 0; c < 2; c++) {
            uint32_t tp[8];
#pragma unroll
            for (int e = 0; e < 8; e++) tp[e] = gd.prep(c ? acc[1][e] : acc[0][e]);      // :50-51 decompto!
            for (int j = 0; j < l; j++) {
                uint64_t z[8];
#pragma unroll
                for (int e = 0; e < 8; e++) { const int d = gd.digit(tp[e], j); z[e] = d >= 0 ? (uint64_t)d : GL_P - (uint64_t)(-d); }
                ntt_forward<LOGN>(z, psi_l, lds, t);
                const uint64_t *row = brk + (((size_t)i * 2 * l + (size_t)(c * l + j)) * 2) * N + 8 * t;
#pragma unroll
                for (int e = 0; e < 8; e++) {                            // :63-68, exactly
                    tacc[0][e] = gl_add(tacc[0][e], gl_mul(z[e], row[e]));
                    tacc[1][e] = gl_add(tacc[1][e], gl_mul(z[e], row[N + e]));
                }
            }
        }
        const uint64_t *mrow = mono + (size_t)(at - 1) * N + 8 * t;
#pragma unroll
        for (int pp = 0; pp < 2; pp++) {
            uint64_t s2[8];
#pragma unroll
            for (int e = 0; e < 8; e++) s2[e] = gl_mul(tacc[pp][e], mrow[e]);   // :71
            ntt_inverse<LOGN, Plan<LOGN, NLR>::NPASS - 1>(s2, psii_l, lds, t);      // :72
#pragma unroll
            for (int e = 0; e < 8; e++) acc[pp][e] += from_residue<uint32_t>(gl_mul(s2[e], ninv));   // :73
        }
    }
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int e = 0; e < 8; e++) accg[c * N + e * NT + t] = acc[c][e];
}

template <typename K>
static hipError_t ntt_set_lds(K kern, size_t bytes) {
    if (bytes > 48 * 1024) return hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return hipSuccess;
}

}  // namespace

#define MKT_NTT_DISPATCH(logN, ...)                   \
    switch (logN) {                                   \
    case 5:  { constexpr int LN = 5;  __VA_ARGS__; } break; \
    case 6:  { constexpr int LN = 6;  __VA_ARGS__; } break; \
    case 7:  { constexpr int LN = 7;  __VA_ARGS__; } break; \
    case 8:  { constexpr int LN = 8;  __VA_ARGS__; } break; \
    case 9:  { constexpr int LN = 9;  __VA_ARGS__; } break; \
    case 10: { constexpr int LN = 10; __VA_ARGS__; } break; \
    case 11: { constexpr int LN = 11; __VA_ARGS__; } break; \
    case 12: { constexpr int LN = 12; __VA_ARGS__; } break; \
    default: return hipErrorInvalidValue;             \
    }

hipError_t launch_ntt_fwd(int logN, int W, const uint64_t *tab, const void *p, uint64_t *t, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    const int grid = (int)(B < 32768 ? B : 32768);
    MKT_NTT_DISPATCH(logN, {
        const size_t lds = (size_t)2 * (1 << LN) * 8;
        if (W == 64) { hipError_t e = ntt_set_lds(ntt_fwd_kernel<LN, uint64_t>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((ntt_fwd_kernel<LN, uint64_t>), dim3(grid), dim3(1 << (LN - NLR)), lds, s, tab, (const uint64_t *)p, t, B); }
        else { hipError_t e = ntt_set_lds(ntt_fwd_kernel<LN, uint32_t>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((ntt_fwd_kernel<LN, uint32_t>), dim3(grid), dim3(1 << (LN - NLR)), lds, s, tab, (const uint32_t *)p, t, B); }
    });
    return hipGetLastError();
}
hipError_t launch_ntt_inv(int logN, int W, const uint64_t *tab, const uint64_t *t, void *p, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    const int grid = (int)(B < 32768 ? B : 32768);
    MKT_NTT_DISPATCH(logN, {
        const size_t lds = (size_t)2 * (1 << LN) * 8;
        if (W == 64) { hipError_t e = ntt_set_lds(ntt_inv_kernel<LN, uint64_t>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((ntt_inv_kernel<LN, uint64_t>), dim3(grid), dim3(1 << (LN - NLR)), lds, s, tab, t, (uint64_t *)p, B); }
        else { hipError_t e = ntt_set_lds(ntt_inv_kernel<LN, uint32_t>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((ntt_inv_kernel<LN, uint32_t>), dim3(grid), dim3(1 << (LN - NLR)), lds, s, tab, t, (uint32_t *)p, B); }
    });
    return hipGetLastError();
}
hipError_t launch_exact_polymul(int logN, int W, const uint64_t *tab, const void *a, const void *b, void *out, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    const int grid = (int)(B < 32768 ? B : 32768);
    MKT_NTT_DISPATCH(logN, {
        const size_t lds = (size_t)3 * (1 << LN) * 8;
        if (W == 64) { hipError_t e = ntt_set_lds(exact_polymul_kernel<LN, uint64_t>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((exact_polymul_kernel<LN, uint64_t>), dim3(grid), dim3(1 << (LN - NLR)), lds, s, tab, (const uint64_t *)a, (const uint64_t *)b, (uint64_t *)out, B); }
        else { hipError_t e = ntt_set_lds(exact_polymul_kernel<LN, uint32_t>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((exact_polymul_kernel<LN, uint32_t>), dim3(grid), dim3(1 << (LN - NLR)), lds, s, tab, (const uint32_t *)a, (const uint32_t *)b, (uint32_t *)out, B); }
    });
    return hipGetLastError();
}

hipError_t launch_exact_blindrotate(int logN, const uint64_t *tab, const uint64_t *brk, const uint64_t *mono, const uint32_t *lwe, int lwe_stride,
                                    int pre_switched, int n, int l, int logB, uint32_t *acc, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    MKT_NTT_DISPATCH(logN, {
        const size_t lds = (size_t)3 * (1 << LN) * 8;
        hipError_t e = ntt_set_lds(exact_blindrotate_kernel<LN>, lds); if (e != hipSuccess) return e;
        hipLaunchKernelGGL((exact_blindrotate_kernel<LN>), dim3((unsigned)B), dim3(1 << (LN - NLR)), lds, s, tab, brk, mono, lwe, lwe_stride, pre_switched, n, l, logB, acc);
    });
    return hipGetLastError();
}

}  // namespace mktd
