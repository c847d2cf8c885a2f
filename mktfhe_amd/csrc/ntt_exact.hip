// MKT_ARITH_EXACT: the negacyclic number-theoretic transform over Z_P[X]/(X^N + 1) in residue form, P = p1 * p2 with the
// two 31-bit NTT primes p1 = 15 * 2^27 + 1 and p2 = 63 * 2^25 + 1 (P = 2^61.88), batched HBM -> HBM; the exact negacyclic
// product of a gadget-digit polynomial with a ring polynomial mod 2^W built on it (the operation the reference's Float64
// transform approximates: src/ring/fft.jl:57-81 + polynomial.jl:99-113; the MultiFloat option of README.md:9 aims at the
// same exact value); and the CGGI blind rotation with exact products.
//
// Why two 31-bit primes and not one 64-bit prime: gfx950 multiplies 32 x 32 bits per lane and instruction
// (v_mul_lo_u32 / v_mul_hi_u32, 4.4 cycles per wave each); a product mod p = 2^64 - 2^32 + 1 costs ~33 instructions, a
// Shoup product mod a 31-bit prime 6, so a butterfly over both residues is ~2.1x cheaper than the Goldilocks one
// (tools/ntt_probe.hip) and the batched transform is bound by HBM, not by integer issue.  A point is the pair
// (x mod p1, x mod p2) packed into 64 bits -- the same 8 N bytes per polynomial as M complex doubles.
//
// Same butterfly network as the Float64 transform (fft_device.h), so the same pass / window / staging machinery: stage
// with stride 2^b multiplies by psi_rev[m + i] (Cooley-Tukey, bit-reversed output), the inverse runs Gentleman-Sande
// with the inverse table and a final N^-1.  8 points per thread, N / 8 threads per polynomial, passes of 3 stages local
// to a thread, LDS exchanges between passes.  Twiddles carry their Shoup companions floor(w * 2^32 / p); resident tables
// (keys, monomials) are stored in Montgomery form (x * 2^32 mod p) so that data x table products need no companion.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_api.h"
#include "fft_device.h"

namespace mktd {

namespace {

constexpr uint32_t P1 = 2013265921u, P2 = 2113929217u;      // 15 * 2^27 + 1, 63 * 2^25 + 1: both = 1 mod 2^13
constexpr uint64_t PP = (uint64_t)P1 * P2;
constexpr int NLR = 3;   // points per thread = 8

// host-evaluable helpers for the compile-time constants
constexpr uint32_t inv_mod_2_32(uint32_t p) { uint32_t x = p; for (int i = 0; i < 5; i++) x *= 2u - p * x; return x; }   // p^-1 mod 2^32 (Newton)
constexpr uint32_t PI1 = inv_mod_2_32(P1), PI2 = inv_mod_2_32(P2);
constexpr uint64_t powmod_c(uint64_t a, uint64_t e, uint64_t p) { uint64_t r = 1; a %= p; while (e) { if (e & 1) r = r * a % p; a = a * a % p; e >>= 1; } return r; }
constexpr uint32_t shoup_c(uint32_t w, uint32_t p) { return (uint32_t)(((uint64_t)w << 32) / p); }
constexpr uint32_t R1 = (uint32_t)(((uint64_t)1 << 32) % P1), R2 = (uint32_t)(((uint64_t)1 << 32) % P2);          // 2^32 mod p
constexpr uint32_t RR1 = (uint32_t)((uint64_t)R1 * R1 % P1), RR2 = (uint32_t)((uint64_t)R2 * R2 % P2);            // 2^64 mod p
constexpr uint32_t CRT_C = (uint32_t)powmod_c(P1 % P2, P2 - 2, P2), CRT_CS = shoup_c(CRT_C, P2);                   // p1^-1 mod p2

struct Pt { uint32_t a, b; };                                    // residues mod p1, p2
__device__ __forceinline__ uint64_t pack(Pt x) { return (uint64_t)x.a | ((uint64_t)x.b << 32); }
__device__ __forceinline__ Pt unpack(uint64_t v) { Pt x; x.a = (uint32_t)v; x.b = (uint32_t)(v >> 32); return x; }

template <uint32_t P> __device__ __forceinline__ uint32_t red1(uint32_t v) { const uint32_t w = v - P; return w < v ? w : v; }   // v < 2P -> [0, P)
template <uint32_t P> __device__ __forceinline__ uint32_t addm(uint32_t x, uint32_t y) { return red1<P>(x + y); }               // 2P < 2^32
template <uint32_t P> __device__ __forceinline__ uint32_t subm(uint32_t x, uint32_t y) { const uint32_t d = x - y, e = d + P; return e < d ? e : d; }
// x * w mod P for a constant w with companion ws = floor(w * 2^32 / P); any 32-bit x
template <uint32_t P> __device__ __forceinline__ uint32_t shoup(uint32_t x, uint32_t w, uint32_t ws) {
    const uint32_t q = __umulhi(x, ws);
    return red1<P>(x * w - q * P);
}
// Montgomery product x * y * 2^-32 mod P (x, y < P)
template <uint32_t P, uint32_t PINV> __device__ __forceinline__ uint32_t montmul(uint32_t x, uint32_t y) {
    const uint32_t lo = x * y, hi = __umulhi(x, y);
    const uint32_t m = lo * PINV, u = __umulhi(m, P);
    const uint32_t d = hi - u, e = d + P;
    return e < d ? e : d;                                        // hi - u in (-P, P)
}
__device__ __forceinline__ Pt pt_add(Pt x, Pt y) { Pt r; r.a = addm<P1>(x.a, y.a); r.b = addm<P2>(x.b, y.b); return r; }
__device__ __forceinline__ Pt pt_sub(Pt x, Pt y) { Pt r; r.a = subm<P1>(x.a, y.a); r.b = subm<P2>(x.b, y.b); return r; }
__device__ __forceinline__ Pt pt_shoup(Pt x, uint4 w) { Pt r; r.a = shoup<P1>(x.a, w.x, w.y); r.b = shoup<P2>(x.b, w.z, w.w); return r; }
__device__ __forceinline__ Pt pt_mont(Pt x, Pt y) { Pt r; r.a = montmul<P1, PI1>(x.a, y.a); r.b = montmul<P2, PI2>(x.b, y.b); return r; }

extern __shared__ __attribute__((aligned(16))) unsigned char ntt_smem[];

template <int LOGN>
__device__ __forceinline__ void ntt_exchange(Pt (&z)[8], uint64_t *lds, int t, int lo_from, int lo_to) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; e++) lds[lds_pos<NLR>(pt_index<NLR>(t, e, lo_from))] = pack(z[e]);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; e++) z[e] = unpack(lds[lds_pos<NLR>(pt_index<NLR>(t, e, lo_to))]);
}

// In: slot e = point e*NT + t.  Out: slot e = point 8t + e (bit-reversed order of the transform, as the reference's).
// psi[k] = (w mod p1, its companion, w mod p2, its companion), w = psi^bitrev(k)
template <int LOGN, int PASS = 0>
__device__ __forceinline__ void ntt_forward(Pt (&z)[8], const uint4 *__restrict__ psi, uint64_t *lds, int t) {
    using P = Plan<LOGN, NLR>;
    constexpr int p = PASS, lo = P::lo(p);
#pragma unroll
    for (int s = 0; s < P::nst(p); s++) {
        const int b = P::hib(p) - s, sb = b - lo;
        const int twbase = (1 << (LOGN - 1 - b)) + ((t >> lo) << (NLR - 1 - sb));
#pragma unroll
        for (int g = 0; g < (1 << (NLR - 1 - sb)); g++) {
            const uint4 w = psi[twbase + g];
#pragma unroll
            for (int q = 0; q < (1 << sb); q++) {
                const int e = (g << (sb + 1)) | q, e2 = e | (1 << sb);
                const Pt u = pt_shoup(z[e2], w), a = z[e];
                z[e] = pt_add(a, u); z[e2] = pt_sub(a, u);
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
__device__ __forceinline__ void ntt_inverse(Pt (&z)[8], const uint4 *__restrict__ psiinv, uint64_t *lds, int t) {
    using P = Plan<LOGN, NLR>;
    constexpr int p = PASS, lo = P::lo(p);
#pragma unroll
    for (int s = P::nst(p) - 1; s >= 0; s--) {
        const int b = P::hib(p) - s, sb = b - lo;
        const int twbase = (1 << (LOGN - 1 - b)) + ((t >> lo) << (NLR - 1 - sb));
#pragma unroll
        for (int g = 0; g < (1 << (NLR - 1 - sb)); g++) {
            const uint4 w = psiinv[twbase + g];
#pragma unroll
            for (int q = 0; q < (1 << sb); q++) {
                const int e = (g << (sb + 1)) | q, e2 = e | (1 << sb);
                const Pt a = z[e], u = z[e2];
                z[e] = pt_add(a, u); z[e2] = pt_shoup(pt_sub(a, u), w);
            }
        }
    }
    if constexpr (p > 0) {
        ntt_exchange<LOGN>(z, lds, t, P::lo(p), P::lo(p - 1));
        ntt_inverse<LOGN, PASS - 1>(z, psiinv, lds, t);
    }
}

// any 32-bit value -> [0, P)
template <uint32_t P> __device__ __forceinline__ uint32_t red_u32(uint32_t v) { return red1<P>(red1<P>(v)); }   // v < 2^32 < 3P
// signed 32-bit integer -> residue
template <uint32_t P> __device__ __forceinline__ uint32_t res_s32(int32_t s) { return red1<P>(s < 0 ? (uint32_t)s + 2u * P : (uint32_t)s); }   // s + 2P in (0, 2P)
__device__ __forceinline__ Pt res_small(int d) { Pt r; r.a = res_s32<P1>(d); r.b = res_s32<P2>(d); return r; }
// signed W-bit ring word -> residues (64-bit words: hi * 2^32 + lo with hi signed)
template <typename WORD> __device__ __forceinline__ Pt to_residue(WORD x);
template <> __device__ __forceinline__ Pt to_residue<uint32_t>(uint32_t x) { return res_small((int32_t)x); }
template <> __device__ __forceinline__ Pt to_residue<uint64_t>(uint64_t x) {
    const int32_t hi = (int32_t)(x >> 32); const uint32_t lo = (uint32_t)x;
    Pt r;
    r.a = addm<P1>(shoup<P1>(res_s32<P1>(hi), R1, shoup_c(R1, P1)), red_u32<P1>(lo));
    r.b = addm<P2>(shoup<P2>(res_s32<P2>(hi), R2, shoup_c(R2, P2)), red_u32<P2>(lo));
    return r;
}
// residues -> the integer of least magnitude they stand for (Garner), two's complement in 64 bits
__device__ __forceinline__ uint64_t crt_signed(Pt r) {
    const uint32_t a2 = red1<P2>(r.a);                           // r.a < p1 < 2 p2
    const uint32_t tq = shoup<P2>(subm<P2>(r.b, a2), CRT_C, CRT_CS);
    const uint64_t x = (uint64_t)r.a + (uint64_t)P1 * tq;        // in [0, P)
    return x > (PP >> 1) ? x - PP : x;
}

// constants at the tail of the table: N^-1 (+ companions) and N^-1 * 2^32 (for products of two plain operands taken with montmul)
struct NttConsts { uint4 ninv, ninv_r; };
// tables (32-bit words): psi_rev[N] x uint4 | psiinv_rev[N] x uint4 | NttConsts
template <int LOGN> __device__ __forceinline__ const uint4 *tab_psi(const uint4 *tab) { return tab; }
template <int LOGN> __device__ __forceinline__ const uint4 *tab_psiinv(const uint4 *tab) { return tab + (1 << LOGN); }
template <int LOGN> __device__ __forceinline__ NttConsts tab_consts(const uint4 *tab) { NttConsts c; c.ninv = tab[2 << LOGN]; c.ninv_r = tab[(2 << LOGN) + 1]; return c; }

// twiddles resident in LDS up to N = 2048 (one table: 16 N bytes); above, they are read through the caches
template <int LOGN> struct TwLds { static constexpr bool on = LOGN <= 11; };
template <int LOGN, int NTAB>
__device__ __forceinline__ void stage_tables(const uint4 *tab, uint4 *dst, int t, int NT, const uint4 *(&out)[NTAB], const int (&which)[NTAB]) {
    constexpr int N = 1 << LOGN;
#pragma unroll
    for (int k = 0; k < NTAB; k++) {
        const uint4 *src = tab + (size_t)which[k] * N;
        if (TwLds<LOGN>::on) { for (int i = t; i < N; i += NT) dst[k * N + i] = src[i]; out[k] = dst + k * N; }
        else out[k] = src;
    }
    __syncthreads();
}
template <int LOGN> constexpr size_t lds_bytes(int ntab, int ppw = 1) { return (size_t)ppw * (1 << LOGN) * 8 + (TwLds<LOGN>::on ? (size_t)ntab * (1 << LOGN) * 16 : 0); }

// Batched transforms: PPW polynomials per workgroup side by side (N / 8 threads each) share the staged twiddle table, which
// lifts the number of resident waves per CU from 12 to 20 at N = 1024.
// MONT: the output goes to a resident table (keys, monomials): Montgomery form
#ifndef MKT_NTT_PPW
#define MKT_NTT_PPW 2
#endif
template <int LOGN> struct Ppw { static constexpr int v = (LOGN <= 10 && LOGN >= 6) ? MKT_NTT_PPW : 1; };
template <int LOGN, typename WORD>
__global__ __launch_bounds__((Ppw<LOGN>::v << (LOGN - NLR))) void ntt_fwd_kernel(const uint4 *__restrict__ tab, const WORD *__restrict__ p,
                                                                                 uint64_t *__restrict__ out, size_t B, int mont) {
    constexpr int N = 1 << LOGN, NT = N >> NLR, PPW = Ppw<LOGN>::v;
    const int sub = PPW > 1 ? threadIdx.x / NT : 0, t = PPW > 1 ? threadIdx.x % NT : threadIdx.x;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem) + (size_t)sub * N;
    const uint4 *tw[1]; const int which[1] = {0};
    stage_tables<LOGN, 1>(tab, reinterpret_cast<uint4 *>(reinterpret_cast<uint64_t *>(ntt_smem) + (size_t)PPW * N), threadIdx.x, PPW * NT, tw, which);
    const size_t groups = (B + PPW - 1) / PPW;
    for (size_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const size_t b0 = g * PPW + sub, b = b0 < B ? b0 : B - 1;       // a ragged last group repeats the last polynomial (same values, same address)
        Pt z[8];
#pragma unroll
        for (int e = 0; e < 8; e++) z[e] = to_residue<WORD>(__builtin_nontemporal_load(&p[b * N + e * NT + t]));
        ntt_forward<LOGN>(z, tw[0], lds, t);
        if (mont) {
#pragma unroll
            for (int e = 0; e < 8; e++) { z[e].a = montmul<P1, PI1>(z[e].a, RR1); z[e].b = montmul<P2, PI2>(z[e].b, RR2); }
        }
        ntt_exchange<LOGN>(z, lds, t, 0, Plan<LOGN, NLR>::lo(0));        // thread-contiguous stores: point e*NT + t of the output order
#pragma unroll
        for (int e = 0; e < 8; e++) __builtin_nontemporal_store(pack(z[e]), &out[b * N + e * NT + t]);
    }
}
template <int LOGN, typename WORD>
__global__ __launch_bounds__((Ppw<LOGN>::v << (LOGN - NLR))) void ntt_inv_kernel(const uint4 *__restrict__ tab, const uint64_t *__restrict__ in,
                                                                                 WORD *__restrict__ p, size_t B) {
    constexpr int N = 1 << LOGN, NT = N >> NLR, PPW = Ppw<LOGN>::v;
    const int sub = PPW > 1 ? threadIdx.x / NT : 0, t = PPW > 1 ? threadIdx.x % NT : threadIdx.x;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem) + (size_t)sub * N;
    const uint4 *tw[1]; const int which[1] = {1};
    stage_tables<LOGN, 1>(tab, reinterpret_cast<uint4 *>(reinterpret_cast<uint64_t *>(ntt_smem) + (size_t)PPW * N), threadIdx.x, PPW * NT, tw, which);
    const NttConsts k = tab_consts<LOGN>(tab);
    const size_t groups = (B + PPW - 1) / PPW;
    for (size_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const size_t b0 = g * PPW + sub, b = b0 < B ? b0 : B - 1;
        Pt z[8];
#pragma unroll
        for (int e = 0; e < 8; e++) z[e] = unpack(__builtin_nontemporal_load(&in[b * N + e * NT + t]));
        ntt_exchange<LOGN>(z, lds, t, Plan<LOGN, NLR>::lo(0), 0);
        ntt_inverse<LOGN, Plan<LOGN, NLR>::NPASS - 1>(z, tw[0], lds, t);
#pragma unroll
        for (int e = 0; e < 8; e++) __builtin_nontemporal_store((WORD)crt_signed(pt_shoup(z[e], k.ninv)), &p[b * N + e * NT + t]);
    }
}

// exact negacyclic product mod 2^W of a digit polynomial a (signed, small) and a ring polynomial b: the 32-bit halves of b
// go through separate transforms so that every true coefficient stays below P / 2 (N * max|a| * 2^32 < 2^60.8)
template <int LOGN, typename WORD>
__global__ __launch_bounds__((1 << (LOGN - NLR))) void exact_polymul_kernel(const uint4 *__restrict__ tab, const WORD *__restrict__ a,
                                                                            const WORD *__restrict__ bp, WORD *__restrict__ out, size_t B) {
    constexpr int N = 1 << LOGN, NT = N >> NLR, W = WordTraits<WORD>::W, H = W == 64 ? 2 : 1;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem);
    const int t = threadIdx.x;
    const uint4 *tw[2]; const int which[2] = {0, 1};
    stage_tables<LOGN, 2>(tab, reinterpret_cast<uint4 *>(lds + N), t, NT, tw, which);
    const NttConsts k = tab_consts<LOGN>(tab);
    for (size_t b = blockIdx.x; b < B; b += gridDim.x) {
        Pt za[8];
#pragma unroll
        for (int e = 0; e < 8; e++) za[e] = to_residue<WORD>(a[b * N + e * NT + t]);
        ntt_forward<LOGN>(za, tw[0], lds, t);
        WORD acc[8];
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = 0;
#pragma unroll
        for (int h = 0; h < H; h++) {
            Pt zb[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint64_t w = (uint64_t)bp[b * N + e * NT + t];
                const uint32_t piece = (uint32_t)(W == 64 ? (w >> (32 * h)) : w);      // unsigned 32-bit pieces
                zb[e].a = red_u32<P1>(piece); zb[e].b = red_u32<P2>(piece);
            }
            ntt_forward<LOGN>(zb, tw[0], lds, t);
#pragma unroll
            for (int e = 0; e < 8; e++) zb[e] = pt_mont(zb[e], za[e]);                 // x y 2^-32: undone by N^-1 2^32 below
            ntt_inverse<LOGN, Plan<LOGN, NLR>::NPASS - 1>(zb, tw[1], lds, t);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint64_t v = crt_signed(pt_shoup(zb[e], k.ninv_r));              // the exact integer, two's complement mod 2^64
                acc[e] = (WORD)(acc[e] + (WORD)(W == 64 ? v << (32 * h) : v));
            }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) out[b * N + e * NT + t] = acc[e];
    }
}


// ------------------------------------------------------------------------------------------------
// Blind rotation with EXACT products (CGGI and LMSS, RLWE length 1, 32-bit ring): bootstrapping.jl:32-76 / :114-165 with
// every transform-domain product replaced by the exact negacyclic product mod 2^32 -- digit transforms, row MACs
// (:63-68 / :146-154), monomial multiply (:71 / :157) and inverse (:72 / :162) all over Z_P, one exact lift per CMux step
// (per block of LB key bits for LMSS: one decomposition, LB accumulators, tacc2 = sum of monomial * tacc, :131-163).  True
// coefficients stay below 2 * LB * 2l * N * 2^(logB-1) * 2^31 < P / 2 (checked on the host for the context's gadget).
// One workgroup of N / 8 threads per rotation; the accumulator lives in registers (slot e = coefficient e*NT + t).  Key
// and monomial tables are in the transform's natural order, Montgomery form.
// ------------------------------------------------------------------------------------------------
template <int LOGN, int LB>
__global__ __launch_bounds__((1 << (LOGN - NLR))) void exact_blindrotate_kernel(const uint4 *__restrict__ tab, const uint64_t *__restrict__ brk,
                                                                              const uint64_t *__restrict__ mono, const uint32_t *__restrict__ lwe,
                                                                              int lwe_stride, int pre_switched, int n, int l, int logB, uint32_t *__restrict__ acc_io) {
    constexpr int N = 1 << LOGN, NT = N >> NLR;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem);
    const int t = threadIdx.x;
    const uint4 *tw[2]; const int which[2] = {0, 1};
    stage_tables<LOGN, 2>(tab, reinterpret_cast<uint4 *>(lds + N), t, NT, tw, which);
    const NttConsts k = tab_consts<LOGN>(tab);
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
    for (int blk = 0; blk < n / LB; blk++) {
        uint32_t ats[LB];
        bool any = false;
#pragma unroll
        for (int q = 0; q < LB; q++) {
            const uint32_t v0 = at_src[blk * LB + q];
            ats[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pre_switched ? v0 : divbits<uint32_t>(v0, msbit)));
            any |= ats[q] != 0;
        }
        if (!any) continue;                                              // :48 / :145 (an all-zero block adds 0)
        Pt tacc[LB][2][8];
#pragma unroll
        for (int q = 0; q < LB; q++)
#pragma unroll
            for (int pp = 0; pp < 2; pp++)
#pragma unroll
                for (int e = 0; e < 8; e++) { tacc[q][pp][e].a = 0; tacc[q][pp][e].b = 0; }
        for (int c = 0; c < 2; c++) {
            uint32_t tp[8];
#pragma unroll
            for (int e = 0; e < 8; e++) tp[e] = gd.prep(c ? acc[1][e] : acc[0][e]);      // :50-51 / :131-132 decompto!
            for (int j = 0; j < l; j++) {
                Pt z[8];
#pragma unroll
                for (int e = 0; e < 8; e++) z[e] = res_small(gd.digit(tp[e], j));
                ntt_forward<LOGN>(z, tw[0], lds, t);
#pragma unroll
                for (int q = 0; q < LB; q++) {
                    if (ats[q] == 0) continue;
                    const uint64_t *row = brk + (((size_t)(blk * LB + q) * 2 * l + (size_t)(c * l + j)) * 2) * N + 8 * t;
#pragma unroll
                    for (int e = 0; e < 8; e++) {                        // :63-68 / :146-154, exactly
                        tacc[q][0][e] = pt_add(tacc[q][0][e], pt_mont(z[e], unpack(row[e])));
                        tacc[q][1][e] = pt_add(tacc[q][1][e], pt_mont(z[e], unpack(row[N + e])));
                    }
                }
            }
        }
#pragma unroll
        for (int pp = 0; pp < 2; pp++) {
            Pt s2[8];
#pragma unroll
            for (int e = 0; e < 8; e++) { s2[e].a = 0; s2[e].b = 0; }
#pragma unroll
            for (int q = 0; q < LB; q++) {
                if (ats[q] == 0) continue;
                const uint64_t *mrow = mono + (size_t)(ats[q] - 1) * N + 8 * t;
#pragma unroll
                for (int e = 0; e < 8; e++) s2[e] = pt_add(s2[e], pt_mont(tacc[q][pp][e], unpack(mrow[e])));   // :71 / :157
            }
            ntt_inverse<LOGN, Plan<LOGN, NLR>::NPASS - 1>(s2, tw[1], lds, t);            // :72 / :162
#pragma unroll
            for (int e = 0; e < 8; e++) acc[pp][e] += (uint32_t)crt_signed(pt_shoup(s2[e], k.ninv));   // :73 / :163
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

hipError_t launch_ntt_fwd(int logN, int W, const uint64_t *tab, const void *p, uint64_t *t, size_t B, int montgomery, hipStream_t s) {
    if (!B) return hipSuccess;
    const uint4 *tb = reinterpret_cast<const uint4 *>(tab);
    MKT_NTT_DISPATCH(logN, {
        constexpr int PPW = Ppw<LN>::v; const size_t lds = lds_bytes<LN>(1, PPW); const size_t groups = (B + PPW - 1) / PPW; const int grid = (int)(groups < 32768 ? groups : 32768);
        if (W == 64) { hipError_t e = ntt_set_lds(ntt_fwd_kernel<LN, uint64_t>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((ntt_fwd_kernel<LN, uint64_t>), dim3(grid), dim3(PPW << (LN - NLR)), lds, s, tb, (const uint64_t *)p, t, B, montgomery); }
        else { hipError_t e = ntt_set_lds(ntt_fwd_kernel<LN, uint32_t>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((ntt_fwd_kernel<LN, uint32_t>), dim3(grid), dim3(PPW << (LN - NLR)), lds, s, tb, (const uint32_t *)p, t, B, montgomery); }
    });
    return hipGetLastError();
}
hipError_t launch_ntt_inv(int logN, int W, const uint64_t *tab, const uint64_t *t, void *p, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    const uint4 *tb = reinterpret_cast<const uint4 *>(tab);
    MKT_NTT_DISPATCH(logN, {
        constexpr int PPW = Ppw<LN>::v; const size_t lds = lds_bytes<LN>(1, PPW); const size_t groups = (B + PPW - 1) / PPW; const int grid = (int)(groups < 32768 ? groups : 32768);
        if (W == 64) { hipError_t e = ntt_set_lds(ntt_inv_kernel<LN, uint64_t>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((ntt_inv_kernel<LN, uint64_t>), dim3(grid), dim3(PPW << (LN - NLR)), lds, s, tb, t, (uint64_t *)p, B); }
        else { hipError_t e = ntt_set_lds(ntt_inv_kernel<LN, uint32_t>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((ntt_inv_kernel<LN, uint32_t>), dim3(grid), dim3(PPW << (LN - NLR)), lds, s, tb, t, (uint32_t *)p, B); }
    });
    return hipGetLastError();
}
hipError_t launch_exact_polymul(int logN, int W, const uint64_t *tab, const void *a, const void *b, void *out, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    const int grid = (int)(B < 32768 ? B : 32768);
    const uint4 *tb = reinterpret_cast<const uint4 *>(tab);
    MKT_NTT_DISPATCH(logN, {
        const size_t lds = lds_bytes<LN>(2);
        if (W == 64) { hipError_t e = ntt_set_lds(exact_polymul_kernel<LN, uint64_t>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((exact_polymul_kernel<LN, uint64_t>), dim3(grid), dim3(1 << (LN - NLR)), lds, s, tb, (const uint64_t *)a, (const uint64_t *)b, (uint64_t *)out, B); }
        else { hipError_t e = ntt_set_lds(exact_polymul_kernel<LN, uint32_t>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((exact_polymul_kernel<LN, uint32_t>), dim3(grid), dim3(1 << (LN - NLR)), lds, s, tb, (const uint32_t *)a, (const uint32_t *)b, (uint32_t *)out, B); }
    });
    return hipGetLastError();
}

hipError_t launch_exact_blindrotate(int logN, const uint64_t *tab, const uint64_t *brk, const uint64_t *mono, const uint32_t *lwe, int lwe_stride,
                                    int pre_switched, int n, int l, int logB, int blk_len, uint32_t *acc, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    if (blk_len != 1 && blk_len != 3) return hipErrorInvalidValue;
    const uint4 *tb = reinterpret_cast<const uint4 *>(tab);
    MKT_NTT_DISPATCH(logN, {
        const size_t lds = lds_bytes<LN>(2);
        if (blk_len == 1) {
            hipError_t e = ntt_set_lds(exact_blindrotate_kernel<LN, 1>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((exact_blindrotate_kernel<LN, 1>), dim3((unsigned)B), dim3(1 << (LN - NLR)), lds, s, tb, brk, mono, lwe, lwe_stride, pre_switched, n, l, logB, acc);
        } else {
            hipError_t e = ntt_set_lds(exact_blindrotate_kernel<LN, 3>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((exact_blindrotate_kernel<LN, 3>), dim3((unsigned)B), dim3(1 << (LN - NLR)), lds, s, tb, brk, mono, lwe, lwe_stride, pre_switched, n, l, logB, acc);
        }
    });
    return hipGetLastError();
}

}  // namespace mktd
