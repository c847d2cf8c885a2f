// Evaluation-key generation on the device: the two large keys of a party -- the RGSW / UniEnc
// bootstrapping key and the LWE key-switching key -- sampled with exact integer arithmetic on the GPU from the party's
// secret keys.  Reference: keygen.jl:13-23, :39-51, :71-79, :106-114, :143-151; lwe.jl:11-22, :78-93; gsw.jl:174-178;
// lev.jl:31-37, :88-102; unienc.jl:36-55.  The seeded streams, their consumption order and every arithmetic step are
// those of mkt_client_party_keygen (client.cpp), so the generated words are identical to the host path's; the
// bootstrapping key is left in coefficient form for the caller to pre-transform (context.cpp).
//
// Randomness: the ChaCha20 streams of client.cpp (rng_chacha.h).  The streams are counter-based, so the workgroup
// fills a whole polynomial in parallel -- lane t generates keystream blocks t, t + 256, ... (8 uniform draws or
// 4 Gaussian deviates each) -- and still produces exactly the words the sequential host loop draws; the exact
// negacyclic product with the binary key is N^2/2 word additions spread over the workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_api.h"
#include "rng_chacha.h"

#pragma clang fp contract(off)

namespace mktd {
namespace {

using mktrng::Rng;
constexpr int KG_THREADS = 256;

// dst[q] = draw number (first + q) of the stream & wm, q < N  (N a multiple of 8, first a multiple of 8)
__device__ __forceinline__ void fill_uniform(const Rng &proto, uint64_t first, uint64_t *dst, int N, uint64_t wm) {
    for (int b = threadIdx.x; b < N / 8; b += KG_THREADS) {
        uint32_t w[16];
        mktrng::chacha20_block(proto.key, (uint32_t)(first / 8) + (uint32_t)b, proto.nonce, w);
#pragma unroll
        for (int i = 0; i < 8; i++) dst[8 * b + i] = ((uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32)) & wm;
    }
}
// dst[q] = noise(sigma) from draws first + 2q, first + 2q + 1  (Rng::noise order)
__device__ __forceinline__ void fill_noise(const Rng &proto, uint64_t first, uint64_t *dst, int N, double sigma) {
    for (int b = threadIdx.x; b < N / 4; b += KG_THREADS) {
        uint32_t w[16];
        mktrng::chacha20_block(proto.key, (uint32_t)(first / 8) + (uint32_t)b, proto.nonce, w);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t r1 = (uint64_t)w[4 * i] | ((uint64_t)w[4 * i + 1] << 32), r2 = (uint64_t)w[4 * i + 2] | ((uint64_t)w[4 * i + 3] << 32);
            dst[4 * b + i] = (uint64_t)(int64_t)rint(sigma * mktrng::box_muller(r1, r2));
        }
    }
}

template <typename WORD> __device__ __forceinline__ uint64_t wmask_of() { return sizeof(WORD) == 8 ? ~0ull : 0xFFFFFFFFull; }

// acc[r] (+)= sum_i s_i * a[(m - i) mod N] with the negacyclic sign, m = t + r * KG_THREADS; s in {-1, 0, 1};
// client.cpp mul_small_acc (out[i + j] -/+= a[j]) restated per output coefficient.  `negate` flips the sign.
template <int MAXR>
__device__ __forceinline__ void mul_small_acc_dev(const uint64_t *a_lds, const int8_t *s, uint64_t (&acc)[MAXR], int N, bool negate) {
    const int t = threadIdx.x, nr = N / KG_THREADS > 0 ? N / KG_THREADS : 1;
    for (int i = 0; i < N; i++) {
        const int si = s[i];                        // wave-uniform
        if (!si) continue;
        const bool neg = (si < 0) != negate;
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
            if (r >= nr) break;
            const int m = t + r * KG_THREADS;
            if (m >= N) break;
            const uint64_t v = a_lds[(m - i) & (N - 1)];
            const bool wrap = i > m;                // the term came around X^N = -1
            acc[r] += (neg != wrap) ? (uint64_t)0 - v : v;
        }
    }
}

// One RLWE sample of an RGSW row (gsw.jl:174-178, lev.jl:88-102, lwe.jl:78-93) per workgroup:
// sample index = (i * rows + c * l + j); out polys (b, a_0..a_{kr-1}) at native width.
template <typename WORD, int MAXR>
__global__ __launch_bounds__(KG_THREADS) void keygen_rgsw_kernel(KeygenArgs a) {
    extern __shared__ uint64_t kg_smem[];
    uint64_t *al = kg_smem;                        // [N] mask polynomial, then the noise
    const int N = a.N, t = threadIdx.x;
    const uint64_t wm = wmask_of<WORD>();
    const int rows = (a.kr + 1) * a.l, polys = a.kr + 1;
    const int sample = blockIdx.x, i = sample / rows, cj = sample % rows, c = cj / a.l, j = cj % a.l;
    WORD *out = reinterpret_cast<WORD *>(a.out) + (size_t)sample * polys * N;
    const Rng rng(a.key, (uint32_t)a.party, 2, (uint32_t)i, (uint32_t)cj);
    uint64_t acc[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; r++) acc[r] = 0;
    for (int cc = 0; cc < a.kr; cc++) {
        fill_uniform(rng, (uint64_t)cc * N, al, N, wm);
        __syncthreads();
        for (int q = t; q < N; q += KG_THREADS) out[(size_t)(1 + cc) * N + q] = (WORD)al[q];
        mul_small_acc_dev<MAXR>(al, a.zring + (size_t)(a.zoff + cc) * N, acc, N, true);
        __syncthreads();
    }
    fill_noise(rng, (uint64_t)a.kr * N, al, N, a.sigma_ring);
    __syncthreads();
    const uint64_t g = (uint64_t)1 << (a.W - (j + 1) * a.logB);
#pragma unroll
    for (int r = 0; r < MAXR; r++) {
        const int m = t + r * KG_THREADS;
        if (m >= N) break;
        out[m] = (WORD)((acc[r] + al[m]) & wm);
    }
    __syncthreads();
    // message: + s_i * g_j on coefficient 0 of polynomial c (c = 0: b; c >= 1: a_{c-1})
    if (t == 0) out[(size_t)c * N] = (WORD)((out[(size_t)c * N] + (uint64_t)a.lwekey[i] * g) & wm);
}

// UniEnc_z(s_i) (unienc.jl:36-55) per workgroup: d[j] = crs[j] * r + s_i g_j + e ; f[j] = RLWE_z(g_j r)
template <typename WORD, int MAXR>
__global__ __launch_bounds__(KG_THREADS) void keygen_unienc_kernel(KeygenArgs a) {
    extern __shared__ uint64_t kg_smem[];
    const int N = a.N, t = threadIdx.x, l = a.l, i = blockIdx.x;
    uint64_t *al = kg_smem;                        // [N] words
    int8_t *rt = reinterpret_cast<int8_t *>(kg_smem + N);   // [N] ternary r
    const uint64_t wm = wmask_of<WORD>();
    WORD *out = reinterpret_cast<WORD *>(a.out) + (size_t)i * 3 * l * N;
    const WORD *crs = reinterpret_cast<const WORD *>(a.crs);
    const Rng rng(a.key, (uint32_t)a.party, 2, (uint32_t)i);
    // stream layout of the host loop: draws [0, N) ternary r; per j at base = N + 5 N j: noise of d (2 draws each),
    // base + 2N: mask of f[j], base + 3N: noise of f[j]
    fill_uniform(rng, 0, al, N, ~0ull);
    __syncthreads();
    for (int q = t; q < N; q += KG_THREADS) rt[q] = (int8_t)((int)(al[q] % 3) - 1);
    __syncthreads();
    for (int j = 0; j < l; j++) {
        const uint64_t g = (uint64_t)1 << (a.W - (j + 1) * a.logB);
        uint64_t acc[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; r++) acc[r] = 0;
        for (int q = t; q < N; q += KG_THREADS) al[q] = crs[(size_t)j * N + q];
        __syncthreads();
        mul_small_acc_dev<MAXR>(al, rt, acc, N, false);                         // crs[j] * r
        __syncthreads();
        const uint64_t base = (uint64_t)N + (uint64_t)5 * N * j;
        fill_noise(rng, base, al, N, a.sigma_ring);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
            const int m = t + r * KG_THREADS;
            if (m >= N) break;
            uint64_t v = acc[r];
            if (m == 0) v += (uint64_t)a.lwekey[i] * g;
            out[(size_t)j * N + m] = (WORD)((v + al[m]) & wm);
            acc[r] = 0;
        }
        __syncthreads();
        // f.stack[j] = RLWE_z(g_j * r): a uniform, b = -a z + e + g r
        WORD *fb = out + (size_t)(l + 2 * j) * N, *fa = fb + N;
        fill_uniform(rng, base + (uint64_t)2 * N, al, N, wm);
        __syncthreads();
        for (int q = t; q < N; q += KG_THREADS) fa[q] = (WORD)al[q];
        mul_small_acc_dev<MAXR>(al, a.zring + (size_t)a.zoff * N, acc, N, true);
        __syncthreads();
        fill_noise(rng, base + (uint64_t)3 * N, al, N, a.sigma_ring);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
            const int m = t + r * KG_THREADS;
            if (m >= N) break;
            fb[m] = (WORD)((acc[r] + al[m] + g * (uint64_t)(int64_t)rt[m]) & wm);
        }
        __syncthreads();
    }
}

// Key-switching key (keygen.jl:17-23 etc., lev.jl:31-37, lwe.jl:11-22): one lane per extracted coefficient (c, j),
// rows [(c*N + j)][d][t] of n + 1 words at stride n1p.
__global__ void keygen_ksk_kernel(KeygenArgs a, uint32_t *ksk, int n1p, int kk, int dr, int is_block) {
    const int cj = blockIdx.x * blockDim.x + threadIdx.x;
    if (cj >= kk * a.N) return;
    const int c = cj / a.N, j = cj % a.N, n = a.n;
    if (is_block && (long)c * a.N + j < n) return;                    // keygen.jl:46, :147
    Rng rng(a.key, (uint32_t)a.party, 5, (uint32_t)cj);
    const uint32_t zj = (uint32_t)a.zring[(size_t)(a.zoff + c) * a.N + j];
    for (int d = 0; d < dr; d++)
        for (int t = 0; t < a.f; t++) {
            uint32_t *row = ksk + ((((size_t)c * a.N + j) * dr + d) * a.f + t) * n1p;
            const uint32_t msg = (uint32_t)(zj * (uint32_t)(d + 1)) << (32 - (t + 1) * a.logD);
            uint32_t dot = 0;
            for (int q = 0; q < n; q++) { const uint32_t w = (uint32_t)rng.next(); row[q] = w; dot += w * a.lwekey[q]; }
            row[n] = (uint32_t)rng.noise(a.sigma_lwe) - dot + msg;
        }
}

}  // namespace

hipError_t launch_keygen_brk(const KeygenArgs &a, int unienc, hipStream_t s) {
    const int N = a.N;
    if (N > 16 * KG_THREADS) return hipErrorInvalidValue;
    const size_t lds = (size_t)N * 8 + (unienc ? (size_t)N : 0);
    const unsigned grid = unienc ? (unsigned)a.n : (unsigned)(a.n * (a.kr + 1) * a.l);
#define MKT_KG_LAUNCH(K, WORD) \
    do { if (N <= 4 * KG_THREADS) hipLaunchKernelGGL((K<WORD, 4>), dim3(grid), dim3(KG_THREADS), lds, s, a); \
         else hipLaunchKernelGGL((K<WORD, 16>), dim3(grid), dim3(KG_THREADS), lds, s, a); } while (0)
    if (unienc) { if (a.W == 64) MKT_KG_LAUNCH(keygen_unienc_kernel, uint64_t); else MKT_KG_LAUNCH(keygen_unienc_kernel, uint32_t); }
    else { if (a.W == 64) MKT_KG_LAUNCH(keygen_rgsw_kernel, uint64_t); else MKT_KG_LAUNCH(keygen_rgsw_kernel, uint32_t); }
#undef MKT_KG_LAUNCH
    return hipGetLastError();
}

hipError_t launch_keygen_ksk(const KeygenArgs &a, uint32_t *ksk, int n1p, int kk, int dr, int is_block, hipStream_t s) {
    const int total = kk * a.N;
    hipLaunchKernelGGL(keygen_ksk_kernel, dim3((total + 63) / 64), dim3(64), 0, s, a, ksk, n1p, kk, dr, is_block);
    return hipGetLastError();
}

}  // namespace mktd
