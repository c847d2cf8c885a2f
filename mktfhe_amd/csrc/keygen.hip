// Evaluation-key generation on the device: the two large keys of a party -- the RGSW / UniEnc
// bootstrapping key and the LWE key-switching key -- sampled with exact integer arithmetic on the GPU from the party's
// secret keys.  Reference: keygen.jl:13-23, :39-51, :71-79, :106-114, :143-151; lwe.jl:11-22, :78-93; gsw.jl:174-178;
// lev.jl:31-37, :88-102; unienc.jl:36-55.  The seeded streams, their consumption order and every arithmetic step are
// those of mkt_client_party_keygen (client.cpp), so the generated words are identical to the host path's; the
// bootstrapping key is left in coefficient form for the caller to pre-transform (context.cpp).
//
// Randomness is a sequential xoshiro256** stream per sample (one lane draws, the workgroup multiplies): the draw is
// ~35 N generator steps per RLWE sample, the exact negacyclic product with the binary key N^2/2 word additions spread
// over the workgroup; thousands of samples run side by side, so a whole party key takes milliseconds.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_api.h"

#pragma clang fp contract(off)

namespace mktd {
namespace {

struct DRng {  // client.cpp Rng, operation for operation
    uint64_t s[4];
    __device__ static uint64_t splitmix(uint64_t &x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    __device__ DRng(uint64_t seed, uint64_t a, uint64_t b = 0, uint64_t c = 0) {
        uint64_t x = seed;
        x = splitmix(x) ^ (a * 0xD6E8FEB86659FD93ull); x = splitmix(x) ^ (b * 0xA0761D6478BD642Full);
        x = splitmix(x) ^ (c * 0xE7037ED1A0B428DBull);
        for (int i = 0; i < 4; i++) s[i] = splitmix(x);
    }
    __device__ static uint64_t rotl(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }
    __device__ uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    __device__ double uniform() { return (double)(next() >> 11) * 0x1p-53; }
    __device__ double gauss() {
        double acc = 0.0;
        for (int i = 0; i < 16; i++) acc += uniform();
        return (acc - 8.0) * 0.8660254037844386;
    }
    __device__ uint64_t noise(double sigma) { return (uint64_t)(int64_t)rint(sigma * gauss()); }
};

template <typename WORD> __device__ __forceinline__ uint64_t wmask_of() { return sizeof(WORD) == 8 ? ~0ull : 0xFFFFFFFFull; }

constexpr int KG_THREADS = 256;

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
    DRng rng(a.ps, 2, (uint64_t)i, (uint64_t)cj);
    uint64_t acc[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; r++) acc[r] = 0;
    for (int cc = 0; cc < a.kr; cc++) {
        if (t == 0) for (int q = 0; q < N; q++) al[q] = rng.next() & wm;
        __syncthreads();
        for (int q = t; q < N; q += KG_THREADS) out[(size_t)(1 + cc) * N + q] = (WORD)al[q];
        mul_small_acc_dev<MAXR>(al, a.zring + (size_t)(a.zoff + cc) * N, acc, N, true);
        __syncthreads();
    }
    if (t == 0) for (int q = 0; q < N; q++) al[q] = rng.noise(a.sigma_ring);
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
    DRng rng(a.ps, 2, (uint64_t)i);
    if (t == 0) for (int q = 0; q < N; q++) rt[q] = (int8_t)((int)(rng.next() % 3) - 1);
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
        if (t == 0) for (int q = 0; q < N; q++) al[q] = rng.noise(a.sigma_ring);
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
        if (t == 0) for (int q = 0; q < N; q++) al[q] = rng.next() & wm;
        __syncthreads();
        for (int q = t; q < N; q += KG_THREADS) fa[q] = (WORD)al[q];
        mul_small_acc_dev<MAXR>(al, a.zring + (size_t)a.zoff * N, acc, N, true);
        __syncthreads();
        if (t == 0) for (int q = 0; q < N; q++) al[q] = rng.noise(a.sigma_ring);
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
    DRng rng(a.ps, 5, (uint64_t)cj);
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
