// EXACT-vs-F64REF decision probe: register-resident butterfly throughput on gfx950 of
//   (a) the Float64 complex radix-2 butterfly the engine ships (u = b * w; a + u, a - u: 10 flops, no FMA), and
//   (b) a 64-bit-prime NTT butterfly, best case: the Goldilocks prime p = 2^64 - 2^32 + 1 (special-form reduction,
//       no Montgomery constants), u = b * w mod p; a + u mod p, a - u mod p.
// One complex butterfly advances 4 real coefficients of a folded negacyclic transform of N reals by one stage (M = N/2
// complex points, log2 M stages); one NTT butterfly advances 2 coefficients (N points, log2 N stages).  An exact
// product over a 64-bit ring with 16-bit digits additionally needs 92 > 64 bits of head room, i.e. two such primes.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ntt_probe.hip -o tools/bin/ntt_probe && tools/bin/ntt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#pragma clang fp contract(off)

constexpr uint64_t P = 0xFFFFFFFF00000001ull, EPS = 0xFFFFFFFFull;

__device__ __forceinline__ uint64_t gl_add(uint64_t a, uint64_t b) { uint64_t s = a + b; if (s < a) s += EPS; if (s >= P) s -= P; return s; }
__device__ __forceinline__ uint64_t gl_sub(uint64_t a, uint64_t b) { uint64_t d = a - b; if (a < b) d -= EPS; return d; }
__device__ __forceinline__ uint64_t gl_mul(uint64_t a, uint64_t b) {
    const uint64_t lo = a * b, hi = __umul64hi(a, b);
    const uint64_t hh = hi >> 32, hl = hi & EPS;
    uint64_t t0 = lo - hh; if (lo < hh) t0 -= EPS;
    const uint64_t t1 = hl * EPS;
    uint64_t r = t0 + t1; if (r < t0) r += EPS;
    return r >= P ? r - P : r;
}

__global__ __launch_bounds__(256) void fft_bfly(double *out, int iters, double seed) {
    double ar[4], ai[4], br[4], bi[4];
    for (int i = 0; i < 4; i++) { ar[i] = seed + threadIdx.x + i; ai[i] = seed * 0.5 + i; br[i] = 1.0 + 1e-3 * i; bi[i] = 0.25 + 1e-3 * threadIdx.x; }
    const double wr = 0.9999999, wi = 4.0e-4;
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const double p0 = br[i] * wr, p1 = bi[i] * wi, p2 = br[i] * wi, p3 = bi[i] * wr;
            const double ur = p0 - p1, ui = p2 + p3;
            const double nr = ar[i] + ur, ni = ai[i] + ui;
            br[i] = ar[i] - ur; bi[i] = ai[i] - ui; ar[i] = nr; ai[i] = ni;
        }
    double s = 0; for (int i = 0; i < 4; i++) s += ar[i] + ai[i] + br[i] + bi[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void ntt_bfly(uint64_t *out, int iters, uint64_t seed) {
    uint64_t a[4], b[4];
    for (int i = 0; i < 4; i++) { a[i] = (seed * (threadIdx.x + 7 + i)) % P; b[i] = (seed ^ (0x9E3779B97F4A7C15ull * (threadIdx.x + i + 1))) % P; }
    const uint64_t w = 0x0123456789ABCDEFull % P;
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t u = gl_mul(b[i], w);
            const uint64_t n = gl_add(a[i], u);
            b[i] = gl_sub(a[i], u); a[i] = n;
        }
    uint64_t s = 0; for (int i = 0; i < 4; i++) s += a[i] ^ b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int blocks = 256 * 8, threads = 256, iters = 20000;
    void *d; hipMalloc(&d, (size_t)blocks * threads * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    for (int which = 0; which < 2; which++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(fft_bfly, dim3(blocks), dim3(threads), 0, 0, (double *)d, iters, 1.25);
            else hipLaunchKernelGGL(ntt_bfly, dim3(blocks), dim3(threads), 0, 0, (uint64_t *)d, iters, 0x1234567ull);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[which], e0, e1);
        }
    }
    const double nb = (double)blocks * threads * 4.0 * iters;
    printf("f64 complex butterfly (10 flops, no FMA): %.3f ms  %.1f G butterflies/s  = %.1f G real-coefficient-stages/s  (%.2f TFLOP/s)\n",
           ms[0], nb / ms[0] / 1e6, 4 * nb / ms[0] / 1e6, 10 * nb / ms[0] / 1e9);
    printf("Goldilocks 64-bit NTT butterfly:          %.3f ms  %.1f G butterflies/s  = %.1f G coefficient-stages/s\n",
           ms[1], nb / ms[1] / 1e6, 2 * nb / ms[1] / 1e6);
    printf("per coefficient and stage the integer butterfly is %.1fx slower; with log2(N) = log2(M) + 1 stages and two primes\n"
           "for a 64-bit ring with 16-bit digits the exact transform costs ~%.0fx the Float64 one\n",
           (4 * nb / ms[0]) / (2 * nb / ms[1]), 2.0 * (10.0 / 9.0) * (4 * nb / ms[0]) / (2 * nb / ms[1]));
    return 0;
}
