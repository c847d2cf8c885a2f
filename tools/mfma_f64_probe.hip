// Can the idle matrix pipe carry the Float64 butterflies' additions?  v_mfma_f64_16x16x4_f64 computes
// D[i][j] = C[i][j] + sum_k A[i][k] B[k][j]; with A[i][k] = +-1 at k = i / 2 (rows 0..7, zero elsewhere) row 2m gives C + u_m and row
// 2m + 1 gives C - u_m for the four u_m = B[m][j] of column j: 128 butterfly outputs per instruction (rows 8..15 idle).
//   (i)  bit identity with v_add_f64 on random operands, signed zeros, denormals, infinities, cancellation;
//   (ii) cycles per MFMA alone, per v_mul_f64 alone, and of the two interleaved (co-issue);
// The register-layout cost (iii) is arithmetic, not measured: A and B want lane = 16 k + i, C / D want col = lane & 15, row = (lane >> 4) + 4 reg.
//   make -C tools bin/mfma_f64_probe && tools/bin/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

typedef double d4 __attribute__((ext_vector_type(4)));

// one wave: C[16][16] given per lane as the D layout (reg r of lane L = row (L >> 4) + 4 r, column L & 15); u[4][16] as B (lane 16 k + j)
__global__ void exact_check(const double *cmat, const double *umat, double *dmat, int ncase) {
    const int L = threadIdx.x;
    for (int c = 0; c < ncase; c++) {
        const double *C = cmat + (size_t)c * 256, *U = umat + (size_t)c * 64;
        const int i = L & 15, k = L >> 4;                      // A: row i, column k
        double a = 0.0;
        if (i < 8 && (i >> 1) == k) a = (i & 1) ? -1.0 : 1.0;
        else a = (L & 1) ? -0.0 : 0.0;                         // unused slots: signed zeros of both kinds
        const double b = U[k * 16 + (L & 15)];
        d4 acc;
        for (int r = 0; r < 4; r++) acc[r] = C[((L >> 4) + 4 * r) * 16 + (L & 15)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        for (int r = 0; r < 4; r++) dmat[(size_t)c * 256 + ((L >> 4) + 4 * r) * 16 + (L & 15)] = acc[r];
    }
}

template <int MODE>   // 0: MFMA only, 1: v_mul_f64 only, 2: one MFMA + 16 v_mul_f64 interleaved, 3: one MFMA + 8 v_mul_f64
__global__ __launch_bounds__(1024) void rate(uint64_t *out, int iters, double seed) {
    double a = seed + threadIdx.x, b = 1.0000001;
    d4 c0 = {a, a + 1, a + 2, a + 3}, c1 = c0, c2 = c0, c3 = c0;
    double m0 = a, m1 = a + 1, m2 = a + 2, m3 = a + 3, m4 = a + 4, m5 = a + 5, m6 = a + 6, m7 = a + 7;
    asm volatile("s_nop 0" ::: "memory");
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#define MF(c) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#define MU8 asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4), "+v"(m5), "+v"(m6), "+v"(m7) : "v"(b));
        if (MODE == 0) { MF(c0) MF(c1) MF(c2) MF(c3) }
        if (MODE == 1) { MU8 MU8 MU8 MU8 MU8 MU8 MU8 MU8 }
        if (MODE == 2) { MF(c0) MU8 MU8 MF(c1) MU8 MU8 MF(c2) MU8 MU8 MF(c3) MU8 MU8 }
        if (MODE == 3) { MF(c0) MU8 MF(c1) MU8 MF(c2) MU8 MF(c3) MU8 }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const double s = c0[0] + c1[1] + c2[2] + c3[3] + m0 + m1 + m2 + m3 + m4 + m5 + m6 + m7;
    if (s == 12345.678) out[1 << 16] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE> double run(uint64_t *d, int waves) {
    const int iters = 2000, blocks = 256;
    std::vector<uint64_t> h(blocks * 16);
    hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(64 * waves), 0, 0, d, iters, 1.25);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> v;
    for (int b = 0; b < blocks; b++) for (int w = 0; w < waves; w++) v.push_back((double)h[b * 16 + w]);
    std::sort(v.begin(), v.end());
    return v.back() / iters;            // the slowest wave: what a fixed amount of work takes
}

int main() {
    // (i) exactness
    const int ncase = 4096;
    std::vector<double> C((size_t)ncase * 256), U((size_t)ncase * 64), D((size_t)ncase * 256);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto special = [&](uint64_t r) -> double {
        static const uint64_t sp[] = {0x0000000000000000ull, 0x8000000000000000ull, 0x0000000000000001ull, 0x800fffffffffffffull, 0x7ff0000000000000ull, 0xfff0000000000000ull,
                                      0x7fefffffffffffffull, 0x3ff0000000000000ull, 0xbff0000000000000ull, 0x4340000000000000ull, 0x0010000000000000ull};
        uint64_t bits = (r % 7 == 0) ? sp[(r >> 8) % 11] : ((r % 5 == 0) ? (r & 0x800fffffffffffffull) | ((uint64_t)(1000 + (r >> 52) % 60) << 52) : r);
        if (((bits >> 52) & 0x7ff) == 0x7ff && (bits & 0xfffffffffffffull)) bits &= ~0xfffffffffffffull;      // no NaN payloads
        double d; memcpy(&d, &bits, 8); return d;
    };
    for (auto &x : C) x = special(rnd());
    for (auto &x : U) { x = special(rnd()); if (x - x != 0.0) x = 1.5; }      // u finite: an infinity in ANOTHER k slot of the column would meet a zero of A (NaN), which says nothing about the add
    for (int c = 0; c < ncase; c += 3) for (int j = 0; j < 16; j++) for (int m = 0; m < 4; m++) C[(size_t)c * 256 + (2 * m) * 16 + j] = U[(size_t)c * 64 + m * 16 + j];   // exact cancellation in the minus rows
    double *dc, *du, *dd; uint64_t *dt;
    (void)hipMalloc(&dc, C.size() * 8); (void)hipMalloc(&du, U.size() * 8); (void)hipMalloc(&dd, D.size() * 8); (void)hipMalloc(&dt, ((1 << 16) + 8) * 8);
    (void)hipMemcpy(dc, C.data(), C.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(du, U.data(), U.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(exact_check, dim3(1), dim3(64), 0, 0, dc, du, dd, ncase);
    (void)hipMemcpy(D.data(), dd, D.size() * 8, hipMemcpyDeviceToHost);
    size_t bad = 0, badzero = 0, checked = 0, idle_bad = 0;
    for (int c = 0; c < ncase; c++)
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                const double cc = C[(size_t)c * 256 + i * 16 + j], got = D[(size_t)c * 256 + i * 16 + j];
                double want;
                if (i < 8) { const double u = U[(size_t)c * 64 + (i >> 1) * 16 + j]; want = (i & 1) ? cc - u : cc + u; }
                else want = cc;                                  // idle rows: C + zeros (with infinities in U: inf * 0 = NaN -- counted apart)
                const bool nan_w = want != want, nan_g = got != got;
                const bool same = (nan_w && nan_g) || (!nan_w && !nan_g && memcmp(&want, &got, 8) == 0);
                if (i < 8) { checked++; if (!same) { bad++; if (want == 0.0 && got == 0.0) badzero++; } }
                else if (!same) idle_bad++;
            }
    printf("(i) %zu butterfly outputs through the MFMA against v_add_f64 (host IEEE add): %zu differ (%zu of them only in the sign of a zero); idle rows changed: %zu\n", checked, bad, badzero, idle_bad);
    // (ii) rates: cycles per loop trip (4 MFMAs, 64 v_mul_f64, or both)
    for (int waves : {4, 8}) {
        const double mf = run<0>(dt, waves), mu = run<1>(dt, waves), both = run<2>(dt, waves), half = run<3>(dt, waves);
        printf("(ii) %d wave(s) per SIMD: 4 MFMA alone %.0f cycles (%.1f each); 64 v_mul_f64 alone %.0f (%.2f each); 4 MFMA + 64 v_mul_f64 interleaved %.0f (sum would be %.0f); 4 MFMA + 32 v_mul_f64 %.0f\n",
               waves / 4, mf, mf / 4, mu, mu / 64, both, mf + mu, half);
    }
    return 0;
}
