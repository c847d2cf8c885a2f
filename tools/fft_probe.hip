// In-register transform throughput probe (gfx950): the forward / inverse transform of fft_device.h in a loop, alone and
// with the blind rotation's surrounding work (digit extraction + twist before, key-row multiply-adds after), at the
// occupancy the rotation kernels run at (4 workgroups of 128 threads per CU, 2 waves/SIMD).  Which part of a CMux step
// keeps the VALU from being busy?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I mktfhe_amd/csrc tools/fft_probe.hip -o tools/bin/fft_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>
#include "fft_device.h"
using namespace mktd;
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

template <int LOGM, int MODE, int NB>
__global__ __launch_bounds__((Plan<LOGM, 2>::NT)) __attribute__((amdgpu_waves_per_eu(2, 2)))
void probe(const cplx *psi, const cplx *rows, double *out, uint32_t *gacc, int iters, int l, int rowmask) {
    using P = Plan<LOGM, 2, NB>;
    constexpr int R = P::R, NT = P::NT, M = P::M;
    cplx *lds = reinterpret_cast<cplx *>(smem);
    cplx *psi_l = lds + Plan<LOGM, 2, 2>::LDS_CPLX;
    const int t = threadIdx.x;
    const LaneX lx = make_lanex();
    for (int i = t; i < M; i += NT) psi_l[i] = psi[i];
    __syncthreads();
    cplx z[NB][R], acc0[R], acc1[R], rt[R];
    uint32_t w[R][2];
    for (int e = 0; e < R; e++) {
        for (int b = 0; b < NB; b++) { z[b][e].re = 1.0 + 1e-3 * (t + e + b); z[b][e].im = 0.5 - 1e-3 * e; }
        acc0[e].re = acc0[e].im = acc1[e].re = acc1[e].im = 0.0; rt[e] = psi[(e * NT + t) & (M - 1)];
        w[e][0] = 0x9E3779B9u * (t + e + 1); w[e][1] = 0x85EBCA6Bu * (t + 3 * e + 7);
    }
    int dp[R];
    for (int e = 0; e < R; e++) dp[e] = dev_pos(t * R + e, NT);
    const Gadget<uint32_t> gd(l, 8);
    for (int it = 0; it < ((MODE == 5 || MODE == 6) ? 0 : iters); it++) {
        if (MODE == 0) fft_forward<LOGM, 2, NB>(z, psi_l, lds, t, lx);
        if (MODE == 1) fft_inverse<LOGM, 2, NB, true>(z, psi_l, lds, t, lx);
        if (MODE == 2 || MODE == 3) {           // the digit loop of a decomposition: [digits + twist] forward, 2 multiply-adds per point
            uint32_t tp[R][2];
            for (int e = 0; e < R; e++) { tp[e][0] = gd.prep(w[e][0] + it); tp[e][1] = gd.prep(w[e][1] ^ it); }
            for (int j = 0; j < l; j++) {
                for (int e = 0; e < R; e++) {
                    cplx v; v.re = (double)gd.digit(tp[e][0], j); v.im = (double)(-gd.digit(tp[e][1], j));
                    z[0][e] = cmul(v, rt[e]);
                }
                fft_forward<LOGM, 2, 1>(reinterpret_cast<cplx(&)[1][R]>(z[0]), psi_l, lds, t, lx);
                const cplx *r0 = rows + (size_t)((it * l + j) & 63) * 2 * M, *r1 = r0 + M;
                for (int e = 0; e < R; e++) {
                    const cplx k0 = MODE == 3 ? r0[dp[e]] : rt[e], k1 = MODE == 3 ? r1[dp[e]] : rt[R - 1 - e];
                    acc0[e] = cadd(acc0[e], cmul(z[0][e], k0)); acc1[e] = cadd(acc1[e], cmul(z[0][e], k1));
                }
            }
        }
        if (MODE == 4 || MODE == 7) {            // one whole plain CMux shape (7: times a row picked at random from a 16 MiB table, as the monomial is): 2 decompositions of l digits + 2 inverses (+ native)
            for (int c = 0; c < 2; c++) {
                uint32_t tp[R][2];
                for (int e = 0; e < R; e++) { tp[e][0] = gd.prep(w[e][0] + c); tp[e][1] = gd.prep(w[e][1] ^ c); }
                for (int j = 0; j < l; j++) {
                    for (int e = 0; e < R; e++) {
                        cplx v; v.re = (double)gd.digit(tp[e][0], j); v.im = (double)(-gd.digit(tp[e][1], j));
                        z[0][e] = cmul(v, rt[e]);
                    }
                    fft_forward<LOGM, 2, 1>(reinterpret_cast<cplx(&)[1][R]>(z[0]), psi_l, lds, t, lx);
                    const cplx *r0 = rows + (size_t)((it * 2 * l + c * l + j) & rowmask) * 2 * M, *r1 = r0 + M;
                    for (int e = 0; e < R; e++) { acc0[e] = cadd(acc0[e], cmul(z[0][e], r0[dp[e]])); acc1[e] = cadd(acc1[e], cmul(z[0][e], r1[dp[e]])); }
                }
            }
            if (MODE == 7) {
                const unsigned h = (blockIdx.x * 2654435761u + (unsigned)it * 40503u) >> 7;
                const cplx *mrow = rows + (size_t)(h & 2047) * M;          // 2048 rows of M points = 16 MiB
                for (int e = 0; e < R; e++) { const cplx mv = mrow[dp[e]]; acc0[e] = cmul(mv, acc0[e]); acc1[e] = cmul(mv, acc1[e]); }
            }
            fft_inverse<LOGM, 2, 1, true>(reinterpret_cast<cplx(&)[1][R]>(acc0), psi_l, lds, t, lx);
            fft_inverse<LOGM, 2, 1, true>(reinterpret_cast<cplx(&)[1][R]>(acc1), psi_l, lds, t, lx);
            for (int e = 0; e < R; e++) {
                const cplx v0 = cmul(acc0[e], rt[e]), v1 = cmul(acc1[e], rt[e]);
                w[e][0] += native<uint32_t>(v0.re); w[e][1] += native<uint32_t>(-v0.im);
                w[e][0] ^= native<uint32_t>(v1.re); w[e][1] ^= native<uint32_t>(-v1.im);
                acc0[e].re = acc0[e].im = acc1[e].re = acc1[e].im = 0.0;
            }
        }
    }
    if (MODE == 5 || MODE == 6) {               // the CCS step at np = 1 (bootstrapping.jl:263-324): accumulator (2 polynomials) in global memory
        uint32_t *accg = gacc + (size_t)blockIdx.x * 3 * 2 * M;
        uint32_t *vsc = accg + 2 * 2 * M;
        for (int it = 0; it < iters; it++) {
            cplx ta[R], tb[R], tu[R], tvq[R];
            uint32_t vw[R][2];
            auto uv = [&](int q, cplx (&tu_)[R], cplx (&tv_)[R]) {
                uint32_t tp[R][2];
                for (int e = 0; e < R; e++) { tp[e][0] = gd.prep(accg[q * 2 * M + e * NT + t]); tp[e][1] = gd.prep(accg[q * 2 * M + M + e * NT + t]); }
                for (int e = 0; e < R; e++) { tu_[e].re = tu_[e].im = 0.0; tv_[e].re = tv_[e].im = 0.0; }
                for (int j = 0; j < l; j++) {
                    cplx zz[R];
                    for (int e = 0; e < R; e++) { cplx v; v.re = (double)gd.digit(tp[e][0], j); v.im = (double)(-gd.digit(tp[e][1], j)); zz[e] = cmul(v, rt[e]); }
                    fft_forward<LOGM, 2, 1>(reinterpret_cast<cplx(&)[1][R]>(zz), psi_l, lds, t, lx);
                    const cplx *r0 = rows + (size_t)((it * l + j + q) & 31) * 2 * M, *r1 = r0 + M;
                    for (int e = 0; e < R; e++) { tu_[e] = cadd(tu_[e], cmul(zz[e], r0[dp[e]])); const cplx pr = cmul(zz[e], r1[dp[e]]); tv_[e] = q == 0 ? csub(tv_[e], pr) : cadd(tv_[e], pr); }
                }
            };
            auto wpart = [&](const uint32_t (&vw_)[R][2]) {
                uint32_t tp[R][2];
                for (int e = 0; e < R; e++) { tp[e][0] = gd.prep(vw_[e][0]); tp[e][1] = gd.prep(vw_[e][1]); }
                for (int j = 0; j < l; j++) {
                    cplx zz[R];
                    for (int e = 0; e < R; e++) { cplx v; v.re = (double)gd.digit(tp[e][0], j); v.im = (double)(-gd.digit(tp[e][1], j)); zz[e] = cmul(v, rt[e]); }
                    fft_forward<LOGM, 2, 1>(reinterpret_cast<cplx(&)[1][R]>(zz), psi_l, lds, t, lx);
                    const cplx *r0 = rows + (size_t)(32 + ((it * l + j) & 31)) * 2 * M, *r1 = r0 + M;
                    for (int e = 0; e < R; e++) { tb[e] = cadd(tb[e], cmul(zz[e], r0[dp[e]])); ta[e] = cadd(ta[e], cmul(zz[e], r1[dp[e]])); }
                }
            };
            auto inv_words = [&](cplx (&zz)[R], uint32_t (&ww)[R][2]) {
                fft_inverse<LOGM, 2, 1, true>(reinterpret_cast<cplx(&)[1][R]>(zz), psi_l, lds, t, lx);
                for (int e = 0; e < R; e++) { const cplx v = cmul(zz[e], MODE == 6 ? rt[e] : psi[(e * NT + t)]); ww[e][0] = native<uint32_t>(v.re); ww[e][1] = native<uint32_t>(-v.im); }
            };
            uv(1, ta, tvq);
            inv_words(tvq, vw);
            for (int e = 0; e < R; e++) { vsc[e * NT + t] = vw[e][0]; vsc[M + e * NT + t] = vw[e][1]; }
            uv(0, tb, tvq);
            inv_words(tvq, vw);
            wpart(vw);
            for (int e = 0; e < R; e++) { vw[e][0] = vsc[e * NT + t]; vw[e][1] = vsc[M + e * NT + t]; }
            wpart(vw);
            const cplx *mono = rows + (size_t)(it & 63) * 2 * M;
            for (int q = 0; q <= 1; q++) {
                cplx sgm[R];
                for (int e = 0; e < R; e++) sgm[e] = cmul(mono[dp[e]], q == 0 ? tb[e] : ta[e]);
                uint32_t ww[R][2];
                inv_words(sgm, ww);
                for (int e = 0; e < R; e++) { accg[q * 2 * M + e * NT + t] += ww[e][0]; accg[q * 2 * M + M + e * NT + t] += ww[e][1]; }
            }
        }
    }
    double s = 0;
    for (int e = 0; e < R; e++) { for (int b = 0; b < NB; b++) s += z[b][e].re + z[b][e].im; s += acc0[e].re + acc1[e].im + (double)w[e][0] + (double)w[e][1]; }
    out[(size_t)blockIdx.x * NT + t] = s;
}

template <int LOGM, int MODE, int NB>
void run(const char *name, const cplx *psi, const cplx *rows, double *out, uint32_t *gacc, int blocks, int iters, int l, double transforms_per_iter, int rowmask = 63) {
    using P2 = Plan<LOGM, 2, 2>;
    const size_t lds = P2::LDS_BYTES + (size_t)P2::M * sizeof(cplx);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<LOGM, MODE, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<LOGM, MODE, NB>), dim3(blocks), dim3(P2::NT), lds, 0, psi, rows, out, gacc, iters, l, rowmask);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double ntr = (double)blocks * iters * transforms_per_iter;
    const int M = 1 << LOGM, lg = LOGM;
    printf("%-58s blocks %5d: %8.3f ms  %7.1f M transforms/s  %6.3f us per transform per workgroup slot (x%d resident)\n", name, blocks, ms, ntr / ms / 1e3,
           ms * 1e3 / (iters * transforms_per_iter) * (blocks >= 1024 ? 1.0 : 1.0), blocks / 256);
    (void)M; (void)lg;
}

int main() {
    constexpr int LOGM = 9, M = 1 << LOGM;
    const size_t NROWS = 4096;   // 4096 row pairs of 16 KiB = 64 MiB: a key the size of the real ones
    std::vector<cplx> h(NROWS * 2 * M);
    for (size_t i = 0; i < h.size(); i++) { h[i].re = 0.001 * (double)(i % 977) - 0.4; h[i].im = 0.002 * (double)(i % 613) - 0.6; }
    cplx *psi, *rows; double *out; uint32_t *gacc;
    hipMalloc(&gacc, (size_t)1024 * 3 * 2 * M * 4); hipMemset(gacc, 0x5a, (size_t)1024 * 3 * 2 * M * 4);
    hipMalloc(&psi, M * sizeof(cplx)); hipMalloc(&rows, h.size() * sizeof(cplx)); hipMalloc(&out, (size_t)1024 * 128 * 8);
    hipMemcpy(psi, h.data(), M * sizeof(cplx), hipMemcpyHostToDevice);
    hipMemcpy(rows, h.data(), h.size() * sizeof(cplx), hipMemcpyHostToDevice);
    const int iters = getenv("PROBE_ITERS") ? atoi(getenv("PROBE_ITERS")) : 2000;
    for (int blocks : {256, 1024}) {
        run<LOGM, 0, 1>("forward, single", psi, rows, out, gacc, blocks, iters, 3, 1);
        run<LOGM, 0, 2>("forward, pairs", psi, rows, out, gacc, blocks, iters, 3, 2);
        run<LOGM, 1, 1>("inverse, single", psi, rows, out, gacc, blocks, iters, 3, 1);
        run<LOGM, 1, 2>("inverse, pairs", psi, rows, out, gacc, blocks, iters, 3, 2);
        run<LOGM, 2, 1>("digit loop (l = 3): digits + twist + forward + 2 MACs, register rows", psi, rows, out, gacc, blocks, iters / 2, 3, 3);
        run<LOGM, 3, 1>("digit loop (l = 3): ... rows from global memory", psi, rows, out, gacc, blocks, iters / 2, 3, 3);
        run<LOGM, 4, 1>("CMux shape: 2 x 3 digit transforms + 2 inverses + native", psi, rows, out, gacc, blocks, iters / 4, 3, 8);
        run<LOGM, 4, 1>("CMux shape, key rows streamed from a 64 MiB table", psi, rows, out, gacc, blocks, iters / 4, 3, 8, 4095);
        run<LOGM, 7, 1>("CMux shape, streamed key rows + a random row of a 16 MiB table per step", psi, rows, out, gacc, blocks, iters / 4, 3, 8, 4095);
        run<LOGM, 5, 1>("CCS step shape, np = 1 (16 transforms), accumulator in global memory", psi, rows, out, gacc, blocks, iters / 8, 3, 16);
        run<LOGM, 6, 1>("CCS step shape, untwist factors from registers", psi, rows, out, gacc, blocks, iters / 8, 3, 16);
    }
    return 0;
}
