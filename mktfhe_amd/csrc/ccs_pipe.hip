// CCS blind rotation (bootstrapping.jl:234-328) for batches that leave compute units idle: ONE ciphertext on TWO thread
// groups of a workgroup, as a two-stage pipeline over the step's input polynomials.
//
// What can run side by side in a step (party idx, key bit i; np = idx + 1 mask polynomials).  For every input polynomial q:
//   stage 1   decompose acc[q], l forward transforms, u_q = sum_j dig_j d[j], v_q = -/+ sum_j dig_j (crs | b_{q-1})[j], inverse of v_q
//   stage 2   decompose v_q, l forward transforms, tacc.b += dig_j f[j].b, tacc.a[idx] += dig_j f[j].a
// The stage-1 jobs are independent; the stage-2 jobs form ONE chain, because tacc.b and tacc.a[idx] are Float64 sums that
// the reference forms in a fixed order (u_0 / u_np first, then w(v_0), w(v_1) ... w(v_np): :279-284, :313-320) and every
// rounding must be the reference's.  So group B owns tacc.b and tacc.a[idx] in registers and runs the stage-2 chain in
// that order, group A runs stage 1 one polynomial ahead of it and hands the v words over; the first slot has both groups
// on stage 1 (A: polynomial np, whose u opens tacc.a[idx]; B: polynomial 0, whose u opens tacc.b), the np + 1 output
// polynomials (monomial, inverse, add: :322-324) alternate between the groups.  A step costs (np + 2) slots of l + 1
// transforms plus ceil((np + 1) / 2) inverses instead of (np + 1)(2 l + 2) transforms: 1.23x at two parties, 1.6x at eight.
// Same operations on the same values in the same order as ccs_blindrotate_kernel: bit-identical (tests force both).
//
// Both groups meet at every workgroup barrier, so each slot has ONE barrier shape (l forward transforms + one inverse);
// a group with nothing to transform walks the barriers only.  Hand-offs (v words, u_np, the finished tacc.b / tacc.a[idx],
// the accumulator) go through the per-ciphertext scratch in global memory: the waves of a workgroup share their CU's L1,
// so a workgroup barrier orders them.
#include "kernel_common.h"

#pragma clang fp contract(off)

namespace mktd {

// (Paired digit transforms inside the stages were built and measured: both groups' live state shares one register allocation, the
// pairs spill 200-470 B and run 9-40 % slower.)
template <int LOGM, typename WORD, int LT, int BT>
__global__ __launch_bounds__((2 * Plan<LOGM, LOGR>::NT)) __attribute__((amdgpu_waves_per_eu(2, 2))) void ccs_pipe_kernel(const CcsArgs a) {
    using P = Plan<LOGM, LOGR, 1>;
    constexpr int R = P::R, NT = P::NT, M = P::M, N = 2 * M;
    constexpr int MO = 0xff;
    cplx *psi_l = reinterpret_cast<cplx *>(mkt_smem);
    const int tid = threadIdx.x, grp = tid / NT, t = tid % NT;          // grp 0 = A (stage 1), 1 = B (stage-2 chain)
    cplx *lds = psi_l + M + (size_t)grp * P::LDS_CPLX;                  // this group's FFT staging
    XS xs = make_xs();
    for (int i = tid; i < M; i += 2 * NT) psi_l[i] = a.tw.psi[i];
    __syncthreads();
    const size_t g = blockIdx.x;
    const int k = a.k, l = LT ? LT : a.l, n = a.n;
    WORD *acc = reinterpret_cast<WORD *>(a.acc) + g * (size_t)(k + 1) * N;
    cplx *sc = a.scratch + g * (size_t)(k + 1) * M;
    WORD *vsc = reinterpret_cast<WORD *>(a.vscratch) + g * (size_t)3 * N;   // [0]: parked v_np, [1], [2]: hand-off slots
    const Gadget<WORD> gd(l, (LT && BT) ? BT : a.logB);
    const int msbit = 32 - a.logN - 1;
    int dp[R];
#pragma unroll
    for (int e = 0; e < R; e++) dp[e] = dev_pos(MKT_DEVORDER, t * R + e, NT);
    cplx rt[R];
#pragma unroll
    for (int e = 0; e < R; e++) rt[e] = a.tw.roots[e * NT + t];

    // stage 1 without its inverse: tu = sum_j dig_j * d[j]; tv = -/+ sum_j dig_j * (crs | b_{q-1})[j]   (:279-294)
    auto uv = [&](int q, const cplx *ud, cplx (&tu)[R], cplx (&tvq)[R]) {
        WORD tp[R][2];
#pragma unroll
        for (int e = 0; e < R; e++) { tp[e][0] = gd.prep(acc[(size_t)q * N + e * NT + t]); tp[e][1] = gd.prep(acc[(size_t)q * N + M + e * NT + t]); }
#pragma unroll
        for (int e = 0; e < R; e++) { tu[e].re = tu[e].im = 0.0; tvq[e].re = tvq[e].im = 0.0; }
        const cplx *vk = q == 0 ? a.crs : a.pub_b + (size_t)(q - 1) * l * M;
#pragma unroll 1
        for (int j = 0; j < l; j++) {
            cplx z[R];
            const cplx *kd = ud + (size_t)j * M, *kv = vk + (size_t)j * M;
            cplx kdr[R], kvr[R];
#pragma unroll
            for (int e = 0; e < R; e++) { kdr[e] = kd[dp[e]]; kvr[e] = kv[dp[e]]; }
            __builtin_amdgcn_sched_barrier(0);
            digit_points<WORD, R>(z, tp, gd, j, rt);
            fft_forward<LOGM, LOGR, 1, MO>(reinterpret_cast<cplx(&)[1][R]>(z), psi_l, lds, t, xs.lx);
#pragma unroll
            for (int e = 0; e < R; e++) {
                tu[e] = cadd(tu[e], cmul(z[e], kdr[e]));
                const cplx pr = cmul(z[e], kvr[e]);
                tvq[e] = q == 0 ? csub(tvq[e], pr) : cadd(tvq[e], pr);          // :290 mulsubto!, :293 muladdto!
            }
        }
    };
    // stage 2: w contribution of one v polynomial given as words (:313-320)
    auto wpart = [&](const WORD (&vw)[R][2], const cplx *uf, cplx (&tb)[R], cplx (&ta)[R]) {
        WORD tp[R][2];
#pragma unroll
        for (int e = 0; e < R; e++) { tp[e][0] = gd.prep(vw[e][0]); tp[e][1] = gd.prep(vw[e][1]); }
#pragma unroll 1
        for (int j = 0; j < l; j++) {
            cplx z[R];
            const cplx *fb = uf + (size_t)(2 * j) * M, *fa = fb + M;
            cplx fbr[R], far[R];
#pragma unroll
            for (int e = 0; e < R; e++) { fbr[e] = fb[dp[e]]; far[e] = fa[dp[e]]; }
            __builtin_amdgcn_sched_barrier(0);
            digit_points<WORD, R>(z, tp, gd, j, rt);
            fft_forward<LOGM, LOGR, 1, MO>(reinterpret_cast<cplx(&)[1][R]>(z), psi_l, lds, t, xs.lx);
#pragma unroll
            for (int e = 0; e < R; e++) { tb[e] = cadd(tb[e], cmul(z[e], fbr[e])); ta[e] = cadd(ta[e], cmul(z[e], far[e])); }
        }
    };
    auto inv_words = [&](cplx (&z)[R], WORD (&w)[R][2]) {                        // fft.jl:74-81
        fft_inverse<LOGM, LOGR, 1, true, MO>(reinterpret_cast<cplx(&)[1][R]>(z), psi_l, lds, t, xs.lx);
#pragma unroll
        for (int e = 0; e < R; e++) {
            const cplx v = cmul(z[e], a.tw.rootsinv[e * NT + t]);
            w[e][0] = native<WORD>(v.re); w[e][1] = native<WORD>(-v.im);
        }
    };
    auto idle_forwards = [&]() { for (int j = 0; j < l; j++) fft_forward_barriers_only<LOGM, LOGR, 1, MO>(); };   // the barrier shape of a stage's forward transforms
    auto idle_inverse = [&]() { fft_inverse_barriers_only<LOGM, LOGR, 1, MO>(); };
    auto put_words = [&](WORD *dst, const WORD (&w)[R][2]) {
#pragma unroll
        for (int e = 0; e < R; e++) { dst[e * NT + t] = w[e][0]; dst[M + e * NT + t] = w[e][1]; }
    };
    auto get_words = [&](const WORD *src, WORD (&w)[R][2]) {
#pragma unroll
        for (int e = 0; e < R; e++) { w[e][0] = src[e * NT + t]; w[e][1] = src[M + e * NT + t]; }
    };

    for (int idx = 0; idx < k; idx++) {
        const int np = idx + 1;
        const uint32_t *at_src = a.lwe + g * (size_t)a.lwe_stride + (size_t)idx * n;
        for (int i = 0; i < n; i++) {
            const uint32_t v0 = at_src[i];
            const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.pre_switched ? v0 : divbits<uint32_t>(v0, msbit)));
            if (at == 0) continue;                                               // :261
            const cplx *uni = a.brk + (size_t)idx * a.brk_party_stride + (size_t)i * 3 * l * M;
            const cplx *ud = uni, *uf = uni + (size_t)l * M;
            cplx ta[R], tb[R];                                                   // live in group B
            WORD vkeep[R][2];                                                    // B: its own v_0
            // ---- slot 0: both groups on stage 1 -- A: polynomial np (u opens tacc.a[idx], v parked), B: polynomial 0 (u opens tacc.b)
            {
                cplx tu[R], tvq[R];
                uv(grp == 0 ? np : 0, ud, tu, tvq);
                if (grp == 0) {
#pragma unroll
                    for (int e = 0; e < R; e++) sc[(size_t)np * M + dp[e]] = tu[e];
                } else {
#pragma unroll
                    for (int e = 0; e < R; e++) tb[e] = tu[e];
                }
                WORD vw[R][2];
                inv_words(tvq, vw);                                              // :297-300
                if (grp == 0) put_words(vsc, vw);
                else {
#pragma unroll
                    for (int e = 0; e < R; e++) { vkeep[e][0] = vw[e][0]; vkeep[e][1] = vw[e][1]; }
                }
            }
            __syncthreads();
            if (grp == 1) {
#pragma unroll
                for (int e = 0; e < R; e++) ta[e] = sc[(size_t)np * M + dp[e]];
            }
            // ---- slots 1 .. np + 1: B runs the stage-2 chain v_0, v_1 ... v_{np-1}, v_np; A stage 1 of polynomials 1 .. np - 1
            for (int s = 1; s <= np + 1; s++) {
                if (grp == 0) {
                    if (s <= np - 1) {
                        cplx tu[R], tvq[R];
                        uv(s, ud, tu, tvq);
#pragma unroll
                        for (int e = 0; e < R; e++) sc[(size_t)s * M + dp[e]] = tu[e];
                        WORD vw[R][2];
                        inv_words(tvq, vw);
                        put_words(vsc + (size_t)(1 + (s & 1)) * N, vw);
                    } else { idle_forwards(); idle_inverse(); }
                } else {
                    WORD vw[R][2];
                    if (s == 1) {
#pragma unroll
                        for (int e = 0; e < R; e++) { vw[e][0] = vkeep[e][0]; vw[e][1] = vkeep[e][1]; }
                    } else if (s <= np) get_words(vsc + (size_t)(1 + ((s - 1) & 1)) * N, vw);
                    else get_words(vsc, vw);
                    wpart(vw, uf, tb, ta);                                       // :313-316 (v_0), :317-320 (j1 = 1 .. np)
                    idle_inverse();
                }
                __syncthreads();
            }
            // ---- outputs (:322-324 mul!(monomial, tacc); ifftto!; add!): tacc.b and tacc.a[idx] join the others in the scratch
            if (grp == 1) {
#pragma unroll
                for (int e = 0; e < R; e++) { sc[dp[e]] = tb[e]; sc[(size_t)np * M + dp[e]] = ta[e]; }
            }
            __syncthreads();
            const cplx *mono = a.monomial + (size_t)(at - 1) * M;
            cplx mrow[R];
#pragma unroll
            for (int e = 0; e < R; e++) mrow[e] = mono[dp[e]];
            for (int u = 0; 2 * u <= np; u++) {
                const int q = 2 * u + grp;
                if (q <= np) {
                    cplx sv[R];
                    WORD aw[R][2];
#pragma unroll
                    for (int e = 0; e < R; e++) { sv[e] = cmul(mrow[e], sc[(size_t)q * M + dp[e]]); aw[e][0] = acc[(size_t)q * N + e * NT + t]; aw[e][1] = acc[(size_t)q * N + M + e * NT + t]; }
                    WORD w[R][2];
                    inv_words(sv, w);
#pragma unroll
                    for (int e = 0; e < R; e++) {
                        acc[(size_t)q * N + e * NT + t] = (WORD)(aw[e][0] + w[e][0]);
                        acc[(size_t)q * N + M + e * NT + t] = (WORD)(aw[e][1] + w[e][1]);
                    }
                } else idle_inverse();
            }
            __syncthreads();                                                     // the accumulator and the scratch are settled for the next step
        }
    }
}

template <int LM, typename WORD, int LT, int BT>
static hipError_t launch_ccs_pipe_one(const CcsArgs &a, size_t B, hipStream_t s) {
    using P = Plan<LM, LOGR, 1>;
    constexpr size_t LB = ((size_t)P::M + 2 * P::LDS_CPLX) * sizeof(cplx);
    if constexpr (2 * P::NT > 1024 || LB > 160 * 1024) { return hipErrorInvalidValue; } else {
        hipError_t e = set_lds(ccs_pipe_kernel<LM, WORD, LT, BT>, LB); if (e != hipSuccess) return e;
        last_rot_kernel = "ccs_pipe_kernel";
        hipLaunchKernelGGL((ccs_pipe_kernel<LM, WORD, LT, BT>), dim3((unsigned)B), dim3(2 * P::NT), LB, s, a);
        return hipGetLastError();
    }
}
hipError_t launch_ccs_pipe(int logM, int W, const CcsArgs &a, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    MKT_DISPATCH_LOGM(logM, {
        if (W == 64) return launch_ccs_pipe_one<LM, uint64_t, 0, 0>(a, B, s);
        if constexpr (LM == 9 || LM == 10) {
            if (a.l == 3 && a.logB == 8) return launch_ccs_pipe_one<LM, uint32_t, 3, 8>(a, B, s);
            if (a.l == 4 && a.logB == 8) return launch_ccs_pipe_one<LM, uint32_t, 4, 8>(a, B, s);
            if (a.l == 5 && a.logB == 6) return launch_ccs_pipe_one<LM, uint32_t, 5, 6>(a, B, s);
        }
        return launch_ccs_pipe_one<LM, uint32_t, 0, 0>(a, B, s);
    });
    return hipGetLastError();
}

}  // namespace mktd
