// Blind rotation of the block-binary schemes for gfx950: LMSS (bootstrapping.jl:114-165) and the phase-1 rows of
// KMS_block (:599-659), RLWE length 1.  G rotations that share a key (same party slot, G different ciphertexts) per
// workgroup of G thread groups.
//
// Why its own kernel.  A block of LB key bits multiplies every digit transform into LB * 2 key rows, so the
// multiply-adds are 45 % of a block's flop (22 % in the plain CMux) and each needs its own 16 bytes of key: with one
// rotation per workgroup every thread needs LB * 2 * 4 key elements per digit (96 VGPRs at LB = 3) next to LB * 2 * 4
// transform-domain accumulators (96 more) -- nothing can be requested ahead, every multiply-add waits for its own load,
// and each compute unit pulls every key row in once per resident rotation.  Here the work of a block is split two ways:
//   * transforms (digits -> forward, inverse -> accumulator update): thread group r owns rotation r, exactly as in the
//     plain kernel (accumulator words in registers for the whole rotation);
//   * multiply-adds: thread tid owns the transform-domain points tid + p * T (p < 4 / G) of EVERY rotation of the
//     workgroup, for all LB key bits and both polynomials.  The digit transforms reach it through LDS; each key element
//     it loads serves G rotations, it needs LB * 2 * 4 / G of them per digit -- few enough to be requested before the
//     digit's forward transform -- and the transform-domain accumulators (LB * 2 * 4, the same count as before) now
//     belong to G rotations.  The sums over the key bits of a block (:157 / :648) stay inside one thread, in the
//     reference's order; the products reach the inverse transforms through LDS.
// Arithmetic is the reference's operation for operation (IEEE double, no contraction), so the accumulator is bit-identical
// to blindrotate_k1_kernel<LB> (tests force both).
#include "kernel_common.h"

#include <type_traits>

#pragma clang fp contract(off)

#ifndef MKT_BLK_ROOTS_LDS
#define MKT_BLK_ROOTS_LDS 1   // forward twist factors resident in LDS (their own wait counter: a global load behind the key requests would wait for those)
#endif
#ifndef MKT_BLK_MONO_PF
#define MKT_BLK_MONO_PF 1     // monomial rows requested 1 = before the last digit's multiply-adds (G * LB * 4 / G more live elements there), 0 = after them
#endif
#ifndef MKT_BLK_KPF_INV
#define MKT_BLK_KPF_INV 1     // key elements of the next block's first digit requested 1 = before the inverse transforms of this block, 0 = at the top of the next block
#endif
#ifndef MKT_BLK_KTOP
#define MKT_BLK_KTOP 0        // key elements of a digit requested 1 = at the top of the digit's own step, 0 = at the end of the previous step (software pipeline across the loop edge)
#endif
#ifndef MKT_BLK_ABL
#define MKT_BLK_ABL 0         // development ablations (WRONG results, timing only): 1 = no key-row loads, 2 = no digit exchange through LDS (each group multiplies its own points), 4 = no monomial loads, 8 = every key request hits the same 6 rows (L1 / L2 resident), 16 = the key requests cycle through 8 blocks' rows (2.3 MB: L2 resident)
#endif
#ifndef MKT_BLK_PROBE
#define MKT_BLK_PROBE 0       // development: workgroup 0 prints the s_memtime ticks its first wave spent in each phase of a block
#endif

namespace mktd {

template <int LOGM, typename WORD, int LB, int G, int LT, int BT>
__global__ __launch_bounds__((G * Plan<LOGM, LOGR>::NT)) __attribute__((amdgpu_waves_per_eu(2, 2)))
void blindrotate_blk_kernel(const RotArgs a, int wg_per_slot) {
    using P = Plan<LOGM, LOGR, 1>;
#ifdef MKT_BLK_MO
    constexpr int MO = MKT_BLK_MO;
#else
    constexpr int MO = !(LOGM & 1) ? 1 : -1;     // even sizes keep every legal exchange in the wave (as the one-rotation block kernel)
#endif
    constexpr int R = P::R, NT = P::NT, M = P::M, N = 2 * M, W = WordTraits<WORD>::W;
    constexpr int T = G * NT, PTS = M / T;       // multiply-add ownership: PTS = 4 / G stored positions per thread
    static_assert(R == 4 && (G == 1 || G == 2 || G == 4), "G must divide the points per thread");
    static_assert(MKT_DEVORDER == 1, "a thread's stored positions assume the slot-major device point order");
    // LDS: Psi | roots | FFT staging of group r | published digit transforms [parity][rotation][M] (reused for the products)
    cplx *psi_l = reinterpret_cast<cplx *>(mkt_smem);
    cplx *roots_l = psi_l + M;
    cplx *stg_all = roots_l + (MKT_BLK_ROOTS_LDS ? M : 0);
    cplx *xbuf = stg_all + (size_t)G * P::LDS_CPLX;
    const int tid = threadIdx.x, grp = tid / NT, t = tid % NT;
    cplx *stg = stg_all + (size_t)grp * P::LDS_CPLX;
    XS xs = make_xs();
    for (int i = tid; i < M; i += T) { psi_l[i] = a.tw.psi[i]; if (MKT_BLK_ROOTS_LDS) roots_l[i] = a.tw.roots[i]; }
    __syncthreads();
#if MKT_BLK_PROBE
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = __builtin_amdgcn_s_memtime();
#define BLK_PROBE(K) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pt[K] += now_ - plast; plast = now_; }
#else
#define BLK_PROBE(K)
#endif

    const unsigned bid = blockIdx.x;
    if (a.stagger > 0 && ((bid >> 8) & 1)) {     // de-phase the workgroups that share a compute unit (speed only)
        for (int s = 0; s < a.stagger; s++) __builtin_amdgcn_s_sleep(8);
    }
    // workgroups are dealt slot-major; the G rotations of a workgroup are G consecutive ciphertexts of one slot (one
    // party, one RLEV row: the same key rows).  A ragged tail repeats the last ciphertext and does not store it.
    const int slot = (int)(bid / (unsigned)wg_per_slot);
    const size_t gate0 = (size_t)(bid % (unsigned)wg_per_slot) * G;
    const int party = __builtin_amdgcn_readfirstlane(a.slot_party[slot]), row = __builtin_amdgcn_readfirstlane(a.slot_row[slot]);
    const uint32_t *at_src[G];
#pragma unroll
    for (int r = 0; r < G; r++) {
        const size_t gr = gate0 + r < a.ngates ? gate0 + r : a.ngates - 1;
        at_src[r] = a.lwe + gr * (size_t)a.lwe_stride + (size_t)party * a.n;
    }
    const bool mine_valid = gate0 + grp < a.ngates;
    const size_t my_gate = mine_valid ? gate0 + grp : a.ngates - 1;
    const size_t rot = my_gate * (size_t)a.rows_per_gate + slot;

    const cplx *brk = a.brk + (size_t)party * a.brk_party_stride;
    const __amdgpu_buffer_rsrc_t rs_brk = table_rsrc(brk, (size_t)a.brk_party_stride * sizeof(cplx));
    const __amdgpu_buffer_rsrc_t rs_mono = table_rsrc(a.monomial, (size_t)2 * N * M * sizeof(cplx));
    unsigned vo[PTS];                             // byte offset of this thread's multiply-add positions in a resident row
#pragma unroll
    for (int p = 0; p < PTS; p++) vo[p] = (unsigned)(tid + p * T) * 16u;
    const Gadget<WORD> gd(LT ? LT : a.l, (LT && BT) ? BT : a.logB);
    const int l = LT ? LT : a.l;

    WORD acc[2][R][2];                            // rotation `grp`: b and a, words (e*NT + t) and (e*NT + t + M)
    if (a.init_mode == 0) {
        const WORD *src = reinterpret_cast<const WORD *>(a.acc_io) + rot * 2 * N;
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < R; e++) { acc[c][e][0] = src[c * N + e * NT + t]; acc[c][e][1] = src[c * N + M + e * NT + t]; }
    } else {                                      // bootstrapping.jl:609-612: b = gvec_lev[row] at X^0, a = 0
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < R; e++) { acc[c][e][0] = 0; acc[c][e][1] = 0; }
        if (t == 0) acc[0][0][0] = (WORD)1 << (W - (row + 1) * a.logB_lev);
    }

    const int nblk = a.n / LB;
    const int msbit = 32 - a.logN - 1;
    // The mask words are uniform over the workgroup: scalar buffer loads (their own wait counter, no vector-memory queue
    // slot in front of the key requests), requested a block ahead.
    // (read-only in this kernel: the constant address space makes the compiler select s_load_dword)
    typedef const __attribute__((address_space(4))) uint32_t *cu32p;
    cu32p at_k[G];
#pragma unroll
    for (int r = 0; r < G; r++) at_k[r] = (cu32p)(unsigned long long)at_src[r];
    uint32_t at_next[G][LB];
#pragma unroll
    for (int r = 0; r < G; r++)
#pragma unroll
        for (int q = 0; q < LB; q++) at_next[r][q] = at_k[r][q];

    // this thread's key elements of one digit: all key bits of the block, both polynomials, PTS stored positions -- each
    // serves all G rotations.  Requested a whole forward transform before they are used.
    cplx K[LB][2][PTS];
    auto load_keys = [&](int kb, int g) {
#pragma unroll
        for (int q = 0; q < LB; q++) {
            const unsigned so_row = (MKT_BLK_ABL & 8) ? (unsigned)(q * 2 * M * sizeof(cplx)) : (MKT_BLK_ABL & 16) ? (unsigned)(((size_t)((kb & 7) * LB + q) * 2 * l + (size_t)g) * 2 * M * sizeof(cplx)) : (unsigned)((((size_t)(kb * LB + q) * 2 * l + (size_t)g) * 2) * M * sizeof(cplx));
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int p = 0; p < PTS; p++) {
                    if (MKT_BLK_ABL & 1) { K[q][c][p].re = (double)(q + c) * 0.5; K[q][c][p].im = (double)(p + 1) * 0.25; continue; }
                    K[q][c][p] = table_load(rs_brk, vo[p], so_row + (unsigned)(c * M * sizeof(cplx)));
                }
        }
    };
    int kblk = -1;                                // block whose first digit's key elements are in flight / in K

    for (int blk = 0; blk < nblk; blk++) {
        uint32_t ats[G][LB];
        bool any = false;
#pragma unroll
        for (int r = 0; r < G; r++)
#pragma unroll
            for (int q = 0; q < LB; q++) {
                const uint32_t v = at_next[r][q];
                ats[r][q] = a.pre_switched ? v : divbits<uint32_t>(v, msbit);   // bootstrapping.jl:8
                any |= ats[r][q] != 0;
            }
        {
            const int nb = blk + 1 < nblk ? blk + 1 : blk;
#pragma unroll
            for (int r = 0; r < G; r++)
#pragma unroll
                for (int q = 0; q < LB; q++) at_next[r][q] = at_k[r][nb * LB + q];
        }
        if (!any) continue;                       // :145 / :638 for every rotation of the workgroup: the block adds native(0) = 0
        BLK_PROBE(0)
        if (kblk != blk) load_keys(blk, 0);       // the first block, or the block after skipped ones

        cplx tacc[G][LB][2][PTS];
#pragma unroll
        for (int r = 0; r < G; r++)
#pragma unroll
            for (int q = 0; q < LB; q++)
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int p = 0; p < PTS; p++) { tacc[r][q][c][p].re = 0.0; tacc[r][q][c][p].im = 0.0; }
        cplx mv[G][LB][PTS];                      // monomial rows of the block (:157), requested during the last digit's multiply-adds

        // One digit polynomial: decompose + twist + forward transform by the rotation's own thread group, publish, then
        // every thread multiplies its positions of all G digit transforms into the key elements requested a transform ago
        // and requests the next digit's.  The digit loops stay rolled (b digits, a digits, the last a digit: three bodies);
        // unrolled, the scheduler interleaves the iterations' key requests and spills hundreds of registers.
        auto digit_step = [&](auto c2_, int j, auto last_) {
            constexpr int c2 = decltype(c2_)::value;
            constexpr bool LAST = decltype(last_)::value;
            const int g = c2 * l + j;
            if (MKT_BLK_KTOP && g > 0) { load_keys(blk, g); __builtin_amdgcn_sched_barrier(0); }
            cplx rtw[R];
            if (!MKT_BLK_ROOTS_LDS) {
#pragma unroll
                for (int e = 0; e < R; e++) rtw[e] = a.tw.roots[e * NT + t];
            }
            cplx z[1][R];
#pragma unroll
            for (int e = 0; e < R; e++) {                                // :131-140 decompto!, fft.jl:57-63 twist
                const WORD w0 = acc[c2][e][0], w1 = acc[c2][e][1];
                const int d0 = gd.digit(gd.prep(w0), j), d1 = gd.digit(gd.prep(w1), j);
                cplx v; v.re = (double)d0; v.im = (double)(-d1);
                z[0][e] = cmul(v, MKT_BLK_ROOTS_LDS ? roots_l[e * NT + t] : rtw[e]);
            }
            BLK_PROBE(1)
            fft_forward<LOGM, LOGR, 1, MO>(z, psi_l, stg, t, xs.lx);     // fftto!
            BLK_PROBE(2)
            cplx *xb = xbuf + (size_t)(g & 1) * G * M;
            cplx zr[G][PTS];
            if (MKT_BLK_ABL & 2) {
#pragma unroll
                for (int r = 0; r < G; r++)
#pragma unroll
                    for (int p = 0; p < PTS; p++) zr[r][p] = z[0][(r * PTS + p) % R];
            } else {
#pragma unroll
                for (int e = 0; e < R; e++) xb[grp * M + e * NT + t] = z[0][e];   // stored position of point 4t+e = e*NT + t (device order 1)
                __syncthreads();
                BLK_PROBE(3)
#pragma unroll
                for (int r = 0; r < G; r++)
#pragma unroll
                    for (int p = 0; p < PTS; p++) zr[r][p] = xb[r * M + tid + p * T];
                // the products below are published in the same buffers: every thread must be done with both parities first
                if (LAST) __syncthreads();
            }
            auto load_mono = [&]() {                                     // :157 monomial rows of every rotation and key bit
#pragma unroll
                for (int r = 0; r < G; r++)
#pragma unroll
                    for (int q = 0; q < LB; q++) {
                        const unsigned so_m = (unsigned)((size_t)(ats[r][q] ? ats[r][q] - 1 : 0) * M * sizeof(cplx));
#pragma unroll
                        for (int p = 0; p < PTS; p++) {
                            if (MKT_BLK_ABL & 4) { mv[r][q][p].re = 0.5; mv[r][q][p].im = 0.25 * (double)(so_m & 3u); continue; }
                            mv[r][q][p] = table_load(rs_mono, vo[p], so_m);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            };
            if (LAST && (MKT_BLK_MONO_PF == 1 || G == 4)) load_mono();   // in flight during the multiply-adds
#pragma unroll
            for (int q = 0; q < LB; q++)
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int p = 0; p < PTS; p++)
#pragma unroll
                        for (int r = 0; r < G; r++)                      // :146-154 muladdto!(tacc[q], digit, row); a key bit with atilde = 0 is dropped below
                            tacc[r][q][c][p] = cadd(tacc[r][q][c][p], cmul(zr[r][p], K[q][c][p]));
            if (!LAST && !MKT_BLK_KTOP) { __builtin_amdgcn_sched_barrier(0); load_keys(blk, g + 1); __builtin_amdgcn_sched_barrier(0); }
            if (LAST && !(MKT_BLK_MONO_PF == 1 || G == 4)) { __builtin_amdgcn_sched_barrier(0); load_mono(); }
            BLK_PROBE(4)
        };
        using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
#pragma unroll 1
        for (int j = 0; j < l; j++) digit_step(C0{}, j, std::false_type{});       // b digits (:131-140, :146-154: b rows first)
#pragma unroll 1
        for (int j = 0; j < l - 1; j++) digit_step(C1{}, j, std::false_type{});   // a digits
        digit_step(C1{}, l - 1, std::true_type{});

        // :157 / :648 tacc2 += monomial[atilde_q] * tacc[q], q ascending from zero, per rotation; published for the inverse
#pragma unroll
        for (int r = 0; r < G; r++) {
            cplx t2[2][PTS];
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int p = 0; p < PTS; p++) { t2[c][p].re = 0.0; t2[c][p].im = 0.0; }
#pragma unroll
            for (int q = 0; q < LB; q++) {
                if (ats[r][q] == 0) continue;
#pragma unroll
                for (int p = 0; p < PTS; p++)
#pragma unroll
                    for (int c = 0; c < 2; c++) t2[c][p] = cadd(t2[c][p], cmul(mv[r][q][p], tacc[r][q][c][p]));
            }
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int p = 0; p < PTS; p++) xbuf[(size_t)(c * G + r) * M + tid + p * T] = t2[c][p];
        }
        __syncthreads();
        BLK_PROBE(5)
        cplx s[2][R];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < R; e++) s[c][e] = xbuf[(size_t)(c * G + grp) * M + e * NT + t];
        __syncthreads();                          // the buffers are free for the next block's first digit
        // requested before the inverse transforms that hide them: the untwist factors, then (younger, so the wait for the
        // factors leaves them in flight) the key elements of the next block's first digit
        cplx ri[R];
#pragma unroll
        for (int e = 0; e < R; e++) ri[e] = a.tw.rootsinv[e * NT + t];
        __builtin_amdgcn_sched_barrier(0);
        if (MKT_BLK_KPF_INV && blk + 1 < nblk) { load_keys(blk + 1, 0); kblk = blk + 1; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 2; c++)               // :162-163 ifftto!, add!
            fft_inverse<LOGM, LOGR, 1, true, MO>(reinterpret_cast<cplx(&)[1][R]>(s[c]), psi_l, stg, t, xs.lx);
#pragma unroll
        for (int e = 0; e < R; e++) {
#pragma unroll
            for (int c = 0; c < 2; c++) {         // fft.jl:76-80 untwist + native
                const cplx v = cmul(s[c][e], ri[e]);
                acc[c][e][0] = (WORD)(acc[c][e][0] + native<WORD>(v.re));
                acc[c][e][1] = (WORD)(acc[c][e][1] + native<WORD>(-v.im));
            }
        }
        BLK_PROBE(6)
    }
#if MKT_BLK_PROBE
    if (bid == 0 && tid == 0)
        printf("blk probe G=%d (s_memtime ticks, wave 0 of workgroup 0): head %llu  keys+digits+twist %llu  forward %llu  publish+barrier %llu  read+MACs %llu  monomial+publish+barrier %llu  inverse+native %llu\n",
               G, pt[0], pt[1], pt[2], pt[3], pt[4], pt[5], pt[6]);
#endif

    if (a.out_mode == 0) {
        if (mine_valid) {
            WORD *dst = reinterpret_cast<WORD *>(a.acc_io) + rot * 2 * N;
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int e = 0; e < R; e++) { dst[c * N + e * NT + t] = acc[c][e][0]; dst[c * N + M + e * NT + t] = acc[c][e][1]; }
        }
    } else {                                      // :657 fftto!(tacc, acc): every group runs it (workgroup barriers), valid ones store
#pragma unroll
        for (int c = 0; c < 2; c++) {
            cplx z[1][R];
#pragma unroll
            for (int e = 0; e < R; e++) {
                cplx v; v.re = word_to_f64<WORD>(acc[c][e][0]); v.im = word_to_f64<WORD>((WORD)((WORD)0 - acc[c][e][1]));
                z[0][e] = cmul(v, a.tw.roots[e * NT + t]);
            }
            fft_forward<LOGM, LOGR, 1, MO>(z, psi_l, stg, t, xs.lx);
            if (mine_valid) {
                cplx *o = a.tout + (rot * 2 + c) * M;
#pragma unroll
                for (int e = 0; e < R; e++) o[a.tout_natural ? t * R + e : dev_pos(MKT_DEVORDER, t * R + e, NT)] = z[0][e];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same split with the digit transforms run TWO at a time (shared twiddle reads and barriers, as the plain kernel
// pairs them) and no separate publishing area: a pair of digit transforms is published in the staging buffer of its
// own thread group that the transform's last exchange did not use, the products of the block in the same place.
// LDS: Psi | roots | G x (2 staging buffers of 2 M points).  Key elements: KPF = 2 digits ahead where they fit the
// registers (G = 4: 2 * LB * 2 elements), else the pair's first digit ahead and its second at the top of the
// multiply-adds (G = 2).
// ------------------------------------------------------------------------------------------------
template <int LOGM, typename WORD, int LB, int G, int LT, int BT>
__global__ __launch_bounds__((G * Plan<LOGM, LOGR>::NT)) __attribute__((amdgpu_waves_per_eu(2, 2)))
void blindrotate_blk_pair_kernel(const RotArgs a, int wg_per_slot) {
    using P = Plan<LOGM, LOGR, 2>;
#ifdef MKT_BLK_MO
    constexpr int MO = MKT_BLK_MO;
#else
    constexpr int MO = !(LOGM & 1) ? 1 : -1;
#endif
    using RT = Route<LOGM, LOGR, MO>;
    constexpr int R = P::R, NT = P::NT, M = P::M, N = 2 * M, W = WordTraits<WORD>::W;
    constexpr int T = G * NT, PTS = M / T;
#ifdef MKT_BLKP_KPF
    constexpr int KPF = MKT_BLKP_KPF;
#else
    constexpr int KPF = G == 4 ? 2 : 1;
#endif
#ifdef MKT_BLKP_GATHER2
    constexpr bool GATHER2 = MKT_BLKP_GATHER2;
#else
    constexpr bool GATHER2 = G == 4;
#endif
    static_assert(R == 4 && (G == 2 || G == 4), "G must divide the points per thread");
    static_assert(MKT_DEVORDER == 1, "a thread's stored positions assume the slot-major device point order");
    // the staging buffer a published pair goes to: not the one the forward transform's last LDS exchange used (its readers
    // may still be at it); with a single buffer (a thread group within one wave: in-order LDS) there is no choice to make
    constexpr int PUBOFF = P::NBUF == 2 ? P::buf_off(RT::fwd_last() == 0 ? 1 : 0) : 0;
    cplx *psi_l = reinterpret_cast<cplx *>(mkt_smem);
    cplx *roots_l = psi_l + M;
    cplx *stg_all = roots_l + M;
    const int tid = threadIdx.x, grp = tid / NT, t = tid % NT;
    cplx *stg = stg_all + (size_t)grp * P::LDS_CPLX;
    XS xs = make_xs();
    for (int i = tid; i < M; i += T) { psi_l[i] = a.tw.psi[i]; roots_l[i] = a.tw.roots[i]; }
    __syncthreads();

    const unsigned bid = blockIdx.x;
    if (a.stagger > 0 && ((bid >> 8) & 1)) {
        for (int s = 0; s < a.stagger; s++) __builtin_amdgcn_s_sleep(8);
    }
    const int slot = (int)(bid / (unsigned)wg_per_slot);
    const size_t gate0 = (size_t)(bid % (unsigned)wg_per_slot) * G;
    const int party = __builtin_amdgcn_readfirstlane(a.slot_party[slot]), row = __builtin_amdgcn_readfirstlane(a.slot_row[slot]);
    typedef const __attribute__((address_space(4))) uint32_t *cu32p;
    cu32p at_k[G];
#pragma unroll
    for (int r = 0; r < G; r++) {
        const size_t gr = gate0 + r < a.ngates ? gate0 + r : a.ngates - 1;
        at_k[r] = (cu32p)(unsigned long long)(a.lwe + gr * (size_t)a.lwe_stride + (size_t)party * a.n);
    }
    const bool mine_valid = gate0 + grp < a.ngates;
    const size_t my_gate = mine_valid ? gate0 + grp : a.ngates - 1;
    const size_t rot = my_gate * (size_t)a.rows_per_gate + slot;

    const cplx *brk = a.brk + (size_t)party * a.brk_party_stride;
    const __amdgpu_buffer_rsrc_t rs_brk = table_rsrc(brk, (size_t)a.brk_party_stride * sizeof(cplx));
    const __amdgpu_buffer_rsrc_t rs_mono = table_rsrc(a.monomial, (size_t)2 * N * M * sizeof(cplx));
    unsigned vo[PTS];
#pragma unroll
    for (int p = 0; p < PTS; p++) vo[p] = (unsigned)(tid + p * T) * 16u;
    const Gadget<WORD> gd(LT ? LT : a.l, (LT && BT) ? BT : a.logB);
    const int l = LT ? LT : a.l;

    WORD acc[2][R][2];
    if (a.init_mode == 0) {
        const WORD *src = reinterpret_cast<const WORD *>(a.acc_io) + rot * 2 * N;
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < R; e++) { acc[c][e][0] = src[c * N + e * NT + t]; acc[c][e][1] = src[c * N + M + e * NT + t]; }
    } else {
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < R; e++) { acc[c][e][0] = 0; acc[c][e][1] = 0; }
        if (t == 0) acc[0][0][0] = (WORD)1 << (W - (row + 1) * a.logB_lev);
    }

    const int nblk = a.n / LB;
    const int msbit = 32 - a.logN - 1;
    uint32_t at_next[G][LB];
#pragma unroll
    for (int r = 0; r < G; r++)
#pragma unroll
        for (int q = 0; q < LB; q++) at_next[r][q] = at_k[r][q];

    cplx K[2][LB][2][PTS];                        // key elements of the two digits of a pair
    auto load_keys = [&](int kb, int g, int h) {
#pragma unroll
        for (int q = 0; q < LB; q++) {
            const unsigned so_row = (unsigned)((((size_t)(kb * LB + q) * 2 * l + (size_t)g) * 2) * M * sizeof(cplx));
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int p = 0; p < PTS; p++) K[h][q][c][p] = table_load(rs_brk, vo[p], so_row + (unsigned)(c * M * sizeof(cplx)));
        }
    };
    int kblk = -1;

    for (int blk = 0; blk < nblk; blk++) {
        uint32_t ats[G][LB];
        bool any = false;
#pragma unroll
        for (int r = 0; r < G; r++)
#pragma unroll
            for (int q = 0; q < LB; q++) {
                const uint32_t v = at_next[r][q];
                ats[r][q] = a.pre_switched ? v : divbits<uint32_t>(v, msbit);
                any |= ats[r][q] != 0;
            }
        {
            const int nb = blk + 1 < nblk ? blk + 1 : blk;
#pragma unroll
            for (int r = 0; r < G; r++)
#pragma unroll
                for (int q = 0; q < LB; q++) at_next[r][q] = at_k[r][nb * LB + q];
        }
        if (!any) continue;
        if (kblk != blk) { load_keys(blk, 0, 0); if (KPF == 2) load_keys(blk, 1, 1); }

        cplx tacc[G][LB][2][PTS];
#pragma unroll
        for (int r = 0; r < G; r++)
#pragma unroll
            for (int q = 0; q < LB; q++)
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int p = 0; p < PTS; p++) { tacc[r][q][c][p].re = 0.0; tacc[r][q][c][p].im = 0.0; }
        cplx mv[G][LB][PTS];

        auto gather = [&](cplx (&zr)[G][PTS], int h) {
#pragma unroll
            for (int r = 0; r < G; r++)
#pragma unroll
                for (int p = 0; p < PTS; p++) zr[r][p] = stg_all[(size_t)r * P::LDS_CPLX + PUBOFF + h * M + tid + p * T];
        };
        auto mac = [&](const cplx (&zr)[G][PTS], int h) {
#pragma unroll
            for (int q = 0; q < LB; q++)
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int p = 0; p < PTS; p++)
#pragma unroll
                        for (int r = 0; r < G; r++)                      // :146-154 muladdto!(tacc[q], digit, row)
                            tacc[r][q][c][p] = cadd(tacc[r][q][c][p], cmul(zr[r][p], K[h][q][c][p]));
        };
        // digits g0 = 2i and g0 + 1 of the reference's order (b digits, then a digits)
        auto pair_step = [&](int i, auto last_) {
            constexpr bool LAST = decltype(last_)::value;
            const int g0 = 2 * i;
            cplx z[2][R];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const bool isa = g0 + h >= l;
                const int j = isa ? g0 + h - l : g0 + h;
#pragma unroll
                for (int e = 0; e < R; e++) {                            // :131-140 decompto!, fft.jl:57-63 twist
                    const WORD w0 = isa ? acc[1][e][0] : acc[0][e][0], w1 = isa ? acc[1][e][1] : acc[0][e][1];
                    const int d0 = gd.digit(gd.prep(w0), j), d1 = gd.digit(gd.prep(w1), j);
                    cplx v; v.re = (double)d0; v.im = (double)(-d1);
                    z[h][e] = cmul(v, roots_l[e * NT + t]);
                }
            }
            fft_forward<LOGM, LOGR, 2, MO>(z, psi_l, stg, t, xs.lx);
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int e = 0; e < R; e++) stg[PUBOFF + h * M + e * NT + t] = z[h][e];
            __syncthreads();
            if (KPF == 1) { load_keys(blk, g0 + 1, 1); __builtin_amdgcn_sched_barrier(0); }
            cplx zr0[G][PTS], zr1[G][PTS];
            gather(zr0, 0);
            if (GATHER2) { gather(zr1, 1); __syncthreads(); }            // the staging buffers are free again
            mac(zr0, 0);
            if (!GATHER2) { gather(zr1, 1); __syncthreads(); }
            if (LAST) {                                                  // :157 monomial rows, in flight during the last multiply-adds
#pragma unroll
                for (int r = 0; r < G; r++)
#pragma unroll
                    for (int q = 0; q < LB; q++) {
                        const unsigned so_m = (unsigned)((size_t)(ats[r][q] ? ats[r][q] - 1 : 0) * M * sizeof(cplx));
#pragma unroll
                        for (int p = 0; p < PTS; p++) mv[r][q][p] = table_load(rs_mono, vo[p], so_m);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            mac(zr1, 1);
            if (!LAST) {
                __builtin_amdgcn_sched_barrier(0);
                load_keys(blk, g0 + 2, 0);
                if (KPF == 2) load_keys(blk, g0 + 3, 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#pragma unroll 1
        for (int i = 0; i < l - 1; i++) pair_step(i, std::false_type{});
        pair_step(l - 1, std::true_type{});

        // :157 / :648 tacc2 += monomial[atilde_q] * tacc[q], q ascending from zero; the products go to the rotation's group
#pragma unroll
        for (int r = 0; r < G; r++) {
            cplx t2[2][PTS];
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int p = 0; p < PTS; p++) { t2[c][p].re = 0.0; t2[c][p].im = 0.0; }
#pragma unroll
            for (int q = 0; q < LB; q++) {
                if (ats[r][q] == 0) continue;
#pragma unroll
                for (int p = 0; p < PTS; p++)
#pragma unroll
                    for (int c = 0; c < 2; c++) t2[c][p] = cadd(t2[c][p], cmul(mv[r][q][p], tacc[r][q][c][p]));
            }
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int p = 0; p < PTS; p++) stg_all[(size_t)r * P::LDS_CPLX + PUBOFF + c * M + tid + p * T] = t2[c][p];
        }
        __syncthreads();
        cplx s[2][R];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < R; e++) s[c][e] = stg[PUBOFF + c * M + e * NT + t];
        __syncthreads();
        cplx ri[R];
#pragma unroll
        for (int e = 0; e < R; e++) ri[e] = a.tw.rootsinv[e * NT + t];
        __builtin_amdgcn_sched_barrier(0);
        if (blk + 1 < nblk) { load_keys(blk + 1, 0, 0); if (KPF == 2) load_keys(blk + 1, 1, 1); kblk = blk + 1; }
        __builtin_amdgcn_sched_barrier(0);
        fft_inverse<LOGM, LOGR, 2, true, MO>(s, psi_l, stg, t, xs.lx);   // :162-163 ifftto! (b and a together)
#pragma unroll
        for (int e = 0; e < R; e++) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const cplx v = cmul(s[c][e], ri[e]);
                acc[c][e][0] = (WORD)(acc[c][e][0] + native<WORD>(v.re));
                acc[c][e][1] = (WORD)(acc[c][e][1] + native<WORD>(-v.im));
            }
        }
    }

    if (a.out_mode == 0) {
        if (mine_valid) {
            WORD *dst = reinterpret_cast<WORD *>(a.acc_io) + rot * 2 * N;
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int e = 0; e < R; e++) { dst[c * N + e * NT + t] = acc[c][e][0]; dst[c * N + M + e * NT + t] = acc[c][e][1]; }
        }
    } else {                                      // :657 fftto!(tacc, acc), both polynomials together
        cplx z[2][R];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < R; e++) {
                cplx v; v.re = word_to_f64<WORD>(acc[c][e][0]); v.im = word_to_f64<WORD>((WORD)((WORD)0 - acc[c][e][1]));
                z[c][e] = cmul(v, roots_l[e * NT + t]);
            }
        fft_forward<LOGM, LOGR, 2, MO>(z, psi_l, stg, t, xs.lx);
        if (mine_valid) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                cplx *o = a.tout + (rot * 2 + c) * M;
#pragma unroll
                for (int e = 0; e < R; e++) o[a.tout_natural ? t * R + e : dev_pos(MKT_DEVORDER, t * R + e, NT)] = z[c][e];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// One rotation per workgroup, ONE THREAD GROUP PER POLYNOMIAL of the accumulator (b, a).  Group c owns polynomial c:
// its words live in that group's registers for the whole rotation, the group decomposes and transforms its l digits
// (both groups at the same time), accumulates the LB key bits' products that end in polynomial c (:146-154: the rows'
// own polynomial c), sums them over the key bits with the monomials (:157) and runs the inverse transform and the
// update of polynomial c (:162-163) -- all without leaving the group.  Only the digit transforms cross: the b digits
// through a two-slot exchange buffer as they are produced, the a digits parked in LDS until the b digits are done,
// because every sum runs over the digits in the reference's order (b digits first).
// Per thread: LB * 4 transform-domain accumulators instead of LB * 2 * 4, one polynomial's words, LB * 4 key elements per
// digit -- room to request every digit's key elements a whole step ahead and to keep the twist factors in registers.
// LDS: Psi | 2 x FFT staging | b-digit exchange [2][M] | a-digit store [l][M].
// ------------------------------------------------------------------------------------------------
template <int LOGM, typename WORD, int LB, int LT, int BT>
__global__ __launch_bounds__((2 * Plan<LOGM, LOGR>::NT)) __attribute__((amdgpu_waves_per_eu(2, 2)))
void blindrotate_blk_split_kernel(const RotArgs a) {
    using P = Plan<LOGM, LOGR, 1>;
#ifdef MKT_BLK_MO
    constexpr int MO = MKT_BLK_MO;
#else
    constexpr int MO = !(LOGM & 1) ? 1 : -1;
#endif
    constexpr int R = P::R, NT = P::NT, M = P::M, N = 2 * M, W = WordTraits<WORD>::W;
    static_assert(MKT_DEVORDER == 1, "a thread's stored positions assume the slot-major device point order");
    cplx *psi_l = reinterpret_cast<cplx *>(mkt_smem);
    cplx *stg_all = psi_l + M;
    cplx *bx = stg_all + (size_t)2 * P::LDS_CPLX;
    cplx *ax = bx + (size_t)2 * M;
    const int tid = threadIdx.x, grp = tid / NT, t = tid % NT;     // grp = the polynomial this thread's group owns
    cplx *stg = stg_all + (size_t)grp * P::LDS_CPLX;
    XS xs = make_xs();
    for (int i = tid; i < M; i += 2 * NT) psi_l[i] = a.tw.psi[i];
    __syncthreads();

    const unsigned bid = blockIdx.x + a.block0;
    if (a.stagger > 0 && ((bid >> 8) & 1)) {
        for (int s = 0; s < a.stagger; s++) __builtin_amdgcn_s_sleep(8);
    }
    const size_t gate = bid % (size_t)a.ngates;
    const int slot = (int)(bid / (size_t)a.ngates);
    const size_t rot = gate * (size_t)a.rows_per_gate + slot;
    const int party = __builtin_amdgcn_readfirstlane(a.slot_party[slot]), row = __builtin_amdgcn_readfirstlane(a.slot_row[slot]);
    typedef const __attribute__((address_space(4))) uint32_t *cu32p;
    const cu32p at_k = (cu32p)(unsigned long long)(a.lwe + gate * (size_t)a.lwe_stride + (size_t)party * a.n);
    const cplx *brk = a.brk + (size_t)party * a.brk_party_stride;
    const __amdgpu_buffer_rsrc_t rs_brk = table_rsrc(brk, (size_t)a.brk_party_stride * sizeof(cplx));
    const __amdgpu_buffer_rsrc_t rs_mono = table_rsrc(a.monomial, (size_t)2 * N * M * sizeof(cplx));
    unsigned vo[R];                               // byte offsets of this thread's points in a resident row: point 4t+e at e*NT + t
#pragma unroll
    for (int e = 0; e < R; e++) vo[e] = (unsigned)(e * NT + t) * 16u;
    const Gadget<WORD> gd(LT ? LT : a.l, (LT && BT) ? BT : a.logB);
    const int l = LT ? LT : a.l;
    cplx rt[R], ri[R];                            // twist / untwist factors of this thread's points
#pragma unroll
    for (int e = 0; e < R; e++) { rt[e] = a.tw.roots[e * NT + t]; ri[e] = a.tw.rootsinv[e * NT + t]; }

    WORD acc[R][2];                               // polynomial grp: words (e*NT + t) and (e*NT + t + M)
    if (a.init_mode == 0) {
        const WORD *src = reinterpret_cast<const WORD *>(a.acc_io) + rot * 2 * N + (size_t)grp * N;
#pragma unroll
        for (int e = 0; e < R; e++) { acc[e][0] = src[e * NT + t]; acc[e][1] = src[M + e * NT + t]; }
    } else {                                      // bootstrapping.jl:609-612
#pragma unroll
        for (int e = 0; e < R; e++) { acc[e][0] = 0; acc[e][1] = 0; }
        if (grp == 0 && t == 0) acc[0][0] = (WORD)1 << (W - (row + 1) * a.logB_lev);
    }

    const int nblk = a.n / LB;
    const int msbit = 32 - a.logN - 1;
    uint32_t at_next[LB];
#pragma unroll
    for (int q = 0; q < LB; q++) at_next[q] = at_k[q];

    cplx K[LB][R];                                // key elements of one digit: rows (key bit q, digit g), polynomial grp
    auto load_keys = [&](int kb, int g) {
#pragma unroll
        for (int q = 0; q < LB; q++) {
            const unsigned so_row = (unsigned)(((((size_t)(kb * LB + q) * 2 * l + (size_t)g) * 2) + (size_t)grp) * M * sizeof(cplx));
#pragma unroll
            for (int e = 0; e < R; e++) K[q][e] = table_load(rs_brk, vo[e], so_row);
        }
    };
    int kblk = -1;

    for (int blk = 0; blk < nblk; blk++) {
        uint32_t ats[LB];
        bool any = false;
#pragma unroll
        for (int q = 0; q < LB; q++) {
            const uint32_t v = at_next[q];
            ats[q] = a.pre_switched ? v : divbits<uint32_t>(v, msbit);           // bootstrapping.jl:8
            any |= ats[q] != 0;
        }
        {
            const int nb = blk + 1 < nblk ? blk + 1 : blk;
#pragma unroll
            for (int q = 0; q < LB; q++) at_next[q] = at_k[nb * LB + q];
        }
        if (!any) continue;                       // :145 / :638
        if (kblk != blk) load_keys(blk, 0);

        cplx tacc[LB][R];
#pragma unroll
        for (int q = 0; q < LB; q++)
#pragma unroll
            for (int e = 0; e < R; e++) { tacc[q][e].re = 0.0; tacc[q][e].im = 0.0; }
        auto mac = [&](const cplx (&zd)[R]) {
#pragma unroll
            for (int q = 0; q < LB; q++)
#pragma unroll
                for (int e = 0; e < R; e++) tacc[q][e] = cadd(tacc[q][e], cmul(zd[e], K[q][e]));   // :146-154 muladdto!
        };

        // rounds: both groups transform digit j of their own polynomial; the b digit is consumed at once (g = j), the a
        // digit is parked (g = l + j comes after every b digit)
#pragma unroll 1
        for (int j = 0; j < l; j++) {
            cplx z[1][R];
#pragma unroll
            for (int e = 0; e < R; e++) {                                // :131-140 decompto!, fft.jl:57-63 twist
                const int d0 = gd.digit(gd.prep(acc[e][0]), j), d1 = gd.digit(gd.prep(acc[e][1]), j);
                cplx v; v.re = (double)d0; v.im = (double)(-d1);
                z[0][e] = cmul(v, rt[e]);
            }
            fft_forward<LOGM, LOGR, 1, MO>(z, psi_l, stg, t, xs.lx);
            cplx *dst = grp == 0 ? bx + (size_t)(j & 1) * M : ax + (size_t)j * M;
#pragma unroll
            for (int e = 0; e < R; e++) dst[e * NT + t] = z[0][e];
            __syncthreads();
            cplx zb[R];
#pragma unroll
            for (int e = 0; e < R; e++) zb[e] = bx[(size_t)(j & 1) * M + e * NT + t];
            mac(zb);
            __builtin_amdgcn_sched_barrier(0);
            load_keys(blk, j + 1);                                       // the next b digit, or the first a digit (g = l)
            __builtin_amdgcn_sched_barrier(0);
        }
        cplx mv[LB][R];
#pragma unroll 1
        for (int j = 0; j < l - 1; j++) {                                // a digits 0 .. l-2 from the store
            cplx za[R];
#pragma unroll
            for (int e = 0; e < R; e++) za[e] = ax[(size_t)j * M + e * NT + t];
            mac(za);
            __builtin_amdgcn_sched_barrier(0);
            load_keys(blk, l + j + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            cplx za[R];
#pragma unroll
            for (int e = 0; e < R; e++) za[e] = ax[(size_t)(l - 1) * M + e * NT + t];
            __syncthreads();                      // every read of the exchange buffers and of the store is done: the next block may refill them
#pragma unroll
            for (int q = 0; q < LB; q++) {                               // :157 monomial rows, in flight during the last multiply-adds
                const unsigned so_m = (unsigned)((size_t)(ats[q] ? ats[q] - 1 : 0) * M * sizeof(cplx));
#pragma unroll
                for (int e = 0; e < R; e++) mv[q][e] = table_load(rs_mono, vo[e], so_m);
            }
            __builtin_amdgcn_sched_barrier(0);
            mac(za);
        }
        // :157 / :648 tacc2 += monomial[atilde_q] * tacc[q], q ascending from zero -- already in the inverse transform's ownership
        cplx s[1][R];
#pragma unroll
        for (int e = 0; e < R; e++) { s[0][e].re = 0.0; s[0][e].im = 0.0; }
#pragma unroll
        for (int q = 0; q < LB; q++) {
            if (ats[q] == 0) continue;
#pragma unroll
            for (int e = 0; e < R; e++) s[0][e] = cadd(s[0][e], cmul(mv[q][e], tacc[q][e]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (blk + 1 < nblk) { load_keys(blk + 1, 0); kblk = blk + 1; }   // hidden behind the inverse transform
        __builtin_amdgcn_sched_barrier(0);
        fft_inverse<LOGM, LOGR, 1, true, MO>(s, psi_l, stg, t, xs.lx);   // :162-163 ifftto!, add!
#pragma unroll
        for (int e = 0; e < R; e++) {
            const cplx v = cmul(s[0][e], ri[e]);                         // fft.jl:76-80 untwist + native
            acc[e][0] = (WORD)(acc[e][0] + native<WORD>(v.re));
            acc[e][1] = (WORD)(acc[e][1] + native<WORD>(-v.im));
        }
    }

    if (a.out_mode == 0) {
        WORD *dst = reinterpret_cast<WORD *>(a.acc_io) + rot * 2 * N + (size_t)grp * N;
#pragma unroll
        for (int e = 0; e < R; e++) { dst[e * NT + t] = acc[e][0]; dst[M + e * NT + t] = acc[e][1]; }
    } else {                                      // :657 fftto!(tacc, acc): each group its own polynomial
        cplx z[1][R];
#pragma unroll
        for (int e = 0; e < R; e++) {
            cplx v; v.re = word_to_f64<WORD>(acc[e][0]); v.im = word_to_f64<WORD>((WORD)((WORD)0 - acc[e][1]));
            z[0][e] = cmul(v, rt[e]);
        }
        fft_forward<LOGM, LOGR, 1, MO>(z, psi_l, stg, t, xs.lx);
        cplx *o = a.tout + (rot * 2 + grp) * M;
#pragma unroll
        for (int e = 0; e < R; e++) o[a.tout_natural ? t * R + e : dev_pos(MKT_DEVORDER, t * R + e, NT)] = z[0][e];
    }
}

template <int LM, typename WORD, int LB, int LT, int BT>
static hipError_t launch_blk_split_lt(const RotArgs &a, size_t nrot, hipStream_t s) {
    using P = Plan<LM, LOGR, 1>;
    if (2 * P::NT > 1024) return hipErrorInvalidValue;
    const size_t lds_bytes = ((size_t)P::M + (size_t)2 * P::LDS_CPLX + (size_t)2 * P::M + (size_t)a.l * P::M) * sizeof(cplx);
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = set_lds(blindrotate_blk_split_kernel<LM, WORD, LB, LT, BT>, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((blindrotate_blk_split_kernel<LM, WORD, LB, LT, BT>), dim3((unsigned)nrot), dim3(2 * P::NT), lds_bytes, s, a);
    return hipGetLastError();
}
template <int LM, typename WORD, int LB>
static hipError_t launch_blk_split(const RotArgs &a, size_t nrot, hipStream_t s) {
    if constexpr (2 * (1 << LM) / 4 > 1024) { return hipErrorInvalidValue; } else {
        if constexpr (LB == 3 && LM == 9 && sizeof(WORD) == 4) { if (a.l == 3 && a.logB == 9) return launch_blk_split_lt<LM, WORD, LB, 3, 9>(a, nrot, s); }
        if constexpr (LB == 3 && LM == 10 && sizeof(WORD) == 8) { if (a.l == 3 && a.logB == 12) return launch_blk_split_lt<LM, WORD, LB, 3, 12>(a, nrot, s); }
        return launch_blk_split_lt<LM, WORD, LB, 0, 0>(a, nrot, s);
    }
}

template <int LM, typename WORD, int LB, int G, int LT, int BT>
static hipError_t launch_blk_pair_lt(const RotArgs &a, size_t nslots, hipStream_t s) {
    using P = Plan<LM, LOGR, 2>;
    constexpr size_t lds_bytes = ((size_t)2 * P::M + (size_t)G * P::LDS_CPLX) * sizeof(cplx);
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    hipError_t e = set_lds(blindrotate_blk_pair_kernel<LM, WORD, LB, G, LT, BT>, lds_bytes);
    if (e != hipSuccess) return e;
    const size_t wg_per_slot = (a.ngates + G - 1) / G;
    hipLaunchKernelGGL((blindrotate_blk_pair_kernel<LM, WORD, LB, G, LT, BT>), dim3((unsigned)(wg_per_slot * nslots)), dim3(G * P::NT), lds_bytes, s, a, (int)wg_per_slot);
    return hipGetLastError();
}

template <int LM, typename WORD, int LB, int G, int LT, int BT>
static hipError_t launch_blk_lt(const RotArgs &a, size_t nslots, hipStream_t s) {
    using P = Plan<LM, LOGR, 1>;
    constexpr size_t lds_bytes = ((size_t)(MKT_BLK_ROOTS_LDS ? 2 : 1) * P::M + (size_t)G * P::LDS_CPLX + (size_t)2 * G * P::M) * sizeof(cplx);
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    hipError_t e = set_lds(blindrotate_blk_kernel<LM, WORD, LB, G, LT, BT>, lds_bytes);
    if (e != hipSuccess) return e;
    const size_t wg_per_slot = (a.ngates + G - 1) / G;
    hipLaunchKernelGGL((blindrotate_blk_kernel<LM, WORD, LB, G, LT, BT>), dim3((unsigned)(wg_per_slot * nslots)), dim3(G * P::NT), lds_bytes, s, a, (int)wg_per_slot);
    return hipGetLastError();
}

template <int LM, typename WORD, int LB, int G>
static hipError_t launch_blk_g(const RotArgs &a, size_t nslots, bool pair, hipStream_t s) {
    if (pair) {
        if constexpr (LB == 3 && LM == 9 && sizeof(WORD) == 4) { if (a.l == 3 && a.logB == 9) return launch_blk_pair_lt<LM, WORD, LB, G, 3, 9>(a, nslots, s); }
        if constexpr (LB == 3 && LM == 10 && sizeof(WORD) == 8) { if (a.l == 3 && a.logB == 12) return launch_blk_pair_lt<LM, WORD, LB, G, 3, 12>(a, nslots, s); }
        return launch_blk_pair_lt<LM, WORD, LB, G, 0, 0>(a, nslots, s);
    }
    // the shipped block gadgets as compile-time constants: Blockparam (l = 3, 2^9, 32-bit ring, N = 1024), KMS*partyblock
    // (l = 3, 2^12 at k = 2; params.jl:87-125) on the 64-bit ring, N = 2048
    if constexpr (LB == 3 && LM == 9 && sizeof(WORD) == 4) { if (a.l == 3 && a.logB == 9) return launch_blk_lt<LM, WORD, LB, G, 3, 9>(a, nslots, s); }
    if constexpr (LB == 3 && LM == 10 && sizeof(WORD) == 8) { if (a.l == 3 && a.logB == 12) return launch_blk_lt<LM, WORD, LB, G, 3, 12>(a, nslots, s); }
    return launch_blk_lt<LM, WORD, LB, G, 0, 0>(a, nslots, s);
}

template <int LM, typename WORD, int G>
static hipError_t launch_blk_lb(const RotArgs &a, size_t nslots, bool pair, hipStream_t s) {
    if constexpr (G * (1 << LM) / 4 > 1024 || ((size_t)2 + 2 * G + 2 * G) * (1 << LM) * 16 > 160 * 1024) { return hipErrorInvalidValue; } else {
        switch (a.blk_len) {
        case 2: return launch_blk_g<LM, WORD, 2, G>(a, nslots, pair, s);
        case 3: return launch_blk_g<LM, WORD, 3, G>(a, nslots, pair, s);
        case 4: return launch_blk_g<LM, WORD, 4, G>(a, nslots, pair, s);
        default: return hipErrorInvalidValue;
        }
    }
}

#ifndef MKT_BLK_WORD
#error "compile with -DMKT_BLK_WORD=32 or 64"
#endif
#if MKT_BLK_WORD == 32
// G rotations per workgroup (2 or 4) supported at this size?  (workgroup of G * M / 4 threads, LDS budget)
bool blockg_supported(int logM, int G) {
    if (G == 21) return logM >= 4 && logM <= 11;       // the launcher checks the LDS budget for the gadget length at hand
    G %= 10;
    if (logM < 4 || logM > 11 || (G != 2 && G != 4)) return false;
    const size_t M = (size_t)1 << logM;
    return G * M / 4 <= 1024 && (2 + 4 * (size_t)G) * M * 16 <= 160 * 1024;
}
#endif
#if MKT_BLK_WORD == 32
#define MKT_BLK_T uint32_t
#define MKT_BLK_FN launch_rot_blockg_u32
#else
#define MKT_BLK_T uint64_t
#define MKT_BLK_FN launch_rot_blockg_u64
#endif
// G = 2 / 4: rotations per workgroup, single digit transforms; G = 12 / 14: the same with paired transforms
hipError_t MKT_BLK_FN(int logM, int G, const RotArgs &a, size_t nslots, hipStream_t s) {
    if (!a.ngates || !nslots) return hipSuccess;
    if (G == 21) {                        // one rotation per workgroup, one thread group per polynomial
        const size_t nrot = a.ngates * nslots;
        MKT_DISPATCH_LOGM(logM, {
            switch (a.blk_len) {
            case 2: return launch_blk_split<LM, MKT_BLK_T, 2>(a, nrot, s);
            case 3: return launch_blk_split<LM, MKT_BLK_T, 3>(a, nrot, s);
            case 4: return launch_blk_split<LM, MKT_BLK_T, 4>(a, nrot, s);
            default: return hipErrorInvalidValue;
            }
        });
        return hipSuccess;
    }
    const bool pair = G >= 10;
    G %= 10;
    MKT_DISPATCH_LOGM(logM, {
        if (G == 2) return launch_blk_lb<LM, MKT_BLK_T, 2>(a, nslots, pair, s);
        if (G == 4) return launch_blk_lb<LM, MKT_BLK_T, 4>(a, nslots, pair, s);
        return hipErrorInvalidValue;
    });
    return hipSuccess;
}

}  // namespace mktd
