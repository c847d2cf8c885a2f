// Blind rotation of the block-binary schemes for gfx950: LMSS (bootstrapping.jl:114-165) with RLWE length 1 or 2 (NP = 2 or 3
// accumulator polynomials) and the phase-1 rows of KMS_block (:599-659).  G rotations that share a key (same party slot, G
// different ciphertexts) per workgroup of G thread groups.
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

#ifndef MKT_BLK_ABLATE_SAMEAT
#define MKT_BLK_ABLATE_SAMEAT 0
#endif
namespace mktd {

// NP = 3 (RLWE length 2, round 3): the one-rotation kernel of that shape (blindrotate_kr_kernel) waits on memory 73 % of the time --
// every workgroup pulls the whole 152 MB key through its L1 (64 GB of L2 -> L1 reads per 1024 gates) with (k+1)^2 l rows per key bit
// and no register left to request them ahead; here each key element is loaded once per G rotations, a digit ahead.  The resident
// tables of such a context are in device point order 2 (ORDER), which only moves where a group publishes / collects its points.
template <int LOGM, typename WORD, int LB, int G, int LT, int BT, int NP = 2>
__global__ __launch_bounds__((G * Plan<LOGM, LOGR>::NT)) __attribute__((amdgpu_waves_per_eu(2, 2)))
void blindrotate_blk_kernel(const RotArgs a, int wg_per_slot) {
    using P = Plan<LOGM, LOGR, 1>;
    constexpr int MO = !(LOGM & 1) ? 1 : -1;     // even sizes keep every legal exchange in the wave (as the one-rotation block kernel)
    constexpr int R = P::R, NT = P::NT, M = P::M, N = 2 * M, W = WordTraits<WORD>::W;
    constexpr int T = G * NT, PTS = M / T;       // multiply-add ownership: PTS = 4 / G stored positions per thread
    static_assert(R == 4 && (G == 1 || G == 2 || G == 4), "G must divide the points per thread");
    // register-lean multiply-add / monomial stages: digit points and monomial rows fetched per rotation as they are used (three or four
    // polynomials; and G = 2, where holding them spilled: 64-bit ring 228 -> 56 B per lane, KMS2partyblock 39.9 -> 34.1 ms, on par with G = 1;
    // 32-bit ring 168 -> 36 B, Blockparam 6.18 -> 5.65 ms per 1024 gates and -- TWO independent workgroups per compute unit -- 75.8 ms per
    // 16 384 gates against 79.2 ms for G = 4, whose one eight-wave workgroup idles all four SIMDs at every barrier)
    constexpr bool LEAN = NP > 2 || G == 2;   // (on the shipped Blockparam kernel, G = 4: 5.27 - 5.35 ms either way)
    // Key elements are re-requested for the NEXT digit as soon as their last multiply-add of this digit is done (per key bit; in the lean
    // stages per key bit and stored position), not after the whole stage: the wait for them was 20 % of a wave's time at G = 2
    // (SQ_WAIT_INST_ANY - SQ_WAIT_INST_LDS, profiles/r05_experiments.txt item 12).  -DMKT_BLK_EARLYK=0: the round-4 order (A/B builds).
#ifndef MKT_BLK_EARLYK
#define MKT_BLK_EARLYK 1
#endif
    constexpr bool EARLYQ = !LEAN && MKT_BLK_EARLYK, EARLYK = LEAN && MKT_BLK_EARLYK;
    constexpr int ORDER = NP > 2 ? MKT_DEVORDER_KR : MKT_DEVORDER;   // context.cpp: the RLWE-length-k contexts keep their tables in order 2
    static_assert(MKT_DEVORDER == 1 && MKT_DEVORDER_KR == 2, "device point orders of the resident tables");
    static_assert(NP >= 2 && NP <= 4, "accumulator polynomials");
    // (NP > 2: the products of a block are published over the staging + digit buffers together; the launcher sizes the region for them)
    // LDS: Psi | roots | FFT staging of group r | published digit transforms [parity][rotation][M] (reused for the products)
    cplx *psi_l = reinterpret_cast<cplx *>(mkt_smem);
    cplx *roots_l = psi_l + M;
    cplx *stg_all = roots_l + M;
    cplx *xbuf = stg_all + (size_t)G * P::LDS_CPLX;
    const int tid = threadIdx.x, grp = tid / NT, t = tid % NT;
    cplx *stg = stg_all + (size_t)grp * P::LDS_CPLX;
    XS xs = make_xs();
    for (int i = tid; i < M; i += T) { psi_l[i] = a.tw.psi[i]; roots_l[i] = a.tw.roots[i]; }
    __syncthreads();

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
        // development only (-DMKT_BLK_ABLATE_SAMEAT=1, WRONG results): every rotation of the workgroup reads rotation 0's mask words, so their monomial rows and
        // mask loads coincide -- the ceiling of what pairing the RLEV rows of one (ciphertext, party) in a workgroup could save (profiles/r06_experiments.txt)
        at_src[r] = a.lwe + (MKT_BLK_ABLATE_SAMEAT ? (gate0 < a.ngates ? gate0 : a.ngates - 1) : gr) * (size_t)a.lwe_stride + (size_t)party * a.n;
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

    int spos[R];                                  // where this thread's points 4t+e sit in a published transform (device point order)
#pragma unroll
    for (int e = 0; e < R; e++) spos[e] = dev_pos(ORDER, t * R + e, NT);
    WORD acc[NP][R][2];                           // rotation `grp`: b and a (a_0, a_1), words (e*NT + t) and (e*NT + t + M)
    if (a.init_mode == 0) {
        const WORD *src = reinterpret_cast<const WORD *>(a.acc_io) + rot * NP * N;
#pragma unroll
        for (int c = 0; c < NP; c++)
#pragma unroll
            for (int e = 0; e < R; e++) { acc[c][e][0] = src[c * N + e * NT + t]; acc[c][e][1] = src[c * N + M + e * NT + t]; }
    } else {                                      // bootstrapping.jl:609-612: b = gvec_lev[row] at X^0, a = 0
#pragma unroll
        for (int c = 0; c < NP; c++)
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
    cplx K[LB][NP][PTS];
    auto load_keys = [&](int kb, int g) {
#pragma unroll
        for (int q = 0; q < LB; q++) {
            const unsigned so_row = (unsigned)((((size_t)(kb * LB + q) * NP * l + (size_t)g) * NP) * M * sizeof(cplx));
#pragma unroll
            for (int c = 0; c < NP; c++)
#pragma unroll
                for (int p = 0; p < PTS; p++) K[q][c][p] = table_load(rs_brk, vo[p], so_row + (unsigned)(c * M * sizeof(cplx)));
        }
    };
    auto load_keys_qp = [&](int kb, int g, int q, int p) {   // one key bit, one stored position
        const unsigned so_row = (unsigned)((((size_t)(kb * LB + q) * NP * l + (size_t)g) * NP) * M * sizeof(cplx));
#pragma unroll
        for (int c = 0; c < NP; c++) K[q][c][p] = table_load(rs_brk, vo[p], so_row + (unsigned)(c * M * sizeof(cplx)));
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
        if (kblk != blk) load_keys(blk, 0);       // the first block, or the block after skipped ones

        cplx tacc[G][LB][NP][PTS];
#pragma unroll
        for (int r = 0; r < G; r++)
#pragma unroll
            for (int q = 0; q < LB; q++)
#pragma unroll
                for (int c = 0; c < NP; c++)
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
            cplx z[1][R];
#pragma unroll
            for (int e = 0; e < R; e++) {                                // :131-140 decompto!, fft.jl:57-63 twist
                const WORD w0 = acc[c2][e][0], w1 = acc[c2][e][1];
                const int d0 = gd.digit(gd.prep(w0), j), d1 = gd.digit(gd.prep(w1), j);
                cplx v; v.re = (double)d0; v.im = (double)(-d1);
                z[0][e] = cmul(v, roots_l[e * NT + t]);
            }
            fft_forward<LOGM, LOGR, 1, MO>(z, psi_l, stg, t, xs.lx);     // fftto!
            cplx *xb = xbuf + (size_t)(g & 1) * G * M;
            cplx zr[G][PTS];
#pragma unroll
            for (int e = 0; e < R; e++) xb[grp * M + spos[e]] = z[0][e];          // stored position of point 4t+e (order 1: e*NT + t)
            __syncthreads();
            if constexpr (!LEAN) {
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
                        for (int p = 0; p < PTS; p++) mv[r][q][p] = table_load(rs_mono, vo[p], so_m);
                    }
                __builtin_amdgcn_sched_barrier(0);
            };
            if (LAST && G == 4 && !LEAN) load_mono();                  // in flight during the multiply-adds where the registers allow
            if constexpr (!LEAN) {
#pragma unroll
                for (int q = 0; q < LB; q++) {
#pragma unroll
                    for (int c = 0; c < NP; c++)
#pragma unroll
                        for (int p = 0; p < PTS; p++)
#pragma unroll
                            for (int r = 0; r < G; r++)                  // :146-154 muladdto!(tacc[q], digit, row); a key bit with atilde = 0 is dropped below
                                tacc[r][q][c][p] = cadd(tacc[r][q][c][p], cmul(zr[r][p], K[q][c][p]));
                    if constexpr (EARLYQ && !LAST) {                     // the key bit's elements are free: the next digit's are requested now
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int p = 0; p < PTS; p++) load_keys_qp(blk, g + 1, q, p);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                // three polynomials: 36 accumulators + 9 key elements leave no room to hold all G digit points -- one rotation at a
                // time, its points read from LDS as they are used (the parity buffers keep them until the step after next)
                if constexpr (EARLYK && !LAST) {
                    // (position, key bit)-major with the G digit points of the position in registers: each group of NP key elements is
                    // re-requested for the next digit as soon as its G * NP multiply-adds are done
#pragma unroll
                    for (int p = 0; p < PTS; p++) {
                        cplx zg[G];
#pragma unroll
                        for (int r = 0; r < G; r++) zg[r] = xb[r * M + tid + p * T];
#pragma unroll
                        for (int q = 0; q < LB; q++) {
#pragma unroll
                            for (int r = 0; r < G; r++)
#pragma unroll
                                for (int c = 0; c < NP; c++) tacc[r][q][c][p] = cadd(tacc[r][q][c][p], cmul(zg[r], K[q][c][p]));
                            __builtin_amdgcn_sched_barrier(0); load_keys_qp(blk, g + 1, q, p); __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                } else {
#pragma unroll
                for (int r = 0; r < G; r++) {
#pragma unroll
                    for (int p = 0; p < PTS; p++) {
                        const cplx zz = xb[r * M + tid + p * T];
#pragma unroll
                        for (int q = 0; q < LB; q++)
#pragma unroll
                            for (int c = 0; c < NP; c++) tacc[r][q][c][p] = cadd(tacc[r][q][c][p], cmul(zz, K[q][c][p]));
                    }
                }
                }
                if (LAST) __syncthreads();                               // the products are published over these buffers
            }
            if (!LAST && !EARLYK && !EARLYQ) { __builtin_amdgcn_sched_barrier(0); load_keys(blk, g + 1); __builtin_amdgcn_sched_barrier(0); }
            if (LAST && G != 4 && !LEAN) { __builtin_amdgcn_sched_barrier(0); load_mono(); }
        };
        using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>; using CL = std::integral_constant<int, NP - 1>;
#pragma unroll 1
        for (int j = 0; j < l; j++) digit_step(C0{}, j, std::false_type{});       // b digits (:131-140, :146-154: b rows first)
        if constexpr (NP >= 3) {
#pragma unroll 1
            for (int j = 0; j < l; j++) digit_step(C1{}, j, std::false_type{});   // a_0 digits
        }
        if constexpr (NP >= 4) {
#pragma unroll 1
            for (int j = 0; j < l; j++) digit_step(std::integral_constant<int, 2>{}, j, std::false_type{});   // a_1 digits
        }
#pragma unroll 1
        for (int j = 0; j < l - 1; j++) digit_step(CL{}, j, std::false_type{});   // a digits (the last a polynomial)
        digit_step(CL{}, l - 1, std::true_type{});

        // :157 / :648 tacc2 += monomial[atilde_q] * tacc[q], q ascending from zero, per rotation; published for the inverse
        // (NP = 3: over the staging and digit buffers together -- every group is past its last forward transform, barrier above)
        cplx *pub = NP > 2 ? stg_all : xbuf;
#pragma unroll
        for (int r = 0; r < G; r++) {
            cplx t2[NP][PTS];
#pragma unroll
            for (int c = 0; c < NP; c++)
#pragma unroll
                for (int p = 0; p < PTS; p++) { t2[c][p].re = 0.0; t2[c][p].im = 0.0; }
            if constexpr (LEAN) {                                        // the rotation's monomial rows now (no register to hold all G * LB of them)
#pragma unroll
                for (int q = 0; q < LB; q++) {
                    const unsigned so_m = (unsigned)((size_t)(ats[r][q] ? ats[r][q] - 1 : 0) * M * sizeof(cplx));
#pragma unroll
                    for (int p = 0; p < PTS; p++) mv[r][q][p] = table_load(rs_mono, vo[p], so_m);
                }
            }
#pragma unroll
            for (int q = 0; q < LB; q++) {
                if (ats[r][q] == 0) continue;
#pragma unroll
                for (int p = 0; p < PTS; p++)
#pragma unroll
                    for (int c = 0; c < NP; c++) t2[c][p] = cadd(t2[c][p], cmul(mv[r][q][p], tacc[r][q][c][p]));
            }
#pragma unroll
            for (int c = 0; c < NP; c++)
#pragma unroll
                for (int p = 0; p < PTS; p++) pub[(size_t)(c * G + r) * M + tid + p * T] = t2[c][p];
        }
        __syncthreads();
        cplx s[NP][R];
#pragma unroll
        for (int c = 0; c < NP; c++)
#pragma unroll
            for (int e = 0; e < R; e++) s[c][e] = pub[(size_t)(c * G + grp) * M + spos[e]];
        __syncthreads();                          // the buffers are free for the next block's first digit
        // requested before the inverse transforms that hide them: the untwist factors, then (younger, so the wait for the
        // factors leaves them in flight) the key elements of the next block's first digit
        cplx ri[R];
#pragma unroll
        for (int e = 0; e < R; e++) ri[e] = a.tw.rootsinv[e * NT + t];
        __builtin_amdgcn_sched_barrier(0);
        if (blk + 1 < nblk) { load_keys(blk + 1, 0); kblk = blk + 1; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NP; c++)              // :162-163 ifftto!, add!
            fft_inverse<LOGM, LOGR, 1, true, MO>(reinterpret_cast<cplx(&)[1][R]>(s[c]), psi_l, stg, t, xs.lx);
#pragma unroll
        for (int e = 0; e < R; e++) {
#pragma unroll
            for (int c = 0; c < NP; c++) {        // fft.jl:76-80 untwist + native
                const cplx v = cmul(s[c][e], ri[e]);
                acc[c][e][0] = (WORD)(acc[c][e][0] + native<WORD>(v.re));
                acc[c][e][1] = (WORD)(acc[c][e][1] + native<WORD>(-v.im));
            }
        }
    }

    if (a.out_mode == 0) {
        if (mine_valid) {
            WORD *dst = reinterpret_cast<WORD *>(a.acc_io) + rot * NP * N;
#pragma unroll
            for (int c = 0; c < NP; c++)
#pragma unroll
                for (int e = 0; e < R; e++) { dst[c * N + e * NT + t] = acc[c][e][0]; dst[c * N + M + e * NT + t] = acc[c][e][1]; }
        }
    } else if constexpr (NP == 2) {                                      // :657 fftto!(tacc, acc): every group runs it (workgroup barriers), valid ones store
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

template <int LM, typename WORD, int LB, int G, int LT, int BT, int NP = 2>
static hipError_t launch_blk_lt(const RotArgs &a, size_t nslots, hipStream_t s) {
    using P = Plan<LM, LOGR, 1>;
    constexpr size_t region = (size_t)G * P::LDS_CPLX + (size_t)2 * G * P::M, products = (size_t)NP * G * P::M;
    constexpr size_t lds_bytes = ((size_t)2 * P::M + (region > products ? region : products)) * sizeof(cplx);
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    hipError_t e = set_lds(blindrotate_blk_kernel<LM, WORD, LB, G, LT, BT, NP>, lds_bytes);
    if (e != hipSuccess) return e;
    last_rot_kernel = "blindrotate_blk_kernel";
    const size_t wg_per_slot = (a.ngates + G - 1) / G;
    hipLaunchKernelGGL((blindrotate_blk_kernel<LM, WORD, LB, G, LT, BT, NP>), dim3((unsigned)(wg_per_slot * nslots)), dim3(G * P::NT), lds_bytes, s, a, (int)wg_per_slot);
    return hipGetLastError();
}

template <int LM, typename WORD, int LB, int G>
static hipError_t launch_blk_g(const RotArgs &a, size_t nslots, hipStream_t s) {
    // the shipped block gadgets as compile-time constants: Blockparam (l = 3, 2^9, 32-bit ring, N = 1024), KMS*partyblock
    // (l = 3, 2^12 at k = 2; params.jl:87-125) on the 64-bit ring, N = 2048
    if constexpr (LB == 3 && LM == 9 && sizeof(WORD) == 4) { if (a.l == 3 && a.logB == 9) return launch_blk_lt<LM, WORD, LB, G, 3, 9>(a, nslots, s); }
    if constexpr (LB == 3 && LM == 10 && sizeof(WORD) == 8) { if (a.l == 3 && a.logB == 12) return launch_blk_lt<LM, WORD, LB, G, 3, 12>(a, nslots, s); }
    return launch_blk_lt<LM, WORD, LB, G, 0, 0>(a, nslots, s);
}

// RLWE length 2 and 3 (three / four accumulator polynomials) on the 32-bit ring, four rotations per workgroup: block length 3 at
// RLWE length 2 (Blockparam's shape, BASELINE.json configs[4]) and the plain CMux (CGGI, block length 1: bootstrapping.jl:32-76 is the
// block loop with one key bit per block) at RLWE length 2 and 3; anything else stays on blindrotate_kr_kernel
template <int LM, typename WORD, int G, int NP>
static hipError_t launch_blk_np(const RotArgs &a, size_t nslots, hipStream_t s) {
    if constexpr (G != 4 || sizeof(WORD) != 4 || G * (1 << LM) / 4 > 1024 || ((size_t)2 + 2 * G + 2 * G) * (1 << LM) * 16 > 160 * 1024) { return hipErrorInvalidValue; } else {
        if (a.blk_len == 1) return launch_blk_lt<LM, WORD, 1, G, 0, 0, NP>(a, nslots, s);
        if constexpr (NP == 3) {
            if (a.blk_len != 3) return hipErrorInvalidValue;
            if constexpr (LM == 9) { if (a.l == 3 && a.logB == 7) return launch_blk_lt<LM, WORD, 3, G, 3, 7, 3>(a, nslots, s); }
            return launch_blk_lt<LM, WORD, 3, G, 0, 0, 3>(a, nslots, s);
        }
        return hipErrorInvalidValue;
    }
}

template <int LM, typename WORD, int G>
static hipError_t launch_blk_lb(const RotArgs &a, size_t nslots, hipStream_t s) {
    if constexpr (G * (1 << LM) / 4 > 1024 || ((size_t)2 + 2 * G + 2 * G) * (1 << LM) * 16 > 160 * 1024) { return hipErrorInvalidValue; } else {
        switch (a.blk_len) {
        case 2: return launch_blk_g<LM, WORD, 2, G>(a, nslots, s);
        case 3: return launch_blk_g<LM, WORD, 3, G>(a, nslots, s);
        case 4: return launch_blk_g<LM, WORD, 4, G>(a, nslots, s);
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
hipError_t MKT_BLK_FN(int logM, int G, int npolys, const RotArgs &a, size_t nslots, hipStream_t s) {
    if (!a.ngates || !nslots) return hipSuccess;
    if (npolys < 2 || npolys > 4) return hipErrorInvalidValue;
    if (npolys > 2) {
        MKT_DISPATCH_LOGM(logM, {
            if (G != 4) return hipErrorInvalidValue;
            if (npolys == 3) return launch_blk_np<LM, MKT_BLK_T, 4, 3>(a, nslots, s);
            return launch_blk_np<LM, MKT_BLK_T, 4, 4>(a, nslots, s);
        });
        return hipSuccess;
    }
    MKT_DISPATCH_LOGM(logM, {
        if (G == 2) return launch_blk_lb<LM, MKT_BLK_T, 2>(a, nslots, s);
        if (G == 4) return launch_blk_lb<LM, MKT_BLK_T, 4>(a, nslots, s);
        return hipErrorInvalidValue;
    });
    return hipSuccess;
}

}  // namespace mktd
