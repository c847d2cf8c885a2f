// Device primitives of the F64REF negacyclic transform for gfx950 (wave64, LDS-staged).
//
// Reference semantics (operation for operation, IEEE double, no FMA contraction):
//   forward  fftto!  src/ring/fft.jl:57-63  + fft!  :105-155 (Cooley-Tukey, bit-reversed output)
//   inverse  ifftto! src/ring/fft.jl:74-81  + ifft! :159-209 (Gentleman-Sande) + native arithmetic.jl:1-9
//   complex product  (xr*yr - xi*yi, xr*yi + xi*yr)  (Julia Base complex `*`)
//
// Mapping.  One transform of M = 2^LOGM complex points is owned by NT = M / R threads, each holding
// R = 2^LOGR points in registers.  The radix-2 butterfly network of the reference is executed in
// "passes" of LOGR consecutive stages that are local to a thread; between passes the points are
// re-distributed through LDS.  In a pass with window low bit `lo`, thread t holds in slot e the point
//      idx(t, e, lo) = ((t >> lo) << (lo + LOGR)) | (e << lo) | (t & (2^lo - 1)).
// The first forward pass (window at the top bits) holds idx = e*NT + t, the last holds idx = t*R + e;
// the inverse runs the same windows backwards, so a (forward -> pointwise -> inverse) chain needs no
// re-distribution besides the in-transform exchanges, and the 2 coefficients (idx, idx + M) folded
// into point idx stay with one thread for the whole blind rotation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace mktd {

struct __attribute__((aligned(16))) cplx { double re, im; };

__device__ __forceinline__ cplx cmul(const cplx x, const cplx y) {
    const double a = x.re * y.re, b = x.im * y.im, c = x.re * y.im, d = x.im * y.re;
    cplx r; r.re = a - b; r.im = c + d; return r;
}
// x * conj(w), bit-identical to cmul(x, (w.re, -w.im)): a - (-b) == a + b and (-c) + d == d - c in IEEE-754
__device__ __forceinline__ cplx cmul_conj(const cplx x, const cplx w) {
    const double a = x.re * w.re, b = x.im * w.im, c = x.re * w.im, d = x.im * w.re;
    cplx r; r.re = a + b; r.im = d - c; return r;
}
__device__ __forceinline__ cplx cadd(const cplx x, const cplx y) { cplx r; r.re = x.re + y.re; r.im = x.im + y.im; return r; }
__device__ __forceinline__ cplx csub(const cplx x, const cplx y) { cplx r; r.re = x.re - y.re; r.im = x.im - y.im; return r; }

// The first two stages of the forward network (the last two of the inverse) only ever see three twiddles:
//   Psi[1] = exp(-i pi/2) = (eps, -1) with eps = 0x1.452821e638d01p-257 (the cosine of the 256-bit pi / 2), Psi[2] = (c, -c), Psi[3] = (-c, -c).
// A product by -1 is exact and x * (-c) == -(x * c) bit for bit, so these butterflies need 2 multiplications instead of 4 and
// give the reference's bits (a - (-y) == a + y and (-y) + d == d - y in IEEE-754, signed zeros included).  The host checks that
// the tables have this shape before they are uploaded (context.cpp: check_twiddle_shape); the reference's always do.
#ifndef MKT_FFT_SPECIAL
#define MKT_FFT_SPECIAL 1
#endif
#ifndef MKT_DPP_PAIR
#define MKT_DPP_PAIR 1        // pairwise lane exchanges along lane bits 0..3 as two DPP reads (see swap_pair)
#endif
#ifndef MKT_PERMLANE_SWAP
#define MKT_PERMLANE_SWAP 1   // gfx950 v_permlane16_swap / v_permlane32_swap for exchanges along lane bits 4 and 5
#endif

// Schedule of a transform: NPASS passes over 2-bit (LOGR-bit) windows [lo, lo + LOGR) of the point index, from the top
// window down to [0, LOGR).  When LOGR does not divide LOGM one pass overlaps its predecessor and performs the stages
// of its fresh bits only.  For 4 points per thread and odd LOGM that single-stage pass sits at window OVL = 4
// (sequence ... 7, 5, 4, 2, 0): the one-bit exchange in front of it then trades thread bit 4 = lane bit 4, which
// v_permlane16_swap does in one instruction per dword (at window 0 it is lane bit 0: a DPP move plus three selects).
template <int LOGM, int LOGR, int NB = 1>
struct Plan {
    static constexpr int M = 1 << LOGM, R = 1 << LOGR, NT = M / R;
    static constexpr int NPASS = (LOGM + LOGR - 1) / LOGR;
    static_assert(LOGM >= LOGR, "transform smaller than a thread's share");
    static constexpr int OVL = (MKT_PERMLANE_SWAP && LOGR == 2 && (LOGM & 1) && LOGM >= 7) ? 4 : 0;
    static constexpr int PA = OVL ? (LOGM - OVL - 1) / LOGR : NPASS - 1;        // index of the single-stage / last pass
    __host__ __device__ static constexpr int lo(int p) {
        if (OVL) return p < PA ? LOGM - (p + 1) * LOGR : (p == PA ? OVL : (NPASS - 1 - p) * LOGR);
        return (LOGM - (p + 1) * LOGR) < 0 ? 0 : (LOGM - (p + 1) * LOGR);
    }
    __host__ __device__ static constexpr int nst(int p) {
        if (OVL) return p == PA ? 1 : LOGR;
        return p < NPASS - 1 ? LOGR : LOGM - (NPASS - 1) * LOGR;
    }
    // highest stage bit of pass p
    __host__ __device__ static constexpr int hib(int p) {
        if (OVL) return p == PA ? OVL : lo(p) + LOGR - 1;
        return p < NPASS - 1 ? lo(p) + LOGR - 1 : nst(p) - 1;
    }
    // LDS staging: NB transforms side by side.  Workgroups of more than one wave alternate between two buffers (one
    // barrier per exchange); a single wave's LDS operations complete in order, so one buffer and no barrier do
    static constexpr int BUF = NB * M;
    static constexpr int NBUF = NT > 64 ? 2 : 1;
    __host__ __device__ static constexpr int buf_off(int pass) { return NBUF == 2 ? (pass & 1) * BUF : 0; }
    static constexpr int LDS_CPLX = NBUF * BUF;
    static constexpr size_t LDS_BYTES = (size_t)LDS_CPLX * sizeof(cplx);
};

template <int LOGR>
__device__ __forceinline__ int pt_index(int t, int e, int lo) {
    return ((t >> lo) << (lo + LOGR)) | (e << lo) | (t & ((1 << lo) - 1));
}
// XOR swizzle of the staging slot: conflict-free ds_write_b128 / ds_read_b128 for every exchange pattern of
// the 4- and 8-points-per-thread schedules at M = 128 .. 2048 (found and verified by tools/lds_swizzle_search.py,
// which simulates the gfx950 b128 lane groups and bank widths)
template <int LOGR>
__device__ __forceinline__ int lds_pos(int idx) {
    if (LOGR == 2) return idx ^ ((idx >> 1) & 8) ^ ((idx >> 2) & 15);
    return idx ^ ((idx >> 3) & 15);   // 8 points per thread
}

// ---- point re-distribution between passes -------------------------------------------------------------------
// Through LDS (any pattern): two staging buffers, the exchange after pass p uses buffer p & 1, one s_barrier per
// exchange.  A wave can only reach its NEXT write into a buffer after the barrier of an exchange in between, which
// every wave passes after finishing its reads from that buffer; where two consecutive LDS exchanges (the last of one
// transform, the first of the next) would use the same buffer, the transform starts with a guard barrier (Route).
template <int LOGM, int LOGR, int NB>
__device__ __forceinline__ void exchange_lds(cplx (&z)[NB][1 << LOGR], cplx *buf, int t, int lo_from, int lo_to) {
    constexpr int R = 1 << LOGR, M = 1 << LOGM;
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int e = 0; e < R; e++) buf[b * M + lds_pos<LOGR>(pt_index<LOGR>(t, e, lo_from))] = z[b][e];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int e = 0; e < R; e++) z[b][e] = buf[b * M + lds_pos<LOGR>(pt_index<LOGR>(t, e, lo_to))];
}

// Inside a wave (LOGR == 2): when the thread bits that trade places with the slot bits are lane bits, the exchange
// is a 4x4 transpose among 4 lanes (2 rounds of pairwise swaps with lane ^ d) or, for the odd last window, one
// pairwise swap -- done with DPP / ds_swizzle register moves: no LDS store traffic (the busiest resource of the
// blind rotation: ~25 % of a wave's time sat in ds_write + wait) and no barrier.
struct LaneX { int lane; int m4; };
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ LaneX make_lanex() {
    LaneX x; x.lane = (int)(threadIdx.x & 63);
    x.m4 = dpp_mov<0x124>(x.lane) == (x.lane ^ 4);   // does row_ror:4 deliver lane^4 to this lane? (else row_ror:12 does)
    return x;
}
template <int D>
__device__ __forceinline__ int xor_lane(int v, const LaneX &lx) {
    if constexpr (D == 1) return dpp_mov<0xB1>(v);                 // quad_perm [1,0,3,2]
    else if constexpr (D == 2) return dpp_mov<0x4E>(v);            // quad_perm [2,3,0,1]
    else if constexpr (D == 4) { const int a = dpp_mov<0x124>(v), b = dpp_mov<0x12C>(v); return lx.m4 ? a : b; }   // row_ror:4 / :12
    else if constexpr (D == 8) return dpp_mov<0x128>(v);           // row_ror:8
    else if constexpr (D == 16) return __builtin_amdgcn_ds_swizzle(v, (16 << 10) | 0x1F);   // BitMode xor 16
    else return __shfl_xor(v, 32);
}
template <int D>
__device__ __forceinline__ void swap_pair(cplx &x0, cplx &x1, bool b, const LaneX &lx) {
    // lane with b = 0 gives x1 and receives the partner's x0 into x1; lane with b = 1 gives x0, receives into x0
    int a0[4], a1[4];
    __builtin_memcpy(a0, &x0, 16); __builtin_memcpy(a1, &x1, 16);
    if constexpr (MKT_PERMLANE_SWAP && (D == 16 || D == 32)) {
        // gfx950 v_permlane16_swap / v_permlane32_swap: the odd rows (upper half) of the first register trade places
        // with the even rows (lower half) of the second -- exactly this exchange, one instruction per dword, no selects
#pragma unroll
        for (int w = 0; w < 4; w++) {
            auto r = D == 16 ? __builtin_amdgcn_permlane16_swap((unsigned)a0[w], (unsigned)a1[w], false, false)
                             : __builtin_amdgcn_permlane32_swap((unsigned)a0[w], (unsigned)a1[w], false, false);
            a0[w] = (int)r[0]; a1[w] = (int)r[1];
        }
    } else if constexpr (MKT_DPP_PAIR && (D == 1 || D == 2 || D == 4 || D == 8)) {
        // Two DPP reads instead of one read and three selects: r0 = the partner's x0 as the b = 0 lanes see it, r1 = the
        // partner's x1 as the b = 1 lanes see it; each is consumed only in the lanes for which its lane pattern is the
        // right one, so lane ^ 4 needs no choice between row_ror:4 and row_ror:12 (dst[i] = src[(i - n) mod 16]:
        // row_ror:12 serves the lanes with bit 2 clear, row_ror:4 those with it set).  The compiler folds each DPP
        // move into the v_cndmask that consumes it where it can.
        constexpr int CA = D == 1 ? 0xB1 : D == 2 ? 0x4E : D == 4 ? 0x12C : 0x128;   // seen by the b = 0 lanes
        constexpr int CB = D == 1 ? 0xB1 : D == 2 ? 0x4E : D == 4 ? 0x124 : 0x128;   // seen by the b = 1 lanes
        if constexpr (D >= 4) {
            // lane bits 2 and 3 select whole banks of 4 lanes: the DPP bank mask does the merge, no select at all
            constexpr int BM0 = D == 4 ? 0x5 : 0x3, BM1 = D == 4 ? 0xA : 0xC;       // banks holding the b = 0 / b = 1 lanes
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int n1 = __builtin_amdgcn_update_dpp(a1[w], a0[w], CA, 0xf, BM0, false);
                const int n0 = __builtin_amdgcn_update_dpp(a0[w], a1[w], CB, 0xf, BM1, false);
                a1[w] = n1; a0[w] = n0;
            }
        } else {
            const bool nb = !b;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int r0 = dpp_mov<CA>(a0[w]);
                const int r1 = dpp_mov<CB>(a1[w]);
                a1[w] = b ? a1[w] : r0;
                a0[w] = nb ? a0[w] : r1;
            }
        }
    } else {
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int snd = b ? a0[w] : a1[w];
            const int rcv = xor_lane<D>(snd, lx);
            a0[w] = b ? rcv : a0[w];
            a1[w] = b ? a1[w] : rcv;
        }
    }
    __builtin_memcpy(&x0, a0, 16); __builtin_memcpy(&x1, a1, 16);
}
// full transpose between thread bits [LO, LO+2) (lane bits) and the two slot bits
template <int LO, int NB>
__device__ __forceinline__ void exchange_lane_full(cplx (&z)[NB][4], const LaneX &lx) {
    const bool b0 = (lx.lane >> LO) & 1, b1 = (lx.lane >> (LO + 1)) & 1;
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        swap_pair<(1 << LO)>(z[nb][0], z[nb][1], b0, lx);          // lane bit LO   <-> slot bit 0
        swap_pair<(1 << LO)>(z[nb][2], z[nb][3], b0, lx);
        swap_pair<(2 << LO)>(z[nb][0], z[nb][2], b1, lx);          // lane bit LO+1 <-> slot bit 1
        swap_pair<(2 << LO)>(z[nb][1], z[nb][3], b1, lx);
    }
}
// single-stage window [lo+1, lo+3) -> [lo, lo+2), lo = LB (a lane bit), forward direction:
// slots (b2,b1) with thread bit LB = b0  ->  slots (b1,b0) with thread bit LB = b2
template <int NB, int LB>
__device__ __forceinline__ void exchange_lane_odd_fwd(cplx (&z)[NB][4], const LaneX &lx) {
    const bool lam = (lx.lane >> LB) & 1;
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        cplx n[4];
#pragma unroll
        for (int b1 = 0; b1 < 2; b1++) {
            cplx A = z[nb][b1], Bq = z[nb][2 + b1];                // old slots (b2 = 0, b1), (b2 = 1, b1)
            // keep the one with b2 == lam, trade the other: swap_pair gives/receives exactly that
            swap_pair<(1 << LB)>(A, Bq, lam, lx);                          // lam = 0: receives into Bq ; lam = 1: receives into A
            // after the swap: lam = 0 holds A = own(0,b1), Bq = partner's (0,b1) ; lam = 1 holds A = partner's (1,b1), Bq = own (1,b1)
            n[2 * b1 + 0] = A;                                     // new slot (b1, b0 = 0): b0 = thread bit of the source
            n[2 * b1 + 1] = Bq;                                    // new slot (b1, b0 = 1)
        }
#pragma unroll
        for (int e = 0; e < 4; e++) z[nb][e] = n[e];
    }
}
// inverse direction: slots (b1,b0) with thread bit0 = b2  ->  slots (b2,b1) with thread bit0 = b0
template <int NB, int LB>
__device__ __forceinline__ void exchange_lane_odd_inv(cplx (&z)[NB][4], const LaneX &lx) {
    const bool lam = (lx.lane >> LB) & 1;
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        cplx n[4];
#pragma unroll
        for (int b1 = 0; b1 < 2; b1++) {
            cplx A = z[nb][2 * b1], Bq = z[nb][2 * b1 + 1];        // old slots (b1, b0 = 0), (b1, b0 = 1)
            swap_pair<(1 << LB)>(A, Bq, lam, lx);
            n[b1] = A;                                             // new slot (b2 = 0, b1)
            n[2 + b1] = Bq;                                        // new slot (b2 = 1, b1)
        }
#pragma unroll
        for (int e = 0; e < 4; e++) z[nb][e] = n[e];
    }
}

#ifndef MKT_LANE_EXCHANGE
#define MKT_LANE_EXCHANGE 4   // 0: every exchange through LDS; 1: in-wave wherever legal; 2: only the single-stage window;
                              // 3: 2 + quad-local (d = 1, 2); 5-7: route mixes for tuning; 8: single-stage window + the
                              // transposes along lane bits 2-5 (bank-masked DPP / permlane swaps), quad-local ones via LDS;
                              // 4 (default, measured): 8 at every size (with paired transforms also at M = 1024: +3 % over 1)
#endif

// which route the exchange between two windows of the schedule takes (compile time): 0 LDS, 1 in-wave 4x4, 2 odd window
// MO >= 0: per-kernel override -- low byte = exchange-route mode (0xff = the library default MKT_LANE_EXCHANGE), bit 8 =
// ONE staging buffer with a barrier on both sides of every LDS exchange (kernels whose LDS budget has no room for two),
// bit 9 = single transforms use the staging-buffer stride of paired ones (see exchange)
template <int LOGM, int LOGR, int MO = -1>
struct Route {
    static constexpr bool SB = MO >= 0 && (MO & 0x100) != 0;
    using P = Plan<LOGM, LOGR>;
    static constexpr int LANEBITS = (LOGM - LOGR) < 6 ? (LOGM - LOGR) : 6;   // thread bits that are lane bits
    __host__ __device__ static constexpr int of(int lo_a, int lo_b) {
        const int lomin = lo_a < lo_b ? lo_a : lo_b, diff = lo_a < lo_b ? lo_b - lo_a : lo_a - lo_b;
        constexpr int MODE = (MO >= 0 && (MO & 0xff) != 0xff) ? (MO & 0xff) : MKT_LANE_EXCHANGE == 4 ? 8 : ((LOGM & 1) && MKT_LANE_EXCHANGE >= 5 && MKT_LANE_EXCHANGE <= 7 ? 2 : MKT_LANE_EXCHANGE);
        if ((MODE == 1 || (MODE == 3 && lomin == 0) || (MODE == 5 && (lomin == 4 || lomin == 0)) || (MODE == 6 && lomin == 4) || (MODE == 7 && (lomin == 4 || lomin == 2)) || (MODE == 8 && (lomin == 4 || lomin == 2))) && LOGR == 2 && diff == 2 && lomin + 2 <= LANEBITS && lomin + 2 <= (MKT_PERMLANE_SWAP ? 6 : 5)) return 1;
        if (MODE >= 1 && LOGR == 2 && diff == 1 && lomin == P::OVL && LANEBITS >= lomin + 1) return 2;
        return 0;
    }
    // staging buffer (pass parity) of the first / last LDS exchange of a forward and of an inverse transform, -1 if none
    __host__ __device__ static constexpr int fwd_first() { for (int p = 0; p < P::NPASS - 1; p++) if (of(P::lo(p), P::lo(p + 1)) == 0) return p & 1; return -1; }
    __host__ __device__ static constexpr int fwd_last() { for (int p = P::NPASS - 2; p >= 0; p--) if (of(P::lo(p), P::lo(p + 1)) == 0) return p & 1; return -1; }
    __host__ __device__ static constexpr int inv_first() { for (int p = P::NPASS - 1; p >= 1; p--) if (of(P::lo(p), P::lo(p - 1)) == 0) return p & 1; return -1; }
    __host__ __device__ static constexpr int inv_last() { for (int p = 1; p <= P::NPASS - 1; p++) if (of(P::lo(p), P::lo(p - 1)) == 0) return p & 1; return -1; }
    // a transform must open with a barrier if its first LDS exchange reuses the buffer of the previous transform's last
    static constexpr bool guard_fwd = !SB && fwd_first() >= 0 && (fwd_first() == fwd_last() || fwd_first() == inv_last());
    static constexpr bool guard_inv = !SB && inv_first() >= 0 && (inv_first() == fwd_last() || inv_first() == inv_last());
};

// one exchange between the windows lo_from / lo_to of the schedule, by the cheapest legal route; PASS = the pass
// whose output is exchanged (selects the staging buffer)
template <int LOGM, int LOGR, int NB, int LO_FROM, int LO_TO, bool FWD, int PASS, int MO = -1>
__device__ __forceinline__ void exchange(cplx (&z)[NB][1 << LOGR], cplx *lds, int t, const LaneX &lx) {
    constexpr int R = Route<LOGM, LOGR, MO>::of(LO_FROM, LO_TO);
    if constexpr (R == 1) exchange_lane_full<(LO_FROM < LO_TO ? LO_FROM : LO_TO), NB>(z, lx);
    else if constexpr (R == 2) { constexpr int LB = Plan<LOGM, LOGR>::OVL; if constexpr (FWD) exchange_lane_odd_fwd<NB, LB>(z, lx); else exchange_lane_odd_inv<NB, LB>(z, lx); }
    else {
        constexpr bool SB = Route<LOGM, LOGR, MO>::SB;
        if (SB) __syncthreads();                     // every reader of the single buffer is done before it is rewritten
        // bit 9 of MO: a single transform in a kernel that also runs paired ones keeps the PAIRED buffer stride, so its
        // two staging buffers lie inside the paired ones of the same parity and the guard logic holds across both kinds
        constexpr bool S2 = MO >= 0 && (MO & 0x200) != 0 && NB == 1;
        exchange_lds<LOGM, LOGR, NB>(z, lds + (SB ? 0 : S2 ? Plan<LOGM, LOGR, 2>::buf_off(PASS) : Plan<LOGM, LOGR, NB>::buf_off(PASS)), t, LO_FROM, LO_TO);
    }
}

// fft.jl:105-155: for stage bit b (stride k = 2^b, m = 2^(LOGM-1-b)), butterfly on (j, j+k):
//   u = a[j+k] * Psi[m + (j >> (b+1))];  a[j], a[j+k] = a[j] + u, a[j] - u
// In: slot e = point e*NT + t.  Out: slot e = point t*R + e.  NB independent transforms share the
// twiddle loads and the barriers.
// MO bit 10 (0x400): the twiddles of the NEXT pass are read from the table before the exchange that precedes it, so their LDS
// latency passes under the exchange instead of in front of the pass's first butterfly (the compiler keeps every table read behind
// the staging writes it cannot tell apart from the table, and waits for it 1-13 instructions later: tools/isa.sh)
template <int LOGM, int LOGR, int NB, int PASS>
__device__ __forceinline__ void tw_load_fwd(const cplx *__restrict__ psi, int t, cplx (&w)[(1 << LOGR) - 1]) {
    using P = Plan<LOGM, LOGR, NB>;
    constexpr int lo = P::lo(PASS);
    int n = 0;
#pragma unroll
    for (int s = 0; s < P::nst(PASS); s++) {
        const int b = P::hib(PASS) - s, sb = b - lo;
        const int twbase = (1 << (LOGM - 1 - b)) + ((t >> lo) << (LOGR - 1 - sb));
#pragma unroll
        for (int g = 0; g < (1 << (LOGR - 1 - sb)); g++) w[n++] = psi[twbase + g];
    }
}
template <int LOGM, int LOGR, int NB, int PASS = 0, int MO = -1>
__device__ __forceinline__ void fft_forward_pass(cplx (&z)[NB][1 << LOGR], const cplx *__restrict__ psi, cplx *lds, int t, const LaneX &lx, const cplx (&wpre)[(1 << LOGR) - 1]) {
    using P = Plan<LOGM, LOGR, NB>;
    constexpr bool PF = MO >= 0 && (MO & 0x400) != 0;
    constexpr int p = PASS, lo = P::lo(p);
    int n = 0;
#pragma unroll
    for (int s = 0; s < P::nst(p); s++) {
        const int b = P::hib(p) - s, sb = b - lo;
        const int twbase = (1 << (LOGM - 1 - b)) + ((t >> lo) << (LOGR - 1 - sb));
#pragma unroll
        for (int g = 0; g < (1 << (LOGR - 1 - sb)); g++) {
            const cplx w = PF ? wpre[n] : psi[twbase + g];
            n++;
#pragma unroll
            for (int q = 0; q < (1 << sb); q++) {
                const int e = (g << (sb + 1)) | q, e2 = e | (1 << sb);
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    cplx u;
                    const cplx x = z[nb][e2];
                    if (MKT_FFT_SPECIAL && PASS == 0 && b == LOGM - 1) {           // w = (eps, -1): x.im * -1 and x.re * -1 are exact
                        u.re = x.re * w.re + x.im; u.im = x.im * w.re - x.re;
                    } else if (MKT_FFT_SPECIAL && PASS == 0 && b == LOGM - 2) {    // w = (c, -c) / (-c, -c): two of the four products are the other two, negated
                        const double pp = x.re * w.re, qq = x.im * w.re;
                        if (g == 0) { u.re = pp + qq; u.im = qq - pp; } else { u.re = pp - qq; u.im = pp + qq; }
                    } else u = cmul(x, w);
                    const cplx a = z[nb][e];
                    z[nb][e] = cadd(a, u); z[nb][e2] = csub(a, u);
                }
            }
        }
    }
    if constexpr (p < P::NPASS - 1) {
        cplx wnext[(1 << LOGR) - 1];
        if constexpr (PF) tw_load_fwd<LOGM, LOGR, NB, PASS + 1>(psi, t, wnext);
        exchange<LOGM, LOGR, NB, P::lo(p), P::lo(p + 1), true, PASS, MO>(z, lds, t, lx);
        fft_forward_pass<LOGM, LOGR, NB, PASS + 1, MO>(z, psi, lds, t, lx, wnext);
    }
}
template <int LOGM, int LOGR, int NB, int MO = -1>
__device__ __forceinline__ void fft_forward(cplx (&z)[NB][1 << LOGR], const cplx *__restrict__ psi, cplx *lds, int t, const LaneX &lx) {
    constexpr bool PF = MO >= 0 && (MO & 0x400) != 0;
    cplx w0[(1 << LOGR) - 1];
    if constexpr (PF) tw_load_fwd<LOGM, LOGR, NB, 0>(psi, t, w0);
    if (Route<LOGM, LOGR, MO>::guard_fwd) __syncthreads();
    fft_forward_pass<LOGM, LOGR, NB, 0, MO>(z, psi, lds, t, lx, w0);
}

// fft.jl:159-209: t, u = a[j], a[j+k];  a[j] = t + u;  a[j+k] = (t - u) * Psiinv[m + (j >> (b+1))]
// In: slot e = point t*R + e.  Out: slot e = point e*NT + t.
// CONJ: `psiinv` points at the FORWARD table Psi and the butterflies multiply by its conjugate (Psiinv == conj(Psi)
// entry for entry, fft.jl:33-34), so one table -- e.g. a copy resident in LDS -- serves both directions.
template <int LOGM, int LOGR, int NB, int PASS>
__device__ __forceinline__ void tw_load_inv(const cplx *__restrict__ psiinv, int t, cplx (&w)[(1 << LOGR) - 1]) {
    using P = Plan<LOGM, LOGR, NB>;
    constexpr int lo = P::lo(PASS);
    int n = 0;
#pragma unroll
    for (int s = P::nst(PASS) - 1; s >= 0; s--) {
        const int b = P::hib(PASS) - s, sb = b - lo;
        const int twbase = (1 << (LOGM - 1 - b)) + ((t >> lo) << (LOGR - 1 - sb));
#pragma unroll
        for (int g = 0; g < (1 << (LOGR - 1 - sb)); g++) w[n++] = psiinv[twbase + g];
    }
}
template <int LOGM, int LOGR, int NB, bool CONJ, int PASS, int MO = -1>
__device__ __forceinline__ void fft_inverse_pass(cplx (&z)[NB][1 << LOGR], const cplx *__restrict__ psiinv, cplx *lds, int t, const LaneX &lx, const cplx (&wpre)[(1 << LOGR) - 1]) {
    using P = Plan<LOGM, LOGR, NB>;
    constexpr bool PF = MO >= 0 && (MO & 0x400) != 0;
    constexpr int p = PASS, lo = P::lo(p);
    int n = 0;
#pragma unroll
    for (int s = P::nst(p) - 1; s >= 0; s--) {
        const int b = P::hib(p) - s, sb = b - lo;
        const int twbase = (1 << (LOGM - 1 - b)) + ((t >> lo) << (LOGR - 1 - sb));
#pragma unroll
        for (int g = 0; g < (1 << (LOGR - 1 - sb)); g++) {
            const cplx w = PF ? wpre[n] : psiinv[twbase + g];
            n++;
#pragma unroll
            for (int q = 0; q < (1 << sb); q++) {
                const int e = (g << (sb + 1)) | q, e2 = e | (1 << sb);
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    const cplx a = z[nb][e], u = z[nb][e2];
                    z[nb][e] = cadd(a, u);
                    const cplx x = csub(a, u);
                    cplx r;
                    if (MKT_FFT_SPECIAL && CONJ && PASS == 0 && b == LOGM - 1) {   // conj(w) = (eps, +1)
                        r.re = x.re * w.re - x.im; r.im = x.im * w.re + x.re;
                    } else if (MKT_FFT_SPECIAL && CONJ && PASS == 0 && b == LOGM - 2) {
                        const double pp = x.re * w.re, qq = x.im * w.re;
                        if (g == 0) { r.re = pp - qq; r.im = qq + pp; } else { r.re = pp + qq; r.im = qq - pp; }
                    } else r = CONJ ? cmul_conj(x, w) : cmul(x, w);
                    z[nb][e2] = r;
                }
            }
        }
    }
    if constexpr (p > 0) {
        cplx wnext[(1 << LOGR) - 1];
        if constexpr (PF) tw_load_inv<LOGM, LOGR, NB, PASS - 1>(psiinv, t, wnext);
        exchange<LOGM, LOGR, NB, P::lo(p), P::lo(p - 1), false, PASS, MO>(z, lds, t, lx);
        fft_inverse_pass<LOGM, LOGR, NB, CONJ, PASS - 1, MO>(z, psiinv, lds, t, lx, wnext);
    }
}
template <int LOGM, int LOGR, int NB, bool CONJ = false, int MO = -1>
__device__ __forceinline__ void fft_inverse(cplx (&z)[NB][1 << LOGR], const cplx *__restrict__ psiinv, cplx *lds, int t, const LaneX &lx) {
    constexpr bool PF = MO >= 0 && (MO & 0x400) != 0;
    cplx w0[(1 << LOGR) - 1];
    if constexpr (PF) tw_load_inv<LOGM, LOGR, NB, Plan<LOGM, LOGR, NB>::NPASS - 1>(psiinv, t, w0);
    if (Route<LOGM, LOGR, MO>::guard_inv) __syncthreads();
    fft_inverse_pass<LOGM, LOGR, NB, CONJ, Plan<LOGM, LOGR, NB>::NPASS - 1, MO>(z, psiinv, lds, t, lx, w0);
}

// The barriers of fft_inverse / fft_forward without the transform: thread groups of a workgroup that sit out a transform
// which other groups run must still arrive at every workgroup barrier it contains.
template <int LOGM, int LOGR, int NB, int LO_FROM, int LO_TO, int MO>
__device__ __forceinline__ void exchange_barriers_only() {
    if constexpr (Route<LOGM, LOGR, MO>::of(LO_FROM, LO_TO) == 0) {
        if (Route<LOGM, LOGR, MO>::SB) __syncthreads();
        __syncthreads();                             // the one inside exchange_lds
    }
}
template <int LOGM, int LOGR, int NB, int PASS, int MO>
__device__ __forceinline__ void fft_inverse_barriers_pass() {
    using P = Plan<LOGM, LOGR, NB>;
    if constexpr (PASS > 0) {
        exchange_barriers_only<LOGM, LOGR, NB, P::lo(PASS), P::lo(PASS - 1), MO>();
        fft_inverse_barriers_pass<LOGM, LOGR, NB, PASS - 1, MO>();
    }
}
template <int LOGM, int LOGR, int NB, int PASS, int MO>
__device__ __forceinline__ void fft_forward_barriers_pass() {
    using P = Plan<LOGM, LOGR, NB>;
    if constexpr (PASS < P::NPASS - 1) {
        exchange_barriers_only<LOGM, LOGR, NB, P::lo(PASS), P::lo(PASS + 1), MO>();
        fft_forward_barriers_pass<LOGM, LOGR, NB, PASS + 1, MO>();
    }
}
template <int LOGM, int LOGR, int NB, int MO = -1>
__device__ __forceinline__ void fft_forward_barriers_only() {
    if (Route<LOGM, LOGR, MO>::guard_fwd) __syncthreads();
    fft_forward_barriers_pass<LOGM, LOGR, NB, 0, MO>();
}
template <int LOGM, int LOGR, int NB, int MO = -1>
__device__ __forceinline__ void fft_inverse_barriers_only() {
    if (Route<LOGM, LOGR, MO>::guard_inv) __syncthreads();
    fft_inverse_barriers_pass<LOGM, LOGR, NB, Plan<LOGM, LOGR, NB>::NPASS - 1, MO>();
}

// Device point order of the resident TransPolys (keys, monomial table, phase-1 output): where the
// reference-order point x = 4t+e (slot e of thread t after the forward transform) is stored.
// order 1: slot-major -- every 16-byte load of a wave is one contiguous KiB (the k = 1 rotation kernels, CCS, KMS phase 2:
//          0.5-1 % ahead of order 2, 3-6 % at KMS2partyblock);
// order 2: slot pairs, 32 B per lane (the RLWE-length-k kernels: order 1 runs them 60 % slower).
// A context picks one for all its resident tables (context.cpp); API-visible TransPolys are in the reference's order.
#ifndef MKT_DEVORDER
#define MKT_DEVORDER 1
#endif
#ifndef MKT_DEVORDER_KR
#define MKT_DEVORDER_KR 2
#endif
#ifndef MKT_LOGR
#define MKT_LOGR 2   // points per thread = 2^MKT_LOGR in every transform schedule (and in dev_pos)
#endif
__host__ __device__ __forceinline__ int dev_pos(int order, int x, int NT) {
    constexpr int LR = MKT_LOGR, RM = (1 << LR) - 1;
    if (order == 1) return (x & RM) * NT + (x >> LR);
    return ((x & RM) >> 1) * (2 * NT) + ((x >> LR) << 1) + (x & 1);
}

// ---- ring words ----
template <typename WORD> struct WordTraits;
template <> struct WordTraits<uint32_t> { static constexpr int W = 32; typedef int32_t S; };
template <> struct WordTraits<uint64_t> { static constexpr int W = 64; typedef int64_t S; };

// signed(x) converted to Float64 (fft.jl:60); round-to-nearest for |x| > 2^53
template <typename WORD>
__device__ __forceinline__ double word_to_f64(WORD x) { return (double)(typename WordTraits<WORD>::S)x; }

// arithmetic.jl:1-9 native(x, mask):  x -= floor(x * 2^-W) * 2^W;  x == 2^W ? 0 : trunc_to_unsigned(x)
// The reduction is the reference's two operations (the scaling by a power of two is exact, the subtraction rounds as
// it does there) and leaves y in [0, 2^W].  The conversion is done without v_cvt / v_cmp / v_cndmask: for an integer v
// in [0, 2^32] the low dword of the double v + 2^52 is v mod 2^32 (the sum is exact below 2^53), which maps the one
// special value 2^W -- the rounded-up sum of a tiny negative x and 2^W -- to 0 exactly like the reference's test.
// MKT_NATIVE_MAGIC 0 keeps the literal compare-and-convert form (same bits: tests/csrc/native_check.cpp compares the two
// forms on 46 M inputs of every magnitude class on the host).
#ifndef MKT_NATIVE_MAGIC
#define MKT_NATIVE_MAGIC 1
#endif
#ifndef MKT_NATIVE_CARRY
#define MKT_NATIVE_CARRY 1
#endif
__device__ __forceinline__ uint32_t low_dword_of_sum_2p52(double v) {
    const double s = v + 4503599627370496.0;   // 2^52
    return (uint32_t)__double2loint(s);
}
template <typename WORD> __device__ __forceinline__ WORD native(double x);
template <> __device__ __forceinline__ uint32_t native<uint32_t>(double x) {
    x -= floor(x * 2.3283064365386963e-10) * 4.294967296e9;
#if MKT_NATIVE_MAGIC
    return low_dword_of_sum_2p52(trunc(x));                  // trunc, as the reference (x >= 0 unless x * 2^-W underflowed)
#else
    return x == 4.294967296e9 ? 0u : (uint32_t)x;
#endif
}
template <> __device__ __forceinline__ uint64_t native<uint64_t>(double x) {
    x -= floor(x * 5.421010862427522e-20) * 1.8446744073709552e19;
#if MKT_NATIVE_MAGIC
    const double hi = trunc(x * 2.3283064365386963e-10);    // in [0, 2^32], exact
    const double lo = trunc(x - hi * 4.294967296e9);        // exact difference in [0, 2^32), fraction dropped
    return ((uint64_t)low_dword_of_sum_2p52(hi) << 32) | (uint64_t)low_dword_of_sum_2p52(lo);
#else
    return x == 1.8446744073709552e19 ? (uint64_t)0 : (uint64_t)x;
#endif
}

// acc + native(x) mod 2^W.  64-bit ring: the two 32-bit halves of native(x) are added with a carry (v_add_co_u32, v_addc_co_u32) instead of
// being joined into a register pair first (two v_mov and two 64-bit adds per coefficient in the rotation kernels' CMux)
template <typename WORD> __device__ __forceinline__ WORD native_add(WORD acc, double x);
template <> __device__ __forceinline__ uint32_t native_add<uint32_t>(uint32_t acc, double x) { return acc + native<uint32_t>(x); }
template <> __device__ __forceinline__ uint64_t native_add<uint64_t>(uint64_t acc, double x) {
#if MKT_NATIVE_MAGIC && MKT_NATIVE_CARRY
    x -= floor(x * 5.421010862427522e-20) * 1.8446744073709552e19;
    const double hi = trunc(x * 2.3283064365386963e-10);
    const double lo = trunc(x - hi * 4.294967296e9);
    const uint32_t h = low_dword_of_sum_2p52(hi), l = low_dword_of_sum_2p52(lo);
    const uint32_t al = (uint32_t)acc, s = al + l;
    const uint32_t ah = (uint32_t)(acc >> 32) + h + (s < l ? 1u : 0u);
    return ((uint64_t)ah << 32) | s;
#else
    return acc + native<uint64_t>(x);
#endif
}

// arithmetic.jl:23-27 divbits (bit may be 0: Julia shifts by >= width give 0)
template <typename WORD>
__device__ __forceinline__ WORD divbits(WORD a, int bit) {
    constexpr int W = WordTraits<WORD>::W;
    if (bit <= 0) return a;
    const WORD carry = (WORD)(a << (W - bit)) >> (W - 1);
    return (WORD)((a >> bit) + carry);
}

// Balanced gadget decomposition (gsw.jl:42-52 / :86-96, unienc.jl:4-18), closed form:
//   t' = divbits(x, W - l*logB) + sum_j (B/2) * B^j ;  digit_j = ((t' >> logB*(l-1-j)) & (B-1)) - B/2
// which reproduces the reference's carry chain digit for digit (j = 0 most significant).
template <typename WORD>
struct Gadget {
    int l, logB;
    WORD offset, mask, half;
    int bit;
    __host__ __device__ Gadget() {}
    __host__ __device__ Gadget(int l_, int logB_) : l(l_), logB(logB_) {
        constexpr int W = WordTraits<WORD>::W;
        mask = (WORD)(((WORD)1 << logB) - 1);
        half = (WORD)((WORD)1 << (logB - 1));
        offset = 0;
        for (int j = 0; j < l; j++) offset = (WORD)(offset + (WORD)(half << (logB * j)));
        bit = W - l * logB;
    }
    __device__ __forceinline__ WORD prep(WORD x) const { return (WORD)(divbits<WORD>(x, bit) + offset); }
    // signed digit j of a prepared word, as a (small) int
    __device__ __forceinline__ int digit(WORD tp, int j) const {
        return (int)((tp >> (logB * (l - 1 - j))) & mask) - (int)half;
    }
};

}  // namespace mktd
