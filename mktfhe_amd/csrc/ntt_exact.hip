// MKT_ARITH_EXACT: the negacyclic number-theoretic transform over Z_P[X]/(X^N + 1) in residue form, P = p1 * p2 with the
// two largest NTT primes below 2^30, p1 = 131063 * 2^13 + 1 and p2 = 131066 * 2^13 + 1 (P = 2^59.9998), batched HBM -> HBM; the exact negacyclic
// product of a gadget-digit polynomial with a ring polynomial mod 2^W built on it (the operation the reference's Float64
// transform approximates: src/ring/fft.jl:57-81 + polynomial.jl:99-113; the MultiFloat option of README.md:9 aims at the
// same exact value); and the CGGI blind rotation with exact products.
//
// Why two 30-bit primes and not one 64-bit prime: gfx950 multiplies 32 x 32 bits per lane and instruction
// (v_mul_lo_u32 / v_mul_hi_u32, 4.4 cycles per wave each); a product mod p = 2^64 - 2^32 + 1 costs ~33 instructions, a
// Shoup product mod a 32-bit-word prime 4-6, so a butterfly over both residues is >2x cheaper than the Goldilocks one
// (tools/ntt_probe.hip).  A point is the pair (x mod p1, x mod p2) packed into 64 bits -- the same 8 N bytes per
// polynomial as M complex doubles.
//
// Why BELOW 2^30 (round 3; rounds 1-2 used 31-bit primes): the batched transforms are bound by integer issue, not by HBM
// (software-prefetching the next polynomial moved nothing), and 4p < 2^32 admits Harvey's lazy butterflies -- values stay
// in [0, 4p) through the forward stages and [0, 2p) through the inverse ones, corrections are single v_min_u32 -- 8 and 9
// instructions per residue butterfly instead of 12.  The 1.9 bits of modulus this gives up are not needed by any
// parameter set of the reference (exact_gate_ok, context.cpp: the largest bound is 2^52).
//
// Same butterfly network as the Float64 transform (fft_device.h), so the same pass / window / staging machinery: stage
// with stride 2^b multiplies by psi_rev[m + i] (Cooley-Tukey, bit-reversed output), the inverse runs Gentleman-Sande
// with the inverse table and a final N^-1.  8 points per thread, N / 8 threads per polynomial, passes of 3 stages local
// to a thread, LDS exchanges between passes.  Twiddles carry their Shoup companions floor(w * 2^32 / p); resident tables
// (keys, monomials) are stored in Montgomery form (x * 2^32 mod p) so that data x table products need no companion.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_api.h"
#include <type_traits>

#include "fft_device.h"

// The file is compiled once per translation unit MKT_NTT_TU (Makefile) so that its kernel families build in parallel (one unit took 175 s, the
// critical path of the whole build): 0 batched transforms + exact products, 1 CGGI / LMSS rotations, 2 KMS phase 1, 3 KMS phase 2 + CCS.  Undefined = everything.
#ifdef MKT_NTT_TU
#define MKT_NTT_IN(n) (MKT_NTT_TU == (n))
#else
#define MKT_NTT_IN(n) 1
#endif

namespace mktd {

namespace {

constexpr uint32_t P1 = 1073668097u, P2 = 1073692673u;      // 131063 * 2^13 + 1 < 131066 * 2^13 + 1 < 2^30: both = 1 mod 2^13
static_assert(P1 < P2 && P2 < (1u << 30), "lazy butterflies need 4 p < 2^32; crt_signed takes r.a < p2 as given");
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

// the smaller of two words as unsigned integers: one v_min_u32.  (`w < v ? w : v` with w = v - c is recognised as a borrow test and
// becomes v_sub_co + v_cndmask on vcc -- a 19-cycle instruction on gfx950, tools/valu_probe.hip -- in a third of the places.)
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return __builtin_elementwise_min(a, b); }
template <uint32_t P> __device__ __forceinline__ uint32_t red1(uint32_t v) { return umin32(v, v - P); }   // v < 2P -> [0, P)
template <uint32_t P> __device__ __forceinline__ uint32_t canon4(uint32_t v) { return red1<P>(umin32(v, v - 2u * P)); }   // v < 4P -> [0, P)
template <uint32_t P> __device__ __forceinline__ uint32_t addm(uint32_t x, uint32_t y) { return red1<P>(x + y); }               // 2P < 2^32
template <uint32_t P> __device__ __forceinline__ uint32_t subm(uint32_t x, uint32_t y) { const uint32_t d = x - y; return umin32(d, d + P); }
// x * w mod P for a constant w with companion ws = floor(w * 2^32 / P); any 32-bit x
template <uint32_t P> __device__ __forceinline__ uint32_t shoup(uint32_t x, uint32_t w, uint32_t ws) {
    const uint32_t q = __umulhi(x, ws);
    return red1<P>(x * w - q * P);
}
// Montgomery product x * y * 2^-32 mod P (x < 4P: a lazy forward-transform value is fine; y < P), result in [0, P)
// (two v_mad_u64_u32 and one v_mul_lo_u32: the 64-bit multiply-add issues at the rate of a 32-bit multiply on gfx950)
template <uint32_t P, uint32_t PINV> __device__ __forceinline__ uint32_t montmul(uint32_t x, uint32_t y) {
    const uint64_t z = (uint64_t)x * y;                          // < 4P * P < 2^62
    const uint32_t m = (uint32_t)z * (0u - PINV);                // z + m P = 0 mod 2^32
    return red1<P>((uint32_t)((z + (uint64_t)m * P) >> 32));     // (z + m P) / 2^32 < 4 P^2 / 2^32 + P < 2P
}
__device__ __forceinline__ Pt pt_shoup(Pt x, uint4 w) { Pt r; r.a = shoup<P1>(x.a, w.x, w.y); r.b = shoup<P2>(x.b, w.z, w.w); return r; }
__device__ __forceinline__ Pt pt_mont(Pt x, Pt y) { Pt r; r.a = montmul<P1, PI1>(x.a, y.a); r.b = montmul<P2, PI2>(x.b, y.b); return r; }
// acc + x * y * 2^-32 with a LAZY accumulator: acc in [0, 2P) -> [0, 2P) (x < 4P, y < P): 6 instructions per residue where the
// canonical pt_add(acc, pt_mont(x, y)) takes 8.  Fine as the x of pt_mont and as the input of ntt_inverse.
template <uint32_t P, uint32_t PINV> __device__ __forceinline__ uint32_t mac_lazy(uint32_t acc, uint32_t x, uint32_t y) {
    const uint64_t z = (uint64_t)x * y;
    const uint32_t m = (uint32_t)z * (0u - PINV);
    const uint32_t s = acc + (uint32_t)((z + (uint64_t)m * P) >> 32);      // < 4P
    return umin32(s, s - 2u * P);
}
// acc - x * y * 2^-32 and a + b on lazy values, [0, 2P) -> [0, 2P)
template <uint32_t P, uint32_t PINV> __device__ __forceinline__ uint32_t msub_lazy(uint32_t acc, uint32_t x, uint32_t y) {
    const uint64_t z = (uint64_t)x * y;
    const uint32_t m = (uint32_t)z * (0u - PINV);
    const uint32_t s = acc + 2u * P - (uint32_t)((z + (uint64_t)m * P) >> 32);      // (0, 4P)
    return umin32(s, s - 2u * P);
}
template <uint32_t P> __device__ __forceinline__ uint32_t add_lazy(uint32_t a, uint32_t b) { const uint32_t s = a + b; return umin32(s, s - 2u * P); }
__device__ __forceinline__ Pt pt_msub(Pt acc, Pt x, Pt y) { Pt r; r.a = msub_lazy<P1, PI1>(acc.a, x.a, y.a); r.b = msub_lazy<P2, PI2>(acc.b, x.b, y.b); return r; }
__device__ __forceinline__ Pt pt_add_lazy(Pt x, Pt y) { Pt r; r.a = add_lazy<P1>(x.a, y.a); r.b = add_lazy<P2>(x.b, y.b); return r; }
__device__ __forceinline__ Pt pt_mac(Pt acc, Pt x, Pt y) { Pt r; r.a = mac_lazy<P1, PI1>(acc.a, x.a, y.a); r.b = mac_lazy<P2, PI2>(acc.b, x.b, y.b); return r; }
__device__ __forceinline__ Pt pt_canon4(Pt x) { Pt r; r.a = canon4<P1>(x.a); r.b = canon4<P2>(x.b); return r; }
// WIDE lazy accumulation: the products x_j y_j of a sum over the gadget digits gathered in a 64-bit accumulator (ONE v_mad_u64_u32 per
// term and residue) and reduced ONCE: sum_j x_j y_j 2^-32 mod P -- the value the chain of mac_lazy gives, for 1 + 4 / T
// multiply-class instructions per term instead of 3 (6 in all).  Room: x < 2P (wide_x), y < P < 2^30: a term is below 2^61, the
// Montgomery correction m P below 2^62, so T <= 4 terms stay below 2^64; x < P (wide_x_canon) admits T <= 12.
struct Wide { uint64_t a, b; };
__device__ __forceinline__ Pt wide_x(Pt x) { Pt r; r.a = umin32(x.a, x.a - 2u * P1); r.b = umin32(x.b, x.b - 2u * P2); return r; }     // lazy forward value [0, 4P) -> [0, 2P)
__device__ __forceinline__ void wide_mac(Wide &acc, Pt x, Pt y) { acc.a += (uint64_t)x.a * y.a; acc.b += (uint64_t)x.b * y.b; }
template <uint32_t P, uint32_t PINV> __device__ __forceinline__ uint32_t wide_reduce1(uint64_t z) {
    const uint32_t m = (uint32_t)z * (0u - PINV);
    const uint32_t s = (uint32_t)((z + (uint64_t)m * P) >> 32);      // < 2^32: (T 2^61 + 2^62) / 2^32 <= 3 * 2^30 for T <= 4; x < P: (12 P^2 + 2^32 P) / 2^32 < 4P
    return umin32(s, s - 2u * P);                                     // [0, 2P): fine as the input of ntt_inverse and as the x of another product
}
__device__ __forceinline__ Pt wide_reduce(Wide w) { Pt r; r.a = wide_reduce1<P1, PI1>(w.a); r.b = wide_reduce1<P2, PI2>(w.b); return r; }
// Lazy butterflies (D. Harvey, "Faster arithmetic for number-theoretic transforms", 2014), 4P < 2^32.
// forward (Cooley-Tukey): x, y in [0, 4P) -> x + w y, x - w y in [0, 4P); FIRST: x < 2P already (a transform's first stage).
// The forward table holds the NEGATED twiddle (2^32 - w, with the companion of w): q P - y w is then one multiply-add
template <uint32_t P> __device__ __forceinline__ void bfly_fwd(uint32_t &x, uint32_t &y, uint32_t w, uint32_t ws, bool FIRST) {   // FIRST folds after unrolling
    uint32_t x1 = x;
    if (!FIRST) x1 = umin32(x, x - 2u * P);
    const uint32_t tn = __umulhi(y, ws) * P + y * w;             // w = -twiddle: -(twiddle y mod P) - {0, P}  (mod 2^32), a multiply-add
    x = x1 - tn; y = x1 + 2u * P + tn;
}
// the inverse's last stage with the scale c folded in: x, y in [0, 2P) -> c (x + y), c w (x - y) in [0, P)
template <uint32_t P> __device__ __forceinline__ void bfly_inv_scaled(uint32_t &x, uint32_t &y, uint32_t c, uint32_t cs, uint32_t cw, uint32_t cws) {
    const uint32_t s = x + y, d = x - y + 2u * P;
    x = shoup<P>(s, c, cs); y = shoup<P>(d, cw, cws);
}
// inverse (Gentleman-Sande): x, y in [0, 2P] -> x + y, winv (x - y) in [0, 2P] (closed: every consumer takes 2P -- sums stay <= 4P < 2^32,
// Shoup products take any 32-bit value).  There is NO inverse table: in bit-reversed order psiinv_rev[m + i] = -psi_rev[m + (m - 1 - i)]
// mod p (psi^N = -1), so the inverse reads the FORWARD table's mirrored entry (nw = 2^32 - w, the companion ws of w) and multiplies by
// p - w:  with r = d w - floor(d ws / 2^32) p in [0, 2P) (Shoup),  (p - w) d = -r = 2P - r mod p, and 2P - r = (d nw + 2P) + q p mod 2^32 --
// two multiply-adds and a v_mul_hi, the count of the table-driven form, and a workgroup stages one table instead of two.
template <uint32_t P> __device__ __forceinline__ void bfly_inv(uint32_t &x, uint32_t &y, uint32_t nw, uint32_t ws) {
    const uint32_t s = x + y, d = x - y + 2u * P, v = s - 2u * P;
    x = umin32(s, v);
    y = __umulhi(d, ws) * P + (d * nw + 2u * P);
}

extern __shared__ __attribute__((aligned(16))) unsigned char ntt_smem[];

// Staging slot of point i: the XOR swizzle of the Float64 transform (fft_device.h lds_pos: conflict-free b64 writes, 1.15x the
// conflict-free read cycles).  pt_index builds a point index from disjoint bit fields of t and e and the swizzle is linear over
// GF(2), so slot(t, e) = slot(t, 0) ^ slot(e << lo): one base per exchange side and ONE v_xor per address, where the compiler
// spent 2.3 instructions per address on the composed expression (19 of a point's ~150 in these integer-issue-bound kernels).
// (MKT_NTT_LAYOUT=1: the padded layout i + 2 (i >> 4), whose slots are immediate offsets -- no address arithmetic at all, but
// 1.35x the conflict-free LDS cycles and ds_read2/write2 pairs: +3 % on the batched transforms, -25 % on the gate kernels, which
// run two waves per SIMD and wait on every exchange.  Measured, not shipped.)
#ifndef MKT_NTT_LAYOUT
#define MKT_NTT_LAYOUT 0
#endif
__host__ __device__ constexpr int ntt_pos(int i) { return MKT_NTT_LAYOUT ? i + 2 * (i >> 4) : (i ^ ((i >> 3) & 15)); }
__host__ __device__ constexpr int ntt_join(int base, int off) { return MKT_NTT_LAYOUT ? base + off : (base ^ off); }
template <int LOGN> struct NttLds { static constexpr int WORDS = MKT_NTT_LAYOUT ? ntt_pos(1 << LOGN) : (1 << LOGN); };   // 64-bit words of one polynomial's staging buffer
// Which exchanges cross waves: thread bit j holds point bit j (j < lo) or j + 3 (j >= lo), so with both windows at lo <= 6 every wave bit
// (j >= 6) keeps its point bit (j + 3 >= 9, above the swizzled bits): the exchange moves points between the lanes of ONE wave, inside that
// wave's own part of the staging buffer, and needs no barrier -- a wave's LDS operations complete in order.  At N = 1024 .. 4096 that is
// two of a transform's three exchanges.  What remains: a cross-wave exchange keeps its two barriers, and the first wave-private one
// behind a cross-wave one (LEAD) opens with a barrier, so that no wave overwrites its part while another still reads it: 3 barriers per
// transform instead of 6 (MKT_NTT_XPRIV=0: all six).  Single-wave workgroups keep the plain form (their barriers cost nothing).
#ifndef MKT_NTT_XPRIV
#define MKT_NTT_XPRIV 1
#endif
// LEAN: the thread index is laundered through an empty asm so that the slot addresses (functions of t alone) are NOT hoisted out of the
// caller's loop: in a register-capped kernel (three waves per SIMD: 168) the ~50 hoisted addresses are spilled and reloaded from
// scratch at every use; recomputed they are one v_xor each
#ifndef MKT_LEAN_FINE
#define MKT_LEAN_FINE 0
#endif
// development only: timing builds with parts of the work removed (WRONG results) -- 1: no exchanges through LDS, 2: no key-row loads,
// 4: no twiddle reads, 8: no rotation through LDS (tools/ntt_variant.sh; profiles/r05_experiments.txt)
#ifndef MKT_ABLATE
#define MKT_ABLATE 0
#endif
__device__ __forceinline__ int launder(int v) { asm volatile("" : "+v"(v)); return v; }
// NB transforms side by side (the pair of digit polynomials of one accumulator; the low and the high half of one lifted sum) go through
// the passes together: ONE set of twiddle reads, slot addresses and barriers for NB x the butterflies, and every wait on an exchange or a
// twiddle read is covered by the other transform's arithmetic (what the Float64 rotation kernel's NB = 2 does: fft_device.h).  Transform b
// stages in its own buffer, lds + b * NttLds<LOGN>::WORDS.
template <int LOGN, int LO_FROM, int LO_TO, bool LEAD = true, bool LEAN = false, int NB = 1>
__device__ __forceinline__ void ntt_exchange_n(Pt (&z)[NB][8], uint64_t *lds, int t) {
    if (LEAN && MKT_LEAN_FINE) t = launder(t);
    if (MKT_ABLATE & 1) return;
    constexpr bool priv = MKT_NTT_XPRIV && LOGN - NLR > 6 && LO_FROM <= 6 && LO_TO <= 6;
    // the callers' LEAD rule (the exchange behind pass 0 crosses waves, the others stay in the wave) holds for 64-lane waves and N <= 4096:
    // at N = 8192 the exchange behind pass 1 (lo 7 -> 4) would cross waves too
    static_assert(!priv || LOGN <= 12, "wave-private exchanges: N <= 4096 (and 64-lane waves: gfx950 has no other)");
    if (!priv || LEAD) __syncthreads();
    // byte offsets, so that the XOR that joins a thread's base slot with point e's constant IS the address arithmetic (one v_xor per
    // access; as element indices each access paid a shift-add on top wherever the addresses are not hoisted)
    char *const lb = reinterpret_cast<char *>(lds);
    const int wr = ntt_pos(pt_index<NLR>(t, 0, LO_FROM)) * 8;
#pragma unroll
    for (int e = 0; e < 8; e++)
#pragma unroll
        for (int b = 0; b < NB; b++) *reinterpret_cast<uint64_t *>(lb + b * NttLds<LOGN>::WORDS * 8 + ntt_join(wr, ntt_pos(e << LO_FROM) * 8)) = pack(z[b][e]);
    if (!priv) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }   // compiler-level order only: the reads below stay below the writes above (a wave's LDS operations complete in order)
    const int rd = ntt_pos(pt_index<NLR>(t, 0, LO_TO)) * 8;
#pragma unroll
    for (int e = 0; e < 8; e++)
#pragma unroll
        for (int b = 0; b < NB; b++) z[b][e] = unpack(*reinterpret_cast<const uint64_t *>(lb + b * NttLds<LOGN>::WORDS * 8 + ntt_join(rd, ntt_pos(e << LO_TO) * 8)));
}
template <int LOGN, int LO_FROM, int LO_TO, bool LEAD = true, bool LEAN = false>
__device__ __forceinline__ void ntt_exchange(Pt (&z)[8], uint64_t *lds, int t) { ntt_exchange_n<LOGN, LO_FROM, LO_TO, LEAD, LEAN, 1>(reinterpret_cast<Pt(&)[1][8]>(z), lds, t); }

// In: slot e = point e*NT + t; slots 0..3 in [0, 2P), slots 4..7 any 32-bit value (fwd_in, res_small).  Out: slot e = point 8t + e (bit-reversed
// order of the transform, as the reference's), values LAZY in [0, 4P): fine as the x of montmul and of shoup; pt_canon4
// where the canonical residue is needed.
// psi[k] = (-w mod p1 as 2^32 - w, the companion of w, the same mod p2), w = psi^bitrev(k)
template <int LOGN, int PASS = 0, bool LEAN = false, int NB = 1>
__device__ __forceinline__ void ntt_forward_n(Pt (&z)[NB][8], const uint4 *__restrict__ psi, uint64_t *lds, int t) {
    using P = Plan<LOGN, NLR>;
    constexpr int p = PASS, lo = P::lo(p);
    if (LEAN && (MKT_LEAN_FINE || p == 0)) t = launder(t);
#pragma unroll
    for (int s = 0; s < P::nst(p); s++) {
        const int b = P::hib(p) - s, sb = b - lo;
        const int twbase = (1 << (LOGN - 1 - b)) + ((t >> lo) << (NLR - 1 - sb));
#pragma unroll
        for (int g = 0; g < (1 << (NLR - 1 - sb)); g++) {
            const uint4 w = (MKT_ABLATE & 4) ? make_uint4(twbase * 0x9E3779B1u + g, twbase + 77u * g, twbase * 0x85EBCA6Bu + g, twbase + 99u * g) : psi[twbase + g];
#pragma unroll
            for (int q = 0; q < (1 << sb); q++) {
                const int e = (g << (sb + 1)) | q, e2 = e | (1 << sb);
#pragma unroll
                for (int h = 0; h < NB; h++) {
                    bfly_fwd<P1>(z[h][e].a, z[h][e2].a, w.x, w.y, p == 0 && s == 0);
                    bfly_fwd<P2>(z[h][e].b, z[h][e2].b, w.z, w.w, p == 0 && s == 0);
                }
            }
        }
    }
    if constexpr (p < P::NPASS - 1) {
        ntt_exchange_n<LOGN, P::lo(p), P::lo(p + 1), (p <= 1), LEAN, NB>(z, lds, t);      // the exchange behind pass 0 crosses waves; LEAD for the one after it
        ntt_forward_n<LOGN, PASS + 1, LEAN, NB>(z, psi, lds, t);
    }
}
template <int LOGN, int PASS = 0, bool LEAN = false>
__device__ __forceinline__ void ntt_forward(Pt (&z)[8], const uint4 *__restrict__ psi, uint64_t *lds, int t) { ntt_forward_n<LOGN, PASS, LEAN, 1>(reinterpret_cast<Pt(&)[1][8]>(z), psi, lds, t); }
// In: slot e = point 8t + e, values in [0, 2P].  Out: slot e = point e*NT + t, scaled by the constant sc[0] (N^-1 or N^-1 2^32;
// sc[1] = sc[0] * psiinv[1]: the last stage has ONE twiddle, so the scale rides on its two products), values in [0, P).
// psi: the FORWARD table (bfly_inv)
template <int LOGN, int PASS, bool LEAN = false, int NB = 1>
__device__ __forceinline__ void ntt_inverse_n(Pt (&z)[NB][8], const uint4 *__restrict__ psi, uint64_t *lds, int t, const uint4 (&sc)[2]) {
    using P = Plan<LOGN, NLR>;
    constexpr int p = PASS, lo = P::lo(p);
    if (LEAN && (MKT_LEAN_FINE || p == P::NPASS - 1)) t = launder(t);
#pragma unroll
    for (int s = P::nst(p) - 1; s >= 0; s--) {
        const int b = P::hib(p) - s, sb = b - lo;
        const int twtop = (2 << (LOGN - 1 - b)) - 1 - ((t >> lo) << (NLR - 1 - sb));      // psiinv_rev[m + i] = -psi_rev[2m - 1 - i] (bfly_inv)
#pragma unroll
        for (int g = 0; g < (1 << (NLR - 1 - sb)); g++) {
            uint4 w = make_uint4(0, 0, 0, 0);
            if (!(p == 0 && s == 0)) w = (MKT_ABLATE & 4) ? make_uint4(twtop * 0x9E3779B1u + g, twtop + 77u * g, twtop * 0x85EBCA6Bu + g, twtop + 99u * g) : psi[twtop - g];
#pragma unroll
            for (int q = 0; q < (1 << sb); q++) {
                const int e = (g << (sb + 1)) | q, e2 = e | (1 << sb);
#pragma unroll
                for (int h = 0; h < NB; h++) {
                    if (p == 0 && s == 0) {
                        bfly_inv_scaled<P1>(z[h][e].a, z[h][e2].a, sc[0].x, sc[0].y, sc[1].x, sc[1].y);
                        bfly_inv_scaled<P2>(z[h][e].b, z[h][e2].b, sc[0].z, sc[0].w, sc[1].z, sc[1].w);
                    } else {
                        bfly_inv<P1>(z[h][e].a, z[h][e2].a, w.x, w.y);
                        bfly_inv<P2>(z[h][e].b, z[h][e2].b, w.z, w.w);
                    }
                }
            }
        }
    }
    if constexpr (p > 0) {
        ntt_exchange_n<LOGN, P::lo(p), P::lo(p - 1), (p == P::NPASS - 1), LEAN, NB>(z, lds, t);   // an inverse opens behind whatever used the buffer last: LEAD
        ntt_inverse_n<LOGN, PASS - 1, LEAN, NB>(z, psi, lds, t, sc);
    }
}
template <int LOGN, int PASS, bool LEAN = false>
__device__ __forceinline__ void ntt_inverse(Pt (&z)[8], const uint4 *__restrict__ psi, uint64_t *lds, int t, const uint4 (&sc)[2]) { ntt_inverse_n<LOGN, PASS, LEAN, 1>(reinterpret_cast<Pt(&)[1][8]>(z), psi, lds, t, sc); }

// Inputs of ntt_forward.  Slot e < 4 is the x of a first-stage butterfly and must be below 2P; slot e >= 4 is its y, which the
// Shoup product takes as ANY 32-bit value congruent to the point: conversions stop as early as the slot allows.
// signed 32-bit integer (a ring word of the 32-bit ring, a centered piece of the 64-bit ring)
template <uint32_t P> __device__ __forceinline__ uint32_t res_in(int32_t s, bool x_slot) {
    uint32_t v = (uint32_t)s + ((uint32_t)(s >> 31) & (3u * P));               // [0, 3P): 2^31 < 3P < 2^32
    if (x_slot) v = umin32(v, v - 2u * P);
    return v;
}
// signed 64-bit ring word x: with the sign bit flipped the word is x + 2^63 >= 0 = h 2^32 + l, and 2^32 = R mod P is an 18-bit
// number -- two multiply-adds fold the word to 37 bits, a shift and a multiply-subtract to below 2P; the 2^63 leaves with a constant
template <uint32_t P> __device__ __forceinline__ uint32_t res_in64(uint64_t x, bool x_slot) {
    constexpr uint64_t R = ((uint64_t)1 << 32) % P;
    constexpr uint32_t K63 = (uint32_t)((P - (((uint64_t)1 << 63) % P)) % P);
    constexpr uint64_t V2MAX = (R + 1) * R + ((uint64_t)1 << 32), QMAX = V2MAX >> 30;
    static_assert(((uint64_t)1 << 30) + QMAX * (((uint64_t)1 << 30) - P) < 2ull * P, "the folded word must land below 2P");
    const uint64_t v = (uint64_t)((uint32_t)(x >> 32) ^ 0x80000000u) * (uint32_t)R + (uint32_t)x;   // = x + 2^63 mod P, < 2^51
    const uint64_t v2 = (uint64_t)(uint32_t)(v >> 32) * (uint32_t)R + (uint32_t)v;                  // < 2^37
    uint32_t r = (uint32_t)v2 + K63 - (uint32_t)(v2 >> 30) * P;                                    // [0, 3P)
    if (x_slot) r = umin32(r, r - 2u * P);
    return r;
}
__device__ __forceinline__ Pt fwd_in(int32_t s, int e) { Pt r; r.a = res_in<P1>(s, e < 4); r.b = res_in<P2>(s, e < 4); return r; }
__device__ __forceinline__ Pt fwd_in(uint32_t x, int e) { return fwd_in((int32_t)x, e); }
__device__ __forceinline__ Pt fwd_in(uint64_t x, int e) { Pt r; r.a = res_in64<P1>(x, e < 4); r.b = res_in64<P2>(x, e < 4); return r; }
// a gadget digit (|d| < P): d + P wraps past 2^32 exactly when d < 0
template <uint32_t P> __device__ __forceinline__ uint32_t res_digit(int32_t d) { const uint32_t v = (uint32_t)d; return umin32(v, v + P); }
__device__ __forceinline__ Pt res_small(int d) { Pt r; r.a = res_digit<P1>(d); r.b = res_digit<P2>(d); return r; }
// the 32-bit pieces of a 64-bit ring word, CENTERED: w = lo + 2^32 hi mod 2^64 with lo, hi in [-2^31, 2^31) -- half the
// magnitude of unsigned pieces, i.e. one more bit of room under P / 2 for every product sum
__device__ __forceinline__ int32_t piece_of(uint64_t w, int h) {
    const int32_t lo = (int32_t)(uint32_t)w;
    return h == 0 ? lo : (int32_t)(uint32_t)((w - (uint64_t)(int64_t)lo) >> 32);
}
// residues -> the integer of least magnitude they stand for (Garner), two's complement in 64 bits
__device__ __forceinline__ uint64_t crt_signed(Pt r) {
    const uint32_t tq = shoup<P2>(subm<P2>(r.b, r.a), CRT_C, CRT_CS);   // r.a < p1 < p2
    const uint64_t x = (uint64_t)r.a + (uint64_t)P1 * tq;        // in [0, P)
    // x > P / 2  <=>  tq > (p2 - 1) / 2, or tq == (p2 - 1) / 2 and r.a > (p1 - 1) / 2  (P odd): a 32-bit test instead of a 64-bit
    // compare (the compiler turns the mask into v_cmp_i32 + v_cndmask; hiding it behind inline asm measured 2 % slower)
    const uint32_t u = 2u * tq + (((P1 - 1u) / 2u - r.a) >> 31);
    const int32_t neg = (int32_t)(P2 - 1u - u) >> 31;
    return x - (PP & (uint64_t)(int64_t)neg);
}

// constants at the tail of the table: N^-1 (+ companions) and N^-1 * 2^32 (for products of two plain operands taken with montmul),
// each followed by its product with psiinv_rev[1]
struct NttConsts { uint4 ninv[2], ninv_r[2]; };   // each: the constant c and c * w, w the twiddle of the inverse's last stage (ntt_inverse SCALED)
// tables (32-bit words): psi_rev[N] x uint4 | NttConsts (4 x uint4).  No inverse table: bfly_inv reads the forward one mirrored
template <int LOGN> __device__ __forceinline__ NttConsts tab_consts(const uint4 *tab) { NttConsts c; c.ninv[0] = tab[1 << LOGN]; c.ninv[1] = tab[(1 << LOGN) + 1]; c.ninv_r[0] = tab[(1 << LOGN) + 2]; c.ninv_r[1] = tab[(1 << LOGN) + 3]; return c; }

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
template <int LOGN> constexpr size_t lds_bytes(int ntab, int ppw = 1) { return (size_t)ppw * NttLds<LOGN>::WORDS * 8 + (TwLds<LOGN>::on ? (size_t)ntab * (1 << LOGN) * 16 : 0); }

// Batched transforms: PPW polynomials per workgroup side by side (N / 8 threads each) share the staged twiddle table, which
// lifts the number of resident waves per CU from 12 to 20 at N = 1024.
// MONT: the output goes to a resident table (keys, monomials): Montgomery form
#ifndef MKT_NTT_PPW
#define MKT_NTT_PPW 2
#endif
template <int LOGN> struct Ppw { static constexpr int v = (LOGN <= 10 && LOGN >= 6) ? MKT_NTT_PPW : 1; };
#if MKT_NTT_IN(0)
template <int LOGN, typename WORD, bool MONT>
__global__ __launch_bounds__((Ppw<LOGN>::v << (LOGN - NLR))) void ntt_fwd_kernel(const uint4 *__restrict__ tab, const WORD *__restrict__ p,
                                                                                 uint64_t *__restrict__ out, size_t B) {
    constexpr int N = 1 << LOGN, NT = N >> NLR, PPW = Ppw<LOGN>::v;
    const int sub = PPW > 1 ? threadIdx.x / NT : 0, t = PPW > 1 ? threadIdx.x % NT : threadIdx.x;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem) + (size_t)sub * NttLds<LOGN>::WORDS;
    const uint4 *tw[1]; const int which[1] = {0};
    stage_tables<LOGN, 1>(tab, reinterpret_cast<uint4 *>(reinterpret_cast<uint64_t *>(ntt_smem) + (size_t)PPW * NttLds<LOGN>::WORDS), threadIdx.x, PPW * NT, tw, which);
    const size_t groups = (B + PPW - 1) / PPW;
    // software pipeline: the words of this workgroup's NEXT polynomial are in flight while one is transformed (the loads
    // are HBM latency, ~2 us: resident waves alone do not cover it)
    auto poly_of = [&](size_t g) { const size_t b0 = g * PPW + sub; return b0 < B ? b0 : B - 1; };   // a ragged last group repeats the last polynomial (same values, same address)
    WORD nxt[8];
    if (blockIdx.x < groups) {
        const size_t b = poly_of(blockIdx.x);
#pragma unroll
        for (int e = 0; e < 8; e++) nxt[e] = __builtin_nontemporal_load(&p[b * N + e * NT + t]);
    }
    for (size_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const size_t b = poly_of(g);
        Pt z[8];
#pragma unroll
        for (int e = 0; e < 8; e++) z[e] = fwd_in(nxt[e], e);
        if (g + gridDim.x < groups) {
            const size_t bn = poly_of(g + gridDim.x);
#pragma unroll
            for (int e = 0; e < 8; e++) nxt[e] = __builtin_nontemporal_load(&p[bn * N + e * NT + t]);
        }
        ntt_forward<LOGN>(z, tw[0], lds, t);
        if (MONT) {
#pragma unroll
            for (int e = 0; e < 8; e++) { z[e].a = montmul<P1, PI1>(z[e].a, RR1); z[e].b = montmul<P2, PI2>(z[e].b, RR2); }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) z[e] = pt_canon4(z[e]);
        }
        ntt_exchange<LOGN, 0, Plan<LOGN, NLR>::lo(0)>(z, lds, t);        // thread-contiguous stores: point e*NT + t of the output order
#pragma unroll
        for (int e = 0; e < 8; e++) __builtin_nontemporal_store(pack(z[e]), &out[b * N + e * NT + t]);
    }
}
template <int LOGN, typename WORD>
__global__ __launch_bounds__((Ppw<LOGN>::v << (LOGN - NLR))) void ntt_inv_kernel(const uint4 *__restrict__ tab, const uint64_t *__restrict__ in,
                                                                                 WORD *__restrict__ p, size_t B) {
    constexpr int N = 1 << LOGN, NT = N >> NLR, PPW = Ppw<LOGN>::v;
    const int sub = PPW > 1 ? threadIdx.x / NT : 0, t = PPW > 1 ? threadIdx.x % NT : threadIdx.x;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem) + (size_t)sub * NttLds<LOGN>::WORDS;
    const uint4 *tw[1]; const int which[1] = {0};
    stage_tables<LOGN, 1>(tab, reinterpret_cast<uint4 *>(reinterpret_cast<uint64_t *>(ntt_smem) + (size_t)PPW * NttLds<LOGN>::WORDS), threadIdx.x, PPW * NT, tw, which);
    const NttConsts k = tab_consts<LOGN>(tab);
    const size_t groups = (B + PPW - 1) / PPW;
    auto poly_of = [&](size_t g) { const size_t b0 = g * PPW + sub; return b0 < B ? b0 : B - 1; };
    uint64_t nxt[8];                               // the next polynomial's points in flight (see ntt_fwd_kernel)
    if (blockIdx.x < groups) {
        const size_t b = poly_of(blockIdx.x);
#pragma unroll
        for (int e = 0; e < 8; e++) nxt[e] = __builtin_nontemporal_load(&in[b * N + e * NT + t]);
    }
    for (size_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const size_t b = poly_of(g);
        Pt z[8];
#pragma unroll
        for (int e = 0; e < 8; e++) z[e] = unpack(nxt[e]);
        if (g + gridDim.x < groups) {
            const size_t bn = poly_of(g + gridDim.x);
#pragma unroll
            for (int e = 0; e < 8; e++) nxt[e] = __builtin_nontemporal_load(&in[bn * N + e * NT + t]);
        }
        ntt_exchange<LOGN, Plan<LOGN, NLR>::lo(0), 0>(z, lds, t);
        ntt_inverse<LOGN, Plan<LOGN, NLR>::NPASS - 1>(z, tw[0], lds, t, k.ninv);
#pragma unroll
        for (int e = 0; e < 8; e++) __builtin_nontemporal_store((WORD)crt_signed(z[e]), &p[b * N + e * NT + t]);
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
    const uint4 *tw[1]; const int which[1] = {0};
    stage_tables<LOGN, 1>(tab, reinterpret_cast<uint4 *>(lds + NttLds<LOGN>::WORDS), t, NT, tw, which);
    const NttConsts k = tab_consts<LOGN>(tab);
    for (size_t b = blockIdx.x; b < B; b += gridDim.x) {
        Pt za[8];
#pragma unroll
        for (int e = 0; e < 8; e++) za[e] = fwd_in(a[b * N + e * NT + t], e);
        ntt_forward<LOGN>(za, tw[0], lds, t);
#pragma unroll
        for (int e = 0; e < 8; e++) za[e] = pt_canon4(za[e]);                          // the y of montmul
        WORD acc[8];
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = 0;
#pragma unroll
        for (int h = 0; h < H; h++) {
            Pt zb[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint64_t w = (uint64_t)bp[b * N + e * NT + t];
                zb[e] = fwd_in(W == 64 ? piece_of(w, h) : (int32_t)(uint32_t)w, e);      // centered 32-bit pieces
            }
            ntt_forward<LOGN>(zb, tw[0], lds, t);
#pragma unroll
            for (int e = 0; e < 8; e++) zb[e] = pt_mont(zb[e], za[e]);                 // x y 2^-32: undone by N^-1 2^32 below
            ntt_inverse<LOGN, Plan<LOGN, NLR>::NPASS - 1>(zb, tw[0], lds, t, k.ninv_r);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint64_t v = crt_signed(zb[e]);              // the exact integer, two's complement mod 2^64
                acc[e] = (WORD)(acc[e] + (WORD)(W == 64 ? v << (32 * h) : v));
            }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) out[b * N + e * NT + t] = acc[e];
    }
}


#endif  // unit 0

#if MKT_NTT_IN(1)
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
    const uint4 *tw[1]; const int which[1] = {0};
    stage_tables<LOGN, 1>(tab, reinterpret_cast<uint4 *>(lds + NttLds<LOGN>::WORDS), t, NT, tw, which);
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
        // the monomial row of the (first non-zero) key bit, requested a whole step ahead of its use (at use it was one exposed L2 round
        // trip per step); MKT_EXACT_MONO_AHEAD=0: at use (A/B builds)
#ifndef MKT_EXACT_MONO_AHEAD
#define MKT_EXACT_MONO_AHEAD 1
#endif
        uint64_t mr0[8];
        if constexpr (LB == 1 && MKT_EXACT_MONO_AHEAD) {
            const uint64_t *mrow = mono + (size_t)(ats[0] - 1) * N + 8 * t;
#pragma unroll
            for (int e = 0; e < 8; e++) mr0[e] = mrow[e];
        }
        for (int c = 0; c < 2; c++) {
            uint32_t tp[8];
#pragma unroll
            for (int e = 0; e < 8; e++) tp[e] = gd.prep(c ? acc[1][e] : acc[0][e]);      // :50-51 / :131-132 decompto!
            for (int j = 0; j < l; j++) {
                // the rows of the block's first key bit are requested ahead of the digit's transform (at use, each was one exposed L2
                // round trip per digit); a block's other key bits load theirs at use -- 96 accumulator registers leave no room
                uint64_t kr0[2][8];
                {
                    const uint64_t *row0 = brk + (((size_t)(blk * LB) * 2 * l + (size_t)(c * l + j)) * 2) * N + 8 * t;
#pragma unroll
                    for (int e = 0; e < 8; e++) { kr0[0][e] = row0[e]; kr0[1][e] = row0[N + e]; }
                    __builtin_amdgcn_sched_barrier(0);
                }
                Pt z[8];
#pragma unroll
                for (int e = 0; e < 8; e++) z[e] = res_small(gd.digit(tp[e], j));
                ntt_forward<LOGN>(z, tw[0], lds, t);
#ifndef MKT_EXACT_BLK_PIPE
#define MKT_EXACT_BLK_PIPE 1
#endif
                if constexpr (LB > 1 && MKT_EXACT_BLK_PIPE) {
                    // a block's other key bits: the rows of key bit q + 1 are requested before the multiply-adds of key bit q (two row
                    // buffers in turn) -- at use, each was an exposed round trip per digit and key bit
                    uint64_t kn[2][8];
#pragma unroll
                    for (int q = 0; q < LB; q++) {
                        uint64_t kc[2][8];
#pragma unroll
                        for (int e = 0; e < 8; e++) { kc[0][e] = q == 0 ? kr0[0][e] : kn[0][e]; kc[1][e] = q == 0 ? kr0[1][e] : kn[1][e]; }
                        if (q + 1 < LB) {
                            const uint64_t *row = brk + (((size_t)(blk * LB + q + 1) * 2 * l + (size_t)(c * l + j)) * 2) * N + 8 * t;
#pragma unroll
                            for (int e = 0; e < 8; e++) { kn[0][e] = row[e]; kn[1][e] = row[N + e]; }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (ats[q] == 0) continue;
#pragma unroll
                        for (int e = 0; e < 8; e++) {                    // :63-68 / :146-154, exactly
                            tacc[q][0][e] = pt_mac(tacc[q][0][e], z[e], unpack(kc[0][e]));
                            tacc[q][1][e] = pt_mac(tacc[q][1][e], z[e], unpack(kc[1][e]));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                for (int q = 0; q < LB; q++) {
                    if (ats[q] == 0) continue;
                    const uint64_t *row = brk + (((size_t)(blk * LB + q) * 2 * l + (size_t)(c * l + j)) * 2) * N + 8 * t;
#pragma unroll
                    for (int e = 0; e < 8; e++) {                        // :63-68 / :146-154, exactly
                        tacc[q][0][e] = pt_mac(tacc[q][0][e], z[e], unpack(q == 0 ? kr0[0][e] : row[e]));
                        tacc[q][1][e] = pt_mac(tacc[q][1][e], z[e], unpack(q == 0 ? kr0[1][e] : row[N + e]));
                    }
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
                for (int e = 0; e < 8; e++) s2[e] = pt_mac(s2[e], tacc[q][pp][e], unpack((LB == 1 && MKT_EXACT_MONO_AHEAD) ? mr0[e] : mrow[e]));   // :71 / :157
            }
            ntt_inverse<LOGN, Plan<LOGN, NLR>::NPASS - 1>(s2, tw[0], lds, t, k.ninv);            // :72 / :162
#pragma unroll
            for (int e = 0; e < 8; e++) acc[pp][e] += (uint32_t)crt_signed(s2[e]);   // :73 / :163
        }
    }
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int e = 0; e < 8; e++) accg[c * N + e * NT + t] = acc[c][e];
}

// ------------------------------------------------------------------------------------------------
// The same rotation for any RLWE length KR (accumulator (b, a_0 .. a_{KR-1}); bootstrapping.jl:32-76 / :114-165 with k > 1:
// TFHEparams_bin.k / TFHEparams_block.k, scheme.jl:6-36 -- BASELINE configs[4] is LMSS with k = 2) and any block length.
// Per key bit: the (KR + 1) l digit polynomials of the accumulator are transformed and multiplied into the key bit's
// (KR + 1) l x (KR + 1) rows (:62-68 / :146-154); a block sums its key bits' products times their monomials in the transform
// domain (:157) and lifts ONCE per output polynomial (:162).  The digit transforms are recomputed per key bit of a block (the
// accumulator does not change inside a block: same values) -- (KR + 1) x 8 point accumulators for one key bit and for the
// block are what the registers hold.  True coefficients stay below blk_len * 2 (KR + 1) l N 2^(logB-1) 2^31 < P / 2 (host check).
// ------------------------------------------------------------------------------------------------
template <int LOGN, int KR>
__global__ __launch_bounds__((1 << (LOGN - NLR))) void exact_blindrotate_kr_kernel(const uint4 *__restrict__ tab, const uint64_t *__restrict__ brk,
                                                                                 const uint64_t *__restrict__ mono, const uint32_t *__restrict__ lwe,
                                                                                 int lwe_stride, int pre_switched, int n, int l, int logB, int blk_len,
                                                                                 uint32_t *__restrict__ acc_io) {
    constexpr int N = 1 << LOGN, NT = N >> NLR, NP = KR + 1;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem);
    const int t = threadIdx.x;
    const uint4 *tw[1]; const int which[1] = {0};
    stage_tables<LOGN, 1>(tab, reinterpret_cast<uint4 *>(lds + NttLds<LOGN>::WORDS), t, NT, tw, which);
    const NttConsts k = tab_consts<LOGN>(tab);
    const size_t rot = blockIdx.x;
    const uint32_t *at_src = lwe + rot * (size_t)lwe_stride;
    uint32_t *accg = acc_io + rot * NP * (size_t)N;
    const Gadget<uint32_t> gd(l, logB);
    uint32_t acc[NP][8];
#pragma unroll
    for (int c = 0; c < NP; c++)
#pragma unroll
        for (int e = 0; e < 8; e++) acc[c][e] = accg[c * N + e * NT + t];
    const int msbit = 32 - LOGN - 1;
    for (int blk = 0; blk < n / blk_len; blk++) {
        bool any = false;
        for (int q = 0; q < blk_len; q++) {
            const uint32_t v0 = at_src[blk * blk_len + q];
            any |= (pre_switched ? v0 : divbits<uint32_t>(v0, msbit)) != 0;
        }
        if (!__builtin_amdgcn_readfirstlane((int)any)) continue;                   // :48 / :145 (an all-zero block adds 0)
        Pt sum[NP][8];
#pragma unroll
        for (int pp = 0; pp < NP; pp++)
#pragma unroll
            for (int e = 0; e < 8; e++) { sum[pp][e].a = 0; sum[pp][e].b = 0; }
        if (blk_len > 1) {
            // a block of several key bits: ONE set of digit transforms; sum_q mono_q (*) (sum_g z_g (*) K_qg) = sum_g sum_q (mono_q (*) z_g) (*) K_qg
            // in exact integers mod P -- each transform is multiplied by the key bit's monomial row, then into the key bit's rows, straight
            // into the block's sums (no per-key-bit accumulators, no second transform of the same digits)
#pragma unroll
            for (int c = 0; c < NP; c++)
                for (int j = 0; j < l; j++) {
                    Pt z[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) z[e] = res_small(gd.digit(gd.prep(acc[c][e]), j));   // :131-132 decompto!
                    ntt_forward<LOGN>(z, tw[0], lds, t);
                    for (int q = 0; q < blk_len; q++) {
                        const int i = blk * blk_len + q;
                        const uint32_t v0 = at_src[i];
                        const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pre_switched ? v0 : divbits<uint32_t>(v0, msbit)));
                        if (at == 0) continue;                                     // :145
                        const uint64_t *mrow = mono + (size_t)(at - 1) * N + 8 * t;
                        const uint64_t *row = brk + (((size_t)i * NP * l + (size_t)(c * l + j)) * NP) * N + 8 * t;   // [row c l + j][poly][N]
                        Pt zm[8];
#pragma unroll
                        for (int e = 0; e < 8; e++) zm[e] = pt_mont(z[e], unpack(mrow[e]));
#pragma unroll
                        for (int pp = 0; pp < NP; pp++)
#pragma unroll
                            for (int e = 0; e < 8; e++) sum[pp][e] = pt_mac(sum[pp][e], zm[e], unpack(row[(size_t)pp * N + e]));   // :146-157, re-associated
                    }
                }
        } else
        for (int q = 0; q < blk_len; q++) {
            const int i = blk * blk_len + q;
            const uint32_t v0 = at_src[i];
            const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pre_switched ? v0 : divbits<uint32_t>(v0, msbit)));
            if (at == 0) continue;
            Pt tacc[NP][8];
#pragma unroll
            for (int pp = 0; pp < NP; pp++)
#pragma unroll
                for (int e = 0; e < 8; e++) { tacc[pp][e].a = 0; tacc[pp][e].b = 0; }
#pragma unroll
            for (int c = 0; c < NP; c++)
                for (int j = 0; j < l; j++) {
                    const uint64_t *row = brk + (((size_t)i * NP * l + (size_t)(c * l + j)) * NP) * N + 8 * t;   // [row c l + j][poly][N]
                    uint64_t kr0[8];                                               // the first polynomial's row ahead of the transform
#pragma unroll
                    for (int e = 0; e < 8; e++) kr0[e] = row[e];
                    __builtin_amdgcn_sched_barrier(0);
                    Pt z[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) z[e] = res_small(gd.digit(gd.prep(acc[c][e]), j));   // :50-51 / :131-132 decompto!
                    ntt_forward<LOGN>(z, tw[0], lds, t);
#pragma unroll
                    for (int pp = 0; pp < NP; pp++)
#pragma unroll
                        for (int e = 0; e < 8; e++) tacc[pp][e] = pt_mac(tacc[pp][e], z[e], unpack(pp == 0 ? kr0[e] : row[(size_t)pp * N + e]));   // :63-68 / :146-154, exactly
                }
            const uint64_t *mrow = mono + (size_t)(at - 1) * N + 8 * t;
#pragma unroll
            for (int pp = 0; pp < NP; pp++)
#pragma unroll
                for (int e = 0; e < 8; e++) sum[pp][e] = pt_mac(sum[pp][e], tacc[pp][e], unpack(mrow[e]));       // :71 / :157
        }
#pragma unroll
        for (int pp = 0; pp < NP; pp++) {
            ntt_inverse<LOGN, Plan<LOGN, NLR>::NPASS - 1>(sum[pp], tw[0], lds, t, k.ninv);                       // :72 / :162
#pragma unroll
            for (int e = 0; e < 8; e++) acc[pp][e] += (uint32_t)crt_signed(sum[pp][e]);                           // :73 / :163
        }
    }
#pragma unroll
    for (int c = 0; c < NP; c++)
#pragma unroll
        for (int e = 0; e < 8; e++) accg[c * N + e * NT + t] = acc[c][e];
}

// ------------------------------------------------------------------------------------------------
// Any RLWE length beyond 3 (scheme.jl:6-36 leaves k free; the Float64 mode's counterpart is blindrotate_kany_kernel): np = k + 1 is a
// run-time value, so the transform-domain sums of one key bit (tacc, :62-68 / :146-154) and of the block (:71 / :157) live in a
// per-rotation scratch region [2][np][N] of packed residue pairs and the accumulator stays in its ring-word buffer.  Every thread
// reads and writes only its own elements of all three (words e*NT + t, transform points 8t + e), so no barrier beyond the
// transforms' own is needed.  Same products, same single lift per output polynomial and block as exact_blindrotate_kr_kernel:
// word-identical to it where both run (tests force this kernel at k <= 3).
// ------------------------------------------------------------------------------------------------
template <int LOGN>
__global__ __launch_bounds__((1 << (LOGN - NLR))) void exact_blindrotate_kany_kernel(const uint4 *__restrict__ tab, const uint64_t *__restrict__ brk,
                                                                                   const uint64_t *__restrict__ mono, const uint32_t *__restrict__ lwe,
                                                                                   int lwe_stride, int pre_switched, int n, int np, int l, int logB, int blk_len,
                                                                                   uint32_t *__restrict__ acc_io, uint64_t *__restrict__ scratch) {
    constexpr int N = 1 << LOGN, NT = N >> NLR;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem);
    const int t = threadIdx.x;
    const uint4 *tw[1]; const int which[1] = {0};
    stage_tables<LOGN, 1>(tab, reinterpret_cast<uint4 *>(lds + NttLds<LOGN>::WORDS), t, NT, tw, which);
    const NttConsts k = tab_consts<LOGN>(tab);
    const size_t rot = blockIdx.x;
    const uint32_t *at_src = lwe + rot * (size_t)lwe_stride;
    uint32_t *accg = acc_io + rot * (size_t)np * N;
    uint64_t *tg = scratch + rot * (size_t)2 * np * N + 8 * t, *sg = tg + (size_t)np * N;     // this thread's points of tacc / of the block sum
    const Gadget<uint32_t> gd(l, logB);
    const int msbit = 32 - LOGN - 1;
    for (int blk = 0; blk < n / blk_len; blk++) {
        bool first_bit = true;                                                     // the block sum starts at this key bit's product (:157 from zero)
        for (int q = 0; q < blk_len; q++) {
            const int i = blk * blk_len + q;
            const uint32_t v0 = at_src[i];
            const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pre_switched ? v0 : divbits<uint32_t>(v0, msbit)));
            if (at == 0) continue;                                                 // :48 / :145
            for (int c = 0; c < np; c++)
                for (int j = 0; j < l; j++) {
                    const uint64_t *row = brk + (((size_t)i * np * l + (size_t)(c * l + j)) * np) * N + 8 * t;   // [row c l + j][poly][N]
                    Pt z[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) z[e] = res_small(gd.digit(gd.prep(accg[(size_t)c * N + e * NT + t]), j));   // :50-51 / :131-132 decompto!
                    ntt_forward<LOGN>(z, tw[0], lds, t);
                    const bool first_row = c == 0 && j == 0;
                    for (int pp = 0; pp < np; pp++) {
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            Pt cur; cur.a = 0; cur.b = 0;
                            if (!first_row) cur = unpack(tg[(size_t)pp * N + e]);
                            tg[(size_t)pp * N + e] = pack(pt_mac(cur, z[e], unpack(row[(size_t)pp * N + e])));   // :63-68 / :146-154, exactly
                        }
                    }
                }
            const uint64_t *mrow = mono + (size_t)(at - 1) * N + 8 * t;
            for (int pp = 0; pp < np; pp++) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    Pt cur; cur.a = 0; cur.b = 0;
                    if (!first_bit) cur = unpack(sg[(size_t)pp * N + e]);
                    sg[(size_t)pp * N + e] = pack(pt_mac(cur, unpack(tg[(size_t)pp * N + e]), unpack(mrow[e])));   // :71 / :157
                }
            }
            first_bit = false;
        }
        if (first_bit) continue;                                                   // an all-zero block adds 0
        for (int pp = 0; pp < np; pp++) {
            Pt s[8];
#pragma unroll
            for (int e = 0; e < 8; e++) s[e] = unpack(sg[(size_t)pp * N + e]);
            ntt_inverse<LOGN, Plan<LOGN, NLR>::NPASS - 1>(s, tw[0], lds, t, k.ninv);                             // :72 / :162
#pragma unroll
            for (int e = 0; e < 8; e++) accg[(size_t)pp * N + e * NT + t] += (uint32_t)crt_signed(s[e]);         // :73 / :163
        }
    }
}

#endif  // unit 1

// ------------------------------------------------------------------------------------------------
// The 64-bit ring (KMS) with exact products.  A product  digit polynomial x 64-bit polynomial  exceeds P, so every resident
// 64-bit table is kept as TWO residue polynomials -- the transforms of its low and of its high 32-bit halves (centered
// pieces in [-2^31, 2^31), piece_of) -- every sum of such products as a (low, high) pair of transform-domain accumulators,
// and every inverse runs twice:  sum_j d_j T_j = [sum_j d_j lo(T_j)] + 2^32 [sum_j d_j hi(T_j)]  mod 2^64,  each bracket an
// exact integer below P / 2 (count * N * 2^(logB-1) * 2^31, twice that where a monomial X^a - 1 multiplies in the transform
// domain; checked on the host for the context's gadgets, exact_gate_ok).
// Layout of a split table: logical polynomial i -> residue polynomials 2i (low half) and 2i + 1 (high half), natural
// transform order, Montgomery form.  The monomial table (coefficients -2 .. 1) is not split.
// ------------------------------------------------------------------------------------------------
#if MKT_NTT_IN(0)
template <int LOGN>
__global__ __launch_bounds__((1 << (LOGN - NLR))) void ntt_fwd_split_kernel(const uint4 *__restrict__ tab, const uint64_t *__restrict__ p,
                                                                           uint64_t *__restrict__ out, size_t B) {
    constexpr int N = 1 << LOGN, NT = N >> NLR;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem);
    const int t = threadIdx.x;
    const uint4 *tw[1]; const int which[1] = {0};
    stage_tables<LOGN, 1>(tab, reinterpret_cast<uint4 *>(lds + NttLds<LOGN>::WORDS), t, NT, tw, which);
    for (size_t b = blockIdx.x; b < B; b += gridDim.x) {
        uint64_t w[8];
#pragma unroll
        for (int e = 0; e < 8; e++) w[e] = p[b * N + e * NT + t];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            Pt z[8];
#pragma unroll
            for (int e = 0; e < 8; e++) z[e] = fwd_in(piece_of(w[e], h), e);
            ntt_forward<LOGN>(z, tw[0], lds, t);
#pragma unroll
            for (int e = 0; e < 8; e++) out[(2 * b + h) * N + 8 * t + e] = pack(Pt{montmul<P1, PI1>(z[e].a, RR1), montmul<P2, PI2>(z[e].b, RR2)});
            __syncthreads();
        }
    }
}

#endif  // unit 0

// the exact integers behind a (low, high) accumulator pair, combined mod 2^64: inverse transforms, N^-1, Garner lift
#ifndef MKT_EXACT_WPE
#define MKT_EXACT_WPE 2
#endif
template <int LOGN>
__device__ __forceinline__ void lift_pair(Pt (&x)[2][8], uint64_t (&w)[8], const uint4 *psi, const NttConsts &k, uint64_t *lds, int t) {
    ntt_inverse_n<LOGN, Plan<LOGN, NLR>::NPASS - 1, false, 2>(x, psi, lds, t, k.ninv);      // the two halves side by side (two staging buffers): one set of twiddle reads, addresses and barriers
#pragma unroll
    for (int e = 0; e < 8; e++) w[e] = crt_signed(x[0][e]) + (crt_signed(x[1][e]) << 32);
}

#if MKT_NTT_IN(2)
// KMS phase 1 (bootstrapping.jl:389-443) with exact products: one workgroup per RLEV row rotation, accumulator (b, a) in
// registers (slot e = coefficient e*NT + t); output: the row's two polynomials as split residue tables for phase 2
template <int LOGN, bool BLK>
__global__ __launch_bounds__((1 << (LOGN - NLR))) __attribute__((amdgpu_waves_per_eu(MKT_EXACT_WPE, MKT_EXACT_WPE))) void exact_kms_phase1_kernel(const uint4 *__restrict__ tab, const uint64_t *__restrict__ brk0, size_t brk_party_stride,
                                                                              const uint64_t *__restrict__ mono, const uint32_t *__restrict__ lwe, int lwe_stride,
                                                                              int pre_switched, int n, int l, int logB, int blk_len, size_t ngates, int rows_per_gate,
                                                                              const int *__restrict__ slot_party, const int *__restrict__ slot_row, int logB_lev,
                                                                              uint64_t *__restrict__ lev_out) {
    constexpr int N = 1 << LOGN, NT = N >> NLR;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem);
    const int t = threadIdx.x;
    const uint4 *tw[1]; const int which[1] = {0};
    stage_tables<LOGN, 1>(tab, reinterpret_cast<uint4 *>(lds + 2 * NttLds<LOGN>::WORDS), t, NT, tw, which);   // two staging buffers (lift_pair), then the table
    const NttConsts k = tab_consts<LOGN>(tab);
    const size_t gate = blockIdx.x % ngates;
    const int slot = (int)(blockIdx.x / ngates);
    const size_t rot = gate * (size_t)rows_per_gate + slot;
    const int party = slot_party[slot], row = slot_row[slot];
    const uint32_t *at_src = lwe + gate * (size_t)lwe_stride + (size_t)party * n;
    const uint64_t *brk = brk0 + (size_t)party * brk_party_stride;
    const Gadget<uint64_t> gd(l, logB);
    uint64_t acc[2][8];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int e = 0; e < 8; e++) acc[c][e] = 0;
    if (t == 0) acc[0][0] = (uint64_t)1 << (64 - (row + 1) * logB_lev);           // :403-406 trivial RLEV row
    const int msbit = 32 - LOGN - 1;
    // blk_len = 1: bootstrapping.jl:411-438; > 1 (KMS_block, :623-655): one decomposition per block of key bits, every key bit's
    // product times its monomial summed in the transform domain (:648), one inverse per block.  The digit transforms are
    // recomputed per key bit (same values) instead of being held for the block: registers, not arithmetic, are short here.
    for (int blk = 0; blk < n / blk_len; blk++) {
        bool any = false;
        for (int q = 0; q < blk_len; q++) {
            const uint32_t v0 = at_src[blk * blk_len + q];
            any |= (pre_switched ? v0 : divbits<uint32_t>(v0, msbit)) != 0;
        }
        if (!__builtin_amdgcn_readfirstlane((int)any)) continue;                   // :413 / :638
        Pt sum[2][2][8];                                                           // [output polynomial][half]; BLK only (one key bit: the product itself)
        if (BLK) {
#pragma unroll
            for (int pp = 0; pp < 2; pp++)
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int e = 0; e < 8; e++) { sum[pp][h][e].a = 0; sum[pp][h][e].b = 0; }
        }
        for (int q = 0; q < (BLK ? blk_len : 1); q++) {
            const int i = blk * blk_len + q;
            const uint32_t v0 = at_src[i];
            const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pre_switched ? v0 : divbits<uint32_t>(v0, msbit)));
            if (at == 0) continue;
            Pt tacc[2][2][8];
#pragma unroll
            for (int pp = 0; pp < 2; pp++)
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int e = 0; e < 8; e++) { tacc[pp][h][e].a = 0; tacc[pp][h][e].b = 0; }
            for (int c = 0; c < 2; c++)
                for (int j = 0; j < l; j++) {
                    // the digit's four key rows are REQUESTED here and consumed after its transform: left to the compiler they were loaded
                    // at use, four rows x one exposed L2 round trip each per digit (16 per CMux at two waves per SIMD)
                    const uint64_t *rowp = brk + (((size_t)i * 2 * l + (size_t)(c * l + j)) * 4) * N + 8 * t;   // [poly][half][N]
                    // (all four rows ahead: 108 spilled registers.  Two ahead, the other two requested right after the transform, under the
                    // first two rows' multiply-adds.  KMS_block keeps loading at use: its second accumulator set leaves no room.)
                    constexpr bool PF = !BLK;
                    uint64_t kr[2][8], ks[2][8];
                    if constexpr (PF) {
#pragma unroll
                        for (int ph = 0; ph < 2; ph++)
#pragma unroll
                            for (int e = 0; e < 8; e++) kr[ph][e] = rowp[(size_t)ph * N + e];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    Pt z[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) z[e] = res_small(gd.digit(gd.prep(c ? acc[1][e] : acc[0][e]), j));   // :415-425 / :625-633 decompto! (the rounding offset re-applied per digit: two instructions, sixteen registers less)
                    ntt_forward<LOGN>(z, tw[0], lds, t);
                    if constexpr (PF) {
#pragma unroll
                        for (int ph = 0; ph < 2; ph++)
#pragma unroll
                            for (int e = 0; e < 8; e++) ks[ph][e] = rowp[(size_t)(2 + ph) * N + e];
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int pp = 0; pp < 2; pp++)
#pragma unroll
                        for (int h = 0; h < 2; h++)
#pragma unroll
                            for (int e = 0; e < 8; e++)                                // :427-432 / :639-646, exactly
                                tacc[pp][h][e] = pt_mac(tacc[pp][h][e], z[e], unpack(PF ? (pp == 0 ? kr[h][e] : ks[h][e]) : rowp[(size_t)(pp * 2 + h) * N + e]));
                }
            if (BLK) {
                const uint64_t *mrow = mono + (size_t)(at - 1) * N + 8 * t;
#pragma unroll
                for (int pp = 0; pp < 2; pp++)
#pragma unroll
                    for (int h = 0; h < 2; h++)
#pragma unroll
                        for (int e = 0; e < 8; e++) sum[pp][h][e] = pt_mac(sum[pp][h][e], tacc[pp][h][e], unpack(mrow[e]));   // :648
            } else {
                // one key bit: lift the product sum S itself and apply X^at - 1 on the integers (a rotation through LDS) -- the
                // same words mod 2^64, and the lifted sum is bounded by 2 l N 2^(logB-1) 2^31 instead of twice that: what lets
                // the gadget l = 2, base 2^16 at N = 1024 (BASELINE configs[1]) fit two 30-bit primes
#pragma unroll
                for (int pp = 0; pp < 2; pp++) {
                    uint64_t w[8];
                    lift_pair<LOGN>(tacc[pp], w, tw[0], k, lds, t);
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < 8; e++) lds[e * NT + t] = w[e];
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < 8; e++) {                                  // :435-437: (X^at S)[i] = +-S[i - at mod N]
                        const uint32_t src = (uint32_t)(e * NT + t - (int)at) & (2u * N - 1u);
                        const uint64_t v = lds[src & (N - 1)];
                        acc[pp][e] += (src >= (uint32_t)N ? (uint64_t)0 - v : v) - w[e];
                    }
                }
            }
        }
        if (BLK)
#pragma unroll
        for (int pp = 0; pp < 2; pp++) {
            uint64_t w[8];
            lift_pair<LOGN>(sum[pp], w, tw[0], k, lds, t);          // :436 / :653
#pragma unroll
            for (int e = 0; e < 8; e++) acc[pp][e] += w[e];                        // :437 / :654
        }
    }
    // :441 fftto!(tacc, acc): the row as split residue tables, Montgomery form
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            Pt z[8];
#pragma unroll
            for (int e = 0; e < 8; e++) z[e] = fwd_in(piece_of(acc[c][e], h), e);
            ntt_forward<LOGN>(z, tw[0], lds, t);
            uint64_t *o = lev_out + ((rot * 2 + c) * 2 + h) * (size_t)N + 8 * t;
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = pack(Pt{montmul<P1, PI1>(z[e].a, RR1), montmul<P2, PI2>(z[e].b, RR2)});
            __syncthreads();
        }
}

// KMS_block phase 1 (bootstrapping.jl:599-659) with ONE set of digit transforms per block.  The block adds
//     sum_q mono[at_q] (*) ( sum_g z_g (*) K_{q,g} )        (:639-648; q the block's key bits, g the 2l digit polynomials)
// in exact integers mod P, so the same value is   sum_g sum_q (mono[at_q] (*) z_g) (*) K_{q,g}:  each digit transform is multiplied by
// the key bit's monomial row first (one product per key bit and digit) and then into the key bit's four rows, straight into the block's
// four accumulators -- 2l forward transforms per block instead of blk_len x 2l (exact_kms_phase1_kernel<.., BLK> transforms the digits
// again for every key bit: its per-key-bit accumulators leave no room to keep them), no per-key-bit accumulators, the same register
// footprint.  Word-identical to that kernel (the Float64 mode cannot re-associate: every rounding is pinned there).
template <int LOGN>
__global__ __launch_bounds__((1 << (LOGN - NLR))) __attribute__((amdgpu_waves_per_eu(MKT_EXACT_WPE, MKT_EXACT_WPE))) void exact_kms_block_phase1_kernel(const uint4 *__restrict__ tab, const uint64_t *__restrict__ brk0, size_t brk_party_stride,
                                                                              const uint64_t *__restrict__ mono, const uint32_t *__restrict__ lwe, int lwe_stride,
                                                                              int pre_switched, int n, int l, int logB, int blk_len, size_t ngates, int rows_per_gate,
                                                                              const int *__restrict__ slot_party, const int *__restrict__ slot_row, int logB_lev,
                                                                              uint64_t *__restrict__ lev_out) {
    constexpr int N = 1 << LOGN, NT = N >> NLR;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem);
    const int t = threadIdx.x;
    const uint4 *tw[1]; const int which[1] = {0};
    stage_tables<LOGN, 1>(tab, reinterpret_cast<uint4 *>(lds + 2 * NttLds<LOGN>::WORDS), t, NT, tw, which);   // two staging buffers (lift_pair), then the table
    const NttConsts k = tab_consts<LOGN>(tab);
    const size_t gate = blockIdx.x % ngates;
    const int slot = (int)(blockIdx.x / ngates);
    const size_t rot = gate * (size_t)rows_per_gate + slot;
    const int party = slot_party[slot], row = slot_row[slot];
    const uint32_t *at_src = lwe + gate * (size_t)lwe_stride + (size_t)party * n;
    const uint64_t *brk = brk0 + (size_t)party * brk_party_stride;
    const Gadget<uint64_t> gd(l, logB);
    uint64_t acc[2][8];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int e = 0; e < 8; e++) acc[c][e] = 0;
    if (t == 0) acc[0][0] = (uint64_t)1 << (64 - (row + 1) * logB_lev);           // :609-612 trivial RLEV row
    const int msbit = 32 - LOGN - 1;
    for (int blk = 0; blk < n / blk_len; blk++) {
        bool any = false;
        for (int q = 0; q < blk_len; q++) {
            const uint32_t v0 = at_src[blk * blk_len + q];
            any |= (pre_switched ? v0 : divbits<uint32_t>(v0, msbit)) != 0;
        }
        if (!__builtin_amdgcn_readfirstlane((int)any)) continue;                   // :638 for every key bit of the block
        Pt sum[2][2][8];                                                           // [output polynomial][half]
#pragma unroll
        for (int pp = 0; pp < 2; pp++)
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int e = 0; e < 8; e++) { sum[pp][h][e].a = 0; sum[pp][h][e].b = 0; }
        for (int c = 0; c < 2; c++)
            for (int j = 0; j < l; j++) {
                Pt z[8];
#pragma unroll
                for (int e = 0; e < 8; e++) z[e] = res_small(gd.digit(gd.prep(c ? acc[1][e] : acc[0][e]), j));   // :625-633 decompto!
                ntt_forward<LOGN>(z, tw[0], lds, t);
                for (int q = 0; q < blk_len; q++) {
                    const int i = blk * blk_len + q;
                    const uint32_t v0 = at_src[i];
                    const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pre_switched ? v0 : divbits<uint32_t>(v0, msbit)));
                    if (at == 0) continue;                                         // :638 per key bit
                    const uint64_t *mrow = mono + (size_t)(at - 1) * N + 8 * t;
                    const uint64_t *rowp = brk + (((size_t)i * 2 * l + (size_t)(c * l + j)) * 4) * N + 8 * t;   // [poly][half][N]
                    Pt zm[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) zm[e] = pt_mont(z[e], unpack(mrow[e]));   // mono (*) z: canonical, the x of the multiply-adds below
#pragma unroll
                    for (int pp = 0; pp < 2; pp++)
#pragma unroll
                        for (int h = 0; h < 2; h++)
#pragma unroll
                            for (int e = 0; e < 8; e++) sum[pp][h][e] = pt_mac(sum[pp][h][e], zm[e], unpack(rowp[(size_t)(pp * 2 + h) * N + e]));   // :639-648, re-associated
                }
            }
#pragma unroll
        for (int pp = 0; pp < 2; pp++) {
            uint64_t w[8];
            lift_pair<LOGN>(sum[pp], w, tw[0], k, lds, t);                         // :653
#pragma unroll
            for (int e = 0; e < 8; e++) acc[pp][e] += w[e];                        // :654
        }
    }
    // :657 fftto!(tacc, acc): the row as split residue tables, Montgomery form
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            Pt z[8];
#pragma unroll
            for (int e = 0; e < 8; e++) z[e] = fwd_in(piece_of(acc[c][e], h), e);
            ntt_forward<LOGN>(z, tw[0], lds, t);
            uint64_t *o = lev_out + ((rot * 2 + c) * 2 + h) * (size_t)N + 8 * t;
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = pack(Pt{montmul<P1, PI1>(z[e].a, RR1), montmul<P2, PI2>(z[e].b, RR2)});
            __syncthreads();
        }
}

// KMS phase 1 at l_gsw = 2 with PAIRED transforms (two waves per SIMD, the whole register file; exact_wide != 0, the default): the digit polynomials of one
// accumulator (j = 0, 1) go through the forward transform side by side, the low and the high half of an output polynomial's lifted sum through the
// inverse -- half the twiddle reads, slot addresses, barriers and exposed waits of the one-at-a-time form (exact_kms_phase1_kernel, exact_wide = 0) --
// every sum gathers its four digit products in 64 bits (one v_mad_u64_u32 per term and residue, one reduction), and the two halves of a sum arrive
// together, so X^at - 1 is applied once per output polynomial on the 64-bit words.  The key rows of the FIRST (polynomial, half) are requested ahead:
// its 16 row pieces per thread are asked for before the second forward pair instead of at use, where the wait was an exposed L2 / fabric round trip
// (key rows: 13 % of the step, profiles/r05_experiments.txt item 4).  The other three sums load at use: asked ahead too the kernel spills 60-100
// registers and runs 7-22 % SLOWER.  (Rounds 3-5 also carried this kernel without the early request, a three-waves-per-SIMD form and a 64-bit-gathering
// form of the one-at-a-time kernel -- all superseded by this one in the round-5 A/B and removed in round 6: docs/history.md.)
template <int LOGN>
__global__ __launch_bounds__((1 << (LOGN - NLR))) __attribute__((amdgpu_waves_per_eu(2, 2))) void exact_kms_phase1_p2pf_kernel(const uint4 *__restrict__ tab, const uint64_t *__restrict__ brk0, size_t brk_party_stride,
                                                                              const uint32_t *__restrict__ lwe, int lwe_stride, int pre_switched, int n, int logB, size_t ngates, int rows_per_gate,
                                                                              const int *__restrict__ slot_party, const int *__restrict__ slot_row, int logB_lev,
                                                                              uint64_t *__restrict__ lev_out) {
    constexpr int N = 1 << LOGN, NT = N >> NLR;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem);                        // two staging buffers, then the table
    const int t = threadIdx.x;
    const uint4 *tw[1]; const int which[1] = {0};
    stage_tables<LOGN, 1>(tab, reinterpret_cast<uint4 *>(lds + 2 * NttLds<LOGN>::WORDS), t, NT, tw, which);
    const NttConsts k = tab_consts<LOGN>(tab);
    const size_t gate = blockIdx.x % ngates;
    const int slot = (int)(blockIdx.x / ngates);
    const size_t rot = gate * (size_t)rows_per_gate + slot;
    const int party = slot_party[slot], row = slot_row[slot];
    const uint32_t *at_src = lwe + gate * (size_t)lwe_stride + (size_t)party * n;
    const uint64_t *brk = brk0 + (size_t)party * brk_party_stride;
    const Gadget<uint64_t> gd(2, logB);
    uint64_t acc[2][8];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int e = 0; e < 8; e++) acc[c][e] = 0;
    if (t == 0) acc[0][0] = (uint64_t)1 << (64 - (row + 1) * logB_lev);           // :403-406 trivial RLEV row
    const int msbit = 32 - LOGN - 1;
    for (int i = 0; i < n; i++) {
        const uint32_t v0 = at_src[i];
        const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pre_switched ? v0 : divbits<uint32_t>(v0, msbit)));
        if (at == 0) continue;                                                     // :413
        const uint4 *rowb = reinterpret_cast<const uint4 *>(brk + ((size_t)i * 4 * 4) * N + 8 * t);   // [digit g][poly][half][N], two points per 16 bytes
        // piece (g, ep) of (polynomial pp, half h): K[g * 4 + ep]
        auto ask = [&](uint4 (&K)[16], int pp, int h) {
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int ep = 0; ep < 4; ep++) K[g * 4 + ep] = rowb[(size_t)(g * 4 + pp * 2 + h) * (N / 2) + ep];
            __builtin_amdgcn_sched_barrier(0);
        };
        Pt zz[4][8];
        uint4 Ka[16], Kb[16];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            Pt (&zp)[2][8] = *reinterpret_cast<Pt(*)[2][8]>(&zz[2 * c]);
            if (c == 1) ask(Ka, 0, 0);                                            // (0, low): under the second forward pair
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 8; e++) zp[j][e] = res_small(gd.digit(gd.prep(c ? acc[1][e] : acc[0][e]), j));   // :415-425 decompto!
            ntt_forward_n<LOGN, 0, false, 2>(zp, tw[0], lds, t);
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 8; e++) zp[j][e] = wide_x(zp[j][e]);
        }
        // the sum of one (polynomial, half) from the pieces in K (:427-432), two points at a time
        auto gather = [&](Pt (&th)[8], const uint4 (&K)[16]) {
#pragma unroll
            for (int ep = 0; ep < 4; ep++) {
                Wide w0, w1;
                w0.a = w0.b = w1.a = w1.b = 0;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const uint4 kv = K[g * 4 + ep];
                    Pt y0, y1; y0.a = kv.x; y0.b = kv.y; y1.a = kv.z; y1.b = kv.w;
                    wide_mac(w0, zz[g][2 * ep], y0); wide_mac(w1, zz[g][2 * ep + 1], y1);
                }
                th[2 * ep] = wide_reduce(w0); th[2 * ep + 1] = wide_reduce(w1);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto lift_rotate = [&](Pt (&th)[2][8], auto ppc) {                        // inverse pair, lift, X^at - 1 (:435-437)
            constexpr int pp = decltype(ppc)::value;
            ntt_inverse_n<LOGN, Plan<LOGN, NLR>::NPASS - 1, false, 2>(th, tw[0], lds, t, k.ninv);
            uint64_t w[8];
#pragma unroll
            for (int e = 0; e < 8; e++) w[e] = crt_signed(th[0][e]) + (crt_signed(th[1][e]) << 32);
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; e++) lds[e * NT + t] = w[e];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint32_t src = (uint32_t)(e * NT + t - (int)at) & (2u * N - 1u);
                const uint64_t v = lds[src & (N - 1)];
                acc[pp][e] += (src >= (uint32_t)N ? (uint64_t)0 - v : v) - w[e];
            }
        };
        {
            Pt th[2][8];
            gather(th[0], Ka);
            ask(Kb, 0, 1);
            gather(th[1], Kb);
            lift_rotate(th, std::integral_constant<int, 0>{});
        }
        {
            Pt th[2][8];
            ask(Ka, 1, 0);
            gather(th[0], Ka);
            ask(Kb, 1, 1);
            gather(th[1], Kb);
            lift_rotate(th, std::integral_constant<int, 1>{});
        }
    }
    // :441 fftto!(tacc, acc): the row as split residue tables, Montgomery form; the two halves of a polynomial side by side
#pragma unroll
    for (int c = 0; c < 2; c++) {
        Pt z[2][8];
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int e = 0; e < 8; e++) z[h][e] = fwd_in(piece_of(acc[c][e], h), e);
        ntt_forward_n<LOGN, 0, false, 2>(z, tw[0], lds, t);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            uint64_t *o = lev_out + ((rot * 2 + c) * 2 + h) * (size_t)N + 8 * t;
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = pack(Pt{montmul<P1, PI1>(z[h][e].a, RR1), montmul<P2, PI2>(z[h][e].b, RR2)});
        }
        __syncthreads();
    }
}

#endif  // unit 2

#if MKT_NTT_IN(3)
// KMS phase 2 (bootstrapping.jl:448-558) with exact products; one workgroup per ciphertext, every thread only touches its
// own coefficients (e*NT + t) and transform points (8t + e)
struct ExactPhase2Args {
    const uint32_t *lin; int lwe_stride;
    int k, l_lev, logB_lev, l_uni, logB_uni, rtot;
    const uint64_t *levkey;    // [B][rtot][2 polys][2 halves][N]
    const uint64_t *rlk_d;     // [k][l_uni][2 halves][N]
    const uint64_t *rlk_f;     // [k][l_uni][2 polys][2 halves][N]
    const uint64_t *pub_b;     // [k][l_uni][2 halves][N]
    const uint64_t *crs;       // [l_uni][2 halves][N]
    uint64_t *acc;             // [B][1+k][N]
    uint64_t *scratch;         // [B][2][k+1][2 halves][N]
};
template <int LOGN>
__global__ __launch_bounds__((1 << (LOGN - NLR))) void exact_kms_phase2_kernel(const uint4 *__restrict__ tab, const ExactPhase2Args a) {
    constexpr int N = 1 << LOGN, NT = N >> NLR;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem);
    const int t = threadIdx.x;
    const uint4 *tw[1]; const int which[1] = {0};
    stage_tables<LOGN, 1>(tab, reinterpret_cast<uint4 *>(lds + 2 * NttLds<LOGN>::WORDS), t, NT, tw, which);   // two staging buffers (lift_pair), then the table
    const NttConsts kc = tab_consts<LOGN>(tab);
    const size_t g = blockIdx.x;
    const int k = a.k;
    uint64_t *acc = a.acc + g * (size_t)(k + 1) * N;
    uint64_t *tx = a.scratch + g * (size_t)4 * (k + 1) * N;
    uint64_t *ty2 = tx + (size_t)2 * (k + 1) * N;
    const Gadget<uint64_t> glev(a.l_lev, a.logB_lev), guni(a.l_uni, a.logB_uni);
    if (a.lin) {                                                                   // bootstrapping.jl:11-23
        uint32_t tb = divbits<uint32_t>(a.lin[g * a.lwe_stride + a.lwe_stride - 1], 32 - LOGN - 1);
        const uint64_t ev = (uint64_t)1 << 61, me = (uint64_t)0 - ev;
        uint64_t lo_v = ev, hi_v = me;
        if (tb > (uint32_t)N) { tb -= (uint32_t)N; lo_v = me; hi_v = ev; }
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int i = e * NT + t;
            acc[i] = ((uint32_t)i < tb) ? lo_v : hi_v;
            for (int q = 1; q <= k; q++) acc[(size_t)q * N + i] = 0;
        }
    }
    auto zero2 = [](Pt (&x)[2][8]) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int e = 0; e < 8; e++) { x[h][e].a = 0; x[h][e].b = 0; }
    };
    auto digit_ntt = [&](Pt (&z)[8], const uint64_t (&tp)[8], const Gadget<uint64_t> &gd, int j) {
#pragma unroll
        for (int e = 0; e < 8; e++) z[e] = res_small(gd.digit(tp[e], j));
        ntt_forward<LOGN>(z, tw[0], lds, t);
    };
    auto mac2 = [&](Pt (&dst)[2][8], const Pt (&z)[8], const uint64_t *tbl, bool subtract) {   // dst += / -= z * (low, high) of one split polynomial
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const Pt y = unpack(tbl[(size_t)h * N + 8 * t + e]);               // accumulators are lazy, [0, 2P)
                dst[h][e] = subtract ? pt_msub(dst[h][e], z[e], y) : pt_mac(dst[h][e], z[e], y);
            }
    };
    for (int idx = 0; idx < k; idx++) {
        const int iter = idx == 0 ? 1 : a.l_lev;                                   // :481
        const int rowbase = idx == 0 ? 0 : 1 + (idx - 1) * a.l_lev;
        const uint64_t *lev = a.levkey + (g * (size_t)a.rtot + rowbase) * 4 * N;   // [row][poly][half][N]
        const uint64_t *rd = a.rlk_d + (size_t)idx * a.l_uni * 2 * N;
        const uint64_t *rf = a.rlk_f + (size_t)idx * a.l_uni * 4 * N;
        Pt tv[2][8];
        zero2(tv);
        for (int q = 0; q <= idx; q++) {
            uint64_t tp[8];
#pragma unroll
            for (int e = 0; e < 8; e++) tp[e] = glev.prep(acc[(size_t)q * N + e * NT + t]);   // :470-471
            Pt txq[2][8], tyq[2][8];
            zero2(txq); zero2(tyq);
            for (int j = 0; j < iter; j++) {                                       // :485-499 LEV multiplication
                Pt z[8];
                digit_ntt(z, tp, glev, j);
                mac2(txq, z, lev + (size_t)(2 * j) * 2 * N, false);
                mac2(tyq, z, lev + (size_t)(2 * j + 1) * 2 * N, false);
            }
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int e = 0; e < 8; e++) tx[((size_t)q * 2 + h) * N + 8 * t + e] = pack(txq[h][e]);
            uint64_t yw[8];
            lift_pair<LOGN>(tyq, yw, tw[0], kc, lds, t);                // :501-504
#pragma unroll
            for (int e = 0; e < 8; e++) tp[e] = guni.prep(yw[e]);                  // :508-509
            Pt tyu[2][8];
            zero2(tyu);
            const uint64_t *vk = q == 0 ? a.crs : a.pub_b + (size_t)(q - 1) * a.l_uni * 2 * N;
            for (int j = 0; j < a.l_uni; j++) {                                    // :521-535 u and v
                Pt z[8];
                digit_ntt(z, tp, guni, j);
                mac2(tyu, z, rd + (size_t)j * 2 * N, false);
                mac2(tv, z, vk + (size_t)j * 2 * N, q == 0);                       // mulsubto! with crs, muladdto! with b_i
            }
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int e = 0; e < 8; e++) ty2[((size_t)q * 2 + h) * N + 8 * t + e] = pack(tyu[h][e]);
        }
        uint64_t vw[8];
        lift_pair<LOGN>(tv, vw, tw[0], kc, lds, t);                      // :538
        uint64_t tp[8];
#pragma unroll
        for (int e = 0; e < 8; e++) tp[e] = guni.prep(vw[e]);                      // :541
        Pt tyb[2][8], tya[2][8];
        zero2(tya);
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int e = 0; e < 8; e++) tyb[h][e] = unpack(ty2[(size_t)h * N + 8 * t + e]);
        for (int i = 0; i < a.l_uni; i++) {                                        // :547-550 w
            Pt z[8];
            digit_ntt(z, tp, guni, i);
            mac2(tyb, z, rf + (size_t)(2 * i) * 2 * N, false);
            mac2(tya, z, rf + (size_t)(2 * i + 1) * 2 * N, false);
        }
        for (int q = 0; q <= idx + 1; q++) {                                       // :553 add!(tx, ty); :556 ifftto!(acc, tx)
            Pt s[2][8];
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    Pt xv; xv.a = 0; xv.b = 0;
                    if (q <= idx) xv = unpack(tx[((size_t)q * 2 + h) * N + 8 * t + e]);
                    const Pt yv = q == 0 ? tyb[h][e] : (q == idx + 1 ? tya[h][e] : unpack(ty2[((size_t)q * 2 + h) * N + 8 * t + e]));
                    s[h][e] = pt_add_lazy(xv, yv);
                }
            uint64_t w[8];
            lift_pair<LOGN>(s, w, tw[0], kc, lds, t);
#pragma unroll
            for (int e = 0; e < 8; e++) acc[(size_t)q * N + e * NT + t] = w[e];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// CCS blind rotation with exact products (bootstrapping.jl:234-328), 32-bit ring: one workgroup per ciphertext, the same step
// sequence as ccs_blindrotate_kernel -- per input polynomial q: decompose, l forward transforms, u = sum dig_j d[j],
// v = -/+ sum dig_j (crs | b_{q-1})[j], inverse of v, decompose v, l forward transforms, w into tacc.b and tacc.a[idx]; then
// every polynomial times the monomial, inverse, add.  All sums are integers mod P, so the order in which the reference's
// Float64 sums must be formed plays no role here.  True coefficients stay below 2 (np + 2) l N 2^(logB-1) 2^31 < P / 2 (centered words: context.cpp exact_gate_ok)
// (host check).  The accumulator lives in the caller's buffer, the u of the earlier parties' polynomials in a scratch area.
// ------------------------------------------------------------------------------------------------
struct ExactCcsArgs {
    const uint32_t *lwe; int lwe_stride, pre_switched;
    int n, k, l, logB;
    const uint64_t *brk; size_t brk_party_stride;    // [party][n][3l][N]: d[l], then (f[j].b, f[j].a)
    const uint64_t *pub_b, *crs, *mono;               // [party][l][N], [l][N], [2N][N]
    uint32_t *acc;                                    // [B][1+k][N]
    uint64_t *scratch;                                // [B][k+1][N] residue pairs
};
template <int LOGN>
__global__ __launch_bounds__((1 << (LOGN - NLR))) __attribute__((amdgpu_waves_per_eu(MKT_EXACT_WPE, MKT_EXACT_WPE))) void exact_ccs_kernel(const uint4 *__restrict__ tab, const ExactCcsArgs a) {
    constexpr int N = 1 << LOGN, NT = N >> NLR;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem);
    const int t = threadIdx.x;
    const uint4 *tw[1]; const int which[1] = {0};
    stage_tables<LOGN, 1>(tab, reinterpret_cast<uint4 *>(lds + NttLds<LOGN>::WORDS), t, NT, tw, which);
    const NttConsts kc = tab_consts<LOGN>(tab);
    const size_t g = blockIdx.x;
    const int k = a.k, l = a.l, n = a.n;
    uint32_t *acc = a.acc + g * (size_t)(k + 1) * N;
    uint64_t *sc = a.scratch + g * (size_t)(k + 1) * N;
    const Gadget<uint32_t> gd(l, a.logB);
    const int msbit = 32 - LOGN - 1;
    auto zero = [](Pt (&x)[8]) {
#pragma unroll
        for (int e = 0; e < 8; e++) { x[e].a = 0; x[e].b = 0; }
    };
    for (int idx = 0; idx < k; idx++) {
        const int np = idx + 1;
        const uint32_t *at_src = a.lwe + g * (size_t)a.lwe_stride + (size_t)idx * n;
        for (int i = 0; i < n; i++) {
            const uint32_t v0 = at_src[i];
            const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.pre_switched ? v0 : divbits<uint32_t>(v0, msbit)));
            if (at == 0) continue;                                                 // :261
            const uint64_t *uni = a.brk + (size_t)idx * a.brk_party_stride + (size_t)i * 3 * l * N;
            const uint64_t *ud = uni, *uf = uni + (size_t)l * N;
            Pt tb[8], ta[8];
            zero(tb); zero(ta);
            for (int q = 0; q <= np; q++) {
                uint32_t tp[8];
#pragma unroll
                for (int e = 0; e < 8; e++) tp[e] = gd.prep(acc[(size_t)q * N + e * NT + t]);   // :264-275
                Pt tu[8], tv[8];
                zero(tu); zero(tv);
                const uint64_t *vk = q == 0 ? a.crs : a.pub_b + (size_t)(q - 1) * l * N;
                for (int j = 0; j < l; j++) {                                      // :279-294 u and v
                    uint64_t kd[8], kv[8];                                         // key rows requested before the transform that hides them (at use: an exposed round trip per digit)
#pragma unroll
                    for (int e = 0; e < 8; e++) { kd[e] = ud[(size_t)j * N + 8 * t + e]; kv[e] = vk[(size_t)j * N + 8 * t + e]; }
                    __builtin_amdgcn_sched_barrier(0);
                    Pt z[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) z[e] = res_small(gd.digit(tp[e], j));
                    ntt_forward<LOGN>(z, tw[0], lds, t);
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        tu[e] = pt_mac(tu[e], z[e], unpack(kd[e]));                // accumulators are lazy, [0, 2P)
                        const Pt y = unpack(kv[e]);
                        tv[e] = q == 0 ? pt_msub(tv[e], z[e], y) : pt_mac(tv[e], z[e], y);     // mulsubto! with crs, muladdto! with b_i
                    }
                }
                if (q == 0) {
#pragma unroll
                    for (int e = 0; e < 8; e++) tb[e] = pt_add_lazy(tb[e], tu[e]);
                } else if (q == np) {
#pragma unroll
                    for (int e = 0; e < 8; e++) ta[e] = pt_add_lazy(ta[e], tu[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e++) sc[(size_t)q * N + 8 * t + e] = pack(tu[e]);
                }
                ntt_inverse<LOGN, Plan<LOGN, NLR>::NPASS - 1>(tv, tw[0], lds, t, kc.ninv);   // :297-300
#pragma unroll
                for (int e = 0; e < 8; e++) tp[e] = gd.prep((uint32_t)crt_signed(tv[e]));   // :303-310
                for (int j = 0; j < l; j++) {                                      // :313-320 w
                    uint64_t fb[8], fa[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) { fb[e] = uf[(size_t)(2 * j) * N + 8 * t + e]; fa[e] = uf[(size_t)(2 * j + 1) * N + 8 * t + e]; }
                    __builtin_amdgcn_sched_barrier(0);
                    Pt z[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) z[e] = res_small(gd.digit(tp[e], j));
                    ntt_forward<LOGN>(z, tw[0], lds, t);
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        tb[e] = pt_mac(tb[e], z[e], unpack(fb[e]));
                        ta[e] = pt_mac(ta[e], z[e], unpack(fa[e]));
                    }
                }
            }
            const uint64_t *mrow = a.mono + (size_t)(at - 1) * N + 8 * t;
            uint64_t mr[8];
#pragma unroll
            for (int e = 0; e < 8; e++) mr[e] = mrow[e];
            for (int q = 0; q <= np; q++) {                                        // :322-324 mul!(monomial, tacc); ifftto!; add!
                Pt s[8];
                uint32_t aw[8];                                                    // the words the lift is added to: requested before the inverse transform
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const Pt x = q == 0 ? tb[e] : (q == np ? ta[e] : unpack(sc[(size_t)q * N + 8 * t + e]));
                    s[e] = pt_mont(x, unpack(mr[e]));
                    aw[e] = acc[(size_t)q * N + e * NT + t];
                }
                __builtin_amdgcn_sched_barrier(0);
                ntt_inverse<LOGN, Plan<LOGN, NLR>::NPASS - 1>(s, tw[0], lds, t, kc.ninv);
#pragma unroll
                for (int e = 0; e < 8; e++) acc[(size_t)q * N + e * NT + t] = aw[e] + (uint32_t)crt_signed(s[e]);
            }
        }
    }
}

#endif  // unit 3

template <typename K>
static hipError_t ntt_set_lds(K kern, size_t bytes) {
    if (bytes > 48 * 1024) return hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return hipSuccess;
}

}  // namespace

#ifdef MKT_NTT_ONLY_LOGN   // development builds: instantiate one transform size only (seconds instead of minutes)
#define MKT_NTT_DISPATCH(logN, ...)                   \
    switch (logN) {                                   \
    case MKT_NTT_ONLY_LOGN: { constexpr int LN = MKT_NTT_ONLY_LOGN; __VA_ARGS__; } break; \
    default: return hipErrorInvalidValue;             \
    }
#else
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
#endif

#if MKT_NTT_IN(0)
template <int LN, typename WORD, bool MONT>
static hipError_t ntt_fwd_launch(const uint4 *tb, const void *p, uint64_t *t, size_t B, int grid, size_t lds, hipStream_t s) {
    hipError_t e = ntt_set_lds(ntt_fwd_kernel<LN, WORD, MONT>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((ntt_fwd_kernel<LN, WORD, MONT>), dim3(grid), dim3(Ppw<LN>::v << (LN - NLR)), lds, s, tb, (const WORD *)p, t, B);
    return hipSuccess;
}
// workgroups of a batched transform launch (each walks the batch with the next polynomial's words in flight); MKT_NTT_GRID overrides.
// tools/ntt_grid_sweep.sh: 8192 against 32 768 -- N = 2048 forward 0.402 -> 0.420, inverse 0.453 -> 0.465 of 8 TB/s, N = 1024 0.448 / 0.474 -> 0.450 / 0.483
static size_t ntt_grid_cap() {
    const int g = launch_tuning().ntt_grid;
    return g > 0 ? (size_t)g : 8192;
}

hipError_t launch_ntt_fwd(int logN, int W, const uint64_t *tab, const void *p, uint64_t *t, size_t B, int montgomery, hipStream_t s) {
    if (!B) return hipSuccess;
    const uint4 *tb = reinterpret_cast<const uint4 *>(tab);
    hipError_t e_ = hipSuccess;
    MKT_NTT_DISPATCH(logN, {
        constexpr int PPW = Ppw<LN>::v; const size_t lds = lds_bytes<LN>(1, PPW); const size_t groups = (B + PPW - 1) / PPW; const size_t gcap = ntt_grid_cap(); const int grid = (int)(groups < gcap ? groups : gcap);
if (W == 64) e_ = montgomery ? ntt_fwd_launch<LN, uint64_t, true>(tb, p, t, B, grid, lds, s) : ntt_fwd_launch<LN, uint64_t, false>(tb, p, t, B, grid, lds, s);
        else e_ = montgomery ? ntt_fwd_launch<LN, uint32_t, true>(tb, p, t, B, grid, lds, s) : ntt_fwd_launch<LN, uint32_t, false>(tb, p, t, B, grid, lds, s);
        if (e_ != hipSuccess) return e_;
    });
    return hipGetLastError();
}
hipError_t launch_ntt_inv(int logN, int W, const uint64_t *tab, const uint64_t *t, void *p, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    const uint4 *tb = reinterpret_cast<const uint4 *>(tab);
    MKT_NTT_DISPATCH(logN, {
        constexpr int PPW = Ppw<LN>::v; const size_t lds = lds_bytes<LN>(1, PPW); const size_t groups = (B + PPW - 1) / PPW; const size_t gcap = ntt_grid_cap(); const int grid = (int)(groups < gcap ? groups : gcap);
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
        const size_t lds = lds_bytes<LN>(1);
        if (W == 64) { hipError_t e = ntt_set_lds(exact_polymul_kernel<LN, uint64_t>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((exact_polymul_kernel<LN, uint64_t>), dim3(grid), dim3(1 << (LN - NLR)), lds, s, tb, (const uint64_t *)a, (const uint64_t *)b, (uint64_t *)out, B); }
        else { hipError_t e = ntt_set_lds(exact_polymul_kernel<LN, uint32_t>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((exact_polymul_kernel<LN, uint32_t>), dim3(grid), dim3(1 << (LN - NLR)), lds, s, tb, (const uint32_t *)a, (const uint32_t *)b, (uint32_t *)out, B); }
    });
    return hipGetLastError();
}

#endif  // unit 0
#if MKT_NTT_IN(1)
hipError_t launch_exact_blindrotate(int logN, const uint64_t *tab, const uint64_t *brk, const uint64_t *mono, const uint32_t *lwe, int lwe_stride,
                                    int pre_switched, int n, int l, int logB, int blk_len, uint32_t *acc, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    if (blk_len != 1 && blk_len != 3) return hipErrorInvalidValue;
    last_rot_kernel = "exact_blindrotate_kernel";
    const uint4 *tb = reinterpret_cast<const uint4 *>(tab);
    MKT_NTT_DISPATCH(logN, {
        const size_t lds = lds_bytes<LN>(1);
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

hipError_t launch_exact_blindrotate_kr(int logN, const uint64_t *tab, const uint64_t *brk, const uint64_t *mono, const uint32_t *lwe, int lwe_stride,
                                       int pre_switched, int n, int kr, int l, int logB, int blk_len, uint32_t *acc, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    if (kr < 1 || kr > 3 || blk_len < 1 || n % blk_len) return hipErrorInvalidValue;
    last_rot_kernel = "exact_blindrotate_kr_kernel";
    const uint4 *tb = reinterpret_cast<const uint4 *>(tab);
#define MKT_EXACT_KR_LAUNCH(KRV) do { hipError_t e = ntt_set_lds(exact_blindrotate_kr_kernel<LN, KRV>, lds); if (e != hipSuccess) return e; \
        hipLaunchKernelGGL((exact_blindrotate_kr_kernel<LN, KRV>), dim3((unsigned)B), dim3(1 << (LN - NLR)), lds, s, tb, brk, mono, lwe, lwe_stride, pre_switched, n, l, logB, blk_len, acc); } while (0)
    MKT_NTT_DISPATCH(logN, {
        const size_t lds = lds_bytes<LN>(1);
        if (kr == 1) MKT_EXACT_KR_LAUNCH(1); else if (kr == 2) MKT_EXACT_KR_LAUNCH(2); else MKT_EXACT_KR_LAUNCH(3);
    });
#undef MKT_EXACT_KR_LAUNCH
    return hipGetLastError();
}

hipError_t launch_exact_blindrotate_kany(int logN, const uint64_t *tab, const uint64_t *brk, const uint64_t *mono, const uint32_t *lwe, int lwe_stride,
                                         int pre_switched, int n, int kr, int l, int logB, int blk_len, uint32_t *acc, uint64_t *scratch, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    if (kr < 1 || blk_len < 1 || n % blk_len || !scratch) return hipErrorInvalidValue;
    last_rot_kernel = "exact_blindrotate_kany_kernel";
    const uint4 *tb = reinterpret_cast<const uint4 *>(tab);
    MKT_NTT_DISPATCH(logN, {
        const size_t lds = lds_bytes<LN>(1);
        hipError_t e = ntt_set_lds(exact_blindrotate_kany_kernel<LN>, lds); if (e != hipSuccess) return e;
        hipLaunchKernelGGL((exact_blindrotate_kany_kernel<LN>), dim3((unsigned)B), dim3(1 << (LN - NLR)), lds, s, tb, brk, mono, lwe, lwe_stride, pre_switched, n, kr + 1, l, logB, blk_len, acc, scratch);
    });
    return hipGetLastError();
}

#endif  // unit 1
#if MKT_NTT_IN(0)
hipError_t launch_ntt_fwd_split(int logN, const uint64_t *tab, const void *p, uint64_t *out, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    const int grid = (int)(B < 32768 ? B : 32768);
    const uint4 *tb = reinterpret_cast<const uint4 *>(tab);
    MKT_NTT_DISPATCH(logN, {
        const size_t lds = lds_bytes<LN>(1);
        hipError_t e = ntt_set_lds(ntt_fwd_split_kernel<LN>, lds); if (e != hipSuccess) return e;
        hipLaunchKernelGGL((ntt_fwd_split_kernel<LN>), dim3(grid), dim3(1 << (LN - NLR)), lds, s, tb, (const uint64_t *)p, out, B);
    });
    return hipGetLastError();
}

#endif  // unit 0
#if MKT_NTT_IN(2)
hipError_t launch_exact_kms_phase1(int logN, const uint64_t *tab, const ExactKmsArgs &a, size_t B, hipStream_t s) {      // internal: phase 1 of launch_exact_kms (unit 2)
    if (!B || a.phase2_only) return hipSuccess;
    last_rot_kernel = "exact_kms_phase1_kernel";
    const uint4 *tb = reinterpret_cast<const uint4 *>(tab);
    MKT_NTT_DISPATCH(logN, {
        const size_t lds = lds_bytes<LN>(1, 2);                                      // two staging buffers: the halves of a lifted sum are inverse-transformed side by side
        hipError_t e = hipSuccess;
        if (a.phase2_only) {                                                       // phase 1 ran on the Float64 pipe (fx_exact.hip) and levkey holds its rows
        } else if (a.blk_len > 1 && a.wide != 0) {                                // exact_wide = 0: the per-key-bit kernel below (tests force both)
            last_rot_kernel = "exact_kms_block_phase1_kernel";
            e = ntt_set_lds(exact_kms_block_phase1_kernel<LN>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((exact_kms_block_phase1_kernel<LN>), dim3((unsigned)(B * (size_t)a.rtot)), dim3(1 << (LN - NLR)), lds, s, tb, a.brk, a.brk_party_stride, a.mono,
                               a.lwe, a.lwe_stride, a.pre_switched, a.n, a.l_gsw, a.logB_gsw, a.blk_len, B, a.rtot, a.slot_party, a.slot_row, a.logB_lev, a.levkey);
        } else if (a.blk_len > 1) {
            e = ntt_set_lds(exact_kms_phase1_kernel<LN, true>, lds); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((exact_kms_phase1_kernel<LN, true>), dim3((unsigned)(B * (size_t)a.rtot)), dim3(1 << (LN - NLR)), lds, s, tb, a.brk, a.brk_party_stride, a.mono,
                               a.lwe, a.lwe_stride, a.pre_switched, a.n, a.l_gsw, a.logB_gsw, a.blk_len, B, a.rtot, a.slot_party, a.slot_row, a.logB_lev, a.levkey);
        } else {
            if (a.wide != 0 && a.l_gsw == 2) {                                     // the paired-transform kernel (default); exact_wide = 0: the one-at-a-time kernel below (tests force both)
                const size_t lds2 = lds_bytes<LN>(1, 2);
                e = ntt_set_lds(exact_kms_phase1_p2pf_kernel<LN>, lds2); if (e != hipSuccess) return e;
                hipLaunchKernelGGL((exact_kms_phase1_p2pf_kernel<LN>), dim3((unsigned)(B * (size_t)a.rtot)), dim3(1 << (LN - NLR)), lds2, s, tb, a.brk, a.brk_party_stride,
                                   a.lwe, a.lwe_stride, a.pre_switched, a.n, a.logB_gsw, B, a.rtot, a.slot_party, a.slot_row, a.logB_lev, a.levkey);
            } else {
                e = ntt_set_lds(exact_kms_phase1_kernel<LN, false>, lds); if (e != hipSuccess) return e;
                hipLaunchKernelGGL((exact_kms_phase1_kernel<LN, false>), dim3((unsigned)(B * (size_t)a.rtot)), dim3(1 << (LN - NLR)), lds, s, tb, a.brk, a.brk_party_stride, a.mono,
                                   a.lwe, a.lwe_stride, a.pre_switched, a.n, a.l_gsw, a.logB_gsw, 1, B, a.rtot, a.slot_party, a.slot_row, a.logB_lev, a.levkey);
            }
        }
    });
    return hipGetLastError();
}
#endif  // unit 2
#if MKT_NTT_IN(3)
hipError_t launch_exact_kms_phase1(int logN, const uint64_t *tab, const ExactKmsArgs &a, size_t B, hipStream_t s);
hipError_t launch_exact_kms(int logN, const uint64_t *tab, const ExactKmsArgs &a, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    hipError_t e1 = launch_exact_kms_phase1(logN, tab, a, B, s);
    if (e1 != hipSuccess || a.phase1_only) return e1;
    const uint4 *tb = reinterpret_cast<const uint4 *>(tab);
    MKT_NTT_DISPATCH(logN, {
        const size_t lds = lds_bytes<LN>(1, 2);
        hipError_t e = hipSuccess;
        ExactPhase2Args q;
        q.lin = a.lin_for_tv; q.lwe_stride = a.lwe_len; q.k = a.k; q.l_lev = a.l_lev; q.logB_lev = a.logB_lev; q.l_uni = a.l_uni; q.logB_uni = a.logB_uni; q.rtot = a.rtot;
        q.levkey = a.levkey; q.rlk_d = a.rlk_d; q.rlk_f = a.rlk_f; q.pub_b = a.pub_b; q.crs = a.crs; q.acc = a.acc; q.scratch = a.scratch;
        e = ntt_set_lds(exact_kms_phase2_kernel<LN>, lds); if (e != hipSuccess) return e;
        hipLaunchKernelGGL((exact_kms_phase2_kernel<LN>), dim3((unsigned)B), dim3(1 << (LN - NLR)), lds, s, tb, q);
    });
    return hipGetLastError();
}

hipError_t launch_exact_ccs(int logN, const uint64_t *tab, const ExactCcsHostArgs &h, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    last_rot_kernel = "exact_ccs_kernel";
    const uint4 *tb = reinterpret_cast<const uint4 *>(tab);
    ExactCcsArgs a;
    a.lwe = h.lwe; a.lwe_stride = h.lwe_stride; a.pre_switched = h.pre_switched; a.n = h.n; a.k = h.k; a.l = h.l; a.logB = h.logB;
    a.brk = h.brk; a.brk_party_stride = h.brk_party_stride; a.pub_b = h.pub_b; a.crs = h.crs; a.mono = h.mono; a.acc = h.acc; a.scratch = h.scratch;
    MKT_NTT_DISPATCH(logN, {
        const size_t lds = lds_bytes<LN>(1);
        hipError_t e = ntt_set_lds(exact_ccs_kernel<LN>, lds); if (e != hipSuccess) return e;
        hipLaunchKernelGGL((exact_ccs_kernel<LN>), dim3((unsigned)B), dim3(1 << (LN - NLR)), lds, s, tb, a);
    });
    return hipGetLastError();
}

#endif  // unit 3

}  // namespace mktd
