// HIP kernels of the gate-bootstrapping hot path for gfx950 (CDNA4, wave64).
// Hand-written for MI355X only; no portability layers.  All floating point is IEEE double with
// contraction off (one rounding per reference operation); see fft_device.h for the mapping.
#include "kernel_common.h"

#pragma clang fp contract(off)

// The file is compiled once per translation unit MKT_TU (Makefile) so the template instantiations build in parallel:
//   0 transforms, small kernels, key switch, dispatchers      3 / 5 blind rotation of the block schemes (LMSS, KMS_block), 32- / 64-bit ring
//   1 blind rotation, 32-bit ring, plain schemes              4 KMS phase 2, CCS; 7 general-k rotation (CGGI / LMSS with k > 1)
//   2 blind rotation, 64-bit ring, plain schemes              6 blind rotation, latency variant (one rotation over 2l thread groups)
// MKT_TU undefined = everything in one unit.
#ifdef MKT_TU
#define MKT_IN_TU(n) (MKT_TU == (n))
#else
#define MKT_IN_TU(n) 1
#endif

namespace mktd {

#if MKT_IN_TU(0)
// ------------------------------------------------------------------------------------------------
// batched transforms (fft.jl:57-63 / :74-81): HBM -> HBM, one polynomial per workgroup pass
// ------------------------------------------------------------------------------------------------
// Batched transforms HBM -> HBM.  Loads are contiguous across the wave (8 B or 4 B per lane), software-pipelined one
// polynomial ahead in registers.  The reference-order output (point 4t+e) is either stored as 64 B per lane
// (measured slower: gone) or brought to thread-contiguous ownership by one more staging exchange and stored as
// contiguous 16 B/lane wave accesses (shipped).  tools/membench.hip measures the pattern ceilings on this part:
// copy 4.6-5.4 TB/s, contiguous stores 5.0-5.2, 64 B-strided stores 4.2-4.8.
// Shipped settings (each measured): Psi resident in LDS, contiguous 16 B/lane stores through one more staging
// exchange, nontemporal loads and stores (+10-16 %), the coefficients of the next two polynomial groups in flight.
constexpr int FFT_PF = 2;
typedef double __attribute__((ext_vector_type(2))) mkt_d2;
template <typename T> __device__ __forceinline__ T stream_load(const T *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ cplx stream_load_c(const cplx *p) {
    mkt_d2 v = __builtin_nontemporal_load(reinterpret_cast<const mkt_d2 *>(p));
    return cplx{v.x, v.y};
}
__device__ __forceinline__ void stream_store_c(cplx *p, cplx z) {
    mkt_d2 v = {z.re, z.im};
    __builtin_nontemporal_store(v, reinterpret_cast<mkt_d2 *>(p));
}
template <typename T> __device__ __forceinline__ void stream_store(T *p, T v) { __builtin_nontemporal_store(v, p); }

template <int LOGM, typename WORD, int NBT>
__global__ __launch_bounds__((Plan<LOGM, LOGR>::NT)) void transform_fwd_kernel(TwPtrs tw, const WORD *__restrict__ p,
                                                                            cplx *__restrict__ out, size_t B, int dev_order) {
    using P = Plan<LOGM, LOGR, NBT>;
    constexpr int R = P::R, NT = P::NT, M = P::M, N = 2 * M;
    cplx *lds = reinterpret_cast<cplx *>(mkt_smem);
    const int t = threadIdx.x;
    XS xs = make_xs();
    cplx *psi_l = lds + P::LDS_CPLX;          // the twiddle table stays in LDS for all polynomials of this workgroup
    for (int i = t; i < M; i += NT) psi_l[i] = tw.psi[i];
    __syncthreads();
    const cplx *psi_f = psi_l;
    cplx rt[R];
#pragma unroll
    for (int e = 0; e < R; e++) rt[e] = tw.roots[e * NT + t];
    // software pipeline: the coefficients of this workgroup's next PF polynomial groups are in flight while one is transformed
    constexpr int PF = FFT_PF;
    WORD c0[PF][NBT][R], c1[PF][NBT][R];
    const size_t groups = (B + NBT - 1) / NBT;
    auto load = [&](int d, size_t gi) {
#pragma unroll
        for (int u = 0; u < NBT; u++) {
            const size_t b = gi * NBT + u < B ? gi * NBT + u : B - 1;
#pragma unroll
            for (int e = 0; e < R; e++) { c0[d][u][e] = stream_load(&p[b * N + e * NT + t]); c1[d][u][e] = stream_load(&p[b * N + M + e * NT + t]); }
        }
    };
#pragma unroll
    for (int d = 0; d < PF; d++)
        if (blockIdx.x + (size_t)d * gridDim.x < groups) load(d, blockIdx.x + (size_t)d * gridDim.x);
    for (size_t g0 = blockIdx.x; g0 < groups; g0 += (size_t)PF * gridDim.x) {
#pragma unroll
        for (int d = 0; d < PF; d++) {
            const size_t g = g0 + (size_t)d * gridDim.x;
            if (g >= groups) break;
            cplx z[NBT][R];
#pragma unroll
            for (int u = 0; u < NBT; u++)
#pragma unroll
                for (int e = 0; e < R; e++) {
                    cplx v;
                    v.re = word_to_f64<WORD>(c0[d][u][e]);
                    v.im = word_to_f64<WORD>((WORD)((WORD)0 - c1[d][u][e]));   // subtraction in the integer type (fft.jl:60)
                    z[u][e] = cmul(v, rt[e]);
                }
            if (g + (size_t)PF * gridDim.x < groups) load(d, g + (size_t)PF * gridDim.x);
            fft_forward<LOGM, LOGR, NBT>(z, psi_f, lds, t, xs.lx);
            const bool contig = !dev_order && P::NPASS > 1;
            if (contig) {
                __syncthreads();
                exchange_lds<LOGM, LOGR, NBT>(z, lds + P::buf_off(P::NPASS - 1), t, 0, P::lo(0));
                __syncthreads();      // the next transform's exchanges reuse the buffers
            }
#pragma unroll
            for (int u = 0; u < NBT; u++) {
                const size_t b = g * NBT + u;
                if (b >= B) break;
                cplx *o = out + b * M;
#pragma unroll
                for (int e = 0; e < R; e++)   // device point order for resident tables; else the reference's TransPoly order
                    stream_store_c(&o[dev_order ? dev_pos(dev_order, t * R + e, NT) : (contig ? e * NT + t : t * R + e)], z[u][e]);
            }
        }
    }
}

template <int LOGM, typename WORD>
__global__ __launch_bounds__((Plan<LOGM, LOGR>::NT)) void transform_inv_kernel(TwPtrs tw, const cplx *__restrict__ in,
                                                                            WORD *__restrict__ p, size_t B) {
    using P = Plan<LOGM, LOGR>;
    constexpr int R = P::R, NT = P::NT, M = P::M, N = 2 * M;
    cplx *lds = reinterpret_cast<cplx *>(mkt_smem);
    const int t = threadIdx.x;
    XS xs = make_xs();
    cplx ri[R];
#pragma unroll
    for (int e = 0; e < R; e++) ri[e] = tw.rootsinv[e * NT + t];
    constexpr bool CONTIG = P::NPASS > 1;
    cplx *psi_l = lds + P::LDS_CPLX;          // psiinv[i] == conj(psi[i]) entrywise: one resident table serves both directions
    for (int i = t; i < M; i += NT) psi_l[i] = tw.psi[i];
    __syncthreads();
    constexpr int PF = FFT_PF;
    cplx zn[PF][R];
    auto load = [&](int d, size_t b) {
#pragma unroll
        for (int e = 0; e < R; e++) zn[d][e] = stream_load_c(&in[b * M + (CONTIG ? e * NT + t : t * R + e)]);
    };
#pragma unroll
    for (int d = 0; d < PF; d++)
        if (blockIdx.x + (size_t)d * gridDim.x < B) load(d, blockIdx.x + (size_t)d * gridDim.x);
    for (size_t b0 = blockIdx.x; b0 < B; b0 += (size_t)PF * gridDim.x) {
#pragma unroll
        for (int d = 0; d < PF; d++) {
            const size_t b = b0 + (size_t)d * gridDim.x;
            if (b >= B) break;
            cplx z[R];
#pragma unroll
            for (int e = 0; e < R; e++) z[e] = zn[d][e];
            if (b + (size_t)PF * gridDim.x < B) load(d, b + (size_t)PF * gridDim.x);
            if (CONTIG) {   // contiguous ownership (e*NT + t) -> the inverse transform's first window (4t + e)
                __syncthreads();
                exchange_lds<LOGM, LOGR, 1>(reinterpret_cast<cplx(&)[1][R]>(z), lds + P::buf_off(P::NPASS), t, P::lo(0), 0);
                __syncthreads();
            }
            fft_inverse<LOGM, LOGR, 1, true>(reinterpret_cast<cplx(&)[1][R]>(z), psi_l, lds, t, xs.lx);
            WORD *pp = p + b * N;
#pragma unroll
            for (int e = 0; e < R; e++) {
                const int idx = e * NT + t;
                const cplx v = cmul(z[e], ri[e]);
                stream_store(&pp[idx], native<WORD>(v.re));
                stream_store(&pp[idx + M], native<WORD>(-v.im));
            }
        }
    }
}

// gsw.jl:86-96 decompto!(avec, a, params) on a batch
template <typename WORD>
__global__ void decompose_kernel(const WORD *__restrict__ p, WORD *__restrict__ dig, int N, int l, int logB, size_t total) {
    const Gadget<WORD> gd(l, logB);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / N, c = i % N;
        const WORD tp = gd.prep(p[i]);
        for (int j = 0; j < l; j++) dig[(b * l + j) * N + c] = (WORD)(typename WordTraits<WORD>::S)gd.digit(tp, j);
    }
}

// natural <-> device point order of TransPolys (F64_FFT key upload, table read-back)
template <int LOGM>
__global__ void reorder_kernel(const cplx *__restrict__ in, cplx *__restrict__ out, size_t npolys, int to_device, int order) {
    constexpr int M = 1 << LOGM, R = 1 << LOGR, NT = M / R;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npolys * M; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / M; const int x = (int)(i % M);
        if (to_device) out[b * M + dev_pos(order, x, NT)] = in[i]; else out[i] = in[b * M + dev_pos(order, x, NT)];
    }
}

// ------------------------------------------------------------------------------------------------
// gate.jl:1-58, bootstrapping.jl:8-23: linear part, mod-switch, test vector
// ------------------------------------------------------------------------------------------------
// `ops` (optional): one code per gate -- bits 0-2 the gate, bit 3 / bit 4 = the x / y input is negated first (NOT!, gate.jl:55-58,
// folded into the linear part: -x is what NOT! leaves in memory); `ix` / `iy` (optional): row of the x / y operand in a
// ciphertext pool (a circuit level reads its operands where the earlier levels left them)
__global__ void gate_linear_kernel(int op_all, const uint8_t *__restrict__ ops, const uint32_t *__restrict__ x, const uint32_t *__restrict__ y,
                                   const uint32_t *__restrict__ ix, const uint32_t *__restrict__ iy, size_t pool_rows, uint32_t *__restrict__ out, int len, size_t total) {
    // index arrays in device memory cannot be validated by the host without a copy (mktfhe.h: the caller's precondition); what the
    // engine guarantees regardless is that no index reads outside the pool: rows are clamped to the last one
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t g = i / len; const int c = (int)(i % len);
        const bool isb = c == len - 1;
        const int code = ops ? ops[g] : op_all;
        size_t ra = ix ? (size_t)ix[g] : g, rb = iy ? (size_t)iy[g] : g;
        if (pool_rows) { ra = ra < pool_rows ? ra : pool_rows - 1; rb = rb < pool_rows ? rb : pool_rows - 1; }
        uint32_t a = x[ra * len + c], b = y[rb * len + c];
        if (code & 8) a = 0u - a;
        if (code & 16) b = 0u - b;
        uint32_t r;
        switch (code & 7) {
        case 0:  r = (isb ? (1u << 29) : 0u) - a - b; break;              // NAND gate.jl:1-8
        case 1:  r = (isb ? (7u << 29) : 0u) + a + b; break;              // AND  :10-17
        case 2:  r = (isb ? (1u << 29) : 0u) + a + b; break;              // OR   :19-26
        case 3:  r = (isb ? (1u << 30) : 0u) + 2u * (a + b); break;       // XOR  :28-35
        case 4:  r = (isb ? (3u << 30) : 0u) - 2u * (a + b); break;       // XNOR :37-44
        default: r = (isb ? (7u << 29) : 0u) - a - b; break;              // NOR  :46-53
        }
        out[i] = r;
    }
}

// native MUX: acc[j] += acc[B + j] (two blind-rotation outputs, polynomial by polynomial) and +1/8 at X^0 of the b polynomial
template <typename WORD>
__global__ void mux_combine_kernel(WORD *__restrict__ acc, size_t B, size_t words) {
    const size_t total = B * words;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        WORD v = (WORD)(acc[i] + acc[total + i]);
        if (i % words == 0) v = (WORD)(v + ((WORD)1 << (WordTraits<WORD>::W - 3)));
        acc[i] = v;
    }
}

// the two AND-linear parts of a level of MUX gates gathered from a pool: out[g] = AND(s, a'), out[B + g] = AND(NOT s, b'), a' / b' = the
// operand or its negation (flag bits 0 / 1)
__global__ void mux_linear_kernel(const uint32_t *__restrict__ pool, size_t pool_rows, const uint32_t *__restrict__ is, const uint32_t *__restrict__ ia, const uint32_t *__restrict__ ib,
                                  const uint8_t *__restrict__ fl, uint32_t *__restrict__ out, int len, size_t B) {
    const size_t total = B * (size_t)len;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t g = i / len; const int c = (int)(i % len);
        const uint32_t k = c == len - 1 ? (7u << 29) : 0u;                       // AND, gate.jl:10-17
        const int f = fl ? fl[g] : 0;
        const size_t last = pool_rows - 1;                                       // rows clamped into the pool (see gate_linear_kernel)
        const size_t rs = is[g] < pool_rows ? is[g] : last, ra = ia[g] < pool_rows ? ia[g] : last, rb = ib[g] < pool_rows ? ib[g] : last;
        const uint32_t s = pool[rs * len + c];
        uint32_t a = pool[ra * len + c], b = pool[rb * len + c];
        if (f & 1) a = 0u - a;
        if (f & 2) b = 0u - b;
        out[i] = k + s + a;
        out[total + i] = k + (0u - s) + b;
    }
}

__global__ void negate_kernel(uint32_t *x, size_t total) {   // gate.jl:55-58
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) x[i] = 0u - x[i];
}

__global__ void modswitch_kernel(const uint32_t *__restrict__ lwe, uint32_t *__restrict__ at, uint32_t *__restrict__ bt,
                                 int len, int logN, size_t total) {
    const int bit = 32 - logN - 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t g = i / len; const int c = (int)(i % len);
        const uint32_t v = divbits<uint32_t>(lwe[i], bit);
        if (c == len - 1) bt[g] = v; else at[g * (len - 1) + c] = v;
    }
}

// bootstrapping.jl:11-23: acc = (testvector(btilde), 0 ...)
template <typename WORD>
__global__ void testvector_kernel(const uint32_t *__restrict__ lin, int lwe_stride, int logN, int kacc, WORD *__restrict__ acc) {
    constexpr int W = WordTraits<WORD>::W;
    const int N = 1 << logN;
    const size_t g = blockIdx.x;
    uint32_t tb = divbits<uint32_t>(lin[g * lwe_stride + lwe_stride - 1], 32 - logN - 1);
    const WORD e = (WORD)1 << (W - 3), me = (WORD)((WORD)0 - e);
    WORD lo_v = e, hi_v = me;
    if (tb > (uint32_t)N) { tb -= (uint32_t)N; lo_v = me; hi_v = e; }
    WORD *a = acc + g * (size_t)(1 + kacc) * N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) a[i] = ((uint32_t)i < tb) ? lo_v : hi_v;
    for (int i = threadIdx.x; i < kacc * N; i += blockDim.x) a[N + i] = 0;
}

#endif  // TU 0

#if MKT_IN_TU(1) || MKT_IN_TU(2) || MKT_IN_TU(3) || MKT_IN_TU(5)
// ------------------------------------------------------------------------------------------------
// Blind rotation, RLWE length 1.  bootstrapping.jl:32-76 (CGGI), :114-165 (LMSS), :389-443 and
// :599-659 (KMS / KMS_block phase 1).  One workgroup per rotation; the accumulator (2 polynomials)
// and the transform-domain accumulator stay in registers for all n CMux steps; LDS only stages the
// in-transform exchanges and holds a copy of Psi.  LB = block length (1 for the plain schemes).
// Built and measured slower, so gone: key rows requested before the group's forward transform (+0.5 %:
// 64 more live registers), flat instead of buffer-descriptor loads (+5 %), twiddles from global memory (+21 %), three
// waves per SIMD at M = 512 (same time, lower clock), one key bit's rows requested ahead in the block kernels (+3 % at
// Blockparam, superseded by rot_block.hip), the next block's key rows pulled into L2 ahead of time -- by every workgroup through
// LDS-DMA (KMS2partyblock 33.1 -> 40.2 ms) or by one elected workgroup in 32 into a sink register (34.8 ms): profiles/r05_experiments.txt.
// ------------------------------------------------------------------------------------------------
// Occupancy the register allocator is told to hit EXACTLY (amdgpu_waves_per_eu(min, max)): 3 waves/SIMD pays at
// M >= 1024 with single transforms; elsewhere LDS admits 2 and the allocator should then use all 256 VGPRs --
// builds that stopped at ~186 or chose <= 168 for a third wave LDS cannot host ran up to 25 % slower
// development only: timing build in which every block reads the key rows of block 0 (cache-resident; WRONG results) -- what the L2 / fabric
// latency of the key stream costs the kernel (tools/tu_variant.sh x 1,2,3,5 "-DMKT_ROT_ABLATE_KEYS=1"; profiles/r05_experiments.txt)
#ifndef MKT_ROT_ABLATE_KEYS
#define MKT_ROT_ABLATE_KEYS 0
#endif
template <int LOGM, int NB> struct RotOcc { static constexpr int MINW = (LOGM >= 10 && NB == 1 && MKT_LOGR == 2) ? 3 : 2; };

template <int LOGM, typename WORD, int LB, int LR, int NB, int LT, int BT>
__global__ __launch_bounds__((Plan<LOGM, LR>::NT)) __attribute__((amdgpu_waves_per_eu(RotOcc<LOGM, NB>::MINW, RotOcc<LOGM, NB>::MINW)))
void blindrotate_k1_kernel(const RotArgs a) {
    using P = Plan<LOGM, LR, NB>;   // NB transforms at a time share twiddle loads and barriers
    // exchange routes: the block kernels at even sizes keep every legal exchange in the wave (mode 1: measured +4.5 % at
    // KMS2partyblock), everything else the library default (mode 8)
#ifndef MKT_TW_PF
#define MKT_TW_PF 1       // the next pass's twiddles read ahead of the exchange in front of it (fft_device.h, MO bit 10): 1 = plain kernels, 2 = block kernels too
#endif
    constexpr int PFB = ((MKT_TW_PF >= 1 && LB == 1) || MKT_TW_PF >= 2) ? 0x400 : 0;
    constexpr int MO = (LB > 1 && !(LOGM & 1)) ? (1 | PFB) : (PFB ? 0x4ff : -1);
    constexpr int R = P::R, NT = P::NT, M = P::M, N = 2 * M, W = WordTraits<WORD>::W;
    cplx *lds = reinterpret_cast<cplx *>(mkt_smem);
    const int t = threadIdx.x;
    XS xs = make_xs();
    // co-resident workgroups run the same loop in near lock-step and then fight for the VALU and the LDS at the same
    // moments; delaying every other group of 256 workgroups (= every other workgroup of a CU under round-robin
    // dispatch; speed only, never correctness) de-phases them
    const unsigned bid = blockIdx.x + a.block0;
    if (a.stagger > 0 && ((bid >> 8) & 1)) {
        for (int s = 0; s < a.stagger; s++) __builtin_amdgcn_s_sleep(8);
    }
    // the forward twiddle table stays resident in LDS behind the staging buffers (17.0 vs 20.6 ms from global memory at
    // KMS k=2 N=1024); the inverse multiplies by its conjugate
    cplx *psi_l = lds + P::LDS_CPLX;
    for (int i = t; i < M; i += NT) psi_l[i] = a.tw.psi[i];
    __syncthreads();
    // workgroups are dealt slot-major (all ciphertexts' rotations of one party/row are adjacent), so the workgroups
    // resident at any time stream the SAME party's key rows through L2; results are stored ciphertext-major
    size_t gate; int slot;
    rot_decode(a, bid, gate, slot);
    const size_t rot = gate * (size_t)a.rows_per_gate + slot;
    const int party = __builtin_amdgcn_readfirstlane(a.slot_party[slot]), row = __builtin_amdgcn_readfirstlane(a.slot_row[slot]);
    const uint32_t *at_src = a.lwe + gate * (size_t)a.lwe_stride + (size_t)party * a.n;
    const cplx *brk = a.brk + (size_t)party * a.brk_party_stride;
    // key, monomial and twist tables are read through buffer descriptors (kernel_common.h table_load)
    const __amdgpu_buffer_rsrc_t rs_brk = table_rsrc(brk, (size_t)a.brk_party_stride * sizeof(cplx));
    const __amdgpu_buffer_rsrc_t rs_mono = table_rsrc(a.monomial, (size_t)2 * N * M * sizeof(cplx));
    const __amdgpu_buffer_rsrc_t rs_roots = table_rsrc(a.tw.roots, (size_t)M * sizeof(cplx));
    const __amdgpu_buffer_rsrc_t rs_rinv = table_rsrc(a.tw.rootsinv, (size_t)M * sizeof(cplx));
    unsigned vo_dev[R], vo_nat[R];            // per-lane byte offsets: device point order (tables), e*NT + t (twists)
#pragma unroll
    for (int e = 0; e < R; e++) { vo_dev[e] = (unsigned)dev_pos(MKT_DEVORDER, t * R + e, NT) * 16u; vo_nat[e] = (unsigned)(e * NT + t) * 16u; }
    // LT, BT > 0: gadget length and base known at compile time -- every digit shift / mask is an immediate
    const Gadget<WORD> gd(LT ? LT : a.l, (LT && BT) ? BT : a.logB);
    const int l = LT ? LT : a.l;   // LT > 0: gadget length known at compile time, the digit loop unrolls fully

    // plain kernels: the twist / untwist factors of this thread's points stay in registers (20 of the 51 table loads of a
    // CMux gone); the block kernels have no registers to spare (measured: -12 % at KMS2partyblock) and reload them
    constexpr bool RREG = LB == 1;
    cplx rt_reg[R], ri_reg[R];
#pragma unroll
    for (int e = 0; e < R; e++) {
        if (RREG) { rt_reg[e] = a.tw.roots[e * NT + t]; ri_reg[e] = a.tw.rootsinv[e * NT + t]; }
        else { rt_reg[e].re = rt_reg[e].im = 0.0; ri_reg[e] = rt_reg[e]; }
    }

    WORD acc[2][R][2];
    if (a.init_mode == 0) {
        const WORD *src = reinterpret_cast<const WORD *>(a.acc_io) + rot * 2 * N;
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < R; e++) { acc[c][e][0] = src[c * N + e * NT + t]; acc[c][e][1] = src[c * N + M + e * NT + t]; }
    } else {   // bootstrapping.jl:403-406: b = gvec_lev[row] at X^0, a = 0
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < R; e++) { acc[c][e][0] = 0; acc[c][e][1] = 0; }
        if (t == 0) acc[0][0][0] = (WORD)1 << (W - (row + 1) * a.logB_lev);
    }

    const int nblk = a.n / LB;
    const int msbit = 32 - a.logN - 1;
    uint32_t at_raw[LB];                      // the mask words of the NEXT block are requested a block ahead: nothing waits for them
#pragma unroll
    for (int q = 0; q < LB; q++) at_raw[q] = at_src[q];
    for (int blk = 0; blk < nblk; blk++) {
        uint32_t ats[LB];
        bool any = false;
#pragma unroll
        for (int q = 0; q < LB; q++) {
            ats[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.pre_switched ? at_raw[q] : divbits<uint32_t>(at_raw[q], msbit)));   // bootstrapping.jl:8 (wave-uniform)
            any |= ats[q] != 0;
        }
        {
            const int nb = blk + 1 < nblk ? blk + 1 : blk;
#pragma unroll
            for (int q = 0; q < LB; q++) at_raw[q] = at_src[nb * LB + q];
        }
        if (!any) continue;                                              // :48 / :145 / :413 / :638


        // plain kernels: the monomial row of the step is requested here (its index only depends on the mask word), not
        // right before the multiply where its Infinity-Cache latency was fully exposed (-4 %)
        constexpr bool PF_MONO = LB == 1;
        cplx mono_pf[R];
        if (PF_MONO) {
#pragma unroll
            for (int e = 0; e < R; e++) mono_pf[e] = table_load(rs_mono, vo_dev[e], (unsigned)((size_t)(ats[0] - 1) * M * sizeof(cplx)));
            __builtin_amdgcn_sched_barrier(0);
        }

        cplx tacc[LB][2][R];
#pragma unroll
        for (int q = 0; q < LB; q++)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int e = 0; e < R; e++) { tacc[q][c][e].re = 0.0; tacc[q][c][e].im = 0.0; }

        // the 2l digit polynomials in MAC order (b digits, then a digits: :63-68), NB at a time
#pragma unroll (LT ? 2 * LT : 1)
        for (int g0 = 0; g0 < 2 * l; g0 += NB) {
            cplx z[NB][R];
#pragma unroll
            for (int h2 = 0; h2 < NB; h2++) {
                const int g = g0 + h2;
                const bool isa = g >= l;
                const int j = isa ? g - l : g;
#pragma unroll
                for (int e = 0; e < R; e++) {                            // :50-51 decompto!, fft.jl:57-63 twist
                    const WORD w0 = isa ? acc[1][e][0] : acc[0][e][0], w1 = isa ? acc[1][e][1] : acc[0][e][1];
                    const int d0 = gd.digit(gd.prep(w0), j), d1 = gd.digit(gd.prep(w1), j);
                    cplx v; v.re = (double)d0; v.im = (double)(-d1);
                    z[h2][e] = cmul(v, RREG ? rt_reg[e] : table_load(rs_roots, vo_nat[e], 0));
                }
            }
            fft_forward<LOGM, LR, NB, MO>(z, psi_l, lds, t, xs.lx);      // :54-59 fftto!
#pragma unroll
            for (int h2 = 0; h2 < NB; h2++)
#pragma unroll
                for (int q = 0; q < LB; q++) {
                    if (LB > 1 && ats[q] == 0) continue;
                    const unsigned so_row = (unsigned)((((size_t)((MKT_ROT_ABLATE_KEYS ? 0 : blk) * LB + q) * 2 * l + (size_t)(g0 + h2)) * 2) * M * sizeof(cplx));
#pragma unroll
                    for (int e = 0; e < R; e++) {                        // :63-68 muladdto!(tacc, digit, row)
                        const cplx kb = table_load(rs_brk, vo_dev[e], so_row), ka = table_load(rs_brk, vo_dev[e], so_row + (unsigned)(M * sizeof(cplx)));
                        tacc[q][0][e] = cadd(tacc[q][0][e], cmul(z[h2][e], kb));
                        tacc[q][1][e] = cadd(tacc[q][1][e], cmul(z[h2][e], ka));
                        if (RotOcc<LOGM, NB>::MINW >= 3) __builtin_amdgcn_sched_barrier(0);   // keep the key-row live ranges short at 3 waves/SIMD
                    }
                }
        }

        cplx t2[2][R];
        if (LB == 1 && !a.blk_accum) {                                   // :71 mul!(monomial[atilde], tacc)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int e = 0; e < R; e++) t2[c][e] = cmul(mono_pf[e], tacc[0][c][e]);
        } else {                                                         // :157 / :648 tacc2 += monomial * tacc
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int e = 0; e < R; e++) { t2[c][e].re = 0.0; t2[c][e].im = 0.0; }
#pragma unroll
            for (int q = 0; q < LB; q++) {
                if (ats[q] == 0) continue;
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int e = 0; e < R; e++) t2[c][e] = cadd(t2[c][e], cmul(table_load(rs_mono, vo_dev[e], (unsigned)((size_t)(ats[q] - 1) * M * sizeof(cplx))), tacc[q][c][e]));
            }
        }
        if (NB == 2) {
            fft_inverse<LOGM, LR, 2, true, MO>(t2, psi_l, lds, t, xs.lx);                                   // :72 ifftto! (b and a together)
        } else {
            fft_inverse<LOGM, LR, 1, true, MO>(reinterpret_cast<cplx(&)[1][R]>(t2[0]), psi_l, lds, t, xs.lx);
            fft_inverse<LOGM, LR, 1, true, MO>(reinterpret_cast<cplx(&)[1][R]>(t2[1]), psi_l, lds, t, xs.lx);
        }
#pragma unroll
        for (int e = 0; e < R; e++) {
            const cplx ri = RREG ? ri_reg[e] : table_load(rs_rinv, vo_nat[e], 0);
#pragma unroll
            for (int c = 0; c < 2; c++) {                                // fft.jl:76-80 untwist + native; :73 add!
                const cplx v = cmul(t2[c][e], ri);
                acc[c][e][0] = native_add<WORD>(acc[c][e][0], v.re);
                acc[c][e][1] = native_add<WORD>(acc[c][e][1], -v.im);
            }
        }
    }

    if (a.out_mode == 0) {
        WORD *dst = reinterpret_cast<WORD *>(a.acc_io) + rot * 2 * N;
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < R; e++) { dst[c * N + e * NT + t] = acc[c][e][0]; dst[c * N + M + e * NT + t] = acc[c][e][1]; }
    } else {                                                             // :441 / :657 fftto!(tacc, acc)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            cplx z[1][R];
#pragma unroll
            for (int e = 0; e < R; e++) {
                cplx v; v.re = word_to_f64<WORD>(acc[c][e][0]); v.im = word_to_f64<WORD>((WORD)((WORD)0 - acc[c][e][1]));
                z[0][e] = cmul(v, a.tw.roots[e * NT + t]);
            }
            fft_forward<LOGM, LR, 1, MO>(z, psi_l, lds, t, xs.lx);
            cplx *o = a.tout + (rot * 2 + c) * M;
#pragma unroll
            for (int e = 0; e < R; e++) o[a.tout_natural ? t * R + e : dev_pos(MKT_DEVORDER, t * R + e, NT)] = z[0][e];
        }
    }
}

#endif  // TU 1-3, 5

#if MKT_IN_TU(7)
// ------------------------------------------------------------------------------------------------
// CGGI / LMSS blind rotation with RLWE length KR > 1 (bootstrapping.jl:32-76, :114-165 with k = KR): the general-k
// form of the kernel above for the plain single-key schemes; no shipped parameter set uses it (params.jl:1-13 have
// k = 1), so it is kept simple: one transform at a time, KR+1 accumulators in registers.  BLK (LMSS): the key bits of
// a block share one decomposition of the accumulator (:131-140) and one inverse transform (:162); the digit transforms
// are recomputed per key bit here (same values) instead of being held for the whole block.
// brk layout [n][(KR+1)*l rows][KR+1 polys][M] (device point order), acc [rot][KR+1][N].
// ------------------------------------------------------------------------------------------------
// BL > 0 (LMSS, block length known at compile time): the digit transforms of a block are computed ONCE and multiplied into
// one transform-domain accumulator per key bit of the block ((KR+1) * BL of them in registers), exactly the reference's
// loop nest (:131-158); BL = 0 recomputes them per key bit (any block length).
template <int LOGM, typename WORD, int KR, bool BLK, int BL = 0>
__global__ __launch_bounds__((Plan<LOGM, LOGR>::NT)) void blindrotate_kr_kernel(const RotArgs a) {
    using P = Plan<LOGM, LOGR>;
    constexpr int R = P::R, NT = P::NT, M = P::M, N = 2 * M, NP = KR + 1;
    cplx *lds = reinterpret_cast<cplx *>(mkt_smem);
    cplx *psi_l = lds + P::LDS_CPLX;
    const int t = threadIdx.x;
    XS xs = make_xs();
    for (int i = t; i < M; i += NT) psi_l[i] = a.tw.psi[i];
    __syncthreads();
    const size_t rot = blockIdx.x;
    const uint32_t *at_src = a.lwe + rot * (size_t)a.lwe_stride;
    const Gadget<WORD> gd(a.l, a.logB);
    const int l = a.l;
    int dp[R];
#pragma unroll
    for (int e = 0; e < R; e++) dp[e] = dev_pos(MKT_DEVORDER_KR, t * R + e, NT);   // compile-time: the offsets of the 4 points fold into the addressing
    WORD *accg = reinterpret_cast<WORD *>(a.acc_io) + rot * (size_t)NP * N;
    WORD acc[NP][R][2];
#pragma unroll
    for (int c = 0; c < NP; c++)
#pragma unroll
        for (int e = 0; e < R; e++) { acc[c][e][0] = accg[c * N + e * NT + t]; acc[c][e][1] = accg[c * N + M + e * NT + t]; }
    const int msbit = 32 - a.logN - 1;
    if constexpr (BLK && BL > 0) {
        for (int blk = 0; blk < a.n / BL; blk++) {
            uint32_t ats[BL];
            bool any = false;
#pragma unroll
            for (int q = 0; q < BL; q++) {
                const uint32_t v0 = at_src[blk * BL + q];
                ats[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.pre_switched ? v0 : divbits<uint32_t>(v0, msbit)));
                any |= ats[q] != 0;
            }
            if (!any) continue;                                          // an all-zero block adds native(0) = 0 (:162-163)
            cplx tacc[BL][NP][R];
#pragma unroll
            for (int q = 0; q < BL; q++)
#pragma unroll
                for (int pp = 0; pp < NP; pp++)
#pragma unroll
                    for (int e = 0; e < R; e++) { tacc[q][pp][e].re = 0.0; tacc[q][pp][e].im = 0.0; }
#pragma unroll
            for (int c = 0; c < NP; c++) {                               // :131-140 one decomposition per block
                WORD tp[R][2];
#pragma unroll
                for (int e = 0; e < R; e++) { tp[e][0] = gd.prep(acc[c][e][0]); tp[e][1] = gd.prep(acc[c][e][1]); }
                for (int j = 0; j < l; j++) {
                    cplx z[R];
#pragma unroll
                    for (int e = 0; e < R; e++) {
                        const int d0 = gd.digit(tp[e][0], j), d1 = gd.digit(tp[e][1], j);
                        cplx v; v.re = (double)d0; v.im = (double)(-d1);
                        z[e] = cmul(v, a.tw.roots[e * NT + t]);
                    }
                    fft_forward1<LOGM>(z, psi_l, lds, t, xs);
#pragma unroll
                    for (int q = 0; q < BL; q++) {                       // :146-154, rows in the reference's order for every key bit
                        if (ats[q] == 0) continue;
                        const cplx *row = a.brk + ((size_t)(blk * BL + q) * NP * l + (size_t)(c * l + j)) * NP * M;
#pragma unroll
                        for (int pp = 0; pp < NP; pp++)
#pragma unroll
                            for (int e = 0; e < R; e++) tacc[q][pp][e] = cadd(tacc[q][pp][e], cmul(z[e], row[(size_t)pp * M + dp[e]]));
                    }
                }
            }
#pragma unroll
            for (int pp = 0; pp < NP; pp++) {
                cplx s2[R];
#pragma unroll
                for (int e = 0; e < R; e++) { s2[e].re = 0.0; s2[e].im = 0.0; }
#pragma unroll
                for (int q = 0; q < BL; q++) {                           // :157 tacc2 += monomial * tacc
                    if (ats[q] == 0) continue;
                    const cplx *mono = a.monomial + (size_t)(ats[q] - 1) * M;
#pragma unroll
                    for (int e = 0; e < R; e++) s2[e] = cadd(s2[e], cmul(mono[dp[e]], tacc[q][pp][e]));
                }
                fft_inverse<LOGM, LOGR, 1, true>(reinterpret_cast<cplx(&)[1][R]>(s2), psi_l, lds, t, xs.lx);   // :162-163
#pragma unroll
                for (int e = 0; e < R; e++) {
                    const cplx v = cmul(s2[e], a.tw.rootsinv[e * NT + t]);
                    acc[pp][e][0] = (WORD)(acc[pp][e][0] + native<WORD>(v.re));
                    acc[pp][e][1] = (WORD)(acc[pp][e][1] + native<WORD>(-v.im));
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NP; c++)
#pragma unroll
            for (int e = 0; e < R; e++) { accg[c * N + e * NT + t] = acc[c][e][0]; accg[c * N + M + e * NT + t] = acc[c][e][1]; }
        return;
    }
    const int blen = BLK ? a.blk_len : 1;
    for (int blk = 0; blk < a.n / blen; blk++) {
        cplx t2[BLK ? NP : 1][R];                                        // :142 tacc2 (LMSS only)
        if (BLK) {
#pragma unroll
            for (int q = 0; q < NP; q++)
#pragma unroll
                for (int e = 0; e < R; e++) { t2[BLK ? q : 0][e].re = 0.0; t2[BLK ? q : 0][e].im = 0.0; }
        }
        bool any = false;
        for (int qb = 0; qb < blen; qb++) {
            const int idx = blk * blen + qb;
            const uint32_t v0 = at_src[idx];
            const uint32_t at = a.pre_switched ? v0 : divbits<uint32_t>(v0, msbit);
            if (at == 0) continue;                                       // :48 / :145
            any = true;
            cplx tacc[NP][R];
#pragma unroll
            for (int q = 0; q < NP; q++)
#pragma unroll
                for (int e = 0; e < R; e++) { tacc[q][e].re = 0.0; tacc[q][e].im = 0.0; }
            const cplx *brk = a.brk + (size_t)idx * NP * l * NP * M;
#pragma unroll
            for (int c = 0; c < NP; c++) {                               // b digits, then a_0, a_1 ... (:63-68 / :146-154)
                WORD tp[R][2];
#pragma unroll
                for (int e = 0; e < R; e++) { tp[e][0] = gd.prep(acc[c][e][0]); tp[e][1] = gd.prep(acc[c][e][1]); }
                for (int j = 0; j < l; j++) {
                    cplx z[R];
#pragma unroll
                    for (int e = 0; e < R; e++) {
                        const int d0 = gd.digit(tp[e][0], j), d1 = gd.digit(tp[e][1], j);
                        cplx v; v.re = (double)d0; v.im = (double)(-d1);
                        z[e] = cmul(v, a.tw.roots[e * NT + t]);
                    }
                    fft_forward1<LOGM>(z, psi_l, lds, t, xs);
                    const cplx *row = brk + (size_t)(c * l + j) * NP * M;
#pragma unroll
                    for (int q = 0; q < NP; q++)
#pragma unroll
                        for (int e = 0; e < R; e++) tacc[q][e] = cadd(tacc[q][e], cmul(z[e], row[(size_t)q * M + dp[e]]));
                }
            }
            const cplx *mono = a.monomial + (size_t)(at - 1) * M;
            if (BLK) {                                                   // :157 tacc2 += monomial * tacc
#pragma unroll
                for (int q = 0; q < NP; q++)
#pragma unroll
                    for (int e = 0; e < R; e++) t2[BLK ? q : 0][e] = cadd(t2[BLK ? q : 0][e], cmul(mono[dp[e]], tacc[q][e]));
            } else {
#pragma unroll
                for (int q = 0; q < NP; q++) {                           // :71-73
                    cplx s[R];
#pragma unroll
                    for (int e = 0; e < R; e++) s[e] = cmul(mono[dp[e]], tacc[q][e]);
                    fft_inverse<LOGM, LOGR, 1, true>(reinterpret_cast<cplx(&)[1][R]>(s), psi_l, lds, t, xs.lx);
#pragma unroll
                    for (int e = 0; e < R; e++) {
                        const cplx v = cmul(s[e], a.tw.rootsinv[e * NT + t]);
                        acc[q][e][0] = (WORD)(acc[q][e][0] + native<WORD>(v.re));
                        acc[q][e][1] = (WORD)(acc[q][e][1] + native<WORD>(-v.im));
                    }
                }
            }
        }
        if (BLK && any) {                                                // :162-163 (an all-zero block adds native(0) = 0)
#pragma unroll
            for (int q = 0; q < NP; q++) {
                fft_inverse<LOGM, LOGR, 1, true>(reinterpret_cast<cplx(&)[1][R]>(t2[BLK ? q : 0]), psi_l, lds, t, xs.lx);
#pragma unroll
                for (int e = 0; e < R; e++) {
                    const cplx v = cmul(t2[BLK ? q : 0][e], a.tw.rootsinv[e * NT + t]);
                    acc[q][e][0] = (WORD)(acc[q][e][0] + native<WORD>(v.re));
                    acc[q][e][1] = (WORD)(acc[q][e][1] + native<WORD>(-v.im));
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NP; c++)
#pragma unroll
        for (int e = 0; e < R; e++) { accg[c * N + e * NT + t] = acc[c][e][0]; accg[c * N + M + e * NT + t] = acc[c][e][1]; }
}

// ------------------------------------------------------------------------------------------------
// CGGI / LMSS blind rotation for ANY RLWE length (TFHEparams_bin.k / TFHEparams_block.k are unrestricted, scheme.jl:6-20,
// :22-38): the loop nest of the kernel above with np = k + 1 a run-time value.  What that kernel holds in registers lives in
// memory here -- the accumulator stays in acc_io, the transform-domain sums tacc (:60 / :143) and tacc2 (:142) in a scratch
// area of 2 * np * M points per rotation -- and every thread only ever touches its own coefficients and points, so nothing
// but the transforms needs ordering.  Used above k = 3 only (no shipped set has k > 1); same operation order as
// blindrotate_kr_kernel.
// ------------------------------------------------------------------------------------------------
template <int LOGM, typename WORD>
__global__ __launch_bounds__((Plan<LOGM, LOGR>::NT)) void blindrotate_kany_kernel(const RotArgs a, int np, cplx *scratch) {
    using P = Plan<LOGM, LOGR>;
    constexpr int R = P::R, NT = P::NT, M = P::M, N = 2 * M;
    cplx *lds = reinterpret_cast<cplx *>(mkt_smem);
    cplx *psi_l = lds + P::LDS_CPLX;
    const int t = threadIdx.x;
    XS xs = make_xs();
    for (int i = t; i < M; i += NT) psi_l[i] = a.tw.psi[i];
    __syncthreads();
    const size_t rot = blockIdx.x;
    const uint32_t *at_src = a.lwe + rot * (size_t)a.lwe_stride;
    const Gadget<WORD> gd(a.l, a.logB);
    const int l = a.l;
    int dp[R];
#pragma unroll
    for (int e = 0; e < R; e++) dp[e] = dev_pos(MKT_DEVORDER_KR, t * R + e, NT);
    WORD *accg = reinterpret_cast<WORD *>(a.acc_io) + rot * (size_t)np * N;
    cplx *tacc = scratch + rot * (size_t)2 * np * M, *t2 = tacc + (size_t)np * M;
    const int msbit = 32 - a.logN - 1;
    const bool blk_mode = a.blk_len > 1;
    const int blen = blk_mode ? a.blk_len : 1;
    auto add_native = [&](int q, cplx (&s2)[R]) {                        // acc[q] += native(ifft(s2)) (:71-73 / :162-163)
        fft_inverse<LOGM, LOGR, 1, true>(reinterpret_cast<cplx(&)[1][R]>(s2), psi_l, lds, t, xs.lx);
#pragma unroll
        for (int e = 0; e < R; e++) {
            const cplx v = cmul(s2[e], a.tw.rootsinv[e * NT + t]);
            WORD *lo = accg + (size_t)q * N + e * NT + t, *hi = lo + M;
            *lo = (WORD)(*lo + native<WORD>(v.re));
            *hi = (WORD)(*hi + native<WORD>(-v.im));
        }
    };
    for (int blk = 0; blk < a.n / blen; blk++) {
        if (blk_mode) {
            for (int q = 0; q < np; q++)
#pragma unroll
                for (int e = 0; e < R; e++) { t2[(size_t)q * M + dp[e]].re = 0.0; t2[(size_t)q * M + dp[e]].im = 0.0; }
        }
        bool any = false;
        for (int qb = 0; qb < blen; qb++) {
            const int idx = blk * blen + qb;
            const uint32_t v0 = at_src[idx];
            const uint32_t at = a.pre_switched ? v0 : divbits<uint32_t>(v0, msbit);
            if (at == 0) continue;                                       // :48 / :145
            any = true;
            for (int q = 0; q < np; q++)
#pragma unroll
                for (int e = 0; e < R; e++) { tacc[(size_t)q * M + dp[e]].re = 0.0; tacc[(size_t)q * M + dp[e]].im = 0.0; }
            const cplx *brk = a.brk + (size_t)idx * np * l * np * M;
            for (int c = 0; c < np; c++) {                               // b digits, then a_0, a_1 ... (:63-68 / :146-154)
                WORD tp[R][2];
#pragma unroll
                for (int e = 0; e < R; e++) {
                    tp[e][0] = gd.prep(accg[(size_t)c * N + e * NT + t]);
                    tp[e][1] = gd.prep(accg[(size_t)c * N + M + e * NT + t]);
                }
                for (int j = 0; j < l; j++) {
                    cplx z[R];
#pragma unroll
                    for (int e = 0; e < R; e++) {
                        const int d0 = gd.digit(tp[e][0], j), d1 = gd.digit(tp[e][1], j);
                        cplx v; v.re = (double)d0; v.im = (double)(-d1);
                        z[e] = cmul(v, a.tw.roots[e * NT + t]);
                    }
                    fft_forward1<LOGM>(z, psi_l, lds, t, xs);
                    const cplx *row = brk + (size_t)(c * l + j) * np * M;
                    for (int q = 0; q < np; q++)
#pragma unroll
                        for (int e = 0; e < R; e++) {
                            cplx *ta = tacc + (size_t)q * M + dp[e];
                            *ta = cadd(*ta, cmul(z[e], row[(size_t)q * M + dp[e]]));
                        }
                }
            }
            const cplx *mono = a.monomial + (size_t)(at - 1) * M;
            for (int q = 0; q < np; q++) {
                if (blk_mode) {                                          // :157 tacc2 += monomial * tacc
#pragma unroll
                    for (int e = 0; e < R; e++) {
                        cplx *tb = t2 + (size_t)q * M + dp[e];
                        *tb = cadd(*tb, cmul(mono[dp[e]], tacc[(size_t)q * M + dp[e]]));
                    }
                } else {                                                 // :71-73
                    cplx s2[R];
#pragma unroll
                    for (int e = 0; e < R; e++) s2[e] = cmul(mono[dp[e]], tacc[(size_t)q * M + dp[e]]);
                    add_native(q, s2);
                }
            }
        }
        if (blk_mode && any) {                                           // :162-163 (an all-zero block adds native(0) = 0)
            for (int q = 0; q < np; q++) {
                cplx s2[R];
#pragma unroll
                for (int e = 0; e < R; e++) s2[e] = t2[(size_t)q * M + dp[e]];
                add_native(q, s2);
            }
        }
    }
}

hipError_t launch_blindrotate_kany(int logM, int W, int kr, const RotArgs &a, cplx *scratch, size_t nrot, hipStream_t s) {
    if (!nrot) return hipSuccess;
    if (kr < 1 || !scratch) return hipErrorInvalidValue;
    last_rot_kernel = "blindrotate_kany_kernel";
    MKT_DISPATCH_LOGM(logM, {
        using P = Plan<LM, LOGR>;
        constexpr size_t LB = P::LDS_BYTES + (size_t)P::M * sizeof(cplx);
        if (W == 64) {
            hipError_t e = set_lds(blindrotate_kany_kernel<LM, uint64_t>, LB); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((blindrotate_kany_kernel<LM, uint64_t>), dim3((unsigned)nrot), dim3(P::NT), LB, s, a, kr + 1, scratch);
        } else {
            hipError_t e = set_lds(blindrotate_kany_kernel<LM, uint32_t>, LB); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((blindrotate_kany_kernel<LM, uint32_t>), dim3((unsigned)nrot), dim3(P::NT), LB, s, a, kr + 1, scratch);
        }
    });
    return hipGetLastError();
}
#endif  // TU 7

#if MKT_IN_TU(4)
// ------------------------------------------------------------------------------------------------
// KMS phase 2 (bootstrapping.jl:448-558): k sequential merges, one workgroup per ciphertext.
// < 1 % of the bootstrap's transforms; polynomials stream through a per-ciphertext scratch area,
// every thread only ever touches its own points / coefficients, so no inter-thread ordering is needed
// outside the transforms.
// ------------------------------------------------------------------------------------------------
template <int LOGM, typename WORD>
__global__ __launch_bounds__((Plan<LOGM, LOGR>::NT)) void kms_phase2_kernel(const Phase2Args a) {
    using P = Plan<LOGM, LOGR>;
    constexpr int R = P::R, NT = P::NT, M = P::M, N = 2 * M, W = WordTraits<WORD>::W;
    cplx *lds = reinterpret_cast<cplx *>(mkt_smem);
    const int t = threadIdx.x;
    XS xs = make_xs();
    const size_t g = blockIdx.x;
    const int k = a.k;
    WORD *acc = reinterpret_cast<WORD *>(a.acc) + g * (size_t)(k + 1) * N;
    cplx *tx = a.scratch + g * (size_t)2 * (k + 1) * M;
    cplx *ty2 = tx + (size_t)(k + 1) * M;
    const Gadget<WORD> glev(a.l_lev, a.logB_lev), guni(a.l_uni, a.logB_uni);
    int dp[R];                     // device point order of this thread's slots
#pragma unroll
    for (int e = 0; e < R; e++) dp[e] = dev_pos(MKT_DEVORDER, t * R + e, NT);

    cplx rt[R];
#pragma unroll
    for (int e = 0; e < R; e++) rt[e] = a.tw.roots[e * NT + t];

    if (a.lin) {                                                          // bootstrapping.jl:11-23
        uint32_t tb = divbits<uint32_t>(a.lin[g * a.lwe_stride + a.lwe_stride - 1], 32 - a.logN - 1);
        const WORD ev = (WORD)1 << (W - 3), me = (WORD)((WORD)0 - ev);
        WORD lo_v = ev, hi_v = me;
        if (tb > (uint32_t)N) { tb -= (uint32_t)N; lo_v = me; hi_v = ev; }
#pragma unroll
        for (int e = 0; e < R; e++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int i = h * M + e * NT + t;
                acc[i] = ((uint32_t)i < tb) ? lo_v : hi_v;
                for (int q = 1; q <= k; q++) acc[(size_t)q * N + i] = 0;
            }
    }

    for (int idx = 0; idx < k; idx++) {
        const int iter = idx == 0 ? 1 : a.l_lev;                          // :481
        const int rowbase = idx == 0 ? 0 : 1 + (idx - 1) * a.l_lev;
        const cplx *lev = a.levkey + (g * (size_t)a.rtot + rowbase) * 2 * M;   // stack[r].b, stack[r].a
        const cplx *rd = a.rlk_d + (size_t)idx * a.l_uni * M;
        const cplx *rf = a.rlk_f + (size_t)idx * a.l_uni * 2 * M;
        cplx tv[R];
#pragma unroll
        for (int e = 0; e < R; e++) { tv[e].re = 0.0; tv[e].im = 0.0; }

        for (int q = 0; q <= idx; q++) {                                  // input polys: b, a_0 .. a_{idx-1}
            WORD tp[R][2];
#pragma unroll
            for (int e = 0; e < R; e++) {
                tp[e][0] = glev.prep(acc[(size_t)q * N + e * NT + t]);    // :470-471
                tp[e][1] = glev.prep(acc[(size_t)q * N + M + e * NT + t]);
            }
            cplx txq[R], tyq[R];
#pragma unroll
            for (int e = 0; e < R; e++) { txq[e].re = txq[e].im = 0.0; tyq[e].re = tyq[e].im = 0.0; }
            for (int j = 0; j < iter; j++) {                              // :485-499 LEV multiplication
                cplx z[R];
                digit_points<WORD, R>(z, tp, glev, j, rt);
                fft_forward1<LOGM>(z, a.tw.psi, lds, t, xs);
                const cplx *kb = lev + (size_t)(2 * j) * M, *ka = kb + M;
#pragma unroll
                for (int e = 0; e < R; e++) { txq[e] = cadd(txq[e], cmul(z[e], kb[dp[e]])); tyq[e] = cadd(tyq[e], cmul(z[e], ka[dp[e]])); }
            }
#pragma unroll
            for (int e = 0; e < R; e++) tx[(size_t)q * M + dp[e]] = txq[e];
            WORD yw[R][2];
            inverse_to_words<LOGM, WORD>(tyq, yw, a.tw, lds, t, xs);          // :501-504
#pragma unroll
            for (int e = 0; e < R; e++) { tp[e][0] = guni.prep(yw[e][0]); tp[e][1] = guni.prep(yw[e][1]); }   // :508-509
            cplx tyu[R];
#pragma unroll
            for (int e = 0; e < R; e++) { tyu[e].re = tyu[e].im = 0.0; }
            const cplx *vk = q == 0 ? a.crs : a.pub_b + (size_t)(q - 1) * a.l_uni * M;
            for (int j = 0; j < a.l_uni; j++) {                           // :521-535 u and v
                cplx z[R];
                digit_points<WORD, R>(z, tp, guni, j, rt);
                fft_forward1<LOGM>(z, a.tw.psi, lds, t, xs);
                const cplx *kd = rd + (size_t)j * M, *kv = vk + (size_t)j * M;
#pragma unroll
                for (int e = 0; e < R; e++) {
                    tyu[e] = cadd(tyu[e], cmul(z[e], kd[dp[e]]));
                    const cplx pr = cmul(z[e], kv[dp[e]]);
                    tv[e] = q == 0 ? csub(tv[e], pr) : cadd(tv[e], pr);   // mulsubto! with crs, muladdto! with b_i
                }
            }
#pragma unroll
            for (int e = 0; e < R; e++) ty2[(size_t)q * M + dp[e]] = tyu[e];
        }

        WORD vw[R][2];
        inverse_to_words<LOGM, WORD>(tv, vw, a.tw, lds, t, xs);               // :538
        WORD tp[R][2];
#pragma unroll
        for (int e = 0; e < R; e++) { tp[e][0] = guni.prep(vw[e][0]); tp[e][1] = guni.prep(vw[e][1]); }      // :541
        cplx tyb[R], tya[R];
#pragma unroll
        for (int e = 0; e < R; e++) { tyb[e] = ty2[dp[e]]; tya[e].re = tya[e].im = 0.0; }
        for (int i = 0; i < a.l_uni; i++) {                               // :547-550 w
            cplx z[R];
            digit_points<WORD, R>(z, tp, guni, i, rt);
            fft_forward1<LOGM>(z, a.tw.psi, lds, t, xs);
            const cplx *fb = rf + (size_t)(2 * i) * M, *fa = fb + M;
#pragma unroll
            for (int e = 0; e < R; e++) { tyb[e] = cadd(tyb[e], cmul(z[e], fb[dp[e]])); tya[e] = cadd(tya[e], cmul(z[e], fa[dp[e]])); }
        }
        // :553 add!(tx, ty); :556 ifftto!(acc, tx)
        for (int q = 0; q <= idx + 1; q++) {
            cplx s[R];
#pragma unroll
            for (int e = 0; e < R; e++) {
                cplx xv, yv;
                if (q <= idx) xv = tx[(size_t)q * M + dp[e]]; else { xv.re = 0.0; xv.im = 0.0; }
                if (q == 0) yv = tyb[e]; else if (q == idx + 1) yv = tya[e]; else yv = ty2[(size_t)q * M + dp[e]];
                s[e] = cadd(xv, yv);
            }
            WORD w[R][2];
            inverse_to_words<LOGM, WORD>(s, w, a.tw, lds, t, xs);
#pragma unroll
            for (int e = 0; e < R; e++) { acc[(size_t)q * N + e * NT + t] = w[e][0]; acc[(size_t)q * N + M + e * NT + t] = w[e][1]; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// CCS blind rotation (bootstrapping.jl:234-328): k*n hybrid products on a growing prefix of the mask.
// One workgroup per ciphertext; the accumulator lives in the caller's buffer (L2-resident), transform-domain
// partial results in a per-ciphertext scratch; every thread only touches its own points / coefficients.
// Floating-point summation order is the reference's: u (:279-284), then w with the b digits of v0 (:313-316),
// then w with the digits of v_1..v_np (:317-320) -- which is why u for the current party's mask polynomial is
// computed first and parked in registers.
// ------------------------------------------------------------------------------------------------
// development only: timing builds with parts of the memory traffic removed (WRONG results) -- 1: no transform-domain scratch stores / loads,
// 2: no accumulator stores, 4: every step reads the key rows of step 0 (cache-resident) (tools/tu_variant.sh x 4 "-DMKT_CCS_ABLATE=..."; profiles/r05_experiments.txt)
#ifndef MKT_CCS_ABLATE
#define MKT_CCS_ABLATE 0
#endif
// LT, BT > 0: gadget length and base known at compile time (digit loops unrolled, shifts and masks immediates)
template <int LOGM, typename WORD, int LT = 0, int BT = 0>
__global__ __launch_bounds__((Plan<LOGM, LOGR>::NT)) __attribute__((amdgpu_waves_per_eu(2, 2))) void ccs_blindrotate_kernel(const CcsArgs a) {
    using P = Plan<LOGM, LOGR, 1>;
    constexpr int R = P::R, NT = P::NT, M = P::M, N = 2 * M;
    constexpr int MO1 = 0xff;                                  // library-default exchange routes (twiddles read ahead of the exchanges, MO bit 10: -0.7 % here, not used)
    cplx *lds = reinterpret_cast<cplx *>(mkt_smem);
    cplx *psi_l = lds + P::LDS_CPLX;
    const int t = threadIdx.x;
    XS xs = make_xs();
    for (int i = t; i < M; i += NT) psi_l[i] = a.tw.psi[i];
    __syncthreads();
    const size_t g = blockIdx.x;
    // the workgroups that share a compute unit (every 256th under round-robin dispatch) run the same loop in lock-step
    // and then want the VALU and the LDS at the same moments: a start-up delay de-phases them (speed only)
    for (int s = 0, ns = a.stagger * (int)((g >> 8) & 3); s < ns; s++) __builtin_amdgcn_s_sleep(1);
    const int k = a.k, l = LT ? LT : a.l, n = a.n;
    WORD *acc = reinterpret_cast<WORD *>(a.acc) + g * (size_t)(k + 1) * N;
    cplx *sc = a.scratch + g * (size_t)(k + 1) * M;
    WORD *vsc = reinterpret_cast<WORD *>(a.vscratch) + g * (size_t)N;
    const Gadget<WORD> gd(l, (LT && BT) ? BT : a.logB);
    const int msbit = 32 - a.logN - 1;
    int dp[R];
#pragma unroll
    for (int e = 0; e < R; e++) dp[e] = dev_pos(MKT_DEVORDER, t * R + e, NT);
    cplx rt[R];
#pragma unroll
    for (int e = 0; e < R; e++) rt[e] = a.tw.roots[e * NT + t];

    // u and v of one input polynomial q (:279-294): tu = sum_j dig_j * d[j]; tv = -/+ sum_j dig_j * (crs | b_{q-1})[j]
    auto uv = [&](int q, const cplx *ud, cplx (&tu)[R], cplx (&tvq)[R]) {
        WORD tp[R][2];
#pragma unroll
        for (int e = 0; e < R; e++) {
            tp[e][0] = gd.prep(acc[(size_t)q * N + e * NT + t]); tp[e][1] = gd.prep(acc[(size_t)q * N + M + e * NT + t]);
        }
#pragma unroll
        for (int e = 0; e < R; e++) { tu[e].re = tu[e].im = 0.0; tvq[e].re = tvq[e].im = 0.0; }
        const cplx *vk = (q == 0 || (MKT_CCS_ABLATE & 4)) ? a.crs : a.pub_b + (size_t)(q - 1) * l * M;
#pragma unroll 1
        for (int j = 0; j < l; j++) {
            cplx z[R];
            const cplx *kd = ud + (size_t)j * M, *kv = vk + (size_t)j * M;
            cplx kdr[R], kvr[R];                                                 // key rows requested before the transform that needs them (+1..3 %)
#pragma unroll
            for (int e = 0; e < R; e++) { kdr[e] = kd[dp[e]]; kvr[e] = kv[dp[e]]; }
            __builtin_amdgcn_sched_barrier(0);
            digit_points<WORD, R>(z, tp, gd, j, rt);
            fft_forward<LOGM, LOGR, 1, MO1>(reinterpret_cast<cplx(&)[1][R]>(z), psi_l, lds, t, xs.lx);
#pragma unroll
            for (int e = 0; e < R; e++) {
                tu[e] = cadd(tu[e], cmul(z[e], kdr[e]));
                const cplx pr = cmul(z[e], kvr[e]);
                tvq[e] = q == 0 ? csub(tvq[e], pr) : cadd(tvq[e], pr);          // :290 mulsubto!, :293 muladdto!
            }
        }
    };
    // w contribution of one v polynomial given as words (:313-320)
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
            fft_forward<LOGM, LOGR, 1, MO1>(reinterpret_cast<cplx(&)[1][R]>(z), psi_l, lds, t, xs.lx);
#pragma unroll
            for (int e = 0; e < R; e++) { tb[e] = cadd(tb[e], cmul(z[e], fbr[e])); ta[e] = cadd(ta[e], cmul(z[e], far[e])); }
        }
    };
#ifndef MKT_CCS_RI_AHEAD
#define MKT_CCS_RI_AHEAD 1
#endif
    auto inv_words = [&](cplx (&z)[R], WORD (&w)[R][2]) {                        // fft.jl:74-81
        cplx ri[R];                                                              // untwist factors requested before the transform (at use: an exposed round trip per inverse)
        if (MKT_CCS_RI_AHEAD) {
#pragma unroll
            for (int e = 0; e < R; e++) ri[e] = a.tw.rootsinv[e * NT + t];
            __builtin_amdgcn_sched_barrier(0);
        }
        fft_inverse<LOGM, LOGR, 1, true, MO1>(reinterpret_cast<cplx(&)[1][R]>(z), psi_l, lds, t, xs.lx);
#pragma unroll
        for (int e = 0; e < R; e++) {
            const cplx v = cmul(z[e], MKT_CCS_RI_AHEAD ? ri[e] : a.tw.rootsinv[e * NT + t]);
            w[e][0] = native<WORD>(v.re); w[e][1] = native<WORD>(-v.im);
        }
    };

    for (int idx = 0; idx < k; idx++) {
        const int np = idx + 1;
        const uint32_t *at_src = a.lwe + g * (size_t)a.lwe_stride + (size_t)idx * n;
        for (int i = 0; i < n; i++) {
            const uint32_t v0 = at_src[i];
            const uint32_t at = a.pre_switched ? v0 : divbits<uint32_t>(v0, msbit);
            if (at == 0) continue;                                               // :261
            const cplx *uni = (MKT_CCS_ABLATE & 4) ? a.brk : a.brk + (size_t)idx * a.brk_party_stride + (size_t)i * 3 * l * M;   // ablation 4: every step reads the rows of step 0 (cache-resident)
            const cplx *ud = uni, *uf = uni + (size_t)l * M;
            // One loop over the input polynomials in the order the reference's sums need: the current party's mask
            // polynomial first (its u opens tacc.a[idx], :279-284; its v is parked), then b (u opens tacc.b; w(v0),
            // :313-316), then the earlier parties' mask polynomials (:317-320, j1 = q), last the parked v (j1 = np).
            // Each transform kind appears once in the loop body: the code of a step stays small.
            cplx ta[R], tb[R];
            for (int qi = 0; qi <= np + 1; qi++) {
                WORD vw[R][2];
                if (qi <= np) {
                    const int q = qi == 0 ? np : qi - 1;
                    cplx tu[R], tvq[R];
                    uv(q, ud, tu, tvq);
                    if (q == np) {
#pragma unroll
                        for (int e = 0; e < R; e++) ta[e] = tu[e];
                    } else if (q == 0) {
#pragma unroll
                        for (int e = 0; e < R; e++) tb[e] = tu[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < R; e++) if (!(MKT_CCS_ABLATE & 1)) sc[(size_t)q * M + dp[e]] = tu[e];
                    }
                    inv_words(tvq, vw);                                          // :297-300
                    if (q == np) {
#pragma unroll
                        for (int e = 0; e < R; e++) if (!(MKT_CCS_ABLATE & 1)) { vsc[e * NT + t] = vw[e][0]; vsc[M + e * NT + t] = vw[e][1]; }
                        continue;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < R; e++) { if (MKT_CCS_ABLATE & 1) { vw[e][0] = (WORD)(t * 2654435761u + e); vw[e][1] = (WORD)(t * 40503u + e); } else { vw[e][0] = vsc[e * NT + t]; vw[e][1] = vsc[M + e * NT + t]; } }
                }
                wpart(vw, uf, tb, ta);
            }
            // :322-324 mul!(monomial, tacc); ifftto!; add!  -- tacc.b and tacc.a[idx] join the others in the scratch so
            // the loop below is uniform; the polynomials are independent, the two just stored go last
#pragma unroll
            for (int e = 0; e < R; e++) if (!(MKT_CCS_ABLATE & 1)) { sc[dp[e]] = tb[e]; sc[(size_t)np * M + dp[e]] = ta[e]; }
            const cplx *mono = a.monomial + (size_t)(at - 1) * M;
            // the monomial row is the same for every polynomial; the next polynomial's transform-domain sum and the
            // accumulator words the result is added to are requested before the inverse transform that hides them
            auto fq = [&](int qq) { return qq < np - 1 ? qq + 1 : (qq == np - 1 ? 0 : np); };
            cplx mrow[R], xq[R];
#pragma unroll
            for (int e = 0; e < R; e++) { mrow[e] = mono[dp[e]]; xq[e] = (MKT_CCS_ABLATE & 1) ? tb[e] : sc[(size_t)fq(0) * M + dp[e]]; }
            for (int qq = 0; qq <= np; qq++) {
                const int q = fq(qq), qn = fq(qq < np ? qq + 1 : qq);
                cplx s[R];
#pragma unroll
                for (int e = 0; e < R; e++) s[e] = cmul(mrow[e], xq[e]);
                WORD aw[R][2];
#pragma unroll
                for (int e = 0; e < R; e++) { xq[e] = (MKT_CCS_ABLATE & 1) ? ta[e] : sc[(size_t)qn * M + dp[e]]; aw[e][0] = acc[(size_t)q * N + e * NT + t]; aw[e][1] = acc[(size_t)q * N + M + e * NT + t]; }
                __builtin_amdgcn_sched_barrier(0);
                WORD w[R][2];
                inv_words(s, w);
#pragma unroll
                for (int e = 0; e < R; e++) {
                    if ((MKT_CCS_ABLATE & 2) && w[e][0] != 12345u) continue;
                    acc[(size_t)q * N + e * NT + t] = (WORD)(aw[e][0] + w[e][0]);
                    acc[(size_t)q * N + M + e * NT + t] = (WORD)(aw[e][1] + w[e][1]);
                }
            }
        }
    }
}

#endif  // TU 4

#if MKT_IN_TU(0)
// ------------------------------------------------------------------------------------------------
// Sample extract + LWE key switch.  bootstrapping.jl:81-109 (CGGI), :170-229 (LMSS), :333-364 (CCS),
// :564-594 (KMS), :664-695 (KMS_block).  Gather-accumulate of pre-multiplied LWE rows; u32 wrap adds
// are order independent, so the per-party partial sums (and the atomics on b) are deterministic.
// grid = (B, parties or 1); 256 threads, thread t owns output words t, t+256, ...
// ------------------------------------------------------------------------------------------------
template <typename WORD>
__device__ __forceinline__ uint32_t extract_word(const WORD *a, int j, int N) {   // :91,:99 ; KMS :575,:583
    constexpr int sh = WordTraits<WORD>::W - 32;
    if (j == 0) return (uint32_t)(a[0] >> sh);
    return 0u - (uint32_t)(a[N - j] >> sh);
}

// out = 0 except: b = acc.b[0] >> (W-32) (:86 / :569) and, for the block schemes, the extracted words that are
// copied instead of switched (:180-191, :676-679).  The key-switch kernel then accumulates with atomics.
template <typename WORD>
__global__ void ks_init_kernel(const KsArgs a, size_t B) {
    constexpr int sh = WordTraits<WORD>::W - 32;
    const int nblocks_out = a.mk ? a.kacc : 1;
    const int lwe_len = nblocks_out * a.n + 1;
    const size_t total = B * (size_t)lwe_len;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t g = i / lwe_len; const int q = (int)(i % lwe_len);
        const WORD *accg = reinterpret_cast<const WORD *>(a.acc) + g * (size_t)(1 + a.kacc) * a.N;
        uint32_t v = 0;
        if (q == lwe_len - 1) v = (uint32_t)(accg[0] >> sh);
        else if (a.balanced) {
            if (a.lmss) { const int c = q / a.N, j = q % a.N; v = extract_word<WORD>(accg + (size_t)(1 + c) * a.N, j, a.N); }
            else { const int c = q / a.n, j = q % a.n; v = extract_word<WORD>(accg + (size_t)(1 + c) * a.N, j, a.N); }
        }
        a.out[i] = v;
    }
}

// Multi-gate key switch.  One wave handles G ciphertexts, one slab of the extracted coefficients j and one
// 256-word column chunk of the LWE rows.  Every pre-multiplied row ksk[j][d][t] is loaded ONCE per wave (16 B/lane,
// rows padded to n1p words) and parked in a lane-private LDS table indexed by digit; each ciphertext then picks
// its row with one ds_read_b128 at a wave-uniform offset (LDS is used as an indexable register file: no lane ever
// reads another lane's data, so there are no barriers).  3 row loads per (j,t) instead of 0.75*G is what matters:
// the gather is bound by L2 / Infinity-Cache bandwidth.  Partial sums over slabs meet in u32 atomics; wrap-around
// addition is order independent, so the result is deterministic.
constexpr int KS_LANES = 64, KS_CHUNK_WORDS = 4 * KS_LANES, KS_STAGES = 2;
constexpr int KS_BATCH = 8;

// WAVES > 1: the waves of a workgroup (each with its own G ciphertexts) share one staged table -- every row is fetched
// from L2 / Infinity Cache once per WAVES*G ciphertexts; one barrier per stage.  The two stage buffers alternate on a
// counter that runs over the whole (c, j, td) loop, so the buffer a wave overwrites is always the one whose readers
// have passed the barrier in between (a per-j parity would reuse stage 0 back to back when f is odd).
// BAL: balanced (signed) digits of the block schemes -- a template flag so the unbalanced path carries none of the
// sign handling on the scalar unit (one per CU, and the busiest unit of this kernel)
template <typename WORD, int G, int WAVES, bool BAL>
__global__ __launch_bounds__(KS_LANES * WAVES, G == 32 ? 3 : 1) void keyswitch_mg_kernel(const KsArgs a, int B, int ngroups, int jslab) {
    // digit table: [stage][1 + drows (+ drows negated rows for balanced digits)][lane]
    uint4 *tabp = reinterpret_cast<uint4 *>(mkt_smem);
    const int trows = 1 + a.drows * (BAL ? 2 : 1);
#define tab(s, r, l) tabp[((s) * trows + (r)) * KS_LANES + (l)]
    const int lane = threadIdx.x & (KS_LANES - 1);
    const int wv = WAVES > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x / KS_LANES)) : 0;   // wave-uniform, and known to be
    const int gblocks = (ngroups + WAVES - 1) / WAVES;
    const int gg = (int)(blockIdx.x % (unsigned)gblocks) * WAVES + wv, slab = (int)(blockIdx.x / (unsigned)gblocks);
    const int N = a.N, n = a.n, n1p = a.n1p, f = a.f, logD = a.logD;
    const int q0 = (int)blockIdx.z * KS_CHUNK_WORDS + 4 * lane;
    const bool active = q0 < n1p;
    const int nblocks_out = a.mk ? a.kacc : 1;
    const int lwe_len = nblocks_out * n + 1;
    const int c_begin = a.mk ? (int)blockIdx.y : 0, c_end = a.mk ? c_begin + 1 : a.kacc;
    const uint32_t Dm = (1u << logD) - 1;
    const int half = 1 << (logD - 1);
    const Gadget<uint32_t> gb(f, logD);
    const size_t comp_words = (size_t)N * a.drows * f * n1p;
    const int g_base = gg * G;
    const int drows = a.drows;
    uint4 sum[G];
#pragma unroll
    for (int g = 0; g < G; g++) sum[g] = make_uint4(0, 0, 0, 0);
    if (wv == 0) {
#pragma unroll
        for (int s = 0; s < KS_STAGES; s++) tab(s, 0, lane) = make_uint4(0, 0, 0, 0);   // digit 0 adds nothing
    }

    int it = 0;   // stage counter over the WHOLE (c, j, td) loop: consecutive stagings always alternate buffers, also for odd f
    for (int c = c_begin; c < c_end; c++) {
        const uint32_t *ksk = a.mk ? a.ksk + (size_t)c * a.ksk_party_stride : a.ksk + (size_t)c * comp_words;
        int jstart = 0;
        if (BAL) {
            if (a.lmss) { const long cur = (long)c * N; jstart = cur >= n ? 0 : (cur + N <= n ? N : (int)(n - cur)); }
            else jstart = n;
        }
        int j0 = slab * jslab, j1 = j0 + jslab;
        if (j0 < jstart) j0 = jstart;
        if (j1 > N) j1 = N;
        for (int j = j0; j < j1; j++) {
            uint32_t tt[G];   // wave-uniform: the prepared digit word of every ciphertext of the group
#pragma unroll
            for (int g = 0; g < G; g++) {
                const int gi = g_base + g < B ? g_base + g : B - 1;
                const WORD *ac = reinterpret_cast<const WORD *>(a.acc) + ((size_t)gi * (1 + a.kacc) + 1 + c) * N;
                const uint32_t w = extract_word<WORD>(ac, j, N);
                tt[g] = BAL ? gb.prep(w) : divbits<uint32_t>(w, 32 - f * logD);   // gsw.jl:42-52 / :34-40
            }
            const uint32_t *rowj = ksk + (size_t)j * drows * f * n1p + q0;
            for (int td = 0; td < f; td++) {
                const int st = (it++) & (KS_STAGES - 1);
                const int shift = logD * (f - 1 - td);
                for (int d = 1 + wv; d <= drows; d += WAVES) {              // the waves share the row fetches
                    uint4 r = make_uint4(0, 0, 0, 0);
                    if (active) r = *reinterpret_cast<const uint4 *>(rowj + ((size_t)(d - 1) * f + td) * n1p);
                    tab(st, d, lane) = r;
                    if (BAL) tab(st, drows + d, lane) = make_uint4(0u - r.x, 0u - r.y, 0u - r.z, 0u - r.w);
                }
                if (WAVES > 1) __syncthreads();
                // KS_BATCH table reads are issued back to back before their sums: one LDS latency per batch instead of
                // one per ciphertext
#pragma unroll
                for (int g0 = 0; g0 < G; g0 += KS_BATCH) {
                    uint4 v[KS_BATCH];
#pragma unroll
                    for (int u = 0; u < KS_BATCH; u++) {
                        int idx = (int)((tt[g0 + u] >> shift) & Dm);
                        if (BAL) { idx -= half; if (idx < 0) idx = drows - idx; }   // -1 -> drows+1, -2 -> drows+2
                        v[u] = tab(st, idx, lane);
                    }
#pragma unroll
                    for (int u = 0; u < KS_BATCH; u++) {
                        sum[g0 + u].x += v[u].x; sum[g0 + u].y += v[u].y; sum[g0 + u].z += v[u].z; sum[g0 + u].w += v[u].w;
                    }
                }
            }
        }
    }
#undef tab
    if (!active) return;
    const int blk = a.mk ? c_begin : 0;
#pragma unroll
    for (int g = 0; g < G; g++) {
        if (g_base + g >= B) break;
        uint32_t *outg = a.out + (size_t)(g_base + g) * lwe_len;
        const uint32_t v[4] = {sum[g].x, sum[g].y, sum[g].z, sum[g].w};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = q0 + u;
            if (q < n) { if (v[u]) atomicAdd(&outg[(size_t)blk * n + q], v[u]); }
            else if (q == n) { if (v[u]) atomicAdd(&outg[lwe_len - 1], v[u]); }
        }
    }
}


// The prepared digit word of every extracted coefficient (gsw.jl:34-52 on the sample-extracted mask, bootstrapping.jl:91,:99 / :575,:583),
// once per key switch: [component][group of 32 ciphertexts][j][32] so that a wave of the pair kernel reads the words of its 32
// ciphertexts for one coefficient as 128 contiguous bytes.  Block = 32 ciphertexts x 8 coefficients: reads are one or two 64-byte
// lines per ciphertext, writes 1 KiB contiguous.
template <typename WORD, bool BAL>
__global__ __launch_bounds__(256) void ks_digits_kernel(const KsArgs a, int B, int ngroups) {
    const int g = threadIdx.x & 31, j = (int)blockIdx.x * 8 + (int)(threadIdx.x >> 5);
    if (j >= a.N) return;
    const int grp = blockIdx.y, c = blockIdx.z;
    const int gi = grp * 32 + g < B ? grp * 32 + g : B - 1;
    const WORD *ac = reinterpret_cast<const WORD *>(a.acc) + ((size_t)gi * (1 + a.kacc) + 1 + c) * a.N;
    const uint32_t w = extract_word<WORD>(ac, j, a.N);
    const Gadget<uint32_t> gb(a.f, 2);
    a.digits[(((size_t)c * ngroups + grp) * a.N + j) * 32 + g] = BAL ? gb.prep(w) : divbits<uint32_t>(w, 32 - a.f * 2);
}

// Digit pairs (D = 4, f even: every shipped set has f = 8) -- the key switch of those sets.  Same tiling as the per-digit kernel above
// (a wave = 32 ciphertexts x a slab of coefficients x a 256-word column chunk, sums in registers, rows through a lane-private LDS
// table), but the staged table holds the 16 sums row(d1, td) + row(d2, td+1) of two consecutive digits, built by the waves that
// share it (four row loads, twelve adds, four ds_write_b128 per wave and pair): ONE ds_read_b128 and four adds per ciphertext and
// PAIR of digits.  The index is the raw 4-bit field of the prepared word for both digit kinds (balanced digits: the sign lives in
// the table).  The digit words come prepared (ks_digits_kernel: two s_load_dwordx16 per coefficient), the sums leave as plain
// stores into per-slab partial rows (ks_reduce_kernel adds them: no atomics), and the slabs cut only the coefficients that are
// switched.  Wrap-around addition is associative and commutative, so every grouping gives the per-digit kernel's words.
// 168 VGPRs (three 4-wave workgroups per CU: the launch is sized for exactly one such round), table reads four at a time.
#ifndef MKT_KSP_BATCH
#define MKT_KSP_BATCH 4
#endif
#ifndef MKT_KSP_OCC
#define MKT_KSP_OCC 3
#endif
constexpr int KSP_BATCH = MKT_KSP_BATCH;
template <typename WORD, int G, int WAVES, bool BAL>
__global__ __launch_bounds__(KS_LANES * WAVES, MKT_KSP_OCC) void keyswitch_pair_kernel(const KsArgs a, int B, int ngroups) {
    uint4 *tabp = reinterpret_cast<uint4 *>(mkt_smem);   // [stage][16][lane]
#define tab(s, r, l) tabp[((s) * 16 + (r)) * KS_LANES + (l)]
    const int lane = threadIdx.x & (KS_LANES - 1);
    const int wv = WAVES > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x / KS_LANES)) : 0;
    const int gblocks = (ngroups + WAVES - 1) / WAVES;
    const int gg = (int)(blockIdx.x % (unsigned)gblocks) * WAVES + wv, slab = (int)(blockIdx.x / (unsigned)gblocks);
    const int N = a.N, n = a.n, n1p = a.n1p, f = a.f;
    const int q0 = (int)blockIdx.z * KS_CHUNK_WORDS + 4 * lane;
    const bool active = q0 < n1p;
    const int c_begin = a.mk ? (int)blockIdx.y : 0, c_end = a.mk ? c_begin + 1 : a.kacc;
    const int drows = a.drows;
    const size_t comp_words = (size_t)N * drows * f * n1p;
    const int g_base = gg * G;
    uint4 sum[G];
#pragma unroll
    for (int g = 0; g < G; g++) sum[g] = make_uint4(0, 0, 0, 0);
    const uint4 zero = make_uint4(0, 0, 0, 0);
    auto neg = [](uint4 r) { return make_uint4(0u - r.x, 0u - r.y, 0u - r.z, 0u - r.w); };
    auto add = [](uint4 x, uint4 y) { return make_uint4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w); };

    int it = 0;
    for (int c = c_begin; c < c_end; c++) {
        const uint32_t *ksk = a.mk ? a.ksk + (size_t)c * a.ksk_party_stride : a.ksk + (size_t)c * comp_words;
        int jstart = 0;
        if (BAL) {
            if (a.lmss) { const long cur = (long)c * N; jstart = cur >= n ? 0 : (cur + N <= n ? N : (int)(n - cur)); }
            else jstart = n;
        }
        // the slabs share the coefficients that ARE switched, [jstart, N): equal work per slab also for the block schemes, whose
        // first n coefficients are copied (Blockparam: 337 of 1024 left)
        const int nslabs = (int)(gridDim.x / (unsigned)gblocks), js = (N - jstart + nslabs - 1) / nslabs;
        const int j0 = jstart + slab * js;
        int j1 = j0 + js;
        if (j1 > N) j1 = N;
        // the prepared digit words of this wave's 32 ciphertexts: 128 contiguous bytes per coefficient (ks_digits_kernel), read on
        // the scalar unit -- extracting them here cost 32 scattered scalar loads per coefficient, 40 % of the kernel
        const uint32_t *dg = a.digits + (((size_t)c * ngroups + (gg < ngroups ? gg : ngroups - 1)) * N) * G;
        for (int j = j0; j < j1; j++) {
            uint32_t tt[G];
#pragma unroll
            for (int g = 0; g < G; g++) tt[g] = dg[(size_t)j * G + g];
            const uint32_t *rowj = ksk + (size_t)j * drows * f * n1p + q0;
            auto ld = [=](int r, int t) { return active ? *reinterpret_cast<const uint4 *>(rowj + ((size_t)r * f + t) * n1p) : make_uint4(0, 0, 0, 0); };
            for (int td = 0; td < f; td += 2) {
                const int st = (it++) & (KS_STAGES - 1);
                const int shift = 2 * (f - 2 - td);
                // second digit: values 0, 1, 2, 3 (rows 1..3) or -2, -1, 0, 1 (balanced: rows 2, 1 negated, nothing, row 1)
                uint4 r2[BAL ? 2 : 3];
#pragma unroll
                for (int r = 0; r < (BAL ? 2 : 3); r++) r2[r] = ld(r, td + 1);
                for (int d1 = wv; d1 < 4; d1 += WAVES) {    // the waves share the sixteen sums
                    uint4 e1;
                    if (BAL) e1 = d1 == 2 ? zero : (d1 == 3 ? ld(0, td) : neg(ld(1 - d1, td)));
                    else e1 = d1 ? ld(d1 - 1, td) : zero;
                    if constexpr (BAL) {
                        tab(st, d1 * 4 + 0, lane) = add(e1, neg(r2[1])); tab(st, d1 * 4 + 1, lane) = add(e1, neg(r2[0]));
                        tab(st, d1 * 4 + 2, lane) = e1; tab(st, d1 * 4 + 3, lane) = add(e1, r2[0]);
                    } else {
                        tab(st, d1 * 4 + 0, lane) = e1; tab(st, d1 * 4 + 1, lane) = add(e1, r2[0]);
                        tab(st, d1 * 4 + 2, lane) = add(e1, r2[1]); tab(st, d1 * 4 + 3, lane) = add(e1, r2[2]);
                    }
                }
                if (WAVES > 1) __syncthreads();
#pragma unroll
                for (int g0 = 0; g0 < G; g0 += KSP_BATCH) {
                    uint4 v[KSP_BATCH];
#pragma unroll
                    for (int u = 0; u < KSP_BATCH; u++) v[u] = tab(st, (int)((tt[g0 + u] >> shift) & 15u), lane);
#pragma unroll
                    for (int u = 0; u < KSP_BATCH; u++) {
                        sum[g0 + u].x += v[u].x; sum[g0 + u].y += v[u].y; sum[g0 + u].z += v[u].z; sum[g0 + u].w += v[u].w;
                    }
                }
            }
        }
    }
#undef tab
    if (!active) return;
    // this wave's share of the sum over (component, coefficient, digit): one 16-byte store per ciphertext into the partial-sum rows
    // [slab][party][ciphertext][n1p]; ks_reduce_kernel adds the slabs (atomics on the output words cost 0.2 ms of a 0.6 ms kernel)
    const int blk = a.mk ? c_begin : 0;
#pragma unroll
    for (int g = 0; g < G; g++) {
        if (g_base + g >= B) break;
        *reinterpret_cast<uint4 *>(a.partial + (((size_t)slab * gridDim.y + blk) * B + (g_base + g)) * n1p + q0) = sum[g];
    }
}

// out = what the key switch leaves alone (b, and for the block schemes the extracted words that are copied, :180-191 / :676-679:
// ks_init_kernel's words) + the partial sums of every slab (and, for b, of every party).
template <typename WORD>
__global__ void ks_reduce_kernel(const KsArgs a, size_t B, int slabs, int parties) {
    constexpr int sh = WordTraits<WORD>::W - 32;
    const int nblocks_out = a.mk ? a.kacc : 1;
    const int lwe_len = nblocks_out * a.n + 1;
    const size_t total = B * (size_t)lwe_len;
    const size_t row = (size_t)a.n1p, slab_stride = (size_t)parties * B * row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t g = i / lwe_len; const int q = (int)(i % lwe_len);
        const WORD *accg = reinterpret_cast<const WORD *>(a.acc) + g * (size_t)(1 + a.kacc) * a.N;
        uint32_t v = 0;
        if (q == lwe_len - 1) {
            v = (uint32_t)(accg[0] >> sh);
            for (int sl = 0; sl < slabs; sl++)
                for (int pt = 0; pt < parties; pt++) v += a.partial[sl * slab_stride + ((size_t)pt * B + g) * row + a.n];
        } else {
            if (a.balanced) {
                if (a.lmss) { const int c = q / a.N, j = q % a.N; v = extract_word<WORD>(accg + (size_t)(1 + c) * a.N, j, a.N); }
                else { const int c = q / a.n, j = q % a.n; v = extract_word<WORD>(accg + (size_t)(1 + c) * a.N, j, a.N); }
            }
            const int pt = a.mk ? q / a.n : 0, w = a.mk ? q % a.n : q;
            const uint32_t *src = a.partial + ((size_t)pt * B + g) * row + w;
            for (int sl = 0; sl < slabs; sl++) v += src[sl * slab_stride];
        }
        a.out[i] = v;
    }
}

#endif  // TU 0

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
#if MKT_IN_TU(0)
bool transform_supported(int logM) { return logM >= 4 && logM <= 11; }
#endif

#if MKT_IN_TU(0)
template <int LM, typename WORD, int NBT>
static hipError_t launch_fwd_one(TwPtrs tw, const void *p, cplx *t, size_t B, int dev_order, int gmax, hipStream_t s) {
    using P = Plan<LM, LOGR, NBT>;
    constexpr size_t LB = P::LDS_BYTES + (size_t)P::M * sizeof(cplx);
    const size_t groups = (B + NBT - 1) / NBT;
    const int grid = (int)(groups < (size_t)gmax ? groups : (size_t)gmax);
    hipError_t e = set_lds(transform_fwd_kernel<LM, WORD, NBT>, LB);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((transform_fwd_kernel<LM, WORD, NBT>), dim3(grid), dim3(P::NT), LB, s, tw, (const WORD *)p, t, B, dev_order);
    return hipGetLastError();
}

hipError_t launch_transform_fwd(int logM, int W, TwPtrs tw, const void *p, cplx *t, size_t B, int dev_order, hipStream_t s) {
    if (B == 0) return hipSuccess;
    // swept on MI355X (tools/fft_sweep.py): with nontemporal streams many short-lived workgroups (8-16 polynomials each
    // at a 4 GiB batch) beat few long-lived ones, as a flat copy beats a grid-stride one on this part (tools/membench2.hip)
    const LaunchTuning &lt = launch_tuning();
    // N = 2048 (round 4, tools/fft_w64_sweep.sh / fft_w32_sweep.sh): a cap of 2048 .. 16 384 workgroups is 3-5 % ahead of 32 768 on both rings
    // (64-bit: forward 0.619 -> 0.636, inverse 0.630 -> 0.649 of 8 TB/s); N = 1024 keeps the many short-lived workgroups (0.683 vs 0.585)
    const int gmax = lt.fft_grid > 0 ? lt.fft_grid : (logM >= 10 ? 4096 : 32768), nbt = lt.fft_nb;
    MKT_DISPATCH_LOGM(logM, {
        if (nbt == 2 && LM <= 10) {
            if (W == 64) return launch_fwd_one<LM, uint64_t, 2>(tw, p, t, B, dev_order, gmax, s);
            return launch_fwd_one<LM, uint32_t, 2>(tw, p, t, B, dev_order, gmax, s);
        }
        if (W == 64) return launch_fwd_one<LM, uint64_t, 1>(tw, p, t, B, dev_order, gmax, s);
        return launch_fwd_one<LM, uint32_t, 1>(tw, p, t, B, dev_order, gmax, s);
    });
    return hipSuccess;
}

hipError_t launch_reorder(int logM, const cplx *in, cplx *out, size_t npolys, int to_device, int order, hipStream_t s) {
    if (!npolys) return hipSuccess;
    MKT_DISPATCH_LOGM(logM, {
        hipLaunchKernelGGL((reorder_kernel<LM>), dim3(blocks_for(npolys << LM, 256)), dim3(256), 0, s, in, out, npolys, to_device, order);
    });
    return hipGetLastError();
}

hipError_t launch_transform_inv(int logM, int W, TwPtrs tw, const cplx *t, void *p, size_t B, hipStream_t s) {
    if (B == 0) return hipSuccess;
    const size_t gmax = launch_tuning().fft_igrid > 0 ? (size_t)launch_tuning().fft_igrid : (logM >= 10 ? 4096 : 32768);   // as the forward launch
    const int grid = (int)(B < gmax ? B : gmax);
    MKT_DISPATCH_LOGM(logM, {
        using P = Plan<LM, LOGR>;
        if (W == 64) {
            constexpr size_t LB = P::LDS_BYTES + (size_t)P::M * sizeof(cplx);
            hipError_t e = set_lds(transform_inv_kernel<LM, uint64_t>, LB); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((transform_inv_kernel<LM, uint64_t>), dim3(grid), dim3(P::NT), LB, s, tw, t, (uint64_t *)p, B);
        } else {
            constexpr size_t LB = P::LDS_BYTES + (size_t)P::M * sizeof(cplx);
            hipError_t e = set_lds(transform_inv_kernel<LM, uint32_t>, LB); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((transform_inv_kernel<LM, uint32_t>), dim3(grid), dim3(P::NT), LB, s, tw, t, (uint32_t *)p, B);
        }
    });
    return hipGetLastError();
}

hipError_t launch_decompose(int W, const void *p, void *digits, int N, int l, int logB, size_t B, hipStream_t s) {
    const size_t total = B * (size_t)N;
    if (!total) return hipSuccess;
    if (W == 64) hipLaunchKernelGGL(decompose_kernel<uint64_t>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (const uint64_t *)p, (uint64_t *)digits, N, l, logB, total);
    else hipLaunchKernelGGL(decompose_kernel<uint32_t>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (const uint32_t *)p, (uint32_t *)digits, N, l, logB, total);
    return hipGetLastError();
}

hipError_t launch_gate_linear(int op, const uint8_t *ops, const uint32_t *x, const uint32_t *y, const uint32_t *ix, const uint32_t *iy, size_t pool_rows, uint32_t *out, int len, size_t B, hipStream_t s) {
    const size_t total = B * (size_t)len;
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(gate_linear_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, op, ops, x, y, ix, iy, pool_rows, out, len, total);
    return hipGetLastError();
}

hipError_t launch_mux_linear(const uint32_t *pool, size_t pool_rows, const uint32_t *is, const uint32_t *ia, const uint32_t *ib, const uint8_t *fl, uint32_t *out, int len, size_t B, hipStream_t s) {
    if (!B || !pool_rows) return hipSuccess;
    hipLaunchKernelGGL(mux_linear_kernel, dim3(blocks_for(B * (size_t)len, 256)), dim3(256), 0, s, pool, pool_rows, is, ia, ib, fl, out, len, B);
    return hipGetLastError();
}

hipError_t launch_mux_combine(int W, void *acc, size_t B, size_t words, hipStream_t s) {
    if (!B) return hipSuccess;
    if (W == 64) hipLaunchKernelGGL(mux_combine_kernel<uint64_t>, dim3(blocks_for(B * words, 256)), dim3(256), 0, s, (uint64_t *)acc, B, words);
    else hipLaunchKernelGGL(mux_combine_kernel<uint32_t>, dim3(blocks_for(B * words, 256)), dim3(256), 0, s, (uint32_t *)acc, B, words);
    return hipGetLastError();
}

hipError_t launch_negate(uint32_t *x, size_t words, hipStream_t s) {
    if (!words) return hipSuccess;
    hipLaunchKernelGGL(negate_kernel, dim3(blocks_for(words, 256)), dim3(256), 0, s, x, words);
    return hipGetLastError();
}

hipError_t launch_modswitch(const uint32_t *lwe, uint32_t *atilde, uint32_t *btilde, int len, int logN, size_t B, hipStream_t s) {
    const size_t total = B * (size_t)len;
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(modswitch_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, lwe, atilde, btilde, len, logN, total);
    return hipGetLastError();
}

hipError_t launch_testvector(int W, const uint32_t *lin, int lwe_stride, int logN, int kacc, void *acc, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    if (W == 64) hipLaunchKernelGGL(testvector_kernel<uint64_t>, dim3((unsigned)B), dim3(256), 0, s, lin, lwe_stride, logN, kacc, (uint64_t *)acc);
    else hipLaunchKernelGGL(testvector_kernel<uint32_t>, dim3((unsigned)B), dim3(256), 0, s, lin, lwe_stride, logN, kacc, (uint32_t *)acc);
    return hipGetLastError();
}

#endif  // TU 0

#if MKT_IN_TU(1) || MKT_IN_TU(2) || MKT_IN_TU(3) || MKT_IN_TU(5)
template <int LM, typename WORD, int LB, int LR, int NB, int LT, int BT = 0>
static hipError_t launch_rot_lt(const RotArgs &a, size_t nrot, hipStream_t s) {
    using P = Plan<LM, LR, NB>;
    const size_t lds_bytes = P::LDS_BYTES + (size_t)P::M * sizeof(cplx);
    hipError_t e = set_lds(blindrotate_k1_kernel<LM, WORD, LB, LR, NB, LT, BT>, lds_bytes);
    if (e != hipSuccess) return e;
    last_rot_kernel = "blindrotate_k1_kernel";
    if (a.split == 0 || a.split >= nrot) {
        hipLaunchKernelGGL((blindrotate_k1_kernel<LM, WORD, LB, LR, NB, LT, BT>), dim3((unsigned)nrot), dim3(P::NT), lds_bytes, s, a);
        return hipGetLastError();
    }
    RotArgs b = a;
    for (size_t b0 = 0; b0 < nrot; b0 += a.split) {
        b.block0 = (unsigned)b0;
        const size_t g = nrot - b0 < a.split ? nrot - b0 : a.split;
        hipLaunchKernelGGL((blindrotate_k1_kernel<LM, WORD, LB, LR, NB, LT, BT>), dim3((unsigned)g), dim3(P::NT), lds_bytes, s, b);
    }
    return hipGetLastError();
}

template <int LM, typename WORD, int LB, int LR, int NB>
static hipError_t launch_rot_one(const RotArgs &a, size_t nrot, hipStream_t s) {
    if constexpr (LM < LR) { return hipErrorInvalidValue; } else {
        // gadget length known at compile time (fully unrolled digit loop): measured +8 % at M = 512, l = 2
        // (15.2 -> 14.0 ms, KMS k=2 N=1024)
        if constexpr (LB == 1 && LM == 9) {
            if (a.l == 2 && a.logB == 16 && sizeof(WORD) == 8) return launch_rot_lt<LM, WORD, LB, LR, NB, 2, 16>(a, nrot, s);
            if (a.l == 2) return launch_rot_lt<LM, WORD, LB, LR, NB, 2>(a, nrot, s);
        }
        // the shipped l = 3 shapes: CGGIparam / Blockparam (logB 9, 32-bit ring, N = 1024) and KMS2party / KMS2partyblock
        // (logB 12, 64-bit ring, N = 2048): +5..7 % once the gadget base is a constant too (the early l = 3 unrolling
        // without it had been slower)
        if constexpr ((LB == 1 || LB == 3) && LM == 9 && sizeof(WORD) == 4) {
            if (a.l == 3 && a.logB == 9) return launch_rot_lt<LM, WORD, LB, LR, NB, 3, 9>(a, nrot, s);
        }
        if constexpr ((LB == 1 || LB == 3) && LM == 10 && sizeof(WORD) == 8) {
            if (a.l == 3 && a.logB == 12) return launch_rot_lt<LM, WORD, LB, LR, NB, 3, 12>(a, nrot, s);
            if (a.l == 5 && a.logB == 8) return launch_rot_lt<LM, WORD, LB, LR, NB, 5, 8>(a, nrot, s);    // KMS4party, KMS16party (+7 %)
            if (a.l == 4 && a.logB == 9) return launch_rot_lt<LM, WORD, LB, LR, NB, 4, 9>(a, nrot, s);    // KMS8party
            if (a.l == 6 && a.logB == 7) return launch_rot_lt<LM, WORD, LB, LR, NB, 6, 7>(a, nrot, s);    // KMS32party
        }
        return launch_rot_lt<LM, WORD, LB, LR, NB, 0>(a, nrot, s);
    }
}

// variant = 10*LR + NB for the plain schemes (tuning knob).  Only LR == LOGR variants are built: the device
// point order of the key tables is tied to the points-per-thread of the schedule.
static inline int rot_variant(const RotArgs &a, int LM) {
    int variant = a.variant;
    if (variant == 0) variant = (LM <= 10 || a.blk_len > 1) ? 22 : 21;   // pairs of transforms up to M = 1024 (re-measured with the specialised kernels: +4 % at M = 1024) and for the block schemes; single transforms above (LDS)
    if ((2 * a.l) % (variant % 10) != 0) variant = (variant / 10) * 10 + 1;
    return variant;
}
template <int LM, typename WORD>
static hipError_t launch_rot_plain(const RotArgs &a, size_t nrot, hipStream_t s) {
    switch (rot_variant(a, LM)) {
    case 21: return launch_rot_one<LM, WORD, 1, LOGR, 1>(a, nrot, s);
    case 22: return launch_rot_one<LM, WORD, 1, LOGR, 2>(a, nrot, s);
    default: return hipErrorInvalidValue;
    }
}
template <int LM, typename WORD>
static hipError_t launch_rot_blk(const RotArgs &a, size_t nrot, hipStream_t s) {
    const int variant = rot_variant(a, LM);
    switch (a.blk_len) {
    case 2: return variant % 10 == 1 ? launch_rot_one<LM, WORD, 2, LOGR, 1>(a, nrot, s) : launch_rot_one<LM, WORD, 2, LOGR, 2>(a, nrot, s);
    case 3: return variant % 10 == 1 ? launch_rot_one<LM, WORD, 3, LOGR, 1>(a, nrot, s) : launch_rot_one<LM, WORD, 3, LOGR, 2>(a, nrot, s);
    case 4: return variant % 10 == 1 ? launch_rot_one<LM, WORD, 4, LOGR, 1>(a, nrot, s) : launch_rot_one<LM, WORD, 4, LOGR, 2>(a, nrot, s);
    default: return hipErrorInvalidValue;
    }
}
#endif  // TU 1-3, 5

#if MKT_IN_TU(1)
hipError_t launch_rot_plain_u32(int logM, const RotArgs &a, size_t nrot, hipStream_t s) {
    MKT_DISPATCH_LOGM(logM, { return launch_rot_plain<LM, uint32_t>(a, nrot, s); });
    return hipSuccess;
}
#endif
#if MKT_IN_TU(2)
hipError_t launch_rot_plain_u64(int logM, const RotArgs &a, size_t nrot, hipStream_t s) {
    MKT_DISPATCH_LOGM(logM, { return launch_rot_plain<LM, uint64_t>(a, nrot, s); });
    return hipSuccess;
}
#endif
#if MKT_IN_TU(3)
hipError_t launch_rot_block_u32(int logM, const RotArgs &a, size_t nrot, hipStream_t s) {
    MKT_DISPATCH_LOGM(logM, { return launch_rot_blk<LM, uint32_t>(a, nrot, s); });
    return hipSuccess;
}
#endif
#if MKT_IN_TU(5)
hipError_t launch_rot_block_u64(int logM, const RotArgs &a, size_t nrot, hipStream_t s) {
    MKT_DISPATCH_LOGM(logM, { return launch_rot_blk<LM, uint64_t>(a, nrot, s); });
    return hipSuccess;
}
#endif

#if MKT_IN_TU(0)
hipError_t launch_rot_plain_u32(int logM, const RotArgs &a, size_t nrot, hipStream_t s);
hipError_t launch_rot_plain_u64(int logM, const RotArgs &a, size_t nrot, hipStream_t s);
hipError_t launch_rot_block_u32(int logM, const RotArgs &a, size_t nrot, hipStream_t s);
hipError_t launch_rot_block_u64(int logM, const RotArgs &a, size_t nrot, hipStream_t s);
bool wide_supported(int logM, int l, int blk_len);
hipError_t launch_blindrotate_wide(int logM, int W, const RotArgs &a, size_t nrot, hipStream_t s);
hipError_t launch_blindrotate_k1(int logM, int W, const RotArgs &a, size_t nrot, hipStream_t s) {
    if (!nrot) return hipSuccess;
    // few rotations in flight (a single gate, small batches): the latency variant spreads one rotation over 2l thread
    // groups of one CU; a.wide: 0 = automatic (at most one rotation per CU, M <= 512: measured 3-16 % less latency there,
    // 2x MORE at M = 1024 where the groups need 8 points per thread), 1 = never, 2 = always where supported
    if (a.wide != 1 && wide_supported(logM, a.l, a.blk_len) && (a.wide == 2 || (nrot <= 256 && logM <= 9)))
        return launch_blindrotate_wide(logM, W, a, nrot, s);
    // A batch is run in rounds of one chip-fill (256 CUs x 4 two-wave workgroups at M = 512), and a round costs one
    // rotation's serial latency however few workgroups it holds.  A last round of at most one rotation per CU goes to the
    // latency variant instead (3.2 instead of 5.9 ms at KMS k = 2, N = 1024); same (ciphertext, slot) numbering via block0.
    constexpr size_t FILL = 1024;
    if (a.wide == 0 && a.blk_len == 1 && logM == 9 && a.split == 0 && nrot > FILL && nrot % FILL != 0 && nrot % FILL <= 256 && wide_supported(logM, a.l, a.blk_len)) {
        const size_t rem = nrot % FILL, head = nrot - rem;
        hipError_t e = W == 64 ? launch_rot_plain_u64(logM, a, head, s) : launch_rot_plain_u32(logM, a, head, s);
        if (e != hipSuccess) return e;
        RotArgs b = a;
        b.block0 = (unsigned)head;
        return launch_blindrotate_wide(logM, W, b, rem, s);
    }
    // ONE chip-fill and a remainder of 288..512 rotations run as two EQUAL launches (three workgroups per CU each) instead of four
    // per CU and then two: a round costs 3.4 / 5.0 / 5.35 / 6.3 ms at 1 / 2 / 3 / 4 workgroups per CU (KMS k = 2, N = 1024: two
    // workgroups of a CU land on the same SIMD pair), so 2 x 5.35 beats 6.3 + 5.0 -- measured 512 KMS gates 11.2 -> 10.1 ms, 470:
    // 11.1 -> 10.1, 448: equal, 427 (remainder 257): 9.6 -> 10.0, hence the lower bound; CGGIparam 1300 / 1536 gates 15.6 / 16.3 ->
    // 14.5 / 14.7 ms.  Behind two or more fills the single launch drains better (853 gates: 15.6 vs 16.0 ms) and keeps the batch.
    // Same (ciphertext, slot) numbering via block0; gates/s no longer dips below the 256-gate rate at 512 gates.
    if (a.blk_len == 1 && logM == 9 && a.split == 0 && nrot > FILL && nrot <= 2 * FILL && nrot % FILL >= 288 && nrot % FILL <= 512) {
        const size_t tail = FILL + nrot % FILL, head = nrot - tail, half = (tail + 1) / 2;
        RotArgs b = a;
        const size_t part[3] = {head, half, tail - half};
        size_t b0 = 0;
        for (int i = 0; i < 3; i++) {
            if (!part[i]) continue;
            b.block0 = a.block0 + (unsigned)b0;
            const hipError_t e = W == 64 ? launch_rot_plain_u64(logM, b, part[i], s) : launch_rot_plain_u32(logM, b, part[i], s);
            if (e != hipSuccess) return e;
            b0 += part[i];
        }
        return hipSuccess;
    }
    if (a.blk_len > 1) {
        // block schemes: G rotations of one slot per workgroup (rot_block.hip) once the batch fills the chip that way
        // automatic: four rotations per workgroup where that workgroup exists (M <= 512) and the batch still fills the
        // chip with one workgroup per compute unit (measured: Blockparam 5.54 -> 5.39 ms per 1024 gates; one rotation per
        // workgroup below that, and at M = 1024 where only two rotations fit and run 15 % slower)
        int G = a.blk_group;
        if (G == 0) G = (blockg_supported(logM, 4) && logM == 9 && nrot >= 1024) ? ((W == 32 && nrot >= 4096) ? 2 : 4) : 1;   // from 4096 rotations two rotations per workgroup, two workgroups per CU: 204 k against 196 k gates/s (Blockparam, 8192 and 16 384 gates)
        // 257..512 rotations: one 4-wave workgroup of two rotations per CU instead of two 2-wave workgroups, which share a SIMD pair
        // (tools/simd_place.hip): Blockparam 384 / 512 gates 4.51 -> 3.90 ms; up to 256 and from 513 to 1023 one rotation per workgroup is ahead
        if (a.blk_group == 0 && W == 32 && logM == 9 && nrot > 256 && nrot <= 512 && blockg_supported(logM, 2)) G = 2;
        // 64-bit ring at M = 1024 (KMS_block, params.jl:87-125): two rotations per workgroup halve the key elements a thread holds, which
        // leaves room to re-request each for the next digit right after its last use (rot_block.hip); ahead up to about one KMS2partyblock
        // batch of 1024 gates (512 / 1024 gates: 16.6 / 32.3 ms against 17.2 / 33.3), behind from 2048 (63.4 against 62.5 ms)
        if (a.blk_group == 0 && W == 64 && logM == 10 && nrot >= 1536 && nrot <= 4096 && blockg_supported(logM, 2)) G = 2;
        if (G > 1 && blockg_supported(logM, G) && a.blk_len >= 2 && a.blk_len <= 4) {
            const size_t nslots = (size_t)a.rows_per_gate;
            const hipError_t e = W == 64 ? launch_rot_blockg_u64(logM, G, 2, a, nslots, s) : launch_rot_blockg_u32(logM, G, 2, a, nslots, s);
            if (e != hipErrorInvalidValue) return e;     // a shape the grouped kernels do not cover (LDS budget): one rotation per workgroup below
        }
        return W == 64 ? launch_rot_block_u64(logM, a, nrot, s) : launch_rot_block_u32(logM, a, nrot, s);
    }
    return W == 64 ? launch_rot_plain_u64(logM, a, nrot, s) : launch_rot_plain_u32(logM, a, nrot, s);
}
#endif  // TU 0

#if MKT_IN_TU(7)
#ifndef MKT_KR_BLKG_MIN
#define MKT_KR_BLKG_MIN 1     // rotations from which the grouped kernel is the default: ahead at every batch size (128 gates: 9.1 vs 13.9 ms, tools/blkg_ab.sh)
#endif
template <int LM, typename WORD, int KR, bool BLK, int BL = 0>
static hipError_t launch_kr_one(const RotArgs &a, size_t nrot, hipStream_t s) {
    using P = Plan<LM, LOGR>;
    constexpr size_t LB = P::LDS_BYTES + (size_t)P::M * sizeof(cplx);
    hipError_t e = set_lds(blindrotate_kr_kernel<LM, WORD, KR, BLK, BL>, LB);
    if (e != hipSuccess) return e;
    last_rot_kernel = "blindrotate_kr_kernel";
    hipLaunchKernelGGL((blindrotate_kr_kernel<LM, WORD, KR, BLK, BL>), dim3((unsigned)nrot), dim3(P::NT), LB, s, a);
    return hipGetLastError();
}
template <int LM, typename WORD>
static hipError_t launch_kr_word(int kr, const RotArgs &a, size_t nrot, hipStream_t s) {
    if (a.blk_len == 3 && kr == 2 && sizeof(WORD) == 4) return launch_kr_one<LM, WORD, 2, true, 3>(a, nrot, s);   // Blockparam's block length (params.jl:8-13)
    if (a.blk_len > 1) return kr == 2 ? launch_kr_one<LM, WORD, 2, true>(a, nrot, s) : launch_kr_one<LM, WORD, 3, true>(a, nrot, s);
    return kr == 2 ? launch_kr_one<LM, WORD, 2, false>(a, nrot, s) : launch_kr_one<LM, WORD, 3, false>(a, nrot, s);
}

hipError_t launch_blindrotate_kr(int logM, int W, int kr, const RotArgs &a, size_t nrot, hipStream_t s) {
    if (!nrot) return hipSuccess;
    if (kr < 2 || kr > 3) return hipErrorInvalidValue;
    // 32-bit ring, LMSS with RLWE length 2 and block length 3, CGGI with RLWE length 2 or 3: four rotations per workgroup share every
    // key element (rot_block.hip); blk_group (MKT_ROT_BLKG): 0 = by batch size, 1 = never, 4 = always
    if (W == 32 && a.ngates == nrot && ((kr == 2 && (a.blk_len == 3 || a.blk_len == 1)) || (kr == 3 && a.blk_len == 1))) {
        int G = a.blk_group;
        // by default where it measured ahead (tools/blkg_ab.sh, tools/kr_time.py): block length 3 at RLWE length 2 (14.1 -> 10.0 ms per 1024
        // gates) and the plain CMux at RLWE length 3 (38.8 -> 24.3 ms); the plain CMux at RLWE length 2 stays on the one-rotation kernel
        // (17.0 vs 17.9 ms at 1024 gates, 13.0 vs 17.8 at 256)
        if (G == 0) G = (blockg_supported(logM, 4) && nrot >= MKT_KR_BLKG_MIN && !(kr == 2 && a.blk_len == 1)) ? 4 : 1;
        if (G == 4 && blockg_supported(logM, 4)) {
            const hipError_t e = launch_rot_blockg_u32(logM, 4, kr + 1, a, (size_t)a.rows_per_gate, s);
            if (e != hipErrorInvalidValue) return e;
        }
    }
    MKT_DISPATCH_LOGM(logM, {
        if (W == 64) return launch_kr_word<LM, uint64_t>(kr, a, nrot, s);
        return launch_kr_word<LM, uint32_t>(kr, a, nrot, s);
    });
    return hipSuccess;
}
#endif  // TU 7

#if MKT_IN_TU(4)
hipError_t launch_kms_phase2(int logM, int W, const Phase2Args &a, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    MKT_DISPATCH_LOGM(logM, {
        using P = Plan<LM, LOGR>;
        if (W == 64) {
            hipError_t e = set_lds(kms_phase2_kernel<LM, uint64_t>, P::LDS_BYTES); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((kms_phase2_kernel<LM, uint64_t>), dim3((unsigned)B), dim3(P::NT), P::LDS_BYTES, s, a);
        } else {
            hipError_t e = set_lds(kms_phase2_kernel<LM, uint32_t>, P::LDS_BYTES); if (e != hipSuccess) return e;
            hipLaunchKernelGGL((kms_phase2_kernel<LM, uint32_t>), dim3((unsigned)B), dim3(P::NT), P::LDS_BYTES, s, a);
        }
    });
    return hipGetLastError();
}

template <int LM, typename WORD, int LT, int BT>
static hipError_t launch_ccs_one(const CcsArgs &a, size_t B, hipStream_t s) {
    using P = Plan<LM, LOGR, 1>;
    constexpr size_t LB = P::LDS_BYTES + (size_t)P::M * sizeof(cplx);
    hipError_t e = set_lds(ccs_blindrotate_kernel<LM, WORD, LT, BT>, LB); if (e != hipSuccess) return e;
    last_rot_kernel = "ccs_blindrotate_kernel";
    hipLaunchKernelGGL((ccs_blindrotate_kernel<LM, WORD, LT, BT>), dim3((unsigned)B), dim3(P::NT), LB, s, a);
    return hipGetLastError();
}
hipError_t launch_ccs_blindrotate(int logM, int W, const CcsArgs &a, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    MKT_DISPATCH_LOGM(logM, {
        if (W == 64) return launch_ccs_one<LM, uint64_t, 0, 0>(a, B, s);
        // the shipped gadgets of the 32-bit CCS sets (params.jl:15-45): CCS2party (3, 8), CCS4party (4, 8), CCS8party (5, 6)
        if constexpr (LM == 9 || LM == 10) {
            if (a.l == 3 && a.logB == 8) return launch_ccs_one<LM, uint32_t, 3, 8>(a, B, s);
            if (a.l == 4 && a.logB == 8) return launch_ccs_one<LM, uint32_t, 4, 8>(a, B, s);
            if (a.l == 5 && a.logB == 6) return launch_ccs_one<LM, uint32_t, 5, 6>(a, B, s);
        }
        return launch_ccs_one<LM, uint32_t, 0, 0>(a, B, s);
    });
    return hipGetLastError();
}

#endif  // TU 4

#if MKT_IN_TU(0)
namespace {
struct KsPlan { int G, waves, ngroups, parties, gblocks, slabs, jslab; bool pair; };
// launch shape of a key switch; pair: digit pairs (keyswitch_pair_kernel: D = 4 and an even digit count, 32 ciphertexts per wave,
// scratch present), else the per-digit kernel with atomics
KsPlan ks_plan(const KsArgs &a, size_t B, bool have_scratch) {
    KsPlan q{};
    q.G = 32;
    int target_blocks = 1024;   // swept on MI355X (tools/ks_sweep.sh): 1.6 ms vs 4.7 ms single-gate at KMS k=2 N=1024
    // per-digit kernel, balanced digits (block schemes): sign handling on the scalar unit wants four to eight times the workgroups once the
    // batch leaves few slabs (and one wave per staged table, below): Blockparam 4096 gates 2.12 -> 1.32 ms, 16 384 gates 8.49 -> 4.94 ms,
    // KMS2partyblock 1024 gates 2.73 -> 2.35 ms; the unbalanced variant is fastest at 1024 at every batch size (tools/ks_blocks_sweep*.sh)
    if (a.balanced) target_blocks = a.mk ? 4096 : 8192;
    const LaunchTuning &lt = launch_tuning();
    if (lt.ks_g > 0) q.G = lt.ks_g;
    if (lt.ks_blocks > 0) target_blocks = lt.ks_blocks;
    q.pair = lt.ks_pair != 0 && a.logD == 2 && a.f % 2 == 0 && q.G == 32 && have_scratch;
    // (measured and left out: eight waves of 16 ciphertexts around one table -- half the slabs; KMS k=2 0.94 vs 0.79 ms, CGGIparam 0.51 vs 0.52)
    if (q.pair && lt.ks_blocks <= 0) target_blocks = 1024;   // tools/ks_pair_sweep.sh, every kind of set: one round of three 4-wave workgroups per CU (Blockparam 1024 / 16 384 gates 0.12 / 1.35 ms; 0.19 / 1.44 at 2048)
    q.ngroups = (int)((B + q.G - 1) / q.G);
    q.parties = a.mk ? a.kacc : 1;
    // enough workgroups to fill the chip, slabs of at least 8 coefficients
    auto nslabs = [&] {
        int sl = (target_blocks + q.ngroups * q.parties - 1) / (q.ngroups * q.parties);
        if (sl < 1) sl = 1;
        if (sl > a.N / 8) sl = a.N / 8;
        return sl < 1 ? 1 : sl;
    };
    q.slabs = nslabs();
    q.jslab = (a.N + q.slabs - 1) / q.slabs;
    q.slabs = (a.N + q.jslab - 1) / q.jslab;
    q.waves = lt.ks_waves > 0 ? lt.ks_waves : 4;
    if (q.waves != 2 && q.waves != 4) q.waves = 1;
    if (q.G != 32) q.waves = 1;
    if (a.balanced && lt.ks_waves <= 0 && !q.pair) q.waves = 1;   // per-digit kernel, balanced digits: every wave stages its own table (KMS2partyblock 2.7 ms alone vs 4.0 ms shared; Blockparam 16 384 gates 4.98 -> 4.11 ms, RLWE length 2 19.3 -> 14.9 ms)
    q.gblocks = (q.ngroups + q.waves - 1) / q.waves;
    return q;
}
}  // namespace

void ks_scratch_words(const KsArgs &a, size_t B, size_t *digit_words, size_t *partial_words) {
    const KsPlan q = ks_plan(a, B, true);
    *digit_words = q.pair ? (size_t)a.kacc * q.ngroups * (size_t)a.N * 32 : 0;                    // [kacc][ceil(B / 32)][N][32]
    *partial_words = q.pair ? (size_t)q.slabs * q.parties * B * (size_t)a.n1p : 0;                 // [slab][party][B][n1p]
}

hipError_t launch_keyswitch(int W, const KsArgs &a, size_t B, hipStream_t s) {
    if (!B) return hipSuccess;
    const KsPlan q = ks_plan(a, B, a.digits && a.partial);
    const int G = q.G, waves = q.waves, ngroups = q.ngroups, parties = q.parties, gblocks = q.gblocks, slabs = q.slabs, jslab = q.jslab;
    const bool pair = q.pair;
    const dim3 grid((unsigned)(gblocks * slabs), (unsigned)parties, (unsigned)((a.n1p + KS_CHUNK_WORDS - 1) / KS_CHUNK_WORDS));
    const size_t ks_lds = pair ? (size_t)KS_STAGES * 16 * KS_LANES * sizeof(uint4) : (size_t)KS_STAGES * (1 + a.drows * (a.balanced ? 2 : 1)) * KS_LANES * sizeof(uint4);
    if (ks_lds > 64 * 1024) return hipErrorInvalidValue;   // logD <= 5
    const size_t total = B * (size_t)(parties * a.n + 1);
    if (pair) {      // digit words -> partial sums per slab -> output
#define MKT_KSP_LAUNCH_B(WT, BV) do { if (waves == 4) hipLaunchKernelGGL((keyswitch_pair_kernel<WT, 32, 4, BV>), grid, dim3(KS_LANES * 4), ks_lds, s, a, (int)B, ngroups); \
        else if (waves == 2) hipLaunchKernelGGL((keyswitch_pair_kernel<WT, 32, 2, BV>), grid, dim3(KS_LANES * 2), ks_lds, s, a, (int)B, ngroups); \
        else hipLaunchKernelGGL((keyswitch_pair_kernel<WT, 32, 1, BV>), grid, dim3(KS_LANES), ks_lds, s, a, (int)B, ngroups); } while (0)
        const dim3 dgrid((unsigned)((a.N + 7) / 8), (unsigned)ngroups, (unsigned)a.kacc);
#define MKT_KSD_LAUNCH(WT) do { if (a.balanced) hipLaunchKernelGGL((ks_digits_kernel<WT, true>), dgrid, dim3(256), 0, s, a, (int)B, ngroups); \
        else hipLaunchKernelGGL((ks_digits_kernel<WT, false>), dgrid, dim3(256), 0, s, a, (int)B, ngroups); } while (0)
        if (W == 64) {
            MKT_KSD_LAUNCH(uint64_t);
            if (a.balanced) MKT_KSP_LAUNCH_B(uint64_t, true); else MKT_KSP_LAUNCH_B(uint64_t, false);
            hipLaunchKernelGGL(ks_reduce_kernel<uint64_t>, dim3(blocks_for(total, 256)), dim3(256), 0, s, a, B, slabs, parties);
        } else {
            MKT_KSD_LAUNCH(uint32_t);
            if (a.balanced) MKT_KSP_LAUNCH_B(uint32_t, true); else MKT_KSP_LAUNCH_B(uint32_t, false);
            hipLaunchKernelGGL(ks_reduce_kernel<uint32_t>, dim3(blocks_for(total, 256)), dim3(256), 0, s, a, B, slabs, parties);
        }
#undef MKT_KSD_LAUNCH
#undef MKT_KSP_LAUNCH_B
        return hipGetLastError();
    }
#define MKT_KS_LAUNCH_B(WT, GV, BV) do { if (GV == 32 && waves == 4) hipLaunchKernelGGL((keyswitch_mg_kernel<WT, 32, 4, BV>), grid, dim3(KS_LANES * 4), ks_lds, s, a, (int)B, ngroups, jslab); \
        else if (GV == 32 && waves == 2) hipLaunchKernelGGL((keyswitch_mg_kernel<WT, 32, 2, BV>), grid, dim3(KS_LANES * 2), ks_lds, s, a, (int)B, ngroups, jslab); \
        else hipLaunchKernelGGL((keyswitch_mg_kernel<WT, GV, 1, BV>), grid, dim3(KS_LANES), ks_lds, s, a, (int)B, ngroups, jslab); } while (0)
#define MKT_KS_LAUNCH(WT, GV) do { if (a.balanced) MKT_KS_LAUNCH_B(WT, GV, true); else MKT_KS_LAUNCH_B(WT, GV, false); } while (0)
    if (W == 64) {
        hipLaunchKernelGGL(ks_init_kernel<uint64_t>, dim3(blocks_for(total, 256)), dim3(256), 0, s, a, B);
        if (G == 8) MKT_KS_LAUNCH(uint64_t, 8); else if (G == 32) MKT_KS_LAUNCH(uint64_t, 32); else MKT_KS_LAUNCH(uint64_t, 16);
    } else {
        hipLaunchKernelGGL(ks_init_kernel<uint32_t>, dim3(blocks_for(total, 256)), dim3(256), 0, s, a, B);
        if (G == 8) MKT_KS_LAUNCH(uint32_t, 8); else if (G == 32) MKT_KS_LAUNCH(uint32_t, 32); else MKT_KS_LAUNCH(uint32_t, 16);
    }
#undef MKT_KS_LAUNCH_B
#undef MKT_KS_LAUNCH
    return hipGetLastError();
}
#endif  // TU 0

#if MKT_IN_TU(6)
// ------------------------------------------------------------------------------------------------
// Blind rotation, latency variant (few rotations in flight: a single gate, small batches).  Same arithmetic as
// blindrotate_k1_kernel, LB = 1: bootstrapping.jl:32-76 (CGGI, k = 1), :389-443 (KMS phase 1).  ONE workgroup per
// rotation, but the 2l digit transforms of a CMux step (:54-59) run side by side on 2l thread groups of NT threads:
// group g decomposes its polynomial of the accumulator (b for g < l, a otherwise -- every group keeps its own copy of
// that polynomial in registers), transforms digit g, multiplies by key row g (:63-68).  The 2l products meet in LDS and
// are summed IN THE REFERENCE'S ORDER (g = 0 .. 2l-1, starting from 0), so every rounding is the reference's.  The group
// of digit 0 of each polynomial then multiplies by the monomial, runs the inverse transform and updates its copy of the
// polynomial (:71-73); the other l-1 groups of that polynomial wait at the transform's barriers and copy the new words
// from LDS.
// LDS: Psi | region[2l][M]: FFT staging of group g (one buffer: barriers on both sides of an exchange), reused for the
// product exchange between the forward and the inverse transforms.  Keys are read in the resident (LOGR = 2) device
// point order whatever LR this kernel uses.
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int dev_pos_lr2(int order, int x, int M) {      // dev_pos of the LOGR = 2 schedules, whatever LR this kernel runs
    const int NT2 = M >> 2;
    if (order == 1) return (x & 3) * NT2 + (x >> 2);
    return ((x & 3) >> 1) * (2 * NT2) + ((x >> 2) << 1) + (x & 1);
}

template <int LOGM, typename WORD, int LR, int LT>
__global__ __launch_bounds__((2 * LT * Plan<LOGM, LR>::NT)) void blindrotate_wide_kernel(const RotArgs a) {
    using P = Plan<LOGM, LR, 1>;
    constexpr int R = P::R, NT = P::NT, M = P::M, N = 2 * M, W = WordTraits<WORD>::W, G = 2 * LT;
    constexpr int MO = 0x1ff;                 // library-default exchange routes, single staging buffer
    static_assert(MKT_LOGR == 2, "the resident tables are in the LOGR = 2 device point order");
    cplx *psi_l = reinterpret_cast<cplx *>(mkt_smem);
    cplx *region = psi_l + M;
    const int tid = threadIdx.x, grp = tid / NT, t = tid % NT;
    const int c = grp >= LT ? 1 : 0, j = c ? grp - LT : grp;
    XS xs = make_xs();
    for (int i = tid; i < M; i += G * NT) psi_l[i] = a.tw.psi[i];
    __syncthreads();
    const unsigned bid = blockIdx.x + a.block0;
    size_t gate; int slot;
    rot_decode(a, bid, gate, slot);
    const size_t rot = gate * (size_t)a.rows_per_gate + slot;
    const int party = __builtin_amdgcn_readfirstlane(a.slot_party[slot]), row = __builtin_amdgcn_readfirstlane(a.slot_row[slot]);
    const uint32_t *at_src = a.lwe + gate * (size_t)a.lwe_stride + (size_t)party * a.n;
    const cplx *brk = a.brk + (size_t)party * a.brk_party_stride;
    const Gadget<WORD> gd(LT, a.logB);
    cplx *stage = region + (size_t)grp * M;

    cplx rt[R], ri[R];
    int kp[R];                                // resident-table position of the point slot e holds after a forward transform
#pragma unroll
    for (int e = 0; e < R; e++) { rt[e] = a.tw.roots[e * NT + t]; ri[e] = a.tw.rootsinv[e * NT + t]; kp[e] = dev_pos_lr2(MKT_DEVORDER, t * R + e, M); }

    WORD acc[R][2];                           // this group's polynomial (c) of the accumulator
    if (a.init_mode == 0) {
        const WORD *src = reinterpret_cast<const WORD *>(a.acc_io) + rot * 2 * N + (size_t)c * N;
#pragma unroll
        for (int e = 0; e < R; e++) { acc[e][0] = src[e * NT + t]; acc[e][1] = src[M + e * NT + t]; }
    } else {                                  // bootstrapping.jl:403-406
#pragma unroll
        for (int e = 0; e < R; e++) { acc[e][0] = 0; acc[e][1] = 0; }
        if (c == 0 && t == 0) acc[0][0] = (WORD)1 << (W - (row + 1) * a.logB_lev);
    }

    const int msbit = 32 - a.logN - 1;
    for (int blk = 0; blk < a.n; blk++) {
        const uint32_t v0 = at_src[blk];
        const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.pre_switched ? v0 : divbits<uint32_t>(v0, msbit)));
        if (at == 0) continue;                                           // :48 / :413 (uniform over the workgroup)
        cplx z[1][R];
#pragma unroll
        for (int e = 0; e < R; e++) {                                    // :50-51 decompto!, fft.jl:57-63
            const int d0 = gd.digit(gd.prep(acc[e][0]), j), d1 = gd.digit(gd.prep(acc[e][1]), j);
            cplx v; v.re = (double)d0; v.im = (double)(-d1);
            z[0][e] = cmul(v, rt[e]);
        }
        fft_forward<LOGM, LR, 1, MO>(z, psi_l, stage, t, xs.lx);         // :54-59
        const cplx *krow = brk + (((size_t)blk * G + (size_t)grp) * 2) * M;
        cplx pb[R], pa[R], ts[R];
#pragma unroll
        for (int e = 0; e < R; e++) { pb[e] = cmul(z[0][e], krow[kp[e]]); pa[e] = cmul(z[0][e], krow[M + kp[e]]); }
        const cplx *mono = a.monomial + (size_t)(at - 1) * M;
        cplx mv[R];
#pragma unroll
        for (int e = 0; e < R; e++) mv[e] = mono[kp[e]];
#pragma unroll
        for (int ph = 0; ph < 2; ph++) {                                 // :63-68: sum over the rows, reference order
            __syncthreads();
#pragma unroll
            for (int e = 0; e < R; e++) stage[e * NT + t] = ph ? pa[e] : pb[e];
            __syncthreads();
            if (c == ph && j == 0) {
#pragma unroll
                for (int e = 0; e < R; e++) { ts[e].re = 0.0; ts[e].im = 0.0; }
#pragma unroll
                for (int g = 0; g < G; g++)
#pragma unroll
                    for (int e = 0; e < R; e++) ts[e] = cadd(ts[e], region[(size_t)g * M + e * NT + t]);
            }
        }
        // :71-73 once per polynomial, by the group of its digit 0; the other groups of that polynomial only keep the
        // workgroup barriers company and then pick the new words up from LDS (exact integers: every copy stays identical)
        WORD *xw = reinterpret_cast<WORD *>(region) + (size_t)c * N;      // [2][N] words at the head of the region
        if (j == 0) {
            cplx s[1][R];
#pragma unroll
            for (int e = 0; e < R; e++) s[0][e] = cmul(mv[e], ts[e]);    // :71
            fft_inverse<LOGM, LR, 1, true, MO>(s, psi_l, stage, t, xs.lx);   // :72 (its first barrier also fences the product reads)
#pragma unroll
            for (int e = 0; e < R; e++) {                                // fft.jl:76-80, :73
                const cplx v = cmul(s[0][e], ri[e]);
                acc[e][0] = (WORD)(acc[e][0] + native<WORD>(v.re));
                acc[e][1] = (WORD)(acc[e][1] + native<WORD>(-v.im));
            }
        } else {
            fft_inverse_barriers_only<LOGM, LR, 1, MO>();
        }
        if (LT > 1) {
            __syncthreads();                                             // the last staging reads of the inverse are done
            if (j == 0) {
#pragma unroll
                for (int e = 0; e < R; e++) { xw[e * NT + t] = acc[e][0]; xw[M + e * NT + t] = acc[e][1]; }
            }
            __syncthreads();
            if (j != 0) {
#pragma unroll
                for (int e = 0; e < R; e++) { acc[e][0] = xw[e * NT + t]; acc[e][1] = xw[M + e * NT + t]; }
            }
        }
    }

    if (a.out_mode == 0) {
        if (j == 0) {
            WORD *dst = reinterpret_cast<WORD *>(a.acc_io) + rot * 2 * N + (size_t)c * N;
#pragma unroll
            for (int e = 0; e < R; e++) { dst[e * NT + t] = acc[e][0]; dst[M + e * NT + t] = acc[e][1]; }
        }
    } else {                                                             // :441 fftto!(tacc, acc); every group runs it (barriers), one per polynomial stores
        cplx z[1][R];
#pragma unroll
        for (int e = 0; e < R; e++) {
            cplx v; v.re = word_to_f64<WORD>(acc[e][0]); v.im = word_to_f64<WORD>((WORD)((WORD)0 - acc[e][1]));
            z[0][e] = cmul(v, rt[e]);
        }
        fft_forward<LOGM, LR, 1, MO>(z, psi_l, stage, t, xs.lx);
        if (j == 0) {
            cplx *o = a.tout + (rot * 2 + c) * M;
#pragma unroll
            for (int e = 0; e < R; e++) o[a.tout_natural ? t * R + e : kp[e]] = z[0][e];
        }
    }
}

template <int LM, typename WORD, int LR, int LT>
static hipError_t launch_wide_one(const RotArgs &a, size_t nrot, hipStream_t s) {
    using P = Plan<LM, LR, 1>;
    constexpr int threads = 2 * LT * P::NT;
    static_assert(threads <= 1024, "workgroup too large");
    const size_t lds_bytes = (size_t)(1 + 2 * LT) * P::M * sizeof(cplx);
    hipError_t e = set_lds(blindrotate_wide_kernel<LM, WORD, LR, LT>, lds_bytes);
    if (e != hipSuccess) return e;
    last_rot_kernel = "blindrotate_wide_kernel";
    hipLaunchKernelGGL((blindrotate_wide_kernel<LM, WORD, LR, LT>), dim3((unsigned)nrot), dim3(threads), lds_bytes, s, a);
    return hipGetLastError();
}
template <int LM, typename WORD>
static hipError_t launch_wide_lm(const RotArgs &a, size_t nrot, hipStream_t s) {
    // points per thread chosen so that 2l groups fit one workgroup: 4 up to M = 512, 8 at M = 1024
    constexpr int LR = LM >= 10 ? 3 : 2;
    switch (a.l) {
    case 2: return launch_wide_one<LM, WORD, LR, 2>(a, nrot, s);
    case 3: return launch_wide_one<LM, WORD, LR, 3>(a, nrot, s);
    case 4: return launch_wide_one<LM, WORD, LR, 4>(a, nrot, s);
    default: return hipErrorInvalidValue;
    }
}
bool wide_supported(int logM, int l, int blk_len) { return blk_len == 1 && logM >= 8 && logM <= 10 && l >= 2 && l <= 4; }
hipError_t launch_blindrotate_wide(int logM, int W, const RotArgs &a, size_t nrot, hipStream_t s) {
    if (!nrot) return hipSuccess;
    switch (logM) {
    case 8:  return W == 64 ? launch_wide_lm<8, uint64_t>(a, nrot, s) : launch_wide_lm<8, uint32_t>(a, nrot, s);
    case 9:  return W == 64 ? launch_wide_lm<9, uint64_t>(a, nrot, s) : launch_wide_lm<9, uint32_t>(a, nrot, s);
    case 10: return W == 64 ? launch_wide_lm<10, uint64_t>(a, nrot, s) : launch_wide_lm<10, uint32_t>(a, nrot, s);
    default: return hipErrorInvalidValue;
    }
}
#endif  // TU 6

}  // namespace mktd
