// Pieces shared by the kernel translation units (kernels.hip, rot_block.hip): points per thread, the dynamic LDS symbol,
// exchange state, digit -> transform helpers, buffer-descriptor table loads, launch helpers.  Internal to the library.
#pragma once
#include "device_api.h"
#include "fft_device.h"

#include <cstdlib>

#pragma clang fp contract(off)

namespace mktd {

constexpr int LOGR = MKT_LOGR;  // points per thread (4 by default)

extern __shared__ __attribute__((aligned(16))) unsigned char mkt_smem[];

// per-thread exchange state: lane facts for the in-wave exchanges
struct XS { LaneX lx; };
__device__ __forceinline__ XS make_xs() { XS x; x.lx = make_lanex(); return x; }

template <int LOGM>
__device__ __forceinline__ void fft_forward1(cplx (&z)[1 << LOGR], const cplx *__restrict__ psi, cplx *lds, int t, XS &xs) {
    fft_forward<LOGM, LOGR, 1>(reinterpret_cast<cplx(&)[1][1 << LOGR]>(z), psi, lds, t, xs.lx);
}
template <int LOGM>
__device__ __forceinline__ void fft_inverse1(cplx (&z)[1 << LOGR], const cplx *__restrict__ psiinv, cplx *lds, int t, XS &xs) {
    fft_inverse<LOGM, LOGR, 1>(reinterpret_cast<cplx(&)[1][1 << LOGR]>(z), psiinv, lds, t, xs.lx);
}

// ------------------------------------------------------------------------------------------------
// digit -> transform helper: z[e] = (d(c_idx) - i*d(c_{idx+M})) * roots[idx]   (fft.jl:57-63)
// ------------------------------------------------------------------------------------------------
template <typename WORD, int R>
__device__ __forceinline__ void digit_points(cplx (&z)[R], const WORD (&tp)[R][2], const Gadget<WORD> &gd, int j, const cplx (&rt)[R]) {
#pragma unroll
    for (int e = 0; e < R; e++) {
        const int d0 = gd.digit(tp[e][0], j), d1 = gd.digit(tp[e][1], j);
        cplx v; v.re = (double)d0; v.im = (double)(-d1);
        z[e] = cmul(v, rt[e]);
    }
}

// inverse transform of a transform-domain accumulator followed by native() (fft.jl:74-81)
template <int LOGM, typename WORD>
__device__ __forceinline__ void inverse_to_words(cplx (&z)[1 << LOGR], WORD (&w)[1 << LOGR][2], const TwPtrs &tw, cplx *lds, int t, XS &xs) {
    using P = Plan<LOGM, LOGR>;
    fft_inverse1<LOGM>(z, tw.psiinv, lds, t, xs);
#pragma unroll
    for (int e = 0; e < P::R; e++) {
        const cplx v = cmul(z[e], tw.rootsinv[e * P::NT + t]);
        w[e][0] = native<WORD>(v.re);
        w[e][1] = native<WORD>(-v.im);
    }
}

// Workgroup index -> (ciphertext, rotation slot) of a rotation launch.
// map_mode 0: slot-major -- all ciphertexts' rotations of one slot are adjacent, so the workgroups resident at any time stream
//   the SAME party's key rows through L2.
// map_mode 1: party-major, and inside a party's region the R rows of one (ciphertext, party) sit EIGHT workgroup ids apart
//   (chunks of 8 ciphertexts x R rows: id = chunk * 8R + row * 8 + ciphertext % 8).  Workgroup ids are dealt round-robin over
//   the 8 XCDs, so the rows of one ciphertext -- which share the mask words, hence the monomial row and the key rows of every
//   step (bootstrapping.jl:400-406) -- run on the same XCD at the same time and meet in its L2.
__device__ __forceinline__ void rot_decode(const RotArgs &a, unsigned bid, size_t &gate, int &slot) {
    if (a.map_mode == 0) { gate = bid % (size_t)a.ngates; slot = (int)(bid / (size_t)a.ngates); return; }
    size_t base = 0;
    int s0 = 0;
    for (;;) {
        const int pty = a.slot_party[s0];
        int R = 1;
        while (s0 + R < a.rows_per_gate && a.slot_party[s0 + R] == pty) R++;
        const size_t sz = (size_t)a.ngates * R;
        if (bid < base + sz || s0 + R >= a.rows_per_gate) {
            const size_t idx = bid - base, chunk = idx / (8 * (size_t)R), j = idx % (8 * (size_t)R);
            const size_t left = a.ngates - 8 * chunk, g8 = left < 8 ? left : 8;
            gate = 8 * chunk + j % g8; slot = s0 + (int)(j / g8);
            return;
        }
        base += sz; s0 += R;
    }
}

// Key rows, monomial rows and the twist tables are read through buffer descriptors: SGPR base + 32-bit per-lane
// offset + SGPR row offset, so a load costs no address arithmetic on the VALU (flat loads needed a 64-bit add each:
// ~100 of the ~2000 VALU instructions of a CMux).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t table_rsrc(const void *p, size_t bytes) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0,
                                             (int)(bytes > 0x7fffffffull ? 0x7fffffffull : bytes), 0x00020000);
}
__device__ __forceinline__ cplx table_load(__amdgpu_buffer_rsrc_t rs, unsigned voff_bytes, unsigned soff_bytes) {
    auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff_bytes, (int)soff_bytes, 0);
    cplx r; __builtin_memcpy(&r, &v, 16); return r;
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
static inline int blocks_for(size_t total, int threads) {
    size_t b = (total + threads - 1) / threads;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

template <typename K>
static hipError_t set_lds(K kern, size_t bytes) {
    if (bytes > 48 * 1024) return hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return hipSuccess;
}

#ifdef MKT_ONLY_LOGM   // development builds: instantiate one transform size only (seconds instead of a minute per unit)
#define MKT_DISPATCH_LOGM(logM, ...)                 \
    switch (logM) {                                  \
    case MKT_ONLY_LOGM: { constexpr int LM = MKT_ONLY_LOGM; __VA_ARGS__; } break; \
    default: return hipErrorInvalidValue;            \
    }
#else
#define MKT_DISPATCH_LOGM(logM, ...)                 \
    switch (logM) {                                  \
    case 4:  { constexpr int LM = 4;  __VA_ARGS__; } break; \
    case 5:  { constexpr int LM = 5;  __VA_ARGS__; } break; \
    case 6:  { constexpr int LM = 6;  __VA_ARGS__; } break; \
    case 7:  { constexpr int LM = 7;  __VA_ARGS__; } break; \
    case 8:  { constexpr int LM = 8;  __VA_ARGS__; } break; \
    case 9:  { constexpr int LM = 9;  __VA_ARGS__; } break; \
    case 10: { constexpr int LM = 10; __VA_ARGS__; } break; \
    case 11: { constexpr int LM = 11; __VA_ARGS__; } break; \
    default: return hipErrorInvalidValue;            \
    }
#endif

}  // namespace mktd
