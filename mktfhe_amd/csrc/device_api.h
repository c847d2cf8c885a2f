// Host-callable launchers of the HIP kernels (kernels.hip).  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mktd {

struct cplx;

struct TwPtrs { const cplx *psi, *psiinv, *roots, *rootsinv; };

// Launcher-level A/B switches (grid shapes of the transform and key-switch kernels): process-wide, read from the
// environment on FIRST USE only -- the call path never calls getenv (context.cpp).  0 = the built-in default.
struct LaunchTuning { int fft_grid, fft_nb, fft_igrid, ks_g, ks_blocks, ks_waves, ks_pair, ntt_grid; };
const LaunchTuning &launch_tuning();
// base name of the blind-rotation kernel the calling thread launched last (set by every rotation launcher; read by
// mkt_last_kernel_name so that bench.py's roofline names the kernel that actually ran)
extern thread_local const char *last_rot_kernel;

// Blind rotation with an RLWE accumulator of length 1 (b, a): CGGI/LMSS with k = 1 and every row of
// KMS / KMS_block phase 1.  One workgroup per rotation.
struct RotArgs {
    TwPtrs tw;
    const cplx *brk;          // party 0 base; [n][2l rows][2 polys][M], DEVICE point order
    size_t brk_party_stride;  // in cplx
    const cplx *monomial;     // [2N][M], entry e-1
    const uint32_t *lwe;      // [B][lwe_stride]: LWE words (mod-switched on the fly) or atilde
    int lwe_stride;
    int pre_switched;         // 1: `lwe` already holds atilde values
    int n, logN;
    int l, logB;              // RGSW gadget
    int blk_len;              // 1, or the block length of LMSS / KMS_block
    int blk_accum;            // block schemes: tacc2 += monomial*tacc form (bootstrapping.jl:157,:648)
    int rows_per_gate;        // rotations per ciphertext
    size_t ngates;            // ciphertexts in this launch (grid = ngates * rows_per_gate)
    const int *slot_party;    // [rows_per_gate]
    const int *slot_row;      // [rows_per_gate]
    int init_mode;            // 0: load acc from acc_io; 1: trivial RLEV row b = 2^(W-(row+1)*logB_lev)
    int logB_lev;
    int out_mode;             // 0: write acc to acc_io; 1: write fft(acc) to tout
    void *acc_io;             // [rot][2][N] ring words
    cplx *tout;               // [rot][2][M]
    int tout_natural;         // 1: reference point order (API output); 0: device order (feeds phase 2)
    int variant;              // tuning: 10*LOGR + transforms per group (0 = default)
    int stagger;              // start-up delay of alternate workgroup groups, in units of 512 cycles (0 = off)
    int wide;                 // latency variant: 0 automatic, 1 never, 2 always where supported (MKT_ROT_WIDE)
    unsigned block0;          // first workgroup index of this launch (a rotation batch may be issued as several launches)
    unsigned split;           // workgroups per launch (0 = the whole batch in one launch)
    int dev_order;            // device point order of the resident tables (fft_device.h dev_pos)
    int map_mode;             // workgroup id -> (ciphertext, slot): 0 slot-major, 1 rows of one (ciphertext, party) eight ids apart (kernel_common.h rot_decode)
    int blk_group;            // block schemes: rotations per workgroup -- 0 automatic, 1 one (blindrotate_k1_kernel<LB>), 2 / 4 (rot_block.hip)
};

// KMS phase 2 (bootstrapping.jl:448-558), one workgroup per ciphertext.
struct Phase2Args {
    TwPtrs tw;
    const uint32_t *lin;      // [B][lwe_stride] linear-combined LWE (b last) for the test vector, or NULL
    int lwe_stride;
    int logN;
    int k, l_lev, logB_lev, l_uni, logB_uni;
    const cplx *levkey;       // [B][Rtot][2][M]
    int rtot;
    const cplx *rlk_d;        // [k][l_uni][M]
    const cplx *rlk_f;        // [k][l_uni][2][M]
    const cplx *pub_b;        // [k][l_uni][M]
    const cplx *crs;          // [l_uni][M]
    void *acc;                // [B][1+k][N] ring words (in: test vector unless lin != NULL; out: result)
    cplx *scratch;            // [B][2*(k+1)][M]
    int dev_order;            // device point order of the resident tables
};

// CCS blind rotation (bootstrapping.jl:234-328), one workgroup per ciphertext.
struct CcsArgs {
    TwPtrs tw;
    const uint32_t *lwe;      // [B][lwe_stride] LWE words or atilde
    int lwe_stride, pre_switched;
    int n, logN, k, l, logB;
    const cplx *brk;          // [party][n][3l][M]: d[l], then (f[j].b, f[j].a) j-major; device point order
    size_t brk_party_stride;
    const cplx *pub_b;        // [party][l][M]
    const cplx *crs;          // [l][M]
    const cplx *monomial;
    void *acc;                // [B][1+k][N] ring words, in place
    cplx *scratch;            // [B][k+1][M]
    void *vscratch;           // [B][3][N] ring words (the two-group kernel uses all three, the one-group kernel the first)
    int stagger;              // start-up delay between the workgroups that share a compute unit, in units of 64 cycles (0 = off)
    int dev_order;            // device point order of the resident tables
};

struct KsArgs {
    const void *acc;          // [B][1+kacc][N] ring words
    uint32_t *out;            // [B][lwe_len]
    const uint32_t *ksk;      // party 0 base; [kr][N][drows][f][n1p] rows padded to n1p = 4*ceil((n+1)/4) words
    size_t ksk_party_stride;  // words
    int n1p;
    int N, n, f, logD, drows;
    int kacc;                 // ring components
    int mk;                   // 1: component i -> party i's KSK and mask block i; 0: single block, ksk comp i
    int balanced;             // block schemes: copy the first words, balanced digits
    int lmss;                 // LMSS flavour of the copy rule (global coefficient index across components)
    uint32_t *digits;         // scratch of the digit-pair kernel (both or neither; sizes: ks_scratch_words): prepared digit words
    uint32_t *partial;        //   and the partial sums of every slab; null: the per-digit kernel with atomics
};
// words of KsArgs::digits / KsArgs::partial for a batch of B ciphertexts (0, 0: this shape runs on the per-digit kernel)
void ks_scratch_words(const KsArgs &a, size_t B, size_t *digit_words, size_t *partial_words);

// Evaluation-key generation on the device (keygen.hip): one party's secrets and the stream key of client.cpp
struct KeygenArgs {
    uint32_t key[8];             // the party's 256-bit ChaCha20 stream key (rng_chacha.h)
    int party;                   // party index (carried in the stream nonce)
    int N, n, W;
    int kr, l, logB;             // RGSW: RLWE length, gadget ; UniEnc: l_uni, logB_uni
    int zoff;                    // first ring-key polynomial used
    int f, logD;                 // key-switching gadget
    double sigma_ring, sigma_lwe;
    const uint32_t *lwekey;      // [n]
    const int8_t *zring;         // [nz][N], entries 0/1
    const void *crs;             // [l][N] ring words (UniEnc only)
    void *out;                   // bootstrapping key, coefficient form, native word width
};
hipError_t launch_keygen_brk(const KeygenArgs &a, int unienc, hipStream_t s);
hipError_t launch_keygen_ksk(const KeygenArgs &a, uint32_t *ksk, int n1p, int kk, int dr, int is_block, hipStream_t s);

hipError_t launch_transform_fwd(int logM, int W, TwPtrs tw, const void *p, cplx *t, size_t B, int dev_order, hipStream_t s);
hipError_t launch_reorder(int logM, const cplx *in, cplx *out, size_t npolys, int to_device, int order, hipStream_t s);
hipError_t launch_transform_inv(int logM, int W, TwPtrs tw, const cplx *t, void *p, size_t B, hipStream_t s);
hipError_t launch_decompose(int W, const void *p, void *digits, int N, int l, int logB, size_t B, hipStream_t s);
// ops / ix / iy may be NULL: one op for the batch / operands in batch order (kernels.hip gate_linear_kernel)
hipError_t launch_gate_linear(int op, const uint8_t *ops, const uint32_t *x, const uint32_t *y, const uint32_t *ix, const uint32_t *iy, size_t pool_rows, uint32_t *out, int len, size_t B, hipStream_t s);   // pool_rows > 0: ix / iy are clamped into the pool (no out-of-bounds read whatever a device-side index array holds)
hipError_t launch_negate(uint32_t *x, size_t words, hipStream_t s);
hipError_t launch_mux_linear(const uint32_t *pool, size_t pool_rows, const uint32_t *is, const uint32_t *ia, const uint32_t *ib, const uint8_t *not_ab, uint32_t *out, int len, size_t B, hipStream_t s);   // [2B][len]: AND(s, a'), then AND(NOT s, b')
hipError_t launch_mux_combine(int W, void *acc, size_t B, size_t words, hipStream_t s);   // acc[j] += acc[B + j], + 2^(W-3) at X^0 of b: [2B][words] -> [B][words]
hipError_t launch_modswitch(const uint32_t *lwe, uint32_t *atilde, uint32_t *btilde, int len, int logN, size_t B, hipStream_t s);
hipError_t launch_testvector(int W, const uint32_t *lin, int lwe_stride, int logN, int kacc, void *acc, size_t B, hipStream_t s);
hipError_t launch_blindrotate_k1(int logM, int W, const RotArgs &a, size_t nrot, hipStream_t s);
bool blockg_supported(int logM, int G);
hipError_t launch_rot_blockg_u32(int logM, int G, int npolys, const RotArgs &a, size_t nslots, hipStream_t s);   // npolys = RLWE length + 1 (2; 3 at block length 1 or 3 and 4 at block length 1, 32-bit ring, G = 4)
hipError_t launch_rot_blockg_u64(int logM, int G, int npolys, const RotArgs &a, size_t nslots, hipStream_t s);
hipError_t launch_blindrotate_kr(int logM, int W, int kr, const RotArgs &a, size_t nrot, hipStream_t s);
// any RLWE length (run-time k; accumulators in memory): scratch = 2 * (kr + 1) * M points per rotation
hipError_t launch_blindrotate_kany(int logM, int W, int kr, const RotArgs &a, cplx *scratch, size_t nrot, hipStream_t s);
hipError_t launch_kms_phase2(int logM, int W, const Phase2Args &a, size_t B, hipStream_t s);
hipError_t launch_ccs_blindrotate(int logM, int W, const CcsArgs &a, size_t B, hipStream_t s);
hipError_t launch_ccs_pipe(int logM, int W, const CcsArgs &a, size_t B, hipStream_t s);   // one ciphertext on two thread groups (ccs_pipe.hip); vscratch [B][3][N]
hipError_t launch_keyswitch(int W, const KsArgs &a, size_t B, hipStream_t s);
bool transform_supported(int logM);

// MKT_ARITH_EXACT (ntt_exact.hip): tab = psi_rev[N] | psiinv_rev[N] | N^-1 | N^-1 2^32, each entry (w mod p1, companion, w mod p2, companion)
hipError_t launch_ntt_fwd(int logN, int W, const uint64_t *tab, const void *p, uint64_t *t, size_t B, int montgomery, hipStream_t s);   // montgomery: output for a resident table (keys, monomials)
hipError_t launch_ntt_inv(int logN, int W, const uint64_t *tab, const uint64_t *t, void *p, size_t B, hipStream_t s);
// CGGI blind rotation (RLWE length 1, 32-bit ring) with exact products; brk [n][2l][2][N], mono [2N][N] residues, natural order
hipError_t launch_exact_blindrotate(int logN, const uint64_t *tab, const uint64_t *brk, const uint64_t *mono, const uint32_t *lwe, int lwe_stride,
                                    int pre_switched, int n, int l, int logB, int blk_len, uint32_t *acc, size_t B, hipStream_t s);
// the same for any RLWE length kr = 1 .. 3 and any block length (digit transforms recomputed per key bit of a block); brk [n][(kr+1) l][kr+1][N]
hipError_t launch_exact_blindrotate_kr(int logN, const uint64_t *tab, const uint64_t *brk, const uint64_t *mono, const uint32_t *lwe, int lwe_stride,
                                       int pre_switched, int n, int kr, int l, int logB, int blk_len, uint32_t *acc, size_t B, hipStream_t s);
// any RLWE length (run-time kr): the transform-domain sums in scratch [B][2][kr+1][N] packed residue pairs
hipError_t launch_exact_blindrotate_kany(int logN, const uint64_t *tab, const uint64_t *brk, const uint64_t *mono, const uint32_t *lwe, int lwe_stride,
                                         int pre_switched, int n, int kr, int l, int logB, int blk_len, uint32_t *acc, uint64_t *scratch, size_t B, hipStream_t s);
// the 64-bit ring with exact products (KMS): resident 64-bit tables as (low, high) residue polynomials per logical polynomial
hipError_t launch_ntt_fwd_split(int logN, const uint64_t *tab, const void *p, uint64_t *out, size_t B, hipStream_t s);
struct ExactKmsArgs {
    const uint64_t *brk; size_t brk_party_stride;   // split tables, u64 units; [n][2l][2 polys][2 halves][N] per party
    const uint64_t *mono;                           // [2N][N] (not split)
    const uint32_t *lwe; int lwe_stride, pre_switched;
    int n, k, l_gsw, logB_gsw, l_lev, logB_lev, l_uni, logB_uni, rtot, lwe_len;
    int blk_len;                                    // KMS_block: key bits per block (1 for KMS)
    const int *slot_party, *slot_row;
    uint64_t *levkey;                               // [B][rtot][2][2][N]
    const uint64_t *rlk_d, *rlk_f, *pub_b, *crs;    // split tables
    const uint32_t *lin_for_tv;                     // bootstrapping.jl:11-23 from the linear combination, or NULL (acc holds the test vector)
    uint64_t *acc, *scratch;                        // [B][1+k][N] ; [B][4(k+1)][N]
    int phase1_only;
    int phase2_only;                                // phase 1 was run elsewhere (fx_exact.hip) and levkey is filled
    int wide;                                       // phase 1, l_gsw = 2: the digit products of an accumulator gathered in 64 bits, one reduction (speed only)
};
hipError_t launch_exact_kms(int logN, const uint64_t *tab, const ExactKmsArgs &a, size_t B, hipStream_t s);
// CCS blind rotation with exact products (32-bit ring); tables as residue pairs, natural order, Montgomery form
struct ExactCcsHostArgs {
    const uint32_t *lwe; int lwe_stride, pre_switched;
    int n, k, l, logB;
    const uint64_t *brk; size_t brk_party_stride;   // 8-byte residue pairs
    const uint64_t *pub_b, *crs, *mono;
    uint32_t *acc; uint64_t *scratch;
};
hipError_t launch_exact_ccs(int logN, const uint64_t *tab, const ExactCcsHostArgs &a, size_t B, hipStream_t s);
hipError_t launch_exact_polymul(int logN, int W, const uint64_t *tab, const void *a, const void *b, void *out, size_t B, hipStream_t s);

// MKT_ARITH_EXACT on the Float64 pipe (fx_exact.hip): exact products from FMA complex transforms over 16-bit key limbs.
// Tables (host_internal.h Twiddles::fx_*): om = cyclic forward twiddles by block, twist = rho^j, nat = inverse twiddles by position.
struct FxRotArgs {
    const cplx *om, *twist, *nat;
    const cplx *brk;          // party 0 base; [n][2l][2 polys][W/16 limbs][M], device point order, scaled by 1 / M
    size_t brk_party_stride;  // in cplx
    const uint32_t *lwe; int lwe_stride, pre_switched;
    int n, logN, l, logB;
    int rows_per_gate; size_t ngates;
    const int *slot_party, *slot_row;
    int init_mode;            // 0: load acc from acc_io; 1: trivial RLEV row (bootstrapping.jl:403-406)
    int logB_lev;
    void *acc_io;             // [rot][2][N] ring words, in place
    int stagger; unsigned block0; int map_mode;
    int split;                // workgroups per launch: 0 = one chip-fill, -1 = everything in one launch (fx_exact.hip launch_fx_blindrotate)
};
bool fx_supported(int logM, int W, int l);
hipError_t launch_fx_key_fwd(int logM, int W, const cplx *om, const cplx *twist, const void *p, cplx *out, size_t npolys, unsigned long long *kmax, hipStream_t s);   // out [npolys][W/16][M]; kmax: largest |transform value|^2 (bit pattern, atomicMax) or NULL
hipError_t launch_fx_polymul(int logM, int W, const cplx *om, const cplx *twist, const cplx *nat, const void *a, const void *b, void *out, size_t B, unsigned long long *resid, hipStream_t s);
hipError_t launch_fx_blindrotate(int logM, int W, const FxRotArgs &a, size_t nrot, hipStream_t s);

}  // namespace mktd
