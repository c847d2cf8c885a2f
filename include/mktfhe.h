/*
 * mktfhe.h -- C ABI of the MI355X-native multi-key TFHE gate-bootstrapping engine.
 *
 * Drop-in boundary for the hot path of SNUCP/MKTFHE (reference: Julia, /root/reference/src).
 * The reference has no FFI; its boundary is the Julia dispatch surface
 *     bootstrapping!(ctxt::LWE, scheme)          src/tfhe/bootstrapping.jl:4
 *     NAND/AND/OR/XOR/XNOR/NOR(c1, c2, scheme)   src/tfhe/gate.jl:1-53,  NOT!(c) gate.jl:55
 *     blindrotate!(atilde, acc, scheme)          bootstrapping.jl:32,:114,:234,:369
 *     keyswitch!(res, acc, scheme)               bootstrapping.jl:81,:170,:333,:564,:664
 * A Julia `ccall` shim flattens its pointer-graph objects into the flat
 * layouts documented here and calls these entry points.  Plain pointers and sizes only.
 *
 * Every function returns MKT_OK (0) or a negative mkt_status; none throws.
 * The engine REQUIRES a gfx950 GPU: there is no CPU fallback on any compute entry point.
 *
 * Layouts (0-based, row-major, innermost last)
 *   LWE ciphertext      : uint32 [k*n + 1] = [a_0 .. a_{k*n-1}, b]   (lwe.jl:1-9; MK mask is the
 *                         concatenation of k per-party blocks of n words, scheme.jl:379-386)
 *   ring polynomial     : N ring words; ring word = uint32 if W == 32, uint64 if W == 64
 *   RLWE accumulator    : [1 + kacc][N] ring words = (b, a_0 .. a_{kacc-1})   (lwe.jl:61-76)
 *   TransPoly           : M = N/2 complex doubles (re, im interleaved) in the order the reference's
 *                         transform leaves them (bit-reversed; fft.jl:105-155)
 *   BRK, RGSW schemes   : per party [n][(kr+1)*l_gsw rows][kr+1 polys] polynomials; rows ordered
 *                         basketb.stack[0..l), basketa[0].stack[0..l), ...  (gsw.jl:219-227);
 *                         polys ordered (b, a_0 ..) (lwe.jl:165-179); kr = k (SK) or 1 (KMS)
 *   BRK, CCS            : per party [n][3*l_uni]: d[0..l), then (f.stack[j].b, f.stack[j].a) j-major
 *                         (unienc.jl:92-99)
 *   KSK                 : per party [kr][N][Drows][f][n+1] uint32 LWE rows ([a.., b]); entry d
 *                         encrypts (d+1)*z_j*2^(32-(t+1)logD); Drows = D-1, or D/2 for the block
 *                         schemes (keygen.jl:17-23,:37,:141)
 */
#ifndef MKTFHE_H
#define MKTFHE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MKT_ABI_VERSION 3

typedef enum {
    MKT_OK = 0,
    MKT_ERR_ARG = -1,          /* bad argument / size mismatch (reference: @assert, polynomial.jl:10,19-20) */
    MKT_ERR_UNSUPPORTED = -2,  /* parameter combination not implemented */
    MKT_ERR_NO_DEVICE = -3,    /* no gfx950 device / HIP runtime failure at context creation */
    MKT_ERR_HIP = -4,          /* HIP runtime error, see mkt_last_error */
    MKT_ERR_STATE = -5,        /* keys not loaded */
    MKT_ERR_NOMEM = -6
} mkt_status;

/* scheme kinds: scheme.jl:107 (CGGI), :168 (LMSS), :209 (CCS), :256 (KMS), :301 (KMS_block) */
enum { MKT_CGGI = 0, MKT_LMSS = 1, MKT_CCS = 2, MKT_KMS = 3, MKT_KMS_BLOCK = 4 };
/* gates: gate.jl:1-53 */
enum { MKT_NAND = 0, MKT_AND = 1, MKT_OR = 2, MKT_XOR = 3, MKT_XNOR = 4, MKT_NOR = 5 };
/* per-gate codes of mkt_gate_batch_ops / _gather: a gate of the enum above, optionally with one or both inputs negated first
 * (NOT!, gate.jl:55-58, folded into the gate's linear part: the same words as NOT! followed by the gate) */
enum { MKT_OP_NOT_X = 8, MKT_OP_NOT_Y = 16 };
/* arithmetic modes of the negacyclic transform */
enum {
    MKT_ARITH_F64REF = 0, /* the reference's Float64 twisted FFT, operation for operation (fft.jl) */
    MKT_ARITH_EXACT = 1   /* EXACT products (what the reference's transform approximates, polynomial.jl:99-113; its MultiFloat option, README.md:9), two
                             implementations with the same words (option "exact_impl"): Float64 FMA transforms over centered 16-bit key limbs whose rounding
                             is proven exact for the loaded keys (fx_exact.hip: CGGI with RLWE length 1, KMS phase 1; mkt_get_metric "fx_bound"), and
                             exact integer arithmetic: the negacyclic NTT over Z_P[X]/(X^N+1) in residue form, P = p1 p2 =
                             (131063 * 2^13 + 1)(131066 * 2^13 + 1) = 2^59.9998, the two largest NTT primes below 2^30
                             (4 p < 2^32: lazy butterflies).
                             Transform-level entry points (mkt_transform_*_batch, mkt_exact_polymul_batch,
                             mkt_decompose_batch, mkt_modswitch_batch, mkt_not_batch) for every scheme; the gate path
                             (mkt_load_*, mkt_keygen_device, mkt_gate, mkt_bootstrap, mkt_blindrotate, mkt_keyswitch) for
                             MKT_CGGI / MKT_LMSS (any RLWE length and block length, 32-bit ring), MKT_CCS (32-bit ring) and MKT_KMS /
                             MKT_KMS_BLOCK (64-bit ring: every 64-bit table kept as the transforms of its two centered
                             32-bit pieces), provided the context's gadgets keep every product sum below P / 2 (checked at the
                             first key upload: MKT_ERR_UNSUPPORTED otherwise).  The ciphertexts are valid -- on the
                             64-bit ring 3-4x LESS noisy than the Float64 path, whose transform error dominates there --
                             but NOT the reference's words (no Float64 rounding); keys in MKT_FMT_INT_COEFF only.
                             On such a KMS context mkt_kms_phase1_batch returns the rows as split residue tables
                             [B][rows][2 polynomials][low, high half][N] uint64 (Montgomery form)   */
};
/* where batch pointers live */
enum { MKT_MEM_DEVICE = 0, MKT_MEM_HOST = 1 };
/* key data formats */
enum {
    MKT_FMT_INT_COEFF = 0, /* coefficient-form ring words; the engine transforms on device
                              (replaces keygen.jl:14,:67,:99-108 `fft(..., ffter)`) */
    MKT_FMT_F64_FFT = 1    /* the reference's Trans* values (TransPoly layout above) */
};

/* parameter block: scheme.jl:6-101 / params.jl.  LWE word is 32 bit in every shipped set. */
typedef struct {
    int32_t scheme;          /* MKT_CGGI .. MKT_KMS_BLOCK */
    int32_t n;               /* LWE dimension per party (block schemes: blk_d * blk_len) */
    int32_t N;               /* ring dimension (power of two, 256..4096) */
    int32_t k;               /* SK: RLWE length (any; beyond 3 on a run-time-k kernel, F64REF only); MK: number of parties */
    int32_t W;               /* ring word bits: 32 or 64 */
    int32_t l_gsw, logB_gsw; /* RGSW gadget (CGGI/LMSS/KMS) */
    int32_t l_lev, logB_lev; /* LEV gadget (KMS) */
    int32_t l_uni, logB_uni; /* UniEnc gadget (CCS/KMS) */
    int32_t f, logD;         /* key-switch gadget */
    int32_t blk_len, blk_d;  /* block length and count (LMSS/KMS_block) */
} mkt_params;

typedef struct mkt_ctx mkt_ctx;
typedef struct mkt_client_party mkt_client_party;   /* one party's keys (client section below) */

/* ---- context: replaces the scheme object (scheme.jl:107-116 ...), FFTransformer (fft.jl:18-45)
 *      and getmonomial (scheme.jl:121-146), all built on device `device`. ---- */
int mkt_abi_version(void);
/* which source tree this library was built from: 16 hex digits of the SHA-256 over the engine's sources and build flags (csrc/Makefile).
 * Measurement bookkeeping only: profiles record it, bench.py quotes profile-derived numbers only for the library that produced them */
const char *mkt_build_id(void);
int mkt_ctx_create(const mkt_params *params, int arith_mode, int device, mkt_ctx **out);
int mkt_ctx_destroy(mkt_ctx *ctx);
/* a second context over the SAME resident keys and tables (no copy; freed with the last context that holds them; both
 * arithmetic modes): own stream, workspace and timing -- one per concurrent caller, matching the reference's contract of one read-only scheme
 * shared by concurrent bootstrapping! calls (bootstrapping.jl:38-45: all scratch is per call).  Once forked, the key set
 * is immutable: mkt_load_*, mkt_set_twiddles and mkt_keygen_device return MKT_ERR_STATE on every sharer. */
int mkt_ctx_fork(mkt_ctx *ctx, mkt_ctx **out);
const char *mkt_last_error(const mkt_ctx *ctx); /* ctx may be NULL: last creation error */
/* HIP stream (hipStream_t) all subsequent batch calls are enqueued on; NULL = default stream.  The per-batch workspace
 * belongs to the context: calls on one context must not overlap (one context per host thread / stream: mkt_ctx_fork).
 * STREAM ORDER.  A context made by mkt_ctx_create starts on the NULL stream.  A context made by mkt_ctx_fork starts on its
 * OWN stream, created hipStreamNonBlocking: its calls are NOT ordered against work on the NULL stream or on any other
 * stream.  A caller that hands MKT_MEM_DEVICE buffers to a fork must order producer and consumer itself: point the fork at
 * the producing stream (mkt_set_stream), or wait on the fork's stream (mkt_get_stream + events, or mkt_synchronize). */
int mkt_set_stream(mkt_ctx *ctx, void *hip_stream);
int mkt_get_stream(mkt_ctx *ctx, void **hip_stream);   /* the stream the context enqueues on right now */
int mkt_synchronize(mkt_ctx *ctx);
/* kernel-selection switches (where a step has more than one kernel, parity tests force each; A/B tools): name one of
 * "rot_variant", "rot_stagger", "rot_split", "rot_wide" (latency variant: 0 auto, 1 never, 2 always), "rot_blkg" (block
 * schemes, rotations per workgroup: 0 auto, 1, 2, 4), "rot_map" (workgroup -> (ciphertext, slot) dealing: 0 plain, 1 XCD-aware),
 * "ccs_stagger", "ccs_pipe" (-1 auto, 0 never, 1 always), "exact_wide" (MKT_ARITH_EXACT on the integer NTT, KMS phase 1 at l_gsw = 2 / KMS_block: 1 the default kernel, 0 the one-at-a-time form),
 * "exact_kany" (MKT_ARITH_EXACT CGGI / LMSS: 1 = the run-time-RLWE-length kernel also at k <= 3),
 * "exact_impl" (MKT_ARITH_EXACT blind rotation of CGGI with RLWE length 1 and of KMS phase 1, and mkt_exact_polymul_batch:
 * 0 = the integer NTT over two 30-bit primes, 1 = Float64 FMA transforms over 16-bit key limbs wherever the proven rounding bound
 * certifies the loaded keys, -1 (default) = 1 for the gate paths; both give the same words).  Results never
 * depend on them.  The environment (MKT_ROT_*, MKT_CCS_*) seeds them once, at mkt_ctx_create; no batch call reads it. */
int mkt_set_option(mkt_ctx *ctx, const char *name, int value);
/* base name of the blind-rotation kernel the last batch call of this context launched ("" before the first) */
const char *mkt_last_kernel_name(const mkt_ctx *ctx);
/* diagnostics of the Float64-pipe implementation of MKT_ARITH_EXACT: "fx_available" (1: the loaded keys are certified and exact_impl
 * admits it), "fx_bound" (proven bound on |computed - exact| of a rounded product sum for the loaded keys; must stay below 1/2),
 * "fx_kmax" (largest transform-domain magnitude of the loaded key limbs), "fx_last_resid" (largest rounding distance met by the last
 * mkt_exact_polymul_batch under exact_impl = 1) */
int mkt_get_metric(mkt_ctx *ctx, const char *name, double *out);

/* twiddle tables (fft.jl:31-41): which = 0 Psi, 1 Psiinv, 2 roots, 3 rootsinv; M complex each.
 * mkt_set_twiddles lets a caller install the reference's own ffter tables verbatim. */
int mkt_get_twiddles(mkt_ctx *ctx, int which, double *out_host);
int mkt_set_twiddles(mkt_ctx *ctx, const double *psi, const double *psiinv,
                     const double *roots, const double *rootsinv);

/* host-only (no GPU needed): the engine's table generator for ring dimension N, same `which` */
int mkt_make_twiddles(int N, int which, double *out_host);

/* ---- evaluation keys (host pointers, copied; keygen.jl:3-155) ---- */
int mkt_load_brk(mkt_ctx *ctx, int party, const void *data, int fmt);
int mkt_load_ksk(mkt_ctx *ctx, int party, const uint32_t *data);
int mkt_load_rlk(mkt_ctx *ctx, int party, const void *d, const void *f, int fmt); /* KMS: keygen.jl:103 */
int mkt_load_pubkey(mkt_ctx *ctx, int party, const void *b, int fmt);             /* CCS/KMS: keygen.jl:67,:100 */
int mkt_load_crs(mkt_ctx *ctx, const void *a, int fmt);

/* ---- key generation on the device: replaces keygen.jl:13-23, :39-51, :71-79, :106-114,
 *      :143-151 (RGSW / UniEnc bootstrapping key, LEV key-switching key) for party `party`.  Exact integer arithmetic;
 *      `keys` holds the party's secrets (mkt_client_party_secrets or _keygen); `crs` = the integer CRS
 *      (mkt_client_crs) for CCS, NULL otherwise.  Equivalent to mkt_load_brk + mkt_load_ksk of the host-generated
 *      keys of the same seed, word for word.
 *      TRUST: this call hands party `party`'s SECRET keys to the GPU of this context.  It is a party-local operation:
 *      a party runs it on its own machine / GPU and ships the resulting evaluation keys (key blob) to the evaluator.
 *      An evaluator context that runs it for every party holds all k secrets -- acceptable only in tests and
 *      benchmarks.  The secret buffers are zeroed on the device before they are freed and the host-side copies of the
 *      party's stream key are wiped; the copy of that key carried in the kernel-argument segment of the keygen launches
 *      lives in runtime-owned memory the library cannot erase (it is overwritten by later launches). ---- */
int mkt_keygen_device(mkt_ctx *ctx, int party, const mkt_client_party *keys, const void *crs);
/* the same, and the keys are also copied to the host: brk_out in the MKT_FMT_INT_COEFF layout of mkt_load_brk, ksk_out
 * in the layout of mkt_load_ksk -- what a party ships to the evaluator after generating its keys on its own GPU */
int mkt_keygen_device_export(mkt_ctx *ctx, int party, const mkt_client_party *keys, const void *crs,
                             void *brk_out, uint32_t *ksk_out);
/* read a party's key-switching key back in the host layout of mkt_load_ksk (tests) */
int mkt_get_ksk(mkt_ctx *ctx, int party, uint32_t *out_host);                           /* scheme.jl:409-410 */

/* ---- hot path, batched: B independent ciphertexts per call (the reference does one per call) ---- */
/* gate.jl:1-53: out = bootstrapping!(linear(op, x, y)); x, y, out: [B][k*n+1] */
int mkt_gate_batch(mkt_ctx *ctx, int op, const uint32_t *x, const uint32_t *y, uint32_t *out, size_t B, int mem);
/* a different gate per ciphertext pair, one launch sequence for the batch -- the shape of the reference's tests, which draw a
 * random gate per step (test/KMS.jl:29-34): ops[j] = MKT_NAND .. MKT_NOR, optionally | MKT_OP_NOT_X / MKT_OP_NOT_Y; ops
 * lives where x, y, out live (`mem`) */
int mkt_gate_batch_ops(mkt_ctx *ctx, const uint8_t *ops, const uint32_t *x, const uint32_t *y, uint32_t *out, size_t B, int mem);
/* one level of a gate circuit: gate j reads rows ix[j] and iy[j] of pool [pool_rows][k*n+1] and writes out[j]; out may be a
 * later region of the same pool provided no gate of this call reads a row this call writes.
 * Validation: with MKT_MEM_HOST every index and gate code is checked (MKT_ERR_ARG).  With MKT_MEM_DEVICE the arrays cannot be read by
 * the host without a synchronising copy, so valid indices (< pool_rows) and codes (bits 0-2 <= MKT_NOR, bits 5-7 clear) are the CALLER's
 * precondition (mktfhe_amd/circuit.py builds them from a validated plan); what the engine guarantees regardless is memory safety -- an
 * index beyond the pool is clamped to its last row, a code is read modulo its defined bits -- never an out-of-bounds access.  The same
 * holds for mkt_gate_batch_ops and mkt_mux_batch_gather. */
int mkt_gate_batch_gather(mkt_ctx *ctx, const uint8_t *ops, const uint32_t *pool, size_t pool_rows, const uint32_t *ix,
                          const uint32_t *iy, uint32_t *out, size_t B, int mem);
/* MUX(s, a, b) = s ? a : b -- named by the north star; the reference has no MUX gate (gate.jl:1-57).  Two blind rotations and one
 * key switch, built from the reference's own operators as CGGI16 builds it:
 *   acc = blindrotate!(AND-linear(s, a)) + blindrotate!(AND-linear(NOT! s, b)), + 1/8 at X^0 of acc.b;  out = keyswitch!(acc)
 * (a composite OR(AND(s, a), AND(NOT s, b)) of the reference's gates takes three full bootstraps).  s, a, b, out: [B][k*n+1] */
int mkt_mux_batch(mkt_ctx *ctx, const uint32_t *s, const uint32_t *a, const uint32_t *b, uint32_t *out, size_t B, int mem);
/* a circuit level of MUX gates: gate j = MUX(pool[is[j]], a', b') -> out[j], a' = pool[ia[j]] or its negation (NOT!) if bit 0 of
 * not_ab[j] is set, b' likewise with bit 1; not_ab may be NULL (rows of pool [pool_rows][k*n+1]; a negated selector = a and b swapped) */
int mkt_mux_batch_gather(mkt_ctx *ctx, const uint32_t *pool, size_t pool_rows, const uint32_t *is, const uint32_t *ia, const uint32_t *ib,
                         const uint8_t *not_ab, uint32_t *out, size_t B, int mem);
/* gate.jl:55-58 NOT!: in-place negation, no bootstrap */
int mkt_not_batch(mkt_ctx *ctx, uint32_t *x, size_t B, int mem);
/* bootstrapping.jl:4-27 bootstrapping!: in place on [B][k*n+1] */
int mkt_bootstrap_batch(mkt_ctx *ctx, uint32_t *lwe, size_t B, int mem);
/* bootstrapping.jl:8-9: atilde [B][k*n], btilde [B] */
int mkt_modswitch_batch(mkt_ctx *ctx, const uint32_t *lwe, uint32_t *atilde, uint32_t *btilde, size_t B, int mem);
/* blindrotate!(atilde, acc, scheme): acc [B][1+k][N] ring words, updated in place */
int mkt_blindrotate_batch(mkt_ctx *ctx, const uint32_t *atilde, void *acc, size_t B, int mem);
/* keyswitch!(res, acc, scheme): acc [B][1+k][N] -> out [B][k*n+1] */
int mkt_keyswitch_batch(mkt_ctx *ctx, const void *acc, uint32_t *out, size_t B, int mem);
/* KMS phase_1 (bootstrapping.jl:389-443 / :599-659) of every party: atilde [B][k*n] ->
 * levkey [B][Rtot][2][M] complex, Rtot = 1 + (k-1)*l_lev, party-major rows */
int mkt_kms_phase1_batch(mkt_ctx *ctx, const uint32_t *atilde, double *levkey, size_t B, int mem);

/* ---- unit-level entry points (parity tests, transform roofline) ----
 * On an MKT_ARITH_EXACT context a TransPoly is N residue pairs (x mod p1) | (x mod p2) << 32 (uint64, the same 8 N bytes as
 * M complex doubles), in the bit-reversed order the Cooley-Tukey network with psi_rev[m + i] leaves them; forward reads
 * ring words as SIGNED integers, inverse returns the integer of least magnitude mod P (the identity for |x| < P / 2),
 * reduced mod 2^W. */
/* fft.jl:57-63 fftto!: p [B][N] ring words -> t [B][M] complex */
int mkt_transform_fwd_batch(mkt_ctx *ctx, const void *p, double *t, size_t B, int mem);
/* fft.jl:74-81 ifftto!: t [B][M] complex -> p [B][N] ring words (t is left unmodified) */
int mkt_transform_inv_batch(mkt_ctx *ctx, const double *t, void *p, size_t B, int mem);
/* gsw.jl:86-96 decompto!: p [B][N] -> digits [B][l][N] ring words (wrapped signed digits) */
int mkt_decompose_batch(mkt_ctx *ctx, const void *p, void *digits, int l, int logB, size_t B, int mem);
/* MKT_ARITH_EXACT only: out = a (*) b in Z_{2^W}[X]/(X^N+1), EXACTLY, for a gadget-digit polynomial a (signed words,
 * N * max|a_i| < 2^28) and any ring polynomial b -- the product the reference's Float64 transform approximates
 * (polynomial.jl:99-113); a, b, out: [B][N] ring words */
int mkt_exact_polymul_batch(mkt_ctx *ctx, const void *a, const void *b, void *out, size_t B, int mem);
/* scheme.jl:121-146: copy monomial table entry e (1..2N) to host, M complex */
int mkt_get_monomial(mkt_ctx *ctx, int e, double *out_host);

/* ---- one evaluator over several GPUs, one caller process (the reference's caller is ONE process whose threads
 *      share one read-only scheme: README.md:38-44, bootstrapping.jl:38-45).  Shard i runs on HIP device devices[i]; naming a
 *      device more than once makes logical shards that share that device's ONE key set (forked contexts).  Keys are loaded /
 *      generated once, on devices[0]; mkt_multi_replicate copies the resident pre-transformed tables to the other devices
 *      (hipMemcpyPeer over xGMI; host bounce if refused) and seals them.  A batch call cuts [0, B) into contiguous balanced
 *      slices (the first B mod nshards slices hold one more: mkt_multi_shard_range), one host thread per shard, every shard
 *      writing its slice of the caller's ONE output array; no collective.  MKT_MEM_DEVICE arrays may live on any of the
 *      devices (slices are peer-copied to and from the shards on other devices).  Calls return when all shards are done; one call
 *      at a time per handle (the shards' workspaces and staging buffers belong to it) -- concurrent callers take one handle each, or
 *      fork the shard contexts (mkt_multi_ctx + mkt_ctx_fork). ---- */
typedef struct mkt_multi mkt_multi;
/* flags: MKT_MULTI_PRIVATE_KEYS = shards that share a device do NOT share its key set: each gets its own replicated copy, as
 * shards on distinct devices do (exercises the device-to-device replication on a one-GPU box; costs one key copy per shard) */
/* MKT_MULTI_STAGE_ALWAYS = device-resident arguments always travel through the shards' staging buffers (peer copies), as they do
 * for a shard on another device than the array's (that path on a one-GPU box) */
/* MKT_MULTI_NO_PEER = every device-to-device copy (key replication, staged arguments) goes through a host buffer instead of
 * hipMemcpyPeer: the path the engine falls back to by itself where a peer copy is refused, selectable so that it can be tested */
enum { MKT_MULTI_PRIVATE_KEYS = 1, MKT_MULTI_STAGE_ALWAYS = 2, MKT_MULTI_NO_PEER = 4 };
int mkt_multi_create(const mkt_params *params, int arith_mode, const int *devices, int nshards, int flags, mkt_multi **out);
int mkt_multi_destroy(mkt_multi *m);
const char *mkt_multi_last_error(const mkt_multi *m);  /* m may be NULL: last creation error */
int mkt_multi_nshards(const mkt_multi *m);
int mkt_multi_device(const mkt_multi *m, int shard);
mkt_ctx *mkt_multi_ctx(mkt_multi *m, int shard);        /* borrowed: timing (mkt_enable_timing), kernel names; NULL for a logical shard before mkt_multi_replicate */
int mkt_multi_shard_range(const mkt_multi *m, size_t B, int shard, size_t *lo, size_t *hi);
int mkt_multi_load_brk(mkt_multi *m, int party, const void *data, int fmt);
int mkt_multi_load_ksk(mkt_multi *m, int party, const uint32_t *data);
int mkt_multi_load_rlk(mkt_multi *m, int party, const void *d, const void *f, int fmt);
int mkt_multi_load_pubkey(mkt_multi *m, int party, const void *b, int fmt);
int mkt_multi_load_crs(mkt_multi *m, const void *a, int fmt);
int mkt_multi_keygen_device(mkt_multi *m, int party, const mkt_client_party *keys, const void *crs);
int mkt_multi_replicate(mkt_multi *m);
int mkt_multi_set_option(mkt_multi *m, const char *name, int value);
int mkt_multi_gate_batch(mkt_multi *m, int op, const uint32_t *x, const uint32_t *y, uint32_t *out, size_t B, int mem);
int mkt_multi_gate_batch_ops(mkt_multi *m, const uint8_t *ops, const uint32_t *x, const uint32_t *y, uint32_t *out, size_t B, int mem);
int mkt_multi_mux_batch(mkt_multi *m, const uint32_t *s, const uint32_t *a, const uint32_t *b, uint32_t *out, size_t B, int mem);
int mkt_multi_bootstrap_batch(mkt_multi *m, uint32_t *lwe, size_t B, int mem);
int mkt_multi_not_batch(mkt_multi *m, uint32_t *x, size_t B, int mem);
int mkt_multi_blindrotate_batch(mkt_multi *m, const uint32_t *atilde, void *acc, size_t B, int mem);
int mkt_multi_keyswitch_batch(mkt_multi *m, const void *acc, uint32_t *out, size_t B, int mem);

/* Device-time accounting with hipEvents recorded on the context's stream around each kernel class:
 * mkt_enable_timing(ctx, 1) clears and starts recording, mkt_last_kernel_ms stores the TOTAL ms of
 * class `which` (0 = whole call, 1 = blind rotation, 2 = key switch, 3 = transform, 4 = KMS phase 2)
 * since then and returns the number of recorded launches (average = total / count); <0 on error. */
int mkt_enable_timing(mkt_ctx *ctx, int on);
int mkt_last_kernel_ms(mkt_ctx *ctx, int which, double *ms);

/* ---- client side (host only, no GPU): counterparts of the reference's key generation and encryption,
 *      exact integer arithmetic.  setup/party_keygen scheme.jl:151,:190,:227,:273,:324; keygen.jl;
 *      lwe_encrypt scheme.jl:352-386; lwe_decrypt scheme.jl:388-407; CRS scheme.jl:409
 *
 *      RANDOMNESS.  Every `seed` below is a 256-bit value (32 bytes) keying ChaCha20 streams, or NULL.
 *      NULL -- what a caller should pass -- draws a fresh seed from the OS (getrandom) for that call, as the
 *      reference draws fresh ChaCha20 entropy per call (sampler.jl:1-34).  A pinned seed makes keys and
 *      ciphertexts reproducible and therefore PUBLIC: pinned seeds (mkt_client_test_seed) are for tests and
 *      benchmarks only. ---- */
int mkt_client_random_seed(uint8_t out[32]);            /* 32 bytes of OS entropy */
int mkt_client_test_seed(uint64_t n, uint8_t out[32]);  /* TESTS / BENCHMARKS ONLY: deterministic expansion of n */
/* crs: [l_uni][N] ring words (MK schemes) */
int mkt_client_crs(const mkt_params *params, const uint8_t *seed, void *crs_out);
/* one party's secret + evaluation keys; crs may be NULL for SK schemes; sigma_lwe/sigma_ring are the
 * absolute noise standard deviations alpha/beta of params.jl */
int mkt_client_party_keygen(const mkt_params *params, const uint8_t *seed, int party, const void *crs,
                            double sigma_lwe, double sigma_ring, mkt_client_party **out);
/* the same party WITHOUT the two large keys (bootstrapping key, key-switching key: mkt_client_brk / _ksk are empty):
 * secrets, public key and relinearisation key only -- for mkt_keygen_device, which generates the large keys on the
 * GPU from the same streams (identical words to mkt_client_party_keygen with the same seed) */
int mkt_client_party_secrets(const mkt_params *params, const uint8_t *seed, int party, const void *crs,
                             double sigma_lwe, double sigma_ring, mkt_client_party **out);
int mkt_client_party_destroy(mkt_client_party *p);   /* wipes the secrets */
/* sizes in bytes / pointers to the flat key material (layouts above), valid until destroy */
const uint32_t *mkt_client_lwekey(const mkt_client_party *p);             /* [n] 0/1 */
/* ring secret polynomial idx, N entries 0/1 (SK schemes: idx < k; CCS: 0; KMS: 0 = gsw key, 1 = uni key); key.jl */
const int8_t *mkt_client_ringkey(const mkt_client_party *p, int idx, size_t *bytes);
const void *mkt_client_brk(const mkt_client_party *p, size_t *bytes);     /* INT_COEFF */
const uint32_t *mkt_client_ksk(const mkt_client_party *p, size_t *bytes);
const void *mkt_client_rlk_d(const mkt_client_party *p, size_t *bytes);
const void *mkt_client_rlk_f(const mkt_client_party *p, size_t *bytes);
const void *mkt_client_pubkey(const mkt_client_party *p, size_t *bytes);
/* lwe_encrypt (SK: party = 0) / lwe_ith_encrypt (MK): out [k*n+1] */
int mkt_client_lwe_encrypt(const mkt_params *params, const mkt_client_party *p, int party, int bit,
                           double sigma_lwe, const uint8_t *seed, uint32_t *out);
/* lwe_decrypt: keys = nparties pointers; returns 0/1, <0 on error */
int mkt_client_lwe_decrypt(const mkt_params *params, const mkt_client_party *const *keys, int nparties,
                           const uint32_t *lwe);

#ifdef __cplusplus
}
#endif
#endif
