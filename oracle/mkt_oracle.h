/*
 * mkt_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the SNUCP/MKTFHE gate-bootstrapping hot path
 * (reference: /root/reference/src, Julia).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library, and only as the checker
 * (or the timed CPU baseline) -- the shipped HIP path never calls into it.
 *
 * PARITY STATUS: "parity unpinned" by the reference.  The reference holds no
 * golden vectors / KATs (its tests only assert decryption of random circuits:
 * test/CGGI.jl:34, test/KMS.jl:37 ...), and Julia is not installed in the build
 * image, so the reference cannot be executed here.  The oracle is pinned instead
 * by this repo's own fixtures (tests/golden/): twiddle tables generated with
 * mpmath at 256 bits exactly as fft.jl:31-41 does with BigFloat, exact big-int
 * negacyclic products, decomposition / divbits KATs, decrypt-correctness of
 * random gate circuits (the reference's own test property), and a second,
 * independent numpy transcription of the Julia source (tests/ref_numpy.py) whose
 * CGGI / KMS gate outputs this library equals bit for bit.
 *
 * Conventions
 *   - LWE word is uint32_t everywhere (reference: T = UInt32 in every shipped set).
 *   - Ring words (UInt32 or UInt64 in the reference) are carried in uint64_t and
 *     reduced mod 2^W after every operation (W = 32 or 64).
 *   - Polynomial coefficient indices are 0-based; Julia tables are 1-based.
 *   - A "TransPoly" is M = N/2 interleaved (re, im) doubles in the order the
 *     reference's Cooley-Tukey leaves them (bit-reversed), fft.jl:105-155.
 *   - LWE ciphertext layout: [a_0 .. a_{k*n-1}, b]  (b LAST), k*n+1 words.
 */
#ifndef MKT_ORACLE_H
#define MKT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORA_CGGI = 0, ORA_LMSS = 1, ORA_CCS = 2, ORA_KMS = 3, ORA_KMS_BLOCK = 4 };
enum { ORA_NAND = 0, ORA_AND = 1, ORA_OR = 2, ORA_XOR = 3, ORA_XNOR = 4, ORA_NOR = 5 };

/* scheme parameters: src/tfhe/scheme.jl:6-101, src/tfhe/params.jl */
typedef struct {
    int32_t scheme;   /* ORA_* */
    int32_t n;        /* LWE dimension per party (block variants: d*blk_len) */
    int32_t N;        /* ring dimension */
    int32_t k;        /* SK schemes: RLWE length; MK schemes: number of parties */
    int32_t W;        /* ring word bits: 32 or 64 */
    int32_t l_gsw, logB_gsw;
    int32_t l_lev, logB_lev;
    int32_t l_uni, logB_uni;
    int32_t f, logD;  /* key-switch gadget */
    int32_t blk_len, blk_d; /* l (block length) and d (number of blocks), block variants */
} ora_params;

/* ---- ring: fft.jl, arithmetic.jl, polynomial.jl ---- */
typedef struct ora_ffter ora_ffter;
ora_ffter *ora_ffter_create(int N, int W);                /* fft.jl:18-45 */
void ora_ffter_destroy(ora_ffter *);
/* table access for fixtures: which = 0 Psi, 1 Psiinv, 2 roots, 3 rootsinv; M complex each */
const double *ora_ffter_table(const ora_ffter *, int which);
void ora_fft_fwd(const ora_ffter *, const uint64_t *p, double *t);      /* fft.jl:57-63 + :105-155 */
void ora_fft_inv(const ora_ffter *, double *t /*destroyed*/, uint64_t *p); /* fft.jl:74-81 + :159-209 + arithmetic.jl:1-9 */
uint64_t ora_native(double x, int W);                     /* arithmetic.jl:1-9 */
uint64_t ora_divbits(uint64_t a, int bit, int W);         /* arithmetic.jl:23-27 */
void ora_tp_muladd(double *res, const double *x, const double *y, int M); /* polynomial.jl:105 */
void ora_tp_mulsub(double *res, const double *x, const double *y, int M); /* polynomial.jl:110 */
void ora_tp_mul(double *res, const double *x, const double *y, int M);    /* polynomial.jl:99 */
void ora_monomial(const ora_ffter *, int e /*1..2N*/, double *out);       /* scheme.jl:121-146 */
/* exact negacyclic product mod 2^W (schoolbook) -- test helper, no reference counterpart */
void ora_negacyclic_schoolbook(const uint64_t *a, const uint64_t *b, uint64_t *out, int N, int W);

/* ---- gadget decomposition: gsw.jl:34-110, unienc.jl:4-18 ---- */
void ora_decomp_word(uint64_t a, int l, int logB, int W, uint64_t *out /*l*/);          /* gsw.jl:42-52 */
void ora_unbalanced_decomp_word(uint64_t a, int l, int logB, int W, uint64_t *out);     /* gsw.jl:34-40 */
void ora_decomp_poly(const uint64_t *a, int N, int l, int logB, int W, uint64_t *out /*[l][N]*/); /* gsw.jl:86-96 */

/* ---- scheme object (read-only during evaluation; scheme.jl:107-116,:256-265 ...) ---- */
typedef struct ora_scheme ora_scheme;
ora_scheme *ora_scheme_create(const ora_params *);
void ora_scheme_destroy(ora_scheme *);
const ora_ffter *ora_scheme_ffter(const ora_scheme *);

/* Key loading, integer (coefficient) form; the oracle forward-transforms with its own
 * Float64 transformer exactly as keygen.jl:14,:67,:99-108 do via fft(..., ffter).
 * BRK of one party (SK schemes: party 0):
 *   RGSW schemes (CGGI/LMSS/KMS/KMS_BLOCK): [n][rows=(kr+1)*l_gsw][polys=kr+1][N] ring words,
 *     rows ordered basketb.stack[0..l), basketa[0].stack[0..l), ... (gsw.jl:219-227),
 *     polys ordered (b, a_0..a_{kr-1}) (lwe.jl:165-179); kr = k for SK schemes, 1 for KMS.
 *   CCS (TransUniEnc, unienc.jl:92-99): [n][3*l_uni][N]: d[0..l), then f.stack[j].b, f.stack[j].a (j-major).
 */
int ora_set_brk(ora_scheme *, int party, const uint64_t *brk_int);
/* KSK of one party: [kr][N][Drows][f][n+1] uint32, LWE rows laid out [a_0..a_{n-1}, b];
 * Drows = D-1 (CGGI/CCS/KMS) or D/2 (LMSS/KMS_BLOCK); entry d (0-based) encrypts (d+1)*z_j. keygen.jl:17-23 */
int ora_set_ksk(ora_scheme *, int party, const uint32_t *ksk);
/* KMS relinearisation key (UniEnc_z(z'), keygen.jl:103): d [l_uni][N]; f [l_uni][2][N] (b, a) */
int ora_set_rlk(ora_scheme *, int party, const uint64_t *d_int, const uint64_t *f_int);
/* CCS/KMS public key b (unienc.jl:77-90): [l_uni][N] */
int ora_set_pubkey(ora_scheme *, int party, const uint64_t *b_int);
/* common reference string (scheme.jl:409-410): [l_uni][N] */
int ora_set_crs(ora_scheme *, const uint64_t *a_int);

/* ---- hot path: bootstrapping.jl, gate.jl ---- */
void ora_gate_linear(int op, const uint32_t *x, const uint32_t *y, uint32_t *out, int len /*k*n+1*/); /* gate.jl:1-53 */
void ora_not(uint32_t *x, int len);                                            /* gate.jl:55-58 */
void ora_modswitch(const ora_scheme *, const uint32_t *lwe, uint32_t *atilde /*k*n*/, uint32_t *btilde); /* bootstrapping.jl:8-9 */
void ora_testvector(const ora_scheme *, uint32_t btilde, uint64_t *acc /*(kacc+1)*N, b first*/); /* bootstrapping.jl:11-23 */
void ora_blindrotate(const ora_scheme *, const uint32_t *atilde, uint64_t *acc); /* bootstrapping.jl:32,:114,:234,:369 */
void ora_keyswitch(const ora_scheme *, const uint64_t *acc, uint32_t *lwe_out);  /* bootstrapping.jl:81,:170,:333,:564,:664 */
void ora_bootstrap(const ora_scheme *, uint32_t *lwe /*in place*/);              /* bootstrapping.jl:4-27 */
void ora_gate(const ora_scheme *, int op, const uint32_t *x, const uint32_t *y, uint32_t *out);
/* KMS phase 1 of one party (bootstrapping.jl:389-443 / :599-659); out: [R_p][2][M] complex, returns R_p */
int ora_kms_phase1(const ora_scheme *, int party /*0-based*/, const uint32_t *atilde_party /*n*/, double *levkey);
void ora_kms_phase2(const ora_scheme *, double *const *levkey /*k ptrs*/, uint64_t *acc); /* bootstrapping.jl:448-558 */
/* number of accumulator mask polynomials (k for every scheme) */
int ora_acc_polys(const ora_scheme *);
/* batch drivers used as the timed CPU baseline: `threads` pthreads, one gate per worker */
void ora_gate_batch(const ora_scheme *, int op, const uint32_t *x, const uint32_t *y, uint32_t *out, size_t B, int threads);
void ora_fft_fwd_batch(const ora_ffter *, const uint64_t *p, double *t, size_t B);
void ora_fft_inv_batch(const ora_ffter *, double *t, uint64_t *p, size_t B);

#ifdef __cplusplus
}
#endif
#endif
