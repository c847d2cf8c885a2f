/*
 * mkt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See mkt_oracle.h.
 *
 * Restates, function by function, the Julia reference under /root/reference/src.
 * Every function cites the file:line it follows.  Floating point: IEEE-754 double,
 * one rounding per operation, NO FMA contraction (compile with -ffp-contract=off),
 * complex product (xr*yr - xi*yi, xr*yi + xi*yr) exactly as Julia's Base complex `*`.
 *
 * parity unpinned by the reference (no golden vectors there, Julia not runnable here);
 * pinned by tests/golden/ fixtures -- see the header.
 */
#define _GNU_SOURCE
#include "mkt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <quadmath.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* ring/arithmetic.jl                                                          */
/* ------------------------------------------------------------------------- */

static inline uint64_t wmask(int W) { return W == 64 ? ~(uint64_t)0 : (((uint64_t)1 << W) - 1); }

/* signed(x) for a W-bit word, as int64 (fft.jl:60 `signed(p.coeffs[i])`) */
static inline int64_t wsigned(uint64_t x, int W) {
    return W == 64 ? (int64_t)x : (int64_t)(int32_t)(uint32_t)x;
}

/* arithmetic.jl:1-9  native(x, mask): x -= floor(x*2^-W)*2^W; x == 2^W ? 0 : trunc(x) */
uint64_t ora_native(double x, int W) {
    if (W == 32) {
        x -= floor(x * 2.3283064365386963e-10) * 4.294967296e9;
        return x == 4.294967296e9 ? 0u : (uint64_t)(uint32_t)x;
    }
    x -= floor(x * 5.421010862427522e-20) * 1.8446744073709552e19;
    return x == 1.8446744073709552e19 ? (uint64_t)0 : (uint64_t)x;
}

/* arithmetic.jl:23-27  divbits(a, bit) = (a >> bit) + ((a << (w-bit)) >> (w-1)); Julia shifts
 * by >= width give 0, so bit == 0 yields a unchanged.  Result can equal 2^(w-bit). */
uint64_t ora_divbits(uint64_t a, int bit, int W) {
    uint64_t m = wmask(W);
    a &= m;
    if (bit <= 0) return a;
    uint64_t carry = ((a << (W - bit)) & m) >> (W - 1);
    return ((a >> bit) + carry) & m;
}

/* ------------------------------------------------------------------------- */
/* ring/fft.jl                                                                 */
/* ------------------------------------------------------------------------- */

struct ora_ffter {
    int N, M, W;
    double *psi, *psiinv, *roots, *rootsinv; /* M complex each */
};

/* fft.jl:1-15 bit_reverse!(mu) on complex entries */
static void bit_reverse_c(double *mu, int n) {
    int j = 0;
    for (int i = 1; i <= n - 1; i++) {
        int bit = n >> 1;
        while (j >= bit) { j -= bit; bit >>= 1; }
        j += bit;
        if (i < j) {
            double r = mu[2 * i], im = mu[2 * i + 1];
            mu[2 * i] = mu[2 * j]; mu[2 * i + 1] = mu[2 * j + 1];
            mu[2 * j] = r; mu[2 * j + 1] = im;
        }
    }
}

/* E[t] = (cos, sin)(pi*t/N) as Float64.  The reference evaluates exp(+-im*big(pi)/N * t) in
 * 256-bit BigFloat and rounds to Float64 (fft.jl:33-41).  Here: __float128 (113 bit), which
 * rounds to the same double except where the true value is 0 and BigFloat's rounded pi leaves
 * a tiny residue: cos(RN256(pi)/2) = 0x1.452821e638d01p-257 (t = N/2).  Pinned against the
 * mpmath fixture in tests/golden/. */
static void unit_root(int t, int N, double *c, double *s) {
    if (t == 0) { *c = 1.0; *s = 0.0; return; }
    if (2 * t == N) { *c = 0x1.452821e638d01p-257; *s = 1.0; return; }
    __float128 th = M_PIq * (__float128)t / (__float128)N;
    *c = (double)cosq(th);
    *s = (double)sinq(th);
}

ora_ffter *ora_ffter_create(int N, int W) {
    ora_ffter *f = (ora_ffter *)calloc(1, sizeof *f);
    int M = N >> 1;
    f->N = N; f->M = M; f->W = W;
    f->psi = (double *)malloc(sizeof(double) * 2 * M);
    f->psiinv = (double *)malloc(sizeof(double) * 2 * M);
    f->roots = (double *)malloc(sizeof(double) * 2 * M);
    f->rootsinv = (double *)malloc(sizeof(double) * 2 * M);
    for (int j = 0; j < M; j++) {
        double c, s;
        /* fft.jl:33-34: Psi = exp(-i pi j / M), Psiinv = exp(+i pi j / M); angle pi*(2j)/N */
        unit_root(2 * j, N, &c, &s);
        f->psi[2 * j] = c; f->psi[2 * j + 1] = -s;
        f->psiinv[2 * j] = c; f->psiinv[2 * j + 1] = s;
        /* fft.jl:40-41: roots = exp(+i pi j / N), rootsinv = exp(-i pi j / N) / M */
        unit_root(j, N, &c, &s);
        f->roots[2 * j] = c; f->roots[2 * j + 1] = s;
        f->rootsinv[2 * j] = c / (double)M; f->rootsinv[2 * j + 1] = -s / (double)M;
    }
    /* signed zeros of entry 0, as Julia's exp(::Complex{BigFloat}) returns them:
     * (-im*pi/M)*0 has imaginary part -0.0, exp keeps it (Base complex.jl exp: iszero(zi) branch) */
    f->psi[1] = -0.0; f->psiinv[1] = 0.0; f->roots[1] = 0.0; f->rootsinv[1] = -0.0;
    bit_reverse_c(f->psi, M);     /* fft.jl:36 */
    bit_reverse_c(f->psiinv, M);  /* fft.jl:37 */
    return f;
}

void ora_ffter_destroy(ora_ffter *f) {
    if (!f) return;
    free(f->psi); free(f->psiinv); free(f->roots); free(f->rootsinv); free(f);
}

const double *ora_ffter_table(const ora_ffter *f, int which) {
    switch (which) { case 0: return f->psi; case 1: return f->psiinv; case 2: return f->roots; default: return f->rootsinv; }
}

/* Julia Base complex `*`: (xr*yr - xi*yi, xr*yi + xi*yr) */
#define CMUL(outr, outi, xr, xi, yr, yi) do { \
        double _a = (xr) * (yr), _b = (xi) * (yi), _c = (xr) * (yi), _d = (xi) * (yr); \
        (outr) = _a - _b; (outi) = _c + _d; } while (0)

/* fft.jl:105-155 fft!(a, Psi): Cooley-Tukey; the source unrolls by 8/4/2, the per-element
 * operations are these. */
static void ct_forward(double *a, const double *psi, int M) {
    int m = 1, k = M >> 1;
    while (k >= 1) {
        for (int i = 0; i < m; i++) {
            double wr = psi[2 * (m + i)], wi = psi[2 * (m + i) + 1];
            int j1 = 2 * i * k, j2 = j1 + k;
            for (int j = j1; j < j2; j++) {
                double tr = a[2 * j], ti = a[2 * j + 1];
                double ur, ui;
                CMUL(ur, ui, a[2 * (j + k)], a[2 * (j + k) + 1], wr, wi);
                a[2 * j] = tr + ur; a[2 * j + 1] = ti + ui;
                a[2 * (j + k)] = tr - ur; a[2 * (j + k) + 1] = ti - ui;
            }
        }
        m <<= 1; k >>= 1;
    }
}

/* fft.jl:159-209 ifft!(a, Psiinv): Gentleman-Sande (the later definition, which overrides :85-101) */
static void gs_inverse(double *a, const double *psiinv, int M) {
    int m = M >> 1, k = 1;
    while (m >= 1) {
        for (int i = 0; i < m; i++) {
            double wr = psiinv[2 * (m + i)], wi = psiinv[2 * (m + i) + 1];
            int j1 = 2 * i * k, j2 = j1 + k;
            for (int j = j1; j < j2; j++) {
                double tr = a[2 * j], ti = a[2 * j + 1];
                double ur = a[2 * (j + k)], ui = a[2 * (j + k) + 1];
                a[2 * j] = tr + ur; a[2 * j + 1] = ti + ui;
                double dr = tr - ur, di = ti - ui;
                CMUL(a[2 * (j + k)], a[2 * (j + k) + 1], dr, di, wr, wi);
            }
        }
        m >>= 1; k <<= 1;
    }
}

/* fft.jl:57-63 fftto!: t[i] = (signed(p[i]) - im*signed(p[i+M])) * roots[i]; fft!(t, Psi).
 * The subtraction happens in the integer type (Complex{IntW}), so -signed(p[i+M]) wraps. */
void ora_fft_fwd(const ora_ffter *f, const uint64_t *p, double *t) {
    int M = f->M, W = f->W;
    uint64_t msk = wmask(W);
    for (int i = 0; i < M; i++) {
        double zr = (double)wsigned(p[i] & msk, W);
        double zi = (double)wsigned(((uint64_t)0 - p[i + M]) & msk, W);
        CMUL(t[2 * i], t[2 * i + 1], zr, zi, f->roots[2 * i], f->roots[2 * i + 1]);
    }
    ct_forward(t, f->psi, M);
}

/* fft.jl:74-81 ifftto!: ifft!(t, Psiinv); t .*= rootsinv; p[i] = native(re), p[i+M] = native(-im) */
void ora_fft_inv(const ora_ffter *f, double *t, uint64_t *p) {
    int M = f->M, W = f->W;
    gs_inverse(t, f->psiinv, M);
    for (int i = 0; i < M; i++) {
        double r, im;
        CMUL(r, im, t[2 * i], t[2 * i + 1], f->rootsinv[2 * i], f->rootsinv[2 * i + 1]);
        t[2 * i] = r; t[2 * i + 1] = im;
        p[i] = ora_native(r, W);
        p[i + M] = ora_native(-im, W);
    }
}

void ora_fft_fwd_batch(const ora_ffter *f, const uint64_t *p, double *t, size_t B) {
    for (size_t b = 0; b < B; b++) ora_fft_fwd(f, p + b * (size_t)f->N, t + b * (size_t)f->N);
}
void ora_fft_inv_batch(const ora_ffter *f, double *t, uint64_t *p, size_t B) {
    for (size_t b = 0; b < B; b++) ora_fft_inv(f, t + b * (size_t)f->N, p + b * (size_t)f->N);
}

/* ring/polynomial.jl:99-113 */
void ora_tp_mul(double *res, const double *x, const double *y, int M) {
    for (int i = 0; i < M; i++) {
        double r, im;
        CMUL(r, im, x[2 * i], x[2 * i + 1], y[2 * i], y[2 * i + 1]);
        res[2 * i] = r; res[2 * i + 1] = im;
    }
}
void ora_tp_muladd(double *res, const double *x, const double *y, int M) {
    for (int i = 0; i < M; i++) {
        double r, im;
        CMUL(r, im, x[2 * i], x[2 * i + 1], y[2 * i], y[2 * i + 1]);
        res[2 * i] = res[2 * i] + r; res[2 * i + 1] = res[2 * i + 1] + im;
    }
}
void ora_tp_mulsub(double *res, const double *x, const double *y, int M) {
    for (int i = 0; i < M; i++) {
        double r, im;
        CMUL(r, im, x[2 * i], x[2 * i + 1], y[2 * i], y[2 * i + 1]);
        res[2 * i] = res[2 * i] - r; res[2 * i + 1] = res[2 * i + 1] - im;
    }
}
static void tp_add(double *res, const double *y, int M) { /* polynomial.jl:73-77 addto!(res,res,y) */
    for (int i = 0; i < 2 * M; i++) res[i] = res[i] + y[i];
}

/* scheme.jl:121-146 getmonomial: entry e (1..2N) = fft(X^e - 1) with the listed special cases */
void ora_monomial(const ora_ffter *f, int e, double *out) {
    int N = f->N;
    uint64_t m1 = wmask(f->W);
    uint64_t *tmp = (uint64_t *)calloc((size_t)N, sizeof(uint64_t));
    if (e == 2 * N) {               /* :125 zero polynomial (zerotransnativepoly, not an fft) */
        for (int i = 0; i < N; i++) out[i] = 0.0;
        free(tmp); return;
    }
    if (e < N) { tmp[0] = m1; tmp[e] = 1; }                    /* :128-134  -1 + X^e */
    else if (e == N) { tmp[0] = (m1 - 1) & m1; }               /* :136-137  -2 */
    else { tmp[0] = m1; tmp[e - N] = m1; }                     /* :138-143  -1 - X^(e-N) */
    ora_fft_fwd(f, tmp, out);
    free(tmp);
}

void ora_negacyclic_schoolbook(const uint64_t *a, const uint64_t *b, uint64_t *out, int N, int W) {
    uint64_t m = wmask(W);
    for (int i = 0; i < N; i++) out[i] = 0;
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            uint64_t p = a[i] * b[j];
            if (i + j < N) out[i + j] += p; else out[i + j - N] -= p;
        }
    for (int i = 0; i < N; i++) out[i] &= m;
}

/* ------------------------------------------------------------------------- */
/* ciphertext/gsw.jl, unienc.jl: gadget decomposition                          */
/* ------------------------------------------------------------------------- */

/* gsw.jl:42-52 decompto!(avec, a, params) -- balanced digits, out[0] most significant.
 * (gsw.jl:86-96 poly version and unienc.jl:4-18 decomptoith! compute the same digits.) */
void ora_decomp_word(uint64_t a, int l, int logB, int W, uint64_t *out) {
    uint64_t m = wmask(W), mask = ((uint64_t)1 << logB) - 1, halfB = (uint64_t)1 << (logB - 1);
    uint64_t ai = ora_divbits(a, W - l * logB, W);
    for (int i = l - 1; i >= 1; i--) {
        uint64_t d = ai & mask;
        ai >>= logB;
        ai = (ai + (d >> (logB - 1))) & m;
        d = (d - ((d & halfB) << 1)) & m;
        out[i] = d;
    }
    uint64_t d = ai & mask;
    d = (d - ((d & halfB) << 1)) & m;
    out[0] = d;
}

/* gsw.jl:34-40 unbalanceddecompto! -- digits 0..D-1, out[0] most significant */
void ora_unbalanced_decomp_word(uint64_t a, int l, int logB, int W, uint64_t *out) {
    uint64_t mask = ((uint64_t)1 << logB) - 1;
    uint64_t ai = ora_divbits(a, W - l * logB, W);
    for (int i = l - 1; i >= 0; i--) { out[i] = ai & mask; ai >>= logB; }
}

void ora_decomp_poly(const uint64_t *a, int N, int l, int logB, int W, uint64_t *out) {
    uint64_t d[64];
    for (int i = 0; i < N; i++) {
        ora_decomp_word(a[i], l, logB, W, d);
        for (int j = 0; j < l; j++) out[(size_t)j * N + i] = d[j];
    }
}

/* ------------------------------------------------------------------------- */
/* scheme object                                                               */
/* ------------------------------------------------------------------------- */

struct ora_scheme {
    ora_params p;
    ora_ffter *ff;
    int kr;          /* RLWE length used by the RGSW rotation (k for SK, 1 for KMS) */
    int kacc;        /* number of mask polys of the accumulator (= p.k) */
    int nparty;      /* 1 for SK schemes, k for MK */
    int brk_polys;   /* TransPolys per BRK entry */
    int ksk_drows;   /* D-1 or D/2 */
    double *monomial; /* [2N][M] complex; entry e-1 */
    double **brk;    /* per party: [n][brk_polys][M] complex */
    uint32_t **ksk;  /* per party */
    double **rlk_d;  /* per party: [l_uni][M] */
    double **rlk_f;  /* per party: [l_uni][2][M] */
    double **pub_b;  /* per party: [l_uni][M] */
    double *crs;     /* [l_uni][M] */
};

static int is_mk(int s) { return s == ORA_CCS || s == ORA_KMS || s == ORA_KMS_BLOCK; }
static int is_kms(int s) { return s == ORA_KMS || s == ORA_KMS_BLOCK; }
static int is_block(int s) { return s == ORA_LMSS || s == ORA_KMS_BLOCK; }

ora_scheme *ora_scheme_create(const ora_params *p) {
    ora_scheme *s = (ora_scheme *)calloc(1, sizeof *s);
    s->p = *p;
    s->ff = ora_ffter_create(p->N, p->W);
    s->kacc = p->k;
    s->nparty = is_mk(p->scheme) ? p->k : 1;
    s->kr = is_kms(p->scheme) ? 1 : p->k;
    if (p->scheme == ORA_CCS) s->brk_polys = 3 * p->l_uni;
    else s->brk_polys = (s->kr + 1) * p->l_gsw * (s->kr + 1);
    int D = 1 << p->logD;
    s->ksk_drows = is_block(p->scheme) ? D / 2 : D - 1;
    int N = p->N;
    s->monomial = (double *)malloc(sizeof(double) * (size_t)2 * N * N);
    for (int e = 1; e <= 2 * N; e++) ora_monomial(s->ff, e, s->monomial + (size_t)(e - 1) * N);
    s->brk = (double **)calloc((size_t)s->nparty, sizeof(double *));
    s->ksk = (uint32_t **)calloc((size_t)s->nparty, sizeof(uint32_t *));
    s->rlk_d = (double **)calloc((size_t)s->nparty, sizeof(double *));
    s->rlk_f = (double **)calloc((size_t)s->nparty, sizeof(double *));
    s->pub_b = (double **)calloc((size_t)s->nparty, sizeof(double *));
    return s;
}

void ora_scheme_destroy(ora_scheme *s) {
    if (!s) return;
    for (int i = 0; i < s->nparty; i++) {
        free(s->brk[i]); free(s->ksk[i]); free(s->rlk_d[i]); free(s->rlk_f[i]); free(s->pub_b[i]);
    }
    free(s->brk); free(s->ksk); free(s->rlk_d); free(s->rlk_f); free(s->pub_b);
    free(s->crs); free(s->monomial);
    ora_ffter_destroy(s->ff);
    free(s);
}

const ora_ffter *ora_scheme_ffter(const ora_scheme *s) { return s->ff; }
int ora_acc_polys(const ora_scheme *s) { return s->kacc; }

static double *transform_polys(const ora_ffter *ff, const uint64_t *in, size_t npolys) {
    size_t N = (size_t)ff->N;
    double *out = (double *)malloc(sizeof(double) * N * npolys);
    for (size_t i = 0; i < npolys; i++) ora_fft_fwd(ff, in + i * N, out + i * N);
    return out;
}

int ora_set_brk(ora_scheme *s, int party, const uint64_t *brk_int) {
    if (party < 0 || party >= s->nparty) return -1;
    free(s->brk[party]);
    s->brk[party] = transform_polys(s->ff, brk_int, (size_t)s->p.n * s->brk_polys);
    return 0;
}
int ora_set_ksk(ora_scheme *s, int party, const uint32_t *ksk) {
    if (party < 0 || party >= s->nparty) return -1;
    int kr = is_mk(s->p.scheme) ? 1 : s->p.k;
    size_t words = (size_t)kr * s->p.N * s->ksk_drows * s->p.f * (s->p.n + 1);
    free(s->ksk[party]);
    s->ksk[party] = (uint32_t *)malloc(words * sizeof(uint32_t));
    memcpy(s->ksk[party], ksk, words * sizeof(uint32_t));
    return 0;
}
int ora_set_rlk(ora_scheme *s, int party, const uint64_t *d_int, const uint64_t *f_int) {
    if (party < 0 || party >= s->nparty) return -1;
    free(s->rlk_d[party]); free(s->rlk_f[party]);
    s->rlk_d[party] = transform_polys(s->ff, d_int, (size_t)s->p.l_uni);
    s->rlk_f[party] = transform_polys(s->ff, f_int, (size_t)s->p.l_uni * 2);
    return 0;
}
int ora_set_pubkey(ora_scheme *s, int party, const uint64_t *b_int) {
    if (party < 0 || party >= s->nparty) return -1;
    free(s->pub_b[party]);
    s->pub_b[party] = transform_polys(s->ff, b_int, (size_t)s->p.l_uni);
    return 0;
}
int ora_set_crs(ora_scheme *s, const uint64_t *a_int) {
    free(s->crs);
    s->crs = transform_polys(s->ff, a_int, (size_t)s->p.l_uni);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* tfhe/gate.jl                                                                */
/* ------------------------------------------------------------------------- */

/* gate.jl:1-53; `len` = k*n+1 words, b last.  Constants are w-bit words (w = 32). */
void ora_gate_linear(int op, const uint32_t *x, const uint32_t *y, uint32_t *out, int len) {
    int nb = len - 1;
    uint32_t c;
    switch (op) {
    case ORA_NAND: c = (uint32_t)1 << 29; for (int i = 0; i < nb; i++) out[i] = 0u - x[i] - y[i]; out[nb] = c - x[nb] - y[nb]; break;         /* :1-8 */
    case ORA_AND:  c = (uint32_t)7 << 29; for (int i = 0; i < nb; i++) out[i] = x[i] + y[i];      out[nb] = c + x[nb] + y[nb]; break;         /* :10-17 */
    case ORA_OR:   c = (uint32_t)1 << 29; for (int i = 0; i < nb; i++) out[i] = x[i] + y[i];      out[nb] = c + x[nb] + y[nb]; break;         /* :19-26 */
    case ORA_XOR:  c = (uint32_t)1 << 30; for (int i = 0; i < nb; i++) out[i] = 2u * (x[i] + y[i]); out[nb] = c + 2u * (x[nb] + y[nb]); break; /* :28-35 */
    case ORA_XNOR: c = (uint32_t)3 << 30; for (int i = 0; i < nb; i++) out[i] = (0u - 2u) * (x[i] + y[i]); out[nb] = c - 2u * (x[nb] + y[nb]); break; /* :37-44 */
    default:       c = (uint32_t)7 << 29; for (int i = 0; i < nb; i++) out[i] = 0u - x[i] - y[i]; out[nb] = c - x[nb] - y[nb]; break;         /* NOR :46-53 */
    }
}
void ora_not(uint32_t *x, int len) { for (int i = 0; i < len; i++) x[i] = 0u - x[i]; } /* gate.jl:55-58 */

/* ------------------------------------------------------------------------- */
/* tfhe/bootstrapping.jl                                                       */
/* ------------------------------------------------------------------------- */

/* bootstrapping.jl:8-9 */
void ora_modswitch(const ora_scheme *s, const uint32_t *lwe, uint32_t *atilde, uint32_t *btilde) {
    int logN = __builtin_ctz((unsigned)s->p.N);
    int len = s->nparty * s->p.n;
    if (!is_mk(s->p.scheme)) len = s->p.n;
    for (int i = 0; i < len; i++) atilde[i] = (uint32_t)ora_divbits(lwe[i], 32 - logN - 1, 32);
    *btilde = (uint32_t)ora_divbits(lwe[len], 32 - logN - 1, 32);
}

/* bootstrapping.jl:11-23; acc = [b | a_0 .. a_{k-1}] each N ring words */
void ora_testvector(const ora_scheme *s, uint32_t btilde, uint64_t *acc) {
    int N = s->p.N, W = s->p.W;
    uint64_t m = wmask(W);
    uint64_t e = (uint64_t)1 << (W - 3);
    uint64_t tb = btilde;
    if (tb <= (uint64_t)N) {
        for (int i = 0; i < N; i++) acc[i] = ((uint64_t)i < tb) ? e : ((0 - e) & m);
    } else {
        tb -= (uint64_t)N;
        for (int i = 0; i < N; i++) acc[i] = ((uint64_t)i < tb) ? ((0 - e) & m) : e;
    }
    memset(acc + N, 0, sizeof(uint64_t) * (size_t)N * s->kacc);
}

static inline const double *mono(const ora_scheme *s, uint32_t e) { return s->monomial + (size_t)(e - 1) * s->p.N; }

/* One RGSW external product accumulation, shared by CGGI/LMSS/KMS phase 1:
 *   tacc(b, a_0..) = sum_j tb[j] * basketb.stack[j] + sum_c sum_j ta[c][j] * basketa[c].stack[j]
 * in exactly that order (bootstrapping.jl:62-68, :146-154, :418-432, :639-646).
 * dig: [(kr+1)][l][M] transforms of the digits (b first); brk_e: [(kr+1)*l rows][kr+1][M]. */
static void rgsw_mac(double *tacc, const double *dig, const double *brk_e, int kr, int l, int M) {
    size_t tp = (size_t)2 * M;
    memset(tacc, 0, sizeof(double) * tp * (kr + 1));
    for (int c = 0; c <= kr; c++)           /* c = 0: b digits, then a_0, a_1 ... */
        for (int j = 0; j < l; j++) {
            const double *x = dig + ((size_t)c * l + j) * tp;
            const double *row = brk_e + ((size_t)c * l + j) * (kr + 1) * tp;
            for (int q = 0; q <= kr; q++) ora_tp_muladd(tacc + q * tp, x, row + q * tp, M);
        }
}

/* decompose (kr+1) polys and transform every digit poly: bootstrapping.jl:50-59 */
static void decomp_fft(const ora_scheme *s, const uint64_t *acc, int npolys, int l, int logB,
                       uint64_t *digbuf /*[l][N]*/, double *dig /*[npolys][l][M]*/) {
    int N = s->p.N;
    for (int c = 0; c < npolys; c++) {
        ora_decomp_poly(acc + (size_t)c * N, N, l, logB, s->p.W, digbuf);
        for (int j = 0; j < l; j++) ora_fft_fwd(s->ff, digbuf + (size_t)j * N, dig + ((size_t)c * l + j) * N);
    }
}

static void acc_add(uint64_t *acc, const uint64_t *acc2, size_t words, int W) { /* polynomial.jl:18-22 */
    uint64_t m = wmask(W);
    for (size_t i = 0; i < words; i++) acc[i] = (acc[i] + acc2[i]) & m;
}

/* bootstrapping.jl:32-76 blindrotate!(::CGGI) */
static void blindrotate_cggi(const ora_scheme *s, const uint32_t *atilde, uint64_t *acc) {
    int N = s->p.N, M = N / 2, kr = s->kr, l = s->p.l_gsw;
    size_t tp = (size_t)N;
    uint64_t *digbuf = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)l * N);
    uint64_t *acc2 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(kr + 1) * N);
    double *dig = (double *)malloc(sizeof(double) * tp * (kr + 1) * l);
    double *tacc = (double *)malloc(sizeof(double) * tp * (kr + 1));
    for (int idx = 0; idx < s->p.n; idx++) {
        if (atilde[idx] == 0) continue;                                            /* :48 */
        decomp_fft(s, acc, kr + 1, l, s->p.logB_gsw, digbuf, dig);                  /* :50-59 */
        rgsw_mac(tacc, dig, s->brk[0] + (size_t)idx * s->brk_polys * tp, kr, l, M); /* :62-68 */
        for (int q = 0; q <= kr; q++) {
            ora_tp_mul(tacc + q * tp, mono(s, atilde[idx]), tacc + q * tp, M);      /* :71 */
            ora_fft_inv(s->ff, tacc + q * tp, acc2 + (size_t)q * N);                /* :72 */
        }
        acc_add(acc, acc2, (size_t)(kr + 1) * N, s->p.W);                           /* :73 */
    }
    free(digbuf); free(acc2); free(dig); free(tacc);
}

/* bootstrapping.jl:114-165 blindrotate!(::LMSS) */
static void blindrotate_lmss(const ora_scheme *s, const uint32_t *atilde, uint64_t *acc) {
    int N = s->p.N, M = N / 2, kr = s->kr, l = s->p.l_gsw;
    size_t tp = (size_t)N;
    uint64_t *digbuf = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)l * N);
    uint64_t *acc2 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(kr + 1) * N);
    double *dig = (double *)malloc(sizeof(double) * tp * (kr + 1) * l);
    double *tacc = (double *)malloc(sizeof(double) * tp * (kr + 1));
    double *tacc2 = (double *)malloc(sizeof(double) * tp * (kr + 1));
    for (int idx1 = 0; idx1 < s->p.blk_d; idx1++) {
        decomp_fft(s, acc, kr + 1, l, s->p.logB_gsw, digbuf, dig);                  /* :131-140 */
        memset(tacc2, 0, sizeof(double) * tp * (kr + 1));                           /* :142 */
        for (int idx2 = 0; idx2 < s->p.blk_len; idx2++) {
            int idx = idx1 * s->p.blk_len + idx2;
            if (atilde[idx] == 0) continue;                                        /* :145 */
            rgsw_mac(tacc, dig, s->brk[0] + (size_t)idx * s->brk_polys * tp, kr, l, M); /* :146-154 */
            for (int q = 0; q <= kr; q++)
                ora_tp_muladd(tacc2 + q * tp, mono(s, atilde[idx]), tacc + q * tp, M); /* :157 */
        }
        for (int q = 0; q <= kr; q++) ora_fft_inv(s->ff, tacc2 + q * tp, acc2 + (size_t)q * N); /* :162 */
        acc_add(acc, acc2, (size_t)(kr + 1) * N, s->p.W);                           /* :163 */
    }
    free(digbuf); free(acc2); free(dig); free(tacc); free(tacc2);
}

/* bootstrapping.jl:234-328 blindrotate!(::CCS).  Mask polys beyond the current party are exactly
 * zero in tacc (initialise! then never touched), their inverse transform is exactly 0, so they
 * are skipped here instead of being transformed (:322-324 does transform them). */
static void blindrotate_ccs(const ora_scheme *s, const uint32_t *atilde, uint64_t *acc) {
    int N = s->p.N, M = N / 2, k = s->p.k, n = s->p.n, l = s->p.l_uni, logB = s->p.logB_uni, W = s->p.W;
    size_t tp = (size_t)N;
    uint64_t *digbuf = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)l * N);
    uint64_t *acc2 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(k + 1) * N);
    uint64_t *v = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(k + 1) * N); /* v0, v[0..k) */
    double *dig = (double *)malloc(sizeof(double) * tp * (k + 1) * l);   /* tbvec, tavec */
    double *vdig = (double *)malloc(sizeof(double) * tp * (k + 1) * l);  /* tv0vec, tvvec */
    double *tacc = (double *)malloc(sizeof(double) * tp * (k + 1));
    double *tv = (double *)malloc(sizeof(double) * tp * (k + 1));        /* tv0, tv[0..k) */
    for (int idx = 0; idx < k; idx++) {
        int np = idx + 1; /* number of active mask polys ("idx" in the 1-based source) */
        const double *uni_all = s->brk[idx];
        for (int i = 0; i < n; i++) {
            uint32_t at = atilde[(size_t)idx * n + i];
            if (at == 0) continue;                                                 /* :261 */
            const double *uni = uni_all + (size_t)i * s->brk_polys * tp; /* d[l], then f[j].b,f[j].a */
            const double *ud = uni, *uf = uni + (size_t)l * tp;
            decomp_fft(s, acc, 1 + np, l, logB, digbuf, dig);                      /* :264-275 */
            memset(tacc, 0, sizeof(double) * tp * (k + 1));                        /* :278 */
            for (int j = 0; j < l; j++) ora_tp_muladd(tacc, dig + (size_t)j * tp, ud + (size_t)j * tp, M); /* :279-281 */
            for (int j1 = 0; j1 < np; j1++)
                for (int j2 = 0; j2 < l; j2++)
                    ora_tp_muladd(tacc + (size_t)(1 + j1) * tp, dig + ((size_t)(1 + j1) * l + j2) * tp, ud + (size_t)j2 * tp, M); /* :282-284 */
            memset(tv, 0, sizeof(double) * tp * (k + 1));                          /* :287-288 */
            for (int j = 0; j < l; j++) ora_tp_mulsub(tv, dig + (size_t)j * tp, s->crs + (size_t)j * tp, M); /* :289-291 */
            for (int j1 = 0; j1 < np; j1++)
                for (int j2 = 0; j2 < l; j2++)
                    ora_tp_muladd(tv + (size_t)(1 + j1) * tp, dig + ((size_t)(1 + j1) * l + j2) * tp,
                                  s->pub_b[j1] + (size_t)j2 * tp, M);              /* :292-294 */
            for (int q = 0; q <= np; q++) ora_fft_inv(s->ff, tv + (size_t)q * tp, v + (size_t)q * N); /* :297-300 */
            decomp_fft(s, v, 1 + np, l, logB, digbuf, vdig);                       /* :303-310 */
            double *ta_idx = tacc + (size_t)(1 + idx) * tp;
            for (int j = 0; j < l; j++) {                                          /* :313-316 */
                ora_tp_muladd(tacc, vdig + (size_t)j * tp, uf + (size_t)(2 * j) * tp, M);
                ora_tp_muladd(ta_idx, vdig + (size_t)j * tp, uf + (size_t)(2 * j + 1) * tp, M);
            }
            for (int j1 = 0; j1 < np; j1++)                                        /* :317-320 */
                for (int j2 = 0; j2 < l; j2++) {
                    const double *x = vdig + ((size_t)(1 + j1) * l + j2) * tp;
                    ora_tp_muladd(tacc, x, uf + (size_t)(2 * j2) * tp, M);
                    ora_tp_muladd(ta_idx, x, uf + (size_t)(2 * j2 + 1) * tp, M);
                }
            for (int q = 0; q <= np; q++) {                                        /* :322-323 */
                ora_tp_mul(tacc + (size_t)q * tp, mono(s, at), tacc + (size_t)q * tp, M);
                ora_fft_inv(s->ff, tacc + (size_t)q * tp, acc2 + (size_t)q * N);
            }
            acc_add(acc, acc2, (size_t)(1 + np) * N, W);                           /* :324 */
        }
    }
    free(digbuf); free(acc2); free(v); free(dig); free(vdig); free(tacc); free(tv);
}

/* bootstrapping.jl:389-443 phase_1 (BootKey_KMS) and :599-659 (BootKey_KMS_block) */
int ora_kms_phase1(const ora_scheme *s, int party, const uint32_t *at, double *levkey) {
    int N = s->p.N, M = N / 2, l = s->p.l_gsw, logB = s->p.logB_gsw, W = s->p.W;
    size_t tp = (size_t)N;
    int iter = party == 0 ? 1 : s->p.l_lev;                                        /* :400 */
    uint64_t *acc = (uint64_t *)calloc((size_t)iter * 2 * N, sizeof(uint64_t));
    uint64_t *acc2 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)iter * 2 * N);
    uint64_t *digbuf = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)l * N);
    double *dig = (double *)malloc(sizeof(double) * tp * 2 * l * iter);
    double *tacc = (double *)malloc(sizeof(double) * tp * 2 * iter);
    double *tacc2 = (double *)malloc(sizeof(double) * tp * 2 * iter);
    for (int r = 0; r < iter; r++)                                                 /* :403-406 gvec[r] = 2^(W-(r+1)logB_lev) */
        acc[(size_t)r * 2 * N] = (uint64_t)1 << (W - (r + 1) * s->p.logB_lev);
    const double *brk = s->brk[party];
    if (s->p.scheme == ORA_KMS) {
        for (int idx = 0; idx < s->p.n; idx++) {
            if (at[idx] == 0) continue;                                            /* :413 */
            for (int r = 0; r < iter; r++) {
                decomp_fft(s, acc + (size_t)r * 2 * N, 2, l, logB, digbuf, dig);   /* :415-425 */
                rgsw_mac(tacc + (size_t)r * 2 * tp, dig, brk + (size_t)idx * s->brk_polys * tp, 1, l, M); /* :418,:427-432 */
            }
            for (int q = 0; q < 2 * iter; q++) {
                ora_tp_mul(tacc + (size_t)q * tp, mono(s, at[idx]), tacc + (size_t)q * tp, M); /* :435 */
                ora_fft_inv(s->ff, tacc + (size_t)q * tp, acc2 + (size_t)q * N);   /* :436 */
            }
            acc_add(acc, acc2, (size_t)iter * 2 * N, W);                           /* :437 */
        }
    } else {
        for (int idx1 = 0; idx1 < s->p.blk_d; idx1++) {                            /* :623 */
            for (int r = 0; r < iter; r++) {
                decomp_fft(s, acc + (size_t)r * 2 * N, 2, l, logB, digbuf, dig + (size_t)r * 2 * l * tp); /* :625-633 */
                double *t2 = tacc2 + (size_t)r * 2 * tp;
                memset(t2, 0, sizeof(double) * 2 * tp);                            /* :635 */
                for (int idx2 = 0; idx2 < s->p.blk_len; idx2++) {
                    int idx = idx1 * s->p.blk_len + idx2;
                    if (at[idx] == 0) continue;                                    /* :638 */
                    double *t1 = tacc + (size_t)r * 2 * tp;
                    rgsw_mac(t1, dig + (size_t)r * 2 * l * tp, brk + (size_t)idx * s->brk_polys * tp, 1, l, M); /* :639-646 */
                    ora_tp_muladd(t2, mono(s, at[idx]), t1, M);                    /* :648 */
                    ora_tp_muladd(t2 + tp, mono(s, at[idx]), t1 + tp, M);
                }
            }
            for (int q = 0; q < 2 * iter; q++) ora_fft_inv(s->ff, tacc2 + (size_t)q * tp, acc2 + (size_t)q * N); /* :653 */
            acc_add(acc, acc2, (size_t)iter * 2 * N, W);                           /* :654 */
        }
    }
    for (int q = 0; q < 2 * iter; q++) ora_fft_fwd(s->ff, acc + (size_t)q * N, levkey + (size_t)q * tp); /* :441 / :657 */
    free(acc); free(acc2); free(digbuf); free(dig); free(tacc); free(tacc2);
    return iter;
}

/* bootstrapping.jl:448-558 phase_2! */
void ora_kms_phase2(const ora_scheme *s, double *const *levkey, uint64_t *acc) {
    int N = s->p.N, M = N / 2, k = s->p.k, W = s->p.W;
    int ll = s->p.l_lev, lbl = s->p.logB_lev, lu = s->p.l_uni, lbu = s->p.logB_uni;
    int maxl = ll > lu ? ll : lu;
    size_t tp = (size_t)N;
    uint64_t *digbuf = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)maxl * N);
    uint64_t *y = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(k + 1) * N);
    uint64_t *v = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)N);
    double *dig = (double *)malloc(sizeof(double) * tp * (k + 1) * maxl);
    double *vdig = (double *)malloc(sizeof(double) * tp * lu);
    double *tx = (double *)malloc(sizeof(double) * tp * (k + 1));
    double *ty = (double *)malloc(sizeof(double) * tp * (k + 1));
    double *tv = (double *)malloc(sizeof(double) * tp);
    for (int idx = 0; idx < k; idx++) {              /* `idx` of the source minus one; idx = #earlier parties */
        const double *lk = levkey[idx];              /* [iter][2][M]: stack[r].b, stack[r].a */
        int iter = idx == 0 ? 1 : ll;                                              /* :481 */
        decomp_fft(s, acc, 1 + idx, ll, lbl, digbuf, dig);                          /* :470-479 */
        memset(tx, 0, sizeof(double) * tp * (k + 1));                              /* :484 */
        for (int i = 0; i < iter; i++) ora_tp_muladd(tx, dig + (size_t)i * tp, lk + (size_t)(2 * i) * tp, M); /* :485-487 */
        for (int i = 0; i < idx; i++)
            for (int j = 0; j < iter; j++)
                ora_tp_muladd(tx + (size_t)(1 + i) * tp, dig + ((size_t)(1 + i) * ll + j) * tp, lk + (size_t)(2 * j) * tp, M); /* :488-490 */
        memset(ty, 0, sizeof(double) * tp * (k + 1));                              /* :493 */
        for (int i = 0; i < iter; i++) ora_tp_muladd(ty, dig + (size_t)i * tp, lk + (size_t)(2 * i + 1) * tp, M); /* :494-496 */
        for (int i = 0; i < idx; i++)
            for (int j = 0; j < iter; j++)
                ora_tp_muladd(ty + (size_t)(1 + i) * tp, dig + ((size_t)(1 + i) * ll + j) * tp, lk + (size_t)(2 * j + 1) * tp, M); /* :497-499 */
        for (int q = 0; q <= idx; q++) ora_fft_inv(s->ff, ty + (size_t)q * tp, y + (size_t)q * N); /* :501-504 */
        decomp_fft(s, y, 1 + idx, lu, lbu, digbuf, dig);                            /* :508-517 */
        memset(ty, 0, sizeof(double) * tp * (k + 1));                              /* :520 */
        for (int i = 0; i < lu; i++) ora_tp_muladd(ty, dig + (size_t)i * tp, s->rlk_d[idx] + (size_t)i * tp, M); /* :521-523 */
        for (int i = 0; i < idx; i++)
            for (int j = 0; j < lu; j++)
                ora_tp_muladd(ty + (size_t)(1 + i) * tp, dig + ((size_t)(1 + i) * lu + j) * tp, s->rlk_d[idx] + (size_t)j * tp, M); /* :524-526 */
        memset(tv, 0, sizeof(double) * tp);                                        /* :529 */
        for (int i = 0; i < lu; i++) ora_tp_mulsub(tv, dig + (size_t)i * tp, s->crs + (size_t)i * tp, M); /* :530-532 */
        for (int i = 0; i < idx; i++)
            for (int j = 0; j < lu; j++)
                ora_tp_muladd(tv, dig + ((size_t)(1 + i) * lu + j) * tp, s->pub_b[i] + (size_t)j * tp, M); /* :533-535 */
        ora_fft_inv(s->ff, tv, v);                                                 /* :538 */
        decomp_fft(s, v, 1, lu, lbu, digbuf, vdig);                                 /* :541-544 */
        for (int i = 0; i < lu; i++) {                                             /* :547-550 */
            ora_tp_muladd(ty, vdig + (size_t)i * tp, s->rlk_f[idx] + (size_t)(2 * i) * tp, M);
            ora_tp_muladd(ty + (size_t)(1 + idx) * tp, vdig + (size_t)i * tp, s->rlk_f[idx] + (size_t)(2 * i + 1) * tp, M);
        }
        for (int q = 0; q <= k; q++) tp_add(tx + (size_t)q * tp, ty + (size_t)q * tp, M); /* :553 */
        /* :556 ifftto!(acc, tx): polys beyond idx+1 are exactly zero -> 0 */
        for (int q = 0; q <= idx + 1 && q <= k; q++) ora_fft_inv(s->ff, tx + (size_t)q * tp, acc + (size_t)q * N);
        for (int q = idx + 2; q <= k; q++) memset(acc + (size_t)q * N, 0, sizeof(uint64_t) * N);
    }
    (void)W;
    free(digbuf); free(y); free(v); free(dig); free(vdig); free(tx); free(ty); free(tv);
}

/* bootstrapping.jl:369-384 blindrotate!(::KMSScheme) */
static void blindrotate_kms(const ora_scheme *s, const uint32_t *atilde, uint64_t *acc) {
    int k = s->p.k, n = s->p.n;
    double **lev = (double **)calloc((size_t)k, sizeof(double *));
    for (int i = 0; i < k; i++) {                                                  /* :376-378 */
        lev[i] = (double *)malloc(sizeof(double) * (size_t)s->p.N * 2 * s->p.l_lev);
        ora_kms_phase1(s, i, atilde + (size_t)i * n, lev[i]);
    }
    ora_kms_phase2(s, lev, acc);                                                   /* :381 */
    for (int i = 0; i < k; i++) free(lev[i]);
    free(lev);
}

void ora_blindrotate(const ora_scheme *s, const uint32_t *atilde, uint64_t *acc) {
    switch (s->p.scheme) {
    case ORA_CGGI: blindrotate_cggi(s, atilde, acc); break;
    case ORA_LMSS: blindrotate_lmss(s, atilde, acc); break;
    case ORA_CCS:  blindrotate_ccs(s, atilde, acc); break;
    default:       blindrotate_kms(s, atilde, acc); break;
    }
}

/* key-switch row: ksk[party][c][j][d][t] -> (n+1) words */
static inline const uint32_t *ksk_row(const ora_scheme *s, int party, int c, int j, int d, int t) {
    size_t n1 = (size_t)s->p.n + 1;
    size_t off = ((((size_t)c * s->p.N + j) * s->ksk_drows + d) * s->p.f + t) * n1;
    return s->ksk[party] + off;
}

/* extracted mask word j of ring poly a (bootstrapping.jl:91,:99): c_0 = a[0], c_j = -a[N-j];
 * KMS first truncates the ring word to the LWE word: T(x >> (W-32)) (:575,:583) */
static inline uint32_t extract_word(const ora_scheme *s, const uint64_t *a, int j) {
    int sh = s->p.W - 32;
    if (j == 0) return (uint32_t)(a[0] >> sh);
    return 0u - (uint32_t)(a[s->p.N - j] >> sh);
}

/* bootstrapping.jl:81-109 (CGGI), :333-364 (CCS), :564-594 (KMS): unbalanced digits, add rows */
static void keyswitch_unbalanced(const ora_scheme *s, const uint64_t *acc, uint32_t *out) {
    int n = s->p.n, N = s->p.N, f = s->p.f, logD = s->p.logD;
    int mk = is_mk(s->p.scheme);
    int total = (mk ? s->p.k : 1) * n;
    memset(out, 0, sizeof(uint32_t) * (size_t)(total + 1));
    out[total] = (uint32_t)(acc[0] >> (s->p.W - 32));                              /* :86 / :569 */
    uint64_t dg[32];
    for (int i = 0; i < s->p.k; i++) {
        const uint64_t *a = acc + (size_t)(1 + i) * N;
        uint32_t *oa = mk ? out + (size_t)i * n : out;   /* MK: party block (:362,:592) */
        for (int j = 0; j < N; j++) {
            ora_unbalanced_decomp_word(extract_word(s, a, j), f, logD, 32, dg);
            for (int t = 0; t < f; t++) {
                if (dg[t] == 0) continue;
                const uint32_t *row = mk ? ksk_row(s, i, 0, j, (int)dg[t] - 1, t) : ksk_row(s, 0, i, j, (int)dg[t] - 1, t);
                for (int q = 0; q < n; q++) oa[q] += row[q];
                out[total] += row[n];
            }
        }
    }
}

/* balanced digit add/sub of one extracted word (bootstrapping.jl:194-201, :682-689) */
static void ks_balanced_word(const ora_scheme *s, int party, int c, int j, uint32_t w, uint32_t *oa, uint32_t *ob) {
    int n = s->p.n, f = s->p.f;
    uint64_t dg[32];
    ora_decomp_word(w, f, s->p.logD, 32, dg);
    for (int t = 0; t < f; t++) {
        int32_t d = (int32_t)(uint32_t)dg[t];
        if (d > 0) {
            const uint32_t *row = ksk_row(s, party, c, j, d - 1, t);
            for (int q = 0; q < n; q++) oa[q] += row[q];
            *ob += row[n];
        } else if (d != 0) {
            const uint32_t *row = ksk_row(s, party, c, j, -d - 1, t);
            for (int q = 0; q < n; q++) oa[q] -= row[q];
            *ob -= row[n];
        }
    }
}

/* bootstrapping.jl:170-229 keyswitch!(::LMSS) */
static void keyswitch_lmss(const ora_scheme *s, const uint64_t *acc, uint32_t *out) {
    int n = s->p.n, N = s->p.N;
    memset(out, 0, sizeof(uint32_t) * (size_t)(n + 1));
    out[n] = (uint32_t)acc[0];                                                     /* :174 */
    int current = 0; /* 0-based `current - 1` */
    for (int i = 0; i < s->p.k; i++) {
        const uint64_t *a = acc + (size_t)(1 + i) * N;
        if (current + N <= n) {                                                    /* :179-186  (1-based: current+N <= n with current one larger) */
            for (int j = 0; j < N; j++) out[current + j] = extract_word(s, a, j);
            current += N;
        } else if (current < n) {                                                  /* :187-204 */
            int cnt = n - current;                 /* words copied: j = 0 .. cnt-1 */
            for (int j = 0; j < cnt; j++) out[current + j] = extract_word(s, a, j);
            for (int j = cnt; j < N; j++) ks_balanced_word(s, 0, i, j, extract_word(s, a, j), out, &out[n]);
            current = n;
        } else {                                                                   /* :205-225 */
            for (int j = 0; j < N; j++) ks_balanced_word(s, 0, i, j, extract_word(s, a, j), out, &out[n]);
        }
    }
}

/* bootstrapping.jl:664-695 keyswitch!(::KMS_block) */
static void keyswitch_kms_block(const ora_scheme *s, const uint64_t *acc, uint32_t *out) {
    int n = s->p.n, N = s->p.N, k = s->p.k;
    int total = k * n;
    memset(out, 0, sizeof(uint32_t) * (size_t)(total + 1));
    out[total] = (uint32_t)(acc[0] >> (s->p.W - 32));                              /* :669 */
    for (int i = 0; i < k; i++) {
        const uint64_t *a = acc + (size_t)(1 + i) * N;
        uint32_t *oa = out + (size_t)i * n;
        for (int j = 0; j < n; j++) oa[j] = extract_word(s, a, j);                 /* :676-679 */
        for (int j = n; j < N; j++) ks_balanced_word(s, i, 0, j, extract_word(s, a, j), oa, &out[total]); /* :681-690 */
    }
}

void ora_keyswitch(const ora_scheme *s, const uint64_t *acc, uint32_t *out) {
    switch (s->p.scheme) {
    case ORA_LMSS: keyswitch_lmss(s, acc, out); break;
    case ORA_KMS_BLOCK: keyswitch_kms_block(s, acc, out); break;
    default: keyswitch_unbalanced(s, acc, out); break;
    }
}

/* bootstrapping.jl:4-27 */
void ora_bootstrap(const ora_scheme *s, uint32_t *lwe) {
    int total = (is_mk(s->p.scheme) ? s->p.k : 1) * s->p.n;
    uint32_t *at = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)total);
    uint64_t *acc = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(s->kacc + 1) * s->p.N);
    uint32_t bt;
    ora_modswitch(s, lwe, at, &bt);
    ora_testvector(s, bt, acc);
    ora_blindrotate(s, at, acc);
    ora_keyswitch(s, acc, lwe);
    free(at); free(acc);
}

void ora_gate(const ora_scheme *s, int op, const uint32_t *x, const uint32_t *y, uint32_t *out) {
    int total = (is_mk(s->p.scheme) ? s->p.k : 1) * s->p.n;
    ora_gate_linear(op, x, y, out, total + 1);
    ora_bootstrap(s, out);
}

/* ---- batch driver (CPU baseline): gates are independent, one gate per worker ---- */
typedef struct { const ora_scheme *s; int op; const uint32_t *x, *y; uint32_t *out; size_t B, len; int tid, nth; } gate_job;
static void *gate_worker(void *arg) {
    gate_job *j = (gate_job *)arg;
    for (size_t b = (size_t)j->tid; b < j->B; b += (size_t)j->nth)
        ora_gate(j->s, j->op, j->x + b * j->len, j->y + b * j->len, j->out + b * j->len);
    return NULL;
}
void ora_gate_batch(const ora_scheme *s, int op, const uint32_t *x, const uint32_t *y, uint32_t *out, size_t B, int threads) {
    size_t len = (size_t)(is_mk(s->p.scheme) ? s->p.k : 1) * s->p.n + 1;
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    gate_job *jobs = (gate_job *)malloc(sizeof(gate_job) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        jobs[t] = (gate_job){ s, op, x, y, out, B, len, t, threads };
        pthread_create(&th[t], NULL, gate_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
}
