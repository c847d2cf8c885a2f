// Host-side generation of the F64REF twiddle tables -- the engine's FFTransformer
// (reference: src/ring/fft.jl:18-45).  The reference evaluates exp(+-i*big(pi)*j/N) in 256-bit
// BigFloat and rounds to Float64.  Here the same values are produced with a self-contained
// 576-bit fixed-point sin/cos (no MPFR / libquadmath dependency in the product library):
//   theta_t = RN256(RN256(pi) * t) / N      (the /N is an exact power-of-two scaling)
//   E[t]    = RN53(cos theta_t, sin theta_t)
// and  Psi_nat[j] = conj E[2j], Psiinv_nat[j] = E[2j], roots[j] = E[j], rootsinv[j] = conj E[j] / M,
// followed by the in-place bit-reversal permutation of fft.jl:1-15 on Psi and Psiinv.
// tests/test_twiddles.py pins the result bit-for-bit against an mpmath (256-bit) fixture.
#include "host_internal.h"

#include <cmath>
#include <cstring>

namespace mkt {
namespace {

constexpr int FL = 9;         // fraction limbs (576 bits)
constexpr int NL = FL + 1;    // + one integer limb

struct Fix {                  // unsigned fixed point, little-endian limbs, value = sum l[i] * 2^(64*(i-FL))
    uint64_t l[NL];
    Fix() { std::memset(l, 0, sizeof l); }
    bool is_zero() const { for (int i = 0; i < NL; i++) if (l[i]) return false; return true; }
};

int cmp(const Fix &a, const Fix &b) {
    for (int i = NL - 1; i >= 0; i--) { if (a.l[i] < b.l[i]) return -1; if (a.l[i] > b.l[i]) return 1; }
    return 0;
}
Fix add(const Fix &a, const Fix &b) {
    Fix r; unsigned __int128 c = 0;
    for (int i = 0; i < NL; i++) { c += (unsigned __int128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    return r;
}
Fix sub(const Fix &a, const Fix &b) {   // a >= b
    Fix r; __int128 c = 0;
    for (int i = 0; i < NL; i++) {
        __int128 d = (__int128)a.l[i] - b.l[i] + c;
        r.l[i] = (uint64_t)d; c = d >> 64;  // arithmetic shift: 0 or -1
    }
    return r;
}
Fix mul(const Fix &a, const Fix &b) {   // truncating product
    uint64_t w[2 * NL + 1]; std::memset(w, 0, sizeof w);
    for (int i = 0; i < NL; i++) {
        unsigned __int128 c = 0;
        for (int j = 0; j < NL; j++) {
            c += (unsigned __int128)a.l[i] * b.l[j] + w[i + j];
            w[i + j] = (uint64_t)c; c >>= 64;
        }
        w[i + NL] += (uint64_t)c;
    }
    Fix r; for (int i = 0; i < NL; i++) r.l[i] = w[i + FL];
    return r;
}
Fix div_small(const Fix &a, uint64_t d) {
    Fix r; unsigned __int128 rem = 0;
    for (int i = NL - 1; i >= 0; i--) {
        unsigned __int128 cur = (rem << 64) | a.l[i];
        r.l[i] = (uint64_t)(cur / d); rem = cur % d;
    }
    return r;
}

// RN256(pi) = P * 2^-254, P a 256-bit integer (top bit set); limbs little-endian.
const uint64_t PI256[4] = { 0x020BBEA63B139B22ull, 0x29024E088A67CC74ull, 0xC4C6628B80DC1CD1ull, 0xC90FDAA22168C234ull };

// theta = RN256(P * t) * 2^-254 / N as fixed point
Fix theta_of(int t, int logN) {
    // 320-bit product
    uint64_t w[5] = {0, 0, 0, 0, 0};
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) { c += (unsigned __int128)PI256[i] * (uint64_t)t; w[i] = (uint64_t)c; c >>= 64; }
    w[4] = (uint64_t)c;
    int L = 0;  // bit length
    for (int i = 4; i >= 0; i--) if (w[i]) { L = 64 * i + 64 - __builtin_clzll(w[i]); break; }
    int s = L > 256 ? L - 256 : 0;
    if (s > 0) {  // round to nearest even at bit s
        auto bit = [&](int b) { return (w[b >> 6] >> (b & 63)) & 1; };
        bool half = bit(s - 1), sticky = false, odd = bit(s);
        for (int b = 0; b < s - 1; b++) sticky |= bit(b);
        // shift right by s (s < 64)
        for (int i = 0; i < 5; i++) w[i] = (w[i] >> s) | (i + 1 < 5 && s ? (w[i + 1] << (64 - s)) : 0);
        if (half && (sticky || odd)) { for (int i = 0; i < 5; i++) if (++w[i]) break; }
        // mantissa overflow to 2^256 keeps the value exact (trailing zeros), no renormalisation needed
    }
    // value = w * 2^(s - 254 - logN); place into Fix: bit position 0 of w goes to fixed-point bit (64*FL + s - 254 - logN)
    int pos = 64 * FL + s - 254 - logN;
    Fix r;
    for (int i = 0; i < 5; i++) {
        int b = pos + 64 * i; int li = b >> 6, sh = b & 63;
        if (li < NL) r.l[li] |= w[i] << sh;
        if (sh && li + 1 < NL) r.l[li + 1] |= w[i] >> (64 - sh);
    }
    return r;
}

// round a non-negative fixed-point magnitude to double (nearest even)
double to_double(const Fix &a, bool neg) {
    int top = -1;
    for (int i = NL - 1; i >= 0 && top < 0; i--) if (a.l[i]) top = 64 * i + 63 - __builtin_clzll(a.l[i]);
    if (top < 0) return neg ? -0.0 : 0.0;
    auto bit = [&](int b) -> uint64_t { return b < 0 ? 0 : (a.l[b >> 6] >> (b & 63)) & 1; };
    uint64_t m = 0;
    for (int b = top; b > top - 53; b--) m = (m << 1) | bit(b);
    bool half = bit(top - 53), sticky = false;
    for (int b = top - 54; b >= 0 && !sticky; b--) sticky |= bit(b) != 0;
    if (half && (sticky || (m & 1))) m++;
    double d = std::ldexp((double)m, top - 52 - 64 * FL);
    return neg ? -d : d;
}

// sin and cos of theta in [0, pi) by Taylor series with separate positive / negative sums
void sincos_fix(const Fix &th, double *s_out, double *c_out) {
    Fix x2 = mul(th, th);
    Fix cpos, cneg, spos, sneg;
    Fix term; term.l[FL] = 1;          // x^0/0! = 1
    Fix sterm = th;                     // x^1/1!
    cpos = term; spos = sterm;
    for (int n = 1; n < 200; n++) {
        term = div_small(mul(term, x2), (uint64_t)(2 * n - 1) * (uint64_t)(2 * n));
        sterm = div_small(mul(sterm, x2), (uint64_t)(2 * n) * (uint64_t)(2 * n + 1));
        if (n & 1) { cneg = add(cneg, term); sneg = add(sneg, sterm); }
        else { cpos = add(cpos, term); spos = add(spos, sterm); }
        if (term.is_zero() && sterm.is_zero()) break;
    }
    if (cmp(cpos, cneg) >= 0) *c_out = to_double(sub(cpos, cneg), false); else *c_out = to_double(sub(cneg, cpos), true);
    if (cmp(spos, sneg) >= 0) *s_out = to_double(sub(spos, sneg), false); else *s_out = to_double(sub(sneg, spos), true);
}

void bit_reverse(std::vector<double> &mu, int n) {  // fft.jl:1-15 on complex entries
    int j = 0;
    for (int i = 1; i <= n - 1; i++) {
        int bit = n >> 1;
        while (j >= bit) { j -= bit; bit >>= 1; }
        j += bit;
        if (i < j) { std::swap(mu[2 * i], mu[2 * j]); std::swap(mu[2 * i + 1], mu[2 * j + 1]); }
    }
}

}  // namespace

void make_twiddles(int N, Twiddles &tw) {
    int M = N / 2, logN = __builtin_ctz((unsigned)N);
    std::vector<double> C(N), S(N);
    C[0] = 1.0; S[0] = 0.0;
    for (int t = 1; t < N; t++) sincos_fix(theta_of(t, logN), &S[t], &C[t]);
    tw.psi.assign(2 * M, 0.0); tw.psiinv.assign(2 * M, 0.0); tw.roots.assign(2 * M, 0.0); tw.rootsinv.assign(2 * M, 0.0);
    for (int j = 0; j < M; j++) {
        tw.psi[2 * j] = C[2 * j];    tw.psi[2 * j + 1] = -S[2 * j];     // fft.jl:33
        tw.psiinv[2 * j] = C[2 * j]; tw.psiinv[2 * j + 1] = S[2 * j];   // fft.jl:34
        tw.roots[2 * j] = C[j];      tw.roots[2 * j + 1] = S[j];        // fft.jl:40
        tw.rootsinv[2 * j] = C[j] / (double)M; tw.rootsinv[2 * j + 1] = -S[j] / (double)M;  // fft.jl:41
    }
    // signed zeros of entry 0 as Julia's exp(::Complex{BigFloat}) leaves them
    tw.psi[1] = -0.0; tw.psiinv[1] = 0.0; tw.roots[1] = 0.0; tw.rootsinv[1] = -0.0;
    bit_reverse(tw.psi, M);      // fft.jl:36
    bit_reverse(tw.psiinv, M);   // fft.jl:37

    // fx_exact.hip: cos / sin of pi t / N, with the 1e-77 residue of cos(RN256(pi) / 2) cleared (these tables owe the reference nothing)
    auto cs = [&](int t, double &c, double &s) { c = std::fabs(C[t]) < 1e-60 ? 0.0 : C[t]; s = S[t]; };
    tw.fx_om.assign(2 * M, 0.0); tw.fx_tw.assign(2 * M, 0.0); tw.fx_nat.assign(2 * M, 0.0);
    tw.fx_om[0] = 1.0; tw.fx_nat[0] = 1.0;
    for (int s = 0, m = 1; m < M; s++, m <<= 1)
        for (int i = 0; i < m; i++) {
            int r = 0;
            for (int b = 0; b < s; b++) r |= ((i >> b) & 1) << (s - 1 - b);
            double c, sn;
            cs(r * (N / m), c, sn);
            tw.fx_om[2 * (m + i)] = c; tw.fx_om[2 * (m + i) + 1] = -sn;
            cs(i * (N / m), c, sn);
            tw.fx_nat[2 * (m + i)] = c; tw.fx_nat[2 * (m + i) + 1] = sn;
        }
    for (int j = 0; j < M; j++) { double c, sn; cs(j, c, sn); tw.fx_tw[2 * j] = c; tw.fx_tw[2 * j + 1] = -sn; }
}

}  // namespace mkt
