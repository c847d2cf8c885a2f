// Client side of the boundary (host only, no GPU): key generation, encryption, decryption.
// Counterpart of the reference's setup / party_keygen / lwe_encrypt / lwe_ith_encrypt / lwe_decrypt /
// CRS (src/tfhe/scheme.jl:151-410, src/tfhe/keygen.jl, src/ciphertext/{lwe,lev,gsw,unienc,key}.jl).
// Randomness: ChaCha20 streams keyed by a 256-bit seed (rng_chacha.h); a NULL seed draws fresh entropy from
// the OS for that call, as the reference does per call (ChaCha20Stream, sampler.jl:1-34) -- pinned seeds are
// for tests and benchmarks only.  Difference by design: the RLWE products are exact integer arithmetic
// mod 2^W (the reference approximates them with a Float64x2 FFT, params.jl:1).  Keys leave in integer (coefficient)
// form; the device pre-transforms them (mkt_load_* with MKT_FMT_INT_COEFF).
#include <cmath>
#include <cstring>
#include <string.h>   // explicit_bzero
#include <functional>
#include <thread>

#include <sys/random.h>

#include "host_internal.h"
#include "rng_chacha.h"

namespace mkt {

int validate_params(const mkt_params &p, std::string &why) {
    auto bad = [&](const char *m) { why = m; return (int)MKT_ERR_ARG; };
    if (p.scheme < MKT_CGGI || p.scheme > MKT_KMS_BLOCK) return bad("unknown scheme");
    if (p.N < 16 || p.N > 4096 || (p.N & (p.N - 1))) return bad("N must be a power of two in [16, 4096]");
    if (p.W != 32 && p.W != 64) return bad("W must be 32 or 64");
    if (p.k < 1 || p.k > 64) return bad("k out of range");
    if (p.n < 1) return bad("n must be positive");
    if (p.f < 1 || p.logD < 1 || p.f * p.logD > 32 || p.logD > 5) return bad("bad key-switch gadget (f*logD <= 32, logD <= 5)");
    if (p.n + 1 > 16384) return bad("n too large");
    auto gadget = [&](int l, int logB) { return l >= 1 && logB >= 1 && logB <= 31 && l * logB <= p.W && l <= 32; };
    if (p.scheme == MKT_CCS) { if (!gadget(p.l_uni, p.logB_uni)) return bad("bad uni gadget"); }
    else if (!gadget(p.l_gsw, p.logB_gsw)) return bad("bad gsw gadget");
    if (is_kms(p.scheme) && (!gadget(p.l_lev, p.logB_lev) || !gadget(p.l_uni, p.logB_uni))) return bad("bad lev/uni gadget");
    if (is_block(p.scheme)) {
        if (p.blk_len < 1 || p.blk_d < 1 || p.blk_len * p.blk_d != p.n) return bad("block schemes need n == blk_len*blk_d");
    }
    if (p.scheme == MKT_KMS_BLOCK && p.n > p.N) return bad("KMS_block needs n <= N");
    return MKT_OK;
}

namespace {

using mktrng::Rng;

// 256-bit seed -> key words; NULL = fresh OS entropy (getrandom), never a fixed default
int seed_to_key(const uint8_t *seed, uint32_t key[8]) {
    uint8_t tmp[32];
    if (!seed) {
        size_t got = 0;
        while (got < sizeof tmp) {
            ssize_t r = getrandom(tmp + got, sizeof tmp - got, 0);
            if (r <= 0) return MKT_ERR_STATE;
            got += (size_t)r;
        }
        seed = tmp;
    }
    std::memcpy(key, seed, 32);
    explicit_bzero(tmp, sizeof tmp);
    return MKT_OK;
}

inline uint64_t wmask(int W) { return W == 64 ? ~0ull : ((1ull << W) - 1); }

// out (+)= a * s in Z[X]/(X^N+1), s with entries in {-1,0,1}
void mul_small_acc(const uint64_t *a, const int8_t *s, uint64_t *out, int N, bool negate) {
    for (int i = 0; i < N; i++) {
        if (!s[i]) continue;
        bool neg = (s[i] < 0) != negate;
        if (!neg) {
            for (int j = 0; j < N - i; j++) out[i + j] += a[j];
            for (int j = N - i; j < N; j++) out[i + j - N] -= a[j];
        } else {
            for (int j = 0; j < N - i; j++) out[i + j] -= a[j];
            for (int j = N - i; j < N; j++) out[i + j - N] += a[j];
        }
    }
}

struct RingKey { std::vector<std::vector<int8_t>> z; };  // k polynomials, binary

}  // namespace
}  // namespace mkt

using namespace mkt;

namespace {

void store_poly(std::vector<uint8_t> &dst, size_t poly_index, const uint64_t *v, int N, int W) {
    if (W == 64) std::memcpy(dst.data() + poly_index * N * 8, v, (size_t)N * 8);
    else { uint32_t *d = (uint32_t *)dst.data() + poly_index * N; for (int i = 0; i < N; i++) d[i] = (uint32_t)v[i]; }
}
void load_poly(const void *src, size_t poly_index, uint64_t *v, int N, int W) {
    if (W == 64) std::memcpy(v, (const uint8_t *)src + poly_index * N * 8, (size_t)N * 8);
    else { const uint32_t *s = (const uint32_t *)src + poly_index * N; for (int i = 0; i < N; i++) v[i] = s[i]; }
}

// RLWEsample (lwe.jl:78-93): a_c uniform, b = -sum a_c z_c + e; polys laid out (b, a_0..a_{kz-1}) in out
void rlwe_sample(Rng &rng, const std::vector<std::vector<int8_t>> &z, int zoff, int kz, double sigma, int N, int W, uint64_t *out) {
    uint64_t m = wmask(W);
    std::memset(out, 0, sizeof(uint64_t) * (size_t)N);
    for (int c = 0; c < kz; c++) {
        uint64_t *a = out + (size_t)(1 + c) * N;
        for (int i = 0; i < N; i++) a[i] = rng.next() & m;
        mul_small_acc(a, z[zoff + c].data(), out, N, true);
    }
    for (int i = 0; i < N; i++) out[i] = (out[i] + rng.noise(sigma)) & m;
}

void parallel_for(int n, const std::function<void(int)> &fn) {
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)(hw ? hw : 4); if (nt > 16) nt = 16; if (nt > n) nt = n; if (nt < 1) nt = 1;
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([=, &fn] { for (int i = t; i < n; i += nt) fn(i); });
    for (auto &x : th) x.join();
}
}  // namespace

extern "C" {

int mkt_make_twiddles(int N, int which, double *out_host) {
    if (!out_host || which < 0 || which > 3 || N < 4 || N > 8192 || (N & (N - 1))) return MKT_ERR_ARG;
    Twiddles tw;
    make_twiddles(N, tw);
    const std::vector<double> &v = which == 0 ? tw.psi : which == 1 ? tw.psiinv : which == 2 ? tw.roots : tw.rootsinv;
    std::memcpy(out_host, v.data(), v.size() * sizeof(double));
    return MKT_OK;
}

// a fresh 256-bit seed from the OS (what a NULL seed argument uses internally)
int mkt_client_random_seed(uint8_t out[32]) {
    if (!out) return MKT_ERR_ARG;
    uint32_t key[8];
    if (seed_to_key(nullptr, key)) return MKT_ERR_STATE;
    std::memcpy(out, key, 32);
    explicit_bzero(key, sizeof key);
    return MKT_OK;
}

// TESTS AND BENCHMARKS ONLY: expands a small integer into a 256-bit seed so that keys and ciphertexts are
// reproducible.  Anything encrypted under such a seed is public.
int mkt_client_test_seed(uint64_t n, uint8_t out[32]) {
    if (!out) return MKT_ERR_ARG;
    uint64_t x = n ^ 0x6D6B746668655F74ull;
    for (int i = 0; i < 4; i++) {   // splitmix64
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        std::memcpy(out + 8 * i, &z, 8);
    }
    return MKT_OK;
}

int mkt_client_crs(const mkt_params *params, const uint8_t *seed, void *crs_out) {
    if (!params || !crs_out) return MKT_ERR_ARG;
    std::string why; if (validate_params(*params, why)) return MKT_ERR_ARG;
    const mkt_params &p = *params;
    if (!is_mk(p.scheme)) return MKT_ERR_ARG;
    uint32_t key[8];
    if (seed_to_key(seed, key)) return MKT_ERR_STATE;
    Rng rng(key, 0, 0xC125);
    uint64_t m = wmask(p.W);
    for (size_t i = 0; i < (size_t)p.l_uni * p.N; i++) {   // scheme.jl:409-410
        uint64_t v = rng.next() & m;
        if (p.W == 64) ((uint64_t *)crs_out)[i] = v; else ((uint32_t *)crs_out)[i] = (uint32_t)v;
    }
    explicit_bzero(key, sizeof key);
    return MKT_OK;
}

}  // extern "C"

// heavy = false: secrets and the small keys only (public key, relinearisation key); the bootstrapping and
// key-switching keys are then generated on the device (mkt_keygen_device) from the same seeded streams
static int party_keygen_impl(const mkt_params *params, const uint8_t *seed, int party, const void *crs,
                             double sigma_lwe, double sigma_ring, bool heavy, mkt_client_party **out) {
    if (!params || !out) return MKT_ERR_ARG;
    std::string why; if (validate_params(*params, why)) return MKT_ERR_ARG;
    const mkt_params &p = *params;
    Shape sh = shape_of(p);
    if (party < 0 || party >= sh.nparty) return MKT_ERR_ARG;
    if (is_mk(p.scheme) && !crs) return MKT_ERR_ARG;
    const int N = p.N, n = p.n, W = p.W;
    const uint64_t wm = wmask(W);
    auto *K = new mkt_client_party();
    K->p = p; K->sh = sh; K->party = party;
    if (seed_to_key(seed, K->key)) { delete K; return MKT_ERR_STATE; }
    const uint32_t *ps = K->key; const uint32_t pa = (uint32_t)party;   // stream key, party index (in the nonce)
    K->sigma_lwe = sigma_lwe; K->sigma_ring = sigma_ring; K->heavy = heavy;

    // ---- secret keys: key.jl:12-19 (binary), sampler.jl:7-21 (block binary), key.jl:52-87 (partial ring key)
    {
        Rng r(ps, pa, 1);
        K->lwekey.assign(n, 0);
        if (is_block(p.scheme)) {
            for (int i = 0; i < p.blk_d; i++) {
                int idx = (int)(r.next() % (uint64_t)(p.blk_len + 1));   // rand(0:l)
                if (idx) K->lwekey[(size_t)i * p.blk_len + idx - 1] = 1;
            }
        } else for (int i = 0; i < n; i++) K->lwekey[i] = (uint32_t)(r.next() >> 63);
        auto binary_poly = [&](std::vector<int8_t> &z) { z.assign(N, 0); for (int i = 0; i < N; i++) z[i] = (int8_t)(r.next() >> 63); };
        auto partial = [&](std::vector<std::vector<int8_t>> &zs, int kz) {   // ring key embeds the LWE key (key.jl:52-69)
            zs.resize(kz);
            for (int c = 0; c < kz; c++) { binary_poly(zs[c]); for (int i = 0; i < N; i++) { long g = (long)c * N + i; if (g < n) zs[c][i] = (int8_t)K->lwekey[g]; } }
        };
        if (p.scheme == MKT_CGGI) { K->zring.resize(p.k); for (auto &z : K->zring) binary_poly(z); }
        else if (p.scheme == MKT_LMSS) partial(K->zring, p.k);
        else if (p.scheme == MKT_CCS) { K->zring.resize(1); binary_poly(K->zring[0]); }
        else if (p.scheme == MKT_KMS) { K->zring.resize(2); binary_poly(K->zring[0]); binary_poly(K->zring[1]); }
        else { K->zring.resize(2); binary_poly(K->zring[0]); std::vector<std::vector<int8_t>> u; partial(u, 1); K->zring[1] = u[0]; }
    }

    // ---- bootstrapping key
    if (!heavy) {
    } else if (p.scheme != MKT_CCS) {
        // brk[i] = RGSW_z(s_i): keygen.jl:13-15,:39-41,:106-108,:143-145; gsw.jl:174-178; lev.jl:88-102
        const int kr = sh.kr, l = p.l_gsw, rows = (kr + 1) * l, polys = kr + 1;
        K->brk.assign((size_t)n * rows * polys * N * sh.word, 0);
        parallel_for(n, [&](int i) {
            std::vector<uint64_t> buf((size_t)polys * N);
            for (int c = 0; c <= kr; c++) for (int j = 0; j < l; j++) {
                Rng r(ps, pa, 2, (uint32_t)i, (uint32_t)(c * l + j));
                rlwe_sample(r, K->zring, 0, kr, sigma_ring, N, W, buf.data());
                uint64_t g = 1ull << (W - (j + 1) * p.logB_gsw);
                buf[(size_t)c * N] = (buf[(size_t)c * N] + (uint64_t)K->lwekey[i] * g) & wm;  // c = 0: b[0]; c >= 1: a_{c-1}[0]
                for (int q = 0; q < polys; q++)
                    store_poly(K->brk, ((size_t)i * rows + (size_t)c * l + j) * polys + q, buf.data() + (size_t)q * N, N, W);
            }
        });
    } else {
        // brk[i] = UniEnc_z(s_i): keygen.jl:71-73; unienc.jl:36-55
        const int l = p.l_uni;
        K->brk.assign((size_t)n * 3 * l * N * sh.word, 0);
        parallel_for(n, [&](int i) {
            Rng r(ps, pa, 2, (uint32_t)i);
            std::vector<int8_t> rt(N);
            for (int q = 0; q < N; q++) rt[q] = (int8_t)((int)(r.next() % 3) - 1);      // ternary r
            std::vector<uint64_t> a(N), d(N), rl((size_t)2 * N);
            std::vector<std::vector<int8_t>> rk{rt};
            for (int j = 0; j < l; j++) {
                uint64_t g = 1ull << (W - (j + 1) * p.logB_uni);
                load_poly(crs, j, a.data(), N, W);
                std::fill(d.begin(), d.end(), 0);
                mul_small_acc(a.data(), rt.data(), d.data(), N, false);                 // crs[j]*r
                d[0] += (uint64_t)K->lwekey[i] * g;
                for (int q = 0; q < N; q++) d[q] = (d[q] + r.noise(sigma_ring)) & wm;
                store_poly(K->brk, (size_t)i * 3 * l + j, d.data(), N, W);
                rlwe_sample(r, K->zring, 0, 1, sigma_ring, N, W, rl.data());             // f.stack[j] = RLWE_z(g_j * r)
                for (int q = 0; q < N; q++) rl[q] = (rl[q] + g * (uint64_t)(int64_t)rt[q]) & wm;
                store_poly(K->brk, (size_t)i * 3 * l + l + 2 * j, rl.data(), N, W);
                store_poly(K->brk, (size_t)i * 3 * l + l + 2 * j + 1, rl.data() + N, N, W);
            }
        });
    }

    // ---- CCS/KMS: public key b (unienc.jl:77-90) and KMS relinearisation key (keygen.jl:103)
    if (is_mk(p.scheme)) {
        const int l = p.l_uni;
        const int zi = is_kms(p.scheme) ? 1 : 0;   // uni key
        K->pub.assign((size_t)l * N * sh.word, 0);
        std::vector<uint64_t> a(N), b(N);
        Rng r(ps, pa, 3);
        for (int j = 0; j < l; j++) {
            load_poly(crs, j, a.data(), N, W);
            std::fill(b.begin(), b.end(), 0);
            mul_small_acc(a.data(), K->zring[zi].data(), b.data(), N, true);            // -z*crs[j]
            for (int q = 0; q < N; q++) b[q] = (b[q] + r.noise(sigma_ring)) & wm;
            store_poly(K->pub, j, b.data(), N, W);
        }
        if (is_kms(p.scheme)) {
            K->rlk_d.assign((size_t)l * N * sh.word, 0);
            K->rlk_f.assign((size_t)l * 2 * N * sh.word, 0);
            Rng r2(ps, pa, 4);
            std::vector<int8_t> rt(N);
            for (int q = 0; q < N; q++) rt[q] = (int8_t)((int)(r2.next() % 3) - 1);
            std::vector<uint64_t> d(N), rl((size_t)2 * N);
            for (int j = 0; j < l; j++) {
                uint64_t g = 1ull << (W - (j + 1) * p.logB_uni);
                load_poly(crs, j, a.data(), N, W);
                std::fill(d.begin(), d.end(), 0);
                mul_small_acc(a.data(), rt.data(), d.data(), N, false);
                for (int q = 0; q < N; q++) d[q] = (d[q] + g * (uint64_t)K->zring[0][q] + r2.noise(sigma_ring)) & wm;  // + g_j * z'
                store_poly(K->rlk_d, j, d.data(), N, W);
                rlwe_sample(r2, K->zring, 1, 1, sigma_ring, N, W, rl.data());
                for (int q = 0; q < N; q++) rl[q] = (rl[q] + g * (uint64_t)(int64_t)rt[q]) & wm;
                store_poly(K->rlk_f, 2 * j, rl.data(), N, W);
                store_poly(K->rlk_f, 2 * j + 1, rl.data() + N, N, W);
            }
        }
    }

    // ---- key-switching key: keygen.jl:17-23,:43-51,:75-79,:110-114,:147-151; lev.jl:31-37; lwe.jl:11-22
    if (heavy) {
        const int f = p.f, logD = p.logD, dr = sh.ksk_drows, kk = sh.ksk_kr;
        const int zoff = is_kms(p.scheme) ? 1 : 0;
        const size_t n1 = (size_t)n + 1;
        K->ksk.assign((size_t)kk * N * dr * f * n1, 0);
        parallel_for(kk * N, [&](int cj) {
            int c = cj / N, j = cj % N;
            if (is_block(p.scheme) && (long)c * N + j < n) return;       // keygen.jl:46,:147: only beyond the embedded LWE key
            Rng r(ps, pa, 5, (uint32_t)cj);
            for (int d = 0; d < dr; d++) for (int t = 0; t < f; t++) {
                uint32_t *row = K->ksk.data() + ((((size_t)c * N + j) * dr + d) * f + t) * n1;
                uint32_t msg = (uint32_t)((uint32_t)K->zring[zoff + c][j] * (uint32_t)(d + 1)) << (32 - (t + 1) * logD);
                uint32_t dot = 0;
                for (int q = 0; q < n; q++) { row[q] = (uint32_t)r.next(); dot += row[q] * K->lwekey[q]; }
                row[n] = (uint32_t)r.noise(sigma_lwe) - dot + msg;
            }
        });
    }
    *out = K;
    return MKT_OK;
}

extern "C" {

int mkt_client_party_keygen(const mkt_params *params, const uint8_t *seed, int party, const void *crs,
                            double sigma_lwe, double sigma_ring, mkt_client_party **out) {
    return party_keygen_impl(params, seed, party, crs, sigma_lwe, sigma_ring, true, out);
}

int mkt_client_party_secrets(const mkt_params *params, const uint8_t *seed, int party, const void *crs,
                             double sigma_lwe, double sigma_ring, mkt_client_party **out) {
    return party_keygen_impl(params, seed, party, crs, sigma_lwe, sigma_ring, false, out);
}

int mkt_client_party_destroy(mkt_client_party *p) {
    if (!p) return MKT_OK;
    // wipe the secrets before the memory goes back to the allocator
    explicit_bzero(p->key, sizeof p->key);
    if (!p->lwekey.empty()) explicit_bzero(p->lwekey.data(), p->lwekey.size() * sizeof(uint32_t));
    for (auto &z : p->zring) if (!z.empty()) explicit_bzero(z.data(), z.size());
    delete p;
    return MKT_OK;
}
const uint32_t *mkt_client_lwekey(const mkt_client_party *p) { return p ? p->lwekey.data() : nullptr; }
// ring secret key polynomial idx (SK schemes: idx < k; CCS: 0; KMS: 0 = gsw key z', 1 = uni key z), N entries 0/1
const int8_t *mkt_client_ringkey(const mkt_client_party *p, int idx, size_t *bytes) {
    if (!p || idx < 0 || idx >= (int)p->zring.size()) { if (bytes) *bytes = 0; return nullptr; }
    if (bytes) *bytes = p->zring[idx].size();
    return p->zring[idx].data();
}
const void *mkt_client_brk(const mkt_client_party *p, size_t *bytes) { if (bytes) *bytes = p->brk.size(); return p->brk.data(); }
const uint32_t *mkt_client_ksk(const mkt_client_party *p, size_t *bytes) { if (bytes) *bytes = p->ksk.size() * 4; return p->ksk.data(); }
const void *mkt_client_rlk_d(const mkt_client_party *p, size_t *bytes) { if (bytes) *bytes = p->rlk_d.size(); return p->rlk_d.data(); }
const void *mkt_client_rlk_f(const mkt_client_party *p, size_t *bytes) { if (bytes) *bytes = p->rlk_f.size(); return p->rlk_f.data(); }
const void *mkt_client_pubkey(const mkt_client_party *p, size_t *bytes) { if (bytes) *bytes = p->pub.size(); return p->pub.data(); }

// scheme.jl:352-386 lwe_encrypt / lwe_ith_encrypt: b = e - <a,s> + (2m-1)*2^29, mask in the party's block
int mkt_client_lwe_encrypt(const mkt_params *params, const mkt_client_party *K, int party, int bit,
                           double sigma_lwe, const uint8_t *seed, uint32_t *out) {
    if (!params || !K || !out) return MKT_ERR_ARG;
    const mkt_params &p = *params;
    Shape sh = shape_of(p);
    if (party < 0 || party >= sh.nparty) return MKT_ERR_ARG;
    std::memset(out, 0, sizeof(uint32_t) * (size_t)sh.lwe_len);
    uint32_t key[8];
    if (seed_to_key(seed, key)) return MKT_ERR_STATE;
    Rng r(key, (uint32_t)party, 7);
    uint32_t *a = out + (size_t)party * p.n;
    uint32_t dot = 0;
    uint32_t e = (uint32_t)r.noise(sigma_lwe);
    for (int i = 0; i < p.n; i++) { a[i] = (uint32_t)r.next(); dot += a[i] * K->lwekey[i]; }
    uint32_t mu = (uint32_t)(2 * (bit ? 1 : 0) - 1);
    out[sh.lwe_len - 1] = e + (0u - dot + (mu << 29));
    explicit_bzero(key, sizeof key);
    return MKT_OK;
}

// scheme.jl:388-407
int mkt_client_lwe_decrypt(const mkt_params *params, const mkt_client_party *const *keys, int nparties, const uint32_t *lwe) {
    if (!params || !keys || !lwe) return MKT_ERR_ARG;
    const mkt_params &p = *params;
    Shape sh = shape_of(p);
    if (nparties != sh.nparty) return MKT_ERR_ARG;
    uint32_t b = lwe[sh.lwe_len - 1];
    for (int i = 0; i < nparties; i++)
        for (int q = 0; q < p.n; q++) b += keys[i]->lwekey[q] * lwe[(size_t)i * p.n + q];
    if (!is_mk(p.scheme)) {                       // divbits(phase, 29) == 1
        uint32_t carry = (b << 3) >> 31;
        return ((b >> 29) + carry) == 1 ? 1 : 0;
    }
    return b < (1u << 31) ? 1 : 0;
}

}  // extern "C"
