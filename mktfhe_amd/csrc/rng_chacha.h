// Client-side randomness shared by the host generator (client.cpp) and the device generator (keygen.hip):
// ChaCha20 (RFC 8439 block function, 32-bit block counter, 96-bit nonce) keyed by a 256-bit seed, one
// independent stream per (party, purpose, index, index) tuple carried in the nonce.  The reference draws from
// ChaCha20Stream (scheme.jl:352-386, sampler.jl:1-34); like it, every keygen / encryption call takes fresh
// entropy unless the caller pins the seed (tests, benchmarks).
//
// The stream is counter-based, so draw number i is addressable directly (`at`): the device fills a whole
// polynomial in parallel and still produces the words the sequential host loop does.
//
// gauss(): Box-Muller with ln / sqrt / cos written out in IEEE add, mul, div only (contraction off on both
// compilers), so host and device produce identical bits; tails reach 8.5 sigma (the reference uses randn,
// sampler.jl:24-28).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define MKT_HD __host__ __device__ inline
#else
#define MKT_HD inline
#endif

namespace mktrng {

MKT_HD uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

#define MKT_QR(a, b, c, d)                                   \
    a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8);  c += d; b ^= c; b = rotl32(b, 7);

// RFC 8439 2.3: state = constants | key | counter | nonce; 10 double rounds; add the input state
MKT_HD void chacha20_block(const uint32_t key[8], uint32_t counter, const uint32_t nonce[3], uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                      key[4], key[5], key[6], key[7], counter, nonce[0], nonce[1], nonce[2]};
    uint32_t x0 = s[0], x1 = s[1], x2 = s[2], x3 = s[3], x4 = s[4], x5 = s[5], x6 = s[6], x7 = s[7];
    uint32_t x8 = s[8], x9 = s[9], x10 = s[10], x11 = s[11], x12 = s[12], x13 = s[13], x14 = s[14], x15 = s[15];
    for (int r = 0; r < 10; r++) {
        MKT_QR(x0, x4, x8, x12) MKT_QR(x1, x5, x9, x13) MKT_QR(x2, x6, x10, x14) MKT_QR(x3, x7, x11, x15)
        MKT_QR(x0, x5, x10, x15) MKT_QR(x1, x6, x11, x12) MKT_QR(x2, x7, x8, x13) MKT_QR(x3, x4, x9, x14)
    }
    out[0] = x0 + s[0]; out[1] = x1 + s[1]; out[2] = x2 + s[2]; out[3] = x3 + s[3];
    out[4] = x4 + s[4]; out[5] = x5 + s[5]; out[6] = x6 + s[6]; out[7] = x7 + s[7];
    out[8] = x8 + s[8]; out[9] = x9 + s[9]; out[10] = x10 + s[10]; out[11] = x11 + s[11];
    out[12] = x12 + s[12]; out[13] = x13 + s[13]; out[14] = x14 + s[14]; out[15] = x15 + s[15];
}
#undef MKT_QR

MKT_HD double bits_to_double(uint64_t b) { double d; __builtin_memcpy(&d, &b, 8); return d; }
MKT_HD uint64_t double_to_bits(double d) { uint64_t b; __builtin_memcpy(&b, &d, 8); return b; }

// unit-variance normal deviate from two 64-bit draws (Box-Muller, cosine branch)
MKT_HD double box_muller(uint64_t r1, uint64_t r2) {
    const double u1 = (double)((r1 >> 11) + 1) * 0x1p-53;   // (0, 1]
    const double u2 = (double)(r2 >> 11) * 0x1p-53;         // [0, 1)
    // ln(u1) = e*ln2 + ln(m), m in [sqrt(1/2), sqrt(2)); ln(m) = 2 atanh(s), s = (m-1)/(m+1)
    uint64_t b = double_to_bits(u1);
    int e = (int)((b >> 52) & 0x7ff) - 1022;
    double m = bits_to_double((b & 0x000FFFFFFFFFFFFFull) | 0x3FE0000000000000ull);   // [0.5, 1)
    if (m < 0.70710678118654752) { m = m * 2.0; e -= 1; }
    const double s = (m - 1.0) / (m + 1.0), s2 = s * s;
    double p = 1.0 / 21.0;
    p = p * s2 + 1.0 / 19.0; p = p * s2 + 1.0 / 17.0; p = p * s2 + 1.0 / 15.0; p = p * s2 + 1.0 / 13.0;
    p = p * s2 + 1.0 / 11.0; p = p * s2 + 1.0 / 9.0;  p = p * s2 + 1.0 / 7.0;  p = p * s2 + 1.0 / 5.0;
    p = p * s2 + 1.0 / 3.0;  p = p * s2 + 1.0;
    const double lnu = (double)e * 0.6931471805599453 + 2.0 * s * p;
    const double x = -2.0 * lnu;                            // [0, 73.5]
    double r = 0.0;
    if (x > 0.0) {                                          // sqrt by Newton steps on a bit-level first guess
        r = bits_to_double((double_to_bits(x) >> 1) + 0x1FF8000000000000ull);
        for (int i = 0; i < 6; i++) r = 0.5 * (r + x / r);
    }
    // cos(2 pi u2): quadrant q, angle th = (4 u2 - q) * pi/2 in [0, pi/2), Taylor series in th^2
    const double f4 = u2 * 4.0;
    const int q = (int)f4;
    const double th = (f4 - (double)q) * 1.5707963267948966, t2 = th * th;
    double c = 1.0 / 2432902008176640000.0;                 // 1/20!  (cos up to th^20, sin up to th^21)
    c = c * t2 - 1.0 / 6402373705728000.0;                  // 1/18!
    c = c * t2 + 1.0 / 20922789888000.0;                    // 1/16!
    c = c * t2 - 1.0 / 87178291200.0;                       // 1/14!
    c = c * t2 + 1.0 / 479001600.0;                         // 1/12!
    c = c * t2 - 1.0 / 3628800.0;                           // 1/10!
    c = c * t2 + 1.0 / 40320.0;                             // 1/8!
    c = c * t2 - 1.0 / 720.0;                               // 1/6!
    c = c * t2 + 1.0 / 24.0;                                // 1/4!
    c = c * t2 - 0.5;
    c = c * t2 + 1.0;
    double sn = 1.0 / 51090942171709440000.0;               // 1/21!
    sn = sn * t2 - 1.0 / 121645100408832000.0;              // 1/19!
    sn = sn * t2 + 1.0 / 355687428096000.0;                 // 1/17!
    sn = sn * t2 - 1.0 / 1307674368000.0;                   // 1/15!
    sn = sn * t2 + 1.0 / 6227020800.0;                      // 1/13!
    sn = sn * t2 - 1.0 / 39916800.0;                        // 1/11!
    sn = sn * t2 + 1.0 / 362880.0;                          // 1/9!
    sn = sn * t2 - 1.0 / 5040.0;                            // 1/7!
    sn = sn * t2 + 1.0 / 120.0;                             // 1/5!
    sn = sn * t2 - 1.0 / 6.0;
    sn = sn * t2 + 1.0;
    sn = sn * th;
    const double cv = q == 0 ? c : (q == 1 ? -sn : (q == 2 ? -c : sn));
    return r * cv;
}

// one stream: key + nonce (purpose a | party << 16, index b, index c); 64-bit draws are word pairs of the keystream
struct Rng {
    uint32_t key[8], nonce[3];
    uint32_t buf[16];
    uint32_t blk;       // next block to generate
    int pos;            // next unread word of buf (16 = empty)
    MKT_HD Rng(const uint32_t k[8], uint32_t party, uint32_t a, uint32_t b = 0, uint32_t c = 0) {
        for (int i = 0; i < 8; i++) key[i] = k[i];
        nonce[0] = (a & 0xffffu) | (party << 16); nonce[1] = b; nonce[2] = c;
        blk = 0; pos = 16;
    }
    MKT_HD uint64_t next() {
        if (pos >= 16) { chacha20_block(key, blk++, nonce, buf); pos = 0; }
        const uint64_t v = (uint64_t)buf[pos] | ((uint64_t)buf[pos + 1] << 32);
        pos += 2;
        return v;
    }
    // continue reading at 64-bit draw number `idx` of the stream (random access)
    MKT_HD void seek(uint64_t idx) {
        blk = (uint32_t)(idx >> 3);
        chacha20_block(key, blk++, nonce, buf);
        pos = (int)(idx & 7) * 2;
    }
    MKT_HD double gauss() { const uint64_t r1 = next(), r2 = next(); return box_muller(r1, r2); }
    // round(signed(T), sigma * randn)  (sampler.jl:24-28)
    MKT_HD uint64_t noise(double sigma) { return (uint64_t)(int64_t)__builtin_rint(sigma * gauss()); }
};

}  // namespace mktrng
