// Internal host-side declarations shared by the C-ABI translation units.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mktfhe.h"

namespace mkt {

struct Twiddles {  // fft.jl:18-45, M interleaved complex each
    std::vector<double> psi, psiinv, roots, rootsinv;
};
void make_twiddles(int N, Twiddles &tw);

inline bool is_mk(int s) { return s == MKT_CCS || s == MKT_KMS || s == MKT_KMS_BLOCK; }
inline bool is_kms(int s) { return s == MKT_KMS || s == MKT_KMS_BLOCK; }
inline bool is_block(int s) { return s == MKT_LMSS || s == MKT_KMS_BLOCK; }

// derived shape facts used on both sides of the ABI
struct Shape {
    int nparty;     // 1 (SK) or k (MK)
    int kr;         // RLWE length of the RGSW rotation: k (SK), 1 (KMS)
    int kacc;       // mask polys of the accumulator
    int brk_polys;  // polynomials per BRK entry
    int ksk_drows;  // D-1 or D/2
    int ksk_kr;     // ring components covered by one party's KSK
    int lwe_len;    // k*n+1 (MK) or n+1
    size_t word;    // ring word bytes
};
inline Shape shape_of(const mkt_params &p) {
    Shape s;
    s.nparty = is_mk(p.scheme) ? p.k : 1;
    s.kr = is_kms(p.scheme) ? 1 : p.k;
    s.kacc = p.k;
    s.brk_polys = p.scheme == MKT_CCS ? 3 * p.l_uni : (s.kr + 1) * p.l_gsw * (s.kr + 1);
    int D = 1 << p.logD;
    s.ksk_drows = is_block(p.scheme) ? D / 2 : D - 1;
    s.ksk_kr = is_mk(p.scheme) ? 1 : p.k;
    s.lwe_len = s.nparty * p.n + 1;
    s.word = p.W == 64 ? 8 : 4;
    return s;
}
int validate_params(const mkt_params &p, std::string &why);

}  // namespace mkt
