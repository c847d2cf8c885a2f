// C ABI of the engine (include/mktfhe.h): per-device context, key upload + on-device pre-transform,
// batched hot-path entry points.  Host code; kernels live in kernels.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string.h>   // explicit_bzero
#include <cmath>
#include <memory>
#include <string>
#include <vector>

#include "device_api.h"
#include "fft_device.h"
#include "host_internal.h"

using mktd::cplx;

namespace {
thread_local std::string g_create_error;

struct TimedSpan { int cls; hipEvent_t a, b; };

int env_int(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }

// Kernel-selection switches of one context (A/B runs and the parity tests that force every kernel variant).  The
// environment seeds them ONCE, at mkt_ctx_create; afterwards only mkt_set_option changes them -- the call path never
// reads the environment.
struct Tune {
    int rot_variant = 0;
    int rot_stagger = 16;   // tools/stagger.sh: 16.99 -> 15.63 ms at KMS k=2 N=1024 on one device, neutral elsewhere
    int rot_split = 0;
    int rot_wide = 0;       // latency variant: 0 automatic, 1 never, 2 always where supported
    int rot_blkg = 0;       // block schemes: rotations per workgroup, 0 automatic
    int ccs_stagger = 0;
    int ccs_pipe = -1;      // two-group CCS kernel: -1 automatic (below one chip-fill), 0 never, 1 always
    int rot_map = 1;        // workgroup id -> (ciphertext, slot) mapping of the k = 1 rotation kernels (kernel_common.h rot_decode): 1 = the RLEV rows of one
                            // (ciphertext, party) on one XCD at one time -- 25 % less fabric traffic at KMS k = 2 (FETCH_SIZE 15.3 -> 11.4 GB per launch,
                            // L2 misses -27 %), time -0.3 ... -2.5 % (profiles/r04j_bench_kms2_n1024_map{0,1}_pmc.txt)
    int exact_wide = 1;     // EXACT (integer NTT) KMS phase 1 at l_gsw = 2 and KMS_block phase 1: 1 = the paired-transform kernel / one set of digit transforms per block (default), 0 = the one-at-a-time kernel (reference loop order; tests force both)
    int exact_kany = 0;     // EXACT CGGI / LMSS: 1 = the run-time-RLWE-length kernel (sums in memory) also where the register kernels serve (k <= 3); tests
    int exact_impl = -1;    // EXACT blind rotation of CGGI (RLWE length 1) and KMS phase 1: 0 = integer NTT over two 30-bit primes (ntt_exact.hip), 1 / -1 = the Float64 pipe (ahead at every measured shape: profiles/r06_fx_shapes.txt)
                            // (fx_exact.hip: FMA transforms over 16-bit key limbs) wherever its error bound certifies the loaded keys (fx_usable), the integer NTT elsewhere
    void from_env() {
        exact_impl = env_int("MKT_EXACT_IMPL", exact_impl);
        rot_variant = env_int("MKT_ROT_VARIANT", rot_variant); rot_stagger = env_int("MKT_ROT_STAGGER", rot_stagger);
        rot_split = env_int("MKT_ROT_SPLIT", rot_split); rot_wide = env_int("MKT_ROT_WIDE", rot_wide);
        rot_blkg = env_int("MKT_ROT_BLKG", rot_blkg); ccs_stagger = env_int("MKT_CCS_STAGGER", ccs_stagger);
        ccs_pipe = env_int("MKT_CCS_PIPE", ccs_pipe); exact_wide = env_int("MKT_EXACT_WIDE", exact_wide); rot_map = env_int("MKT_ROT_MAP", rot_map); exact_kany = env_int("MKT_EXACT_KANY", exact_kany);
    }
};
}  // namespace

thread_local const char *mktd::last_rot_kernel = "";

// launcher-level switches (grid shapes of the transform and key-switch kernels): process-wide, read from the environment
// on first use only (device_api.h)
const mktd::LaunchTuning &mktd::launch_tuning() {
    static const LaunchTuning t = [] {
        LaunchTuning q{};
        q.fft_grid = env_int("MKT_FFT_GRID", 0); q.fft_nb = env_int("MKT_FFT_NB", 1); q.fft_igrid = env_int("MKT_FFT_IGRID", 0);
        q.ks_g = env_int("MKT_KS_G", 32); q.ks_blocks = env_int("MKT_KS_BLOCKS", 0); q.ks_waves = env_int("MKT_KS_WAVES", 0); q.ks_pair = env_int("MKT_KS_PAIR", -1); q.ntt_grid = env_int("MKT_NTT_GRID", 0);
        return q;
    }();
    return t;
}

// The evaluation keys and tables of one scheme on one device: immutable once a second context shares them
// (mkt_ctx_fork), freed when the last context that holds them is destroyed.  This is the reference's scheme object
// proper -- read-only during evaluation, shared by concurrent callers (bootstrapping.jl:38-45 allocates all scratch per
// call) -- while mkt_ctx adds what a caller must not share: stream, workspace, timing spans, error string.
struct KeySet {
    int device = 0;
    mkt::Twiddles tw;
    cplx *d_tw = nullptr;        // psi | psiinv | roots | rootsinv, M each
    cplx *d_monomial = nullptr;  // [2N][M]
    cplx *d_brk = nullptr;  size_t brk_party_cplx = 0;  std::vector<char> brk_loaded;
    uint32_t *d_ksk = nullptr; size_t ksk_party_words = 0; int n1p = 0; std::vector<char> ksk_loaded;
    cplx *d_rlk_d = nullptr, *d_rlk_f = nullptr, *d_pub = nullptr, *d_crs = nullptr;
    std::vector<char> rlk_loaded, pub_loaded; bool crs_loaded = false;
    // rotation slots (KMS phase 1: party-major rows)
    int rtot = 1;
    int *d_slot_party = nullptr, *d_slot_row = nullptr;
    uint64_t *d_ntt = nullptr;   // MKT_ARITH_EXACT: psi_rev (negated) | N^-1, N^-1 w | N^-1 2^32, N^-1 2^32 w, with Shoup companions
    // MKT_ARITH_EXACT on the Float64 pipe (fx_exact.hip), where the shape has the kernel: the engine's own tables and the bootstrapping key as limb transforms
    cplx *d_fx_tab = nullptr;    // fx_om | fx_tw | fx_nat, M each
    cplx *d_fx_brk = nullptr;  size_t fx_brk_party_cplx = 0;   // [party][n][2l][2][W/16][M], scaled by 1 / M
    unsigned long long *d_fx_stat = nullptr;   // [0] largest |key transform value|^2 over the loaded keys, [1] largest rounding distance of the last fx polymul (bit patterns)
    double fx_kmax = 0.0;        // sqrt of [0], read back after every key load
    ~KeySet() {
        int prev = -1;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) (void)hipSetDevice(device);
        void *ptrs[] = {d_tw, d_monomial, d_brk, d_ksk, d_rlk_d, d_rlk_f, d_pub, d_crs, d_slot_party, d_slot_row, d_ntt, d_fx_tab, d_fx_brk, d_fx_stat};
        for (void *p : ptrs) if (p) (void)hipFree(p);
        if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    }
};

struct mkt_ctx {
    mkt_params p;
    mkt::Shape sh;
    int device = 0;
    int logM = 0, logN = 0, M = 0;
    int dev_order = MKT_DEVORDER;   // device point order of this context's resident tables (fft_device.h dev_pos)
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;   // a fork's own non-blocking stream (destroyed with the context); `stream` may be re-pointed by mkt_set_stream
    std::string err;
    std::shared_ptr<KeySet> ks;  // shared with the contexts forked from this one
    bool exact = false;          // MKT_ARITH_EXACT (integer NTT, two 30-bit primes)
    uint64_t *d_ntt = nullptr;   // = ks->d_ntt (owned by the key set: shared by forks)
    int split = 1;               // EXACT on the 64-bit ring: every resident 64-bit table is kept as (low, high) residue polynomials -> 2 per logical polynomial
    // workspace
    size_t ws_gates = 0;
    uint32_t *ws_lin = nullptr;
    void *ws_acc = nullptr;
    cplx *ws_lev = nullptr, *ws_scratch = nullptr;
    void *ws_fxacc = nullptr;    // fx_exact.hip, KMS: the phase-1 rows as ring words [gates][rtot][2][N] before they become split residue tables
    uint32_t *ws_ksd = nullptr; size_t ws_ksd_words = 0;   // key switch: prepared digit words + partial sums per slab (grows with the largest batch seen)
    // timing
    bool timing = false;
    std::vector<TimedSpan> spans;
    Tune tune;
    const char *last_rot_kernel = "";   // name of the blind-rotation kernel the last call launched (mkt_last_kernel_name)
    double fx_last_resid = 0.0;         // fx polymul: largest |q - round(q)| of the last call

    const cplx *fx_om() const { return ks->d_fx_tab; }
    const cplx *fx_tw() const { return ks->d_fx_tab + M; }
    const cplx *fx_nat() const { return ks->d_fx_tab + 2 * (size_t)M; }
    mktd::TwPtrs twp() const { return mktd::TwPtrs{ks->d_tw, ks->d_tw + M, ks->d_tw + 2 * (size_t)M, ks->d_tw + 3 * (size_t)M}; }
    bool keys_shared() const { return ks.use_count() > 1; }
};

namespace {

int fail(mkt_ctx *c, int code, const std::string &msg) { if (c) c->err = msg; else g_create_error = msg; return code; }
int hipfail(mkt_ctx *c, hipError_t e, const char *what) {
    return fail(c, MKT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(c, call) do { hipError_t _e = (call); if (_e != hipSuccess) return hipfail((c), _e, #call); } while (0)

struct DevGuard {   // make the context's device current for the duration of a call
    int prev = -1; bool ok = true;
    explicit DevGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) ok = hipSetDevice(dev) == hipSuccess; }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct Timer {
    mkt_ctx *c; int cls; hipEvent_t a = nullptr, b = nullptr;
    Timer(mkt_ctx *c_, int cls_) : c(c_), cls(cls_) {
        if (c->timing && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) (void)hipEventRecord(a, c->stream);
        else a = b = nullptr;
    }
    ~Timer() {
        if (cls == 1) c->last_rot_kernel = mktd::last_rot_kernel;   // the blind-rotation launcher noted which kernel it picked
        if (a && b) { (void)hipEventRecord(b, c->stream); c->spans.push_back(TimedSpan{cls, a, b}); }
    }
};

void clear_spans(mkt_ctx *c) {
    for (auto &s : c->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    c->spans.clear();
}

size_t poly_bytes(const mkt_ctx *c) { return (size_t)c->p.N * c->sh.word; }

// transform `npolys` coefficient-form polynomials (host) into TransPolys at `dst` (device)
int fx_after_key_load(mkt_ctx *c);
bool fx_usable(const mkt_ctx *c);
mktd::FxRotArgs fx_rot_args(mkt_ctx *c, const uint32_t *lwe, int stride, int pre);
int upload_polys(mkt_ctx *c, const void *host, size_t npolys, cplx *dst, int fmt, bool small = false, cplx *fx_dst = nullptr) {   // small: coefficients far below 2^32 in magnitude (monomials): never split; fx_dst: also as limb transforms (fx_exact.hip)
    if (c->exact && fmt != MKT_FMT_INT_COEFF) return fail(c, MKT_ERR_UNSUPPORTED, "an MKT_ARITH_EXACT context takes keys in integer form (MKT_FMT_INT_COEFF)");
    if (fmt == MKT_FMT_F64_FFT) {   // the reference's Trans* values: copy, then natural -> device point order
        cplx *tmpc = nullptr;
        const size_t nb = npolys * (size_t)c->M * sizeof(cplx);
        HIPCHK(c, hipMalloc((void **)&tmpc, nb));
        hipError_t e = hipMemcpyAsync(tmpc, host, nb, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = mktd::launch_reorder(c->logM, tmpc, dst, npolys, 1, c->dev_order, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        (void)hipFree(tmpc);
        if (e != hipSuccess) return hipfail(c, e, "key upload");
        return MKT_OK;
    }
    if (fmt != MKT_FMT_INT_COEFF) return fail(c, MKT_ERR_ARG, "unknown key format");
    void *tmp = nullptr;
    HIPCHK(c, hipMalloc(&tmp, npolys * poly_bytes(c)));
    hipError_t e = hipMemcpyAsync(tmp, host, npolys * poly_bytes(c), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = !c->exact ? mktd::launch_transform_fwd(c->logM, c->p.W, c->twp(), tmp, dst, npolys, c->dev_order, c->stream)
                           : (c->split == 2 && !small) ? mktd::launch_ntt_fwd_split(c->logN, c->d_ntt, tmp, reinterpret_cast<uint64_t *>(dst), npolys, c->stream)   // 2 residue polynomials per input
                           : mktd::launch_ntt_fwd(c->logN, c->p.W, c->d_ntt, tmp, reinterpret_cast<uint64_t *>(dst), npolys, 1, c->stream);   // N residues = the bytes of M complex
    if (e == hipSuccess && fx_dst) e = mktd::launch_fx_key_fwd(c->logM, c->p.W, c->fx_om(), c->fx_tw(), tmp, fx_dst, npolys, c->ks->d_fx_stat, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return hipfail(c, e, "key pre-transform");
    if (fx_dst) return fx_after_key_load(c);
    return MKT_OK;
}

int build_monomial(mkt_ctx *c) {   // scheme.jl:121-146
    const int N = c->p.N;
    const size_t wb = c->sh.word;
    std::vector<unsigned char> host((size_t)2 * N * N * wb, 0);
    auto set = [&](int e, int i, uint64_t v) {
        unsigned char *p = host.data() + ((size_t)(e - 1) * N + i) * wb;
        if (wb == 8) std::memcpy(p, &v, 8); else { uint32_t w = (uint32_t)v; std::memcpy(p, &w, 4); }
    };
    const uint64_t m1 = ~0ull;
    for (int e = 1; e < N; e++) { set(e, 0, m1); set(e, e, 1); }            // -1 + X^e
    set(N, 0, m1 - 1);                                                       // -2
    for (int e = N + 1; e < 2 * N; e++) { set(e, 0, m1); set(e, e - N, m1); } // -1 - X^(e-N)
    int r = upload_polys(c, host.data(), (size_t)2 * N, c->ks->d_monomial, MKT_FMT_INT_COEFF, true);
    if (r) return r;
    HIPCHK(c, hipMemsetAsync(c->ks->d_monomial + (size_t)(2 * N - 1) * c->M, 0, (size_t)c->M * sizeof(cplx), c->stream));  // entry 2N = 0
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MKT_OK;
}

// fft_device.h (MKT_FFT_SPECIAL): the butterflies of the first two forward / last two inverse stages are written for
// Psi[1] = (eps, -1), Psi[2] = (c, -c), Psi[3] = (-c, -c) -- the shape of the reference's tables (fft.jl:31-37 at any size)
static bool twiddle_shape_ok(const std::vector<double> &psi, int M) {
    if (M < 4) return false;
    return psi[3] == -1.0 && psi[5] == -psi[4] && psi[4] > 0.0 && psi[7] == psi[6] && psi[6] == -psi[4];
}
int upload_twiddles(mkt_ctx *c) {
    if (!twiddle_shape_ok(c->ks->tw.psi, c->M)) return fail(c, MKT_ERR_ARG, "twiddle table Psi does not have the reference's shape (Psi[1] = (eps,-1), Psi[2] = (c,-c), Psi[3] = (-c,-c))");
    const size_t tb = (size_t)c->M * sizeof(cplx);
    HIPCHK(c, hipMemcpy(c->ks->d_tw, c->ks->tw.psi.data(), tb, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->ks->d_tw + c->M, c->ks->tw.psiinv.data(), tb, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->ks->d_tw + 2 * (size_t)c->M, c->ks->tw.roots.data(), tb, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->ks->d_tw + 3 * (size_t)c->M, c->ks->tw.rootsinv.data(), tb, hipMemcpyHostToDevice));
    return MKT_OK;
}

int ensure_workspace(mkt_ctx *c, size_t gates) {
    if (gates <= c->ws_gates) return MKT_OK;
    if (c->ws_lin) (void)hipFree(c->ws_lin);
    if (c->ws_acc) (void)hipFree(c->ws_acc);
    if (c->ws_lev) (void)hipFree(c->ws_lev);
    if (c->ws_scratch) (void)hipFree(c->ws_scratch);
    if (c->ws_fxacc) (void)hipFree(c->ws_fxacc);
    c->ws_lin = nullptr; c->ws_acc = nullptr; c->ws_lev = nullptr; c->ws_scratch = nullptr; c->ws_fxacc = nullptr; c->ws_gates = 0;
    const mkt_params &p = c->p;
    HIPCHK(c, hipMalloc((void **)&c->ws_lin, gates * (size_t)c->sh.lwe_len * 4));
    HIPCHK(c, hipMalloc(&c->ws_acc, gates * (size_t)(1 + c->sh.kacc) * poly_bytes(c)));
    if (mkt::is_kms(p.scheme)) {
        HIPCHK(c, hipMalloc((void **)&c->ws_lev, gates * (size_t)c->ks->rtot * 2 * c->M * sizeof(cplx) * c->split));
        HIPCHK(c, hipMalloc((void **)&c->ws_scratch, gates * (size_t)2 * (p.k + 1) * c->M * sizeof(cplx) * c->split));
        if (c->ks->d_fx_brk) HIPCHK(c, hipMalloc(&c->ws_fxacc, gates * (size_t)c->ks->rtot * 2 * poly_bytes(c)));
    } else if (p.scheme == MKT_CCS) {
        HIPCHK(c, hipMalloc((void **)&c->ws_lev, gates * 3 * poly_bytes(c)));    // v scratch (ring words): parked v + two hand-off slots
        HIPCHK(c, hipMalloc((void **)&c->ws_scratch, gates * (size_t)(p.k + 1) * c->M * sizeof(cplx)));
    } else if (p.k > 3 || (c->exact && c->tune.exact_kany == 1)) {                      // CGGI / LMSS beyond RLWE length 3: tacc and tacc2 of blindrotate_kany_kernel / exact_blindrotate_kany_kernel (same byte count)
        HIPCHK(c, hipMalloc((void **)&c->ws_scratch, gates * (size_t)2 * (p.k + 1) * c->M * sizeof(cplx)));
    }
    c->ws_gates = gates;
    return MKT_OK;
}

constexpr size_t CHUNK_GATES = 8192;   // bounds the workspace (KMS N=2048, k=2: ~1.9 GiB)

int check_ready(mkt_ctx *c, bool need_brk, bool need_ksk) {
    for (int i = 0; i < c->sh.nparty; i++) {
        if (need_brk && !c->ks->brk_loaded[i]) return fail(c, MKT_ERR_STATE, "bootstrapping key not loaded");
        if (need_ksk && !c->ks->ksk_loaded[i]) return fail(c, MKT_ERR_STATE, "key-switching key not loaded");
        if (need_brk && c->p.scheme == MKT_CCS && !c->ks->pub_loaded[i]) return fail(c, MKT_ERR_STATE, "public key not loaded");
        if (need_brk && mkt::is_kms(c->p.scheme) && (!c->ks->rlk_loaded[i] || !c->ks->pub_loaded[i])) return fail(c, MKT_ERR_STATE, "rlk / public key not loaded");
    }
    if (need_brk && mkt::is_mk(c->p.scheme) && !c->ks->crs_loaded) return fail(c, MKT_ERR_STATE, "crs not loaded");
    return MKT_OK;
}

mktd::RotArgs rot_args(mkt_ctx *c, const uint32_t *lwe, int stride, int pre) {
    const mkt_params &p = c->p;
    mktd::RotArgs a{};
    a.tw = c->twp(); a.brk = c->ks->d_brk; a.brk_party_stride = c->ks->brk_party_cplx; a.monomial = c->ks->d_monomial;
    a.lwe = lwe; a.lwe_stride = stride; a.pre_switched = pre; a.n = p.n; a.logN = c->logN;
    a.l = p.l_gsw; a.logB = p.logB_gsw;
    a.blk_len = mkt::is_block(p.scheme) ? p.blk_len : 1;
    a.blk_accum = mkt::is_block(p.scheme) ? 1 : 0;
    a.rows_per_gate = c->ks->rtot; a.slot_party = c->ks->d_slot_party; a.slot_row = c->ks->d_slot_row;
    a.logB_lev = p.logB_lev; a.dev_order = c->dev_order;
    a.variant = c->tune.rot_variant; a.stagger = c->tune.rot_stagger; a.split = (unsigned)c->tune.rot_split;
    a.wide = c->tune.rot_wide; a.blk_group = c->tune.rot_blkg; a.map_mode = c->tune.rot_map;
    return a;
}

// blind rotation of `B` accumulators resident at `acc` ([B][1+k][N]); atilde source described by (lwe, stride, pre)
int do_blindrotate(mkt_ctx *c, const uint32_t *lwe, int stride, int pre, const uint32_t *lin_for_tv, void *acc, cplx *lev, cplx *scratch, size_t B) {
    const mkt_params &p = c->p;
    if (p.scheme == MKT_CCS && c->exact) {   // hybrid products over Z_P (ntt_exact.hip)
        mktd::ExactCcsHostArgs q{};
        q.lwe = lwe; q.lwe_stride = stride; q.pre_switched = pre; q.n = p.n; q.k = p.k; q.l = p.l_uni; q.logB = p.logB_uni;
        q.brk = reinterpret_cast<const uint64_t *>(c->ks->d_brk); q.brk_party_stride = c->ks->brk_party_cplx * 2;
        q.pub_b = reinterpret_cast<const uint64_t *>(c->ks->d_pub); q.crs = reinterpret_cast<const uint64_t *>(c->ks->d_crs);
        q.mono = reinterpret_cast<const uint64_t *>(c->ks->d_monomial); q.acc = (uint32_t *)acc; q.scratch = reinterpret_cast<uint64_t *>(scratch);
        Timer tm(c, 1);
        HIPCHK(c, mktd::launch_exact_ccs(c->logN, c->d_ntt, q, B, c->stream));
        return MKT_OK;
    }
    if (p.scheme == MKT_CCS) {
        mktd::CcsArgs q{};
        q.tw = c->twp(); q.lwe = lwe; q.lwe_stride = stride; q.pre_switched = pre; q.n = p.n; q.logN = c->logN; q.k = p.k;
        q.l = p.l_uni; q.logB = p.logB_uni; q.brk = c->ks->d_brk; q.brk_party_stride = c->ks->brk_party_cplx; q.pub_b = c->ks->d_pub; q.crs = c->ks->d_crs;
        q.monomial = c->ks->d_monomial; q.acc = acc; q.scratch = scratch; q.vscratch = lev;
        q.stagger = c->tune.ccs_stagger; q.dev_order = c->dev_order;
        // batches that leave compute units idle run each ciphertext on two thread groups (ccs_pipe.hip); option ccs_pipe: 0 never,
        // 1 always, -1: below one chip-fill of one-group workgroups (4 per CU at M = 512, 2 at M = 1024)
        const int pipe = c->tune.ccs_pipe;
        const size_t fill = (size_t)256 * (c->logM <= 9 ? 4 : 2);
        const bool use_pipe = pipe == 1 || (pipe < 0 && B * 2 <= fill);
        Timer tm(c, 1);
        if (use_pipe) {
            const hipError_t e = mktd::launch_ccs_pipe(c->logM, p.W, q, B, c->stream);
            if (e == hipSuccess) return MKT_OK;
            if (e != hipErrorInvalidValue) return hipfail(c, e, "launch_ccs_pipe");
        }
        HIPCHK(c, mktd::launch_ccs_blindrotate(c->logM, p.W, q, B, c->stream));
        return MKT_OK;
    }
    if (c->exact && mkt::is_kms(p.scheme)) {   // 64-bit ring, split tables: phase 1 and phase 2 with exact products (ntt_exact.hip)
        mktd::ExactKmsArgs q{};
        q.brk = reinterpret_cast<const uint64_t *>(c->ks->d_brk); q.brk_party_stride = c->ks->brk_party_cplx * 2 /* in 8-byte residue pairs */; q.mono = reinterpret_cast<const uint64_t *>(c->ks->d_monomial);
        q.lwe = lwe; q.lwe_stride = stride; q.pre_switched = pre; q.n = p.n; q.k = p.k; q.l_gsw = p.l_gsw; q.logB_gsw = p.logB_gsw;
        q.l_lev = p.l_lev; q.logB_lev = p.logB_lev; q.l_uni = p.l_uni; q.logB_uni = p.logB_uni; q.rtot = c->ks->rtot; q.lwe_len = c->sh.lwe_len; q.blk_len = p.scheme == MKT_KMS_BLOCK ? p.blk_len : 1;
        q.slot_party = c->ks->d_slot_party; q.slot_row = c->ks->d_slot_row; q.levkey = reinterpret_cast<uint64_t *>(lev);
        q.rlk_d = reinterpret_cast<const uint64_t *>(c->ks->d_rlk_d); q.rlk_f = reinterpret_cast<const uint64_t *>(c->ks->d_rlk_f);
        q.pub_b = reinterpret_cast<const uint64_t *>(c->ks->d_pub); q.crs = reinterpret_cast<const uint64_t *>(c->ks->d_crs);
        q.lin_for_tv = lin_for_tv; q.acc = reinterpret_cast<uint64_t *>(acc); q.scratch = reinterpret_cast<uint64_t *>(scratch); q.phase1_only = 0; q.wide = c->tune.exact_wide;
        Timer tm(c, 1);
        if (p.scheme == MKT_KMS && fx_usable(c) && c->ws_fxacc) {   // phase 1 on the Float64 pipe: rows as ring words, then as split residue tables for the integer phase 2
            mktd::FxRotArgs f = fx_rot_args(c, lwe, stride, pre);
            f.init_mode = 1; f.acc_io = c->ws_fxacc; f.ngates = B;
            const size_t nrot = B * (size_t)c->ks->rtot;
            HIPCHK(c, mktd::launch_fx_blindrotate(c->logM, p.W, f, nrot, c->stream));
            HIPCHK(c, mktd::launch_ntt_fwd_split(c->logN, c->d_ntt, c->ws_fxacc, q.levkey, nrot * 2, c->stream));
            q.phase2_only = 1;
        }
        HIPCHK(c, mktd::launch_exact_kms(c->logN, c->d_ntt, q, B, c->stream));
        if (q.phase2_only) mktd::last_rot_kernel = "fx_blindrotate_kernel";
        return MKT_OK;
    }
    if (c->exact && (p.k > 3 || c->tune.exact_kany == 1)) {   // CGGI / LMSS, any RLWE length: sums in memory
        Timer tm(c, 1);
        HIPCHK(c, mktd::launch_exact_blindrotate_kany(c->logN, c->d_ntt, reinterpret_cast<const uint64_t *>(c->ks->d_brk), reinterpret_cast<const uint64_t *>(c->ks->d_monomial),
                                                      lwe, stride, pre, p.n, p.k, p.l_gsw, p.logB_gsw, mkt::is_block(p.scheme) ? p.blk_len : 1, (uint32_t *)acc,
                                                      reinterpret_cast<uint64_t *>(scratch), B, c->stream));
        return MKT_OK;
    }
    if (c->exact && (p.k > 1 || (mkt::is_block(p.scheme) && p.blk_len != 3))) {   // CGGI / LMSS with RLWE length 2, 3 or another block length: the general kernel
        Timer tm(c, 1);
        HIPCHK(c, mktd::launch_exact_blindrotate_kr(c->logN, c->d_ntt, reinterpret_cast<const uint64_t *>(c->ks->d_brk), reinterpret_cast<const uint64_t *>(c->ks->d_monomial),
                                                    lwe, stride, pre, p.n, p.k, p.l_gsw, p.logB_gsw, mkt::is_block(p.scheme) ? p.blk_len : 1, (uint32_t *)acc, B, c->stream));
        return MKT_OK;
    }
    if (c->exact && p.scheme == MKT_CGGI && fx_usable(c)) {   // CGGI on the Float64 pipe (fx_exact.hip)
        mktd::FxRotArgs f = fx_rot_args(c, lwe, stride, pre);
        f.init_mode = 0; f.acc_io = acc; f.ngates = B;
        Timer tm(c, 1);
        HIPCHK(c, mktd::launch_fx_blindrotate(c->logM, p.W, f, B, c->stream));
        return MKT_OK;
    }
    if (c->exact) {          // CGGI / LMSS, RLWE length 1, 32-bit ring (exact_gate_ok): every product exact mod 2^32
        Timer tm(c, 1);
        HIPCHK(c, mktd::launch_exact_blindrotate(c->logN, c->d_ntt, reinterpret_cast<const uint64_t *>(c->ks->d_brk), reinterpret_cast<const uint64_t *>(c->ks->d_monomial),
                                                 lwe, stride, pre, p.n, p.l_gsw, p.logB_gsw, mkt::is_block(p.scheme) ? p.blk_len : 1, (uint32_t *)acc, B, c->stream));
        return MKT_OK;
    }
    if (!mkt::is_kms(p.scheme)) {
        mktd::RotArgs a = rot_args(c, lwe, stride, pre);
        a.init_mode = 0; a.out_mode = 0; a.acc_io = acc;
        Timer tm(c, 1);
        if (p.k > 3) {   // any RLWE length: accumulators in memory (blindrotate_kany_kernel)
            a.ngates = B;
            HIPCHK(c, mktd::launch_blindrotate_kany(c->logM, p.W, p.k, a, scratch, B, c->stream));
            return MKT_OK;
        }
        if (p.k > 1) {   // RLWE length 2, 3 (CGGI, LMSS): accumulators in registers
            a.ngates = B;
            HIPCHK(c, mktd::launch_blindrotate_kr(c->logM, p.W, p.k, a, B, c->stream));
            return MKT_OK;
        }
        a.ngates = B;
        HIPCHK(c, mktd::launch_blindrotate_k1(c->logM, p.W, a, B, c->stream));
        return MKT_OK;
    }
    {
        mktd::RotArgs a = rot_args(c, lwe, stride, pre);
        a.init_mode = 1; a.out_mode = 1; a.tout = lev; a.tout_natural = 0; a.ngates = B;
        Timer tm(c, 1);
        HIPCHK(c, mktd::launch_blindrotate_k1(c->logM, p.W, a, B * (size_t)c->ks->rtot, c->stream));
    }
    mktd::Phase2Args q{};
    q.tw = c->twp(); q.lin = lin_for_tv; q.lwe_stride = c->sh.lwe_len; q.logN = c->logN;
    q.k = p.k; q.l_lev = p.l_lev; q.logB_lev = p.logB_lev; q.l_uni = p.l_uni; q.logB_uni = p.logB_uni;
    q.levkey = lev; q.rtot = c->ks->rtot; q.rlk_d = c->ks->d_rlk_d; q.rlk_f = c->ks->d_rlk_f; q.pub_b = c->ks->d_pub; q.crs = c->ks->d_crs;
    q.acc = acc; q.scratch = scratch; q.dev_order = c->dev_order;
    Timer tm(c, 4);
    HIPCHK(c, mktd::launch_kms_phase2(c->logM, p.W, q, B, c->stream));
    return MKT_OK;
}

int do_keyswitch(mkt_ctx *c, const void *acc, uint32_t *out, size_t B) {
    const mkt_params &p = c->p;
    mktd::KsArgs a{};
    a.acc = acc; a.out = out; a.ksk = c->ks->d_ksk; a.ksk_party_stride = c->ks->ksk_party_words; a.n1p = c->ks->n1p;
    a.N = p.N; a.n = p.n; a.f = p.f; a.logD = p.logD; a.drows = c->sh.ksk_drows; a.kacc = c->sh.kacc;
    a.mk = mkt::is_mk(p.scheme) ? 1 : 0; a.balanced = mkt::is_block(p.scheme) ? 1 : 0; a.lmss = p.scheme == MKT_LMSS ? 1 : 0;
    size_t dw = 0, pw = 0;
    mktd::ks_scratch_words(a, B, &dw, &pw);
    if (dw + pw > c->ws_ksd_words) {
        if (c->ws_ksd) (void)hipFree(c->ws_ksd);
        c->ws_ksd = nullptr; c->ws_ksd_words = 0;
        HIPCHK(c, hipMalloc((void **)&c->ws_ksd, (dw + pw) * 4));
        c->ws_ksd_words = dw + pw;
    }
    if (dw) { a.digits = c->ws_ksd; a.partial = c->ws_ksd + dw; }
    Timer tm(c, 2);
    HIPCHK(c, mktd::launch_keyswitch(p.W, a, B, c->stream));
    return MKT_OK;
}

// bootstrapping.jl:8-24 (mod-switch, test vector, blindrotate!) for a device-resident chunk of linear combinations: ws_acc <- accumulators
int rotate_chunk(mkt_ctx *c, const uint32_t *lin, size_t B) {
    const mkt_params &p = c->p;
    if (!mkt::is_kms(p.scheme)) {
        HIPCHK(c, mktd::launch_testvector(p.W, lin, c->sh.lwe_len, c->logN, c->sh.kacc, c->ws_acc, B, c->stream));
        return do_blindrotate(c, lin, c->sh.lwe_len, 0, nullptr, c->ws_acc, c->ws_lev, c->ws_scratch, B);
    }
    return do_blindrotate(c, lin, c->sh.lwe_len, 0, lin, c->ws_acc, c->ws_lev, c->ws_scratch, B);
}

// bootstrapping!(lin) -> out for a device-resident chunk
int bootstrap_chunk(mkt_ctx *c, const uint32_t *lin, uint32_t *out, size_t B) {
    int r;
    if ((r = rotate_chunk(c, lin, B))) return r;
    return do_keyswitch(c, c->ws_acc, out, B);
}

// staging helper for MKT_MEM_HOST callers
struct Staged {
    mkt_ctx *c; void *dev = nullptr; void *host_out = nullptr; size_t bytes = 0; bool owned = false;
    int in(const void *ptr, size_t nbytes, int mem, bool copy_in) {
        bytes = nbytes;
        if (mem == MKT_MEM_DEVICE) { dev = const_cast<void *>(ptr); return MKT_OK; }
        hipError_t e = hipMalloc(&dev, nbytes ? nbytes : 1);
        if (e != hipSuccess) return hipfail(c, e, "hipMalloc(staging)");
        owned = true;
        if (copy_in && nbytes) { e = hipMemcpyAsync(dev, ptr, nbytes, hipMemcpyHostToDevice, c->stream); if (e != hipSuccess) return hipfail(c, e, "H2D"); }
        return MKT_OK;
    }
    int out(void *ptr) {
        if (!owned || !bytes) return MKT_OK;
        hipError_t e = hipMemcpyAsync(ptr, dev, bytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return hipfail(c, e, "D2H");
        return MKT_OK;
    }
    ~Staged() { if (owned && dev) { (void)hipStreamSynchronize(c->stream); (void)hipFree(dev); } }
};

bool mem_ok(int mem) { return mem == MKT_MEM_DEVICE || mem == MKT_MEM_HOST; }

// ---- MKT_ARITH_EXACT: tables of the two-prime negacyclic NTT (ntt_exact.hip), computed on the host, uploaded once ----
constexpr uint32_t NTT_P[2] = {1073668097u, 1073692673u};        // 131063 * 2^13 + 1, 131066 * 2^13 + 1: the two largest NTT primes below 2^30 (ntt_exact.hip)
uint32_t ntt_mulmod(uint32_t a, uint32_t b, uint32_t p) { return (uint32_t)((uint64_t)a * b % p); }
uint32_t ntt_powmod(uint32_t a, uint64_t e, uint32_t p) { uint32_t r = 1; while (e) { if (e & 1) r = ntt_mulmod(r, a, p); a = ntt_mulmod(a, a, p); e >>= 1; } return r; }
uint32_t ntt_shoup(uint32_t w, uint32_t p) { return (uint32_t)(((uint64_t)w << 32) / p); }

// per point (w mod p1, companion, w mod p2, companion; the table holds 2^32 - w): psi_rev[N] | N^-1, N^-1 w | N^-1 2^32, N^-1 2^32 w, for the transform of
// size N (no inverse table: psiinv_rev[m + i] = -psi_rev[2m - 1 - i], which the inverse butterflies read off the forward table, ntt_exact.hip bfly_inv); psi = g^((p - 1) / 2N) with g the smallest quadratic non-residue of p (so psi^N = -1: a primitive 2N-th root of unity)
int upload_ntt_tables(mkt_ctx *c) {
    const int N = c->p.N, logN = c->logN;
    std::vector<uint32_t> tab((size_t)(N + 4) * 4);
    for (int k = 0; k < 2; k++) {
        const uint32_t p = NTT_P[k];
        uint32_t g = 2;
        while (ntt_powmod(g, (p - 1) / 2, p) != p - 1) g++;
        const uint32_t psi = ntt_powmod(g, (p - 1) / (2 * (uint64_t)N), p), psiinv = ntt_powmod(psi, p - 2, p);
        if (ntt_powmod(psi, (uint64_t)N, p) != p - 1) return fail(c, MKT_ERR_UNSUPPORTED, "no primitive 2N-th root of unity for this ring dimension");
        for (int i = 0; i < N; i++) {
            int r = 0;
            for (int b = 0; b < logN; b++) r |= ((i >> b) & 1) << (logN - 1 - b);
            const uint32_t w = ntt_powmod(psi, (uint64_t)r, p);
            tab[(size_t)i * 4 + 2 * k] = 0u - w; tab[(size_t)i * 4 + 2 * k + 1] = ntt_shoup(w, p);   // the NEGATED twiddle (ntt_exact.hip bfly_fwd, bfly_inv)
        }
        // N^-1 and N^-1 2^32, each followed by its product with the one twiddle of the inverse's last stage (psiinv_rev[1])
        const uint32_t ninv = ntt_powmod((uint32_t)N, p - 2, p), ninv_r = (uint32_t)(((uint64_t)ninv << 32) % p);
        const uint32_t wlast = ntt_powmod(psiinv, (uint64_t)N / 2, p);                    // bitrev(1) = N / 2
        const uint32_t cs[4] = {ninv, ntt_mulmod(ninv, wlast, p), ninv_r, ntt_mulmod(ninv_r, wlast, p)};
        for (int q = 0; q < 4; q++) { tab[(size_t)(N + q) * 4 + 2 * k] = cs[q]; tab[(size_t)(N + q) * 4 + 2 * k + 1] = ntt_shoup(cs[q], p); }
    }
    HIPCHK(c, hipMalloc((void **)&c->ks->d_ntt, tab.size() * 4));
    c->d_ntt = c->ks->d_ntt;
    HIPCHK(c, hipMemcpy(c->d_ntt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    return MKT_OK;
}
// the gate path of an EXACT context: CGGI with RLWE length 1 on the 32-bit ring (every true product coefficient < p / 2)
// (every true product coefficient below P / 2 = 2^58.9998: 2l polynomials of N digits of magnitude <= 2^(logB-1) against 32-bit words)
bool exact_gate_ok(const mkt_ctx *c) {
    const double half_P = 0.5 * (double)NTT_P[0] * (double)NTT_P[1];
    const mkt_params &p = c->p;
    // ring words and their 32-bit pieces enter as centered integers (magnitude <= 2^31: ntt_exact.hip res_word / piece_of)
    const double n31 = (double)p.N * 2147483648.0;
    if (mkt::is_kms(p.scheme) && p.W == 64) {
        // 64-bit ring: tables split into 32-bit halves, every accumulated product sum of one half must stay below P / 2:
        // phase 1 (KMS: the sum of the 2l products, the monomial X^a - 1 is applied after the lift; KMS_block: a block sums
        // its key bits' products times their monomials before the inverse), the LEV multiplication + relinearisation sums,
        // the v sum over the parties
        const double ph1 = (p.scheme == MKT_KMS_BLOCK ? 2.0 * p.blk_len : 1.0) * 2.0 * p.l_gsw * std::ldexp(1.0, p.logB_gsw - 1) * n31;
        const double acc = (p.l_lev * std::ldexp(1.0, p.logB_lev - 1) + 2.0 * p.l_uni * std::ldexp(1.0, p.logB_uni - 1)) * n31;
        const double tv = (double)p.k * p.l_uni * std::ldexp(1.0, p.logB_uni - 1) * n31;
        return ph1 < half_P && acc < half_P && tv < half_P;
    }
    if (p.scheme == MKT_CCS && p.W == 32)    // tacc.b gathers u_0 and the w of all np + 1 polynomials, then the monomial doubles it
        return 2.0 * (p.k + 2.0) * p.l_uni * std::ldexp(1.0, p.logB_uni - 1) * n31 < half_P;
    const bool lmss = p.scheme == MKT_LMSS;
    if (!((p.scheme == MKT_CGGI || lmss) && p.k >= 1 && p.W == 32)) return false;     // any RLWE length (exact_blindrotate_kr_kernel beyond the k = 1, block-length-3 shapes; exact_blindrotate_kany_kernel beyond k = 3)
    // the kernels multiply the product sum by the monomial X^a - 1 in the transform domain BEFORE the one lift (ntt_exact.hip
    // exact_blindrotate_kernel: s2 = tacc * mono), so the lifted integer is up to twice the sum -- for CGGI as for a block;
    // (k + 1) l digit polynomials per key bit
    const double bound = 2.0 * (lmss ? p.blk_len : 1.0) * (p.k + 1.0) * p.l_gsw * std::ldexp(1.0, p.logB_gsw - 1) * n31;
    return bound < half_P;
}
#define MKT_EXACT_GATE(c) do { if ((c) && (c)->exact && !exact_gate_ok(c)) return fail((c), MKT_ERR_UNSUPPORTED, "MKT_ARITH_EXACT evaluates gates for CGGI and LMSS (32-bit ring), for CCS (32-bit ring) and for KMS / KMS_block (64-bit ring, tables split in 32-bit halves), gadgets within the two-prime modulus; other schemes offer the transform-level entry points (mkt_transform_*_batch, mkt_exact_polymul_batch)"); } while (0)
#define MKT_F64_OR_EXACT_KMS(c) do { if ((c) && (c)->exact && !(mkt::is_mk((c)->p.scheme) && exact_gate_ok(c))) return fail((c), MKT_ERR_UNSUPPORTED, "on an MKT_ARITH_EXACT context this entry point serves the multi-key gate paths only"); } while (0)
#define MKT_F64_ONLY(c) do { if ((c) && (c)->exact) return fail((c), MKT_ERR_UNSUPPORTED, "this entry point is the Float64-reference gate path; not offered by an MKT_ARITH_EXACT context"); } while (0)

// ---- MKT_ARITH_EXACT on the Float64 pipe (fx_exact.hip) ----
// Which shapes keep a second copy of the bootstrapping key as limb transforms: CGGI with RLWE length 1 and KMS (phase 1), gadget lengths the kernel holds in registers.
bool fx_shape(const mkt_ctx *c) {
    const mkt_params &p = c->p;
    return c->exact && ((p.scheme == MKT_CGGI && p.k == 1) || p.scheme == MKT_KMS) && mktd::fx_supported(c->logM, p.W, p.l_gsw);
}
// Proven bound on |computed coefficient - exact integer| of one rounded sum  sum_g d_g (*) limb_g  (DESIGN.md section 2), u = 2^-53:
//   transform-domain errors reach a coefficient through the 1-norm:  (gamma_f + gamma_k + gamma_m) sum_g |d_g|_2 |limb_g|_2
//   the inverse's own roundings are relative to the 2-norm of what it transforms:  gamma_i sum_g |d_g|_2 max_r |K_g[r]|
// with |d_g|_2 <= sqrt(N) 2^(logB-1), |limb_g|_2 <= sqrt(N) 2^15, max_r |K[r]| MEASURED over the loaded key (kmax; a random key sits near
// 4 sqrt(N) 2^15 / sqrt(3), an adversarial one at N 2^15 fails the bound and the integer NTT serves), per-stage constants 5.5 u (6-operation
// butterfly incl. the rounded twiddle), 1.5 u (product-free stages), 3 u (twist / untwist), (2 + 6 l) u for the multiply-add chain.
double fx_bound(const mkt_ctx *c, double kmax) {
    const mkt_params &p = c->p;
    const double u = std::ldexp(1.0, -53), g2 = 2.0 * p.l_gsw;
    const double gt = (3.0 + 5.5 * (c->logM - 2) + 1.5 * 2) * u, gm = (2.0 + 3.0 * g2) * u;     // (chain of 2 g2 fused operations per component, sqrt(2) for the complex value: 2.83 g2 u)
    const double dn = std::sqrt((double)p.N) * std::ldexp(1.0, p.logB_gsw - 1), kn = std::sqrt((double)p.N) * 32768.0;
    return (gt + (gt + u) + gm) * g2 * dn * kn + gt * g2 * dn * kmax * (1.0 + 1e-6);
}
bool fx_usable(const mkt_ctx *c) {
    if (!c->ks->d_fx_brk || c->tune.exact_impl == 0 || c->ks->fx_kmax <= 0.0) return false;
    const mkt_params &p = c->p;
    if (2.0 * p.l_gsw * p.N * std::ldexp(1.0, p.logB_gsw - 1) * 32768.0 >= std::ldexp(1.0, 50)) return false;   // the rounding trick holds integers below 2^51
    return fx_bound(c, c->ks->fx_kmax) < 0.45;
}
int fx_after_key_load(mkt_ctx *c) {   // the key's largest transform magnitude, for fx_bound
    unsigned long long bits = 0;
    HIPCHK(c, hipMemcpy(&bits, c->ks->d_fx_stat, 8, hipMemcpyDeviceToHost));
    double v; std::memcpy(&v, &bits, 8);
    c->ks->fx_kmax = std::sqrt(v);
    return MKT_OK;
}
mktd::FxRotArgs fx_rot_args(mkt_ctx *c, const uint32_t *lwe, int stride, int pre) {
    const mkt_params &p = c->p;
    mktd::FxRotArgs q{};
    q.om = c->fx_om(); q.twist = c->fx_tw(); q.nat = c->fx_nat(); q.brk = c->ks->d_fx_brk; q.brk_party_stride = c->ks->fx_brk_party_cplx;
    q.lwe = lwe; q.lwe_stride = stride; q.pre_switched = pre; q.n = p.n; q.logN = c->logN; q.l = p.l_gsw; q.logB = p.logB_gsw;
    q.rows_per_gate = c->ks->rtot; q.slot_party = c->ks->d_slot_party; q.slot_row = c->ks->d_slot_row; q.logB_lev = p.logB_lev;
    q.stagger = c->tune.rot_stagger; q.map_mode = c->tune.rot_map; q.split = c->tune.rot_split;
    return q;
}

}  // namespace

extern "C" {

int mkt_abi_version(void) { return MKT_ABI_VERSION; }

const char *mkt_last_error(const mkt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int mkt_ctx_create(const mkt_params *params, int arith_mode, int device, mkt_ctx **out) {
    if (!params || !out) return fail(nullptr, MKT_ERR_ARG, "null argument");
    *out = nullptr;
    std::string why;
    if (mkt::validate_params(*params, why)) return fail(nullptr, MKT_ERR_ARG, why);
    if (arith_mode != MKT_ARITH_F64REF && arith_mode != MKT_ARITH_EXACT) return fail(nullptr, MKT_ERR_ARG, "unknown arithmetic mode");
    const int logN = __builtin_ctz((unsigned)params->N);
    if (!mktd::transform_supported(logN - 1)) return fail(nullptr, MKT_ERR_UNSUPPORTED, "ring dimension not instantiated (N must be 32..4096)");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(nullptr, MKT_ERR_NO_DEVICE, "no HIP device available: the engine has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(nullptr, MKT_ERR_ARG, "device index out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, MKT_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, MKT_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", the engine is built for gfx950 only");

    auto *c = new mkt_ctx();
    c->ks = std::make_shared<KeySet>();
    c->ks->device = device;
    c->p = *params; c->sh = mkt::shape_of(*params); c->device = device;
    c->logN = logN; c->logM = logN - 1; c->M = params->N / 2;
    c->exact = arith_mode == MKT_ARITH_EXACT;
    c->split = (c->exact && params->W == 64) ? 2 : 1;
    // the RLWE-length-k kernels of the plain schemes want the slot-pair order, everything else the slot-major one (fft_device.h)
    c->dev_order = ((params->scheme == MKT_CGGI || params->scheme == MKT_LMSS) && params->k > 1) ? MKT_DEVORDER_KR : MKT_DEVORDER;
    c->tune.from_env();   // the one place the MKT_ROT_* / MKT_CCS_* environment is read; mkt_set_option afterwards
    DevGuard dg(device);
    auto bail = [&](int code) { std::string m = c->err; mkt_ctx_destroy(c); g_create_error = m; return code; };
    if (!dg.ok) { c->err = "hipSetDevice failed"; return bail(MKT_ERR_HIP); }
    const mkt_params &p = c->p;
    const int np = c->sh.nparty, N = p.N, M = c->M;
    c->ks->brk_loaded.assign(np, 0); c->ks->ksk_loaded.assign(np, 0); c->ks->rlk_loaded.assign(np, 0); c->ks->pub_loaded.assign(np, 0);
    mkt::make_twiddles(N, c->ks->tw);
#define CK(call) do { hipError_t _e = (call); if (_e != hipSuccess) { c->err = std::string(#call) + ": " + hipGetErrorString(_e); return bail(_e == hipErrorOutOfMemory ? MKT_ERR_NOMEM : MKT_ERR_HIP); } } while (0)
    CK(hipMalloc((void **)&c->ks->d_tw, (size_t)4 * M * sizeof(cplx)));
    CK(hipMalloc((void **)&c->ks->d_monomial, (size_t)2 * N * M * sizeof(cplx)));
    c->ks->brk_party_cplx = (size_t)p.n * c->sh.brk_polys * M * c->split;
    // the rotation kernels read a party's key rows through one buffer descriptor (kernel_common.h table_rsrc: 31-bit record
    // count, 32-bit row offsets); a larger key would read zeros silently, so it is refused here (largest shipped set: 0.25 GB)
    if (c->ks->brk_party_cplx * sizeof(cplx) > 0x7fffffffull) { c->err = "per-party bootstrapping key exceeds the 2 GiB window of the rotation kernels' buffer descriptors"; return bail(MKT_ERR_UNSUPPORTED); }
    CK(hipMalloc((void **)&c->ks->d_brk, (size_t)np * c->ks->brk_party_cplx * sizeof(cplx)));
    c->ks->n1p = (p.n + 1 + 3) / 4 * 4;   // device rows padded to 16 B
    c->ks->ksk_party_words = (size_t)c->sh.ksk_kr * N * c->sh.ksk_drows * p.f * c->ks->n1p;
    CK(hipMalloc((void **)&c->ks->d_ksk, (size_t)np * c->ks->ksk_party_words * sizeof(uint32_t)));
    if (mkt::is_mk(p.scheme)) {
        CK(hipMalloc((void **)&c->ks->d_pub, (size_t)np * p.l_uni * M * sizeof(cplx) * c->split));
        CK(hipMalloc((void **)&c->ks->d_crs, (size_t)p.l_uni * M * sizeof(cplx) * c->split));
    }
    if (mkt::is_kms(p.scheme)) {
        CK(hipMalloc((void **)&c->ks->d_rlk_d, (size_t)np * p.l_uni * M * sizeof(cplx) * c->split));
        CK(hipMalloc((void **)&c->ks->d_rlk_f, (size_t)np * p.l_uni * 2 * M * sizeof(cplx) * c->split));
    }
    // rotation slots: KMS phase 1 runs 1 row for party 0 and l_lev rows for the others (bootstrapping.jl:400)
    std::vector<int> sp, sr;
    if (mkt::is_kms(p.scheme)) {
        for (int i = 0; i < p.k; i++) { int rows = i == 0 ? 1 : p.l_lev; for (int r = 0; r < rows; r++) { sp.push_back(i); sr.push_back(r); } }
    } else { sp.push_back(0); sr.push_back(0); }
    c->ks->rtot = (int)sp.size();
    CK(hipMalloc((void **)&c->ks->d_slot_party, sp.size() * sizeof(int)));
    CK(hipMalloc((void **)&c->ks->d_slot_row, sr.size() * sizeof(int)));
    CK(hipMemcpy(c->ks->d_slot_party, sp.data(), sp.size() * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemcpy(c->ks->d_slot_row, sr.data(), sr.size() * sizeof(int), hipMemcpyHostToDevice));
    if (fx_shape(c) && c->tune.exact_impl != 0) {   // second copy of the bootstrapping key as limb transforms + the engine's own tables
        const size_t per = (size_t)p.n * 2 * p.l_gsw * 2 * (p.W / 16) * M;
        if (per * sizeof(cplx) <= 0x7fffffffull) {
            c->ks->fx_brk_party_cplx = per;
            CK(hipMalloc((void **)&c->ks->d_fx_tab, (size_t)3 * M * sizeof(cplx)));
            CK(hipMalloc((void **)&c->ks->d_fx_brk, (size_t)np * per * sizeof(cplx)));
            CK(hipMalloc((void **)&c->ks->d_fx_stat, 16));
            CK(hipMemset(c->ks->d_fx_stat, 0, 16));
            const size_t tb = (size_t)M * sizeof(cplx);
            CK(hipMemcpy(c->ks->d_fx_tab, c->ks->tw.fx_om.data(), tb, hipMemcpyHostToDevice));
            CK(hipMemcpy(c->ks->d_fx_tab + M, c->ks->tw.fx_tw.data(), tb, hipMemcpyHostToDevice));
            CK(hipMemcpy(c->ks->d_fx_tab + 2 * (size_t)M, c->ks->tw.fx_nat.data(), tb, hipMemcpyHostToDevice));
        }
    }
#undef CK
    int r = upload_twiddles(c);
    if (!r && c->exact) r = upload_ntt_tables(c);
    if (!r) r = build_monomial(c);
    if (r) return bail(r);
    *out = c;
    return MKT_OK;
}

int mkt_ctx_destroy(mkt_ctx *c) {
    if (!c) return MKT_OK;
    DevGuard dg(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->own_stream && c->own_stream != c->stream) (void)hipStreamSynchronize(c->own_stream);   // before the workspace goes: work queued on the fork's own stream may still use it
    clear_spans(c);
    void *ptrs[] = {c->ws_lin, c->ws_acc, c->ws_lev, c->ws_scratch, c->ws_ksd, c->ws_fxacc};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;                      // drops this context's reference to the key set; the last one frees it
    return MKT_OK;
}

// A second context over the SAME resident keys and tables (no copy): own stream, own workspace, own timing -- one per
// concurrent caller / host thread / stream, as the reference's read-only scheme object is shared by concurrent
// bootstrapping! calls.  From the first fork on the key set is immutable (mkt_load_*, mkt_set_twiddles,
// mkt_keygen_device return MKT_ERR_STATE on every context that shares it).
int mkt_ctx_fork(mkt_ctx *c, mkt_ctx **out) {
    if (!c || !out) return fail(c, MKT_ERR_ARG, "null argument");
    auto *f = new mkt_ctx();
    f->p = c->p; f->sh = c->sh; f->device = c->device; f->logM = c->logM; f->logN = c->logN; f->M = c->M; f->dev_order = c->dev_order;
    f->ks = c->ks; f->exact = c->exact; f->split = c->split; f->d_ntt = c->d_ntt; f->tune = c->tune;
    // the fork's own stream: non-blocking, so forks driven from several host threads neither serialise on the NULL stream
    // nor against each other; mkt_set_stream may re-point the context at a caller's stream later
    {
        DevGuard dg(c->device);
        hipError_t e = hipStreamCreateWithFlags(&f->own_stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete f; return hipfail(c, e, "hipStreamCreateWithFlags(fork)"); }
        f->stream = f->own_stream;
    }
    *out = f;
    return MKT_OK;
}

// ---- internal (multi.cpp): replicate the resident, pre-transformed key set of `src` onto `dst`'s device ----
// dst is a fresh context of the same parameters and arithmetic on another device.  Device-to-device with hipMemcpyPeer (xGMI
// when the devices are linked; the runtime stages through the host otherwise); if the peer copy is refused, an explicit host
// bounce.  The key upload and its transforms run ONCE, on src's device ("optional one-time device-to-device key copy").
static int copy_across(mkt_ctx *dst, void *d, int ddev, const void *s_, int sdev, size_t bytes, bool no_peer) {
    if (!bytes) return MKT_OK;
    if (!no_peer) {
        if (hipMemcpyPeer(d, ddev, s_, sdev, bytes) == hipSuccess) return MKT_OK;
        (void)hipGetLastError();
    }
    std::vector<unsigned char> bounce(bytes);
    { DevGuard g(sdev); HIPCHK(dst, hipMemcpy(bounce.data(), s_, bytes, hipMemcpyDeviceToHost)); }
    { DevGuard g(ddev); HIPCHK(dst, hipMemcpy(d, bounce.data(), bytes, hipMemcpyHostToDevice)); }
    return MKT_OK;
}
int mkt_internal_clone_keys(mkt_ctx *src, mkt_ctx *dst, int no_peer) {
    if (!src || !dst) return MKT_ERR_ARG;
    if (std::memcmp(&src->p, &dst->p, sizeof(mkt_params)) != 0 || src->exact != dst->exact) return fail(dst, MKT_ERR_ARG, "key replication between contexts of different parameters");
    if (dst->keys_shared()) return fail(dst, MKT_ERR_STATE, "the key set is shared with forked contexts and immutable");
    { DevGuard g(src->device); HIPCHK(dst, hipStreamSynchronize(src->stream)); }
    const mkt_params &p = src->p;
    const int np = src->sh.nparty, M = src->M, N = p.N, sd = src->device, dd = dst->device;
    KeySet &a = *src->ks, &b = *dst->ks;
    const size_t cb = sizeof(cplx);
    int r;
    // tables a caller may have replaced on src (mkt_set_twiddles), and the monomial table that depends on them
    b.tw = a.tw;
    if ((r = copy_across(dst, b.d_tw, dd, a.d_tw, sd, (size_t)4 * M * cb, no_peer != 0))) return r;
    if ((r = copy_across(dst, b.d_monomial, dd, a.d_monomial, sd, (size_t)2 * N * M * cb, no_peer != 0))) return r;
    if ((r = copy_across(dst, b.d_brk, dd, a.d_brk, sd, (size_t)np * a.brk_party_cplx * cb, no_peer != 0))) return r;
    if ((r = copy_across(dst, b.d_ksk, dd, a.d_ksk, sd, (size_t)np * a.ksk_party_words * 4, no_peer != 0))) return r;
    if (mkt::is_mk(p.scheme)) {
        if ((r = copy_across(dst, b.d_pub, dd, a.d_pub, sd, (size_t)np * p.l_uni * M * cb * src->split, no_peer != 0))) return r;
        if ((r = copy_across(dst, b.d_crs, dd, a.d_crs, sd, (size_t)p.l_uni * M * cb * src->split, no_peer != 0))) return r;
    }
    if (mkt::is_kms(p.scheme)) {
        if ((r = copy_across(dst, b.d_rlk_d, dd, a.d_rlk_d, sd, (size_t)np * p.l_uni * M * cb * src->split, no_peer != 0))) return r;
        if ((r = copy_across(dst, b.d_rlk_f, dd, a.d_rlk_f, sd, (size_t)np * p.l_uni * 2 * M * cb * src->split, no_peer != 0))) return r;
    }
    if (a.d_fx_brk && b.d_fx_brk) {
        if ((r = copy_across(dst, b.d_fx_brk, dd, a.d_fx_brk, sd, (size_t)np * a.fx_brk_party_cplx * cb, no_peer != 0))) return r;
        if ((r = copy_across(dst, b.d_fx_stat, dd, a.d_fx_stat, sd, 16, no_peer != 0))) return r;
        b.fx_kmax = a.fx_kmax;
    }
    b.brk_loaded = a.brk_loaded; b.ksk_loaded = a.ksk_loaded; b.rlk_loaded = a.rlk_loaded; b.pub_loaded = a.pub_loaded; b.crs_loaded = a.crs_loaded;
    dst->tune = src->tune;
    // hipMemcpyPeer may return before the copy has landed, and the shards evaluate on non-blocking streams that the NULL stream does
    // not order: both devices are drained before the replica may be used (one-time cost, off the evaluation path)
    { DevGuard g(sd); HIPCHK(dst, hipDeviceSynchronize()); }
    { DevGuard g(dd); HIPCHK(dst, hipDeviceSynchronize()); }
    return MKT_OK;
}
#ifndef MKT_BUILD_ID
#define MKT_BUILD_ID "unknown"
#endif
const char *mkt_build_id(void) { return MKT_BUILD_ID; }
int mkt_internal_device_of(const mkt_ctx *c) { return c ? c->device : -1; }
size_t mkt_internal_lwe_len(const mkt_ctx *c) { return c ? (size_t)c->sh.lwe_len : 0; }
size_t mkt_internal_acc_bytes(const mkt_ctx *c) { return c ? (size_t)(1 + c->sh.kacc) * poly_bytes(c) : 0; }

int mkt_set_stream(mkt_ctx *c, void *hip_stream) { if (!c) return MKT_ERR_ARG; c->stream = (hipStream_t)hip_stream; return MKT_OK; }

int mkt_get_stream(mkt_ctx *c, void **hip_stream) { if (!c || !hip_stream) return MKT_ERR_ARG; *hip_stream = (void *)c->stream; return MKT_OK; }

// kernel-selection switches (parity tests force every kernel variant through this; A/B tools may seed them from the
// MKT_ROT_* / MKT_CCS_* environment, which is read once at mkt_ctx_create)
int mkt_set_option(mkt_ctx *c, const char *name, int value) {
    if (!c || !name) return fail(c, MKT_ERR_ARG, "null argument");
    const std::string k(name);
    Tune &t = c->tune;
    if (k == "rot_variant") t.rot_variant = value;
    else if (k == "rot_stagger") t.rot_stagger = value;
    else if (k == "rot_split") t.rot_split = value;
    else if (k == "rot_wide") t.rot_wide = value;
    else if (k == "rot_blkg") t.rot_blkg = value;
    else if (k == "ccs_stagger") t.ccs_stagger = value;
    else if (k == "ccs_pipe") t.ccs_pipe = value;
    else if (k == "exact_wide") t.exact_wide = value;
    else if (k == "exact_impl") t.exact_impl = value;
    else if (k == "rot_map") t.rot_map = value;
    else if (k == "exact_kany") { if (t.exact_kany != value) c->ws_gates = 0; t.exact_kany = value; }   // the workspace gains / loses the kernel's scratch at the next call
    else return fail(c, MKT_ERR_ARG, "mkt_set_option: unknown option '" + k + "'");
    return MKT_OK;
}

const char *mkt_last_kernel_name(const mkt_ctx *c) { return c ? c->last_rot_kernel : ""; }

// diagnostics of the Float64-pipe EXACT implementation (fx_exact.hip): "fx_available" (1 if the loaded keys are certified and exact_impl admits it),
// "fx_bound" (proven bound on |computed - exact| of a rounded sum for the loaded keys), "fx_kmax" (largest |key transform value|), "fx_last_resid"
int mkt_get_metric(mkt_ctx *c, const char *name, double *out) {
    if (!c || !name || !out) return fail(c, MKT_ERR_ARG, "null argument");
    const std::string k(name);
    if (k == "fx_available") *out = fx_usable(c) ? 1.0 : 0.0;
    else if (k == "fx_bound") *out = c->ks->d_fx_brk ? fx_bound(c, c->ks->fx_kmax) : -1.0;
    else if (k == "fx_kmax") *out = c->ks->fx_kmax;
    else if (k == "fx_last_resid") *out = c->fx_last_resid;
    else return fail(c, MKT_ERR_ARG, "mkt_get_metric: unknown metric '" + k + "'");
    return MKT_OK;
}

int mkt_synchronize(mkt_ctx *c) {
    if (!c) return MKT_ERR_ARG;
    DevGuard dg(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MKT_OK;
}

int mkt_get_twiddles(mkt_ctx *c, int which, double *out_host) {
    if (!c || !out_host || which < 0 || which > 3) return fail(c, MKT_ERR_ARG, "bad argument");
    MKT_F64_ONLY(c);
    DevGuard dg(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out_host, c->ks->d_tw + (size_t)which * c->M, (size_t)c->M * sizeof(cplx), hipMemcpyDeviceToHost));
    return MKT_OK;
}

int mkt_set_twiddles(mkt_ctx *c, const double *psi, const double *psiinv, const double *roots, const double *rootsinv) {
    if (!c || !psi || !psiinv || !roots || !rootsinv) return fail(c, MKT_ERR_ARG, "null table");
    MKT_F64_ONLY(c);
    if (c->keys_shared()) return fail(c, MKT_ERR_STATE, "the key set is shared with forked contexts and immutable");
    DevGuard dg(c->device);
    const size_t nd = (size_t)2 * c->M;
    // the kernels derive the inverse twiddles from the forward table: Psiinv must be conj(Psi) entry for entry,
    // which holds for the reference's tables (fft.jl:33-34: exp(-i*theta) and exp(+i*theta) of the same theta)
    for (int i = 1; i < c->M; i++)
        if (std::memcmp(&psi[2 * i], &psiinv[2 * i], 8) != 0 || psiinv[2 * i + 1] != -psi[2 * i + 1])
            return fail(c, MKT_ERR_ARG, "mkt_set_twiddles: Psiinv is not the conjugate of Psi");
    if (!twiddle_shape_ok(std::vector<double>(psi, psi + nd), c->M))   // checked BEFORE the host tables are replaced: a refused call leaves the context as it was
        return fail(c, MKT_ERR_ARG, "twiddle table Psi does not have the reference's shape (Psi[1] = (eps,-1), Psi[2] = (c,-c), Psi[3] = (-c,-c))");
    c->ks->tw.psi.assign(psi, psi + nd); c->ks->tw.psiinv.assign(psiinv, psiinv + nd);
    c->ks->tw.roots.assign(roots, roots + nd); c->ks->tw.rootsinv.assign(rootsinv, rootsinv + nd);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int r = upload_twiddles(c);
    if (r) return r;
    return build_monomial(c);   // the monomial table depends on the tables (scheme.jl:121-146)
}

int mkt_get_monomial(mkt_ctx *c, int e, double *out_host) {
    if (!c || !out_host || e < 1 || e > 2 * c->p.N) return fail(c, MKT_ERR_ARG, "bad argument");
    MKT_F64_ONLY(c);
    DevGuard dg(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<cplx> dev((size_t)c->M);
    HIPCHK(c, hipMemcpy(dev.data(), c->ks->d_monomial + (size_t)(e - 1) * c->M, (size_t)c->M * sizeof(cplx), hipMemcpyDeviceToHost));
    const int NT = c->M >> MKT_LOGR;              // device order -> the reference's order
    cplx *o = reinterpret_cast<cplx *>(out_host);
    for (int x = 0; x < c->M; x++) o[x] = dev[(size_t)mktd::dev_pos(c->dev_order, x, NT)];
    return MKT_OK;
}

int mkt_load_brk(mkt_ctx *c, int party, const void *data, int fmt) {
    if (!c || !data || party < 0 || party >= c->sh.nparty) return fail(c, MKT_ERR_ARG, "bad argument");
    MKT_EXACT_GATE(c);
    if (c->keys_shared()) return fail(c, MKT_ERR_STATE, "the key set is shared with forked contexts and immutable");
    DevGuard dg(c->device);
    int r = upload_polys(c, data, (size_t)c->p.n * c->sh.brk_polys, c->ks->d_brk + (size_t)party * c->ks->brk_party_cplx, fmt, false,
                         c->ks->d_fx_brk ? c->ks->d_fx_brk + (size_t)party * c->ks->fx_brk_party_cplx : nullptr);
    if (!r) c->ks->brk_loaded[party] = 1;
    return r;
}

int mkt_load_ksk(mkt_ctx *c, int party, const uint32_t *data) {
    if (!c || !data || party < 0 || party >= c->sh.nparty) return fail(c, MKT_ERR_ARG, "bad argument");
    MKT_EXACT_GATE(c);
    if (c->keys_shared()) return fail(c, MKT_ERR_STATE, "the key set is shared with forked contexts and immutable");
    DevGuard dg(c->device);
    const size_t rows = (size_t)c->sh.ksk_kr * c->p.N * c->sh.ksk_drows * c->p.f, n1 = (size_t)c->p.n + 1;
    HIPCHK(c, hipMemset(c->ks->d_ksk + (size_t)party * c->ks->ksk_party_words, 0, c->ks->ksk_party_words * sizeof(uint32_t)));
    HIPCHK(c, hipMemcpy2D(c->ks->d_ksk + (size_t)party * c->ks->ksk_party_words, (size_t)c->ks->n1p * 4, data, n1 * 4, n1 * 4, rows, hipMemcpyHostToDevice));
    c->ks->ksk_loaded[party] = 1;
    return MKT_OK;
}

int mkt_load_rlk(mkt_ctx *c, int party, const void *d, const void *f, int fmt) {
    if (!c || !d || !f || party < 0 || party >= c->sh.nparty || !mkt::is_kms(c->p.scheme)) return fail(c, MKT_ERR_ARG, "bad argument");
    MKT_F64_OR_EXACT_KMS(c);
    if (c->keys_shared()) return fail(c, MKT_ERR_STATE, "the key set is shared with forked contexts and immutable");
    DevGuard dg(c->device);
    const size_t l = (size_t)c->p.l_uni;
    int r = upload_polys(c, d, l, c->ks->d_rlk_d + (size_t)party * l * c->M * c->split, fmt);
    if (!r) r = upload_polys(c, f, 2 * l, c->ks->d_rlk_f + (size_t)party * 2 * l * c->M * c->split, fmt);
    if (!r) c->ks->rlk_loaded[party] = 1;
    return r;
}

int mkt_load_pubkey(mkt_ctx *c, int party, const void *b, int fmt) {
    if (!c || !b || party < 0 || party >= c->sh.nparty || !mkt::is_mk(c->p.scheme)) return fail(c, MKT_ERR_ARG, "bad argument");
    MKT_F64_OR_EXACT_KMS(c);
    if (c->keys_shared()) return fail(c, MKT_ERR_STATE, "the key set is shared with forked contexts and immutable");
    DevGuard dg(c->device);
    int r = upload_polys(c, b, (size_t)c->p.l_uni, c->ks->d_pub + (size_t)party * c->p.l_uni * c->M * c->split, fmt);
    if (!r) c->ks->pub_loaded[party] = 1;
    return r;
}

int mkt_load_crs(mkt_ctx *c, const void *a, int fmt) {
    if (!c || !a || !mkt::is_mk(c->p.scheme)) return fail(c, MKT_ERR_ARG, "bad argument");
    MKT_F64_OR_EXACT_KMS(c);
    if (c->keys_shared()) return fail(c, MKT_ERR_STATE, "the key set is shared with forked contexts and immutable");
    DevGuard dg(c->device);
    int r = upload_polys(c, a, (size_t)c->p.l_uni, c->ks->d_crs, fmt);
    if (!r) c->ks->crs_loaded = true;
    return r;
}

// Bootstrapping key and key-switching key of party `party` generated on the device from the party's secrets
// (keygen.hip: the seeded streams of mkt_client_party_keygen, identical words), pre-transformed in place of an upload.
static int keygen_device_impl(mkt_ctx *c, int party, const mkt_client_party *K, const void *crs, void *brk_out, uint32_t *ksk_out) {
    if (!c || !K || party < 0 || party >= c->sh.nparty) return fail(c, MKT_ERR_ARG, "bad argument");
    MKT_EXACT_GATE(c);
    if (c->keys_shared()) return fail(c, MKT_ERR_STATE, "the key set is shared with forked contexts and immutable");
    const mkt_params &p = c->p;
    if (std::memcmp(&K->p, &p, sizeof(mkt_params)) != 0 || K->party != party) return fail(c, MKT_ERR_ARG, "mkt_keygen_device: the party's keys were made for other parameters / another party index");
    const bool unienc = p.scheme == MKT_CCS;
    if (unienc && !crs) return fail(c, MKT_ERR_ARG, "mkt_keygen_device: CCS needs the integer CRS");
    DevGuard dg(c->device);
    const int N = p.N, nz = (int)K->zring.size();
    uint32_t *d_lwe = nullptr; int8_t *d_z = nullptr; void *d_crs_int = nullptr, *d_out = nullptr;
    const size_t brk_polys_total = (size_t)p.n * c->sh.brk_polys;
    // the secrets are wiped on the device before their buffers are released
    auto cleanup = [&] {
        if (d_lwe) (void)hipMemsetAsync(d_lwe, 0, (size_t)p.n * 4, c->stream);
        if (d_z) (void)hipMemsetAsync(d_z, 0, (size_t)nz * N, c->stream);
        (void)hipStreamSynchronize(c->stream);
        (void)hipFree(d_lwe); (void)hipFree(d_z); (void)hipFree(d_crs_int); (void)hipFree(d_out);
    };
    hipError_t e = hipMalloc((void **)&d_lwe, (size_t)p.n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_z, (size_t)nz * N);
    if (e == hipSuccess) e = hipMalloc(&d_out, brk_polys_total * poly_bytes(c));
    if (e == hipSuccess && unienc) e = hipMalloc(&d_crs_int, (size_t)p.l_uni * poly_bytes(c));
    if (e == hipSuccess) e = hipMemcpyAsync(d_lwe, K->lwekey.data(), (size_t)p.n * 4, hipMemcpyHostToDevice, c->stream);
    for (int q = 0; q < nz && e == hipSuccess; q++) e = hipMemcpyAsync(d_z + (size_t)q * N, K->zring[q].data(), (size_t)N, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && unienc) e = hipMemcpyAsync(d_crs_int, crs, (size_t)p.l_uni * poly_bytes(c), hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { cleanup(); return hipfail(c, e, "device keygen setup"); }
    mktd::KeygenArgs a{};
    std::memcpy(a.key, K->key, sizeof a.key); a.party = K->party; a.N = N; a.n = p.n; a.W = p.W; a.f = p.f; a.logD = p.logD;
    a.sigma_ring = K->sigma_ring; a.sigma_lwe = K->sigma_lwe;
    a.lwekey = d_lwe; a.zring = d_z; a.crs = d_crs_int; a.out = d_out;
    if (unienc) { a.kr = 1; a.l = p.l_uni; a.logB = p.logB_uni; a.zoff = 0; }
    else { a.kr = c->sh.kr; a.l = p.l_gsw; a.logB = p.logB_gsw; a.zoff = 0; }
    e = mktd::launch_keygen_brk(a, unienc ? 1 : 0, c->stream);
    if (e == hipSuccess && brk_out) e = hipMemcpyAsync(brk_out, d_out, brk_polys_total * poly_bytes(c), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = (c->exact && c->split == 2) ? mktd::launch_ntt_fwd_split(c->logN, c->d_ntt, d_out, reinterpret_cast<uint64_t *>(c->ks->d_brk + (size_t)party * c->ks->brk_party_cplx), brk_polys_total, c->stream)
                           : c->exact ? mktd::launch_ntt_fwd(c->logN, p.W, c->d_ntt, d_out, reinterpret_cast<uint64_t *>(c->ks->d_brk + (size_t)party * c->ks->brk_party_cplx), brk_polys_total, 1, c->stream)
                               : mktd::launch_transform_fwd(c->logM, p.W, c->twp(), d_out, c->ks->d_brk + (size_t)party * c->ks->brk_party_cplx, brk_polys_total, c->dev_order, c->stream);
    if (e == hipSuccess && c->ks->d_fx_brk) e = mktd::launch_fx_key_fwd(c->logM, p.W, c->fx_om(), c->fx_tw(), d_out, c->ks->d_fx_brk + (size_t)party * c->ks->fx_brk_party_cplx, brk_polys_total, c->ks->d_fx_stat, c->stream);
    uint32_t *ksk = c->ks->d_ksk + (size_t)party * c->ks->ksk_party_words;
    if (e == hipSuccess) e = hipMemsetAsync(ksk, 0, c->ks->ksk_party_words * sizeof(uint32_t), c->stream);
    a.zoff = mkt::is_kms(p.scheme) ? 1 : 0;      // the key switch targets the uni key of the KMS schemes
    if (e == hipSuccess) e = mktd::launch_keygen_ksk(a, ksk, c->ks->n1p, c->sh.ksk_kr, c->sh.ksk_drows, mkt::is_block(p.scheme) ? 1 : 0, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    explicit_bzero(&a, sizeof a);            // the host copy of the party's stream key (the kernel-argument copy: see the TRUST note in mktfhe.h)
    if (e != hipSuccess) return hipfail(c, e, "device keygen");
    c->ks->brk_loaded[party] = 1; c->ks->ksk_loaded[party] = 1;
    if (c->ks->d_fx_brk) { int r = fx_after_key_load(c); if (r) return r; }
    if (ksk_out) return mkt_get_ksk(c, party, ksk_out);
    return MKT_OK;
}

int mkt_keygen_device(mkt_ctx *c, int party, const mkt_client_party *K, const void *crs) {
    return keygen_device_impl(c, party, K, crs, nullptr, nullptr);
}

// the same, and the generated keys are also copied out in the host layouts of mkt_load_brk (MKT_FMT_INT_COEFF) /
// mkt_load_ksk: a party generates its evaluation keys on its OWN GPU and ships them (key blob) to the evaluator,
// which never sees a secret
int mkt_keygen_device_export(mkt_ctx *c, int party, const mkt_client_party *K, const void *crs, void *brk_out, uint32_t *ksk_out) {
    if (!brk_out || !ksk_out) return fail(c, MKT_ERR_ARG, "null output");
    return keygen_device_impl(c, party, K, crs, brk_out, ksk_out);
}

// debug / test read-back of a party's key-switching key in the host layout of mkt_load_ksk
int mkt_get_ksk(mkt_ctx *c, int party, uint32_t *out_host) {
    if (!c || !out_host || party < 0 || party >= c->sh.nparty) return fail(c, MKT_ERR_ARG, "bad argument");
    MKT_EXACT_GATE(c);
    DevGuard dg(c->device);
    const size_t rows = (size_t)c->sh.ksk_kr * c->p.N * c->sh.ksk_drows * c->p.f, n1 = (size_t)c->p.n + 1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy2D(out_host, n1 * 4, c->ks->d_ksk + (size_t)party * c->ks->ksk_party_words, (size_t)c->ks->n1p * 4, n1 * 4, rows, hipMemcpyDeviceToHost));
    return MKT_OK;
}

// ---- batched hot path ----

// the gate entry points share one body: `op` for the whole batch or per-gate `ops`; operands in batch order (x, y: [B][len]) or
// picked by row index from a pool (x = y = pool, [pool_rows][len])
static int gate_impl(mkt_ctx *c, int op, const uint8_t *ops, const uint32_t *x, const uint32_t *y, size_t rows_xy, const uint32_t *ix, const uint32_t *iy,
                     uint32_t *out, size_t B, int mem) {
    MKT_EXACT_GATE(c);
    int r;
    if ((r = check_ready(c, true, true))) return r;
    DevGuard dg(c->device);
    Timer whole(c, 0);
    const size_t len = (size_t)c->sh.lwe_len;
    const bool pool = ix != nullptr;
    Staged sx{c}, sy{c}, so{c}, sops{c}, six{c}, siy{c};
    if ((r = sx.in(x, rows_xy * len * 4, mem, true))) return r;
    if (pool) sy.dev = sx.dev; else if ((r = sy.in(y, rows_xy * len * 4, mem, true))) return r;
    if ((r = so.in(out, B * len * 4, mem, false))) return r;
    if (ops && (r = sops.in(ops, B, mem, true))) return r;
    if (pool && ((r = six.in(ix, B * 4, mem, true)) || (r = siy.in(iy, B * 4, mem, true)))) return r;
    for (size_t off = 0; off < B; off += CHUNK_GATES) {
        const size_t nb = std::min(CHUNK_GATES, B - off);
        if ((r = ensure_workspace(c, nb))) return r;
        const size_t xoff = pool ? 0 : off * len;
        HIPCHK(c, mktd::launch_gate_linear(op, ops ? (const uint8_t *)sops.dev + off : nullptr, (const uint32_t *)sx.dev + xoff, (const uint32_t *)sy.dev + xoff,
                                           pool ? (const uint32_t *)six.dev + off : nullptr, pool ? (const uint32_t *)siy.dev + off : nullptr, pool ? rows_xy : 0, c->ws_lin, (int)len, nb, c->stream));
        if ((r = bootstrap_chunk(c, c->ws_lin, (uint32_t *)so.dev + off * len, nb))) return r;
    }
    return so.out(out);
}

int mkt_gate_batch(mkt_ctx *c, int op, const uint32_t *x, const uint32_t *y, uint32_t *out, size_t B, int mem) {
    if (!c || !x || !y || !out || !mem_ok(mem) || op < MKT_NAND || op > MKT_NOR) return fail(c, MKT_ERR_ARG, "bad argument");
    return gate_impl(c, op, nullptr, x, y, B, nullptr, nullptr, out, B, mem);
}

// a different gate per ciphertext pair -- the shape of the reference's own tests (test/KMS.jl:29-34 draws a random gate per step;
// gate.jl:1-53) -- in ONE launch sequence: ops[j] = MKT_NAND .. MKT_NOR, optionally | MKT_OP_NOT_X / MKT_OP_NOT_Y
static bool ops_valid_host(const uint8_t *ops, size_t B) {
    for (size_t j = 0; j < B; j++) if ((ops[j] & 7) > MKT_NOR || (ops[j] & ~31u)) return false;
    return true;
}
int mkt_gate_batch_ops(mkt_ctx *c, const uint8_t *ops, const uint32_t *x, const uint32_t *y, uint32_t *out, size_t B, int mem) {
    if (!c || !ops || !x || !y || !out || !mem_ok(mem)) return fail(c, MKT_ERR_ARG, "bad argument");
    if (mem == MKT_MEM_HOST && !ops_valid_host(ops, B)) return fail(c, MKT_ERR_ARG, "mkt_gate_batch_ops: unknown gate code");
    return gate_impl(c, 0, ops, x, y, B, nullptr, nullptr, out, B, mem);
}

// one circuit level: gate j reads pool[ix[j]] and pool[iy[j]] (rows of [pool_rows][k*n+1]) and writes out[j]; `out` may be a
// later region of the same pool as long as no gate of THIS call reads a row this call writes
int mkt_gate_batch_gather(mkt_ctx *c, const uint8_t *ops, const uint32_t *pool, size_t pool_rows, const uint32_t *ix, const uint32_t *iy,
                          uint32_t *out, size_t B, int mem) {
    if (!c || !ops || !pool || !ix || !iy || !out || !mem_ok(mem)) return fail(c, MKT_ERR_ARG, "bad argument");
    if (B && !pool_rows) return fail(c, MKT_ERR_ARG, "mkt_gate_batch_gather: gates over an empty pool");     // (the kernel clamps indices into [0, pool_rows): there must be a row to clamp to, in either memory kind)
    if (mem == MKT_MEM_HOST) {
        if (!ops_valid_host(ops, B)) return fail(c, MKT_ERR_ARG, "mkt_gate_batch_gather: unknown gate code");
        for (size_t j = 0; j < B; j++) if (ix[j] >= pool_rows || iy[j] >= pool_rows) return fail(c, MKT_ERR_ARG, "mkt_gate_batch_gather: operand index outside the pool");
    }
    return gate_impl(c, 0, ops, pool, pool, pool_rows, ix, iy, out, B, mem);
}

// MUX(s, a, b) = s ? a : b with TWO blind rotations and ONE key switch (the reference has no MUX gate, gate.jl:1-57; this is the
// CGGI16 construction written with the reference's own operators):
//   acc = blindrotate!(AND-linear(s, a)) + blindrotate!(AND-linear(NOT! s, b)), + 1/8 at X^0 of acc.b;  out = keyswitch!(acc)
// Each rotation leaves +-1/8 and at most one of the two ANDs holds, so the sum + 1/8 is +-1/8 again.  A composite of the
// reference's gates, OR(AND(s, a), AND(NOT s, b)), costs three full bootstraps.
int mkt_mux_batch(mkt_ctx *c, const uint32_t *sel, const uint32_t *a, const uint32_t *b, uint32_t *out, size_t B, int mem) {
    if (!c || !sel || !a || !b || !out || !mem_ok(mem)) return fail(c, MKT_ERR_ARG, "bad argument");
    MKT_EXACT_GATE(c);
    int r;
    if ((r = check_ready(c, true, true))) return r;
    DevGuard dg(c->device);
    Timer whole(c, 0);
    const size_t len = (size_t)c->sh.lwe_len, words = (size_t)(1 + c->sh.kacc) * c->p.N;
    Staged ss{c}, sa{c}, sb{c}, so{c};
    if ((r = ss.in(sel, B * len * 4, mem, true)) || (r = sa.in(a, B * len * 4, mem, true)) || (r = sb.in(b, B * len * 4, mem, true)) || (r = so.in(out, B * len * 4, mem, false))) return r;
    constexpr size_t HALF = CHUNK_GATES / 2;                  // two rotations per gate share the workspace chunk
    for (size_t off = 0; off < B; off += HALF) {
        const size_t nb = std::min(HALF, B - off);
        if ((r = ensure_workspace(c, 2 * nb))) return r;
        const uint32_t *ps = (const uint32_t *)ss.dev + off * len;
        HIPCHK(c, mktd::launch_gate_linear(MKT_AND, nullptr, ps, (const uint32_t *)sa.dev + off * len, nullptr, nullptr, 0, c->ws_lin, (int)len, nb, c->stream));
        HIPCHK(c, mktd::launch_gate_linear(MKT_AND | MKT_OP_NOT_X, nullptr, ps, (const uint32_t *)sb.dev + off * len, nullptr, nullptr, 0, c->ws_lin + nb * len, (int)len, nb, c->stream));
        if ((r = rotate_chunk(c, c->ws_lin, 2 * nb))) return r;
        HIPCHK(c, mktd::launch_mux_combine(c->p.W, c->ws_acc, nb, words, c->stream));
        if ((r = do_keyswitch(c, c->ws_acc, (uint32_t *)so.dev + off * len, nb))) return r;
    }
    return so.out(out);
}

// the same MUX with its operands picked by row index from a ciphertext pool (a circuit level of MUX gates): gate j = MUX(pool[is[j]],
// pool[ia[j]], pool[ib[j]]) -> out[j]; out may be a later region of the pool that no gate of this call reads
// not_ab (optional, lives where the indices live): bit 0 / bit 1 of not_ab[j] = the a / b operand of gate j is negated first (a circuit's
// free NOTs; a negated SELECTOR is the caller swapping a and b)
int mkt_mux_batch_gather(mkt_ctx *c, const uint32_t *pool, size_t pool_rows, const uint32_t *is, const uint32_t *ia, const uint32_t *ib, const uint8_t *not_ab,
                         uint32_t *out, size_t B, int mem) {
    if (!c || !pool || !is || !ia || !ib || !out || !mem_ok(mem)) return fail(c, MKT_ERR_ARG, "bad argument");
    if (B && !pool_rows) return fail(c, MKT_ERR_ARG, "mkt_mux_batch_gather: gates over an empty pool");
    if (mem == MKT_MEM_HOST)
        for (size_t j = 0; j < B; j++) if (is[j] >= pool_rows || ia[j] >= pool_rows || ib[j] >= pool_rows || (not_ab && (not_ab[j] & ~3u))) return fail(c, MKT_ERR_ARG, "mkt_mux_batch_gather: operand index outside the pool, or unknown flag");
    MKT_EXACT_GATE(c);
    int r;
    if ((r = check_ready(c, true, true))) return r;
    DevGuard dg(c->device);
    Timer whole(c, 0);
    const size_t len = (size_t)c->sh.lwe_len, words = (size_t)(1 + c->sh.kacc) * c->p.N;
    Staged sp{c}, ss{c}, sa{c}, sb{c}, so{c}, sf{c};
    if ((r = sp.in(pool, pool_rows * len * 4, mem, true)) || (r = ss.in(is, B * 4, mem, true)) || (r = sa.in(ia, B * 4, mem, true)) || (r = sb.in(ib, B * 4, mem, true)) || (r = so.in(out, B * len * 4, mem, false))) return r;
    if (not_ab && (r = sf.in(not_ab, B, mem, true))) return r;
    constexpr size_t HALF = CHUNK_GATES / 2;
    for (size_t off = 0; off < B; off += HALF) {
        const size_t nb = std::min(HALF, B - off);
        if ((r = ensure_workspace(c, 2 * nb))) return r;
        const uint32_t *pp = (const uint32_t *)sp.dev, *js = (const uint32_t *)ss.dev + off;
        const uint8_t *fl = not_ab ? (const uint8_t *)sf.dev + off : nullptr;
        HIPCHK(c, mktd::launch_mux_linear(pp, pool_rows, js, (const uint32_t *)sa.dev + off, (const uint32_t *)sb.dev + off, fl, c->ws_lin, (int)len, nb, c->stream));
        if ((r = rotate_chunk(c, c->ws_lin, 2 * nb))) return r;
        HIPCHK(c, mktd::launch_mux_combine(c->p.W, c->ws_acc, nb, words, c->stream));
        if ((r = do_keyswitch(c, c->ws_acc, (uint32_t *)so.dev + off * len, nb))) return r;
    }
    return so.out(out);
}

int mkt_not_batch(mkt_ctx *c, uint32_t *x, size_t B, int mem) {
    if (!c || !x || !mem_ok(mem)) return fail(c, MKT_ERR_ARG, "bad argument");
    DevGuard dg(c->device);
    Staged sx{c};
    int r;
    const size_t words = B * (size_t)c->sh.lwe_len;
    if ((r = sx.in(x, words * 4, mem, true))) return r;
    HIPCHK(c, mktd::launch_negate((uint32_t *)sx.dev, words, c->stream));
    return sx.out(x);
}

int mkt_bootstrap_batch(mkt_ctx *c, uint32_t *lwe, size_t B, int mem) {
    if (!c || !lwe || !mem_ok(mem)) return fail(c, MKT_ERR_ARG, "bad argument");
    MKT_EXACT_GATE(c);
    int r;
    if ((r = check_ready(c, true, true))) return r;
    DevGuard dg(c->device);
    Timer whole(c, 0);
    const size_t len = (size_t)c->sh.lwe_len;
    Staged sx{c};
    if ((r = sx.in(lwe, B * len * 4, mem, true))) return r;
    for (size_t off = 0; off < B; off += CHUNK_GATES) {
        const size_t nb = std::min(CHUNK_GATES, B - off);
        if ((r = ensure_workspace(c, nb))) return r;
        uint32_t *chunk = (uint32_t *)sx.dev + off * len;
        // the rotation reads the masks, phase 2 / the test vector read b, all before key switching overwrites them
        if ((r = bootstrap_chunk(c, chunk, chunk, nb))) return r;
    }
    return sx.out(lwe);
}

int mkt_modswitch_batch(mkt_ctx *c, const uint32_t *lwe, uint32_t *atilde, uint32_t *btilde, size_t B, int mem) {
    if (!c || !lwe || !atilde || !btilde || !mem_ok(mem)) return fail(c, MKT_ERR_ARG, "bad argument");
    DevGuard dg(c->device);
    const size_t len = (size_t)c->sh.lwe_len;
    Staged sl{c}, sa{c}, sb{c};
    int r;
    if ((r = sl.in(lwe, B * len * 4, mem, true)) || (r = sa.in(atilde, B * (len - 1) * 4, mem, false)) || (r = sb.in(btilde, B * 4, mem, false))) return r;
    HIPCHK(c, mktd::launch_modswitch((const uint32_t *)sl.dev, (uint32_t *)sa.dev, (uint32_t *)sb.dev, (int)len, c->logN, B, c->stream));
    if ((r = sa.out(atilde))) return r;
    return sb.out(btilde);
}

int mkt_blindrotate_batch(mkt_ctx *c, const uint32_t *atilde, void *acc, size_t B, int mem) {
    if (!c || !atilde || !acc || !mem_ok(mem)) return fail(c, MKT_ERR_ARG, "bad argument");
    MKT_EXACT_GATE(c);
    int r;
    if ((r = check_ready(c, true, false))) return r;
    DevGuard dg(c->device);
    Timer whole(c, 0);
    const size_t alen = (size_t)c->sh.lwe_len - 1, accb = (size_t)(1 + c->sh.kacc) * poly_bytes(c);
    Staged sa{c}, sc{c};
    if ((r = sa.in(atilde, B * alen * 4, mem, true)) || (r = sc.in(acc, B * accb, mem, true))) return r;
    for (size_t off = 0; off < B; off += CHUNK_GATES) {
        const size_t nb = std::min(CHUNK_GATES, B - off);
        if ((r = ensure_workspace(c, nb))) return r;
        if ((r = do_blindrotate(c, (const uint32_t *)sa.dev + off * alen, (int)alen, 1, nullptr, (char *)sc.dev + off * accb, c->ws_lev, c->ws_scratch, nb))) return r;
    }
    return sc.out(acc);
}

int mkt_kms_phase1_batch(mkt_ctx *c, const uint32_t *atilde, double *levkey, size_t B, int mem) {
    if (!c || !atilde || !levkey || !mem_ok(mem) || !mkt::is_kms(c->p.scheme)) return fail(c, MKT_ERR_ARG, "bad argument");
    MKT_F64_OR_EXACT_KMS(c);
    int r;
    if ((r = check_ready(c, true, false))) return r;
    DevGuard dg(c->device);
    const size_t alen = (size_t)c->sh.lwe_len - 1, lb = (size_t)c->ks->rtot * 2 * c->M * sizeof(cplx) * c->split;
    Staged sa{c}, sl{c};
    if ((r = sa.in(atilde, B * alen * 4, mem, true)) || (r = sl.in(levkey, B * lb, mem, false))) return r;
    if (c->exact) {   // the rows as split residue tables [B][rows][2 polys][2 halves][N] (uint64 residue pairs, Montgomery form)
        const mkt_params &p = c->p;
        mktd::ExactKmsArgs q{};
        q.brk = reinterpret_cast<const uint64_t *>(c->ks->d_brk); q.brk_party_stride = c->ks->brk_party_cplx * 2 /* in 8-byte residue pairs */; q.mono = reinterpret_cast<const uint64_t *>(c->ks->d_monomial);
        q.lwe = (const uint32_t *)sa.dev; q.lwe_stride = (int)alen; q.pre_switched = 1; q.n = p.n; q.k = p.k; q.l_gsw = p.l_gsw; q.logB_gsw = p.logB_gsw;
        q.l_lev = p.l_lev; q.logB_lev = p.logB_lev; q.l_uni = p.l_uni; q.logB_uni = p.logB_uni; q.rtot = c->ks->rtot; q.lwe_len = c->sh.lwe_len; q.blk_len = p.scheme == MKT_KMS_BLOCK ? p.blk_len : 1;
        q.slot_party = c->ks->d_slot_party; q.slot_row = c->ks->d_slot_row; q.levkey = (uint64_t *)sl.dev; q.phase1_only = 1; q.wide = c->tune.exact_wide;
        if (p.scheme == MKT_KMS && fx_usable(c)) {
            if ((r = ensure_workspace(c, B < CHUNK_GATES ? B : CHUNK_GATES))) return r;
            for (size_t off = 0; off < B; off += CHUNK_GATES) {
                const size_t nb = B - off < CHUNK_GATES ? B - off : CHUNK_GATES, nrot = nb * (size_t)c->ks->rtot;
                mktd::FxRotArgs f = fx_rot_args(c, (const uint32_t *)sa.dev + off * alen, (int)alen, 1);
                f.init_mode = 1; f.acc_io = c->ws_fxacc; f.ngates = nb;
                Timer tm(c, 1);
                HIPCHK(c, mktd::launch_fx_blindrotate(c->logM, p.W, f, nrot, c->stream));
                HIPCHK(c, mktd::launch_ntt_fwd_split(c->logN, c->d_ntt, c->ws_fxacc, (uint64_t *)sl.dev + off * (lb / 8), nrot * 2, c->stream));
            }
            return sl.out(levkey);
        }
        { Timer tm(c, 1); HIPCHK(c, mktd::launch_exact_kms(c->logN, c->d_ntt, q, B, c->stream)); }
        return sl.out(levkey);
    }
    mktd::RotArgs a = rot_args(c, (const uint32_t *)sa.dev, (int)alen, 1);
    a.init_mode = 1; a.out_mode = 1; a.tout = (cplx *)sl.dev; a.tout_natural = 1; a.ngates = B;
    { Timer tm(c, 1); HIPCHK(c, mktd::launch_blindrotate_k1(c->logM, c->p.W, a, B * (size_t)c->ks->rtot, c->stream)); }
    return sl.out(levkey);
}

int mkt_keyswitch_batch(mkt_ctx *c, const void *acc, uint32_t *out, size_t B, int mem) {
    if (!c || !acc || !out || !mem_ok(mem)) return fail(c, MKT_ERR_ARG, "bad argument");
    MKT_EXACT_GATE(c);
    int r;
    if ((r = check_ready(c, false, true))) return r;
    DevGuard dg(c->device);
    const size_t accb = (size_t)(1 + c->sh.kacc) * poly_bytes(c), len = (size_t)c->sh.lwe_len;
    Staged sc{c}, so{c};
    if ((r = sc.in(acc, B * accb, mem, true)) || (r = so.in(out, B * len * 4, mem, false))) return r;
    if ((r = do_keyswitch(c, sc.dev, (uint32_t *)so.dev, B))) return r;
    return so.out(out);
}

int mkt_transform_fwd_batch(mkt_ctx *c, const void *p, double *t, size_t B, int mem) {
    if (!c || !p || !t || !mem_ok(mem)) return fail(c, MKT_ERR_ARG, "bad argument");
    DevGuard dg(c->device);
    Staged sp{c}, st{c};
    int r;
    if ((r = sp.in(p, B * poly_bytes(c), mem, true)) || (r = st.in(t, B * (size_t)c->M * sizeof(cplx), mem, false))) return r;
    if (c->exact) { Timer tm(c, 3); HIPCHK(c, mktd::launch_ntt_fwd(c->logN, c->p.W, c->d_ntt, sp.dev, (uint64_t *)st.dev, B, 0, c->stream)); }
    else { Timer tm(c, 3); HIPCHK(c, mktd::launch_transform_fwd(c->logM, c->p.W, c->twp(), sp.dev, (cplx *)st.dev, B, 0, c->stream)); }
    return st.out(t);
}

int mkt_transform_inv_batch(mkt_ctx *c, const double *t, void *p, size_t B, int mem) {
    if (!c || !p || !t || !mem_ok(mem)) return fail(c, MKT_ERR_ARG, "bad argument");
    DevGuard dg(c->device);
    Staged st{c}, sp{c};
    int r;
    if ((r = st.in(t, B * (size_t)c->M * sizeof(cplx), mem, true)) || (r = sp.in(p, B * poly_bytes(c), mem, false))) return r;
    if (c->exact) { Timer tm(c, 3); HIPCHK(c, mktd::launch_ntt_inv(c->logN, c->p.W, c->d_ntt, (const uint64_t *)st.dev, sp.dev, B, c->stream)); }
    else { Timer tm(c, 3); HIPCHK(c, mktd::launch_transform_inv(c->logM, c->p.W, c->twp(), (const cplx *)st.dev, sp.dev, B, c->stream)); }
    return sp.out(p);
}

int mkt_decompose_batch(mkt_ctx *c, const void *p, void *digits, int l, int logB, size_t B, int mem) {
    if (!c || !p || !digits || !mem_ok(mem) || l < 1 || logB < 1 || l * logB > c->p.W) return fail(c, MKT_ERR_ARG, "bad argument");
    DevGuard dg(c->device);
    Staged sp{c}, sd{c};
    int r;
    if ((r = sp.in(p, B * poly_bytes(c), mem, true)) || (r = sd.in(digits, B * (size_t)l * poly_bytes(c), mem, false))) return r;
    HIPCHK(c, mktd::launch_decompose(c->p.W, sp.dev, sd.dev, c->p.N, l, logB, B, c->stream));
    return sd.out(digits);
}

// MKT_ARITH_EXACT: out = a (*) b in Z_{2^W}[X]/(X^N + 1), exact, for a gadget-digit polynomial a (signed, N * max|a_i| < 2^28: true coefficients below P / 2) and any b
int mkt_exact_polymul_batch(mkt_ctx *c, const void *a, const void *b, void *out, size_t B, int mem) {
    if (!c || !a || !b || !out || !mem_ok(mem)) return fail(c, MKT_ERR_ARG, "bad argument");
    if (!c->exact) return fail(c, MKT_ERR_UNSUPPORTED, "mkt_exact_polymul_batch needs an MKT_ARITH_EXACT context");
    DevGuard dg(c->device);
    Staged sa{c}, sb{c}, so{c};
    int r;
    if ((r = sa.in(a, B * poly_bytes(c), mem, true)) || (r = sb.in(b, B * poly_bytes(c), mem, true)) || (r = so.in(out, B * poly_bytes(c), mem, false))) return r;
    if (c->ks->d_fx_tab && c->tune.exact_impl == 1) {   // the Float64-pipe product (fx_exact.hip), certified per call by its measured rounding distance
        HIPCHK(c, hipMemsetAsync(c->ks->d_fx_stat + 1, 0, 8, c->stream));
        { Timer tm(c, 3); HIPCHK(c, mktd::launch_fx_polymul(c->logM, c->p.W, c->fx_om(), c->fx_tw(), c->fx_nat(), sa.dev, sb.dev, so.dev, B, c->ks->d_fx_stat + 1, c->stream)); }
        unsigned long long bits = 0;
        HIPCHK(c, hipMemcpyAsync(&bits, c->ks->d_fx_stat + 1, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        std::memcpy(&c->fx_last_resid, &bits, 8);
        if (!(c->fx_last_resid < 0.25)) return fail(c, MKT_ERR_UNSUPPORTED, "mkt_exact_polymul_batch (Float64 pipe): a rounding distance of 1/4 or more -- operands beyond what 16-bit limbs certify; use exact_impl = 0");
        return so.out(out);
    }
    { Timer tm(c, 3); HIPCHK(c, mktd::launch_exact_polymul(c->logN, c->p.W, c->d_ntt, sa.dev, sb.dev, so.dev, B, c->stream)); }
    return so.out(out);
}

int mkt_enable_timing(mkt_ctx *c, int on) {
    if (!c) return MKT_ERR_ARG;
    DevGuard dg(c->device);
    (void)hipStreamSynchronize(c->stream);
    clear_spans(c);
    c->timing = on != 0;
    return MKT_OK;
}

// sum of the recorded spans of class `which` since timing was enabled (ms); *ms / count = average launch
int mkt_last_kernel_ms(mkt_ctx *c, int which, double *ms) {
    if (!c || !ms) return MKT_ERR_ARG;
    if (!c->timing) return fail(c, MKT_ERR_STATE, "timing not enabled");
    DevGuard dg(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double tot = 0.0; int cnt = 0;
    for (auto &s : c->spans) if (s.cls == which) { float f = 0; if (hipEventElapsedTime(&f, s.a, s.b) == hipSuccess) { tot += f; cnt++; } }
    *ms = tot;
    return cnt;
}

}  // extern "C"
