// The clock the part holds INSIDE the batched integer transforms ("explain the 1.6-1.8 GHz").  A diagnostic copy of
// ntt_fwd_kernel's loop (the device functions are the engine's own: this file includes csrc/ntt_exact.hip) with s_memtime (shader
// cycles) and s_memrealtime (100 MHz) stamped around the loop of every workgroup; in-kernel clock = d memtime / d memrealtime x 100 MHz,
// median over workgroups, after >= 2 s of back-to-back launches on random data (MI355X_MICROARCH.md, DVFS give-back item 6).  The stamps go
// to a buffer of their own; the shipped kernels carry none.
//   make -C tools bin/ntt_clock_probe && tools/bin/ntt_clock_probe
#include "../mktfhe_amd/csrc/ntt_exact.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>

namespace mktd {
const LaunchTuning &launch_tuning() { static LaunchTuning t{}; return t; }
thread_local const char *last_rot_kernel = "";
namespace {
template <int LOGN, bool COMPUTE>
__global__ __launch_bounds__((Ppw<LOGN>::v << (LOGN - NLR))) void ntt_fwd_stamped(const uint4 *__restrict__ tab, const uint64_t *__restrict__ p, uint64_t *__restrict__ out, size_t B, uint64_t *stamps) {
    constexpr int N = 1 << LOGN, NT = N >> NLR, PPW = Ppw<LOGN>::v;
    const int sub = PPW > 1 ? threadIdx.x / NT : 0, t = PPW > 1 ? threadIdx.x % NT : threadIdx.x;
    uint64_t *lds = reinterpret_cast<uint64_t *>(ntt_smem) + (size_t)sub * NttLds<LOGN>::WORDS;
    const uint4 *tw[1]; const int which[1] = {0};
    stage_tables<LOGN, 1>(tab, reinterpret_cast<uint4 *>(reinterpret_cast<uint64_t *>(ntt_smem) + (size_t)PPW * NttLds<LOGN>::WORDS), threadIdx.x, PPW * NT, tw, which);
    const size_t groups = (B + PPW - 1) / PPW;
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (size_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const size_t b = g * PPW + sub < B ? g * PPW + sub : B - 1;
        Pt z[8];
#pragma unroll
        for (int e = 0; e < 8; e++) z[e] = fwd_in(__builtin_nontemporal_load(&p[b * N + e * NT + t]), e);
        if (COMPUTE) {
            ntt_forward<LOGN>(z, tw[0], lds, t);
#pragma unroll
            for (int e = 0; e < 8; e++) z[e] = pt_canon4(z[e]);
            ntt_exchange<LOGN, 0, Plan<LOGN, NLR>::lo(0)>(z, lds, t);
        }
#pragma unroll
        for (int e = 0; e < 8; e++) __builtin_nontemporal_store(pack(z[e]), &out[b * N + e * NT + t]);
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
}  // namespace
}  // namespace mktd

template <int LOGN, bool COMPUTE>
static void run(const char *what) {
    using namespace mktd;
    constexpr int N = 1 << LOGN, PPW = Ppw<LOGN>::v;
    const size_t B = ((size_t)4 << 30) / (16 * N);
    const int grid = 8192;
    uint64_t *p, *out, *stamps; uint4 *tab;
    (void)hipMalloc(&p, B * N * 8); (void)hipMalloc(&out, B * N * 8); (void)hipMalloc(&stamps, (size_t)grid * 16); (void)hipMalloc(&tab, (size_t)(N + 4) * 16);
    std::vector<uint64_t> h(1 << 20); std::mt19937_64 rng(7);
    for (auto &x : h) x = rng();
    for (size_t off = 0; off < B * N; off += h.size()) (void)hipMemcpy(p + off, h.data(), std::min(h.size(), B * N - off) * 8, hipMemcpyHostToDevice);
    std::vector<uint32_t> tb((size_t)(N + 4) * 4);
    for (auto &x : tb) x = (uint32_t)rng() % P1;                                // any residues: the clock does not depend on the table's meaning
    (void)hipMemcpy(tab, tb.data(), tb.size() * 4, hipMemcpyHostToDevice);
    const size_t lds = lds_bytes<LOGN>(1, PPW);
    (void)ntt_set_lds(ntt_fwd_stamped<LOGN, COMPUTE>, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int launches = 1600;                                                   // ~2 s back to back before the launch that is read
    (void)hipEventRecord(e0);
    for (int i = 0; i < launches; i++) hipLaunchKernelGGL((ntt_fwd_stamped<LOGN, COMPUTE>), dim3(grid), dim3(PPW << (LOGN - NLR)), lds, 0, tab, p, out, B, stamps);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> s((size_t)grid * 2);
    (void)hipMemcpy(s.data(), stamps, s.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int b = 0; b < grid; b++) if (s[2 * b + 1] > 50) ghz.push_back((double)s[2 * b] / (double)s[2 * b + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    printf("%-34s N %d: %.3f ms per launch (%.2f TB/s of 16 N bytes), in-kernel clock median %.3f GHz (p10 %.3f, p90 %.3f) over %zu workgroups, %d launches in %.2f s\n", what, N,
           ms / launches, (double)B * 16 * N / (ms / launches * 1e-3) / 1e12, ghz[ghz.size() / 2], ghz[ghz.size() / 10], ghz[ghz.size() * 9 / 10], ghz.size(), launches, ms * 1e-3);
    (void)hipFree(p); (void)hipFree(out); (void)hipFree(stamps); (void)hipFree(tab);
}

// the engine's own launcher (ntt_fwd_kernel: the next polynomial's words requested a transform ahead), timed the same way
template <int LOGN>
static void run_shipped() {
    using namespace mktd;
    constexpr int N = 1 << LOGN;
    const size_t B = ((size_t)4 << 30) / (16 * N);
    uint64_t *p, *out; uint4 *tab;
    (void)hipMalloc(&p, B * N * 8); (void)hipMalloc(&out, B * N * 8); (void)hipMalloc(&tab, (size_t)(N + 4) * 16);
    std::vector<uint64_t> h(1 << 20); std::mt19937_64 rng(7);
    for (auto &x : h) x = rng();
    for (size_t off = 0; off < B * N; off += h.size()) (void)hipMemcpy(p + off, h.data(), std::min(h.size(), B * N - off) * 8, hipMemcpyHostToDevice);
    std::vector<uint32_t> tb((size_t)(N + 4) * 4);
    for (auto &x : tb) x = (uint32_t)rng() % P1;
    (void)hipMemcpy(tab, tb.data(), tb.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int launches = 800;
    (void)launch_ntt_fwd(LOGN, 64, reinterpret_cast<const uint64_t *>(tab), p, out, B, 0, 0);
    (void)hipEventRecord(e0);
    for (int i = 0; i < launches; i++) (void)launch_ntt_fwd(LOGN, 64, reinterpret_cast<const uint64_t *>(tab), p, out, B, 0, 0);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s N %d: %.3f ms per launch (%.2f TB/s of 16 N bytes), %d launches back to back\n", "shipped ntt_fwd_kernel", N, ms / launches, (double)B * 16 * N / (ms / launches * 1e-3) / 1e12, launches);
    (void)hipFree(p); (void)hipFree(out); (void)hipFree(tab);
}

int main() {
    run_shipped<10>();
    run<10, true>("integer NTT forward (64-bit words)");
    run<11, true>("integer NTT forward (64-bit words)");
    run<10, false>("the same loads and stores, no NTT");
    return 0;
}
