// Where do the waves of small workgroups land?  (gfx950 / MI355X)  A grid of W-wave workgroups, each holding LDS so that K of them
// fit a CU, every wave records HW_ID (SIMD, CU, SE) and XCC_ID while all workgroups are resident (they spin ~1 ms); the host prints
// how many CUs carry which waves-per-SIMD pattern.  Background: a round of the two-wave rotation kernel costs 3.4 / 5.0 / 5.35 / 6.3 ms
// at 1 / 2 / 3 / 4 workgroups per CU (profiles/r04_experiments.txt item 13).
//   hipcc -O3 --offload-arch=gfx950 tools/simd_place.hip -o /tmp/simd_place && /tmp/simd_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
#include <array>
#include <string>

extern __shared__ char smem[];
// live: waves of the workgroup that stay (the others leave at once): does a 4-wave launch with two live waves spread better than a 2-wave one?
__global__ void place(uint32_t *out, int spin, int live) {
    if ((int)(threadIdx.x / 64) >= live) return;
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    smem[threadIdx.x] = (char)hw;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    while ((int64_t)(__builtin_amdgcn_s_memtime() - t0) < spin) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * live + threadIdx.x / 64;
        out[2 * w] = hw; out[2 * w + 1] = xcc;
    }
}

int main() {
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    for (int mode : {0, 1, 2}) for (int per_cu : {1, 2, 3, 4, 5, 6, 8}) {
        const int waves = mode == 0 ? 2 : 4, live = mode == 2 ? 2 : waves;      // mode 2: four waves launched, two stay
        const int wgs = ncu * per_cu;
        const size_t lds = (size_t)(160 * 1024 / (per_cu <= 4 ? 4 : 8)) - 512;   // four (eight) workgroups per CU at most
        uint32_t *d; (void)hipMalloc((void **)&d, (size_t)wgs * live * 8);
        (void)hipFuncSetAttribute((const void *)place, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(place, dim3(wgs), dim3(64 * waves), lds, 0, d, 100000, live);     // 100 MHz counter: 1 ms
        (void)hipDeviceSynchronize();
        std::vector<uint32_t> h((size_t)wgs * live * 2);
        (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        std::map<uint32_t, std::array<int, 4>> cu;                              // (xcc, se, sh, cu) -> waves per SIMD
        for (size_t w = 0; w < (size_t)wgs * live; w++) {
            const uint32_t hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
            const uint32_t simd = (hw >> 4) & 3, cuid = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            cu[(xcc << 16) | (se << 8) | (sh << 4) | cuid][simd]++;
        }
        std::map<std::string, int> pat;
        for (auto &kv : cu) { char b[32]; snprintf(b, sizeof b, "%d %d %d %d", kv.second[0], kv.second[1], kv.second[2], kv.second[3]); pat[b]++; }
        printf("%d-wave workgroups (%d live), %d per CU (%d workgroups on %zu CUs seen): waves on SIMD 0 1 2 3 -> CUs:", waves, live, per_cu, wgs, cu.size());
        for (auto &kv : pat) printf("  [%s] x %d", kv.first.c_str(), kv.second);
        printf("\n");
        (void)hipFree(d);
    }
    return 0;
}
