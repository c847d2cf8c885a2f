// What a two-CU split of ONE blind rotation would pay per hand-off: two workgroups on two CUs of one XCD pass a
// buffer of PAYLOAD bytes back and forth -- sc1 (write-through) 16-byte stores, every storing wave's s_waitcnt vmcnt(0), a workgroup
// barrier, one sc1 flag store; the consumer polls the flag with sc1 loads, passes a barrier and reads the payload with sc1 loads
// (MI355X_MICROARCH.md, "Valid forms": the row `ONE lane of each storing workgroup ... sc1 flag store`).  Reported: microseconds per
// one-way hand-off (half a round trip), median over the pairs, with idle CUs around (the best case for a single-gate rotation).
//   make -C tools bin/handoff_probe && tools/bin/handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st16_sc1(void *p, u4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u4 ld16_sc1(const void *p) { u4 v; asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void st4_sc1(uint32_t *p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ uint32_t ld4_sc1(const uint32_t *p) { uint32_t v; asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }

// pair q = blockIdx.x % npairs, side = blockIdx.x / npairs (blocks b and b + npairs with npairs a multiple of 8 share an XCD under round-robin dispatch)
__global__ __launch_bounds__(256) void pingpong(u4 *buf, uint32_t *flags, uint64_t *out, uint32_t *xcc, int npairs, int payload16, int iters) {
    const int q = blockIdx.x % npairs, side = blockIdx.x / npairs, t = threadIdx.x;
    u4 *mine = buf + ((size_t)q * 2 + side) * payload16, *theirs = buf + ((size_t)q * 2 + (side ^ 1)) * payload16;
    uint32_t *fmine = flags + ((size_t)q * 2 + side) * 64, *ftheirs = flags + ((size_t)q * 2 + (side ^ 1)) * 64;     // a flag per 256-byte line
    u4 acc = {(unsigned)t, (unsigned)side, (unsigned)q, 1u};
    if (t == 0) { uint32_t x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); xcc[blockIdx.x] = x & 0xf; }
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();        // 100 MHz
    for (int it = 1; it <= iters; it++) {
        if ((it & 1) == side) {              // my turn to produce
            for (int i = t; i < payload16; i += 256) st16_sc1(&mine[i], u4{acc.x + (unsigned)i, acc.y, acc.z, (unsigned)it});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) st4_sc1(fmine, (uint32_t)it);
        } else {                             // my turn to consume
            if (t == 0) { int spin = 0; while (ld4_sc1(ftheirs) != (uint32_t)it && ++spin < (1 << 22)) __builtin_amdgcn_s_sleep(1); }
            __syncthreads();
            for (int i = t; i < payload16; i += 256) { const u4 v = ld16_sc1(&theirs[i]); acc.x += v.x; acc.y ^= v.w; }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
    if (t == 0) out[blockIdx.x] = t1 - t0;
    if (acc.x == 0x12345678u && acc.y == 77u) out[blockIdx.x] = 0;
}

int main() {
    const int npairs = 8, iters = 2000;
    for (int payload : {64, 4096, 8192, 16384, 32768}) {
        const int p16 = payload / 16;
        u4 *buf; uint32_t *flags, *xcc; uint64_t *out;
        (void)hipMalloc(&buf, (size_t)npairs * 2 * payload); (void)hipMalloc(&flags, (size_t)npairs * 2 * 256); (void)hipMalloc(&out, npairs * 2 * 8); (void)hipMalloc(&xcc, npairs * 2 * 4);
        (void)hipMemset(flags, 0, (size_t)npairs * 2 * 256);
        hipLaunchKernelGGL(pingpong, dim3(npairs * 2), dim3(256), 0, 0, buf, flags, out, xcc, npairs, p16, iters);
        (void)hipDeviceSynchronize();
        std::vector<uint64_t> h(npairs * 2); std::vector<uint32_t> x(npairs * 2);
        (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(x.data(), xcc, x.size() * 4, hipMemcpyDeviceToHost);
        std::vector<double> us; int same = 0;
        for (int q = 0; q < npairs; q++) { us.push_back((double)h[q] / 100.0 / iters); same += x[q] == x[q + npairs]; }
        std::sort(us.begin(), us.end());
        printf("payload %6d B: %.2f us per one-way hand-off (median of %d pairs, min %.2f, max %.2f; %d pairs on one XCD), %d hand-offs each\n", payload, us[npairs / 2], npairs, us.front(), us.back(), same, iters);
        (void)hipFree(buf); (void)hipFree(flags); (void)hipFree(out); (void)hipFree(xcc);
    }
    return 0;
}
