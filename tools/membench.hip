// Memory-pattern ceilings on MI355X for the transform kernels' access shapes (diagnostic tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// (1) plain copy, 16 B/lane, grid-stride
__global__ void copy16(const uint4 *in, uint4 *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
// (2) per-block "polynomial" of 8 KiB in / 8 KiB out: 8 B/lane loads (e*NT+t and +M), 64 B-strided 16 B stores
template <int NT>
__global__ void poly_pattern_a(const uint64_t *p, double2 *out, size_t B) {
    constexpr int M = NT * 4, N = 2 * M; const int t = threadIdx.x;
    for (size_t b = blockIdx.x; b < B; b += gridDim.x) {
        double2 z[4];
        for (int e = 0; e < 4; e++) { z[e].x = (double)(int64_t)p[b * N + e * NT + t]; z[e].y = (double)(int64_t)p[b * N + M + e * NT + t]; }
        for (int e = 0; e < 4; e++) out[b * M + t * 4 + e] = z[e];
    }
}
// (3) same but contiguous 16 B stores (e*NT + t)
template <int NT>
__global__ void poly_pattern_b(const uint64_t *p, double2 *out, size_t B) {
    constexpr int M = NT * 4, N = 2 * M; const int t = threadIdx.x;
    for (size_t b = blockIdx.x; b < B; b += gridDim.x) {
        double2 z[4];
        for (int e = 0; e < 4; e++) { z[e].x = (double)(int64_t)p[b * N + e * NT + t]; z[e].y = (double)(int64_t)p[b * N + M + e * NT + t]; }
        for (int e = 0; e < 4; e++) out[b * M + e * NT + t] = z[e];
    }
}
// (4) 16 B loads + contiguous 16 B stores
template <int NT>
__global__ void poly_pattern_c(const uint4 *p, double2 *out, size_t B) {
    constexpr int M = NT * 4; const int t = threadIdx.x;
    for (size_t b = blockIdx.x; b < B; b += gridDim.x) {
        uint4 v[4];
        for (int k = 0; k < 4; k++) v[k] = p[b * M + k * NT + t];
        for (int e = 0; e < 4; e++) { double2 z; z.x = (double)(int64_t)(((uint64_t)v[e].y << 32) | v[e].x); z.y = (double)(int64_t)(((uint64_t)v[e].w << 32) | v[e].z); out[b * M + e * NT + t] = z; }
    }
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const size_t bytes = 2ull << 30;   // 2 GiB in, 2 GiB out
    void *in, *out; CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    const size_t n16 = bytes / 16; const size_t B = bytes / 8192;
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        float t1 = timeit([&] { hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const uint4 *)in, (uint4 *)out, n16); }, 5);
        float ta = timeit([&] { hipLaunchKernelGGL(poly_pattern_a<128>, dim3(grid), dim3(128), 0, 0, (const uint64_t *)in, (double2 *)out, B); }, 5);
        float tb = timeit([&] { hipLaunchKernelGGL(poly_pattern_b<128>, dim3(grid), dim3(128), 0, 0, (const uint64_t *)in, (double2 *)out, B); }, 5);
        float tc = timeit([&] { hipLaunchKernelGGL(poly_pattern_c<128>, dim3(grid), dim3(128), 0, 0, (const uint4 *)in, (double2 *)out, B); }, 5);
        printf("grid %5d  copy16 %.0f GB/s | 8B-load+strided-store %.0f | 8B-load+contig-store %.0f | 16B-load+contig-store %.0f\n", grid,
               2 * bytes / t1 / 1e6, 2 * bytes / ta / 1e6, 2 * bytes / tb / 1e6, 2 * bytes / tc / 1e6);
    }
    return 0;
}
