// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 for the access patterns of the fused kernel.
// hipcc -O3 --offload-arch=gfx950 tools/mfma_ubench.hip -o /tmp/mfma_ubench && /tmp/mfma_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
using f4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f4 mfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int NACC, int MODE>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    f4 acc[NACC];
    f4 a[NACC];
    f4 b = {1.f + threadIdx.x, 2.f, 3.f, 4.f};
#pragma unroll
    for (int u = 0; u < NACC; ++u) { acc[u] = f4{0, 0, 0, 0}; a[u] = f4{1.f * u, 2.f + threadIdx.x, 3.f, 4.f}; }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {        // t-major over NACC independent accumulators (mma_group)
#pragma unroll
            for (int u = 0; u < NACC; ++u) acc[u] = mfma(a[u].x, b.x, acc[u]);
#pragma unroll
            for (int u = 0; u < NACC; ++u) acc[u] = mfma(a[u].y, b.y, acc[u]);
#pragma unroll
            for (int u = 0; u < NACC; ++u) acc[u] = mfma(a[u].z, b.z, acc[u]);
#pragma unroll
            for (int u = 0; u < NACC; ++u) acc[u] = mfma(a[u].w, b.w, acc[u]);
        } else {                // accumulator-major: 4 dependent MFMAs per accumulator
#pragma unroll
            for (int u = 0; u < NACC; ++u) {
                acc[u] = mfma(a[u].x, b.x, acc[u]); acc[u] = mfma(a[u].y, b.y, acc[u]);
                acc[u] = mfma(a[u].z, b.z, acc[u]); acc[u] = mfma(a[u].w, b.w, acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < NACC; ++u) asm volatile("" : "+v"(a[u]));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int u = 0; u < NACC; ++u) s += acc[u].x + acc[u].y + acc[u].z + acc[u].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int NACC, int MODE>
void run(int threads, const char* name) {
    const int blocks = 256, iters = 20000;
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, blocks * threads * 4)); CK(hipMalloc(&cyc, blocks * 16 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * threads / 64);
    CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double m = 0; for (auto v : h) m += v; m /= h.size();
    const double per = m / (iters * 4.0 * NACC);
    const double flops = (double)blocks * (threads / 64) * iters * 4.0 * NACC * 2048.0;
    printf("%-30s waves/SIMD=%d  memtime ticks per MFMA per wave = %7.2f | wall %.3f ms -> %.1f TFLOP/s, implied clock %.2f GHz\n", name,
           threads / 256, per, ms, flops / (ms * 1e-3) / 1e12, m / (ms * 1e-3) / 1e9);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    run<5, 0>(256, "5 acc, t-major");
    run<5, 0>(512, "5 acc, t-major");
    run<4, 0>(256, "4 acc, t-major");
    run<2, 0>(256, "2 acc, t-major");
    run<2, 0>(512, "2 acc, t-major");
    run<1, 0>(256, "1 acc (dependent chain)");
    run<1, 0>(512, "1 acc (dependent chain)");
    run<5, 1>(256, "5 acc, acc-major (4-chains)");
    run<5, 1>(512, "5 acc, acc-major (4-chains)");
    run<13, 0>(256, "13 acc, t-major");
    run<13, 0>(512, "13 acc, t-major");
    run<5, 0>(768, "5 acc, t-major");
    run<5, 0>(1024, "5 acc, t-major");
    run<20, 0>(256, "20 acc, t-major");
    run<20, 0>(512, "20 acc, t-major");
    return 0;
}
