// Micro-benchmark of the fused kernel's inner-loop structure: FragStream refills + LDS B-fragment reads + mma_group<5>.
// hipcc -O3 --offload-arch=gfx950 tools/stream_ubench.hip -o tools/stream_ubench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f4 = __attribute__((ext_vector_type(4))) float;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ f4 mfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
template <int N> __device__ __forceinline__ void mma_group(f4* acc, const f4* a, const f4 b) {
#pragma unroll
    for (int u = 0; u < N; ++u) acc[u] = mfma(a[u].x, b.x, acc[u]);
#pragma unroll
    for (int u = 0; u < N; ++u) acc[u] = mfma(a[u].y, b.y, acc[u]);
#pragma unroll
    for (int u = 0; u < N; ++u) acc[u] = mfma(a[u].z, b.z, acc[u]);
#pragma unroll
    for (int u = 0; u < N; ++u) acc[u] = mfma(a[u].w, b.w, acc[u]);
}
// MODE bit0: global refills, bit1: LDS B reads, bit2: all 4 waves of a half read the SAME stream (L1 sharing as in the real kernel)
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ w, float* out, unsigned long long* cyc, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1.0f + i;
    __syncthreads();
    const int sid = (MODE & 4) ? (wave >> 2) : wave;
    const float* p = w + (size_t)sid * 1024 * 256 + lane * 4;     // 1 MiB stream per id (wraps)
    f4 ring[10];
#pragma unroll
    for (int u = 0; u < 10; ++u) ring[u] = *reinterpret_cast<const f4*>(p + u * 256);
    int pos = 10;
    f4 acc[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) acc[u] = f4{0, 0, 0, 0};
    const float* brow = lds + (wave & 3) * 16 * 204 + (lane & 15) * 204 + (lane >> 4) * 4;
    f4 bb = *reinterpret_cast<const f4*>(brow);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        f4 b1 = bb, b2 = bb;
        if (MODE & 2) { b1 = *reinterpret_cast<const f4*>(brow + 16 * ((2 * it + 1) % 12)); b2 = *reinterpret_cast<const f4*>(brow + 16 * ((2 * it + 2) % 12)); }
        f4 af[5], ag[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) { af[u] = ring[u]; if (MODE & 1) { ring[u] = *reinterpret_cast<const f4*>(p + (size_t)(pos & 1023) * 256); ++pos; } }
        mma_group<5>(acc, af, bb);
#pragma unroll
        for (int u = 0; u < 5; ++u) { ag[u] = ring[5 + u]; if (MODE & 1) { ring[5 + u] = *reinterpret_cast<const f4*>(p + (size_t)(pos & 1023) * 256); ++pos; } }
        mma_group<5>(acc, ag, b1);
        bb = b2;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int u = 0; u < 5; ++u) s += acc[u].x + acc[u].y + acc[u].z + acc[u].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int MODE> void run(const float* w, const char* name, int blocks) {
    const int iters = 4000, threads = 512;
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, blocks * threads * 4)); CK(hipMalloc(&cyc, blocks * 8 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 65536, 0, w, out, cyc, iters);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 65536, 0, w, out, cyc, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)blocks * 8 * iters * 40.0 * 2048.0;
    printf("%-52s blocks=%d wall %.3f ms -> %.1f TFLOP/s (%.1f%% of %d-CU peak)\n", name, blocks, ms, flops / (ms * 1e-3) / 1e12,
           100.0 * flops / (ms * 1e-3) / (157.3e12 * blocks / 256.0), blocks);
    CK(hipFree(out)); CK(hipFree(cyc));
}
int main() {
    float* w; CK(hipMalloc(&w, (size_t)8 * 1024 * 256 * 4)); CK(hipMemset(w, 0, (size_t)8 * 1024 * 256 * 4));
    run<0>(w, "MFMA only (ring never refilled)", 200);
    run<2>(w, "MFMA + LDS B reads", 200);
    run<1>(w, "MFMA + global refills (8 private streams / WG)", 200);
    run<5>(w, "MFMA + global refills (2 streams shared by 4 waves)", 200);
    run<7>(w, "MFMA + refills shared + LDS reads", 200);
    run<7>(w, "MFMA + refills shared + LDS reads", 256);
    return 0;
}
