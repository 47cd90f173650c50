// Micro-benchmark (round 3): do VALU instructions issue in the shadow of v_mfma_f32_16x16x4_f32, or do they take matrix-pipe time?
// Per loop iteration: 16 independent MFMAs (4 accumulators x 4 k) and NV VALU instructions of kind KIND spread between them, on 1 or 2
// waves per SIMD.  hipcc -O3 --offload-arch=gfx950 tools/coissue_ubench.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f4 = __attribute__((ext_vector_type(4))) float;
using f2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f4 mfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// KIND 0: v_fma_f32 chain   1: v_pk_fma_f32 chain   2: integer v_add_u32 / v_xor chain   3: v_exp_f32 (transcendental)   4: v_cndmask / v_cmp
template <int NV, int KIND, bool MF>
__global__ void k(float* out, int iters) {
    f4 acc[4], a[4];
    f4 b = {1.f + threadIdx.x, 2.f, 3.f, 4.f};
    float v[8]; int iv[8]; f2 pv[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) { acc[u] = f4{0, 0, 0, 0}; a[u] = f4{1.f * u, 2.f + threadIdx.x, 3.f, 4.f}; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u] = 1.0f + 0.001f * (threadIdx.x + u); iv[u] = threadIdx.x + u; pv[u] = f2{v[u], v[u] + 1.f}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (MF) acc[u] = mfma(a[u][t], b[t], acc[u]);
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int r = (j + u) & 7;
                    if (KIND == 0) v[r] = __builtin_fmaf(v[r], 1.0000001f, 1e-7f);
                    if (KIND == 1) pv[r] = __builtin_elementwise_fma(pv[r], f2{1.0000001f, 1.0000001f}, f2{1e-7f, 1e-7f});
                    if (KIND == 2) iv[r] = (iv[r] + 12345) ^ (iv[r] >> 3);
                    if (KIND == 3) v[r] = __builtin_amdgcn_exp2f(v[r]) * 0.5f;
                    if (KIND == 4) v[r] = v[r] > 1.5f ? v[(r + 1) & 7] : v[r];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(a[u]));
    }
    float s = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) s += acc[u].x + acc[u].y + acc[u].z + acc[u].w;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u] + iv[u] + pv[u].x + pv[u].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int NV, int KIND, bool MF>
void run(int threads) {
    const int blocks = 256, iters = 4000;
    float* out; CK(hipMalloc(&out, blocks * threads * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<NV, KIND, MF>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<NV, KIND, MF>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: waves per SIMD = threads / 256; MFMAs per wave = 16 * iters
    const double wps = threads / 256.0;
    const double us_per_iter_simd = ms * 1e3 / iters;                       // one iteration of all waves of a SIMD (they run concurrently)
    const double tfl = MF ? 2048.0 * 16 * iters * (threads / 64.0) * blocks / (ms * 1e-3) / 1e12 : 0.0;
    static const char* kn[] = {"v_fma_f32", "v_pk_fma_f32", "int add/xor", "v_exp_f32+mul", "cmp+cndmask"};
    printf("waves/SIMD %.0f  MFMA %d  VALU per MFMA %d (%s): %8.3f ms  %7.4f us per iteration  %6.1f TFLOP/s\n", wps, (int)MF, NV, kn[KIND], ms, us_per_iter_simd, tfl);
    CK(hipFree(out));
}
int main() {
    for (int threads : {256, 512}) {
        run<0, 0, true>(threads);
        run<2, 0, true>(threads); run<4, 0, true>(threads); run<7, 0, true>(threads);
        run<4, 1, true>(threads); run<4, 2, true>(threads); run<2, 3, true>(threads); run<2, 4, true>(threads);
        run<4, 0, false>(threads); run<4, 1, false>(threads); run<4, 2, false>(threads); run<2, 3, false>(threads);
    }
    return 0;
}
