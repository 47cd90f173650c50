// Micro-benchmark of the row-block chains' weight stream (tgat_chain.hip): 8 waves per workgroup, every wave streams 1-KiB operand
// fragments of its tiles from a packed buffer through a ring of U registers and feeds MFMAs from one LDS operand.
//   hipcc -O3 --offload-arch=gfx950 tools/chain_ubench.hip -o tools/chain_ubench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f4 = __attribute__((ext_vector_type(4))) float;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ f4 mfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// MODE bit0: MFMAs on; bit1: LDS operand reads on; bit2: all workgroups stream the SAME fragments (else: workgroup-private offset)
template <int U, int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k(const f4* __restrict__ w, float* out, int steps, int frags_total) {
    __shared__ float lds[16 * 452];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * 452; i += blockDim.x) lds[i] = 1.0f + (i & 7);
    __syncthreads();
    const int c = lane & 15, g = lane >> 4;
    unsigned pos = (unsigned)wave * steps + ((MODE & 4) ? 0u : (unsigned)blockIdx.x * 977u);
    f4 ring[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { ring[u] = w[(size_t)(pos % frags_total) * 64 + lane]; ++pos; }
    f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    const float* ab = lds + c * 452 + 4 * g;
    f4 bn = *reinterpret_cast<const f4*>(ab);
    for (int s0 = 0; s0 + U <= steps; s0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f4 x = ring[u];
            const f4 b = bn;
            if (MODE & 2) bn = *reinterpret_cast<const f4*>(ab + 16 * ((s0 + u + 1) % 27));
            if (MODE & 1) { a0 = mfma(x.x, b.x, a0); a1 = mfma(x.y, b.y, a1); a0 = mfma(x.z, b.z, a0); a1 = mfma(x.w, b.w, a1); }
            else { a0 += x * b; }
            ring[u] = w[(size_t)(pos % frags_total) * 64 + lane]; ++pos;
        }
    }
    const f4 r = a0 + a1;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r.x + r.y + r.z + r.w;
}
__global__ void k_write(f4* w, size_t n) { const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) w[i] = f4{1.f, 0.5f, 0.25f, 0.125f}; }

template <int U, int MODE, int NW>
void run(const f4* w, float* out, int grid, int steps, int frags, const char* tag, bool rewrite) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        if (rewrite) { hipLaunchKernelGGL(k_write, dim3((frags * 64 + 255) / 256), dim3(256), 0, 0, const_cast<f4*>(w), (size_t)frags * 64); }
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<U, MODE, NW>), dim3(grid), dim3(NW * 64), 0, 0, w, out, steps, frags);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    printf("%-34s U=%2d NW=%2d grid=%5d steps/wave=%4d buffer=%5.1f MB %s: %8.1f us  %6.1f ns/step  %7.1f GB/s/CU-equivalent total %7.2f TB/s\n", tag, U, NW, grid, steps,
           frags / 1024.0, rewrite ? "rewritten" : "static   ", best * 1e3, best * 1e6 / steps, NW * steps * 1024.0 / (best * 1e-3) / 1e9,
           (double)grid * NW * steps * 1024.0 / (best * 1e-3) / 1e12);
}
int main() {
    const int frags = 2048 * 2;      // 4 MB
    f4* w; float* out;
    CK(hipMalloc(&w, (size_t)frags * 1024)); CK(hipMalloc(&out, 4096 * 1024 * sizeof(float)));
    hipLaunchKernelGGL(k_write, dim3((frags * 64 + 255) / 256), dim3(256), 0, 0, w, (size_t)frags * 64);
    CK(hipDeviceSynchronize());
    const int F2 = 2048;   // 2 MB working set
    for (int grid : {50, 256, 1024}) {
        run<8, 7, 8>(w, out, grid, 208, F2, "mfma+lds same-stream", grid == 50);
        run<8, 7, 8>(w, out, grid, 208, F2, "mfma+lds same-stream", false);
        run<8, 3, 8>(w, out, grid, 208, F2, "mfma+lds private-offset", false);
        run<8, 4, 8>(w, out, grid, 208, F2, "loads only same-stream", false);
        run<16, 7, 8>(w, out, grid, 208, F2, "mfma+lds same-stream", false);
        run<16, 4, 8>(w, out, grid, 208, F2, "loads only same-stream", false);
        run<8, 7, 16>(w, out, grid, 104, F2, "mfma+lds same-stream 16 waves", false);
        run<16, 7, 16>(w, out, grid, 104, F2, "mfma+lds same-stream 16 waves", false);
        run<8, 5, 8>(w, out, grid, 208, F2, "mfma no-lds same-stream", false);
    }
    return 0;
}
