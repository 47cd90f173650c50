// Micro-benchmark for the "token-owner" wave layout: every wave owns 16 tokens x all 208 channels in registers
// (LayerNorm output xn[13] is the MFMA B operand straight from VGPRs), all 8 waves of a workgroup consume the SAME
// weight fragments.  Measures the FFN loop (2 hidden tiles per step: 26 W1 fragments, GELU, 26 W2 fragments) with the
// weight (A) operand fed from: registers only / a static LDS buffer / LDS stages filled by LDS-DMA / a global ring.
// hipcc -O3 --offload-arch=gfx950 tools/v3_ubench.hip -o tools/v3_ubench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
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
__device__ __forceinline__ f4 lds4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ void dma_frag(const float* gsrc_lane, float* lds_dst_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 0);
}
__device__ __forceinline__ float gelu_fast(float v) {
    const float x = v * 0.70710678f, ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
    const float e = __expf(-ax * ax);
    return 0.5f * v * (1.0f + copysignf(fmaf(-p * t, e, 1.0f), x));
}
constexpr int kFrag = 256, kStage = 26;

// MODE 0: registers only; 1: static LDS; 2: LDS-DMA stages + barriers; 3: global ring (R = 13)
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ w, float* out, int reps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * kStage * kFrag; i += 512) lds[i] = 1e-3f * (i & 255);
    __syncthreads();
    f4 xn[13], y[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) { xn[i] = f4{1e-3f * lane, 2e-3f, 1e-3f * i, 1e-3f}; y[i] = f4{0, 0, 0, 0}; }
    float* stg = lds;                                  // [2 buffers][26 frags][256]
    const float* wl = w + lane * 4;
    auto issue_part = [&](int st, int f) {             // wave's share of stage st: fragments wave, wave+8, ...
        if (8 * f + wave < kStage)
            dma_frag(wl + ((size_t)(st & 127) * kStage + 8 * f + wave) * kFrag, stg + (size_t)((st & 1) * kStage + 8 * f + wave) * kFrag);
    };
    f4 ring[13];
    const float* rp = wl;
    if (MODE == 3) {
#pragma unroll
        for (int u = 0; u < 13; ++u) ring[u] = *reinterpret_cast<const f4*>(rp + (size_t)u * kFrag);
        rp += 13 * kFrag;
    }
    auto take = [&](int slot) -> f4 {
        const f4 v = ring[slot];
        ring[slot] = *reinterpret_cast<const f4*>(rp);
        rp += kFrag;
        return v;
    };
    if (MODE == 2) {
#pragma unroll
        for (int f = 0; f < 4; ++f) issue_part(0, f);
        __syncthreads();
    }
    int st = 0;
#pragma unroll 1
    for (int rep = 0; rep < reps; ++rep) {
        if (MODE == 3) rp = wl + 13 * kFrag;
#pragma unroll 1
        for (int p = 0; p < 25; ++p) {
            // ---------------- A stage: h[u] = W1[tile u] . xn   (fragments [kc][u])
            f4 h[2] = {f4{0.1f, 0.2f, 0.3f, 0.4f}, f4{0.4f, 0.3f, 0.2f, 0.1f}};
            {
                const float* abuf = stg + lane * 4;     // buffer 0
                f4 sa[2][2];
                if (MODE == 1 || MODE == 2) { sa[0][0] = lds4(abuf); sa[0][1] = lds4(abuf + kFrag); }
                if (MODE == 0) { sa[0][0] = xn[1]; sa[0][1] = xn[2]; sa[1][0] = xn[3]; sa[1][1] = xn[4]; }
#pragma unroll
                for (int kc = 0; kc < 13; ++kc) {
                    const int cur = kc & 1;
                    if (MODE == 1 || MODE == 2) {
                        if (kc + 1 < 13) { sa[cur ^ 1][0] = lds4(abuf + (size_t)(2 * (kc + 1)) * kFrag); sa[cur ^ 1][1] = lds4(abuf + (size_t)(2 * (kc + 1) + 1) * kFrag); }
                    }
                    if (MODE == 3) { sa[cur][0] = take((2 * kc) % 13); sa[cur][1] = take((2 * kc + 1) % 13); }
                    if (MODE == 2 && (kc & 3) == 0) issue_part(st + 1, kc >> 2);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_group<2>(h, sa[cur], xn[kc]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) { h[u].x = gelu_fast(h[u].x); h[u].y = gelu_fast(h[u].y); h[u].z = gelu_fast(h[u].z); h[u].w = gelu_fast(h[u].w); }
            }
            if (MODE == 2) { __syncthreads(); }
            ++st;
            // ---------------- B stage: y[i] += W2[i][tile u] . h[u]   (fragments [u][i]), groups (4,3,3,3) x 2
            {
                const float* bbuf = stg + (size_t)kStage * kFrag + lane * 4;   // buffer 1
                f4 fs[2][4];
                if (MODE == 1 || MODE == 2) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) fs[0][v] = lds4(bbuf + (size_t)v * kFrag);
                }
                if (MODE == 0) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) { fs[0][v] = xn[v]; fs[1][v] = xn[4 + v]; }
                }
#pragma unroll
                for (int gi = 0; gi < 8; ++gi) {
                    const int u = gi >> 2, q = gi & 3;
                    const int i0 = q == 0 ? 0 : 4 + 3 * (q - 1), n = q == 0 ? 4 : 3;
                    if (MODE == 1 || MODE == 2) {
                        if (gi + 1 < 8) {
                            const int u2 = (gi + 1) >> 2, q2 = (gi + 1) & 3;
                            const int j0 = q2 == 0 ? 0 : 4 + 3 * (q2 - 1), n2 = q2 == 0 ? 4 : 3;
#pragma unroll
                            for (int v = 0; v < 4; ++v) if (v < n2) fs[(gi + 1) & 1][v] = lds4(bbuf + (size_t)(u2 * 13 + j0 + v) * kFrag);
                        }
                    }
                    if (MODE == 3) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) if (v < n) fs[gi & 1][v] = take((u * 13 + i0 + v) % 13);
                    }
                    if (MODE == 2 && gi < 4) issue_part(st + 1, gi);
                    __builtin_amdgcn_sched_barrier(0);
                    if (n == 4) mma_group<4>(&y[i0], fs[gi & 1], h[u]); else mma_group<3>(&y[i0], fs[gi & 1], h[u]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (MODE == 2) { __syncthreads(); }
            ++st;
        }
    }
    float s = 0;
#pragma unroll
    for (int u = 0; u < 13; ++u) s += y[u].x + y[u].y + y[u].z + y[u].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MODE 4: NB stage buffers, DMA issued two stages ahead, first operands of the next stage prefetched BEFORE the barrier,
// GELU of the previous tile pair spread over the A stage of the next one (software pipelined).
template <int NB, bool PIPE_GELU, bool BARRIER, bool ILV = false>
__global__ __launch_bounds__(512, 2) void k4(const float* __restrict__ w, float* out, int reps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < NB * kStage * kFrag; i += 512) lds[i] = 1e-3f * (i & 255);
    __syncthreads();
    f4 xn[13], y[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) { xn[i] = f4{1e-3f * lane, 2e-3f, 1e-3f * i, 1e-3f}; y[i] = f4{0, 0, 0, 0}; }
    float* stg = lds;
    const float* wl = w + lane * 4;
    auto issue_part = [&](int st, int f) {
        if (8 * f + wave < kStage)
            dma_frag(wl + ((size_t)(st & 127) * kStage + 8 * f + wave) * kFrag, stg + (size_t)((st % NB) * kStage + 8 * f + wave) * kFrag);
    };
#pragma unroll
    for (int f = 0; f < 4; ++f) { issue_part(0, f); issue_part(1, f); }
    __syncthreads();
    int st = 0;
    f4 ha[2] = {f4{0.1f, 0.2f, 0.3f, 0.4f}, f4{0.4f, 0.3f, 0.2f, 0.1f}}, hb[2] = {f4{0.1f, 0.2f, 0.3f, 0.4f}, f4{0.4f, 0.3f, 0.2f, 0.1f}};
    f4 sa[2][2], fs[2][4];
    sa[0][0] = lds4(stg + lane * 4); sa[0][1] = lds4(stg + kFrag + lane * 4);
    auto body = [&](f4 (&h)[2], f4 (&hn)[2]) {
        // ---------------- A stage (stage st): hn[u] = W1[tile u] . xn ; gelu(h) interleaved
        {
            const float* abuf = stg + (size_t)(st % NB) * kStage * kFrag + lane * 4;
            const float* nbuf = stg + (size_t)((st + 1) % NB) * kStage * kFrag + lane * 4;
#pragma unroll
            for (int kc = 0; kc < 13; ++kc) {
                const int cur = kc & 1;
                if (kc + 1 < 13) { sa[cur ^ 1][0] = lds4(abuf + (size_t)(2 * (kc + 1)) * kFrag); sa[cur ^ 1][1] = lds4(abuf + (size_t)(2 * (kc + 1) + 1) * kFrag); }
                else {
#pragma unroll
                    for (int v = 0; v < 4; ++v) fs[0][v] = lds4(nbuf + (size_t)v * kFrag);       // first group of the B stage
                }
                if ((kc & 3) == 0) issue_part(st + 2, kc >> 2);
                if (ILV) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (kc < 8) h[kc >> 2][kc & 3] = gelu_fast(h[kc >> 2][kc & 3]);
                    mma_group<2>(hn, sa[cur], xn[kc]);
                    // 8 x { 1 MFMA, 3 VALU }: the GELU of the previous tile pair issues in the shadow of this wave's own MFMAs
#pragma unroll
                    for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                if (PIPE_GELU && kc < 8) h[kc >> 2][kc & 3] = gelu_fast(h[kc >> 2][kc & 3]);
                __builtin_amdgcn_sched_barrier(0);
                mma_group<2>(hn, sa[cur], xn[kc]);
                __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!PIPE_GELU) {
#pragma unroll
                for (int u = 0; u < 2; ++u) { h[u].x = gelu_fast(h[u].x); h[u].y = gelu_fast(h[u].y); h[u].z = gelu_fast(h[u].z); h[u].w = gelu_fast(h[u].w); }
            }
        }
        if (BARRIER) __syncthreads();
        ++st;
        // ---------------- B stage: y[i] += W2[i][tile u] . h[u]
        {
            const float* bbuf = stg + (size_t)(st % NB) * kStage * kFrag + lane * 4;
            const float* nbuf = stg + (size_t)((st + 1) % NB) * kStage * kFrag + lane * 4;
#pragma unroll
            for (int gi = 0; gi < 8; ++gi) {
                const int u = gi >> 2, q = gi & 3;
                const int i0 = q == 0 ? 0 : 4 + 3 * (q - 1), n = q == 0 ? 4 : 3;
                if (gi + 1 < 8) {
                    const int u2 = (gi + 1) >> 2, q2 = (gi + 1) & 3;
                    const int j0 = q2 == 0 ? 0 : 4 + 3 * (q2 - 1), n2 = q2 == 0 ? 4 : 3;
#pragma unroll
                    for (int v = 0; v < 4; ++v) if (v < n2) fs[(gi + 1) & 1][v] = lds4(bbuf + (size_t)(u2 * 13 + j0 + v) * kFrag);
                } else {
                    sa[0][0] = lds4(nbuf); sa[0][1] = lds4(nbuf + kFrag);                          // first group of the next A stage
                }
                if (gi < 4) issue_part(st + 2, gi);
                __builtin_amdgcn_sched_barrier(0);
                if (n == 4) mma_group<4>(&y[i0], fs[gi & 1], h[u]); else mma_group<3>(&y[i0], fs[gi & 1], h[u]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (BARRIER) __syncthreads();
        ++st;
    };
#pragma unroll 1
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll 1
        for (int p = 0; p < 26; p += 2) { body(ha, hb); body(hb, ha); }
    }
    float s = 0;
#pragma unroll
    for (int u = 0; u < 13; ++u) s += y[u].x + y[u].y + y[u].z + y[u].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + ha[0].x + hb[1].y;
}
template <int NB, bool PG, bool BAR, bool ILV = false> void run4(const float* w, const char* name, int blocks) {
    const int reps = 40, threads = 512, ldsb = 160 * 1024;
    float* out;
    CK(hipMalloc(&out, blocks * threads * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k4<NB, PG, BAR, ILV>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k4<NB, PG, BAR, ILV>), dim3(blocks), dim3(threads), ldsb, 0, w, out, reps);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k4<NB, PG, BAR, ILV>), dim3(blocks), dim3(threads), ldsb, 0, w, out, reps);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)blocks * 8 * reps * 26 * 208.0 * 2048.0;
    printf("%-48s blocks=%d wall %.3f ms -> %.1f TFLOP/s (%.1f%% of %d-CU peak)\n", name, blocks, ms, flops / (ms * 1e-3) / 1e12,
           100.0 * flops / (ms * 1e-3) / (157.3e12 * (blocks < 256 ? blocks : 256) / 256.0), blocks);
    CK(hipFree(out));
}
template <int MODE> void run(const float* w, const char* name, int blocks) {
    const int reps = 40, threads = 512, ldsb = 160 * 1024;
    float* out;
    CK(hipMalloc(&out, blocks * threads * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), ldsb, 0, w, out, reps);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), ldsb, 0, w, out, reps);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)blocks * 8 * reps * 25 * 208.0 * 2048.0;
    printf("%-48s blocks=%d wall %.3f ms -> %.1f TFLOP/s (%.1f%% of %d-CU peak)\n", name, blocks, ms, flops / (ms * 1e-3) / 1e12,
           100.0 * flops / (ms * 1e-3) / (157.3e12 * (blocks < 256 ? blocks : 256) / 256.0), blocks);
    CK(hipFree(out));
}
int main() {
    float* w; const size_t n = (size_t)201 * kStage * kFrag + 64 * kFrag;
    CK(hipMalloc(&w, n * 4)); CK(hipMemset(w, 0, n * 4));
    run<0>(w, "registers only", 64);
    run<0>(w, "registers only", 128);
    run<0>(w, "registers only", 200);
    run<0>(w, "registers only", 256);
    run<2>(w, "A via LDS-DMA stages + barriers", 128);
    run<1>(w, "A from static LDS (ds_read_b128)", 256);
    run<2>(w, "A via LDS-DMA stages + barriers", 256);
    run<3>(w, "A via global ring R=13", 256);
    run4<3, false, true>(w, "k4: 3 buffers, prefetch over barrier", 256);
    run4<3, true, true>(w, "k4: 3 buffers, prefetch, pipelined GELU", 256);
    run4<3, true, false>(w, "k4: same, NO barriers (racy; upper bound)", 256);
    run4<3, true, true, true>(w, "k4: 3 buffers, GELU interleaved by sched_group_barrier", 256);
    return 0;
}
