// DyGFormer forward, generic path (any F/C/P/L/H, T <= 128 tokens per pair): a correctness-first
// multi-kernel pipeline with activations in HBM.  The fused MFMA kernel (dygformer_fused3.hip)
// is the fast path for the headline shape; this path covers every other shape and is the
// on-device cross-check for the fused one.  No host synchronisation anywhere: the batch-wide
// padded lengths S_src/S_dst (models/DyGFormer.py:219-226) stay on the device in CallDims and
// every kernel is launched for the worst case and skips rows beyond the actual token count.
#include "dygformer_layout.h"

namespace dygnn {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ------------------------------------------------------------------------------------------------
// pack: transposes + co-occurrence LUT
// ------------------------------------------------------------------------------------------------
__global__ void k_transpose(const float* __restrict__ W, float* __restrict__ Wt, int N, int K) {
    // W [N][K] -> Wt [K][N]
    __shared__ float tile[32][33];
    const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int n = n0 + i, k = k0 + threadIdx.x;
        tile[i][threadIdx.x] = (n < N && k < K) ? W[(size_t)n * K + k] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int k = k0 + i, n = n0 + threadIdx.x;
        if (k < K && n < N) Wt[(size_t)k * N + n] = tile[threadIdx.x][i];
    }
}

// f(c)[j] = b1[j] + sum_i W1[j][i] * relu(W0[i]*c + b0[i])     (models/DyGFormer.py:332-335)
// evaluated with the same operation order for every count c, so the table entry IS the value the
// reference's MLP produces for that count (fma for the K=1 Linear like PyTorch's CPU addmm).
__global__ void k_cooc_lut(const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ w1,
                           const float* __restrict__ b1, int C, int rows, float* __restrict__ lut) {
    const int c = blockIdx.x;
    const int j = threadIdx.x;
    if (c >= rows || j >= C) return;
    float acc = b1[j];
    for (int i = 0; i < C; ++i) {
        const float h = fmaxf(fmaf((float)c, w0[i], b0[i]), 0.f);
        acc = fmaf(h, w1[(size_t)j * C + i], acc);
    }
    lut[(size_t)c * C + j] = acc;
}

// ------------------------------------------------------------------------------------------------
// per-call sizes
// ------------------------------------------------------------------------------------------------
// one entry per group: a call may hold several independent batches ("groups" of G consecutive pairs), each padded to
// its own S_src / S_dst exactly as if it had been a separate reference call
__global__ void k_call_dims(CallDims* cds, int P, int64_t ngroups) {
    const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi < ngroups) {
        CallDims* cd = cds + gi;
        int Ss = cd->maxw_s + 1, Sd = cd->maxw_d + 1;            // + the target node (DyGFormer.py:223)
        if (Ss % P) Ss += P - Ss % P;                            // :224-225
        if (Sd % P) Sd += P - Sd % P;
        cd->S_s = Ss; cd->S_d = Sd; cd->T_s = Ss / P; cd->T_d = Sd / P; cd->T = cd->T_s + cd->T_d;
    }
}

struct CsrView2 { const int64_t* indptr; const int32_t* nbr; const int32_t* eid; const double* ts; int64_t num_nodes; };

// 64-ary wave search, see sampler.hip
__device__ __forceinline__ int64_t wave_lower_bound2(const double* __restrict__ ts, int64_t lo, int64_t hi, double t, int lane) {
    while (hi - lo > kWave) {
        const int64_t step = (hi - lo + kWave - 1) / kWave;
        const int64_t p = lo + (int64_t)lane * step;
        const bool pred = (p < hi) && (ts[p] < t);
        const int c = __popcll(__ballot(pred));
        if (c == 0) return lo;
        const int64_t nlo = lo + (int64_t)(c - 1) * step + 1;
        const int64_t nhi = lo + (int64_t)c * step;
        hi = nhi < hi ? nhi : hi;
        lo = nlo;
    }
    const int64_t p = lo + lane;
    const bool pred = (p < hi) && (ts[p] < t);
    return lo + __popcll(__ballot(pred));
}

// queries 0..B-1 = src side, B..2B-1 = dst side
__global__ __launch_bounds__(256) void k_window_lengths2(CsrView2 g, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                                           const double* __restrict__ times, int64_t B, int64_t G, int32_t L,
                                                           int32_t* __restrict__ hist_len, int64_t* __restrict__ end_pos, CallDims* cds) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= 2 * B) return;
    const bool is_dst = q >= B;
    const int64_t r = is_dst ? q - B : q;
    int64_t node = is_dst ? dst[r] : src[r];
    if (node < 0 || node >= g.num_nodes) node = 0;
    const int64_t lo = g.indptr[node], hi = g.indptr[node + 1];
    const int64_t i = wave_lower_bound2(g.ts, lo, hi, times[r], lane);
    if (lane == 0) {
        const int32_t len = (int32_t)(i - lo);
        hist_len[q] = len;
        end_pos[q] = i;
        CallDims* cd = cds + r / G;
        int32_t* mw = is_dst ? &cd->maxw_d : &cd->maxw_s;
        const int32_t v = len < L - 1 ? len : L - 1;
        if (v > __atomic_load_n(mw, __ATOMIC_RELAXED)) atomicMax(mw, v);      // monotone maximum: most queries skip the atomic
    }
}

// ------------------------------------------------------------------------------------------------
// embed: windows + co-occurrence + features + patch projection  -> X [B][Tmax][D]
// (models/DyGFormer.py:89-174)
// ------------------------------------------------------------------------------------------------
struct EmbedArgs {
    CsrView2 g;
    const int64_t *src, *dst;
    const double* times;
    const int32_t* hist_len;
    const int64_t* end_pos;
    const CallDims* cd;
    const float *node_feat, *edge_feat;
    const float *time_w, *time_b;
    const float* lut;
    const float* projT[4];
    const float* proj_b[4];
    float* X;
    int64_t B, G;
    int Fn, Fe, Ft, C, D, P, L, Tmax, Smax, lut_rows;
    int stage_floats;   // per-wave staging buffer = P * max(Fn,Fe,Ft,C)
};

__global__ __launch_bounds__(256) void k_embed(EmbedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t b = blockIdx.x;
    const CallDims cd = a.cd[b / a.G];
    const int Ss = cd.S_s, Sd = cd.S_d, S = Ss + Sd;
    int32_t* ids = reinterpret_cast<int32_t*>(smem);            // [2*Smax]
    int32_t* eids = ids + 2 * a.Smax;
    float* dts = reinterpret_cast<float*>(eids + 2 * a.Smax);
    int32_t* c0 = reinterpret_cast<int32_t*>(dts + 2 * a.Smax);
    int32_t* c1 = c0 + 2 * a.Smax;
    float* stage = reinterpret_cast<float*>(c1 + 2 * a.Smax);   // [4 waves][stage_floats]

    const double t = a.times[b];
    // ---- windows (pad_sequences, DyGFormer.py:228-245), left aligned, position 0 = target node
    for (int p = threadIdx.x; p < S; p += blockDim.x) {
        const bool is_dst = p >= Ss;
        const int j = is_dst ? p - Ss : p;
        const int64_t q = is_dst ? a.B + b : b;
        const int32_t len = a.hist_len[q];
        const int32_t m = len < a.L - 1 ? len : a.L - 1;
        int32_t id = 0, e = 0;
        float tn = 0.f;
        if (j == 0) {
            const int64_t qid = is_dst ? a.dst[b] : a.src[b];
            id = qid < 0 || qid >= a.g.num_nodes ? 0 : (int32_t)qid;      // a bad query id is the padding node (never a fault)
            tn = (float)t;
        } else if (j <= m) {
            const int64_t pos = a.end_pos[q] - m + (j - 1);
            id = a.g.nbr[pos]; e = a.g.eid[pos]; tn = (float)a.g.ts[pos];
        }
        ids[p] = id; eids[p] = e;
        dts[p] = (float)(t - (double)tn);                       // DyGFormer.py:263: f64 - f32 -> f64 -> .float()
    }
    __syncthreads();
    // ---- co-occurrence counts (DyGFormer.py:337-393)
    for (int p = threadIdx.x; p < S; p += blockDim.x) {
        const int32_t v = ids[p];
        int32_t cs = 0, cdn = 0;
        for (int q = 0; q < Ss; ++q) cs += (ids[q] == v);
        for (int q = Ss; q < S; ++q) cdn += (ids[q] == v);
        if (v == 0) { cs = 0; cdn = 0; }
        c0[p] = cs; c1[p] = cdn;
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* st = stage + (size_t)wave * a.stage_floats;
    const int T = cd.T, Ts = cd.T_s;
    const int iters = (T + 3) / 4;
    for (int it = 0; it < iters; ++it) {
        const int tok = it * 4 + wave;
        const bool live = tok < T;
        // first position of this token's patch in the concatenated [src | dst] position array
        const int p0 = live ? (tok < Ts ? tok * a.P : Ss + (tok - Ts) * a.P) : 0;
        for (int ch = 0; ch < 4; ++ch) {
            const int F = ch == 0 ? a.Fn : ch == 1 ? a.Fe : ch == 2 ? a.Ft : a.C;
            const int K = a.P * F;
            if (live) {
                for (int k = lane; k < K; k += kWave) {
                    const int pp = p0 + k / F, f = k % F;
                    float v;
                    if (ch == 0) v = a.node_feat[(size_t)ids[pp] * a.Fn + f];                      // DyGFormer.py:259
                    else if (ch == 1) v = a.edge_feat[(size_t)eids[pp] * a.Fe + f];                // :261
                    else if (ch == 2) v = ids[pp] == 0 ? 0.f : cosf(fmaf(dts[pp], a.time_w[f], a.time_b[f]));  // :263-266
                    else v = a.lut[(size_t)c0[pp] * a.C + f] + a.lut[(size_t)c1[pp] * a.C + f];    // :409-411
                    st[k] = v;
                }
            }
            __syncthreads();
            if (live) {
                const float* Wt = a.projT[ch];
                for (int j = lane; j < a.C; j += kWave) {
                    float acc = 0.f;
                    for (int k = 0; k < K; ++k) acc = fmaf(st[k], Wt[(size_t)k * a.C + j], acc);
                    a.X[((size_t)b * a.Tmax + tok) * a.D + ch * a.C + j] = acc + a.proj_b[ch][j];   // :148-174
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm rows (eps 1e-5, biased variance)   models/DyGFormer.py:452, :458
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ X, float* __restrict__ Y, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int64_t B, int Tmax, int D, const CallDims* cd, int64_t G) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * Tmax) return;
    if ((int)(row % Tmax) >= cd[(row / Tmax) / G].T) return;
    const float* x = X + row * D;
    float s = 0.f;
    for (int k = lane; k < D; k += kWave) s += x[k];
    const float mean = wave_sum(s) / (float)D;
    float v = 0.f;
    for (int k = lane; k < D; k += kWave) { const float d = x[k] - mean; v = fmaf(d, d, v); }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)D + 1e-5f);
    for (int k = lane; k < D; k += kWave) Y[row * D + k] = (x[k] - mean) * rstd * gamma[k] + beta[k];
}

// ------------------------------------------------------------------------------------------------
// C[M][N] (+)= act(A[M][K] * Wt[K][N] + bias)      64x64x16 LDS-tiled fp32 FMA GEMM
// ------------------------------------------------------------------------------------------------
template <bool GELU, bool RESIDUAL>
__global__ __launch_bounds__(256) void k_gemm(const float* __restrict__ A, const float* __restrict__ Wt, const float* __restrict__ bias,
                                                float* __restrict__ Cm, int64_t M, int N, int K, int Tmax, const CallDims* cd, int64_t G) {
    __shared__ float As[16][64 + 1];
    __shared__ float Bs[16][64 + 1];
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    const int n0 = blockIdx.y * 64;
    auto live = [&](int64_t r) { return r < M && (int)(r % Tmax) < cd[(r / Tmax) / G].T; };
    // skip tiles whose rows are all padding tokens
    {
        bool any = false;
        for (int i = 0; i < 64 && !any; i += 1) {
            if (live(m0 + i)) any = true;
        }
        if (!any) return;
    }
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int i = threadIdx.x; i < 64 * 16; i += 256) {
            const int r = i >> 4, kk = i & 15;
            const int64_t gr = m0 + r;
            As[kk][r] = (gr < M && k0 + kk < K) ? A[gr * K + k0 + kk] : 0.f;
        }
        for (int i = threadIdx.x; i < 16 * 64; i += 256) {
            const int kk = i >> 6, c = i & 63;
            Bs[kk][c] = (k0 + kk < K && n0 + c < N) ? Wt[(size_t)(k0 + kk) * N + n0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = Bs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t r = m0 + ty * 4 + i;
        if (!live(r)) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = n0 + tx * 4 + j;
            if (c >= N) continue;
            float v = acc[i][j] + bias[c];
            if (GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));   // exact erf GELU (F.gelu default)
            if (RESIDUAL) v += Cm[r * N + c];
            Cm[r * N + c] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// attention for one (pair, head): softmax(q/sqrt(hd) . k^T) v, no mask   (nn.MultiheadAttention)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_attention(const float* __restrict__ QKV, float* __restrict__ O, int Tmax, int D, int hd,
                                                     const CallDims* cd, int64_t G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t b = blockIdx.x;
    const int T = cd[b / G].T;
    const int h = blockIdx.y;
    const int ks = hd + 1;                       // odd row stride -> conflict-free column reads
    float* Ks = reinterpret_cast<float*>(smem);  // [Tmax][hd+1]
    float* Vs = Ks + (size_t)Tmax * ks;          // [Tmax][hd]
    float* Ps = Vs + (size_t)Tmax * hd;          // [4][Tmax]
    const float* base = QKV + (size_t)b * Tmax * 3 * D;
    for (int i = threadIdx.x; i < T * hd; i += blockDim.x) {
        const int tk = i / hd, d = i % hd;
        Ks[tk * ks + d] = base[(size_t)tk * 3 * D + D + h * hd + d];
        Vs[tk * hd + d] = base[(size_t)tk * 3 * D + 2 * D + h * hd + d];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float scale = sqrtf(1.0f / (float)hd);            // q * sqrt(1/head_dim) before q.k^T
    float* P = Ps + wave * Tmax;
    for (int q = wave; q < T; q += 4) {
        const float* qv = base + (size_t)q * 3 * D + h * hd;
        float s[2] = {-INFINITY, -INFINITY};                 // keys lane, lane+64 (T <= 128)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int key = lane + r * 64;
            if (key < T) {
                float acc = 0.f;
                for (int d = 0; d < hd; ++d) acc = fmaf(qv[d] * scale, Ks[key * ks + d], acc);
                s[r] = acc;
            }
        }
        const float mx = wave_max(fmaxf(s[0], s[1]));
        float e0 = lane < T ? expf(s[0] - mx) : 0.f;
        float e1 = lane + 64 < T ? expf(s[1] - mx) : 0.f;
        const float inv = 1.0f / wave_sum(e0 + e1);
        if (lane < T) P[lane] = e0 * inv;
        if (lane + 64 < T) P[lane + 64] = e1 * inv;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int d = lane; d < hd; d += kWave) {
            float acc = 0.f;
            for (int key = 0; key < T; ++key) acc = fmaf(P[key], Vs[key * hd + d], acc);
            O[((size_t)b * Tmax + q) * D + h * hd + d] = acc;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// ------------------------------------------------------------------------------------------------
// per-side mean over tokens + output layer   (models/DyGFormer.py:181-192)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pool_output(const float* __restrict__ X, const float* __restrict__ WoT, const float* __restrict__ bo,
                                                       float* __restrict__ out_src, float* __restrict__ out_dst, int Tmax, int D, int Fn,
                                                       const CallDims* cd, int64_t G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* mean = reinterpret_cast<float*>(smem);      // [2][D]
    const int64_t b = blockIdx.x;
    const int Ts = cd[b / G].T_s, Td = cd[b / G].T_d;
    for (int i = threadIdx.x; i < 2 * D; i += blockDim.x) {
        const int side = i / D, k = i % D;
        const int t0 = side ? Ts : 0, n = side ? Td : Ts;
        float s = 0.f;
        for (int tk = 0; tk < n; ++tk) s += X[((size_t)b * Tmax + t0 + tk) * D + k];
        mean[i] = s / (float)n;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * Fn; i += blockDim.x) {
        const int side = i / Fn, j = i % Fn;
        float acc = 0.f;
        for (int k = 0; k < D; ++k) acc = fmaf(mean[side * D + k], WoT[(size_t)k * Fn + j], acc);
        (side ? out_dst : out_src)[b * Fn + j] = acc + bo[j];
    }
}

__global__ void k_copy_seq_lens(const CallDims* cd, int32_t* out) {
    if (threadIdx.x == 0) { out[0] = cd->S_s; out[1] = cd->S_d; }
}

// ------------------------------------------------------------------------------------------------
// host drivers
// ------------------------------------------------------------------------------------------------
static int transpose_into(const float* W, float* Wt, int N, int K, hipStream_t s) {
    DYGNN_REQUIRE(W != nullptr, "pack: null weight pointer");
    hipLaunchKernelGGL(k_transpose, dim3((K + 31) / 32, (N + 31) / 32), dim3(32, 8), 0, s, W, Wt, N, K);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

int pack_generic(const Dims& d, const PackedLayout& pl, const dygnn_dygformer_weights* w, float* packed, hipStream_t s) {
    DYGNN_REQUIRE(w->cooc_w0 && w->cooc_b0 && w->cooc_w1 && w->cooc_b1, "pack: null co-occurrence weights");
    DYGNN_REQUIRE(d.C <= 1024, "pack: channel_embedding_dim too large");
    hipLaunchKernelGGL(k_cooc_lut, dim3(d.lut_rows), dim3(((d.C + 63) / 64) * 64), 0, s, w->cooc_w0, w->cooc_b0, w->cooc_w1,
                       w->cooc_b1, d.C, d.lut_rows, packed + pl.lut);
    DYGNN_LAUNCH_CHECK();
    const float* pw[4] = {w->proj_node_w, w->proj_edge_w, w->proj_time_w, w->proj_cooc_w};
    const int K[4] = {d.P * d.Fn, d.P * d.Fe, d.P * d.Ft, d.P * d.C};
    for (int c = 0; c < 4; ++c)
        if (int rc = transpose_into(pw[c], packed + pl.projT[c], d.C, K[c], s)) return rc;
    for (int l = 0; l < d.NL; ++l) {
        const dygnn_encoder_layer_weights& L = w->layers[l];
        if (int rc = transpose_into(L.in_proj_weight, packed + pl.inT[l], 3 * d.D, d.D, s)) return rc;
        if (int rc = transpose_into(L.out_proj_weight, packed + pl.outT[l], d.D, d.D, s)) return rc;
        if (int rc = transpose_into(L.ffn0_weight, packed + pl.f0T[l], 4 * d.D, d.D, s)) return rc;
        if (int rc = transpose_into(L.ffn1_weight, packed + pl.f1T[l], d.D, 4 * d.D, s)) return rc;
    }
    return transpose_into(w->output_w, packed + pl.outputT, d.Fn, d.D, s);
}

template <bool GELU, bool RES>
static int launch_gemm(const float* A, const float* Wt, const float* bias, float* C, int64_t M, int N, int K, int Tmax,
                       const CallDims* cd, int64_t G, hipStream_t s) {
    hipLaunchKernelGGL((k_gemm<GELU, RES>), dim3((unsigned)ceil_div(M, 64), (unsigned)ceil_div(N, 64)), dim3(256), 0, s, A, Wt, bias,
                       C, M, N, K, Tmax, cd, G);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

int window_lengths_device(const Dims& d, const dygnn_csr* csr, const int64_t* src, const int64_t* dst, const double* times,
                          int64_t B, int64_t G, char* ws, const WorkspaceLayout& wl, hipStream_t s) {
    CallDims* cd = reinterpret_cast<CallDims*>(ws + wl.dims);
    const int64_t ngroups = ceil_div(B, G);
    DYGNN_HIP(hipMemsetAsync(cd, 0, ngroups * sizeof(CallDims), s));
    CsrView2 g{csr->indptr, csr->nbr, csr->eid, csr->ts, csr->num_nodes};
    hipLaunchKernelGGL(k_window_lengths2, dim3((unsigned)ceil_div(2 * B, 4)), dim3(256), 0, s, g, src, dst, times, B, G, d.L,
                       reinterpret_cast<int32_t*>(ws + wl.hist_len), reinterpret_cast<int64_t*>(ws + wl.end_pos), cd);
    DYGNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_call_dims, dim3((unsigned)ceil_div(ngroups, 64)), dim3(64), 0, s, cd, d.P, ngroups);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

int forward_generic(const Dims& d, const PackedLayout& pl, const dygnn_dygformer_weights* w, const float* packed,
                    const dygnn_csr* csr, const float* node_feat, const float* edge_feat, const int64_t* src,
                    const int64_t* dst, const double* times, int64_t B, int64_t G, int64_t /*pair_stride: a fused-kernel launch option, results are the same*/, float* out_src, float* out_dst, char* ws,
                    const WorkspaceLayout& wl, const dygnn_dygformer_taps* taps, hipStream_t s) {
    DYGNN_REQUIRE(d.Tmax <= 128, "generic path supports at most 128 tokens per pair (2*ceil(L/P) = %d)", d.Tmax);
    if (int rc = window_lengths_device(d, csr, src, dst, times, B, G, ws, wl, s)) return rc;
    const CallDims* cd = reinterpret_cast<const CallDims*>(ws + wl.dims);
    float* X = reinterpret_cast<float*>(ws + wl.X);
    float* Xn = reinterpret_cast<float*>(ws + wl.Xn);
    float* QKV = reinterpret_cast<float*>(ws + wl.QKV);
    float* Hid = reinterpret_cast<float*>(ws + wl.Hid);

    EmbedArgs ea;
    ea.g = CsrView2{csr->indptr, csr->nbr, csr->eid, csr->ts, csr->num_nodes};
    ea.src = src; ea.dst = dst; ea.times = times;
    ea.hist_len = reinterpret_cast<const int32_t*>(ws + wl.hist_len);
    ea.end_pos = reinterpret_cast<const int64_t*>(ws + wl.end_pos);
    ea.cd = cd; ea.node_feat = node_feat; ea.edge_feat = edge_feat;
    ea.time_w = w->time_w; ea.time_b = w->time_b; ea.lut = packed + pl.lut;
    const float* pb[4] = {w->proj_node_b, w->proj_edge_b, w->proj_time_b, w->proj_cooc_b};
    for (int c = 0; c < 4; ++c) { ea.projT[c] = packed + pl.projT[c]; ea.proj_b[c] = pb[c]; }
    ea.X = X; ea.B = B; ea.G = G;
    ea.Fn = d.Fn; ea.Fe = d.Fe; ea.Ft = d.Ft; ea.C = d.C; ea.D = d.D; ea.P = d.P; ea.L = d.L; ea.Tmax = d.Tmax; ea.Smax = d.Smax;
    ea.lut_rows = d.lut_rows;
    int fmax = d.Fn > d.Fe ? d.Fn : d.Fe; fmax = fmax > d.Ft ? fmax : d.Ft; fmax = fmax > d.C ? fmax : d.C;
    ea.stage_floats = d.P * fmax;
    const size_t embed_lds = (size_t)2 * d.Smax * 5 * 4 + (size_t)4 * ea.stage_floats * 4;
    if (embed_lds > 160 * 1024) { set_error("generic path: window arrays + patch staging need %zu bytes of LDS (> 160 KiB)", embed_lds); return DYGNN_E_UNSUPPORTED; }
    DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_embed), hipFuncAttributeMaxDynamicSharedMemorySize, (int)embed_lds));
    hipLaunchKernelGGL(k_embed, dim3((unsigned)B), dim3(256), embed_lds, s, ea);
    DYGNN_LAUNCH_CHECK();

    const int64_t M = B * d.Tmax;
    const size_t act_bytes = (size_t)M * d.D * sizeof(float);
    if (taps && taps->seq_lens) {
        hipLaunchKernelGGL(k_copy_seq_lens, dim3(1), dim3(64), 0, s, cd, taps->seq_lens);
        DYGNN_LAUNCH_CHECK();
    }
    if (taps && taps->encoder_input) DYGNN_HIP(hipMemcpyAsync(taps->encoder_input, X, act_bytes, hipMemcpyDeviceToDevice, s));

    const size_t att_lds = ((size_t)d.Tmax * (d.hd + 1) + (size_t)d.Tmax * d.hd + 4 * d.Tmax) * sizeof(float);
    if (att_lds > 160 * 1024) { set_error("generic path: attention needs %zu bytes of LDS (> 160 KiB)", att_lds); return DYGNN_E_UNSUPPORTED; }
    DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attention), hipFuncAttributeMaxDynamicSharedMemorySize, (int)att_lds));

    for (int l = 0; l < d.NL; ++l) {
        const dygnn_encoder_layer_weights& L = w->layers[l];
        hipLaunchKernelGGL(k_layernorm, dim3((unsigned)ceil_div(M, 4)), dim3(256), 0, s, X, Xn, L.norm0_weight, L.norm0_bias, B, d.Tmax, d.D, cd, G);
        DYGNN_LAUNCH_CHECK();
        if (int rc = launch_gemm<false, false>(Xn, packed + pl.inT[l], L.in_proj_bias, QKV, M, 3 * d.D, d.D, d.Tmax, cd, G, s)) return rc;
        hipLaunchKernelGGL(k_attention, dim3((unsigned)B, d.H), dim3(256), att_lds, s, QKV, Xn, d.Tmax, d.D, d.hd, cd, G);
        DYGNN_LAUNCH_CHECK();
        if (int rc = launch_gemm<false, true>(Xn, packed + pl.outT[l], L.out_proj_bias, X, M, d.D, d.D, d.Tmax, cd, G, s)) return rc;
        hipLaunchKernelGGL(k_layernorm, dim3((unsigned)ceil_div(M, 4)), dim3(256), 0, s, X, Xn, L.norm1_weight, L.norm1_bias, B, d.Tmax, d.D, cd, G);
        DYGNN_LAUNCH_CHECK();
        if (int rc = launch_gemm<true, false>(Xn, packed + pl.f0T[l], L.ffn0_bias, Hid, M, 4 * d.D, d.D, d.Tmax, cd, G, s)) return rc;
        if (int rc = launch_gemm<false, true>(Hid, packed + pl.f1T[l], L.ffn1_bias, X, M, d.D, 4 * d.D, d.Tmax, cd, G, s)) return rc;
        if (taps && taps->layer_out[l]) DYGNN_HIP(hipMemcpyAsync(taps->layer_out[l], X, act_bytes, hipMemcpyDeviceToDevice, s));
    }
    hipLaunchKernelGGL(k_pool_output, dim3((unsigned)B), dim3(256), (size_t)2 * d.D * sizeof(float), s, X, packed + pl.outputT,
                       w->output_b, out_src, out_dst, d.Tmax, d.D, d.Fn, cd, G);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

}  // namespace dygnn
