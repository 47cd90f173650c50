// C-ABI entry points for the DyGFormer forward path + the fused link-predictor head.
#include "dygformer_layout.h"

namespace dygnn {
// dygformer_generic.hip
int pack_generic(const Dims&, const PackedLayout&, const dygnn_dygformer_weights*, float* packed, hipStream_t);
int forward_generic(const Dims&, const PackedLayout&, const dygnn_dygformer_weights*, const float* packed, const dygnn_csr*,
                    const float* node_feat, const float* edge_feat, const int64_t* src, const int64_t* dst, const double* times,
                    int64_t B, int64_t G, int64_t pair_stride, float* out_src, float* out_dst, char* ws, const WorkspaceLayout&,
                    const dygnn_dygformer_taps*, hipStream_t);
// dygformer_fused3.hip
bool fused3_supported(const Dims&);
int pack_fused3(const Dims&, const PackedLayout&, const dygnn_dygformer_weights*, float* packed, hipStream_t, bool reuse_desc);
int forward_fused3(const Dims&, const PackedLayout&, const dygnn_dygformer_weights*, const float* packed, const dygnn_csr*,
                   const float* node_feat, const float* edge_feat, const int64_t* src, const int64_t* dst, const double* times,
                   int64_t B, int64_t G, int64_t pair_stride, float* out_src, float* out_dst, char* ws, const WorkspaceLayout&,
                   const dygnn_dygformer_taps*, hipStream_t);

static int check_weights(const Dims& d, const dygnn_dygformer_weights* w) {
    DYGNN_REQUIRE(w != nullptr, "weights is NULL");
    const void* p[] = {w->time_w, w->time_b, w->cooc_w0, w->cooc_b0, w->cooc_w1, w->cooc_b1, w->proj_node_w, w->proj_node_b,
                       w->proj_edge_w, w->proj_edge_b, w->proj_time_w, w->proj_time_b, w->proj_cooc_w, w->proj_cooc_b,
                       w->output_w, w->output_b};
    for (const void* q : p) DYGNN_REQUIRE(q != nullptr, "weights: null parameter pointer");
    for (int l = 0; l < d.NL; ++l) {
        const dygnn_encoder_layer_weights& L = w->layers[l];
        const void* r[] = {L.in_proj_weight, L.in_proj_bias, L.out_proj_weight, L.out_proj_bias, L.ffn0_weight, L.ffn0_bias,
                           L.ffn1_weight, L.ffn1_bias, L.norm0_weight, L.norm0_bias, L.norm1_weight, L.norm1_bias};
        for (const void* q : r) DYGNN_REQUIRE(q != nullptr, "weights: null parameter pointer in layer %d", l);
    }
    return DYGNN_OK;
}

// sigmoid(fc2(relu(fc1(cat(a,b)))))   models/modules.py:57-68 + evaluate_models_utils.py:140-141
__global__ __launch_bounds__(256) void k_merge_sigmoid(const float* __restrict__ a, const float* __restrict__ b, int dim, int hidden,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2,
                                                         float* __restrict__ out, int sig) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* x = reinterpret_cast<float*>(smem);          // [2*dim]
    float* red = x + 2 * dim;                           // [4]
    const int64_t r = blockIdx.x;
    for (int i = threadIdx.x; i < 2 * dim; i += blockDim.x) x[i] = i < dim ? a[r * dim + i] : b[r * dim + (i - dim)];
    __syncthreads();
    float part = 0.f;
    for (int j = threadIdx.x; j < hidden; j += blockDim.x) {
        float acc = 0.f;
        const float* wr = w1 + (size_t)j * 2 * dim;
        for (int k = 0; k < 2 * dim; ++k) acc = fmaf(x[k], wr[k], acc);
        acc = fmaxf(acc + b1[j], 0.f);
        part = fmaf(acc, w2[j], part);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        float z = b2[0];
        for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) z += red[wv];
        out[r] = sig ? 1.0f / (1.0f + expf(-z)) : z;
    }
}

// The same head on the matrix cores: one wave = 16 rows.  Transposed product H^T[n][m] = sum_k W1[n][k] cat(a,b)[m][k]
// (A operand = fc1 rows, B operand = the row's features, both K-contiguous float4 reads), four hidden tiles at a time;
// relu, the fc2 dot product and the sigmoid are applied to the accumulators.  Needs dim % 4 == 0.
using mf4 = __attribute__((ext_vector_type(4))) float;
__global__ __launch_bounds__(256) void k_merge_sigmoid_mfma(const float* __restrict__ a, const float* __restrict__ b, int64_t n_rows, int dim,
                                                              int hidden, const float* __restrict__ w1, const float* __restrict__ b1,
                                                              const float* __restrict__ w2, const float* __restrict__ b2,
                                                              float* __restrict__ out, int sig) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int64_t m0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
    if (m0 >= n_rows) return;
    const int64_t m = m0 + c;
    const bool mv = m < n_rows;
    const int K = 2 * dim;
    float z = 0.f;
    for (int n0 = 0; n0 < hidden; n0 += 64) {
        mf4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = mf4{0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < K; k0 += 16) {
            const int kk = k0 + 4 * g;
            mf4 bf = mf4{0.f, 0.f, 0.f, 0.f};
            if (mv && kk < K) bf = kk < dim ? *reinterpret_cast<const mf4*>(a + m * dim + kk) : *reinterpret_cast<const mf4*>(b + m * dim + (kk - dim));
            mf4 af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + 16 * i + c;
                af[i] = (n < hidden && kk < K) ? *reinterpret_cast<const mf4*>(w1 + (size_t)n * K + kk) : mf4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][t], bf[t], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 16 * i + 4 * g + r;
                if (n < hidden) z = fmaf(fmaxf(acc[i][r] + b1[n], 0.f), w2[n], z);
            }
    }
    z += __shfl_xor(z, 16, 64);
    z += __shfl_xor(z, 32, 64);
    if (g == 0 && mv) out[m] = sig ? 1.0f / (1.0f + expf(-(z + b2[0]))) : z + b2[0];
}


// The same head for a few hundred rows (one evaluation step: 400 .. 800): one workgroup = 16 rows, its 4 waves split the hidden units
// (wave w: three 16-wide tiles from 48 w), so the rows' features and fc1 are read once per 16 rows instead of once per row (the
// one-workgroup-per-row kernel moves hidden x 2 dim x 4 B = 237 KB of fc1 through L2 for EVERY row) and 50 workgroups run side by side
// where the wave-per-16-rows form above would run 13.  Operands come straight from global memory / L2, four k-steps ahead in registers.
__global__ __launch_bounds__(256) void k_merge_sigmoid_mid(const float* __restrict__ a, const float* __restrict__ b, int64_t n_rows, int dim,
                                                             int hidden, const float* __restrict__ w1, const float* __restrict__ b1,
                                                             const float* __restrict__ w2, const float* __restrict__ b2,
                                                             float* __restrict__ out, int sig) {
    __shared__ float zpart[4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int64_t m = (int64_t)blockIdx.x * 16 + c;
    const bool mv = m < n_rows;
    const int K = 2 * dim, nsteps = (K + 15) >> 4;
    constexpr int PF = 4;
    const int n0 = 48 * wave;
    auto load_b = [&](int st) -> mf4 {
        const int kk = 16 * st + 4 * g;
        if (!(mv && kk < K)) return mf4{0.f, 0.f, 0.f, 0.f};
        return kk < dim ? *reinterpret_cast<const mf4*>(a + m * dim + kk) : *reinterpret_cast<const mf4*>(b + m * dim + (kk - dim));
    };
    auto load_a = [&](int st, int i) -> mf4 {
        const int kk = 16 * st + 4 * g, n = n0 + 16 * i + c;
        return (n < hidden && kk < K) ? *reinterpret_cast<const mf4*>(w1 + (size_t)n * K + kk) : mf4{0.f, 0.f, 0.f, 0.f};
    };
    mf4 acc[3] = {mf4{0.f, 0.f, 0.f, 0.f}, mf4{0.f, 0.f, 0.f, 0.f}, mf4{0.f, 0.f, 0.f, 0.f}};
    mf4 bq[PF], aq[PF][3];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        bq[u] = load_b(u);
#pragma unroll
        for (int i = 0; i < 3; ++i) aq[u][i] = load_a(u, i);
    }
    for (int st0 = 0; st0 < nsteps; st0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (st0 + u < nsteps) {
                const mf4 bf = bq[u];
                mf4 af[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) af[i] = aq[u][i];
                bq[u] = load_b(st0 + u + PF);                  // beyond K: zeros, no access
#pragma unroll
                for (int i = 0; i < 3; ++i) aq[u][i] = load_a(st0 + u + PF, i);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][t], bf[t], acc[i], 0, 0, 0);
            }
        }
    }
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + 16 * i + 4 * g + r;
            if (n < hidden) z = fmaf(fmaxf(acc[i][r] + b1[n], 0.f), w2[n], z);
        }
    z += __shfl_xor(z, 16, 64);
    z += __shfl_xor(z, 32, 64);
    if (g == 0) zpart[wave][c] = z;
    __syncthreads();
    if (wave == 0 && g == 0 && mv) {
        const float z = ((zpart[0][c] + zpart[1][c]) + (zpart[2][c] + zpart[3][c])) + b2[0];
        out[m] = sig ? 1.0f / (1.0f + expf(-z)) : z;
    }
}

// Backward of the link predictor z = fc2(relu(fc1(cat(a, b)))) for the training step (train_link_prediction.py:241-257; models/modules.py:57-68):
// given g = dL/dz [n] it produces da, db [n][dim] and the four parameter gradients.  The whole head is 0.1 GFLOP per step, so this is plain
// FMA code organised around its memory shapes, in two launches and without atomics (float atomics take ~3,000 cycles to retire under load and
// every later load of the wave waits behind them: a first single-kernel version with one atomic per dfc1 element spent 80 us on that):
//   k_merge_bwd_w : one workgroup = 8 hidden units j, ALL rows.  Per block of 256 rows: thread = row recomputes the 8 pre-activations
//                   (fc1 rows from LDS) and leaves dh = g w2 [pre > 0] in LDS and in `dh` [n][hidden]; thread = input feature k then
//                   accumulates dfc1[j][k] += dh[row][j] cat[row][k] over the block (coalesced feature reads).  Plain stores at the end;
//                   dfc1_b, dfc2_w by wave reductions, dfc2_b by workgroup 0.
//   k_merge_bwd_x : one workgroup = 8 rows: dcat[row][k] = sum_j dh[row][j] fc1[j][k], thread = k, coalesced fc1 reads.
constexpr int kMergeJ = 8, kMergeRows = 8;
__global__ __launch_bounds__(256) void k_merge_bwd_w(const float* __restrict__ a, const float* __restrict__ b, int64_t n, int dim, int hidden,
                                                      const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const float* __restrict__ gz, float* __restrict__ dh, float* __restrict__ dw1,
                                                      float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2) {
    extern __shared__ __attribute__((aligned(16))) float msm[];
    const int K = 2 * dim, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0 = blockIdx.x * kMergeJ;
    float* wj = msm;                          // [8][K] fc1 rows of this workgroup's hidden units
    float* dhs = wj + kMergeJ * K;            // [256][8]
    float* red = dhs + 256 * kMergeJ;         // [4 waves][2][8] (+ 4 for dfc2_b)
    for (int idx = tid; idx < kMergeJ * K; idx += 256) {
        const int j = idx / K;
        wj[idx] = j0 + j < hidden ? w1[(size_t)(j0 + j) * K + (idx - j * K)] : 0.f;
    }
    float bj[kMergeJ], vj[kMergeJ];
#pragma unroll
    for (int j = 0; j < kMergeJ; ++j) { bj[j] = j0 + j < hidden ? b1[j0 + j] : 0.f; vj[j] = j0 + j < hidden ? w2[j0 + j] : 0.f; }
    float accw[2][kMergeJ];                   // dfc1[j0 + j][k], k = tid and tid + 256
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < kMergeJ; ++j) accw[q][j] = 0.f;
    float sb[kMergeJ], sw[kMergeJ], sg = 0.f;  // this thread's share of dfc1_b, dfc2_w, dfc2_b
#pragma unroll
    for (int j = 0; j < kMergeJ; ++j) { sb[j] = 0.f; sw[j] = 0.f; }
    __syncthreads();
    for (int64_t r0 = 0; r0 < n; r0 += 256) {
        const int64_t row = r0 + tid;
        float pre[kMergeJ];
#pragma unroll
        for (int j = 0; j < kMergeJ; ++j) pre[j] = bj[j];
        if (row < n) {
            // these loops are chains of global-load latencies unless the loads are batched: eight float4 of the row in flight at a time
            constexpr int LB = 8;
            for (int k0 = 0; k0 < K; k0 += 4 * LB) {
                mf4 cq[LB];
#pragma unroll
                for (int u = 0; u < LB; ++u) {
                    const int k = k0 + 4 * u;
                    cq[u] = k >= K ? mf4{0.f, 0.f, 0.f, 0.f} : (k < dim ? *reinterpret_cast<const mf4*>(a + row * dim + k) : *reinterpret_cast<const mf4*>(b + row * dim + (k - dim)));
                }
#pragma unroll
                for (int u = 0; u < LB; ++u) {
                    const int k = k0 + 4 * u;
                    if (k < K) {
#pragma unroll
                        for (int j = 0; j < kMergeJ; ++j) {
                            const mf4 wv = *reinterpret_cast<const mf4*>(wj + j * K + k);
                            pre[j] = fmaf(wv.x, cq[u].x, fmaf(wv.y, cq[u].y, fmaf(wv.z, cq[u].z, fmaf(wv.w, cq[u].w, pre[j]))));
                        }
                    }
                }
            }
        }
        const float g = row < n ? gz[row] : 0.f;
        sg += g;
#pragma unroll
        for (int j = 0; j < kMergeJ; ++j) {
            const float d = pre[j] > 0.f ? g * vj[j] : 0.f;
            dhs[tid * kMergeJ + j] = d;
            if (row < n && j0 + j < hidden) dh[row * hidden + j0 + j] = d;
            sb[j] += d; sw[j] = fmaf(fmaxf(pre[j], 0.f), g, sw[j]);
        }
        __syncthreads();
        const int nr = n - r0 < 256 ? (int)(n - r0) : 256;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = tid + 256 * q;
            if (k < K) {
                const float* src = k < dim ? a + r0 * dim + k : b + r0 * dim + (k - dim);
                constexpr int RB = 16;
                for (int rb = 0; rb < nr; rb += RB) {
                    float cq[RB];
#pragma unroll
                    for (int u = 0; u < RB; ++u) cq[u] = rb + u < nr ? src[(size_t)(rb + u) * dim] : 0.f;
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const int r = rb + u < nr ? rb + u : 0;          // beyond the block: cq = 0
                        const float cv = cq[u];
                        const mf4 d0 = *reinterpret_cast<const mf4*>(dhs + r * kMergeJ), d1 = *reinterpret_cast<const mf4*>(dhs + r * kMergeJ + 4);
                        accw[q][0] = fmaf(d0.x, cv, accw[q][0]); accw[q][1] = fmaf(d0.y, cv, accw[q][1]); accw[q][2] = fmaf(d0.z, cv, accw[q][2]); accw[q][3] = fmaf(d0.w, cv, accw[q][3]);
                        accw[q][4] = fmaf(d1.x, cv, accw[q][4]); accw[q][5] = fmaf(d1.y, cv, accw[q][5]); accw[q][6] = fmaf(d1.z, cv, accw[q][6]); accw[q][7] = fmaf(d1.w, cv, accw[q][7]);
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int k = tid + 256 * q;
        if (k < K)
#pragma unroll
            for (int j = 0; j < kMergeJ; ++j)
                if (j0 + j < hidden) dw1[(size_t)(j0 + j) * K + k] = accw[q][j];
    }
#pragma unroll
    for (int j = 0; j < kMergeJ; ++j) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sb[j] += __shfl_xor(sb[j], o, 64); sw[j] += __shfl_xor(sw[j], o, 64); }
        if (lane == 0) { red[(wave * 2 + 0) * kMergeJ + j] = sb[j]; red[(wave * 2 + 1) * kMergeJ + j] = sw[j]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sg += __shfl_xor(sg, o, 64);
    if (lane == 0) red[8 * kMergeJ + wave] = sg;
    __syncthreads();
    if (tid < 2 * kMergeJ) {
        const int which = tid / kMergeJ, j = tid % kMergeJ;
        const float t = (red[(0 * 2 + which) * kMergeJ + j] + red[(1 * 2 + which) * kMergeJ + j]) + (red[(2 * 2 + which) * kMergeJ + j] + red[(3 * 2 + which) * kMergeJ + j]);
        if (j0 + j < hidden) (which ? dw2 : db1)[j0 + j] = t;
    }
    if (blockIdx.x == 0 && tid == 0) db2[0] = (red[8 * kMergeJ] + red[8 * kMergeJ + 1]) + (red[8 * kMergeJ + 2] + red[8 * kMergeJ + 3]);
}
__global__ __launch_bounds__(256) void k_merge_bwd_x(int64_t n, int dim, int hidden, const float* __restrict__ w1, const float* __restrict__ dh,
                                                      float* __restrict__ da, float* __restrict__ db) {
    extern __shared__ __attribute__((aligned(16))) float msm[];      // dh rows [8][hidden]
    const int K = 2 * dim, tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * kMergeRows;
    for (int idx = tid; idx < kMergeRows * hidden; idx += 256) {
        const int r = idx / hidden;
        msm[idx] = row0 + r < n ? dh[(row0 + r) * hidden + (idx - r * hidden)] : 0.f;
    }
    __syncthreads();
    for (int k = tid; k < K; k += 256) {
        float acc[kMergeRows];
#pragma unroll
        for (int r = 0; r < kMergeRows; ++r) acc[r] = 0.f;
        constexpr int JB = 16;               // fc1 values in flight per thread
        for (int j0 = 0; j0 < hidden; j0 += JB) {
            float wq[JB];
#pragma unroll
            for (int u = 0; u < JB; ++u) wq[u] = j0 + u < hidden ? w1[(size_t)(j0 + u) * K + k] : 0.f;
#pragma unroll
            for (int u = 0; u < JB; ++u) {
                const int j = j0 + u < hidden ? j0 + u : 0;
#pragma unroll
                for (int r = 0; r < kMergeRows; ++r) acc[r] = fmaf(msm[r * hidden + j], wq[u], acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < kMergeRows; ++r) {
            const int64_t row = row0 + r;
            if (row < n) { if (k < dim) da[row * dim + k] = acc[r]; else db[row * dim + (k - dim)] = acc[r]; }
        }
    }
}

}  // namespace dygnn

using namespace dygnn;

extern "C" size_t dygnn_dygformer_packed_bytes(const dygnn_dygformer_config* cfg) {
    if (check_config(cfg) != DYGNN_OK) return 0;
    const Dims d = make_dims(*cfg);
    return make_packed_layout(d).total * sizeof(float);
}

extern "C" size_t dygnn_dygformer_workspace_bytes(const dygnn_dygformer_config* cfg, int64_t batch) {
    if (check_config(cfg) != DYGNN_OK || batch < 0) return 0;
    const Dims d = make_dims(*cfg);
    return make_workspace_layout(d, batch).total;
}

// impl as in dygnn_dygformer_forward.  The fused kernels keep every activation on chip: they only need the per-query search results
// (a few bytes per pair); the generic path also needs its HBM activation buffers (~29 KB per token).
extern "C" size_t dygnn_dygformer_workspace_bytes_for(const dygnn_dygformer_config* cfg, int64_t batch, int32_t impl) {
    if (check_config(cfg) != DYGNN_OK || batch < 0 || !(impl == 0 || impl == 1 || impl == 3)) return 0;
    const Dims d = make_dims(*cfg);
    const WorkspaceLayout wl = make_workspace_layout(d, batch);
    const bool generic = impl == 1 || !fused3_supported(d);
    return generic ? wl.total : wl.X;
}

static int pack_impl(const dygnn_dygformer_config* cfg, const dygnn_dygformer_weights* w, void* packed, size_t packed_bytes, dygnn_stream_t stream, bool reuse_desc,
                     bool fused_only = false) {
    if (int rc = check_config(cfg)) return rc;
    const Dims d = make_dims(*cfg);
    if (int rc = check_weights(d, w)) return rc;
    const PackedLayout pl = make_packed_layout(d);
    DYGNN_REQUIRE(packed != nullptr, "pack: packed buffer is NULL");
    if (packed_bytes < pl.total * sizeof(float)) {
        set_error("pack: buffer too small (%zu < %zu bytes)", packed_bytes, pl.total * sizeof(float));
        return DYGNN_E_WORKSPACE;
    }
    if (!(fused_only && fused3_supported(d)))
        if (int rc = pack_generic(d, pl, w, static_cast<float*>(packed), as_stream(stream))) return rc;
    if (fused3_supported(d))
        if (int rc = pack_fused3(d, pl, w, static_cast<float*>(packed), as_stream(stream), reuse_desc)) return rc;
    return DYGNN_OK;
}

extern "C" int dygnn_dygformer_pack(const dygnn_dygformer_config* cfg, const dygnn_dygformer_weights* w, void* packed,
                                    size_t packed_bytes, dygnn_stream_t stream) {
    return pack_impl(cfg, w, packed, packed_bytes, stream, false);
}

extern "C" int dygnn_dygformer_repack(const dygnn_dygformer_config* cfg, const dygnn_dygformer_weights* w, void* packed,
                                      size_t packed_bytes, int32_t fused_only, dygnn_stream_t stream) {
    return pack_impl(cfg, w, packed, packed_bytes, stream, true, fused_only != 0);
}

extern "C" int dygnn_dygformer_forward(const dygnn_dygformer_config* cfg, const dygnn_dygformer_weights* w, const void* packed,
                                       const dygnn_csr* csr, const float* node_feat, const float* edge_feat, const int64_t* src,
                                       const int64_t* dst, const double* times, int64_t batch, int64_t group_size, int64_t pair_stride,
                                       float* out_src, float* out_dst, void* workspace, size_t workspace_bytes, const dygnn_dygformer_taps* taps,
                                       int32_t impl, dygnn_stream_t stream) {
    if (int rc = check_config(cfg)) return rc;
    const Dims d = make_dims(*cfg);
    if (int rc = check_weights(d, w)) return rc;
    DYGNN_REQUIRE(csr && csr->indptr && csr->num_nodes >= 1, "forward: bad csr");
    DYGNN_REQUIRE(batch >= 0, "forward: negative batch");
    DYGNN_REQUIRE(group_size >= 0, "forward: negative group_size");
    if (group_size == 0 || group_size > batch) group_size = batch;      // one group = the reference's single call
    DYGNN_REQUIRE(pair_stride == 0 || (2 * pair_stride == batch && pair_stride % group_size == 0),
                  "forward: pair_stride must be 0 or batch / 2, a whole number of groups (pairs i and i + pair_stride = the positive and negative call of one edge)");
    DYGNN_REQUIRE(packed && node_feat && edge_feat, "forward: null table / packed pointer");
    DYGNN_REQUIRE(batch == 0 || (src && dst && times && out_src && out_dst && workspace), "forward: null pointer");
    DYGNN_REQUIRE(impl == 0 || impl == 1 || impl == 3, "forward: impl must be 0 (auto), 1 (generic) or 3 (fused, token-owner layout)");
    if (batch == 0) return DYGNN_OK;
    const WorkspaceLayout wl = make_workspace_layout(d, batch);
    const size_t need = dygnn_dygformer_workspace_bytes_for(cfg, batch, impl);
    if (workspace_bytes < need) {
        set_error("forward: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
        return DYGNN_E_WORKSPACE;
    }
    const PackedLayout pl = make_packed_layout(d);
    const bool can_fuse3 = fused3_supported(d);
    if (impl == 3 && !can_fuse3) {
        set_error("forward: token-owner fused kernel does not support this shape (D=%d H=%d tokens<=%d)", d.D, d.H, d.Tmax);
        return DYGNN_E_UNSUPPORTED;
    }
    auto fn = forward_generic;
    if (impl == 3 || (impl == 0 && can_fuse3)) fn = forward_fused3;
    return fn(d, pl, w, static_cast<const float*>(packed), csr, node_feat, edge_feat, src, dst, times, batch, group_size, pair_stride, out_src, out_dst,
              static_cast<char*>(workspace), wl, taps, as_stream(stream));
}

static int merge_forward(const float* a, const float* b, int64_t n, int32_t dim, int32_t hidden, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                         const float* fc2_b, float* out, int sig, dygnn_stream_t stream) {
    DYGNN_REQUIRE(n >= 0 && dim > 0 && hidden > 0, "merge_layer: bad sizes");
    DYGNN_REQUIRE(n == 0 || (a && b && fc1_w && fc1_b && fc2_w && fc2_b && out), "merge_layer: null pointer");
    if (n == 0) return DYGNN_OK;
    if (dim % 4 == 0 && n >= 2048) {        // few rows: the one-workgroup-per-row kernel below has the shorter critical path
        hipLaunchKernelGGL(k_merge_sigmoid_mfma, dim3((unsigned)ceil_div(n, 64)), dim3(256), 0, as_stream(stream), a, b, n, dim, hidden,
                           fc1_w, fc1_b, fc2_w, fc2_b, out, sig);
        DYGNN_LAUNCH_CHECK();
        return DYGNN_OK;
    }
    if (dim % 4 == 0 && hidden <= 192 && n >= 64) {     // an evaluation step's worth of rows
        hipLaunchKernelGGL(k_merge_sigmoid_mid, dim3((unsigned)ceil_div(n, 16)), dim3(256), 0, as_stream(stream), a, b, n, dim, hidden,
                           fc1_w, fc1_b, fc2_w, fc2_b, out, sig);
        DYGNN_LAUNCH_CHECK();
        return DYGNN_OK;
    }
    const size_t lds = (size_t)(2 * dim + 4) * sizeof(float);
    DYGNN_REQUIRE(lds <= 64 * 1024, "merge_layer: dim too large");
    hipLaunchKernelGGL(k_merge_sigmoid, dim3((unsigned)n), dim3(256), lds, as_stream(stream), a, b, dim, hidden, fc1_w, fc1_b, fc2_w,
                       fc2_b, out, sig);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

extern "C" int dygnn_merge_layer_sigmoid(const float* a, const float* b, int64_t n, int32_t dim, int32_t hidden,
                                         const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b,
                                         float* out, dygnn_stream_t stream) {
    return merge_forward(a, b, n, dim, hidden, fc1_w, fc1_b, fc2_w, fc2_b, out, 1, stream);
}

extern "C" int dygnn_merge_layer_logits(const float* a, const float* b, int64_t n, int32_t dim, int32_t hidden,
                                        const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b,
                                        float* out, dygnn_stream_t stream) {
    return merge_forward(a, b, n, dim, hidden, fc1_w, fc1_b, fc2_w, fc2_b, out, 0, stream);
}

extern "C" int dygnn_merge_layer_backward(const float* a, const float* b, int64_t n, int32_t dim, int32_t hidden, const float* fc1_w, const float* fc1_b,
                                          const float* fc2_w, const float* grad_logits, float* grad_a, float* grad_b, float* grad_fc1_w, float* grad_fc1_b,
                                          float* grad_fc2_w, float* grad_fc2_b, float* workspace, dygnn_stream_t stream) {
    DYGNN_REQUIRE(n >= 0 && dim > 0 && hidden > 0 && dim % 4 == 0 && 2 * dim <= 512, "merge_layer_backward: bad sizes (dim: a multiple of 4, at most 256)");
    DYGNN_REQUIRE(a && b && fc1_w && fc1_b && fc2_w && grad_logits && grad_a && grad_b && grad_fc1_w && grad_fc1_b && grad_fc2_w && grad_fc2_b && workspace,
                  "merge_layer_backward: null pointer");
    const size_t lds_w = (size_t)(kMergeJ * 2 * dim + 256 * kMergeJ + 8 * kMergeJ + 4) * sizeof(float), lds_x = (size_t)kMergeRows * hidden * sizeof(float);
    DYGNN_REQUIRE(lds_w <= 64 * 1024 && lds_x <= 64 * 1024, "merge_layer_backward: dim / hidden too large");
    hipLaunchKernelGGL(k_merge_bwd_w, dim3((unsigned)ceil_div(hidden, kMergeJ)), dim3(256), lds_w, as_stream(stream), a, b, n, dim, hidden, fc1_w, fc1_b, fc2_w,
                       grad_logits, workspace, grad_fc1_w, grad_fc1_b, grad_fc2_w, grad_fc2_b);
    DYGNN_LAUNCH_CHECK();
    if (n == 0) return DYGNN_OK;
    hipLaunchKernelGGL(k_merge_bwd_x, dim3((unsigned)ceil_div(n, (int64_t)kMergeRows)), dim3(256), lds_x, as_stream(stream), n, dim, hidden, fc1_w, workspace, grad_a,
                       grad_b);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}
