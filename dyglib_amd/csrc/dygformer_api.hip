// C-ABI entry points for the DyGFormer forward path + the fused link-predictor head.
#include "dygformer_layout.h"
#include "gemm.h"

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


// The same head for a few hundred rows (one evaluation step: 400 .. 800).  FOUR rows per workgroup and one wave per 16-wide tile of hidden
// units (hidden = 172: 11 waves), on `v_mfma_f32_4x4x1_16b_f32` used as 4 groups of hidden units x 4 slices of k against the four rows
// (tgat_chain.hip has the long form of this): lane L = 16 ng + 4 ks + i holds fc1[16 wave + 4 ng + i][16 chunk + 4 ks ..] -- the 16 lanes of a
// quarter-wave read 4 rows x 64 contiguous bytes (the 16x16x4 form: 16 rows x 16 bytes, four times the L1 lookups per byte) -- and
// multiplies it with cat(a, b)[row j][16 chunk + 4 ks ..] out of LDS.  400 rows = 100 workgroups; a wave's stream is 22 steps.
// (The previous form, 16 rows per workgroup with 4 waves over the hidden units, took 16 us for 400 rows on 25 workgroups: every wave's
// float4 operand loads touched 16 rows, and 4 or 8 k-steps in flight made no difference.)
template <int MT>      // 4 MT rows per workgroup: 4 for an evaluation step's few hundred rows, 16 for thousands (the fc1 stream is shared by more rows)
__global__ __launch_bounds__(1024) void k_merge_sigmoid_rows4(const float* __restrict__ a, const float* __restrict__ b, int64_t n_rows, int dim,
                                                               int hidden, const float* __restrict__ w1, const float* __restrict__ b1,
                                                               const float* __restrict__ w2, const float* __restrict__ b2,
                                                               float* __restrict__ out, int sig, int ldx) {
    extern __shared__ __attribute__((aligned(16))) float mlds[];
    constexpr int R = 4 * MT;
    float* x = mlds;                      // [R][ldx] cat(a, b) rows, zero-padded to the 16-k chunk
    float* zp = x + R * ldx;              // [waves][R] per-tile parts of the fc2 dot product
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int j = lane & 3, ks = (lane >> 2) & 3, ng = lane >> 4;
    const int K = 2 * dim, nch = (K + 15) >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * R;
    const int n_lane = 16 * wave + 4 * ng + j;
    const bool rowok = n_lane < hidden;
    const float* wp = w1 + (size_t)(rowok ? n_lane : 0) * K + 4 * ks;
    constexpr int PF = 4;
    mf4 ring[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) ring[u] = *reinterpret_cast<const mf4*>(wp + ((16 * u + 4 * ks < K) ? 16 * u : 0));
    for (int rr = wave; rr < R; rr += nwaves) {
        const int64_t m = r0 + rr;
        for (int f = lane; f < ldx; f += 64) x[rr * ldx + f] = (m < n_rows && f < K) ? (f < dim ? a[m * dim + f] : b[m * dim + (f - dim)]) : 0.f;
    }
    __syncthreads();
    mf4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = mf4{0.f, 0.f, 0.f, 0.f};
    const float* xb = x + j * ldx + 4 * ks;
    for (int c0 = 0; c0 < nch; c0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int ch = c0 + u;
            if (ch < nch) {
                const bool kok = 16 * ch + 4 * ks < K;
                const mf4 w = (rowok && kok) ? ring[u] : mf4{0.f, 0.f, 0.f, 0.f};
                const int nx = ch + PF;
                ring[u] = *reinterpret_cast<const mf4*>(wp + ((nx < nch && 16 * nx + 4 * ks < K) ? 16 * nx : 0));
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const mf4 xv = *reinterpret_cast<const mf4*>(xb + m * 4 * ldx + 16 * ch);
                    acc[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(w.x, xv.x, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(w.y, xv.y, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(w.z, xv.z, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w, xv.w, acc[m], 0, 0, 0);
                }
            }
        }
    }
    // k-slices 0 + 1, 2 + 3, then the pairs; then relu(h + b1) . w2 over the lane's four hidden units, the four groups, the tiles
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[m][e] += __shfl_xor(acc[m][e], 4, 64);
            acc[m][e] += __shfl_xor(acc[m][e], 8, 64);
        }
        float z = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = 16 * wave + 4 * ng + e;
            if (n < hidden) z = fmaf(fmaxf(acc[m][e] + b1[n], 0.f), w2[n], z);
        }
        z += __shfl_xor(z, 16, 64);
        z += __shfl_xor(z, 32, 64);
        if (lane < 4) zp[wave * R + 4 * m + lane] = z;      // ng = 0, ks = 0, row 4 m + lane
    }
    __syncthreads();
    if (threadIdx.x < R && r0 + threadIdx.x < n_rows) {
        float t = 0.f;
        for (int w = 0; w < nwaves; ++w) t += zp[w * R + threadIdx.x];
        t += b2[0];
        out[r0 + threadIdx.x] = sig ? 1.0f / (1.0f + expf(-t)) : t;
    }
}

// Backward of the link predictor z = fc2(relu(fc1(cat(a, b)))) for the training step (train_link_prediction.py:241-257; models/modules.py:57-68):
// given g = dL/dz [n] it produces da, db [n][dim] and ACCUMULATES the four parameter gradients (buffers zeroed by the caller).
// k_merge_bwd_rows, one workgroup = 16 rows (the forward kernel above with a different ending):
//   * hidden pre-activations on the matrix cores, 16 rows per workgroup (wave w: three 16-wide hidden tiles from 48 w);
//   * dh = g w2 [pre > 0] leaves as rows [n][hidden] (the operand of the weight-gradient product below) and, transposed, into LDS; dfc2_w from the
//     accumulators (DPP row sums, one atomic per hidden unit and workgroup), dfc2_b likewise;
//   * dcat = dh fc1 on the matrix cores: out^T[k][row] = sum_j fc1[j][k] dh^T[j][row].  One float4 of an fc1 row (4 consecutive k) per lane feeds FOUR
//     MFMAs whose output rows are k = 4 c + t: the 16 lanes of a group cover 64 consecutive k, each lane ends up with 16 consecutive k per lane group;
// dfc1 = dh^T [a | b] (+ dfc1_b = column sums of dh) is a weight-gradient-shaped product: train::dw_grouped (k_dw_grouped), two problems.
// (Two earlier FMA-only versions — one atomic per dfc1 element, then two atomics-free kernels — took 80-90 us per call: chains of load and atomic
// latencies on 25 workgroups.)
__global__ __launch_bounds__(256) void k_merge_bwd_rows(const float* __restrict__ a, const float* __restrict__ b, int64_t n_rows, int dim, int hidden,
                                                         const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                         const float* __restrict__ gz, float* __restrict__ dh, int ldh, float* __restrict__ da,
                                                         float* __restrict__ db, float* __restrict__ dw2, float* __restrict__ db2) {
    __shared__ float dhs[192][17];             // dh^T [hidden][row] (+1: the transposed reads below hit 16 different banks)
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
    {
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
                    bq[u] = load_b(st0 + u + PF);
#pragma unroll
                    for (int i = 0; i < 3; ++i) aq[u][i] = load_a(st0 + u + PF, i);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][t], bf[t], acc[i], 0, 0, 0);
                }
            }
        }
    }
    // dh, dfc2_w, dfc2_b from the accumulators (rows = hidden units n0 + 16 i + 4 g + r, columns = the 16 rows of the workgroup)
    const float gm = mv ? gz[m] : 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        mf4 d, hr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + 16 * i + 4 * g + r;
            const float pre = n < hidden ? acc[i][r] + b1[n] : 0.f;
            d[r] = (n < hidden && pre > 0.f) ? gm * w2[n] : 0.f;
            hr[r] = fmaxf(pre, 0.f) * gm;
            dhs[n][c] = d[r];
        }
        const int nb = n0 + 16 * i + 4 * g;
        if (mv && nb < hidden) *reinterpret_cast<mf4*>(dh + m * ldh + nb) = d;          // hidden % 4 == 0: a float4 never straddles the end
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = hr[r];                    // sum over the 16 rows (lanes of one group)
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if (c == 0 && nb + r < hidden) atomicAdd(dw2 + nb + r, v);
        }
    }
    if (wave == 0) {
        float v = g == 0 ? gm : 0.f;
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        if (lane == 0) atomicAdd(db2, v);
    }
    __syncthreads();
    // dcat: sets of 64 consecutive input features k, dealt to the waves; per k-step of 4 hidden units one float4 of fc1 per lane and four MFMAs
    const int nset = (K + 63) >> 6, hsteps = (hidden + 3) >> 2;
    for (int set = wave; set < nset; set += 4) {
        const int kbase = 64 * set, kl = kbase + 4 * c;            // this lane's four features
        mf4 o[4] = {mf4{0.f, 0.f, 0.f, 0.f}, mf4{0.f, 0.f, 0.f, 0.f}, mf4{0.f, 0.f, 0.f, 0.f}, mf4{0.f, 0.f, 0.f, 0.f}};
        auto load_w = [&](int hs) -> mf4 {
            const int j = 4 * hs + g;
            return (j < hidden && kl < K) ? *reinterpret_cast<const mf4*>(w1 + (size_t)j * K + kl) : mf4{0.f, 0.f, 0.f, 0.f};
        };
        constexpr int WQ = 8;
        mf4 wq[WQ];
#pragma unroll
        for (int u = 0; u < WQ; ++u) wq[u] = load_w(u);
        for (int hs0 = 0; hs0 < hsteps; hs0 += WQ) {
#pragma unroll
            for (int u = 0; u < WQ; ++u) {
                if (hs0 + u < hsteps) {
                    const mf4 wv = wq[u];
                    wq[u] = load_w(hs0 + u + WQ);
                    const int j = 4 * (hs0 + u) + g;
                    const float bv = j < hidden ? dhs[j][c] : 0.f;      // B[k = hidden unit j][n = row c]
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t], bv, o[t], 0, 0, 0);
                }
            }
        }
        // o[t][r] = dcat[row c][k = kbase + 4 (4 g + r) + t]: 16 consecutive features per lane
        if (mv) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = kbase + 16 * g + 4 * r;
                if (k < K) {
                    const mf4 v = mf4{o[0][r], o[1][r], o[2][r], o[3][r]};
                    if (k < dim) *reinterpret_cast<mf4*>(da + m * dim + k) = v; else *reinterpret_cast<mf4*>(db + m * dim + (k - dim)) = v;
                }
            }
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
    if (dim % 4 == 0 && hidden <= 256 && n >= 64) {     // an evaluation step's worth of rows (4 per workgroup) or thousands of them (16 per workgroup)
        const int tiles = (hidden + 15) / 16, k16 = (2 * dim + 15) & ~15, ldx = (k16 & 16) ? k16 : k16 + 16;      // row stride % 32 == 16: the four rows on distinct banks
        const int R = n >= 2048 ? 16 : 4;
        const size_t lds4 = ((size_t)R * ldx + (size_t)R * tiles) * sizeof(float);
        DYGNN_REQUIRE(lds4 <= 64 * 1024, "merge_layer: dim too large");
        if (R == 16)
            hipLaunchKernelGGL(k_merge_sigmoid_rows4<4>, dim3((unsigned)ceil_div(n, 16)), dim3(64 * tiles), lds4, as_stream(stream), a, b, n, dim, hidden,
                               fc1_w, fc1_b, fc2_w, fc2_b, out, sig, ldx);
        else
            hipLaunchKernelGGL(k_merge_sigmoid_rows4<1>, dim3((unsigned)ceil_div(n, 4)), dim3(64 * tiles), lds4, as_stream(stream), a, b, n, dim, hidden,
                               fc1_w, fc1_b, fc2_w, fc2_b, out, sig, ldx);
        DYGNN_LAUNCH_CHECK();
        return DYGNN_OK;
    }
    if (dim % 4 == 0 && n >= 2048) {        // (hidden > 256) the wave-per-16-rows form
        hipLaunchKernelGGL(k_merge_sigmoid_mfma, dim3((unsigned)ceil_div(n, 64)), dim3(256), 0, as_stream(stream), a, b, n, dim, hidden,
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
    DYGNN_REQUIRE(n >= 0 && dim > 0 && hidden > 0 && dim % 4 == 0 && hidden % 4 == 0 && hidden <= 192, "merge_layer_backward: dim and hidden must be multiples of 4, hidden <= 192");
    DYGNN_REQUIRE(a && b && fc1_w && fc1_b && fc2_w && grad_logits && grad_a && grad_b && grad_fc1_w && grad_fc1_b && grad_fc2_w && grad_fc2_b && workspace,
                  "merge_layer_backward: null pointer");
    DYGNN_REQUIRE(((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0,
                  "merge_layer_backward: a, b and the workspace must be 16-byte aligned");
    if (n == 0) return DYGNN_OK;
    hipLaunchKernelGGL(k_merge_bwd_rows, dim3((unsigned)ceil_div(n, (int64_t)16)), dim3(256), 0, as_stream(stream), a, b, n, dim, hidden, fc1_w, fc1_b, fc2_w, grad_logits,
                       workspace, hidden, grad_a, grad_b, grad_fc2_w, grad_fc2_b);
    DYGNN_LAUNCH_CHECK();
    const train::DwPair pairs[2] = {{workspace, hidden, hidden, a, dim, dim, grad_fc1_w, 2 * dim, grad_fc1_b},
                                    {workspace, hidden, hidden, b, dim, dim, grad_fc1_w + dim, 2 * dim, nullptr}};
    return train::dw_grouped(as_stream(stream), (int)n, pairs, 2);
}
