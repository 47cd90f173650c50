// TGAT forward (BASELINE config 3; reference models/TGAT.py:48-136 + MultiHeadAttention models/modules.py:99-206 +
// MergeLayer models/modules.py:42-68), eval mode.
//
// The reference recurses per node set: layer-l embeddings of a set need layer-(l-1) embeddings of the set itself and of
// its k most recent neighbours.  Here the recursion is unrolled top-down into level sets (level L = the 2B query nodes,
// level l-1 = [level-l nodes ; their k neighbours]; neighbour times are the float32 values the sampler returns,
// models/TGAT.py:107-110) and evaluated bottom-up, one batched pass per layer:
//   sample_recent -> query projection -> W_k^T q per head -> attention over the k neighbours' INPUT rows (gathered on the
//   fly; K and V are never materialised, see k_tgat_attn_lin) -> W_v z per head -> residual_fc + residual + LayerNorm ->
//   MergeLayer.  All products go through the library's general fp32-MFMA GEMM (gemm.h).
#include <cstdlib>
#include "common.h"
#include "gemm.h"
#include "tgat_chain.h"
#include "tgat_attn.h"

namespace dygnn {

using f4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f4 tmfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// sampler kernel shared with sampler.hip (one wave per query, 64-ary search)
__device__ __forceinline__ int64_t wave_lower_bound3(const double* __restrict__ ts, int64_t lo, int64_t hi, double t, int lane) {
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

// level expansion: for the n nodes of a level (ids/times), sample the k most recent neighbours (utils/utils.py:200-209)
// and append them to the next-lower level: lower = [this level ; neighbours (row-major n x k)].
// nbr ids / edge ids int32, times float64 holding float32-rounded values (models/TGAT.py:107-110).
// TGN (MemoryModel.py:108-109, :609): the level-0 set of a call is what it reads.  Every level-0 slot (entry q, position j) stores
// owner[id] = slot -- plain stores, some slot of a node wins; chain::pack's list blocks (tgat_chain.h: ListArgs) let the winner list its
// node.  (Flags claimed with returning atomics cost 25 us per step here: 8,800 device-scope atomics on a 28-KB array.)
struct TgnTouch {
    int32_t* owner;                // [N]; entries of nodes outside this call's level-0 set are stale and never read
    int32_t* counts;               // the two list lengths, zeroed here for the list pass of the next launch
    int64_t N;
};
// `src` given: this is the top level [src ; dst] read straight from the caller's int64 / float64 arrays (B pairs)
__global__ __launch_bounds__(256) void k_tgat_expand(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cnbr,
                                                       const int32_t* __restrict__ ceid, const double* __restrict__ cts, int64_t num_nodes,
                                                       const int32_t* __restrict__ ids, const double* __restrict__ times, int64_t n, int k,
                                                       int32_t* __restrict__ lower_ids, double* __restrict__ lower_times,
                                                       int32_t* __restrict__ nbr_eid, float* __restrict__ nbr_dt, const int32_t* __restrict__ n_live,
                                                       const int64_t* __restrict__ src, const int64_t* __restrict__ dst, const double* __restrict__ tq,
                                                       int64_t B, const TgnTouch tt, int per_root) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tt.owner && blockIdx.x == 0 && threadIdx.x < 2) tt.counts[threadIdx.x] = 0;
    if (q >= n || (n_live && q >= *n_live)) return;      // n = layout size of the level, *n_live = entries in use (de-duplicated level)
    int64_t node = src ? (q < B ? src[q] : dst[q - B]) : (int64_t)ids[q];
    if (node < 0 || node >= num_nodes) node = 0;
    const double t = src ? tq[(per_root || q < B) ? q : q - B] : times[q];
    const int64_t lo = indptr[node], hi = indptr[node + 1];
    const int64_t i = wave_lower_bound3(cts, lo, hi, t, lane);
    const int64_t len = i - lo;
    const int m = (int)(len < k ? len : k), pad = k - m;
    if (lane == 0) {
        lower_ids[q] = (int32_t)node; lower_times[q] = t;
        if (tt.owner && node < tt.N) tt.owner[node] = (int32_t)(q * (k + 1) + k);
    }
    for (int j = lane; j < k; j += kWave) {
        int32_t nb = 0, e = 0;
        float tn = 0.f;
        if (j >= pad) {
            const int64_t p = i - m + (j - pad);
            nb = cnbr[p]; e = ceid[p]; tn = (float)cts[p];              // float32 on store, utils/utils.py:167
        }
        lower_ids[n + q * k + j] = nb;
        lower_times[n + q * k + j] = (double)tn;                        // hop-(l+1) queries use the float32 time
        nbr_eid[q * k + j] = e;
        nbr_dt[q * k + j] = (float)(t - (double)tn);                    // models/TGAT.py:116-119: f64 - f32 -> f64 -> .float()
        if (tt.owner && nb >= 0 && nb < tt.N) tt.owner[nb] = (int32_t)(q * (k + 1) + j);
    }
}

// TGN on PRE-SAMPLED levels (random sampling strategies): what k_tgat_expand's `tt` does while it writes level 0 — every level-0 slot names
// itself owner of its node (slot code: level-1 entry q, neighbour column j, or k for the entry itself), the two list counters are zeroed
__global__ void k_tgn_touch_levels(const int32_t* __restrict__ ids0, int64_t n1, int k, TgnTouch tt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2) tt.counts[i] = 0;
    if (i >= n1 * (k + 1)) return;
    const int32_t node = ids0[i];
    if (node < 0 || node >= tt.N) return;
    if (i < n1) tt.owner[node] = (int32_t)(i * (k + 1) + k);
    else { const int64_t q = (i - n1) / k; tt.owner[node] = (int32_t)(q * (k + 1) + (i - n1 - q * k)); }
}

// the query-input rows alone, [n][Fn+Ft] = [h(self) | cos(b)]: one wave per row, four rows per workgroup, float4 copies (its predecessor, one
// 256-thread workgroup per 1.1-KB row, ran a 119 k-row level at 1.3 TB/s: 163 us) (models/modules.py:150-157)
__global__ __launch_bounds__(256) void k_tgat_qrows(const float* __restrict__ h_lower, const float* __restrict__ node_feat, const int32_t* __restrict__ lower_ids,
                                                      const float* __restrict__ tw, const float* __restrict__ tb, int64_t n, int Fn, int Ft,
                                                      float* __restrict__ q_in, const int32_t* __restrict__ n_live, const int32_t* __restrict__ lower_map) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n || (n_live && i >= *n_live)) return;
    const float* hsrc = h_lower ? h_lower + (lower_map ? (int64_t)lower_map[i] : i) * Fn : node_feat + (size_t)lower_ids[i] * Fn;
    float* o = q_in + i * (Fn + Ft);
    for (int x = lane; x < (Fn >> 2); x += kWave) *reinterpret_cast<f4*>(o + 4 * x) = *reinterpret_cast<const f4*>(hsrc + 4 * x);
    for (int f = lane; f < Ft; f += kWave) o[Fn + f] = cosf(fmaf(0.0f, tw[f], tb[f]));      // the encoding of dt = 0 (models/TGAT.py:84)
}

using attn::cos_time_t;

// C[M][N] = act(A[M][K] . W[N][K]^T + bias): fp32 MFMA, operands straight from global memory (both are K-contiguous, so
// lane (c,g) of a 16x16x4 fragment reads the float4 at [row c][k0 + 4g]); a wave computes a 64 x 64 block (4x4 tiles).
// Transposed product: accumulator tile = C^T[n = 4g+r][m = c]  ->  float4 store at C[m][n0 + 4g].
template <bool RELU>
__global__ __launch_bounds__(256) void k_gemm_nt(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                                                   float* __restrict__ C, int64_t M, int N, int K, int ldc, int ldw = 0) {
    if (ldw == 0) ldw = K;                   // row stride of W (a product over the first K columns of wider rows passes it)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int64_t m0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
    const int n0 = blockIdx.y * 64;
    if (m0 >= M) return;
    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 16) {
        const int kk = k0 + 4 * g;
        f4 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + 16 * i + c;
            a[i] = (n < N && kk < K) ? *reinterpret_cast<const f4*>(W + (size_t)n * ldw + kk) : f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t m = m0 + 16 * j + c;
            b[j] = (m < M && kk < K) ? *reinterpret_cast<const f4*>(A + (size_t)m * K + kk) : f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = tmfma(a[i].x, b[j].x, acc[i][j]);
                acc[i][j] = tmfma(a[i].y, b[j].y, acc[i][j]);
                acc[i][j] = tmfma(a[i].z, b[j].z, acc[i][j]);
                acc[i][j] = tmfma(a[i].w, b[j].w, acc[i][j]);
            }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + 16 * i + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t m = m0 + 16 * j + c;
            if (m >= M) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r < N) {
                    float v = acc[i][j][r] + (bias ? bias[n + r] : 0.f);
                    if (RELU) v = fmaxf(v, 0.f);
                    C[(size_t)m * ldc + n + r] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Attention over the k neighbours WITHOUT materialising K and V.  The key / value projections are bias-free linear maps
// (models/modules.py:126-128), so
//     score_ijh = q_ih . (W_k,h x_ij) = (W_k,h^T q_ih) . x_ij          and          o_ih = sum_j p_ijh (W_v,h x_ij) = W_v,h (sum_j p_ijh x_ij)
// with x_ij = [h(nbr) | edge | cos(w dt + b)] the neighbour's input row: the two [n*k, 444] x [444, 272] products (81 GFLOP per
// layer-1 pass at Reddit size) become two [n, 136] x [136, 444] products per head (4 GFLOP) around this kernel, which is
// bound by gathering the input rows twice.  One wave per node; lanes sweep a row as float4 (Dkv/4 <= 128 columns).
// qk [n][H][Dkv] = W_k,h^T q_ih ; z [n][H][Dkv] = sum_j p_ijh x_ij.
// ------------------------------------------------------------------------------------------------
template <int KCACHE>      // > 0: the k <= KCACHE input rows stay in registers between the score pass and the weighted sum (one gather)
__global__ __launch_bounds__(256) void k_tgat_attn_lin(const float* __restrict__ qk, const float* __restrict__ h_lower, const float* __restrict__ node_feat,
                                                         const float* __restrict__ edge_feat, const int32_t* __restrict__ lower_ids,
                                                         const int32_t* __restrict__ nbr_eid, const float* __restrict__ nbr_dt, const float* __restrict__ tw,
                                                         const float* __restrict__ tb, int64_t n, int k, int Fn, int Fe, int Ft, int H, float scale,
                                                         float* __restrict__ z) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= n) return;
    float* pw = reinterpret_cast<float*>(smem) + wave * H * k;       // [H][k] scores -> probabilities
    const int Dkv = Fn + Fe + Ft, D4 = Dkv >> 2;
    const int x0 = lane, x1 = lane + 64;
    const bool v0 = x0 < D4, v1 = x1 < D4;
    auto fetch = [&](int64_t r, int x) -> f4 {                       // float4 column x of the input row of neighbour entry r
        const int kk = 4 * x;
        if (kk < Fn) {
            const int64_t le = n + r;
            const float* hp = h_lower ? h_lower + le * Fn : node_feat + (size_t)lower_ids[le] * Fn;
            return *reinterpret_cast<const f4*>(hp + kk);
        }
        if (kk < Fn + Fe) return *reinterpret_cast<const f4*>(edge_feat + (size_t)nbr_eid[r] * Fe + (kk - Fn));
        const int f = kk - Fn - Fe;
        const float dt = nbr_dt[r];
        const f4 w = *reinterpret_cast<const f4*>(tw + f), b = *reinterpret_cast<const f4*>(tb + f);
        return f4{cos_time_t(fmaf(dt, w.x, b.x)), cos_time_t(fmaf(dt, w.y, b.y)), cos_time_t(fmaf(dt, w.z, b.z)), cos_time_t(fmaf(dt, w.w, b.w))};
    };
    auto dot4 = [](const f4 a, const f4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); };
    const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
    f4 xs[KCACHE > 0 ? KCACHE : 1][2];
    if (KCACHE > 0) {
        // all k rows are requested back to back (one memory latency for the whole neighbourhood), then every (row, head)
        // score is reduced: 2*k independent butterfly chains that pipeline instead of one chain per loop iteration
        // Branch-free gathers: which table a lane reads (node / edge / none) depends only on its column, so every lane
        // issues exactly one unconditional float4 load per row and pass (lanes of the time columns and beyond read row 0 of
        // the edge table and discard it); the row indices are wave-uniform scalars.  The cosines are filled in afterwards.
        const float* ntab = h_lower ? h_lower : node_feat;
        const float* bp[2]; size_t st[2]; int cls[2];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int kk = 4 * (lane + 64 * ps);
            cls[ps] = kk < Fn ? 0 : kk < Fn + Fe ? 1 : kk < Dkv ? 2 : 3;
            bp[ps] = cls[ps] == 0 ? ntab + kk : cls[ps] == 1 ? edge_feat + (kk - Fn) : edge_feat;
            st[ps] = cls[ps] == 0 ? (size_t)Fn : cls[ps] == 1 ? (size_t)Fe : 0;
        }
#pragma unroll
        for (int j = 0; j < KCACHE; ++j) {
            const int64_t r = i * k + (j < k ? j : 0);
            const int64_t nrow = h_lower ? n + r : (int64_t)lower_ids[n + r];
            const int64_t erow = nbr_eid[r];
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) xs[j][ps] = *reinterpret_cast<const f4*>(bp[ps] + (cls[ps] == 0 ? nrow : erow) * st[ps]);
        }
        // time encoding: computed in a ROLLED loop into LDS (the cosine code exists once; unrolled over 20 rows it alone was
        // 100 KB of straight-line code, more than the instruction cache, executed once per wave), then read back per row
        float* tf = reinterpret_cast<float*>(smem) + 4 * H * k + (size_t)wave * KCACHE * Ft;          // [KCACHE][Ft]
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            if (cls[ps] == 2) {
                const int f = 4 * (lane + 64 * ps) - Fn - Fe;
                const f4 w = *reinterpret_cast<const f4*>(tw + f), b = *reinterpret_cast<const f4*>(tb + f);
#pragma unroll 1
                for (int j = 0; j < k; ++j) {
                    const float dt = nbr_dt[i * k + j];
                    *reinterpret_cast<f4*>(tf + j * Ft + f) = f4{cos_time_t(fmaf(dt, w.x, b.x)), cos_time_t(fmaf(dt, w.y, b.y)), cos_time_t(fmaf(dt, w.z, b.z)),
                                                                cos_time_t(fmaf(dt, w.w, b.w))};
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            if (cls[ps] == 2) {
                const int f = 4 * (lane + 64 * ps) - Fn - Fe;
#pragma unroll
                for (int j = 0; j < KCACHE; ++j) xs[j][ps] = *reinterpret_cast<const f4*>(tf + (j < k ? j : 0) * Ft + f);
            } else if (cls[ps] == 3) {
#pragma unroll
                for (int j = 0; j < KCACHE; ++j) xs[j][ps] = zero;
            }
        }
#pragma unroll
        for (int j = 0; j < KCACHE; ++j)
            if (j >= k) { xs[j][0] = zero; xs[j][1] = zero; }
        for (int h = 0; h < H; ++h) {
            const f4* qh = reinterpret_cast<const f4*>(qk + ((size_t)i * H + h) * Dkv);
            const f4 qa = v0 ? qh[x0] : zero, qb = v1 ? qh[x1] : zero;
            float sc[KCACHE > 0 ? KCACHE : 1];
#pragma unroll
            for (int j = 0; j < KCACHE; ++j) sc[j] = dot4(qa, xs[j][0]) + dot4(qb, xs[j][1]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                for (int j = 0; j < KCACHE; ++j) sc[j] += __shfl_xor(sc[j], o, 64);
#pragma unroll
            for (int j = 0; j < KCACHE; ++j)
                if (lane == j && j < k) pw[h * k + j] = lower_ids[n + i * k + j] == 0 ? -1e10f : sc[j] * scale;      // modules.py:173, :176-184
        }
    }
#pragma unroll
    for (int j = 0; j < (KCACHE > 0 ? 0 : 1 << 30); ++j) {
        if (j >= k) break;
        const int64_t r = i * k + j;
        const f4 xa = v0 ? fetch(r, x0) : zero, xb = v1 ? fetch(r, x1) : zero;
        for (int h = 0; h < H; ++h) {
            const f4* qh = reinterpret_cast<const f4*>(qk + ((size_t)i * H + h) * Dkv);
            float sc = (v0 ? dot4(qh[x0], xa) : 0.f) + (v1 ? dot4(qh[x1], xb) : 0.f);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sc += __shfl_xor(sc, o, 64);
            if (lane == 0) {
                sc *= scale;                                              // modules.py:173
                if (lower_ids[n + r] == 0) sc = -1e10f;                   // modules.py:176-184
                pw[h * k + j] = sc;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < H) {
        float mx = -INFINITY;
        for (int j = 0; j < k; ++j) mx = fmaxf(mx, pw[lane * k + j]);
        float sum = 0.f;
        for (int j = 0; j < k; ++j) { const float e = expf(pw[lane * k + j] - mx); pw[lane * k + j] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int j = 0; j < k; ++j) pw[lane * k + j] *= inv;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int h = 0; h < H; ++h) {
        f4 za = zero, zb = zero;
#pragma unroll
        for (int j = 0; j < (KCACHE > 0 ? KCACHE : 1 << 30); ++j) {
            if (j >= k) break;
            const int64_t r = i * k + j;
            const float p = pw[h * k + j];
            if (KCACHE > 0) {
                const f4 xa = xs[j][0], xb = xs[j][1];
                za.x = fmaf(p, xa.x, za.x); za.y = fmaf(p, xa.y, za.y); za.z = fmaf(p, xa.z, za.z); za.w = fmaf(p, xa.w, za.w);
                zb.x = fmaf(p, xb.x, zb.x); zb.y = fmaf(p, xb.y, zb.y); zb.z = fmaf(p, xb.z, zb.z); zb.w = fmaf(p, xb.w, zb.w);
                continue;
            }
            if (v0) { const f4 xv = fetch(r, x0); za.x = fmaf(p, xv.x, za.x); za.y = fmaf(p, xv.y, za.y); za.z = fmaf(p, xv.z, za.z); za.w = fmaf(p, xv.w, za.w); }
            if (v1) { const f4 xv = fetch(r, x1); zb.x = fmaf(p, xv.x, zb.x); zb.y = fmaf(p, xv.y, zb.y); zb.z = fmaf(p, xv.z, zb.z); zb.w = fmaf(p, xv.w, zb.w); }
        }
        f4* zo = reinterpret_cast<f4*>(z + ((size_t)i * H + h) * Dkv);
        if (v0) zo[x0] = za;
        if (v1) zo[x1] = zb;
    }
}

// The same attention with TWO waves per node (k <= KC, H <= 2, Dkv <= 512; tgat_attn.h: attn::pair_node).  Workgroup = 2 nodes.
template <int KC, bool FULL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_tgat_attn_pair(const float* __restrict__ qk, const float* __restrict__ h_lower, const float* __restrict__ node_feat,
                                                          const float* __restrict__ edge_feat, const int32_t* __restrict__ lower_ids,
                                                          const int32_t* __restrict__ nbr_eid, const float* __restrict__ nbr_dt, const float* __restrict__ tw,
                                                          const float* __restrict__ tb, int64_t n, int k, int Fn, int Fe, int Ft, int H, float scale,
                                                          float* __restrict__ z, const int32_t* __restrict__ n_live, const int32_t* __restrict__ lower_map) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t i = (int64_t)blockIdx.x * 2 + (wave >> 1);
    const int64_t nl = n_live ? (int64_t)*n_live : n;      // entries in use (a de-duplicated level keeps the layout size n)
    if ((int64_t)blockIdx.x * 2 >= nl) return;              // whole workgroup beyond the live entries
    const bool live = i < nl;
    if (!live) i = nl - 1;                                  // keeps the barrier uniform; nothing is written
    const int Dkv = Fn + Fe + Ft;
    attn::pair_node<KC, FULL>(qk, h_lower, node_feat, edge_feat, lower_ids, nbr_eid, nbr_dt, tw, tb, n, k, Fn, Fe, Ft, H, scale, lower_map, i, live, false, wave, 4, lane,
                        reinterpret_cast<float*>(smem), z + (size_t)i * H * Dkv, Dkv);
}

// The query input of a node is [h(node) | timeenc(0)] (models/TGAT.py:84, models/modules.py:150-157) and timeenc(0) = cos(b) is the SAME vector
// for every row: q = W_q [h | cos(b)] = W_q[:, :Fn] h + W_q[:, Fn:] cos(b).  The second term is a constant of the call (cq, the bias of the
// product over the Fn feature columns, which gathers its rows itself: no [n][Fn + Ft] query-input matrix is written or read), and the
// residual branch of the layer (models/modules.py:196-199) rebuilds [h | cos(b)] from the feature row and ct = cos(b).
__global__ void k_tgat_const_q(const float* __restrict__ Wq, const float* __restrict__ tw, const float* __restrict__ tb, int Dq, int Fn, int Ft,
                               float* __restrict__ cq, float* __restrict__ ct) {
    // one wave per output row (4 per workgroup): lanes stride over the Ft time columns (coalesced), partial sums meet by shuffles in a fixed order
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0) for (int f = threadIdx.x; f < Ft; f += blockDim.x) ct[f] = cosf(fmaf(0.0f, tw[f], tb[f]));
    if (j >= Dq) return;
    float a = 0.f;
    for (int f = lane; f < Ft; f += 64) a = fmaf(Wq[(size_t)j * Dq + Fn + f], cosf(fmaf(0.0f, tw[f], tb[f])), a);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0) cq[j] = a;
}

// y = LayerNorm(fc_out + residual) (models/modules.py:196-199), written into the first Dq columns of the MergeLayer input
// row [Dq + Fn]; the raw node features fill the rest (models/TGAT.py:134, models/modules.py:64).  residual = [h(node) | cos(b)]: h from the
// layer below (h_lower, through the row map of a de-duplicated level) or, for layer 1, the raw feature row itself
__global__ __launch_bounds__(256) void k_tgat_post(const float* __restrict__ fc_out, const float* __restrict__ h_lower, const int32_t* __restrict__ lower_map,
                                                     const float* __restrict__ ct, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ node_feat,
                                                     const int32_t* __restrict__ lower_ids, int64_t n, int Dq, int Fn, float* __restrict__ merge_in,
                                                     const int32_t* __restrict__ n_live) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n || (n_live && i >= *n_live)) return;
    // the row lives in registers (Dq <= 272: five elements per lane) between the three passes; a lane adds its elements in the order of the
    // three-loop form this replaces, so the bits are the same
    constexpr int NV = 5;
    float x[NV];
    const float* raw = node_feat + (size_t)lower_ids[i] * Fn;
    const float* hrow = h_lower ? h_lower + (lower_map ? (int64_t)lower_map[i] : i) * Fn : raw;
    float rv[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int f = lane + kWave * u;
        rv[u] = f < Fn ? raw[f] : 0.f;
        const float res = f < Fn ? (h_lower ? hrow[f] : rv[u]) : (f < Dq ? ct[f - Fn] : 0.f);
        x[u] = f < Dq ? fc_out[i * Dq + f] + res : 0.f;
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u)
        if (lane + kWave * u < Dq) s += x[u];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)Dq;
    float v = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u)
        if (lane + kWave * u < Dq) { const float d = x[u] - mean; v = fmaf(d, d, v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const float rstd = 1.0f / sqrtf(v / (float)Dq + 1e-5f);
    float* o = merge_in + i * (Dq + Fn);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int f = lane + kWave * u;
        if (f < Dq) o[f] = (x[u] - mean) * rstd * gamma[f] + beta[f];
        if (f < Fn) o[Dq + f] = rv[u];
    }
}

// ---- de-duplication of a level (recent sampling only) --------------------------------------------------------------------
// Entries of a level are (node, time) queries.  With `recent` sampling an entry's embedding is a function of that pair alone, and
// the level below the roots repeats itself: consecutive interactions of a node share 19 of their 20 most recent neighbours, and
// both endpoints of an edge list it (at Reddit shape, 16 steps per call: 128,000 queries, 46,000 distinct).  Distinct entries
// are found with an open-addressing table keyed by (node, time bits): the first thread to claim a slot is the representative,
// every other entry with an equal key maps to it.  Which duplicate wins is a race, but all of them would compute the same row,
// so results do not depend on it.  canon[i] = representative entry; the representatives are then numbered (cidx) and copied
// into the compact level (cids, ctimes); map[i] = compact index of entry i.
__device__ __forceinline__ uint32_t dedup_hash(int32_t id, double t) {
    uint64_t z = (uint64_t)__double_as_longlong(t) ^ ((uint64_t)(uint32_t)id * 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)(z ^ (z >> 31));
}
__global__ void k_dedup_insert(const int32_t* __restrict__ ids, const double* __restrict__ times, int64_t n, int32_t* __restrict__ slots, uint32_t cap_mask,
                               int32_t* __restrict__ canon) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t id = ids[i];
    const double t = times[i];
    uint32_t h = dedup_hash(id, t) & cap_mask;
    for (;;) {
        // look before claiming: a level repeats its popular (node, time) entries hundreds of times, and a compare-and-swap per duplicate on the
        // representative's slot serialises them (~8 ns each); a slot never changes once claimed, so a plain read of a claimed slot is final
        int32_t prev = __atomic_load_n(&slots[h], __ATOMIC_RELAXED);
        if (prev == -1) prev = atomicCAS(&slots[h], -1, (int32_t)i);
        if (prev == -1) { canon[i] = (int32_t)i; return; }
        if (ids[prev] == id && __double_as_longlong(times[prev]) == __double_as_longlong(t)) { canon[i] = prev; return; }
        h = (h + 1) & cap_mask;
    }
}
__global__ void k_dedup_number(const int32_t* __restrict__ ids, const double* __restrict__ times, const int32_t* __restrict__ canon, int64_t n,
                               int32_t* __restrict__ count, int32_t* __restrict__ cidx, int32_t* __restrict__ cids, double* __restrict__ ctimes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool rep = i < n && canon[i] == (int32_t)i;
    // ONE atomic per 1024-thread workgroup reserves the compact indices of its representatives: lanes count inside their wave (ballot), waves
    // inside the workgroup (LDS).  Same-address device atomics cost ~8 ns each here: one per representative was 54 us for 10^5 of them, one
    // per wave still 51 us.
    __shared__ int32_t s_cnt, s_base;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint64_t m = __ballot(rep);
    int32_t off = 0;
    if (lane == 0 && m) off = atomicAdd(&s_cnt, __popcll(m));
    off = __shfl(off, 0, 64);
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_cnt ? atomicAdd(count, s_cnt) : 0;
    __syncthreads();
    if (!rep) return;
    const int32_t c = s_base + off + __popcll(m & ((1ull << lane) - 1));
    cidx[i] = c;
    cids[c] = ids[i];
    ctimes[c] = times[i];
}
__global__ void k_dedup_map(const int32_t* __restrict__ canon, const int32_t* __restrict__ cidx, int64_t n, int32_t* __restrict__ map) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) map[i] = cidx[canon[i]];
}

__global__ void k_split_out(const float* __restrict__ h, int64_t B, int Fn, float* __restrict__ out_src, float* __restrict__ out_dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * B * Fn) return;
    const int64_t r = i / Fn;
    (r < B ? out_src : out_dst)[(r < B ? r : r - B) * Fn + i % Fn] = h[i];
}

struct TgatPlan {
    int L, k, Fn, Fe, Ft, H, hd, Dq, Dkv;
    int64_t n[DYGNN_MAX_LAYERS + 1];       // level sizes: n[L] = 2B, n[l-1] = n[l] * (1 + k)
    // byte offsets
    size_t ids[DYGNN_MAX_LAYERS + 1], times[DYGNN_MAX_LAYERS + 1], eid[DYGNN_MAX_LAYERS + 1], dt[DYGNN_MAX_LAYERS + 1], h[DYGNN_MAX_LAYERS + 1];
    size_t q_in, q, att, fc, merge_in, hid, qk, z, pack, cq, total;
    // de-duplication of level L-1 (two-layer models, recent sampling): hash slots, representative / compact index / map per entry, compact level
    size_t dd_slots, dd_canon, dd_cidx, dd_map, dd_count, dd_ids, dd_times;
    uint32_t dd_cap;
};

static TgatPlan make_tgat_plan(const dygnn_tgat_config& c, int64_t B) {
    TgatPlan p{};
    p.L = c.num_layers; p.k = c.num_neighbors; p.Fn = c.node_feat_dim; p.Fe = c.edge_feat_dim; p.Ft = c.time_feat_dim; p.H = c.num_heads;
    p.Dq = p.Fn + p.Ft; p.Dkv = p.Fn + p.Fe + p.Ft; p.hd = p.H > 0 ? p.Dq / p.H : 0;
    p.n[p.L] = 2 * B;
    for (int l = p.L; l >= 1; --l) p.n[l - 1] = p.n[l] * (1 + p.k);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~size_t(255); return r; };
    for (int l = 0; l <= p.L; ++l) {
        p.ids[l] = take(p.n[l] * sizeof(int32_t));
        p.times[l] = take(p.n[l] * sizeof(double));
        p.h[l] = l >= 1 ? take((size_t)p.n[l] * p.Fn * sizeof(float)) : 0;
        p.eid[l] = l >= 1 ? take((size_t)p.n[l] * p.k * sizeof(int32_t)) : 0;
        p.dt[l] = l >= 1 ? take((size_t)p.n[l] * p.k * sizeof(float)) : 0;
    }
    const int64_t nmax = p.L >= 1 ? p.n[1] : 0;                        // the largest computed level
    p.q_in = take((size_t)nmax * p.Dq * sizeof(float));
    p.qk = take((size_t)nmax * p.H * p.Dkv * sizeof(float));
    p.z = take((size_t)nmax * p.H * p.Dkv * sizeof(float));
    p.q = take((size_t)nmax * p.Dq * sizeof(float));
    p.att = take((size_t)nmax * p.Dq * sizeof(float));
    p.fc = take((size_t)nmax * p.Dq * sizeof(float));
    p.merge_in = take((size_t)nmax * (p.Dq + p.Fn) * sizeof(float));
    p.hid = take((size_t)nmax * p.Fn * sizeof(float));
    p.cq = take((size_t)(p.L > 0 ? p.L : 1) * (p.Dq + p.Ft) * sizeof(float));      // per layer: the query's constant term [Dq] | cos(b) [Ft]
    if (p.L == 2) {
        p.dd_cap = 1024;
        while ((int64_t)p.dd_cap < 2 * p.n[1]) p.dd_cap <<= 1;
        p.dd_slots = take((size_t)p.dd_cap * sizeof(int32_t));
        p.dd_canon = take((size_t)p.n[1] * sizeof(int32_t));
        p.dd_cidx = take((size_t)p.n[1] * sizeof(int32_t));
        p.dd_map = take((size_t)p.n[1] * sizeof(int32_t));
        p.dd_count = take(sizeof(int32_t));
        p.dd_ids = take((size_t)p.n[1] * sizeof(int32_t));
        p.dd_times = take((size_t)p.n[1] * sizeof(double));
    }
    // packed weight fragments of the row-block chains (tgat_chain.h), with room for the GRU of a TGN call
    p.pack = take(chain::pack_bytes(chain::plan_pack(p.L, p.Fn, p.Ft, p.Dkv, p.H > 0 ? p.H : 1, 2 * p.Fn + p.Ft + p.Fe)));
    p.total = o;
    return p;
}

static int check_tgat(const dygnn_tgat_config* c) {
    DYGNN_REQUIRE(c != nullptr, "tgat: config is NULL");
    DYGNN_REQUIRE(c->node_feat_dim > 0 && c->edge_feat_dim > 0 && c->time_feat_dim > 0, "tgat: feature dims must be positive");
    DYGNN_REQUIRE(c->node_feat_dim % 4 == 0 && c->edge_feat_dim % 4 == 0 && c->time_feat_dim % 4 == 0, "tgat: feature dims must be multiples of 4");
    DYGNN_REQUIRE(c->num_layers >= 1 && c->num_layers <= 3, "tgat: num_layers must be in [1,3]");
    // models/modules.py:120
    DYGNN_REQUIRE(c->num_heads >= 1 && (c->node_feat_dim + c->time_feat_dim) % c->num_heads == 0,
                  "The sum of node_feat_dim and time_feat_dim should be divided by num_heads!");
    // utils/utils.py:157
    DYGNN_REQUIRE(c->num_neighbors > 0, "Number of sampled neighbors for each node should be greater than 0!");
    DYGNN_REQUIRE(c->num_neighbors <= 64, "tgat: num_neighbors > 64 not supported");
    DYGNN_REQUIRE(c->node_feat_dim + c->time_feat_dim <= 16 * 17, "tgat: node_feat_dim + time_feat_dim > 272 not supported");
    DYGNN_REQUIRE(((c->node_feat_dim + c->time_feat_dim) / c->num_heads) % 4 == 0, "tgat: head dim must be a multiple of 4");
    return DYGNN_OK;
}

template <bool RELU>
static int gemm_nt(const float* A, const float* W, const float* bias, float* C, int64_t M, int N, int K, int ldc, hipStream_t s, const int32_t* m_dev = nullptr) {
    if (M == 0) return DYGNN_OK;
    if (M >= 48 || m_dev) return train::mm(s, A, K, false, W, K, true, C, ldc, (int)M, N, K, bias, 1.f, 0.f, 1, 1, 0, 0, 0, 0, 0, 0, RELU, false, nullptr, m_dev);      // the LDS-tiled general GEMM
    hipLaunchKernelGGL((k_gemm_nt<RELU>), dim3((unsigned)ceil_div(M, 256), (unsigned)ceil_div(N, 64)), dim3(256), 0, s, A, W, bias, C, M, N, K, ldc);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

}  // namespace dygnn

using namespace dygnn;

extern "C" size_t dygnn_tgat_workspace_bytes(const dygnn_tgat_config* cfg, int64_t batch) {
    if (check_tgat(cfg) != DYGNN_OK || batch < 0) return 0;
    return make_tgat_plan(*cfg, batch).total;
}

namespace dygnn {
static bool tgat_dedup_active(const TgatPlan& p, bool presampled) {
    const char* dd_env = getenv("DYGNN_TGAT_DEDUP");                 // "0" switches it off (read per call: the A/B switch of tests/test_tgat.py)
    const bool dedup_on = !(dd_env && dd_env[0] == '0');
    return dedup_on && !presampled && p.L == 2 && p.k <= 20 && p.H <= 2 && p.Dkv <= 512;
}
static int tgat_forward_impl(const dygnn_tgat_config* cfg, const dygnn_tgat_weights* w, const dygnn_csr* csr, const float* node_feat,
                             const float* edge_feat, const int64_t* src, const int64_t* dst, const double* times, int64_t batch,
                             float* out_src, float* out_dst, void* workspace, size_t workspace_bytes, dygnn_stream_t stream,
                             const dygnn_tgat_levels* levels = nullptr, bool levels_in_workspace = false, bool expand_only = false,
                             const TgnTouch* touch = nullptr, bool packed = false, bool times_per_root = false, bool presampled = false) {
    if (int rc = check_tgat(cfg)) return rc;
    DYGNN_REQUIRE(w && w->time_w && w->time_b, "tgat: null weights");
    DYGNN_REQUIRE(levels || levels_in_workspace || (csr && csr->indptr && csr->num_nodes >= 1), "tgat: bad csr");
    DYGNN_REQUIRE(batch >= 0 && node_feat && edge_feat, "tgat: bad arguments");
    if (batch == 0) return DYGNN_OK;
    DYGNN_REQUIRE((levels || levels_in_workspace || (src && dst && times)) && out_src && out_dst && workspace, "tgat: null pointer");
    const TgatPlan p = make_tgat_plan(*cfg, batch);
    if (workspace_bytes < p.total) {
        set_error("tgat: workspace too small (%zu < %zu bytes)", workspace_bytes, p.total);
        return DYGNN_E_WORKSPACE;
    }
    for (int l = 0; l < p.L; ++l) {
        const dygnn_tgat_layer_weights& Lw = w->layers[l];
        DYGNN_REQUIRE(Lw.query_w && Lw.key_w && Lw.value_w && Lw.ln_w && Lw.ln_b && Lw.res_w && Lw.res_b && Lw.fc1_w && Lw.fc1_b && Lw.fc2_w && Lw.fc2_b,
                      "tgat: null layer weights (layer %d)", l);
    }
    hipStream_t s = as_stream(stream);
    char* ws = static_cast<char*>(workspace);
    auto I32 = [&](size_t off) { return reinterpret_cast<int32_t*>(ws + off); };
    auto F64 = [&](size_t off) { return reinterpret_cast<double*>(ws + off); };
    auto F32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    // Distinct entries of level 1 are computed once (see k_dedup_insert): only when the library samples itself (`recent` is a function of
    // (node, time); pre-sampled random levels draw independently per entry) and the pair attention kernel, which knows the row map, applies.
    const bool dedup = tgat_dedup_active(p, levels != nullptr || presampled);

    if (levels_in_workspace) {
        // the caller ran this function's own expansion on this workspace already (TGN: it needs the level-0 node set before the features exist)
    } else if (levels) {
        // pre-sampled levels (random strategies): copy them where the sampling kernels would have written them
        for (int l = 0; l <= p.L; ++l) {
            DYGNN_REQUIRE(levels->ids[l] && (l == 0 || (levels->nbr_eid[l] && levels->nbr_dt[l])), "tgat: null level array (level %d)", l);
            DYGNN_HIP(hipMemcpyAsync(I32(p.ids[l]), levels->ids[l], (size_t)p.n[l] * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
            if (l >= 1) {
                DYGNN_HIP(hipMemcpyAsync(I32(p.eid[l]), levels->nbr_eid[l], (size_t)p.n[l] * p.k * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
                DYGNN_HIP(hipMemcpyAsync(F32(p.dt[l]), levels->nbr_dt[l], (size_t)p.n[l] * p.k * sizeof(float), hipMemcpyDeviceToDevice, s));
            }
        }
    } else {
    // top-down: sample neighbours of every level, building the level below; level L = [src ; dst] is read from the caller's arrays
    for (int l = p.L; l >= 1; --l) {
        const bool dd = dedup && l == 1;          // level 1 is expanded from its distinct entries only
        const bool top = l == p.L;
        const bool tch = touch && l == 1;
        hipLaunchKernelGGL(k_tgat_expand, dim3((unsigned)ceil_div(p.n[l], 4)), dim3(256), 0, s, csr->indptr, csr->nbr, csr->eid, csr->ts, csr->num_nodes,
                           dd ? I32(p.dd_ids) : I32(p.ids[l]), dd ? F64(p.dd_times) : F64(p.times[l]), p.n[l], p.k, I32(p.ids[l - 1]), F64(p.times[l - 1]),
                           I32(p.eid[l]), F32(p.dt[l]), dd ? I32(p.dd_count) : (const int32_t*)nullptr, top ? src : nullptr, top ? dst : nullptr,
                           top ? times : nullptr, batch, tch ? *touch : TgnTouch{}, times_per_root ? 1 : 0);
        DYGNN_LAUNCH_CHECK();
        if (dedup && l == 2) {                    // level 1 is complete: find its distinct (node, time) entries
            const int64_t n1 = p.n[1];
            DYGNN_HIP(hipMemsetAsync(I32(p.dd_slots), 0xFF, (size_t)p.dd_cap * sizeof(int32_t), s));
            DYGNN_HIP(hipMemsetAsync(I32(p.dd_count), 0, sizeof(int32_t), s));
            hipLaunchKernelGGL(k_dedup_insert, dim3((unsigned)ceil_div(n1, 256)), dim3(256), 0, s, I32(p.ids[1]), F64(p.times[1]), n1, I32(p.dd_slots), p.dd_cap - 1,
                               I32(p.dd_canon));
            DYGNN_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_dedup_number, dim3((unsigned)ceil_div(n1, 1024)), dim3(1024), 0, s, I32(p.ids[1]), F64(p.times[1]), I32(p.dd_canon), n1, I32(p.dd_count),
                               I32(p.dd_cidx), I32(p.dd_ids), F64(p.dd_times));
            DYGNN_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_dedup_map, dim3((unsigned)ceil_div(n1, 256)), dim3(256), 0, s, I32(p.dd_canon), I32(p.dd_cidx), n1, I32(p.dd_map));
            DYGNN_LAUNCH_CHECK();
        }
    }
    }
    if (expand_only) return DYGNN_OK;
    // bottom-up: layer l turns level-(l-1) embeddings (raw features for l = 1) into level-l embeddings
    const float scale = (float)pow((double)p.hd, -0.5);
    // Two forms of a layer.  Row-block chains (tgat_chain.hip; three launches, intermediates in LDS): every workgroup streams the layer's
    // 2 MB of weights through its own CU, which pays when a level has few rows -- TGN calls (one step of a few hundred roots; they come
    // with their weights packed).  Product by product through the general GEMM: TGAT's levels of 10^4..10^5 rows.  The choice depends on
    // the caller, never on the batch, so a row keeps its bits at every batch size; DYGNN_TGAT_CHAIN=0/1 forces one (A/B switch, read per call).
    const char* ch_env = getenv("DYGNN_TGAT_CHAIN");
    // chain_from: the first layer that runs as chains.  TGN: all (1).  TGAT: none (L + 1); DYGNN_TGAT_CHAIN=1: all, =2: the top layer only (the
    // roots: a few hundred rows per evaluation step) -- by layer, never by batch size.
    const bool fits = chain::fits(p.Fn, p.Ft, p.Dkv, p.H);
    int chain_from = packed ? 1 : p.L + 1;
    if (ch_env && ch_env[0] >= '0' && ch_env[0] <= '2') chain_from = ch_env[0] == '0' ? p.L + 1 : ch_env[0] == '1' ? 1 : p.L;
    if (!fits) chain_from = p.L + 1;
    const chain::PackPlan pp = chain::plan_pack(p.L, p.Fn, p.Ft, p.Dkv, p.H, 2 * p.Fn + p.Ft + p.Fe);
    if (chain_from <= p.L && !packed)
        if (int rc = chain::pack(s, pp, p.L, p.Fn, p.Ft, p.Dkv, p.H, w, nullptr, 0, F32(p.pack))) return rc;
    for (int l = 1; l <= p.L; ++l) {
        const bool chain = l >= chain_from;
        const dygnn_tgat_layer_weights& Lw = w->layers[l - 1];
        const int64_t n = p.n[l];
        const float* h_lower = l >= 2 ? F32(p.h[l - 1]) : nullptr;
        const int32_t* nl = dedup && l == 1 ? I32(p.dd_count) : nullptr;       // layer 1 runs over the distinct level-1 entries (count on the device)
        const int32_t* lmap = dedup && l == 2 ? I32(p.dd_map) : nullptr;       // layer 2 finds an entry's layer-1 row through the map
        // the top level is [src rows ; dst rows]: when the caller's two outputs are one [2B, Fn] block it is written in place
        const bool direct = l == p.L && out_dst == out_src + (size_t)batch * p.Fn;
        float* h_out = direct ? out_src : F32(p.h[l]);
        float* cq = F32(p.cq) + (size_t)(l - 1) * (p.Dq + p.Ft);
        float* ct = cq + p.Dq;
        if (chain) {
            // q_in -> q -> W_k^T q inside one workgroup per 16 / 32 rows (tgat_chain.hip): intermediates stay in LDS
            chain::PreArgs pa{h_lower, node_feat, I32(p.ids[l - 1]), lmap, nl, w->time_w, w->time_b, F32(p.pack), pp.layer[l - 1].q, pp.layer[l - 1].k, F32(p.qk), n,
                              p.Fn, p.Ft, p.Dkv, p.H};
            if (int rc = chain::launch_pre(s, pa)) return rc;
        } else {
        // q = W_q[:, :Fn] h + cq (see k_tgat_const_q): the product gathers the feature rows itself
        hipLaunchKernelGGL(k_tgat_const_q, dim3((unsigned)ceil_div(p.Dq, 4)), dim3(256), 0, s, Lw.query_w, w->time_w, w->time_b, p.Dq, p.Fn, p.Ft, cq, ct);
        DYGNN_LAUNCH_CHECK();
        if (n >= 48) {
            if (int rc = train::mm(s, h_lower ? h_lower : node_feat, p.Fn, false, Lw.query_w, p.Dq, true, F32(p.q), p.Dq, (int)n, p.Dq, p.Fn, cq, 1.f, 0.f, 1, 1, 0, 0, 0, 0,
                                   0, 0, false, false, nullptr, nl, false, h_lower ? lmap : I32(p.ids[l - 1]))) return rc;
        } else {      // a handful of rows: the feature rows are staged ([n][Fn], Ft = 0: no time columns), then the small-M kernel
            hipLaunchKernelGGL(k_tgat_qrows, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, s, h_lower, node_feat, I32(p.ids[l - 1]), w->time_w, w->time_b, n, p.Fn, 0,
                               F32(p.q_in), nl, lmap);
            DYGNN_LAUNCH_CHECK();
            hipLaunchKernelGGL((k_gemm_nt<false>), dim3((unsigned)ceil_div(n, 256), (unsigned)ceil_div(p.Dq, 64)), dim3(256), 0, s, F32(p.q_in), Lw.query_w, cq, F32(p.q), n,
                               p.Dq, p.Fn, p.Dq, p.Dq);
            DYGNN_LAUNCH_CHECK();
        }
        // qk[i][h] = W_k,h^T q_ih : per head [n][hd] x [hd][Dkv] (rows h*hd .. of key_w), one batched launch over the heads
        if (int rc = train::mm(s, F32(p.q), p.Dq, false, Lw.key_w, p.Dkv, false, F32(p.qk), p.H * p.Dkv, (int)n, p.Dkv, p.hd, nullptr, 1.f, 0.f, p.H, p.H, 0, p.hd,
                               0, (int64_t)p.hd * p.Dkv, 0, p.Dkv, false, false, nullptr, nl)) return rc;
        }
        const dim3 grid((unsigned)ceil_div(n, 4));
        const size_t lds = (size_t)4 * p.H * p.k * sizeof(float);
        const bool attn_in_post = chain && chain::post_fuses_attention(n, p.Fn, p.Ft, p.Dkv, p.H, p.k);      // the chain's last kernel runs it on its own rows
        if (attn_in_post) {
        } else if (p.k <= 20 && p.H <= 2 && p.Dkv <= 512) {
            const int TW = 4 * ((p.Ft / 4 + 1) / 2);
            // KC = the row slots a lane keeps in registers: 10 for k <= 10 (TGN's configuration; half the gathers of the 20-slot form, whose
            // idle slots re-read row 0), 20 otherwise.  Idle slots contribute exact zeros, so a row's bits do not depend on KC.
            const int KC = p.k <= 10 ? 10 : 20;
            const size_t lds2 = ((size_t)8 * p.H * KC + (size_t)4 * KC * TW) * sizeof(float);
            // (k == KC: the instantiation without idle row slots; the bits of a row are the same)
#define DYGNN_ATTN_LAUNCH(KC_, FULL_)                                                                                                                           \
    hipLaunchKernelGGL((k_tgat_attn_pair<KC_, FULL_>), dim3((unsigned)ceil_div(n, 2)), dim3(256), lds2, s, F32(p.qk), h_lower, node_feat, edge_feat, I32(p.ids[l - 1]), \
                       I32(p.eid[l]), F32(p.dt[l]), w->time_w, w->time_b, n, p.k, p.Fn, p.Fe, p.Ft, p.H, scale, F32(p.z), nl, lmap)
            if (KC == 10) { if (p.k == 10) DYGNN_ATTN_LAUNCH(10, true); else DYGNN_ATTN_LAUNCH(10, false); }
            else { if (p.k == 20) DYGNN_ATTN_LAUNCH(20, true); else DYGNN_ATTN_LAUNCH(20, false); }
#undef DYGNN_ATTN_LAUNCH
        } else if (p.k <= 20)
            hipLaunchKernelGGL((k_tgat_attn_lin<20>), grid, dim3(256), lds + (size_t)4 * 20 * p.Ft * sizeof(float), s, F32(p.qk), h_lower, node_feat, edge_feat, I32(p.ids[l - 1]), I32(p.eid[l]), F32(p.dt[l]),
                               w->time_w, w->time_b, n, p.k, p.Fn, p.Fe, p.Ft, p.H, scale, F32(p.z));
        else
            hipLaunchKernelGGL((k_tgat_attn_lin<0>), grid, dim3(256), lds, s, F32(p.qk), h_lower, node_feat, edge_feat, I32(p.ids[l - 1]), I32(p.eid[l]), F32(p.dt[l]),
                               w->time_w, w->time_b, n, p.k, p.Fn, p.Fe, p.Ft, p.H, scale, F32(p.z));
        DYGNN_LAUNCH_CHECK();
        if (chain) {
            // W_v z -> residual_fc + q_in -> LayerNorm -> MergeLayer, one workgroup per 16 / 32 rows
            const chain::LayerPack& y = pp.layer[l - 1];
            chain::PostArgs po{F32(p.z), h_lower, node_feat, I32(p.ids[l - 1]), lmap, nl, w->time_w, w->time_b, F32(p.pack), y.v, y.r, y.f1, y.f2, Lw.res_b, Lw.ln_w,
                               Lw.ln_b, Lw.fc1_b, Lw.fc2_b, h_out, n, p.Fn, p.Ft, p.Dkv, p.H, nullptr,
                               attn_in_post ? F32(p.qk) : nullptr, edge_feat, I32(p.eid[l]), F32(p.dt[l]), p.k, p.Fe, scale};
#ifdef DYGNN_STAMPS      // diagnostic build only (tools/chain_stamps.py, DYGNN_LIB_VARIANT=stamps): the product library never takes a device address from the environment
            if (const char* st_env = getenv("DYGNN_CHAIN_STAMPS")) po.stamps = reinterpret_cast<unsigned long long*>(strtoull(st_env, nullptr, 0));
#endif
            if (int rc = chain::launch_post(s, po)) return rc;
            if (direct) return DYGNN_OK;
            continue;
        }
        // att[i][h*hd ..] = W_v,h z_ih : per head [n][Dkv] x [Dkv][hd]
        if (int rc = train::mm(s, F32(p.z), p.H * p.Dkv, false, Lw.value_w, p.Dkv, true, F32(p.att), p.Dq, (int)n, p.hd, p.Dkv, nullptr, 1.f, 0.f, p.H, p.H, 0, p.Dkv,
                               0, (int64_t)p.hd * p.Dkv, 0, p.hd, false, false, nullptr, nl)) return rc;
        if (int rc = gemm_nt<false>(F32(p.att), Lw.res_w, Lw.res_b, F32(p.fc), n, p.Dq, p.Dq, p.Dq, s, nl)) return rc;
        hipLaunchKernelGGL(k_tgat_post, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, s, F32(p.fc), h_lower, lmap, ct, Lw.ln_w, Lw.ln_b, node_feat, I32(p.ids[l - 1]),
                           n, p.Dq, p.Fn, F32(p.merge_in), nl);
        DYGNN_LAUNCH_CHECK();
        if (int rc = gemm_nt<true>(F32(p.merge_in), Lw.fc1_w, Lw.fc1_b, F32(p.hid), n, p.Fn, p.Dq + p.Fn, p.Fn, s, nl)) return rc;
        if (int rc = gemm_nt<false>(F32(p.hid), Lw.fc2_w, Lw.fc2_b, h_out, n, p.Fn, p.Fn, p.Fn, s, nl)) return rc;
        if (direct) return DYGNN_OK;
    }
    hipLaunchKernelGGL(k_split_out, dim3((unsigned)ceil_div(2 * batch * p.Fn, 256)), dim3(256), 0, s, F32(p.h[p.L]), batch, p.Fn, out_src, out_dst);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

// ================================================================================================
// TGN (BASELINE config 5; reference models/MemoryModel.py): memory bank + last-message aggregation + GRU update +
// the same temporal graph attention over (memory + raw) node features.
// State (owned by the caller, one per model): memory M [N][Fn], last-update time U [N] (float32), and ONE pending raw
// message per node (msg [N][2Fn+Ft+Fe], its float64 time, a flag): the reference keeps a Python list per node
// (MemoryModel.py:389-407) but its aggregator only ever reads the last element (:284-291), and lists are cleared
// whole (:400-407), so the last message is the entire observable state.
// ================================================================================================
// End of a positive call, one launch: for every batch node (a) persist the updated memory if it had a pending message and (b) store its
// new raw message (MemoryModel.py:142-161, :223-241, :425-459).  Messages are stored source role first, then destination role, and only a
// node's LAST stored message is ever read (:284-291): entry e = role * B + i (role 0 = source); the last entry of a node does the work
// for that node, the others exit.  "Had a pending message" is read from the call's pendf (written by the list pass), not from
// has_msg, which this kernel rewrites; the other endpoint's updated memory is Mnew when it was pending, M when not (nobody writes it then),
// so no workgroup reads what another one writes.
__global__ __launch_bounds__(256) void k_tgn_commit(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, const double* __restrict__ times,
                                                      const int64_t* __restrict__ eids, int64_t B, const float* __restrict__ Mnew, const int32_t* __restrict__ pendf,
                                                      float* __restrict__ M, float* __restrict__ U, const float* __restrict__ edge_feat,
                                                      const float* __restrict__ tw, const float* __restrict__ tb, int Fn, int Fe, int Ft, float* __restrict__ msg,
                                                      double* __restrict__ msg_t, int32_t* __restrict__ has_msg, int64_t N) {
    const int64_t e = blockIdx.x;
    const bool role = e >= B;
    const int64_t i = role ? e - B : e;
    const int64_t node = role ? dst[i] : src[i];
    const int64_t o = role ? src[i] : dst[i];
    if (node < 0 || node >= N || o < 0 || o >= N) return;      // never index the state tables out of range (the host API raises IndexError for such ids)
    int later = 0;
    for (int64_t e2 = e + 1 + threadIdx.x; e2 < 2 * B; e2 += blockDim.x) {
        const int64_t i2 = e2 >= B ? e2 - B : e2;
        const int64_t n2 = e2 >= B ? dst[i2] : src[i2], o2 = e2 >= B ? src[i2] : dst[i2];
        later |= (n2 == node && o2 >= 0 && o2 < N) ? 1 : 0;
    }
    if (__syncthreads_or(later)) return;
    const bool pend = pendf[node] != 0, pend_o = pendf[o] != 0;
    const float* mn = pend ? Mnew + node * Fn : M + node * Fn;      // the node's updated memory (MemoryModel.py:142-145 persists exactly this)
    const float* mo = pend_o ? Mnew + o * Fn : M + o * Fn;
    const float u = pend ? (float)msg_t[node] : U[node];            // last-update time after the persist (:447-459)
    const int D = 2 * Fn + Ft + Fe;
    const float dt = (float)times[i] - u;                           // float32 - float32 (MemoryModel.py:232-233)
    float* m = msg + node * D;
    for (int f = threadIdx.x; f < D; f += blockDim.x) {
        float v;
        if (f < Fn) v = mn[f];
        else if (f < 2 * Fn) v = mo[f - Fn];
        else if (f < 2 * Fn + Ft) v = cosf(fmaf(dt, tw[f - 2 * Fn], tb[f - 2 * Fn]));
        else v = edge_feat[(size_t)eids[i] * Fe + (f - 2 * Fn - Ft)];
        m[f] = v;
        if (pend && f < Fn) M[node * Fn + f] = v;                   // persist
    }
    __syncthreads();                                                // msg_t[node] was read above by every thread
    if (threadIdx.x == 0) {
        if (pend) U[node] = u;
        msg_t[node] = times[i];
        has_msg[node] = 1;
    }
}

// ---- the nodes a call reads (TGN) --------------------------------------------------------------------------------------------
// The reference updates the memory of every node with a pending message on every call (get_updated_memories over range(num_nodes),
// MemoryModel.py:108-109) although a call only reads the rows of its level-0 set (roots and sampled neighbours).  Here the GRU runs
// over exactly those: every level-0 slot names itself owner of its node (k_tgat_expand); the winning slot lists the node (the list blocks
// of chain::pack's launch): with a pending message for the GRU row-block kernel (tgat_chain.hip: gathers the listed rows, runs both gate
// products and scatters the new memory and feat0 = memory + raw), without one for the plain feat0 = memory + raw rows of the same launch.
// Rows of a product do not depend on which other rows are in it, so every row read later is bit-identical to the all-nodes update.
struct TgnPlan { size_t Mnew, feat0, tgat, owner, pendf, list, count, list2, count2, total; };
static TgnPlan make_tgn_plan(const dygnn_tgat_config& c, int64_t N, int64_t B) {
    TgnPlan p{};
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~size_t(255); return r; };
    p.Mnew = take((size_t)N * c.node_feat_dim * sizeof(float));
    p.feat0 = take((size_t)N * c.node_feat_dim * sizeof(float));
    p.owner = take((size_t)N * sizeof(int32_t));             // the level-0 slot that speaks for a node / whether the node had a pending message
    p.pendf = take((size_t)N * sizeof(int32_t));
    p.count = take(2 * sizeof(int32_t));                     // the two list lengths
    p.count2 = p.count + sizeof(int32_t);
    p.list = take((size_t)N * sizeof(int32_t));
    p.list2 = take((size_t)N * sizeof(int32_t));
    p.tgat = take(make_tgat_plan(c, B).total);
    p.total = o;
    return p;
}
}  // namespace dygnn

extern "C" int dygnn_tgat_forward_levels(const dygnn_tgat_config* cfg, const dygnn_tgat_weights* w, const dygnn_tgat_levels* levels, const float* node_feat,
                                         const float* edge_feat, int64_t batch, float* out_src, float* out_dst, void* workspace, size_t workspace_bytes,
                                         dygnn_stream_t stream) {
    DYGNN_REQUIRE(levels != nullptr, "tgat_forward_levels: levels is NULL");
    return tgat_forward_impl(cfg, w, nullptr, node_feat, edge_feat, nullptr, nullptr, nullptr, batch, out_src, out_dst, workspace, workspace_bytes, stream, levels);
}

extern "C" int dygnn_tgat_level_entries(const dygnn_tgat_config* cfg, int64_t batch, const void* workspace, int64_t* total, int64_t* computed,
                                        dygnn_stream_t stream) {
    if (int rc = check_tgat(cfg)) return rc;
    DYGNN_REQUIRE(batch > 0 && workspace && total && computed, "tgat_level_entries: bad arguments");
    const TgatPlan p = make_tgat_plan(*cfg, batch);
    *total = 0;
    for (int l = 1; l <= p.L; ++l) *total += p.n[l];
    *computed = *total;
    if (p.L == 2) {
        int32_t u = 0;
        DYGNN_HIP(hipMemcpyAsync(&u, static_cast<const char*>(workspace) + p.dd_count, sizeof(u), hipMemcpyDeviceToHost, as_stream(stream)));
        DYGNN_HIP(hipStreamSynchronize(as_stream(stream)));
        if (u > 0 && u <= p.n[1]) *computed = p.n[2] + u;
    }
    return DYGNN_OK;
}

extern "C" int dygnn_tgat_forward(const dygnn_tgat_config* cfg, const dygnn_tgat_weights* w, const dygnn_csr* csr, const float* node_feat,
                                  const float* edge_feat, const int64_t* src, const int64_t* dst, const double* times, int64_t batch,
                                  float* out_src, float* out_dst, void* workspace, size_t workspace_bytes, dygnn_stream_t stream) {
    return tgat_forward_impl(cfg, w, csr, node_feat, edge_feat, src, dst, times, batch, out_src, out_dst, workspace, workspace_bytes, stream);
}

// Embeddings of a LIST of (node, time) roots [n_roots] (n_roots even: the roots are the level [first half ; second half] of n_roots / 2
// "pairs", every root with its own time).  An evaluation step's positive and negative call share their source rows: the caller hands
// [sources ; destinations ; negative destinations] over once instead of [sources ; destinations] + [sources ; negatives].
extern "C" int dygnn_tgat_forward_roots(const dygnn_tgat_config* cfg, const dygnn_tgat_weights* w, const dygnn_csr* csr, const float* node_feat,
                                        const float* edge_feat, const int64_t* ids, const double* times, int64_t n_roots, float* out,
                                        void* workspace, size_t workspace_bytes, dygnn_stream_t stream) {
    DYGNN_REQUIRE(n_roots >= 0 && n_roots % 2 == 0, "tgat_forward_roots: n_roots must be even (pad with a repeated root)");
    DYGNN_REQUIRE(n_roots == 0 || (ids && times && out && cfg), "tgat_forward_roots: null pointer");
    const int64_t b = n_roots / 2;
    return tgat_forward_impl(cfg, w, csr, node_feat, edge_feat, ids, ids + b, times, b, out, out + (size_t)b * (cfg ? cfg->node_feat_dim : 0), workspace, workspace_bytes,
                             stream, nullptr, false, false, nullptr, false, true);
}

extern "C" size_t dygnn_tgn_workspace_bytes(const dygnn_tgat_config* cfg, int64_t num_nodes, int64_t batch) {
    if (check_tgat(cfg) != DYGNN_OK || batch < 0 || num_nodes < 1) return 0;
    return make_tgn_plan(*cfg, num_nodes, batch).total;
}

// The first n_pos pairs of the batch are positive edges (they update the state), the rest only read it.
static int tgn_forward_impl(const dygnn_tgat_config* cfg, const dygnn_tgat_weights* w, const dygnn_gru_weights* gru, const dygnn_csr* csr,
                            const float* node_feat, const float* edge_feat, const dygnn_tgn_state* st, const int64_t* src, const int64_t* dst,
                            const double* times, const int64_t* edge_ids, int64_t batch, int64_t n_pos, float* out_src,
                            float* out_dst, void* workspace, size_t workspace_bytes, dygnn_stream_t stream, const dygnn_tgat_levels* levels = nullptr) {
    const bool edges_are_positive = n_pos > 0;
    if (int rc = check_tgat(cfg)) return rc;
    DYGNN_REQUIRE(n_pos >= 0 && n_pos <= batch, "tgn: n_positive must be in [0, batch]");
    DYGNN_REQUIRE(gru && gru->weight_ih && gru->weight_hh && gru->bias_ih && gru->bias_hh, "tgn: null GRU weights");
    DYGNN_REQUIRE(st && st->memory && st->last_update && st->msg && st->msg_time && st->has_msg && st->num_nodes >= 1, "tgn: bad state");
    DYGNN_REQUIRE(batch >= 0 && node_feat && edge_feat && workspace, "tgn: bad arguments");
    DYGNN_REQUIRE(!edges_are_positive || edge_ids != nullptr, "tgn: edge_ids required for positive edges");   // MemoryModel.py:140
    if (batch == 0) return DYGNN_OK;
    const int64_t N = st->num_nodes;
    const int Fn = cfg->node_feat_dim, Fe = cfg->edge_feat_dim, Ft = cfg->time_feat_dim, Dm = 2 * Fn + Ft + Fe;
    const TgnPlan p = make_tgn_plan(*cfg, N, batch);
    if (workspace_bytes < p.total) { set_error("tgn: workspace too small (%zu < %zu bytes)", workspace_bytes, p.total); return DYGNN_E_WORKSPACE; }
    hipStream_t s = as_stream(stream);
    char* ws = static_cast<char*>(workspace);
    float* Mnew = reinterpret_cast<float*>(ws + p.Mnew);
    float* feat0 = reinterpret_cast<float*>(ws + p.feat0);
    int32_t* owner = reinterpret_cast<int32_t*>(ws + p.owner);
    int32_t* pendf = reinterpret_cast<int32_t*>(ws + p.pendf);
    int32_t* list = reinterpret_cast<int32_t*>(ws + p.list);
    int32_t* count = reinterpret_cast<int32_t*>(ws + p.count);
    int32_t* list2 = reinterpret_cast<int32_t*>(ws + p.list2);
    int32_t* count2 = reinterpret_cast<int32_t*>(ws + p.count2);
    // 0. the levels of this call (they depend on the graph only).  Their level-0 set is what the call reads: every slot of it names itself
    //    owner of its node (k_tgat_expand)
    char* wt = ws + p.tgat;
    const size_t wt_bytes = p.total - p.tgat;
    const TgnTouch touch{owner, count, N};
    const TgatPlan tp = make_tgat_plan(*cfg, batch);
    if (levels) {
        // pre-sampled levels (`uniform` / `time_interval_aware`: the caller replayed the sampler's RandomState, MemoryModel.py:626-629): copied to
        // where the expansion would have written them, then every level-0 slot names itself owner of its node
        if (int rc = tgat_forward_impl(cfg, w, nullptr, feat0, edge_feat, nullptr, nullptr, nullptr, batch, out_src, out_dst, wt, wt_bytes, stream, levels, false, true)) return rc;
        hipLaunchKernelGGL(k_tgn_touch_levels, dim3((unsigned)ceil_div(tp.n[0], 256)), dim3(256), 0, s, reinterpret_cast<const int32_t*>(wt + tp.ids[0]), tp.n[1], tp.k, touch);
        DYGNN_LAUNCH_CHECK();
    } else if (int rc = tgat_forward_impl(cfg, w, csr, feat0, edge_feat, src, dst, times, batch, out_src, out_dst, wt, wt_bytes, stream, nullptr, false, true, &touch)) return rc;
    // 1. one launch: the owners list their nodes -- pending message: GRU rows; none: feat0 = memory + raw (MemoryModel.py:609) -- and the weights
    //    of the GRU and of the layers are packed into operand fragments (tgat_chain.h); then the updated memories of the listed nodes (the
    //    reference updates all nodes, MemoryModel.py:108-109): one launch, row count on the device
    DYGNN_REQUIRE(chain::fits(tp.Fn, tp.Ft, tp.Dkv, tp.H), "tgn: feature dims do not fit the row-block kernels");
    const chain::PackPlan pp = chain::plan_pack(tp.L, tp.Fn, tp.Ft, tp.Dkv, tp.H, Dm);
    float* pk = reinterpret_cast<float*>(wt + tp.pack);
    const int32_t* live = tgat_dedup_active(tp, levels != nullptr) ? reinterpret_cast<const int32_t*>(wt + tp.dd_count) : nullptr;
    const chain::ListArgs la{reinterpret_cast<const int32_t*>(wt + tp.ids[0]), live, owner, st->has_msg, pendf, count, list, count2, list2, tp.n[1], N, tp.k};
    if (int rc = chain::pack(s, pp, tp.L, tp.Fn, tp.Ft, tp.Dkv, tp.H, w, gru, Dm, pk, &la)) return rc;
    const int64_t ub = N < tp.n[0] ? N : tp.n[0];              // the list cannot be longer than the level-0 set
    const chain::GruArgs ga{list, count, list2, count2, st->msg, st->memory, node_feat, pk, pp.ih, pp.hh, gru->bias_ih, gru->bias_hh, Mnew, feat0, ub, Dm, Fn};
    if (int rc = chain::launch_gru(s, ga)) return rc;
    // 2. temporal graph attention over (memory + raw) features (GraphAttentionEmbedding, MemoryModel.py:548-664) on the levels built above
    if (int rc = tgat_forward_impl(cfg, w, csr, feat0, edge_feat, src, dst, times, batch, out_src, out_dst, wt, wt_bytes, stream, nullptr, true, false, nullptr,
                                   true, false, levels != nullptr)) return rc;
    if (!edges_are_positive) return DYGNN_OK;
    // 3. persist the updated memories of the batch nodes and store their new raw messages (MemoryModel.py:142-161)
    hipLaunchKernelGGL(k_tgn_commit, dim3((unsigned)(2 * n_pos)), dim3(256), 0, s, src, dst, times, edge_ids, n_pos, Mnew, pendf, st->memory, st->last_update, edge_feat,
                       w->time_w, w->time_b, Fn, Fe, Ft, st->msg, st->msg_time, st->has_msg, N);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

extern "C" int dygnn_tgn_forward(const dygnn_tgat_config* cfg, const dygnn_tgat_weights* w, const dygnn_gru_weights* gru, const dygnn_csr* csr,
                                 const float* node_feat, const float* edge_feat, const dygnn_tgn_state* st, const int64_t* src, const int64_t* dst,
                                 const double* times, const int64_t* edge_ids, int64_t batch, int32_t edges_are_positive, float* out_src,
                                 float* out_dst, void* workspace, size_t workspace_bytes, dygnn_stream_t stream) {
    return tgn_forward_impl(cfg, w, gru, csr, node_feat, edge_feat, st, src, dst, times, edge_ids, batch, edges_are_positive ? batch : 0, out_src, out_dst,
                            workspace, workspace_bytes, stream);
}

// dygnn_tgn_forward_step on PRE-SAMPLED levels: the reference accepts any sampling strategy for TGN (MemoryModel.py:626-629); `uniform` and
// `time_interval_aware` draw from the sampler's numpy RandomState, which the host mirror replays in the reference's order (one call on
// [src ; dst], MemoryModel.py:104-131) and hands over as level arrays in dygnn_tgat_forward_levels' layout.  src / dst / times are still read:
// by the commit of a positive call (messages, last-update times).
extern "C" int dygnn_tgn_forward_levels(const dygnn_tgat_config* cfg, const dygnn_tgat_weights* w, const dygnn_gru_weights* gru,
                                        const dygnn_tgat_levels* levels, const float* node_feat, const float* edge_feat, const dygnn_tgn_state* st,
                                        const int64_t* src, const int64_t* dst, const double* times, const int64_t* edge_ids, int64_t batch,
                                        int64_t n_positive, float* out_src, float* out_dst, void* workspace, size_t workspace_bytes, dygnn_stream_t stream) {
    DYGNN_REQUIRE(levels != nullptr, "tgn_forward_levels: levels is NULL");
    DYGNN_REQUIRE(n_positive == 0 || (src && dst && times), "tgn_forward_levels: a positive call needs src / dst / times");
    return tgn_forward_impl(cfg, w, gru, nullptr, node_feat, edge_feat, st, src, dst, times, edge_ids, batch, n_positive, out_src, out_dst, workspace,
                            workspace_bytes, stream, levels);
}

extern "C" int dygnn_tgn_forward_step(const dygnn_tgat_config* cfg, const dygnn_tgat_weights* w, const dygnn_gru_weights* gru, const dygnn_csr* csr,
                                      const float* node_feat, const float* edge_feat, const dygnn_tgn_state* st, const int64_t* src, const int64_t* dst,
                                      const double* times, const int64_t* edge_ids, int64_t batch, int64_t n_positive, float* out_src,
                                      float* out_dst, void* workspace, size_t workspace_bytes, dygnn_stream_t stream) {
    return tgn_forward_impl(cfg, w, gru, csr, node_feat, edge_feat, st, src, dst, times, edge_ids, batch, n_positive, out_src, out_dst, workspace,
                            workspace_bytes, stream);
}
