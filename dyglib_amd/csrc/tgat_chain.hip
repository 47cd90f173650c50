// Row-block chains of the TGAT / TGN layer (see tgat_chain.h).  gfx950 only.
//
// One workgroup (8 wave64) owns R = 16*MT rows of a level.  Every product of the chain is Out[R][N] = Act[R][K] . W^T with the
// activations in LDS and the weight operand read from global memory (L2-resident: a layer's weights are 2 MB) straight into the
// MFMA A operand, `v_mfma_f32_16x16x4_f32`, transposed form: accumulator tile = Out^T[n = 4g+r][m = c] for lane (c = lane&15,
// g = lane>>4), so a lane ends with four consecutive n of row c = one float4 store into the next product's LDS operand.
//  * weights [N][K] (K contiguous; nn.Linear): lane (c,g) reads the float4 W[n0+c][k0+4g..] and Act[c][k0+4g..]: 4 MFMAs per pair of
//    loads, 16 k per step (tile_kc);
//  * weights [K][N] (contraction over ROWS: W_k,h^T q): lane (c,g) reads the float4 W[k0+g][nb+4c..] and the scalar Act[c][k0+g]; its
//    four elements feed four accumulators whose tiles interleave to 64 consecutive n (block_km).
// LDS row strides are 4 (mod 8) floats: the 16 rows of a b128 read (and the 16 x 4 words of the b32 read) fall on distinct banks.
// A row's result depends on that row's data only (an MFMA column never mixes with another), in the same order for either MT, so rows
// are bit-identical whatever batch or block they sit in.
#include "tgat_chain.h"

namespace dygnn {
namespace chain {

using f4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f4 cmfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
__host__ __device__ inline int pad_ld(int K) {           // smallest row stride >= K with stride % 8 == 4
    const int l = (K + 3) & ~3;
    return (l & 4) ? l : l + 4;
}

// acc[j] (+)= tile n0..n0+15 of Act[16j..16j+15][0..K) . W[n][0..K)^T, weights K-contiguous.  Weight float4s run U chunks ahead.
template <int MT>
__device__ __forceinline__ void tile_kc(const float* __restrict__ W, int ldw, int N, int K, int n0, const float* act, int lda, int lane, f4 (&acc)[MT]) {
    constexpr int U = 4;
    const int c = lane & 15, g = lane >> 4;
    const bool vn = n0 + c < N;
    const float* wp = W + (size_t)(vn ? n0 + c : 0) * ldw + 4 * g;
    const float* ap = act + c * lda + 4 * g;
    const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[j] = zero;
    const int nch = (K + 15) >> 4;
    f4 wc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool vk = 16 * u + 4 * g < K;
        wc[u] = *reinterpret_cast<const f4*>(wp + (vk ? 16 * u : 0));
    }
    for (int ch0 = 0; ch0 < nch; ch0 += U) {
        f4 wn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ch = ch0 + U + u;
            const bool vk = 16 * ch + 4 * g < K;
            wn[u] = *reinterpret_cast<const f4*>(wp + (vk ? 16 * ch : 0));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ch = ch0 + u;
            if (ch < nch) {
                const bool vk = 16 * ch + 4 * g < K;
                const f4 w = (vk && vn) ? wc[u] : zero;
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    f4 b = *reinterpret_cast<const f4*>(ap + j * 16 * lda + (vk ? 16 * ch : 0));
                    b = vk ? b : zero;
                    acc[j] = cmfma(w.x, b.x, acc[j]);
                    acc[j] = cmfma(w.y, b.y, acc[j]);
                    acc[j] = cmfma(w.z, b.z, acc[j]);
                    acc[j] = cmfma(w.w, b.w, acc[j]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) wc[u] = wn[u];
    }
}

// acc[t][j][r] = Out[16j + c][nb + 16g + 4r + t], Out = Act[.][0..K) . W[0..K)[n], weight rows are the contraction index (N % 4 == 0)
template <int MT>
__device__ __forceinline__ void block_km(const float* __restrict__ W, int ldw, int N, int K, int nb, const float* act, int lda, int lane, f4 (&acc)[4][MT]) {
    constexpr int U = 4;
    const int c = lane & 15, g = lane >> 4;
    const bool vn = nb + 4 * c < N;
    const float* wp = W + (vn ? nb + 4 * c : 0) + (size_t)g * ldw;
    const float* ap = act + c * lda + g;
    const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[t][j] = zero;
    const int nst = (K + 3) >> 2;
    f4 wc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool vk = 4 * u + g < K;
        wc[u] = *reinterpret_cast<const f4*>(wp + (vk ? (size_t)4 * u * ldw : 0));
    }
    for (int s0 = 0; s0 < nst; s0 += U) {
        f4 wn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int st = s0 + U + u;
            const bool vk = 4 * st + g < K;
            wn[u] = *reinterpret_cast<const f4*>(wp + (vk ? (size_t)4 * st * ldw : 0));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int st = s0 + u;
            if (st < nst) {
                const bool vk = 4 * st + g < K;
                const f4 w = (vk && vn) ? wc[u] : zero;
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    float b = ap[j * 16 * lda + (vk ? 4 * st : 0)];
                    b = vk ? b : 0.f;
                    acc[0][j] = cmfma(w.x, b, acc[0][j]);
                    acc[1][j] = cmfma(w.y, b, acc[1][j]);
                    acc[2][j] = cmfma(w.z, b, acc[2][j]);
                    acc[3][j] = cmfma(w.w, b, acc[3][j]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) wc[u] = wn[u];
    }
}

// query-input rows [h(self) | cos(w*0 + b)] of the block's rows into LDS (zero rows beyond the live count; zero padding columns)
template <int MT>
__device__ __forceinline__ void fill_qin(float* qin, int ldq, const float* tcos, const float* __restrict__ h_lower, const float* __restrict__ node_feat,
                                         const int32_t* __restrict__ lower_ids, const int32_t* __restrict__ lower_map, int64_t i0, int64_t nl, int Fn,
                                         int Dq, int wave, int lane) {
    for (int rr = wave; rr < 16 * MT; rr += kWaves) {
        const int64_t i = i0 + rr;
        const bool valid = i < nl;
        const float* hsrc = nullptr;
        if (valid) hsrc = h_lower ? h_lower + (lower_map ? (int64_t)lower_map[i] : i) * Fn : node_feat + (size_t)lower_ids[i] * Fn;
        for (int f = lane; f < ldq; f += 64) qin[rr * ldq + f] = !valid ? 0.f : f < Fn ? hsrc[f] : f < Dq ? tcos[f - Fn] : 0.f;
    }
}

template <int MT>
__global__ __launch_bounds__(kThreads) void k_tgat_pre(const PreArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    constexpr int R = 16 * MT;
    const int64_t i0 = (int64_t)blockIdx.x * R;
    const int64_t nl = a.n_live ? (int64_t)*a.n_live : a.n;
    if (i0 >= nl) return;
    const int Fn = a.Fn, Ft = a.Ft, Dq = Fn + Ft, Dkv = a.Dkv, H = a.H, hd = Dq / H;
    const int ldq = pad_ld(Dq);
    float* qin = lds;
    float* q = qin + R * ldq;
    float* tcos = q + R * ldq;
    for (int f = threadIdx.x; f < Ft; f += kThreads) tcos[f] = cosf(fmaf(0.0f, a.tw[f], a.tb[f]));      // the query's time feature: dt = 0 (models/TGAT.py:84)
    __syncthreads();
    fill_qin<MT>(qin, ldq, tcos, a.h_lower, a.node_feat, a.lower_ids, a.lower_map, i0, nl, Fn, Dq, wave, lane);
    __syncthreads();
    // q = W_q q_in (bias-free, models/modules.py:126)
    const int ntq = (Dq + 15) >> 4;
    for (int nt = wave; nt < ntq; nt += kWaves) {
        f4 acc[MT];
        tile_kc<MT>(a.query_w, Dq, Dq, Dq, nt * 16, qin, ldq, lane, acc);
        const int n = nt * 16 + 4 * g;
        if (n < Dq) {
#pragma unroll
            for (int j = 0; j < MT; ++j) *reinterpret_cast<f4*>(q + (16 * j + c) * ldq + n) = acc[j];
        }
    }
    __syncthreads();
    // qk[i][h][:] = W_k,h^T q_ih: rows h*hd .. of key_w [Dq][Dkv] are the contraction index
    const int nblk = (Dkv + 63) >> 6, jobs = H * nblk;
    for (int job = wave; job < jobs; job += kWaves) {
        const int h = job / nblk, nb = (job - h * nblk) * 64;
        f4 acc[4][MT];
        block_km<MT>(a.key_w + (size_t)h * hd * Dkv, Dkv, Dkv, hd, nb, q + h * hd, ldq, lane, acc);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int64_t i = i0 + 16 * j + c;
            if (i >= nl) continue;
            float* o = a.qk + ((size_t)i * H + h) * Dkv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nb + 16 * g + 4 * r;
                if (n < Dkv) *reinterpret_cast<f4*>(o + n) = f4{acc[0][j][r], acc[1][j][r], acc[2][j][r], acc[3][j][r]};
            }
        }
    }
}

template <int MT>
__global__ __launch_bounds__(kThreads) void k_tgat_post(const PostArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    constexpr int R = 16 * MT;
    const int64_t i0 = (int64_t)blockIdx.x * R;
    const int64_t nl = a.n_live ? (int64_t)*a.n_live : a.n;
    if (i0 >= nl) return;
    const int Fn = a.Fn, Ft = a.Ft, Dq = Fn + Ft, Dkv = a.Dkv, H = a.H, hd = Dq / H, Dm = Dq + Fn;
    const int ldz = pad_ld(H * Dkv), ldq = pad_ld(Dq), ldm = pad_ld(Dm), ldh = pad_ld(Fn);
    // region 0: z rows, later (z is dead after the W_v product) q_in rows | MergeLayer input rows ; region 1: att rows, later hid rows
    float* zb = lds;
    float* qin = lds;
    float* mrg = lds + R * ldq;
    const int r0f = R * ldz > R * (ldq + ldm) ? R * ldz : R * (ldq + ldm);
    float* att = lds + r0f;
    float* hid = att;
    float* tcos = att + R * ldq;
    for (int f = threadIdx.x; f < Ft; f += kThreads) tcos[f] = cosf(fmaf(0.0f, a.tw[f], a.tb[f]));
    {   // z rows of the block (float4, coalesced); rows beyond the live count and the padding columns are zero
        const int z4 = (H * Dkv) >> 2, l4 = ldz >> 2;
        for (int rr = wave; rr < R; rr += kWaves) {
            const int64_t i = i0 + rr;
            const f4* src = reinterpret_cast<const f4*>(a.z + (size_t)(i < nl ? i : 0) * H * Dkv);
            for (int x = lane; x < l4; x += 64)
                *reinterpret_cast<f4*>(zb + rr * ldz + 4 * x) = (i < nl && x < z4) ? src[x] : f4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
    // att[i][h*hd + e] = W_v,h z_ih (value_w [Dq][Dkv], bias-free)
    const int nth = (hd + 15) >> 4;
    for (int job = wave; job < H * nth; job += kWaves) {
        const int h = job / nth, n0 = (job - h * nth) * 16;
        f4 acc[MT];
        tile_kc<MT>(a.value_w + (size_t)h * hd * Dkv, Dkv, hd, Dkv, n0, zb + h * Dkv, ldz, lane, acc);
        const int n = n0 + 4 * g;
        if (n < hd) {
#pragma unroll
            for (int j = 0; j < MT; ++j) *reinterpret_cast<f4*>(att + (16 * j + c) * ldq + h * hd + n) = acc[j];
        }
    }
    __syncthreads();
    fill_qin<MT>(qin, ldq, tcos, a.h_lower, a.node_feat, a.lower_ids, a.lower_map, i0, nl, Fn, Dq, wave, lane);      // the residual (models/modules.py:150, :196)
    __syncthreads();
    // x = residual_fc(att) + q_in, into the MergeLayer input rows (normalised in place below)
    const int ntq = (Dq + 15) >> 4;
    for (int nt = wave; nt < ntq; nt += kWaves) {
        f4 acc[MT];
        tile_kc<MT>(a.res_w, Dq, Dq, Dq, nt * 16, att, ldq, lane, acc);
        const int n = nt * 16 + 4 * g;
        if (n < Dq) {
            const f4 b = *reinterpret_cast<const f4*>(a.res_b + n);
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const f4 r = *reinterpret_cast<const f4*>(qin + (16 * j + c) * ldq + n);
                *reinterpret_cast<f4*>(mrg + (16 * j + c) * ldm + n) = f4{acc[j].x + b.x + r.x, acc[j].y + b.y + r.y, acc[j].z + b.z + r.z, acc[j].w + b.w + r.w};
            }
        }
    }
    __syncthreads();
    // LayerNorm (eps 1e-5) per row; the raw node features fill the rest of the MergeLayer input (models/TGAT.py:134, models/modules.py:64)
    for (int rr = wave; rr < R; rr += kWaves) {
        const int64_t i = i0 + rr;
        float* row = mrg + rr * ldm;
        float s = 0.f;
        for (int f = lane; f < Dq; f += 64) s += row[f];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s / (float)Dq;
        float v = 0.f;
        for (int f = lane; f < Dq; f += 64) { const float d = row[f] - mean; v = fmaf(d, d, v); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        const float rstd = 1.0f / sqrtf(v / (float)Dq + 1e-5f);
        for (int f = lane; f < Dq; f += 64) row[f] = (row[f] - mean) * rstd * a.ln_w[f] + a.ln_b[f];
        const float* raw = a.node_feat + (size_t)(i < nl ? a.lower_ids[i] : 0) * Fn;
        for (int f = lane; f < ldm - Dq; f += 64) row[Dq + f] = (i < nl && f < Fn) ? raw[f] : 0.f;
    }
    __syncthreads();
    // hid = relu(fc1 [y | raw] + b1)
    const int ntf = (Fn + 15) >> 4;
    for (int nt = wave; nt < ntf; nt += kWaves) {
        f4 acc[MT];
        tile_kc<MT>(a.fc1_w, Dm, Fn, Dm, nt * 16, mrg, ldm, lane, acc);
        const int n = nt * 16 + 4 * g;
        if (n < Fn) {
            const f4 b = *reinterpret_cast<const f4*>(a.fc1_b + n);
#pragma unroll
            for (int j = 0; j < MT; ++j)
                *reinterpret_cast<f4*>(hid + (16 * j + c) * ldh + n) = f4{fmaxf(acc[j].x + b.x, 0.f), fmaxf(acc[j].y + b.y, 0.f), fmaxf(acc[j].z + b.z, 0.f), fmaxf(acc[j].w + b.w, 0.f)};
        }
    }
    __syncthreads();
    // out = fc2 hid + b2
    for (int nt = wave; nt < ntf; nt += kWaves) {
        f4 acc[MT];
        tile_kc<MT>(a.fc2_w, Fn, Fn, Fn, nt * 16, hid, ldh, lane, acc);
        const int n = nt * 16 + 4 * g;
        if (n < Fn) {
            const f4 b = *reinterpret_cast<const f4*>(a.fc2_b + n);
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int64_t i = i0 + 16 * j + c;
                if (i < nl) *reinterpret_cast<f4*>(a.out + (size_t)i * Fn + n) = f4{acc[j].x + b.x, acc[j].y + b.y, acc[j].z + b.z, acc[j].w + b.w};
            }
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_tgn_gru_chain(const GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * 16;
    const int64_t cnt = *a.count;
    if (r0 >= cnt) return;
    const int Dm = a.Dm, Fn = a.Fn, G = 3 * Fn;
    const int ldm = pad_ld(Dm), ldh = pad_ld(Fn), ldg = pad_ld(G);
    float* am = lds;                  // [16][ldm] aggregated (= last) message rows
    float* ah = am + 16 * ldm;        // [16][ldh] memory rows
    float* gi = ah + 16 * ldh;        // [16][ldg] W_ih m + b_ih
    float* gh = gi + 16 * ldg;        // [16][ldg] W_hh h + b_hh
    for (int rr = wave; rr < 16; rr += kWaves) {
        const bool valid = r0 + rr < cnt;
        const int64_t node = valid ? a.list[r0 + rr] : 0;
        for (int f = lane; f < ldm; f += 64) am[rr * ldm + f] = (valid && f < Dm) ? a.msg[node * Dm + f] : 0.f;
        for (int f = lane; f < ldh; f += 64) ah[rr * ldh + f] = (valid && f < Fn) ? a.M[node * Fn + f] : 0.f;
    }
    __syncthreads();
    const int nt = (G + 15) >> 4;
    for (int job = wave; job < 2 * nt; job += kWaves) {
        const bool hh = job >= nt;
        const int n0 = (hh ? job - nt : job) * 16;
        f4 acc[1];
        if (hh) tile_kc<1>(a.w_hh, Fn, G, Fn, n0, ah, ldh, lane, acc);
        else tile_kc<1>(a.w_ih, Dm, G, Dm, n0, am, ldm, lane, acc);
        const int n = n0 + 4 * g;
        if (n < G) {
            const f4 b = *reinterpret_cast<const f4*>((hh ? a.b_hh : a.b_ih) + n);
            *reinterpret_cast<f4*>((hh ? gh : gi) + c * ldg + n) = f4{acc[0].x + b.x, acc[0].y + b.y, acc[0].z + b.z, acc[0].w + b.w};
        }
    }
    __syncthreads();
    // nn.GRUCell gates (r, z, n order) -> new memory; feat0 = new memory + raw features (MemoryModel.py:609)
    for (int idx = threadIdx.x; idx < 16 * Fn; idx += kThreads) {
        const int rr = idx / Fn, f = idx - rr * Fn;
        if (r0 + rr >= cnt) continue;
        const int64_t node = a.list[r0 + rr];
        const float* x = gi + rr * ldg;
        const float* y = gh + rr * ldg;
        const float h = ah[rr * ldh + f];
        const float rg = 1.0f / (1.0f + expf(-(x[f] + y[f])));
        const float zg = 1.0f / (1.0f + expf(-(x[Fn + f] + y[Fn + f])));
        const float ng = tanhf(x[2 * Fn + f] + rg * y[2 * Fn + f]);
        const float hn = (1.0f - zg) * ng + zg * h;
        a.Mnew[node * Fn + f] = hn;
        a.feat0[node * Fn + f] = hn + a.raw[node * Fn + f];
    }
}

static size_t pre_lds(const PreArgs& a, int MT) { return ((size_t)2 * 16 * MT * pad_ld(a.Fn + a.Ft) + a.Ft) * sizeof(float); }
static size_t post_lds(const PostArgs& a, int MT) {
    const int R = 16 * MT, Dq = a.Fn + a.Ft;
    const size_t r0 = (size_t)R * pad_ld(a.H * a.Dkv), r0b = (size_t)R * (pad_ld(Dq) + pad_ld(Dq + a.Fn));
    return ((r0 > r0b ? r0 : r0b) + (size_t)R * pad_ld(Dq) + a.Ft) * sizeof(float);
}
static size_t gru_lds(const GruArgs& a) { return (size_t)16 * (pad_ld(a.Dm) + pad_ld(a.Fn) + 2 * pad_ld(3 * a.Fn)) * sizeof(float); }
constexpr size_t kLdsMax = 160 * 1024;

template <class K>
static int set_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return DYGNN_OK;
}
static int pick_mt(int64_t n, size_t lds2) { return (n > 4096 && lds2 <= kLdsMax) ? 2 : 1; }

bool fits(int Fn, int Ft, int Dkv, int H) {
    if (H < 1 || (Fn + Ft) % H || Fn % 4 || Ft % 4 || Dkv % 4 || ((Fn + Ft) / H) % 4) return false;
    PreArgs pa{}; pa.Fn = Fn; pa.Ft = Ft; pa.Dkv = Dkv; pa.H = H;
    PostArgs po{}; po.Fn = Fn; po.Ft = Ft; po.Dkv = Dkv; po.H = H;
    return pre_lds(pa, 1) <= kLdsMax && post_lds(po, 1) <= kLdsMax;
}

int launch_pre(hipStream_t s, const PreArgs& a) {
    if (a.n == 0) return DYGNN_OK;
    DYGNN_REQUIRE(a.H >= 1 && (a.Fn + a.Ft) % a.H == 0 && a.Dkv % 4 == 0 && a.Fn % 4 == 0 && a.Ft % 4 == 0 && ((a.Fn + a.Ft) / a.H) % 4 == 0,
                  "tgat chain: dims must be multiples of 4");
    const int MT = pick_mt(a.n, pre_lds(a, 2));
    const size_t bytes = pre_lds(a, MT);
    DYGNN_REQUIRE(bytes <= kLdsMax, "tgat chain: feature dims too large for the row-block kernels (%zu bytes of LDS)", bytes);
    const dim3 grid((unsigned)ceil_div(a.n, 16 * MT));
    if (MT == 2) { if (int rc = set_lds(k_tgat_pre<2>, bytes)) return rc; hipLaunchKernelGGL(k_tgat_pre<2>, grid, dim3(kThreads), bytes, s, a); }
    else { if (int rc = set_lds(k_tgat_pre<1>, bytes)) return rc; hipLaunchKernelGGL(k_tgat_pre<1>, grid, dim3(kThreads), bytes, s, a); }
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

int launch_post(hipStream_t s, const PostArgs& a) {
    if (a.n == 0) return DYGNN_OK;
    const int MT = pick_mt(a.n, post_lds(a, 2));
    const size_t bytes = post_lds(a, MT);
    DYGNN_REQUIRE(bytes <= kLdsMax, "tgat chain: feature dims too large for the row-block kernels (%zu bytes of LDS)", bytes);
    const dim3 grid((unsigned)ceil_div(a.n, 16 * MT));
    if (MT == 2) { if (int rc = set_lds(k_tgat_post<2>, bytes)) return rc; hipLaunchKernelGGL(k_tgat_post<2>, grid, dim3(kThreads), bytes, s, a); }
    else { if (int rc = set_lds(k_tgat_post<1>, bytes)) return rc; hipLaunchKernelGGL(k_tgat_post<1>, grid, dim3(kThreads), bytes, s, a); }
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

int launch_gru(hipStream_t s, const GruArgs& a) {
    if (a.max_rows == 0) return DYGNN_OK;
    const size_t bytes = gru_lds(a);
    DYGNN_REQUIRE(bytes <= kLdsMax && a.Fn % 4 == 0 && a.Dm % 4 == 0, "tgn chain: feature dims too large for the GRU row-block kernel (%zu bytes of LDS)", bytes);
    if (int rc = set_lds(k_tgn_gru_chain, bytes)) return rc;
    hipLaunchKernelGGL(k_tgn_gru_chain, dim3((unsigned)ceil_div(a.max_rows, 16)), dim3(kThreads), bytes, s, a);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

}  // namespace chain
}  // namespace dygnn
