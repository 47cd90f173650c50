// Fused DyGFormer forward for gfx950: ONE workgroup (8 wave64) per (src,dst,t) pair runs
// models/DyGFormer.py:68-194 end to end — window gather, co-occurrence counts, feature gathers,
// time encoding, patch projection, every pre-LN encoder layer, mean-pool and output layer — with all
// dense math on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32) and no activation ever written
// to HBM.
//
// Formulation (everything is computed TRANSPOSED: Out^T[n][tok] = sum_k W[n][k] * Act[tok][k]):
//   * MFMA A operand = weights, streamed from L2 in a pre-packed fragment order (one coalesced
//     1-KiB read per 16x16 fragment, see pack kernels below); B operand = activations.
//   * An accumulator tile holds Out^T[4g+r][c] in lane (c = lane&15, g = lane>>4), register r:
//     exactly the B-operand layout of the NEXT product that sums over its row index.  So Q^T feeds
//     K.Q^T, softmax(P)^T feeds V^T.P^T, O^T feeds the out-projection and gelu(H)^T feeds the second
//     FFN GEMM straight from registers: Q, P, O and the 800-wide FFN hidden never touch LDS.
//   * wave (tt, hf) = (token tile of 16 tokens, half): owns the residual stream X^T for its tokens
//     (n-tiles 0..6 / 7..12) in registers for the whole kernel, head `hf` in attention, hidden half
//     `hf` in the FFN (K-split, partial sums exchanged through LDS).
//   * LDS (160 KiB): Xn (LayerNorm output, [tok][204]), K, V ([tok][204]) + 7 KiB of small state.
// Shape limits (else the dispatcher uses the generic path): D = 200 (C = 50), 2 heads, <= 64 tokens
// per pair, feature dims multiples of 4.
#include "dygformer_layout.h"

namespace dygnn {

using f4 = __attribute__((ext_vector_type(4))) float;

constexpr int kD = 200;        // model dim
constexpr int kDP = 208;       // padded to 13 n-tiles
constexpr int kNT = 13;        // n-tiles of 16 rows
constexpr int kKC = 13;        // k-chunks of 16 for K = 200
constexpr int kHD = 100;       // head dim
constexpr int kXS = 204;       // LDS row stride (floats): 16-B aligned rows, conflict-free b128 column reads
constexpr int kTok = 64;       // max tokens per pair
constexpr int kHid = 800;
constexpr int kHT = 50;        // hidden tiles
constexpr int kBufFloats = kTok * kXS;                 // 13056 floats = 52224 B
constexpr int kLdsXn = 0, kLdsK = kBufFloats, kLdsV = 2 * kBufFloats, kLdsMisc = 3 * kBufFloats;
constexpr int kLdsBytes = 160 * 1024;
constexpr int kFrag = 256;     // floats per packed 16x16 fragment

__device__ __forceinline__ f4 mfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// one 16-wide k-chunk: acc += A(16 x 16k) . B(16k x 16); a,b hold this lane's 4 consecutive k values
__device__ __forceinline__ void mma_chunk(f4& acc, const f4 a, const f4 b) {
    acc = mfma(a.x, b.x, acc);
    acc = mfma(a.y, b.y, acc);
    acc = mfma(a.z, b.z, acc);
    acc = mfma(a.w, b.w, acc);
}
// two independent accumulators interleaved (dependent-accumulate latency 40 cyc > issue 32 cyc)
__device__ __forceinline__ void mma_chunk2(f4& acc0, f4& acc1, const f4 a0, const f4 a1, const f4 b) {
    acc0 = mfma(a0.x, b.x, acc0); acc1 = mfma(a1.x, b.x, acc1);
    acc0 = mfma(a0.y, b.y, acc0); acc1 = mfma(a1.y, b.y, acc1);
    acc0 = mfma(a0.z, b.z, acc0); acc1 = mfma(a1.z, b.z, acc1);
    acc0 = mfma(a0.w, b.w, acc0); acc1 = mfma(a1.w, b.w, acc1);
}

// N independent accumulators share one B chunk: issue t-major so consecutive MFMAs never depend on each other
template <int N>
__device__ __forceinline__ void mma_group(f4* acc, const f4* a, const f4 b) {
#pragma unroll
    for (int u = 0; u < N; ++u) acc[u] = mfma(a[u].x, b.x, acc[u]);
#pragma unroll
    for (int u = 0; u < N; ++u) acc[u] = mfma(a[u].y, b.y, acc[u]);
#pragma unroll
    for (int u = 0; u < N; ++u) acc[u] = mfma(a[u].z, b.z, acc[u]);
#pragma unroll
    for (int u = 0; u < N; ++u) acc[u] = mfma(a[u].w, b.w, acc[u]);
}
// y[0..13) += Wfrag[i] . B  for the 13 n-tiles, fragments contiguous with stride `stride` floats
__device__ __forceinline__ void mma_all_ntiles(f4 (&y)[13], const float* wfrag, size_t stride, const f4 b, int lane) {
    {
        f4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const f4*>(wfrag + (size_t)u * stride + lane * 4);
        mma_group<4>(&y[0], a, b);
    }
#pragma unroll
    for (int i0 = 4; i0 < 13; i0 += 3) {
        f4 a[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) a[u] = *reinterpret_cast<const f4*>(wfrag + (size_t)(i0 + u) * stride + lane * 4);
        mma_group<3>(&y[i0], a, b);
    }
}

__device__ __forceinline__ f4 ldg4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ f4 lds4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ f4 zero4() { return f4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

struct FusedLayer {
    const float *ln0g, *ln0b, *ln1g, *ln1b;   // [208] zero padded
    const float* wqkv;                        // [39 tiles][13 chunks][256]  tiles: Q 0..12, K 13..25, V 26..38
    const float* bqkv;                        // [39*16]
    const float* wo;                          // [2 heads][13 n-tiles][7 d-chunks][256]
    const float* bo;                          // [208]
    const float* w1;                          // [50 hidden tiles][13 chunks][256]
    const float* b1;                          // [800]
    const float* w2;                          // [50 hidden chunks][13 n-tiles][256]
    const float* b2;                          // [208]
};

struct FusedArgs {
    // graph + queries
    const int64_t* indptr; const int32_t* nbr; const int32_t* eid; const double* ts; int64_t num_nodes;
    const int64_t *src, *dst; const double* times;
    const int32_t* hist_len; const int64_t* end_pos; const CallDims* cd;
    // tables + small weights
    const float *node_feat, *edge_feat, *time_w, *time_b, *lut;
    const float* proj[4];       // per channel: [4 tiles][nchunk_ch][256]
    const float* bias_x;        // [208] projection biases in model-dim order
    FusedLayer layer[DYGNN_MAX_LAYERS];
    const float *outT, *outb;   // output layer: transposed [200][Fn], bias [Fn]
    float *out_src, *out_dst;
    float* tap_enc; float* tap_layer[DYGNN_MAX_LAYERS];
    int64_t B;
    int Fn, Fe, Ft, P, L, NL, Tmax;
    int nchunk[4];
    float qscale;
};

// ------------------------------------------------------------------------------------------------
// LayerNorm over the register-resident X^T (two-pass, biased variance, eps 1e-5) -> Xn in LDS
// ------------------------------------------------------------------------------------------------
template <int NTILES>
__device__ __forceinline__ void layernorm_to_lds(const f4 (&x)[7], float* lds, const float* gamma, const float* beta,
                                                 int tile0, int tt, int hf, int c, int g, int which) {
    float* st = lds + kLdsMisc + which * 256;      // [sum: 2*64][var: 2*64]
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NTILES; ++i) s += (x[i].x + x[i].y) + (x[i].z + x[i].w);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (g == 0) st[hf * 64 + 16 * tt + c] = s;
    __syncthreads();
    const float mean = (st[16 * tt + c] + st[64 + 16 * tt + c]) * (1.0f / kD);
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NTILES; ++i) {
        const int n = 16 * (tile0 + i) + 4 * g;
        if (n < kD) {   // rows 200..207 are padding
            const float d0 = x[i].x - mean, d1 = x[i].y - mean, d2 = x[i].z - mean, d3 = x[i].w - mean;
            v += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (g == 0) st[128 + hf * 64 + 16 * tt + c] = v;
    __syncthreads();
    const float var = (st[128 + 16 * tt + c] + st[128 + 64 + 16 * tt + c]) * (1.0f / kD);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    float* row = lds + kLdsXn + (16 * tt + c) * kXS;
#pragma unroll
    for (int i = 0; i < NTILES; ++i) {
        const int n = 16 * (tile0 + i) + 4 * g;
        if (n < kXS) {
            const f4 gm = ldg4(gamma + n), bt = ldg4(beta + n);
            f4 y;
            y.x = (x[i].x - mean) * rstd * gm.x + bt.x;
            y.y = (x[i].y - mean) * rstd * gm.y + bt.y;
            y.z = (x[i].z - mean) * rstd * gm.z + bt.z;
            y.w = (x[i].w - mean) * rstd * gm.w + bt.w;
            *reinterpret_cast<f4*>(row + n) = y;
        }
    }
}

// B fragments of this wave's token tile for a K=200 product, from an LDS [tok][kXS] buffer
__device__ __forceinline__ void load_bfrags(f4 (&b)[kKC], const float* buf, int tt, int c, int g) {
    const float* row = buf + (16 * tt + c) * kXS + 4 * g;
#pragma unroll
    for (int kc = 0; kc < kKC; ++kc) b[kc] = lds4(row + 16 * kc);
}

// acc0/acc1 = W[tile0] . Xn^T, W[tile1] . Xn^T  (K = 200, 13 chunks), packed fragments at w0/w1
__device__ __forceinline__ void gemm_pair_k200(f4& acc0, f4& acc1, const float* w0, const float* w1, const f4 (&b)[kKC], int lane) {
#pragma unroll
    for (int kc = 0; kc < kKC; ++kc) {
        const f4 a0 = ldg4(w0 + kc * kFrag + lane * 4);
        const f4 a1 = ldg4(w1 + kc * kFrag + lane * 4);
        mma_chunk2(acc0, acc1, a0, a1, b[kc]);
    }
}

template <int NTILES>
__device__ __forceinline__ void tap_store(const f4 (&x)[7], float* base, int64_t b, int Tmax, int T, int tile0, int tt, int c, int g) {
    if (base == nullptr) return;
    const int tok = 16 * tt + c;
    if (tok >= T) return;
#pragma unroll
    for (int i = 0; i < NTILES; ++i) {
        const int n = 16 * (tile0 + i) + 4 * g;
        if (n < kD) *reinterpret_cast<f4*>(base + ((size_t)b * Tmax + tok) * kD + n) = x[i];
    }
}

// ------------------------------------------------------------------------------------------------
// patch-projection of one channel into NT consecutive resident tiles x[LOCAL0 .. LOCAL0+NT)
// bfn(kc) returns this lane's 4 feature values k = 16kc+4g .. +3 of its token
// ------------------------------------------------------------------------------------------------
template <int LOCAL0, int NTL, typename BF>
__device__ __forceinline__ void project_channel(f4 (&x)[7], const float* wp, int first_pack_tile, int nchunk, int lane, BF bfn) {
    for (int kc = 0; kc < nchunk; ++kc) {
        const f4 b = bfn(kc);
        f4 a[NTL];
#pragma unroll
        for (int i = 0; i < NTL; ++i) a[i] = ldg4(wp + ((size_t)(first_pack_tile + i) * nchunk + kc) * kFrag + lane * 4);
#pragma unroll
        for (int i = 0; i < NTL; ++i) x[LOCAL0 + i] = mfma(a[i].x, b.x, x[LOCAL0 + i]);
#pragma unroll
        for (int i = 0; i < NTL; ++i) x[LOCAL0 + i] = mfma(a[i].y, b.y, x[LOCAL0 + i]);
#pragma unroll
        for (int i = 0; i < NTL; ++i) x[LOCAL0 + i] = mfma(a[i].z, b.z, x[LOCAL0 + i]);
#pragma unroll
        for (int i = 0; i < NTL; ++i) x[LOCAL0 + i] = mfma(a[i].w, b.w, x[LOCAL0 + i]);
    }
}

// ================================================================================================
__global__ __launch_bounds__(512, 2) void k_dygformer_fused(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tt = wave & 3, hf = wave >> 2;
    const int c = lane & 15, g = lane >> 4;
    const int64_t b = blockIdx.x;

    const CallDims cd = *a.cd;
    const int Ss = cd.S_s, Sd = cd.S_d, Ts = cd.T_s, T = cd.T;
    const int S = Ss + Sd;

    // ---- zero LDS once: padding columns / rows of never-written tokens are read as MFMA operands
    for (int i = tid; i < kLdsBytes / 16; i += 512) reinterpret_cast<f4*>(lds)[i] = zero4();
    __syncthreads();

    // ---- windows (pad_sequences, DyGFormer.py:228-245) into the (still unused) V buffer
    int32_t* ids = reinterpret_cast<int32_t*>(lds + kLdsV);
    int32_t* eids = ids + S;
    float* dts = reinterpret_cast<float*>(eids + S);
    int32_t* c0 = reinterpret_cast<int32_t*>(dts + S);
    int32_t* c1 = c0 + S;
    const double tq = a.times[b];
    for (int p = tid; p < S; p += 512) {
        const bool is_dst = p >= Ss;
        const int j = is_dst ? p - Ss : p;
        const int64_t q = is_dst ? a.B + b : b;
        const int32_t len = a.hist_len[q];
        const int32_t m = len < a.L - 1 ? len : a.L - 1;
        int32_t id = 0, e = 0;
        float tn = 0.f;
        if (j == 0) {
            id = (int32_t)(is_dst ? a.dst[b] : a.src[b]); tn = (float)tq;
        } else if (j <= m) {
            const int64_t pos = a.end_pos[q] - m + (j - 1);
            id = a.nbr[pos]; e = a.eid[pos]; tn = (float)a.ts[pos];
        }
        ids[p] = id; eids[p] = e;
        dts[p] = (float)(tq - (double)tn);                      // DyGFormer.py:263
    }
    __syncthreads();
    // ---- co-occurrence counts (DyGFormer.py:337-393)
    for (int p = tid; p < S; p += 512) {
        const int32_t v = ids[p];
        int32_t cs = 0, cdn = 0;
        for (int q = 0; q < Ss; ++q) cs += (ids[q] == v);
        for (int q = Ss; q < S; ++q) cdn += (ids[q] == v);
        if (v == 0) { cs = 0; cdn = 0; }
        c0[p] = cs; c1[p] = cdn;
    }
    __syncthreads();

    // ---- resident residual stream X^T: tiles tile0 .. tile0+6 (hf=0) / +5 (hf=1) for tokens 16tt..16tt+15
    const int tile0 = hf ? 7 : 0;
    const bool active = 16 * tt < T;                 // wave-uniform: this token tile holds real tokens
    f4 x[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) x[i] = (hf == 0 || i < 6) ? ldg4(a.bias_x + 16 * (tile0 + i) + 4 * g) : zero4();

    if (active) {
        const int tok = 16 * tt + c;
        const bool tv = tok < T;
        const int pos0 = tv ? (tok < Ts ? tok * a.P : Ss + (tok - Ts) * a.P) : 0;
        const int P = a.P;
        auto gather = [&](const float* table, const int32_t* idx, int F, int kc) -> f4 {
            const int k = 16 * kc + 4 * g;
            const int pp = k / F;
            if (!tv || pp >= P) return zero4();
            return ldg4(table + (size_t)idx[pos0 + pp] * F + (k - pp * F));                          // DyGFormer.py:259-261
        };
        auto timef = [&](int kc) -> f4 {
            const int k = 16 * kc + 4 * g;
            const int pp = k / a.Ft;
            if (!tv || pp >= P || ids[pos0 + pp] == 0) return zero4();                                 // DyGFormer.py:266
            const int f = k - pp * a.Ft;
            const float dt = dts[pos0 + pp];
            const f4 w = ldg4(a.time_w + f), bb = ldg4(a.time_b + f);
            f4 r;
            r.x = cosf(fmaf(dt, w.x, bb.x)); r.y = cosf(fmaf(dt, w.y, bb.y));
            r.z = cosf(fmaf(dt, w.z, bb.z)); r.w = cosf(fmaf(dt, w.w, bb.w));
            return r;
        };
        auto coocf = [&](int kc) -> f4 {
            f4 r;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = 16 * kc + 4 * g + t;
                const int pp = k / 50;
                float v = 0.f;
                if (tv && pp < P) {
                    const int j = k - pp * 50;
                    v = a.lut[(size_t)c0[pos0 + pp] * 50 + j] + a.lut[(size_t)c1[pos0 + pp] * 50 + j];   // DyGFormer.py:409-411
                }
                r[t] = v;
            }
            return r;
        };
        if (hf == 0) {
            project_channel<0, 4>(x, a.proj[0], 0, a.nchunk[0], lane, [&](int kc) { return gather(a.node_feat, ids, a.Fn, kc); });
            project_channel<3, 4>(x, a.proj[1], 0, a.nchunk[1], lane, [&](int kc) { return gather(a.edge_feat, eids, a.Fe, kc); });
            project_channel<6, 1>(x, a.proj[2], 0, a.nchunk[2], lane, timef);
        } else {
            project_channel<0, 3>(x, a.proj[2], 1, a.nchunk[2], lane, timef);
            project_channel<2, 4>(x, a.proj[3], 0, a.nchunk[3], lane, coocf);
        }
    }
    __syncthreads();     // everyone is done with ids/eids/dts/c0/c1 (they live in the V buffer)
    // re-zero the part of V that held the window arrays (rows of absent tokens must stay finite zeros)
    for (int i = tid; i < (5 * S + 3) / 4; i += 512) reinterpret_cast<f4*>(lds + kLdsV)[i] = zero4();
    if (hf == 0) tap_store<7>(x, a.tap_enc, b, a.Tmax, T, tile0, tt, c, g); else tap_store<6>(x, a.tap_enc, b, a.Tmax, T, tile0, tt, c, g);

    float* Xn = lds + kLdsXn;
    float* Kb = lds + kLdsK;
    float* Vb = lds + kLdsV;

    for (int l = 0; l < a.NL; ++l) {
        const FusedLayer& W = a.layer[l];
        // ================= LN0 -> Xn =================
        if (hf == 0) layernorm_to_lds<7>(x, lds, W.ln0g, W.ln0b, tile0, tt, hf, c, g, 0);
        else layernorm_to_lds<6>(x, lds, W.ln0g, W.ln0b, tile0, tt, hf, c, g, 0);
        __syncthreads();

        // ================= QKV =================
        f4 qa[7];     // Q^T tiles 6hf .. 6hf+6 (rows outside head hf are zeroed below), already scaled
        if (active) {
            f4 bf[kKC];
            load_bfrags(bf, Xn, tt, c, g);
            // 20 tiles per wave, two at a time: Q tiles 6hf..6hf+6 (kept in registers, scaled), then the 13 K/V
            // tiles of this half (hf=0 -> K 0..6, V 0..5 ; hf=1 -> K 7..12, V 6..12) written to LDS.
            const int nk = hf ? 6 : 7, k_first = hf ? 7 : 0, v_first = hf ? 6 : 0;
            auto kv_tile = [&](int s, int& ptile, float*& dbuf, int& ncol) {      // slot s in 0..12
                const bool isk = s < nk;
                const int tile = isk ? k_first + s : v_first + (s - nk);
                ptile = (isk ? 13 : 26) + tile; dbuf = isk ? Kb : Vb; ncol = 16 * tile + 4 * g;
            };
            auto wq = [&](int ptile) { return W.wqkv + (size_t)ptile * kKC * kFrag; };
#pragma unroll
            for (int j = 0; j < 6; j += 2) {
                const int t0 = 6 * hf + j;
                f4 acc0 = ldg4(W.bqkv + 16 * t0 + 4 * g);
                f4 acc1 = ldg4(W.bqkv + 16 * (t0 + 1) + 4 * g);
                gemm_pair_k200(acc0, acc1, wq(t0), wq(t0 + 1), bf, lane);
                qa[j] = acc0 * a.qscale;
                qa[j + 1] = acc1 * a.qscale;
            }
            {   // Q tile 6 paired with K/V slot 0
                int pt, nc; float* db;
                kv_tile(0, pt, db, nc);
                f4 acc0 = ldg4(W.bqkv + 16 * (6 * hf + 6) + 4 * g);
                f4 acc1 = ldg4(W.bqkv + 16 * pt + 4 * g);
                gemm_pair_k200(acc0, acc1, wq(6 * hf + 6), wq(pt), bf, lane);
                qa[6] = acc0 * a.qscale;
                if (nc < kXS) *reinterpret_cast<f4*>(db + (16 * tt + c) * kXS + nc) = acc1;
            }
            // rows of tile 6 that belong to the other head contribute nothing to this head's q.k
            if (hf == 0) { if (g != 0) qa[6] = zero4(); } else { if (g == 0) qa[0] = zero4(); }
            for (int s2 = 1; s2 < 13; s2 += 2) {
                int pt0, pt1, nc0, nc1; float *db0, *db1;
                kv_tile(s2, pt0, db0, nc0);
                kv_tile(s2 + 1, pt1, db1, nc1);
                f4 acc0 = ldg4(W.bqkv + 16 * pt0 + 4 * g);
                f4 acc1 = ldg4(W.bqkv + 16 * pt1 + 4 * g);
                gemm_pair_k200(acc0, acc1, wq(pt0), wq(pt1), bf, lane);
                if (nc0 < kXS) *reinterpret_cast<f4*>(db0 + (16 * tt + c) * kXS + nc0) = acc0;
                if (nc1 < kXS) *reinterpret_cast<f4*>(db1 + (16 * tt + c) * kXS + nc1) = acc1;
            }
        }
        __syncthreads();

        // ================= attention for (token tile tt, head hf) =================
        f4 y[kNT];
#pragma unroll
        for (int i = 0; i < kNT; ++i) y[i] = zero4();
        if (active) {
            f4 sa[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) sa[kt] = zero4();
            // S^T[key][query] = sum_d K[key][d] * Q^T[d][query]
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int col = 16 * (6 * hf + j) + 4 * g;
#pragma unroll
                for (int kt = 0; kt < 4; kt += 2) {
                    const f4 k0 = lds4(Kb + (16 * kt + c) * kXS + col);
                    const f4 k1 = lds4(Kb + (16 * (kt + 1) + c) * kXS + col);
                    mma_chunk2(sa[kt], sa[kt + 1], k0, k1, qa[j]);
                }
            }
            // softmax over keys (rows: 16kt + 4g + r); keys >= T do not exist
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * kt + 4 * g + r;
                    if (key >= T) sa[kt][r] = -INFINITY;
                    mx = fmaxf(mx, sa[kt][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { sa[kt][r] = expf(sa[kt][r] - mx); sum += sa[kt][r]; }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) sa[kt] *= inv;
            // O^T[d][query] = sum_key V[key][100hf + d] * P^T[key][query]   (7 d-tiles, rows >= 100 are junk x 0-weights)
            f4 oa[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) oa[j] = zero4();
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                f4 va[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const float* vp = Vb + (16 * kt + 4 * g) * kXS + kHD * hf + 16 * j + c;
                    va[j].x = vp[0]; va[j].y = vp[kXS]; va[j].z = vp[2 * kXS]; va[j].w = vp[3 * kXS];
                }
                mma_group<4>(&oa[0], &va[0], sa[kt]);
                mma_group<3>(&oa[4], &va[4], sa[kt]);
            }
            // out-projection, K-split over heads: y^T[n][q] += Wo[n][100hf + d] * O^T[d][q]
            const float* wo = W.wo + (size_t)hf * kNT * 7 * kFrag;
#pragma unroll
            for (int j = 0; j < 7; ++j) mma_all_ntiles(y, wo + (size_t)j * kFrag, (size_t)7 * kFrag, oa[j], lane);
        }
        // exchange partial sums with the partner wave (tt, 1-hf): 52 fragment slots in the (dead) Xn buffer
        // (51 fit; the last one lives in the misc region)
        {
            auto slot = [&](int tile) -> float* {
                const int idx = tt * kNT + tile;
                return idx < 51 ? Xn + (size_t)idx * kFrag + lane * 4 : lds + kLdsMisc + 512 + lane * 4;
            };
#pragma unroll
            for (int i = 0; i < kNT; ++i) {
                const bool mine = hf ? (i >= 7) : (i < 7);
                if (!mine) *reinterpret_cast<f4*>(slot(i)) = y[i];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                if (hf == 0 || i < 6) {
                    const int tile = tile0 + i;
                    const f4 other = lds4(slot(tile));
                    const f4 bias = ldg4(W.bo + 16 * tile + 4 * g);
                    const f4 mine_v = (i < 6) ? (hf ? y[7 + (i < 6 ? i : 0)] : y[i]) : y[6];
                    x[i] = x[i] + ((mine_v + other) + bias);
                }
            }
        }

        // ================= LN1 -> Xn (barriers inside also order the exchange reads before the Xn writes) =================
        if (hf == 0) layernorm_to_lds<7>(x, lds, W.ln1g, W.ln1b, tile0, tt, hf, c, g, 1);
        else layernorm_to_lds<6>(x, lds, W.ln1g, W.ln1b, tile0, tt, hf, c, g, 1);
        __syncthreads();

        // ================= FFN: hidden half hf, K-split second GEMM =================
#pragma unroll
        for (int i = 0; i < kNT; ++i) y[i] = zero4();
        if (active) {
            f4 bf[kKC];
            load_bfrags(bf, Xn, tt, c, g);
            for (int hh = 0; hh < kHT / 2; ++hh) {
                const int ht = hf * (kHT / 2) + hh;
                const float* w1 = W.w1 + (size_t)ht * kKC * kFrag;
                f4 h0 = ldg4(W.b1 + 16 * ht + 4 * g), h1 = zero4();
#pragma unroll
                for (int kc = 0; kc < 12; kc += 2) {
                    const f4 a0 = ldg4(w1 + kc * kFrag + lane * 4);
                    const f4 a1 = ldg4(w1 + (kc + 1) * kFrag + lane * 4);
                    h0 = mfma(a0.x, bf[kc].x, h0); h1 = mfma(a1.x, bf[kc + 1].x, h1);
                    h0 = mfma(a0.y, bf[kc].y, h0); h1 = mfma(a1.y, bf[kc + 1].y, h1);
                    h0 = mfma(a0.z, bf[kc].z, h0); h1 = mfma(a1.z, bf[kc + 1].z, h1);
                    h0 = mfma(a0.w, bf[kc].w, h0); h1 = mfma(a1.w, bf[kc + 1].w, h1);
                }
                mma_chunk(h0, ldg4(w1 + 12 * kFrag + lane * 4), bf[12]);
                f4 h = h0 + h1;
                h.x = gelu_erf(h.x); h.y = gelu_erf(h.y); h.z = gelu_erf(h.z); h.w = gelu_erf(h.w);   // DyGFormer.py:458
                const float* w2 = W.w2 + (size_t)ht * kNT * kFrag;
                mma_all_ntiles(y, w2, (size_t)kFrag, h, lane);
            }
        }
        {   // K and V are dead after attention (every wave passed LN1's barriers): 52 slots from the start of K
            auto slot = [&](int tile) -> float* { return Kb + (size_t)(tt * kNT + tile) * kFrag + lane * 4; };
#pragma unroll
            for (int i = 0; i < kNT; ++i) {
                const bool mine = hf ? (i >= 7) : (i < 7);
                if (!mine) *reinterpret_cast<f4*>(slot(i)) = y[i];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                if (hf == 0 || i < 6) {
                    const int tile = tile0 + i;
                    const f4 other = lds4(slot(tile));
                    const f4 bias = ldg4(W.b2 + 16 * tile + 4 * g);
                    const f4 mine_v = (i < 6) ? (hf ? y[7 + (i < 6 ? i : 0)] : y[i]) : y[6];
                    x[i] = x[i] + ((mine_v + other) + bias);
                }
            }
        }
        if (hf == 0) tap_store<7>(x, a.tap_layer[l], b, a.Tmax, T, tile0, tt, c, g); else tap_store<6>(x, a.tap_layer[l], b, a.Tmax, T, tile0, tt, c, g);
        __syncthreads();   // exchange reads of K buffer done before the next layer's K writes / the pooling scratch
    }

    // ================= per-side mean over tokens + output layer (DyGFormer.py:181-192) =================
    {
        float* pool = Kb;                       // [side][tt][208]
        const int tok = 16 * tt + c;
        const bool in_src = tok < Ts, in_dst = tok >= Ts && tok < T;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            if (hf == 0 || i < 6) {
                f4 vs = in_src ? x[i] : zero4();
                f4 vd = in_dst ? x[i] : zero4();
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        vs[r] += __shfl_xor(vs[r], o, 64);
                        vd[r] += __shfl_xor(vd[r], o, 64);
                    }
                }
                if (c == 0) {
                    const int n = 16 * (tile0 + i) + 4 * g;
                    *reinterpret_cast<f4*>(pool + (0 * 4 + tt) * kDP + n) = vs;
                    *reinterpret_cast<f4*>(pool + (1 * 4 + tt) * kDP + n) = vd;
                }
            }
        }
        __syncthreads();
        float* mean = lds + kLdsXn;             // [2][208]
        const int Td = T - Ts;
        for (int i = tid; i < 2 * kDP; i += 512) {
            const int side = i / kDP, n = i % kDP;
            const float s = (pool[(side * 4 + 0) * kDP + n] + pool[(side * 4 + 1) * kDP + n]) +
                            (pool[(side * 4 + 2) * kDP + n] + pool[(side * 4 + 3) * kDP + n]);
            mean[i] = s / (float)(side ? Td : Ts);
        }
        __syncthreads();
        for (int i = tid; i < 2 * a.Fn; i += 512) {
            const int side = i / a.Fn, j = i % a.Fn;
            float acc = 0.f;
            for (int k = 0; k < kD; ++k) acc = fmaf(mean[side * kDP + k], a.outT[(size_t)k * a.Fn + j], acc);
            (side ? a.out_dst : a.out_src)[b * a.Fn + j] = acc + a.outb[j];
        }
    }
}

// ================================================================================================
// packing
// ================================================================================================
// dst fragment (slot = chunk_major ? chunk*n_tiles + tile : tile*n_chunks + chunk), lane (c,g), element t:
//   row = r0 + 16*tile + c   valid iff 0 <= row < rmax          (row of src, ld = src row stride)
//   col = c0 + 16*chunk + 4g + t   valid iff cmin <= col < cmax
__global__ void k_pack_frag(const float* __restrict__ src, int ld, int n_tiles, int n_chunks, int r0, int rmax, int c0, int cmin,
                            int cmax, int chunk_major, float* __restrict__ dst) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)n_tiles * n_chunks * kFrag;
    if (idx >= total) return;
    const int t = idx & 3, lane = (idx >> 2) & 63;
    const int64_t slot = idx >> 8;
    const int tile = chunk_major ? (int)(slot % n_tiles) : (int)(slot / n_chunks);
    const int chunk = chunk_major ? (int)(slot / n_tiles) : (int)(slot % n_chunks);
    const int c = lane & 15, g = lane >> 4;
    const int row = r0 + 16 * tile + c;
    const int col = c0 + 16 * chunk + 4 * g + t;
    float v = 0.f;
    if (row >= 0 && row < rmax && col >= cmin && col < cmax) v = src[(size_t)row * ld + col];
    dst[idx] = v;
}

__global__ void k_pack_vec(const float* __restrict__ src, int n_valid, int src_off, float* __restrict__ dst, int dst_off, int n_total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    dst[dst_off + i] = i < n_valid ? src[src_off + i] : 0.f;
}

struct FusedPackLayout {     // float offsets relative to PackedLayout.fused
    size_t proj[4]; int nchunk[4];
    size_t bias_x;
    struct L { size_t ln0g, ln0b, ln1g, ln1b, wqkv, bqkv, wo, bo, w1, b1, w2, b2; } layer[DYGNN_MAX_LAYERS];
    size_t total;
};

static FusedPackLayout make_fused_layout(const Dims& d) {
    FusedPackLayout f;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 63) & ~size_t(63); return r; };
    const int K[4] = {d.P * d.Fn, d.P * d.Fe, d.P * d.Ft, d.P * d.C};
    for (int c = 0; c < 4; ++c) { f.nchunk[c] = (K[c] + 15) / 16; f.proj[c] = take((size_t)4 * f.nchunk[c] * kFrag); }
    f.bias_x = take(kDP);
    for (int l = 0; l < d.NL; ++l) {
        auto& L = f.layer[l];
        L.ln0g = take(kDP); L.ln0b = take(kDP); L.ln1g = take(kDP); L.ln1b = take(kDP);
        L.wqkv = take((size_t)39 * kKC * kFrag); L.bqkv = take(39 * 16);
        L.wo = take((size_t)2 * kNT * 7 * kFrag); L.bo = take(kDP);
        L.w1 = take((size_t)kHT * kKC * kFrag); L.b1 = take(kHid);
        L.w2 = take((size_t)kHT * kNT * kFrag); L.b2 = take(kDP);
    }
    f.total = o;
    return f;
}

bool fused_supported(const Dims& d) {
    return d.C == 50 && d.H == 2 && d.Tmax <= kTok && d.Fn % 4 == 0 && d.Fe % 4 == 0 && d.Ft % 4 == 0 &&
           (size_t)d.Tmax * d.P * 5 * 4 <= (size_t)kBufFloats * 4 && d.NL <= DYGNN_MAX_LAYERS;
}

size_t fused_packed_floats(const Dims& d) { return fused_supported(d) ? make_fused_layout(d).total : 0; }

static int pack_frag(const float* src, int ld, int n_tiles, int n_chunks, int r0, int rmax, int c0, int cmin, int cmax,
                     int chunk_major, float* dst, hipStream_t s) {
    const int64_t total = (int64_t)n_tiles * n_chunks * kFrag;
    hipLaunchKernelGGL(k_pack_frag, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, src, ld, n_tiles, n_chunks, r0, rmax, c0,
                       cmin, cmax, chunk_major, dst);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}
static int pack_vec(const float* src, int n_valid, int src_off, float* dst, int dst_off, int n_total, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_vec, dim3((n_total + 255) / 256), dim3(256), 0, s, src, n_valid, src_off, dst, dst_off, n_total);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

int pack_fused(const Dims& d, const PackedLayout& pl, const dygnn_dygformer_weights* w, float* packed, hipStream_t s) {
    const FusedPackLayout f = make_fused_layout(d);
    float* base = packed + pl.fused;
    const float* pw[4] = {w->proj_node_w, w->proj_edge_w, w->proj_time_w, w->proj_cooc_w};
    const float* pb[4] = {w->proj_node_b, w->proj_edge_b, w->proj_time_b, w->proj_cooc_b};
    const int K[4] = {d.P * d.Fn, d.P * d.Fe, d.P * d.Ft, d.P * d.C};
    for (int c = 0; c < 4; ++c) {
        // model rows 50c .. 50c+49 live in flat tiles (50c)/16 .. +3; src row = flat row - 50c
        const int t0 = (50 * c) / 16;
        if (int rc = pack_frag(pw[c], K[c], 4, f.nchunk[c], 16 * t0 - 50 * c, 50, 0, 0, K[c], 0, base + f.proj[c], s)) return rc;
        if (int rc = pack_vec(pb[c], 50, 0, base + f.bias_x, 50 * c, 50, s)) return rc;
    }
    if (int rc = pack_vec(pb[0], 0, 0, base + f.bias_x, 200, 8, s)) return rc;       // zero the 8 padding rows
    for (int l = 0; l < d.NL; ++l) {
        const dygnn_encoder_layer_weights& L = w->layers[l];
        const auto& F = f.layer[l];
        if (int rc = pack_vec(L.norm0_weight, kD, 0, base + F.ln0g, 0, kDP, s)) return rc;
        if (int rc = pack_vec(L.norm0_bias, kD, 0, base + F.ln0b, 0, kDP, s)) return rc;
        if (int rc = pack_vec(L.norm1_weight, kD, 0, base + F.ln1g, 0, kDP, s)) return rc;
        if (int rc = pack_vec(L.norm1_bias, kD, 0, base + F.ln1b, 0, kDP, s)) return rc;
        for (int part = 0; part < 3; ++part) {      // q | k | v row blocks of in_proj
            if (int rc = pack_frag(L.in_proj_weight + (size_t)part * kD * kD, kD, kNT, kKC, 0, kD, 0, 0, kD, 0,
                                   base + F.wqkv + (size_t)part * kNT * kKC * kFrag, s)) return rc;
            if (int rc = pack_vec(L.in_proj_bias, kD, part * kD, base + F.bqkv, part * kDP, kDP, s)) return rc;
        }
        for (int h = 0; h < 2; ++h)
            if (int rc = pack_frag(L.out_proj_weight, kD, kNT, 7, 0, kD, kHD * h, kHD * h, kHD * (h + 1), 0,
                                   base + F.wo + (size_t)h * kNT * 7 * kFrag, s)) return rc;
        if (int rc = pack_vec(L.out_proj_bias, kD, 0, base + F.bo, 0, kDP, s)) return rc;
        if (int rc = pack_frag(L.ffn0_weight, kD, kHT, kKC, 0, kHid, 0, 0, kD, 0, base + F.w1, s)) return rc;
        if (int rc = pack_vec(L.ffn0_bias, kHid, 0, base + F.b1, 0, kHid, s)) return rc;
        if (int rc = pack_frag(L.ffn1_weight, kHid, kNT, kHT, 0, kD, 0, 0, kHid, 1, base + F.w2, s)) return rc;
        if (int rc = pack_vec(L.ffn1_bias, kD, 0, base + F.b2, 0, kDP, s)) return rc;
    }
    return DYGNN_OK;
}

int window_lengths_device(const Dims& d, const dygnn_csr* csr, const int64_t* src, const int64_t* dst, const double* times,
                          int64_t B, char* ws, const WorkspaceLayout& wl, hipStream_t s);   // dygformer_generic.hip

int forward_fused(const Dims& d, const PackedLayout& pl, const dygnn_dygformer_weights* w, const float* packed,
                  const dygnn_csr* csr, const float* node_feat, const float* edge_feat, const int64_t* src,
                  const int64_t* dst, const double* times, int64_t B, float* out_src, float* out_dst, char* ws,
                  const WorkspaceLayout& wl, const dygnn_dygformer_taps* taps, hipStream_t s) {
    if (!fused_supported(d)) { set_error("fused kernel: unsupported shape"); return DYGNN_E_UNSUPPORTED; }
    if (int rc = window_lengths_device(d, csr, src, dst, times, B, ws, wl, s)) return rc;
    const FusedPackLayout f = make_fused_layout(d);
    const float* base = packed + pl.fused;
    FusedArgs a{};
    a.indptr = csr->indptr; a.nbr = csr->nbr; a.eid = csr->eid; a.ts = csr->ts; a.num_nodes = csr->num_nodes;
    a.src = src; a.dst = dst; a.times = times;
    a.hist_len = reinterpret_cast<const int32_t*>(ws + wl.hist_len);
    a.end_pos = reinterpret_cast<const int64_t*>(ws + wl.end_pos);
    a.cd = reinterpret_cast<const CallDims*>(ws + wl.dims);
    a.node_feat = node_feat; a.edge_feat = edge_feat; a.time_w = w->time_w; a.time_b = w->time_b; a.lut = packed + pl.lut;
    for (int c = 0; c < 4; ++c) { a.proj[c] = base + f.proj[c]; a.nchunk[c] = f.nchunk[c]; }
    a.bias_x = base + f.bias_x;
    for (int l = 0; l < d.NL; ++l) {
        const auto& F = f.layer[l];
        FusedLayer& L = a.layer[l];
        L.ln0g = base + F.ln0g; L.ln0b = base + F.ln0b; L.ln1g = base + F.ln1g; L.ln1b = base + F.ln1b;
        L.wqkv = base + F.wqkv; L.bqkv = base + F.bqkv; L.wo = base + F.wo; L.bo = base + F.bo;
        L.w1 = base + F.w1; L.b1 = base + F.b1; L.w2 = base + F.w2; L.b2 = base + F.b2;
        a.tap_layer[l] = taps ? taps->layer_out[l] : nullptr;
    }
    a.outT = packed + pl.outputT; a.outb = w->output_b;
    a.out_src = out_src; a.out_dst = out_dst;
    a.tap_enc = taps ? taps->encoder_input : nullptr;
    a.B = B; a.Fn = d.Fn; a.Fe = d.Fe; a.Ft = d.Ft; a.P = d.P; a.L = d.L; a.NL = d.NL; a.Tmax = d.Tmax;
    a.qscale = (float)sqrt(1.0 / (double)d.hd);
    if (taps && taps->seq_lens) DYGNN_HIP(hipMemcpyAsync(taps->seq_lens, ws + wl.dims + 2 * sizeof(int32_t), 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    static bool attr_set = false;
    if (!attr_set) {
        DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dygformer_fused), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_dygformer_fused, dim3((unsigned)B), dim3(512), kLdsBytes, s, a);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

}  // namespace dygnn
