// Row-block chains of the TGAT / TGN layer (see tgat_chain.h).  gfx950 only.
//
// One workgroup (8 wave64) owns R = 16*MT rows of a level.  Every product of the chain is Out[R][N] = Act[R][K] . W^T with the
// activations in LDS and the weight operand streamed from the packed buffer (L2-resident: a layer is 2 MB) straight into the MFMA A
// operand, `v_mfma_f32_16x16x4_f32`, transposed form: accumulator tile = Out^T[n = 4g+r][m = c] for lane (c = lane&15, g = lane>>4), so
// a lane ends with four consecutive n of row c = one float4 store into the next product's LDS operand.
//  * "kc" products (nn.Linear weights [N][K], contraction along a weight row): fragment (tile, chunk) holds for lane (c,g) the float4
//    W[16 tile + c][16 chunk + 4g ..]; the lane reads Act[c][16 chunk + 4g ..] from LDS: 4 MFMAs per fragment, 16 k per step;
//  * "km" products (W_k,h^T q: contraction along weight ROWS): fragment (tile, step) holds W[4 step + g][64 tile + 4c ..]; the lane reads the
//    scalar Act[c][4 step + g]; its four elements feed four accumulators whose tiles interleave to 64 consecutive n.
// Why packed: read in place, a wave's float4 load of a [16 rows][16 k] operand is 16 strided 64-byte pieces; the vector L1 looks every
// quarter-wave's 16 lines up separately and the first version of these kernels ran at 270 ns per step whatever the prefetch depth
// (profiles/r02_tgn_chain_notes.md).  A packed fragment is one contiguous KiB.
// LDS row strides are 4 (mod 8) floats: the 16 rows of a b128 read (and the 16 x 4 words of the b32 read) fall on distinct banks; rows
// are zero-padded to the 16-k chunk so no step needs a mask.
// A row's result depends on that row's data only (an MFMA column never mixes with another), in the same order for either MT, so rows
// are bit-identical whatever batch or block they sit in.
#include "tgat_chain.h"

namespace dygnn {
namespace chain {

using f4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f4 cmfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
__host__ __device__ inline int r16(int K) { return (K + 15) & ~15; }
__host__ __device__ inline int pad_ld(int K) {           // smallest row stride >= K with stride % 8 == 4
    const int l = (K + 3) & ~3;
    return (l & 4) ? l : l + 4;
}

// ---- packing ----------------------------------------------------------------------------------------------------------------------
struct PackDesc {
    const float* src;
    int ld, N, K, type;            // type 0: kc (tile = 16 rows n, step = 16 k) ; 1: km (tile = 64 columns n, step = 4 rows k)
    int tph;                       // tiles per head (per-head products: tile t belongs to head t / tph, matrix src + head * hstride)
    int64_t hstride;
    uint32_t off, ntiles, nsteps;
};
constexpr int kMaxDesc = 6 * DYGNN_MAX_LAYERS + 2;
struct PackTable { PackDesc d[kMaxDesc]; int n; uint32_t total; };

__global__ __launch_bounds__(256) void k_pack(const PackTable tb, f4* __restrict__ dst) {
    const uint32_t frag = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frag >= tb.total) return;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    int di = 0;
    for (int i = 1; i < tb.n; ++i)
        if (frag >= tb.d[i].off) di = i;
    const PackDesc& d = tb.d[di];
    const uint32_t rel = frag - d.off, t = rel / d.nsteps, st = rel - t * d.nsteps;
    const int h = (int)t / d.tph, lt = (int)t - h * d.tph;
    const float* m = d.src + (int64_t)h * d.hstride;
    f4 v = f4{0.f, 0.f, 0.f, 0.f};
    if (d.type == 0) {
        const int n = 16 * lt + c, k = 16 * (int)st + 4 * g;
        if (n < d.N && k < d.K) v = *reinterpret_cast<const f4*>(m + (size_t)n * d.ld + k);
    } else {
        const int k = 4 * (int)st + g, n = 64 * lt + 4 * c;
        if (k < d.K && n < d.N) v = *reinterpret_cast<const f4*>(m + (size_t)k * d.ld + n);
    }
    dst[(size_t)frag * 64 + lane] = v;
}

static uint32_t kc_frags(int N, int K, int heads = 1) { return (uint32_t)heads * ((N + 15) / 16) * ((K + 15) / 16); }
static uint32_t km_frags(int N, int K, int heads = 1) { return (uint32_t)heads * ((N + 63) / 64) * ((K + 3) / 4); }

PackPlan plan_pack(int L, int Fn, int Ft, int Dkv, int H, int gru_Dm) {
    PackPlan p{};
    const int Dq = Fn + Ft, hd = Dq / H;
    uint32_t o = 0;
    for (int l = 0; l < L; ++l) {
        LayerPack& y = p.layer[l];
        y.q = o; o += kc_frags(Dq, Dq);
        y.k = o; o += km_frags(Dkv, hd, H);
        y.v = o; o += kc_frags(hd, Dkv, H);
        y.r = o; o += kc_frags(Dq, Dq);
        y.f1 = o; o += kc_frags(Fn, Dq + Fn);
        y.f2 = o; o += kc_frags(Fn, Fn);
    }
    if (gru_Dm > 0) {
        p.ih = o; o += kc_frags(3 * Fn, gru_Dm);
        p.hh = o; o += kc_frags(3 * Fn, Fn);
    }
    p.total = o;
    return p;
}

int pack(hipStream_t s, const PackPlan& p, int L, int Fn, int Ft, int Dkv, int H, const dygnn_tgat_weights* w, const dygnn_gru_weights* gru, int gru_Dm,
         float* dst) {
    PackTable tb{};
    const int Dq = Fn + Ft, hd = Dq / H;
    auto kc = [&](const float* src, int ld, int N, int K, uint32_t off, int heads = 1, int64_t hstride = 0) {
        const int tph = (N + 15) / 16;
        tb.d[tb.n++] = PackDesc{src, ld, N, K, 0, tph, hstride, off, (uint32_t)(heads * tph), (uint32_t)((K + 15) / 16)};
    };
    for (int l = 0; l < L; ++l) {
        const dygnn_tgat_layer_weights& Lw = w->layers[l];
        const LayerPack& y = p.layer[l];
        kc(Lw.query_w, Dq, Dq, Dq, y.q);
        tb.d[tb.n++] = PackDesc{Lw.key_w, Dkv, Dkv, hd, 1, (Dkv + 63) / 64, (int64_t)hd * Dkv, y.k, (uint32_t)(H * ((Dkv + 63) / 64)), (uint32_t)((hd + 3) / 4)};
        kc(Lw.value_w, Dkv, hd, Dkv, y.v, H, (int64_t)hd * Dkv);
        kc(Lw.res_w, Dq, Dq, Dq, y.r);
        kc(Lw.fc1_w, Dq + Fn, Fn, Dq + Fn, y.f1);
        kc(Lw.fc2_w, Fn, Fn, Fn, y.f2);
    }
    if (gru_Dm > 0) {
        kc(gru->weight_ih, gru_Dm, 3 * Fn, gru_Dm, p.ih);
        kc(gru->weight_hh, Fn, 3 * Fn, Fn, p.hh);
    }
    tb.total = p.total;
    hipLaunchKernelGGL(k_pack, dim3((p.total + 3) / 4), dim3(256), 0, s, tb, reinterpret_cast<f4*>(dst));
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

// ---- the weight stream of a wave ------------------------------------------------------------------------------------------------
// A stage gives wave w the tiles t = w, w + 8, ... < T.  Their fragments form ONE sequence of steps (tile by tile, k-step by k-step) that
// runs through a ring of U float4 registers filled U steps ahead -- across tile boundaries, so the stream only drains at the end of a
// stage.  Steady state is branch-free but for the tile epilogue (refills are always issued; beyond the stream they re-read the wave's
// first tile), so the compiler counts the loads in flight (s_waitcnt vmcnt(U-1)) instead of draining them.  The LDS operand of step
// s+1 is read before the MFMAs of step s; a step's MFMAs go to two (kc) / four (km) independent accumulators.
// AF: act(t) = LDS base of the activation operand of tile t.  Ep(t, acc): the finished tile.
constexpr int U = 8;

template <int MT, class AF, class Ep>
__device__ __forceinline__ void stream_kc(const f4* __restrict__ pk, int T, int nch, int wave, int lane, int lda, AF actf, Ep ep) {
    if (wave >= T) return;
    const int c = lane & 15, g = lane >> 4;
    const int S = ((T - wave + kWaves - 1) / kWaves) * nch;
    const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
    int pt = wave, pch = 0;                                           // prefetch cursor
    const f4* pp = pk + (size_t)wave * nch * 64 + lane;
    f4 ring[U];
    auto fetch = [&](f4& dst) {
        dst = *pp;
        pp += 64;
        if (++pch == nch) { pch = 0; pt += kWaves; pp = pk + (size_t)(pt < T ? pt : wave) * nch * 64 + lane; }
    };
#pragma unroll
    for (int u = 0; u < U; ++u) fetch(ring[u]);
    int ct = wave, cch = 0;                                           // consume cursor
    const float* ab = actf(ct) + c * lda + 4 * g;
    f4 acc0[MT], acc1[MT], bn[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) { acc0[j] = zero; acc1[j] = zero; bn[j] = *reinterpret_cast<const f4*>(ab + j * 16 * lda); }
    auto step = [&](f4& slot, bool refill) {
        const f4 w = slot;
        f4 b[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) b[j] = bn[j];
        const bool tile_end = cch + 1 == nch;
        const int nt = tile_end ? ct + kWaves : ct, nc = tile_end ? 0 : cch + 1;
        const float* nab = tile_end ? actf(nt < T ? nt : ct) + c * lda + 4 * g : ab;
#pragma unroll
        for (int j = 0; j < MT; ++j) bn[j] = *reinterpret_cast<const f4*>(nab + j * 16 * lda + 16 * nc);      // (behind the last step: a valid, unused read)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            acc0[j] = cmfma(w.x, b[j].x, acc0[j]);
            acc1[j] = cmfma(w.y, b[j].y, acc1[j]);
            acc0[j] = cmfma(w.z, b[j].z, acc0[j]);
            acc1[j] = cmfma(w.w, b[j].w, acc1[j]);
        }
        if (refill) fetch(slot);
        if (tile_end) {
            f4 r[MT];
#pragma unroll
            for (int j = 0; j < MT; ++j) { r[j] = acc0[j] + acc1[j]; acc0[j] = zero; acc1[j] = zero; }
            ep(ct, r);
        }
        ct = nt; cch = nc; ab = nab;
    };
    int s0 = 0;
    for (; s0 + U <= S; s0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) step(ring[u], true);
    }
#pragma unroll
    for (int u = 0; u < U - 1; ++u)
        if (s0 + u < S) step(ring[u], false);
}

// acc[tt][j][r] = Out[16j + c][64 t' + 16g + 4r + tt] (t' = the tile's block of n)
template <int MT, class AF, class Ep>
__device__ __forceinline__ void stream_km(const f4* __restrict__ pk, int T, int nst, int wave, int lane, int lda, AF actf, Ep ep) {
    if (wave >= T) return;
    const int c = lane & 15, g = lane >> 4;
    const int S = ((T - wave + kWaves - 1) / kWaves) * nst;
    const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
    int pt = wave, pst = 0;
    const f4* pp = pk + (size_t)wave * nst * 64 + lane;
    f4 ring[U];
    auto fetch = [&](f4& dst) {
        dst = *pp;
        pp += 64;
        if (++pst == nst) { pst = 0; pt += kWaves; pp = pk + (size_t)(pt < T ? pt : wave) * nst * 64 + lane; }
    };
#pragma unroll
    for (int u = 0; u < U; ++u) fetch(ring[u]);
    int ct = wave, cst = 0;
    const float* ab = actf(ct) + c * lda + g;
    f4 acc[4][MT];
    float bn[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        bn[j] = ab[j * 16 * lda];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt][j] = zero;
    }
    auto step = [&](f4& slot, bool refill) {
        const f4 w = slot;
        float b[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) b[j] = bn[j];
        const bool tile_end = cst + 1 == nst;
        const int nt = tile_end ? ct + kWaves : ct, ns = tile_end ? 0 : cst + 1;
        const float* nab = tile_end ? actf(nt < T ? nt : ct) + c * lda + g : ab;
#pragma unroll
        for (int j = 0; j < MT; ++j) bn[j] = nab[j * 16 * lda + 4 * ns];
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            acc[0][j] = cmfma(w.x, b[j], acc[0][j]);
            acc[1][j] = cmfma(w.y, b[j], acc[1][j]);
            acc[2][j] = cmfma(w.z, b[j], acc[2][j]);
            acc[3][j] = cmfma(w.w, b[j], acc[3][j]);
        }
        if (refill) fetch(slot);
        if (tile_end) {
            ep(ct, acc);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int j = 0; j < MT; ++j) acc[tt][j] = zero;
        }
        ct = nt; cst = ns; ab = nab;
    };
    int s0 = 0;
    for (; s0 + U <= S; s0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) step(ring[u], true);
    }
#pragma unroll
    for (int u = 0; u < U - 1; ++u)
        if (s0 + u < S) step(ring[u], false);
}

// query-input rows [h(self) | cos(w*0 + b)] of the block's rows into LDS (zero rows beyond the live count; zero padding columns).
// A wave owns rows wave, wave + 8, ...: their indices are loaded together, then their feature rows, column slot by column slot.
template <int MT>
__device__ __forceinline__ void fill_qin(float* qin, int ldq, const float* __restrict__ tw, const float* __restrict__ tb, const float* __restrict__ h_lower,
                                         const float* __restrict__ node_feat, const int32_t* __restrict__ lower_ids, const int32_t* __restrict__ lower_map,
                                         int64_t i0, int64_t nl, int Fn, int Dq, int wave, int lane) {
    constexpr int RW = 2 * MT;
    const float* hsrc[RW];
    bool valid[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int64_t i = i0 + wave + kWaves * r;
        valid[r] = i < nl;
        int64_t idx = 0;
        if (valid[r]) idx = h_lower ? (lower_map ? (int64_t)lower_map[i] : i) : (int64_t)lower_ids[i];
        hsrc[r] = (h_lower ? h_lower : node_feat) + idx * Fn;
    }
    for (int f = lane; f < ldq; f += 64) {
        if (f < Fn) {
            float v[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) v[r] = hsrc[r][f];
#pragma unroll
            for (int r = 0; r < RW; ++r) qin[(wave + kWaves * r) * ldq + f] = valid[r] ? v[r] : 0.f;
        } else {
            const float v = f < Dq ? cosf(fmaf(0.0f, tw[f - Fn], tb[f - Fn])) : 0.f;      // the query's time feature: dt = 0 (models/TGAT.py:84)
#pragma unroll
            for (int r = 0; r < RW; ++r) qin[(wave + kWaves * r) * ldq + f] = valid[r] ? v : 0.f;
        }
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
    const int ldq = pad_ld(r16(Dq));
    float* qin = lds;
    float* q = qin + R * ldq;
    const f4* pk = reinterpret_cast<const f4*>(a.pk);
    fill_qin<MT>(qin, ldq, a.tw, a.tb, a.h_lower, a.node_feat, a.lower_ids, a.lower_map, i0, nl, Fn, Dq, wave, lane);
    __syncthreads();
    // q = W_q q_in (bias-free, models/modules.py:126)
    stream_kc<MT>(pk + (size_t)a.off_q * 64, (Dq + 15) >> 4, (Dq + 15) >> 4, wave, lane, ldq, [&](int) { return (const float*)qin; }, [&](int t, const f4 (&acc)[MT]) {
        const int n = 16 * t + 4 * g;
        if (n < Dq) {
#pragma unroll
            for (int j = 0; j < MT; ++j) *reinterpret_cast<f4*>(q + (16 * j + c) * ldq + n) = acc[j];
        }
    });
    __syncthreads();
    // qk[i][h][:] = W_k,h^T q_ih: rows h*hd .. of key_w [Dq][Dkv] are the contraction index; tile = (head, 64-wide block of Dkv)
    const int nblk = (Dkv + 63) >> 6;
    stream_km<MT>(pk + (size_t)a.off_k * 64, H * nblk, (hd + 3) >> 2, wave, lane, ldq, [&](int t) { return (const float*)q + (t / nblk) * hd; },
                  [&](int t, const f4 (&acc)[4][MT]) {
        const int h = t / nblk, nb = (t - h * nblk) * 64;
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
    });
}

struct PostLds { int ldz, ldq, ldm, ldh, r0, total; };
__host__ __device__ inline PostLds post_layout(int R, int Fn, int Ft, int Dkv, int H) {
    PostLds y;
    const int Dq = Fn + Ft;
    y.ldz = pad_ld((H - 1) * Dkv + r16(Dkv));
    y.ldq = pad_ld(r16(Dq));
    y.ldm = pad_ld(r16(Dq + Fn));
    y.ldh = pad_ld(r16(Fn));
    const int a = R * y.ldz, b = R * (y.ldq + y.ldm);
    y.r0 = a > b ? a : b;                       // region 0: z rows, later (z is dead after the W_v product) q_in rows | MergeLayer input rows
    const int r1 = R * (y.ldq > y.ldh ? y.ldq : y.ldh);      // region 1: att rows, later hid rows
    y.total = y.r0 + r1;
    return y;
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
    const PostLds L = post_layout(R, Fn, Ft, Dkv, H);
    const int ldz = L.ldz, ldq = L.ldq, ldm = L.ldm, ldh = L.ldh;
    float* zb = lds;
    float* qin = lds;
    float* mrg = lds + R * ldq;
    float* att = lds + L.r0;
    float* hid = att;
    const f4* pk = reinterpret_cast<const f4*>(a.pk);
    {   // z rows of the block (float4, coalesced); rows beyond the live count and the padding columns are zero
        const int z4 = (H * Dkv) >> 2, l4 = ldz >> 2;
        for (int rr = wave; rr < R; rr += kWaves) {
            const int64_t i = i0 + rr;
            const f4* src = reinterpret_cast<const f4*>(a.z + (size_t)(i < nl ? i : 0) * H * Dkv);
            for (int x = lane; x < l4; x += 64)
                *reinterpret_cast<f4*>(zb + rr * ldz + 4 * x) = (i < nl && x < z4) ? src[x] : f4{0.f, 0.f, 0.f, 0.f};
            for (int f = Dq + lane; f < ldq; f += 64) att[rr * ldq + f] = 0.f;      // the att rows' padding columns
        }
    }
    __syncthreads();
    // att[i][h*hd + e] = W_v,h z_ih (value_w [Dq][Dkv], bias-free); tile = (head, 16 rows of the head)
    const int nth = (hd + 15) >> 4;
    stream_kc<MT>(pk + (size_t)a.off_v * 64, H * nth, (Dkv + 15) >> 4, wave, lane, ldz, [&](int t) { return (const float*)zb + (t / nth) * Dkv; },
                  [&](int t, const f4 (&acc)[MT]) {
        const int h = t / nth, n = (t - h * nth) * 16 + 4 * g;
        if (n < hd) {
#pragma unroll
            for (int j = 0; j < MT; ++j) *reinterpret_cast<f4*>(att + (16 * j + c) * ldq + h * hd + n) = acc[j];
        }
    });
    __syncthreads();
    fill_qin<MT>(qin, ldq, a.tw, a.tb, a.h_lower, a.node_feat, a.lower_ids, a.lower_map, i0, nl, Fn, Dq, wave, lane);      // the residual (models/modules.py:150, :196)
    __syncthreads();
    // x = residual_fc(att) + q_in, into the MergeLayer input rows (normalised in place below)
    stream_kc<MT>(pk + (size_t)a.off_r * 64, (Dq + 15) >> 4, (Dq + 15) >> 4, wave, lane, ldq, [&](int) { return (const float*)att; }, [&](int t, const f4 (&acc)[MT]) {
        const int n = 16 * t + 4 * g;
        if (n < Dq) {
            const f4 b = *reinterpret_cast<const f4*>(a.res_b + n);
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const f4 r = *reinterpret_cast<const f4*>(qin + (16 * j + c) * ldq + n);
                *reinterpret_cast<f4*>(mrg + (16 * j + c) * ldm + n) = f4{acc[j].x + b.x + r.x, acc[j].y + b.y + r.y, acc[j].z + b.z + r.z, acc[j].w + b.w + r.w};
            }
        }
    });
    __syncthreads();
    // LayerNorm (eps 1e-5) per row; the raw node features fill the rest of the MergeLayer input (models/TGAT.py:134, models/modules.py:64)
    for (int rr = wave; rr < R; rr += kWaves) {
        const int64_t i = i0 + rr;
        float* row = mrg + rr * ldm;
        const float* raw = a.node_feat + (size_t)(i < nl ? a.lower_ids[i] : 0) * Fn;
        float rv[3];
#pragma unroll
        for (int x = 0; x < 3; ++x) rv[x] = (lane + 64 * x < Fn) ? raw[lane + 64 * x] : 0.f;      // in flight during the reductions
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
        if (Fn <= 192) {
#pragma unroll
            for (int x = 0; x < 3; ++x)
                if (lane + 64 * x < ldm - Dq) row[Dq + lane + 64 * x] = i < nl ? rv[x] : 0.f;
        } else {
            for (int f = lane; f < ldm - Dq; f += 64) row[Dq + f] = (i < nl && f < Fn) ? raw[f] : 0.f;
        }
        for (int f = Fn + lane; f < ldh; f += 64) hid[rr * ldh + f] = 0.f;      // the hid rows' padding columns (att is dead)
    }
    __syncthreads();
    // hid = relu(fc1 [y | raw] + b1)
    const int ntf = (Fn + 15) >> 4;
    stream_kc<MT>(pk + (size_t)a.off_f1 * 64, ntf, (Dm + 15) >> 4, wave, lane, ldm, [&](int) { return (const float*)mrg; }, [&](int t, const f4 (&acc)[MT]) {
        const int n = 16 * t + 4 * g;
        if (n < Fn) {
            const f4 b = *reinterpret_cast<const f4*>(a.fc1_b + n);
#pragma unroll
            for (int j = 0; j < MT; ++j)
                *reinterpret_cast<f4*>(hid + (16 * j + c) * ldh + n) = f4{fmaxf(acc[j].x + b.x, 0.f), fmaxf(acc[j].y + b.y, 0.f), fmaxf(acc[j].z + b.z, 0.f), fmaxf(acc[j].w + b.w, 0.f)};
        }
    });
    __syncthreads();
    // out = fc2 hid + b2
    stream_kc<MT>(pk + (size_t)a.off_f2 * 64, ntf, (Fn + 15) >> 4, wave, lane, ldh, [&](int) { return (const float*)hid; }, [&](int t, const f4 (&acc)[MT]) {
        const int n = 16 * t + 4 * g;
        if (n < Fn) {
            const f4 b = *reinterpret_cast<const f4*>(a.fc2_b + n);
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int64_t i = i0 + 16 * j + c;
                if (i < nl) *reinterpret_cast<f4*>(a.out + (size_t)i * Fn + n) = f4{acc[j].x + b.x, acc[j].y + b.y, acc[j].z + b.z, acc[j].w + b.w};
            }
        }
    });
}

// Workgroups 0 .. ceil(count/16)-1: the GRU rows.  The workgroups behind them: feat0 = memory + raw for the nodes of the call WITHOUT a
// pending message (list2), one row per wave and round -- independent of the GRU rows, so it rides in the same launch.
__global__ __launch_bounds__(kThreads) void k_tgn_gru_chain(const GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int64_t cnt = *a.count;
    const int64_t ngru = (cnt + 15) >> 4;
    const int Dm = a.Dm, Fn = a.Fn, G = 3 * Fn;
    if ((int64_t)blockIdx.x >= ngru) {
        const int64_t cnt2 = *a.count2, nb = (int64_t)gridDim.x - ngru, F4 = Fn >> 2;
        for (int64_t r = ((int64_t)blockIdx.x - ngru) * kWaves + wave; r < cnt2; r += nb * kWaves) {
            const int64_t node = a.list2[r];
            for (int x = lane; x < F4; x += 64) {
                const f4 m = *reinterpret_cast<const f4*>(a.M + node * Fn + 4 * x), w = *reinterpret_cast<const f4*>(a.raw + node * Fn + 4 * x);
                *reinterpret_cast<f4*>(a.feat0 + node * Fn + 4 * x) = f4{m.x + w.x, m.y + w.y, m.z + w.z, m.w + w.w};      // MemoryModel.py:609
            }
        }
        return;
    }
    const int64_t r0 = (int64_t)blockIdx.x * 16;
    const int ldm = pad_ld(r16(Dm)), ldh = pad_ld(r16(Fn)), ldg = pad_ld(G);
    float* am = lds;                  // [16][ldm] aggregated (= last) message rows
    float* ah = am + 16 * ldm;        // [16][ldh] memory rows
    float* gi = ah + 16 * ldh;        // [16][ldg] W_ih m + b_ih
    float* gh = gi + 16 * ldg;        // [16][ldg] W_hh h + b_hh
    const f4* pk = reinterpret_cast<const f4*>(a.pk);
    {
        int64_t node[2]; bool valid[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) { valid[r] = r0 + wave + 8 * r < cnt; node[r] = valid[r] ? a.list[r0 + wave + 8 * r] : 0; }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int rr = wave + 8 * r;
            for (int f = lane; f < ldm; f += 64) am[rr * ldm + f] = (valid[r] && f < Dm) ? a.msg[node[r] * Dm + f] : 0.f;
            for (int f = lane; f < ldh; f += 64) ah[rr * ldh + f] = (valid[r] && f < Fn) ? a.M[node[r] * Fn + f] : 0.f;
        }
    }
    __syncthreads();
    const int nt = (G + 15) >> 4;
    stream_kc<1>(pk + (size_t)a.off_ih * 64, nt, (Dm + 15) >> 4, wave, lane, ldm, [&](int) { return (const float*)am; }, [&](int t, const f4 (&acc)[1]) {
        const int n = 16 * t + 4 * g;
        if (n < G) {
            const f4 b = *reinterpret_cast<const f4*>(a.b_ih + n);
            *reinterpret_cast<f4*>(gi + c * ldg + n) = f4{acc[0].x + b.x, acc[0].y + b.y, acc[0].z + b.z, acc[0].w + b.w};
        }
    });
    stream_kc<1>(pk + (size_t)a.off_hh * 64, nt, (Fn + 15) >> 4, kWaves - 1 - wave, lane, ldh, [&](int) { return (const float*)ah; },      // waves in reverse: evens the odd tile out
                 [&](int t, const f4 (&acc)[1]) {
        const int n = 16 * t + 4 * g;
        if (n < G) {
            const f4 b = *reinterpret_cast<const f4*>(a.b_hh + n);
            *reinterpret_cast<f4*>(gh + c * ldg + n) = f4{acc[0].x + b.x, acc[0].y + b.y, acc[0].z + b.z, acc[0].w + b.w};
        }
    });
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

static size_t pre_lds(int Fn, int Ft, int MT) { return (size_t)2 * 16 * MT * pad_ld(r16(Fn + Ft)) * sizeof(float); }
static size_t post_lds(int Fn, int Ft, int Dkv, int H, int MT) { return (size_t)post_layout(16 * MT, Fn, Ft, Dkv, H).total * sizeof(float); }
static size_t gru_lds(const GruArgs& a) { return (size_t)16 * (pad_ld(r16(a.Dm)) + pad_ld(r16(a.Fn)) + 2 * pad_ld(3 * a.Fn)) * sizeof(float); }
constexpr size_t kLdsMax = 160 * 1024;

template <class K>
static int set_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return DYGNN_OK;
}
static int pick_mt(int64_t n, size_t lds2) { return (n > 4096 && lds2 <= kLdsMax) ? 2 : 1; }

bool fits(int Fn, int Ft, int Dkv, int H) {
    if (H < 1 || (Fn + Ft) % H || Fn % 4 || Ft % 4 || Dkv % 4 || ((Fn + Ft) / H) % 4) return false;
    return pre_lds(Fn, Ft, 1) <= kLdsMax && post_lds(Fn, Ft, Dkv, H, 1) <= kLdsMax;
}

int launch_pre(hipStream_t s, const PreArgs& a) {
    if (a.n == 0) return DYGNN_OK;
    DYGNN_REQUIRE(fits(a.Fn, a.Ft, a.Dkv, a.H), "tgat chain: feature dims do not fit the row-block kernels");
    const int MT = pick_mt(a.n, pre_lds(a.Fn, a.Ft, 2));
    const size_t bytes = pre_lds(a.Fn, a.Ft, MT);
    const dim3 grid((unsigned)ceil_div(a.n, 16 * MT));
    if (MT == 2) { if (int rc = set_lds(k_tgat_pre<2>, bytes)) return rc; hipLaunchKernelGGL(k_tgat_pre<2>, grid, dim3(kThreads), bytes, s, a); }
    else { if (int rc = set_lds(k_tgat_pre<1>, bytes)) return rc; hipLaunchKernelGGL(k_tgat_pre<1>, grid, dim3(kThreads), bytes, s, a); }
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

int launch_post(hipStream_t s, const PostArgs& a) {
    if (a.n == 0) return DYGNN_OK;
    DYGNN_REQUIRE(fits(a.Fn, a.Ft, a.Dkv, a.H), "tgat chain: feature dims do not fit the row-block kernels");
    const int MT = pick_mt(a.n, post_lds(a.Fn, a.Ft, a.Dkv, a.H, 2));
    const size_t bytes = post_lds(a.Fn, a.Ft, a.Dkv, a.H, MT);
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
    hipLaunchKernelGGL(k_tgn_gru_chain, dim3((unsigned)ceil_div(a.max_rows, 16) + 8), dim3(kThreads), bytes, s, a);      // + 8: always some workgroups for the plain rows
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

}  // namespace chain
}  // namespace dygnn
