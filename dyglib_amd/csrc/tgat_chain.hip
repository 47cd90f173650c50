// Row-block chains of the TGAT / TGN layer (see tgat_chain.h).  gfx950 only.
//
// One workgroup (8 wave64) owns R = 4*MT rows of a level (R = 4 .. 32, chosen from the level size so that small levels still spread over
// the chip).  Every product of the chain is Out[R][N] = Act[R][K] . W^T with the activations in LDS and the weight operand streamed from
// the packed buffer (a layer is 2 MB: Infinity Cache / L2 traffic, re-streamed by every workgroup) straight into the MFMA A operand of `v_mfma_f32_4x4x1_16b_f32`: sixteen independent
// 4x4 outer products per instruction, used as 4 groups of n (4 outputs each) x 4 slices of k against FOUR rows -- full matrix-core rate
// at four rows per workgroup, where a 16x16x4 tile would need sixteen (the first version of these kernels used 16-row tiles: a TGN
// step's 800 roots were 50 workgroups on 256 CUs, each bound by its own CU's matrix pipe and L1; profiles/r02_tgn_chain_notes.md).
//  * fragment (tile, chunk) = 16 outputs n x 16 k = 1 KiB: lane L = 16 ng + 4 ks + i holds the float4 W[16 tile + 4 ng + i][16 chunk + 4 ks ..];
//    instruction t of a step multiplies element t with Act[4m + j][16 chunk + 4 ks + t] (lane's j = L & 3; one b128 LDS read per 4 rows);
//  * accumulator of lane (ng, ks, j): partial sums over k-slice ks of Out[4m + j][16 tile + 4 ng + 0..3]; at the end of a tile the four
//    slices are added in a fixed order (two cross-lane adds) and the ks = 0 lanes hold a float4 of four consecutive n of one row.
// So a row's value is the same sequence of operations for every MT: rows are bit-identical whatever batch or block they sit in.
// Both weight layouts ([N][K] of nn.Linear, and key_w used as W_k,h^T: contraction along weight ROWS) are packed into this one
// fragment form (k_pack), zero-padded to whole tiles / chunks so no step needs a mask; LDS rows are zero-padded to the 16-k chunk.
#include <cstdlib>
#include "tgat_chain.h"
#include "tgat_attn.h"

namespace dygnn {
namespace chain {

using f4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
__host__ __device__ inline int r16(int K) { return (K + 15) & ~15; }
__host__ __device__ inline int pad_ld(int K) {           // smallest row stride >= K with stride % 32 == 16: the four rows of an operand read fall on distinct banks
    const int l = r16(K);
    return (l & 16) ? l : l + 16;
}

// ---- packing ----------------------------------------------------------------------------------------------------------------------
struct PackDesc {
    const float* src;
    int ld, N, K, type;            // type 0: src[n][k] (nn.Linear) ; 1: src[k][n] (used transposed); tile = 16 n, step = 16 k either way
    int tph;                       // tiles per head (per-head products: tile t belongs to head t / tph, matrix src + head * hstride)
    int64_t hstride;
    uint32_t off, ntiles, nsteps;
};
constexpr int kMaxDesc = 6 * DYGNN_MAX_LAYERS + 2;
struct PackTable { PackDesc d[kMaxDesc]; int n; uint32_t total; };

__global__ __launch_bounds__(256) void k_pack(const PackTable tb, f4* __restrict__ dst, const ListArgs la, unsigned list_blocks) {
    if (blockIdx.x < list_blocks) {       // ---- the call's node lists (TGN), see ListArgs
        __shared__ int32_t s_cnt[2], s_base[2];
        if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        const int lane = threadIdx.x & 63;
        const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
        const int64_t q = slot / (la.k + 1);
        const int jj = (int)(slot - q * (la.k + 1));
        const int64_t nl = la.n_live ? (int64_t)*la.n_live : la.n;
        int32_t id = -1;
        if (q < la.n && q < nl) id = jj == la.k ? la.ids0[q] : la.ids0[la.n + q * la.k + jj];
        const bool first = id >= 0 && id < la.N && la.owner[id] == (int32_t)slot;
        const bool pend = first && la.has_msg[id] != 0;
        if (first) la.pendf[id] = pend ? 1 : 0;
        const uint64_t below = (1ull << lane) - 1;
        const uint64_t pm = __ballot(pend), qm = __ballot(first && !pend);
        int32_t o1 = 0, o2 = 0;
        if (lane == 0) {
            if (pm) o1 = atomicAdd(&s_cnt[0], __popcll(pm));
            if (qm) o2 = atomicAdd(&s_cnt[1], __popcll(qm));
        }
        o1 = __shfl(o1, 0, 64);
        o2 = __shfl(o2, 0, 64);
        __syncthreads();
        if (threadIdx.x == 0) {
            s_base[0] = s_cnt[0] ? atomicAdd(la.count, s_cnt[0]) : 0;
            s_base[1] = s_cnt[1] ? atomicAdd(la.count2, s_cnt[1]) : 0;
        }
        __syncthreads();
        if (pend) la.list[s_base[0] + o1 + __popcll(pm & below)] = id;
        if (first && !pend) la.list2[s_base[1] + o2 + __popcll(qm & below)] = id;
        return;
    }
    const uint32_t frag = (blockIdx.x - list_blocks) * 4 + (threadIdx.x >> 6);
    if (frag >= tb.total) return;
    const int lane = threadIdx.x & 63;
    int di = 0;
    for (int i = 1; i < tb.n; ++i)
        if (frag >= tb.d[i].off) di = i;
    const PackDesc& d = tb.d[di];
    const uint32_t rel = frag - d.off, t = rel / d.nsteps, st = rel - t * d.nsteps;
    const int h = (int)t / d.tph, lt = (int)t - h * d.tph;
    const float* m = d.src + (int64_t)h * d.hstride;
    f4 v = f4{0.f, 0.f, 0.f, 0.f};
    const int n = 16 * lt + 4 * (lane >> 4) + (lane & 3), k = 16 * (int)st + 4 * ((lane >> 2) & 3);
    if (n < d.N && k < d.K) {
        if (d.type == 0) v = *reinterpret_cast<const f4*>(m + (size_t)n * d.ld + k);               // W[n][k ..]       (K % 4 == 0)
        else {                                                                                       // W[k ..][n]: the contraction runs along weight rows
            v.x = m[(size_t)k * d.ld + n];
            if (k + 1 < d.K) v.y = m[(size_t)(k + 1) * d.ld + n];
            if (k + 2 < d.K) v.z = m[(size_t)(k + 2) * d.ld + n];
            if (k + 3 < d.K) v.w = m[(size_t)(k + 3) * d.ld + n];
        }
    }
    dst[(size_t)frag * 64 + lane] = v;
}

static uint32_t kc_frags(int N, int K, int heads = 1) { return (uint32_t)heads * ((N + 15) / 16) * ((K + 15) / 16); }

PackPlan plan_pack(int L, int Fn, int Ft, int Dkv, int H, int gru_Dm) {
    PackPlan p{};
    const int Dq = Fn + Ft, hd = Dq / H;
    uint32_t o = 0;
    for (int l = 0; l < L; ++l) {
        LayerPack& y = p.layer[l];
        y.q = o; o += kc_frags(hd, Dq, H);
        y.k = o; o += kc_frags(Dkv, hd, H);
        y.v = o; o += kc_frags(hd, Dkv, H);
        y.r = o; o += kc_frags(Dq, Dq);
        y.f1 = o; o += kc_frags(Fn, Dq + Fn);
        y.f2 = o; o += kc_frags(Fn, Fn);
    }
    if (gru_Dm > 0) {
        p.ih = o; o += kc_frags(Fn, gru_Dm, 3);      // per gate: a slice of memory dims takes the same tiles of every gate
        p.hh = o; o += kc_frags(Fn, Fn, 3);
    }
    p.total = o;
    return p;
}

int pack(hipStream_t s, const PackPlan& p, int L, int Fn, int Ft, int Dkv, int H, const dygnn_tgat_weights* w, const dygnn_gru_weights* gru, int gru_Dm,
         float* dst, const ListArgs* lists) {
    PackTable tb{};
    const int Dq = Fn + Ft, hd = Dq / H;
    auto kc = [&](const float* src, int ld, int N, int K, uint32_t off, int heads = 1, int64_t hstride = 0) {
        const int tph = (N + 15) / 16;
        tb.d[tb.n++] = PackDesc{src, ld, N, K, 0, tph, hstride, off, (uint32_t)(heads * tph), (uint32_t)((K + 15) / 16)};
    };
    for (int l = 0; l < L; ++l) {
        const dygnn_tgat_layer_weights& Lw = w->layers[l];
        const LayerPack& y = p.layer[l];
        kc(Lw.query_w, Dq, hd, Dq, y.q, H, (int64_t)hd * Dq);
        tb.d[tb.n++] = PackDesc{Lw.key_w, Dkv, Dkv, hd, 1, (Dkv + 15) / 16, (int64_t)hd * Dkv, y.k, (uint32_t)(H * ((Dkv + 15) / 16)), (uint32_t)((hd + 15) / 16)};
        kc(Lw.value_w, Dkv, hd, Dkv, y.v, H, (int64_t)hd * Dkv);
        kc(Lw.res_w, Dq, Dq, Dq, y.r);
        kc(Lw.fc1_w, Dq + Fn, Fn, Dq + Fn, y.f1);
        kc(Lw.fc2_w, Fn, Fn, Fn, y.f2);
    }
    if (gru_Dm > 0) {
        kc(gru->weight_ih, gru_Dm, Fn, gru_Dm, p.ih, 3, (int64_t)Fn * gru_Dm);
        kc(gru->weight_hh, Fn, Fn, Fn, p.hh, 3, (int64_t)Fn * Fn);
    }
    tb.total = gru_Dm > 0 ? p.total : p.ih ? p.ih : p.total;      // (a TGAT call leaves the GRU part of the plan alone)
    const unsigned lb = lists ? (unsigned)ceil_div(lists->n * (lists->k + 1), 256) : 0;
    hipLaunchKernelGGL(k_pack, dim3(lb + (tb.total + 3) / 4), dim3(256), 0, s, tb, reinterpret_cast<f4*>(dst), lists ? *lists : ListArgs{}, lb);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

// ---- the weight stream of a wave ------------------------------------------------------------------------------------------------
// A stage gives wave w the tiles t = w, w + 8, ... < T.  Their fragments form ONE sequence of steps (tile by tile, chunk by chunk) that
// runs through a ring of U float4 registers filled U steps ahead -- across tile boundaries, so the stream only drains at the end of a
// stage.  Steady state is branch-free but for the tile epilogue (refills are always issued; beyond the stream they re-read the wave's
// first tile), so the compiler counts the loads in flight (s_waitcnt vmcnt(U-1)) instead of draining them.  The LDS operands of step
// s+1 are read before the MFMAs of step s.
// AF: act(t) = LDS base of the activation operand of tile t.  Ep(t, acc): the finished tile -- every lane holds the sums of
// Out[4m + j][16t + 4ng .. +3] in acc[m]; the ks == 0 lanes store.
#ifndef DYGNN_CHAIN_U
#define DYGNN_CHAIN_U 4      // ring depth: the chains run at the rate a CU streams fragments through its vector L1 (~35 B/clk measured), 4 .. 16 deep gave the same time per step; 4 is the least code
#endif
constexpr int U = DYGNN_CHAIN_U;

struct IdTile { __device__ __forceinline__ int operator()(int t) const { return t; } };
// TM: local tile index -> tile index in the packed stage (a workgroup that owns a slice of a stage's tiles).
// start() issues the first U fragment loads -- they depend on nothing the kernel computes, so a stage's ring is started BEFORE the
// previous stage runs (its loads are then older than that stage's refills and have landed when it ends); run() consumes.
template <class TM>
struct WStream {
    const f4* pk;
    int T, nch, wave, lane, pt, pch, adv;
    const f4* pp;
    TM tmap;
    f4 ring[U];
    __device__ __forceinline__ WStream(const f4* pk_, int T_, int nch_, int wave_, int lane_, TM tm) : pk(pk_), T(T_), nch(nch_), wave(wave_), lane(lane_), pt(wave_), pch(0), tmap(tm) {
        adv = wave < T ? 64 : 0;
        pp = wave < T ? pk + (size_t)tmap(wave) * nch * 64 + lane : pk;
#pragma unroll
        for (int u = 0; u < U; ++u) fetch(ring[u]);
    }
    // Behind the wave's last fragment every lane re-reads ONE address (a single 16-byte request): the refill stays unconditional -- the
    // compiler keeps counting the loads in flight -- without fetching a junk KiB per step (U junk fragments per stage and wave were 10-70 %
    // of these short streams' traffic).
    __device__ __forceinline__ void fetch(f4& dst) {
        dst = *pp;
        pp += adv;
        if (++pch == nch) {
            pch = 0; pt += kWaves;
            if (pt < T) pp = pk + (size_t)tmap(pt) * nch * 64 + lane;
            else { pp = pk; adv = 0; }
        }
    }
    template <int MT, class AF, class Ep>
    __device__ __forceinline__ void run(int lda, AF actf, Ep ep) {
        if (wave >= T) return;
        const int j = lane & 3, ks = (lane >> 2) & 3;
        const int S = ((T - wave + kWaves - 1) / kWaves) * nch;
        const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
        int ct = wave, cch = 0;                                           // consume cursor
        const float* ab = actf(ct) + j * lda + 4 * ks;
        f4 acc[MT], bn[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) { acc[m] = zero; bn[m] = *reinterpret_cast<const f4*>(ab + m * 4 * lda); }
        auto step = [&](f4& slot, bool refill) {
            const f4 w = slot;
            f4 b[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) b[m] = bn[m];
            const bool tile_end = cch + 1 == nch;
            const int nt = tile_end ? ct + kWaves : ct, nc = tile_end ? 0 : cch + 1;
            const float* nab = tile_end ? actf(nt < T ? nt : ct) + j * lda + 4 * ks : ab;
#pragma unroll
            for (int m = 0; m < MT; ++m) bn[m] = *reinterpret_cast<const f4*>(nab + m * 4 * lda + 16 * nc);      // (behind the last step: a valid, unused read)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = mfma4(w.x, b[m].x, acc[m]);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = mfma4(w.y, b[m].y, acc[m]);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = mfma4(w.z, b[m].z, acc[m]);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = mfma4(w.w, b[m].w, acc[m]);
            if (refill) fetch(slot);
            if (tile_end) {
                f4 r[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {          // k-slices 0 + 1, 2 + 3, then the pairs: fixed order
                    f4 v = acc[m];                      // rotations inside the row of 16 lanes (DPP row_ror 4, then 8): no LDS
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = v[e];
                        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));
                        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));
                        v[e] = x;
                    }
                    r[m] = v;
                    acc[m] = zero;
                }
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
};
template <class TM>
__device__ __forceinline__ WStream<TM> wstream(const f4* pk, int T, int nch, int wave, int lane, TM tm) { return WStream<TM>(pk, T, nch, wave, lane, tm); }
__device__ __forceinline__ WStream<IdTile> wstream(const f4* pk, int T, int nch, int wave, int lane) { return WStream<IdTile>(pk, T, nch, wave, lane, IdTile()); }

// query-input rows [h(self) | cos(w*0 + b)] of the block's rows (zero rows beyond the live count; zero padding columns).
// A wave owns rows wave, wave + 8, ...: load() takes their indices, then their feature rows (and, for k_tgat_post, the raw feature rows
// of the MergeLayer input) into registers -- issued at the start of a kernel, two dependent round trips that overlap whatever comes
// next; store() / store_raw() write them to LDS once the buffer is free.
template <int MT, int NQ, bool RAW>
struct RowRegs {
    static constexpr int R = 4 * MT, RW = (R + kWaves - 1) / kWaves, NR = RAW ? NQ : 1;
    float v[RW][NQ], rv[RW][NR];
    bool valid[RW], have[RW];
    __device__ __forceinline__ void load(const float* __restrict__ tw, const float* __restrict__ tb, const float* __restrict__ h_lower,
                                         const float* __restrict__ node_feat, const int32_t* __restrict__ lower_ids, const int32_t* __restrict__ lower_map,
                                         int64_t i0, int64_t nl, int Fn, int Dq, int wave, int lane) {
        const float* hsrc[RW];
        const float* raw[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int rr = wave + kWaves * r;
            const int64_t i = i0 + rr;
            have[r] = rr < R;
            valid[r] = have[r] && i < nl;
            int64_t idx = 0, id = 0;
            if (valid[r]) {
                id = (!h_lower || RAW) ? (int64_t)lower_ids[i] : 0;
                idx = h_lower ? (lower_map ? (int64_t)lower_map[i] : i) : id;
            }
            hsrc[r] = (h_lower ? h_lower : node_feat) + idx * Fn;
            raw[r] = node_feat + id * Fn;
        }
#pragma unroll
        for (int x = 0; x < NQ; ++x) {
            const int f = lane + 64 * x;
            if (f < Fn) {
#pragma unroll
                for (int r = 0; r < RW; ++r) v[r][x] = hsrc[r][f];
            } else {
                const float c = f < Dq ? cosf(fmaf(0.0f, tw[f - Fn], tb[f - Fn])) : 0.f;      // the query's time feature: dt = 0 (models/TGAT.py:84)
#pragma unroll
                for (int r = 0; r < RW; ++r) v[r][x] = c;
            }
            if (RAW) {
#pragma unroll
                for (int r = 0; r < RW; ++r) rv[r][x] = f < Fn ? raw[r][f] : 0.f;
            }
        }
    }
    __device__ __forceinline__ void store(float* qin, int ldq, int wave, int lane) const {
#pragma unroll
        for (int x = 0; x < NQ; ++x) {
            const int f = lane + 64 * x;
            if (f < ldq) {
#pragma unroll
                for (int r = 0; r < RW; ++r)
                    if (have[r]) qin[(wave + kWaves * r) * ldq + f] = valid[r] ? v[r][x] : 0.f;
            }
        }
    }
    // the raw features behind the Dq LayerNorm outputs of the MergeLayer input rows, zero up to the row stride
    __device__ __forceinline__ void store_raw(float* mrg, int ldm, int Dq, int wave, int lane) const {
#pragma unroll
        for (int x = 0; x < NR; ++x) {
            const int f = lane + 64 * x;
            if (Dq + f < ldm) {
#pragma unroll
                for (int r = 0; r < RW; ++r)
                    if (have[r]) mrg[(wave + kWaves * r) * ldm + Dq + f] = valid[r] ? rv[r][x] : 0.f;
            }
        }
    }
};
constexpr int kNQ = 5;      // column slots of 64 per row held in registers: row strides up to 320 floats (fits() checks)

// grid = (row blocks, heads): a workgroup computes ONE head's q_h = W_q,h q_in and qk_h = W_k,h^T q_h (the heads share nothing but q_in,
// which each re-reads), so a row block's weight stream is split over H CUs
template <int MT>
__global__ __launch_bounds__(kThreads) void k_tgat_pre(const PreArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 3, ks = (lane >> 2) & 3, ng = lane >> 4;
    constexpr int R = 4 * MT;
    const int64_t i0 = (int64_t)blockIdx.x * R;
    const int h = blockIdx.y;
    const int64_t nl = a.n_live ? (int64_t)*a.n_live : a.n;
    if (i0 >= nl) return;
    const int Fn = a.Fn, Ft = a.Ft, Dq = Fn + Ft, Dkv = a.Dkv, H = a.H, hd = Dq / H;
    const int ldq = pad_ld(Dq), ldh = pad_ld(hd);
    float* qin = lds;
    float* q = qin + R * ldq;                 // [R][ldh]: this head's q, zero beyond hd (zero-packed weight rows)
    const f4* pk = reinterpret_cast<const f4*>(a.pk);
    const int ntq = (hd + 15) >> 4, ntk = (Dkv + 15) >> 4;
    auto sq = wstream(pk + (size_t)a.off_q * 64, ntq, (Dq + 15) >> 4, wave, lane, [=](int t) { return h * ntq + t; });
    {
        RowRegs<MT, kNQ, false> rows;
        rows.load(a.tw, a.tb, a.h_lower, a.node_feat, a.lower_ids, a.lower_map, i0, nl, Fn, Dq, wave, lane);
        rows.store(qin, ldq, wave, lane);
    }
    __syncthreads();
    auto sk = wstream(pk + (size_t)a.off_k * 64, ntk, (hd + 15) >> 4, wave, lane, [=](int t) { return h * ntk + t; });
    // q_h = W_q,h q_in (bias-free, models/modules.py:126); tile = 16 rows of the head
    sq.template run<MT>(ldq, [&](int) { return (const float*)qin; }, [&](int t, const f4 (&acc)[MT]) {
        const int n = 16 * t + 4 * ng;
        if (ks == 0 && n < ldh) {
#pragma unroll
            for (int m = 0; m < MT; ++m) *reinterpret_cast<f4*>(q + (4 * m + j) * ldh + n) = acc[m];
        }
    });
    __syncthreads();
    // qk[i][h][:] = W_k,h^T q_ih: rows h*hd .. of key_w [Dq][Dkv] are the contraction index; tile = 16 columns of Dkv
    sk.template run<MT>(ldh, [&](int) { return (const float*)q; }, [&](int t, const f4 (&acc)[MT]) {
        const int n = 16 * t + 4 * ng;
        if (ks == 0 && n < Dkv) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int64_t i = i0 + 4 * m + j;
                if (i < nl) *reinterpret_cast<f4*>(a.qk + ((size_t)i * H + h) * Dkv + n) = acc[m];
            }
        }
    });
}

struct PostLds { int ldz, ldq, ldm, ldh, r0, par, att, total; };
__host__ __device__ inline PostLds post_layout(int R, int Fn, int Ft, int Dkv, int H, int KC = 0) {
    PostLds y;
    const int Dq = Fn + Ft;
    y.ldz = pad_ld((H - 1) * Dkv + r16(Dkv));
    y.ldq = pad_ld(Dq);
    y.ldm = pad_ld(Dq + Fn);
    y.ldh = pad_ld(Fn);
    const int a = R * y.ldz, b = R * (y.ldq + y.ldm);
    y.r0 = a > b ? a : b;                       // region 0: z rows, later (z is dead after the W_v product) q_in rows | MergeLayer input rows
    const int r1 = R * (y.ldq > y.ldh ? y.ldq : y.ldh);      // region 1: att rows, later hid rows
    y.par = y.r0 + r1;                          // then the bias / LayerNorm vectors: res_b | ln_w | ln_b (3 Dq) | fc1_b | fc2_b (2 Fn)
    y.att = y.par + 3 * Dq + 2 * Fn;            // then the staging of the fused attention (tgat_attn.h), KC row slots per lane
    y.total = y.att + (KC ? attn::pair_smem_floats(kWaves, H, KC, Ft) : 0);
    return y;
}

template <int MT, int KC>      // KC > 0: the attention over the rows' k <= KC neighbours runs here (z stays in LDS); 0: z comes from the attention kernel
__global__ __launch_bounds__(kThreads) void k_tgat_post(const PostArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 3, ks = (lane >> 2) & 3, ng = lane >> 4;
    constexpr int R = 4 * MT;
    const int64_t i0 = (int64_t)blockIdx.x * R;
    const int64_t nl = a.n_live ? (int64_t)*a.n_live : a.n;
    if (i0 >= nl) return;
    const int Fn = a.Fn, Ft = a.Ft, Dq = Fn + Ft, Dkv = a.Dkv, H = a.H, hd = Dq / H, Dm = Dq + Fn;
    const PostLds L = post_layout(R, Fn, Ft, Dkv, H, KC);
    const int ldz = L.ldz, ldq = L.ldq, ldm = L.ldm, ldh = L.ldh;
    float* zb = lds;
    float* qin = lds;
    float* mrg = lds + R * ldq;
    float* att = lds + L.r0;
    float* hid = att;
    float* res_b = lds + L.par;
    float* ln_w = res_b + Dq;
    float* ln_b = ln_w + Dq;
    float* fc1_b = ln_b + Dq;
    float* fc2_b = fc1_b + Fn;
    const f4* pk = reinterpret_cast<const f4*>(a.pk);
    const int nth = (hd + 15) >> 4, ntf = (Fn + 15) >> 4;
    int stamp_i = 0;
    auto stamp = [&]() {
        if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + stamp_i] = __builtin_amdgcn_s_memtime();
        ++stamp_i;
    };
    stamp();
    auto sv = wstream(pk + (size_t)a.off_v * 64, H * nth, (Dkv + 15) >> 4, wave, lane);
    RowRegs<MT, kNQ, true> rows;      // residual rows and raw feature rows: requested now, used after the W_v product
    rows.load(a.tw, a.tb, a.h_lower, a.node_feat, a.lower_ids, a.lower_map, i0, nl, Fn, Dq, wave, lane);
    for (int f = threadIdx.x; f < Dq; f += kThreads) { res_b[f] = a.res_b[f]; ln_w[f] = a.ln_w[f]; ln_b[f] = a.ln_b[f]; }
    for (int f = threadIdx.x; f < Fn; f += kThreads) { fc1_b[f] = a.fc1_b[f]; fc2_b[f] = a.fc2_b[f]; }
    if constexpr (KC > 0) {
        // z rows of the block computed here: two waves per row, four rows at a time (attn::pair_node has one workgroup barrier inside)
        for (int rr = wave; rr < R; rr += kWaves) {
            for (int f = H * Dkv + lane; f < ldz; f += 64) zb[rr * ldz + f] = 0.f;      // padding columns
            for (int f = Dq + lane; f < ldq; f += 64) att[rr * ldq + f] = 0.f;
        }
#pragma unroll 1
        for (int g4 = 0; g4 < MT; ++g4) {
            const int row = 4 * g4 + (wave >> 1);
            int64_t i = i0 + row;
            const bool live = i < nl;
            if (!live) i = nl - 1;              // a stand-in keeps the barrier uniform; the row's z is zeros
            attn::pair_node<KC, false>(a.qk, a.h_lower, a.node_feat, a.edge_feat, a.lower_ids, a.nbr_eid, a.nbr_dt, a.tw, a.tb, a.n, a.k, Fn, a.Fe, Ft, H, a.scale, a.lower_map,
                                       i, live, true, wave, kWaves, lane, lds + L.att, zb + row * ldz, Dkv);
            if (g4 + 1 < MT) __syncthreads();   // the staging is reused
        }
    } else {
        // z rows of the block (float4, coalesced); rows beyond the live count and the padding columns are zero
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
    stamp();
    auto sr = wstream(pk + (size_t)a.off_r * 64, (Dq + 15) >> 4, (Dq + 15) >> 4, wave, lane);
    // att[i][h*hd + e] = W_v,h z_ih (value_w [Dq][Dkv], bias-free); tile = (head, 16 rows of the head)
    sv.template run<MT>(ldz, [&](int t) { return (const float*)zb + (t / nth) * Dkv; }, [&](int t, const f4 (&acc)[MT]) {
        const int h = t / nth, n = (t - h * nth) * 16 + 4 * ng;
        if (ks == 0 && n < hd) {
#pragma unroll
            for (int m = 0; m < MT; ++m) *reinterpret_cast<f4*>(att + (4 * m + j) * ldq + h * hd + n) = acc[m];
        }
    });
    __syncthreads();
    stamp();
    rows.store(qin, ldq, wave, lane);      // the residual (models/modules.py:150, :196)
    __syncthreads();
    stamp();
    // x = residual_fc(att) + q_in, into the MergeLayer input rows (normalised in place below)
    auto sf1 = wstream(pk + (size_t)a.off_f1 * 64, ntf, (Dm + 15) >> 4, wave, lane);
    sr.template run<MT>(ldq, [&](int) { return (const float*)att; }, [&](int t, const f4 (&acc)[MT]) {
        const int n = 16 * t + 4 * ng;
        if (ks == 0 && n < Dq) {
            const f4 b = *reinterpret_cast<const f4*>(res_b + n);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f4 r = *reinterpret_cast<const f4*>(qin + (4 * m + j) * ldq + n);
                *reinterpret_cast<f4*>(mrg + (4 * m + j) * ldm + n) = f4{acc[m].x + b.x + r.x, acc[m].y + b.y + r.y, acc[m].z + b.z + r.z, acc[m].w + b.w + r.w};
            }
        }
    });
    __syncthreads();
    stamp();
    // LayerNorm (eps 1e-5) per row; the raw node features fill the rest of the MergeLayer input (models/TGAT.py:134, models/modules.py:64)
    for (int rr = wave; rr < R; rr += kWaves) {
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
        for (int f = lane; f < Dq; f += 64) row[f] = (row[f] - mean) * rstd * ln_w[f] + ln_b[f];
        for (int f = Fn + lane; f < ldh; f += 64) hid[rr * ldh + f] = 0.f;      // the hid rows' padding columns (att is dead)
    }
    rows.store_raw(mrg, ldm, Dq, wave, lane);
    __syncthreads();
    stamp();
    // hid = relu(fc1 [y | raw] + b1)
    auto sf2 = wstream(pk + (size_t)a.off_f2 * 64, ntf, (Fn + 15) >> 4, wave, lane);
    sf1.template run<MT>(ldm, [&](int) { return (const float*)mrg; }, [&](int t, const f4 (&acc)[MT]) {
        const int n = 16 * t + 4 * ng;
        if (ks == 0 && n < Fn) {
            const f4 b = *reinterpret_cast<const f4*>(fc1_b + n);
#pragma unroll
            for (int m = 0; m < MT; ++m)
                *reinterpret_cast<f4*>(hid + (4 * m + j) * ldh + n) = f4{fmaxf(acc[m].x + b.x, 0.f), fmaxf(acc[m].y + b.y, 0.f), fmaxf(acc[m].z + b.z, 0.f), fmaxf(acc[m].w + b.w, 0.f)};
        }
    });
    __syncthreads();
    stamp();
    // out = fc2 hid + b2
    sf2.template run<MT>(ldh, [&](int) { return (const float*)hid; }, [&](int t, const f4 (&acc)[MT]) {
        const int n = 16 * t + 4 * ng;
        if (ks == 0 && n < Fn) {
            const f4 b = *reinterpret_cast<const f4*>(fc2_b + n);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int64_t i = i0 + 4 * m + j;
                if (i < nl) *reinterpret_cast<f4*>(a.out + (size_t)i * Fn + n) = f4{acc[m].x + b.x, acc[m].y + b.y, acc[m].z + b.z, acc[m].w + b.w};
            }
        }
    });
    stamp();
}

// grid = (row-block workgroups + kPlainBlocks, slices).  The GRU rows: row blocks of R listed nodes, dealt round-robin to the row-block workgroups; slice s of a row block owns the memory dims
// f in [16 ft0, 16 ft1) of ALL THREE gates (gate g, tile ft = packed tile g*ntf + ft), so the slices of a row block share nothing but the
// gathered rows and a row block's weight stream is split over kGruSlices CUs.  The workgroups behind the row blocks (slice 0 only):
// feat0 = memory + raw for the nodes of the call WITHOUT a pending message (list2), one row per wave and round -- independent of the
// GRU rows, so it rides in the same launch.
constexpr int kGruSlices = 2;
constexpr int kPlainBlocks = 64;      // x 8 waves = 512 plain rows per round
template <int MT>
__global__ __launch_bounds__(kThreads) void k_tgn_gru_chain(const GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 3, ks = (lane >> 2) & 3, ng = lane >> 4;
    constexpr int R = 4 * MT;
    const int64_t cnt = *a.count;
    const int64_t ngru = (cnt + R - 1) / R;
    const int64_t nrow = (int64_t)gridDim.x - kPlainBlocks;      // workgroups 0 .. nrow-1 walk the GRU row blocks, the last kPlainBlocks (and idle ones) the plain rows
    const int Dm = a.Dm, Fn = a.Fn;
    if ((int64_t)blockIdx.x >= (ngru < nrow ? ngru : nrow)) {
        if (blockIdx.y != 0) return;
        const int64_t first = ngru < nrow ? ngru : nrow;
        const int64_t cnt2 = *a.count2, nb = (int64_t)gridDim.x - first, F4 = Fn >> 2;
        for (int64_t r = ((int64_t)blockIdx.x - first) * kWaves + wave; r < cnt2; r += nb * kWaves) {
            const int64_t node = a.list2[r];
            for (int x = lane; x < F4; x += 64) {
                const f4 m = *reinterpret_cast<const f4*>(a.M + node * Fn + 4 * x), w = *reinterpret_cast<const f4*>(a.raw + node * Fn + 4 * x);
                *reinterpret_cast<f4*>(a.feat0 + node * Fn + 4 * x) = f4{m.x + w.x, m.y + w.y, m.z + w.z, m.w + w.w};      // MemoryModel.py:609
            }
        }
        return;
    }
    const int ntf = (Fn + 15) >> 4, per = (ntf + (int)gridDim.y - 1) / (int)gridDim.y;
    const int ft0 = blockIdx.y * per, ft1 = ft0 + per < ntf ? ft0 + per : ntf, nft = ft1 - ft0;
    if (nft <= 0) return;
    const int ldm = pad_ld(Dm), ldh = pad_ld(Fn), ldg = 3 * per * 16;
    float* am = lds;                  // [R][ldm] aggregated (= last) message rows
    float* ah = am + R * ldm;         // [R][ldh] memory rows
    float* gi = ah + R * ldh;         // [R][ldg] (W_ih m)[gate][f of the slice]: column (gate * nft + ft - ft0) * 16 + f % 16
    float* gh = gi + R * ldg;         // [R][ldg] W_hh h
    const f4* pk = reinterpret_cast<const f4*>(a.pk);
    auto tmap = [=](int t) { return (t / nft) * ntf + ft0 + t % nft; };
    const int f_lo = 16 * ft0, f_hi = 16 * ft1 < Fn ? 16 * ft1 : Fn, fw = f_hi - f_lo;
    for (int64_t blk = blockIdx.x; blk < ngru; blk += nrow) {      // (one round unless the list is longer than the grid's row blocks)
        const int64_t r0 = blk * R;
        auto si = wstream(pk + (size_t)a.off_ih * 64, 3 * nft, (Dm + 15) >> 4, wave, lane, tmap);
        {   // the listed rows: a wave's node ids first, then float4 row pieces of all its rows in flight together
            constexpr int RW = (R + kWaves - 1) / kWaves;
            int64_t node[RW]; bool valid[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int rr = wave + kWaves * r;
                valid[r] = rr < R && r0 + rr < cnt;
                node[r] = valid[r] ? a.list[r0 + rr] : 0;
            }
            for (int x = lane; x < (ldm >> 2); x += 64) {
                f4 v[RW];
#pragma unroll
                for (int r = 0; r < RW; ++r) v[r] = (4 * x < Dm) ? *reinterpret_cast<const f4*>(a.msg + node[r] * Dm + 4 * x) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < RW; ++r)
                    if (wave + kWaves * r < R) *reinterpret_cast<f4*>(am + (wave + kWaves * r) * ldm + 4 * x) = valid[r] ? v[r] : f4{0.f, 0.f, 0.f, 0.f};
            }
            for (int x = lane; x < (ldh >> 2); x += 64) {
                f4 v[RW];
#pragma unroll
                for (int r = 0; r < RW; ++r) v[r] = (4 * x < Fn) ? *reinterpret_cast<const f4*>(a.M + node[r] * Fn + 4 * x) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < RW; ++r)
                    if (wave + kWaves * r < R) *reinterpret_cast<f4*>(ah + (wave + kWaves * r) * ldh + 4 * x) = valid[r] ? v[r] : f4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();
        auto sh = wstream(pk + (size_t)a.off_hh * 64, 3 * nft, (Fn + 15) >> 4, kWaves - 1 - wave, lane, tmap);      // waves in reverse: evens the odd tile out
        si.template run<MT>(ldm, [&](int) { return (const float*)am; }, [&](int t, const f4 (&acc)[MT]) {
            if (ks == 0) {
#pragma unroll
                for (int m = 0; m < MT; ++m) *reinterpret_cast<f4*>(gi + (4 * m + j) * ldg + 16 * t + 4 * ng) = acc[m];
            }
        });
        sh.template run<MT>(ldh, [&](int) { return (const float*)ah; }, [&](int t, const f4 (&acc)[MT]) {
            if (ks == 0) {
#pragma unroll
                for (int m = 0; m < MT; ++m) *reinterpret_cast<f4*>(gh + (4 * m + j) * ldg + 16 * t + 4 * ng) = acc[m];
            }
        });
        __syncthreads();
        // nn.GRUCell gates (r, z, n order) -> new memory; feat0 = new memory + raw features (MemoryModel.py:609)
        for (int idx = threadIdx.x; idx < R * fw; idx += kThreads) {
            const int rr = idx / fw, fl = idx - rr * fw, f = f_lo + fl;
            if (r0 + rr >= cnt) continue;
            const int64_t node = a.list[r0 + rr];
            const float* x = gi + rr * ldg;
            const float* y = gh + rr * ldg;
            const int c0 = fl, c1 = nft * 16 + fl, c2 = 2 * nft * 16 + fl;
            const float h = ah[rr * ldh + f];
            const float xr = x[c0] + a.b_ih[f], yr = y[c0] + a.b_hh[f];
            const float xz = x[c1] + a.b_ih[Fn + f], yz = y[c1] + a.b_hh[Fn + f];
            const float xn = x[c2] + a.b_ih[2 * Fn + f], yn = y[c2] + a.b_hh[2 * Fn + f];
            const float rg = 1.0f / (1.0f + expf(-(xr + yr)));
            const float zg = 1.0f / (1.0f + expf(-(xz + yz)));
            const float ng_ = tanhf(xn + rg * yn);
            const float hn = (1.0f - zg) * ng_ + zg * h;
            a.Mnew[node * Fn + f] = hn;
            a.feat0[node * Fn + f] = hn + a.raw[node * Fn + f];
        }
        __syncthreads();
    }
}

static size_t pre_lds(int Fn, int Ft, int H, int MT) { return (size_t)4 * MT * (pad_ld(Fn + Ft) + pad_ld((Fn + Ft) / H)) * sizeof(float); }
static size_t post_lds(int Fn, int Ft, int Dkv, int H, int MT, int KC = 0) { return (size_t)post_layout(4 * MT, Fn, Ft, Dkv, H, KC).total * sizeof(float); }
static int gru_slices() {          // DYGNN_GRU_SLICES = 1 .. 6: tuning override (read per call)
    const char* e = getenv("DYGNN_GRU_SLICES");
    return (e && e[0] >= '1' && e[0] <= '6' && e[1] == 0) ? e[0] - '0' : kGruSlices;
}
static size_t gru_lds(const GruArgs& a, int MT) {
    const int per = ((a.Fn + 15) / 16 + gru_slices() - 1) / gru_slices();
    return (size_t)4 * MT * (pad_ld(a.Dm) + pad_ld(a.Fn) + 2 * 3 * per * 16) * sizeof(float);
}
constexpr size_t kLdsMax = 160 * 1024;

template <class K>
static int set_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return DYGNN_OK;
}
// Rows per workgroup = 4 MT.  Small levels take few rows so that they still spread over the chip (the weights are re-streamed from L2 by
// every workgroup: 4 rows at 800 rows = 200 workgroups); large levels take 32 rows (every weight fragment feeds 32 MFMAs).  A row's
// bits do not depend on the choice.
static int pick_mt(int64_t n) { return n <= 2048 ? 1 : n <= 8192 ? 2 : 4; }
static int env_mt(const char* name, int mt) {          // DYGNN_CHAIN_MT / DYGNN_GRU_MT = 1 | 2 | 4 | 8: tuning override (read per call)
    const char* e = getenv(name);
    if (e && (e[0] == '1' || e[0] == '2' || e[0] == '4' || e[0] == '8') && e[1] == 0) return e[0] - '0';
    return mt;
}

bool fits(int Fn, int Ft, int Dkv, int H) {
    if (H < 1 || (Fn + Ft) % H || Fn % 4 || Ft % 4 || Dkv % 4 || ((Fn + Ft) / H) % 4) return false;
    if (pad_ld(Fn + Ft) > 64 * kNQ || pad_ld(Fn + Ft + Fn) - (Fn + Ft) > 64 * kNQ) return false;      // RowRegs
    return pre_lds(Fn, Ft, H, 1) <= kLdsMax && post_lds(Fn, Ft, Dkv, H, 1) <= kLdsMax;
}

template <int MT>
static int launch_pre_mt(hipStream_t s, const PreArgs& a) {
    const size_t bytes = pre_lds(a.Fn, a.Ft, a.H, MT);
    if (int rc = set_lds(k_tgat_pre<MT>, bytes)) return rc;
    hipLaunchKernelGGL(k_tgat_pre<MT>, dim3((unsigned)ceil_div(a.n, 4 * MT), (unsigned)a.H), dim3(kThreads), bytes, s, a);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}
int launch_pre(hipStream_t s, const PreArgs& a) {
    if (a.n == 0) return DYGNN_OK;
    DYGNN_REQUIRE(fits(a.Fn, a.Ft, a.Dkv, a.H), "tgat chain: feature dims do not fit the row-block kernels");
    int MT = env_mt("DYGNN_CHAIN_MT", pick_mt(a.n));
    while (MT > 1 && pre_lds(a.Fn, a.Ft, a.H, MT) > kLdsMax) MT >>= 1;
    return MT == 8 ? launch_pre_mt<8>(s, a) : MT == 4 ? launch_pre_mt<4>(s, a) : MT == 2 ? launch_pre_mt<2>(s, a) : launch_pre_mt<1>(s, a);
}

template <int MT, int KC>
static int launch_post_mt(hipStream_t s, const PostArgs& a) {
    const size_t bytes = post_lds(a.Fn, a.Ft, a.Dkv, a.H, MT, KC);
    if (int rc = set_lds(k_tgat_post<MT, KC>, bytes)) return rc;
    hipLaunchKernelGGL((k_tgat_post<MT, KC>), dim3((unsigned)ceil_div(a.n, 4 * MT)), dim3(kThreads), bytes, s, a);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}
static int post_mt(int64_t n, int Fn, int Ft, int Dkv, int H) {
    int MT = env_mt("DYGNN_CHAIN_MT", pick_mt(n));
    while (MT > 1 && post_lds(Fn, Ft, Dkv, H, MT) > kLdsMax) MT >>= 1;
    return MT;
}
// the attention rides in k_tgat_post for small row blocks (4 or 8 rows: levels of up to a few thousand rows) whose neighbourhoods fit the
// two-waves-per-node form; DYGNN_CHAIN_ATTN=0 keeps it a kernel of its own (A/B switch)
static int fused_kc(int64_t n, int Fn, int Ft, int Dkv, int H, int k) {
    const char* e = getenv("DYGNN_CHAIN_ATTN");
    if (e && e[0] == '0') return 0;
    const int MT = post_mt(n, Fn, Ft, Dkv, H);
    if (MT > 2 || k > 20 || H > 2 || Dkv > 512) return 0;
    const int KC = k <= 10 ? 10 : 20;
    return post_lds(Fn, Ft, Dkv, H, MT, KC) <= kLdsMax ? KC : 0;
}
bool post_fuses_attention(int64_t n, int Fn, int Ft, int Dkv, int H, int k) { return fused_kc(n, Fn, Ft, Dkv, H, k) != 0; }
int launch_post(hipStream_t s, const PostArgs& a) {
    if (a.n == 0) return DYGNN_OK;
    DYGNN_REQUIRE(fits(a.Fn, a.Ft, a.Dkv, a.H), "tgat chain: feature dims do not fit the row-block kernels");
    const int MT = post_mt(a.n, a.Fn, a.Ft, a.Dkv, a.H);
    const int KC = a.qk ? fused_kc(a.n, a.Fn, a.Ft, a.Dkv, a.H, a.k) : 0;
    DYGNN_REQUIRE(KC != 0 || a.z != nullptr, "tgat chain: neither z nor a fusable attention");
    if (KC == 10) return MT == 2 ? launch_post_mt<2, 10>(s, a) : launch_post_mt<1, 10>(s, a);
    if (KC == 20) return MT == 2 ? launch_post_mt<2, 20>(s, a) : launch_post_mt<1, 20>(s, a);
    return MT == 8 ? launch_post_mt<8, 0>(s, a) : MT == 4 ? launch_post_mt<4, 0>(s, a) : MT == 2 ? launch_post_mt<2, 0>(s, a) : launch_post_mt<1, 0>(s, a);
}

template <int MT>
static int launch_gru_mt(hipStream_t s, const GruArgs& a) {
    const size_t bytes = gru_lds(a, MT);
    if (int rc = set_lds(k_tgn_gru_chain<MT>, bytes)) return rc;
    const int64_t blocks = ceil_div(a.max_rows, 4 * MT);      // the list length is only known on the device: a bounded grid walks it
    hipLaunchKernelGGL(k_tgn_gru_chain<MT>, dim3((unsigned)(blocks < 192 ? blocks : 192) + kPlainBlocks, gru_slices()), dim3(kThreads), bytes, s, a);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}
int launch_gru(hipStream_t s, const GruArgs& a) {
    if (a.max_rows == 0) return DYGNN_OK;
    DYGNN_REQUIRE(gru_lds(a, 1) <= kLdsMax && a.Fn % 4 == 0 && a.Dm % 4 == 0, "tgn chain: feature dims too large for the GRU row-block kernel");
    int MT = env_mt("DYGNN_GRU_MT", pick_mt(a.max_rows / 8));      // the list is a fraction of the level-0 set
    if (MT > 4) MT = 4;
    while (MT > 1 && gru_lds(a, MT) > kLdsMax) MT >>= 1;
    return MT == 4 ? launch_gru_mt<4>(s, a) : MT == 2 ? launch_gru_mt<2>(s, a) : launch_gru_mt<1>(s, a);
}

}  // namespace chain
}  // namespace dygnn
