// Fused DyGFormer forward for gfx950, "token-owner" layout (models/DyGFormer.py:68-194 end to end).
//
// One workgroup = 8 wave64 = 128 tokens: two (src,dst,t) pairs of <= 64 tokens (TPW = 4 token tiles per pair) or one
// pair of <= 128 tokens (TPW = 8; BASELINE config 4, L=512 / P=8).  Wave w owns 16 tokens x ALL 200 channels:
//   * the residual stream X^T (13 accumulator tiles = 52 VGPRs) never leaves the wave's registers;
//   * LayerNorm is wave-local (register sums + two cross-lane adds) and its output IS the MFMA B operand of the
//     QKV / FFN products — no LDS round trip, no partial-sum exchange between waves, no K-split;
//   * Q^T, softmax(S)^T, O^T and gelu(H)^T feed the next product straight from accumulators (same layout trick
//     an accumulator tile is the B operand of the product that sums over its rows).
// Only K and V of ONE head at a time live in LDS ([128 tokens][100], 2 x 51.2 KB); heads run back to back.
//
// Weights: all 8 waves consume the SAME fragments in the SAME order, so the whole model is ONE linear stream of
// 1-KiB MFMA-A fragments per kernel, brought on chip once per workgroup by LDS-DMA (global_load_lds, no VGPRs)
// into a 52-fragment LDS ring and read with ds_read_b128.  (Measured in tools/v3_ubench.hip: weight fragments
// loaded global->VGPR per wave hold the MFMA pipe at 66 %, LDS-DMA staged at 85 %, registers only 90 %.)
// The ring protocol: stages of 13 fragments; after the step that finishes a stage every wave waits for its own
// DMAs, passes one barrier, and issues its share of the stage four ahead.  Steps never straddle the ring end
// (the packer inserts pad fragments with the same rule the consumer applies).
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "dygformer_layout.h"
#include "dropout.h"

// Build-time switch (tools/ab_fused3.py builds the other arm with -DF3_KSKIP=0 to A/B it in one process):
//   F3_KSKIP     the K = 200 products (QKV, FFN W1) spend 2 instead of 4 MFMAs on their last k-chunk (192..207: only 8 real k), the head-dim
//                contractions (Q K^T, out-projection) 1 instead of 4 on theirs (96..111: only 4 real k)
// Measured and NOT kept (round 2, profiles/r02_fused3_ab.md): a software-pipelined FFN (stream order W1(p+1) before W2(p), GELU of step p
// issued inside the W1(p+1) block — in chunks between MFMA groups, or whole before / after the block's MFMAs with the two waves of a SIMD at
// opposite ends): 1.3-1.6 % SLOWER in every arrangement, the two waves of a SIMD already run the block one after the other (the older or
// prioritised wave takes nearly every matrix-pipe slot), so one GELU of the two is hidden as it is; a static s_setprio 1 for waves 4-7: +-0.2 %;
// stage barriers every 13 instead of 26 fragments in the FFN with the next group's fragments read before the barrier: slower (twice the barriers).
#ifndef F3_XBAR
#define F3_XBAR 0      // 1: the FFN's W1 blocks (2: W2 blocks too) start on fragments read across the stage barrier in front of them.  Measured and
#endif                 // NOT kept (round 3, tools/ab_fused3.py, 32 steps per launch): 1 = -0.3 %, 2 = -2.7 % (16 more live VGPRs spill)
#ifndef F3_KSKIP
#define F3_KSKIP 1
#endif

namespace dygnn {
namespace v3 {

using f4 = __attribute__((ext_vector_type(4))) float;
using i4 = __attribute__((ext_vector_type(4))) int;

constexpr int kD = 200, kDP = 208, kNT = 13, kKC = 13, kHD = 100, kHid = 800, kC = 50;
constexpr int kFrag = 256;            // floats per 16x16 fragment
constexpr int kRing = 52;             // LDS ring, fragments
// F3_STAGE = 26 (two stages of 26 fragments: half the stage barriers) measured +1.25 % at L = 64, +0.8 % at L = 512 (round 3, tools/ab_fused3.py) and
// NOT kept: with two stages the fragments a step reads ahead ACROSS the barrier that ends a stage belong to a stage whose DMAs that very
// barrier publishes (with four stages the barrier one stage earlier did) — a read-ahead that is legal only with three stages in flight;
// re-reading after the barrier costs what the halved barriers save.
#ifndef F3_STAGE
#define F3_STAGE 13
#endif
constexpr int kStage = F3_STAGE;      // DMA / barrier granularity, fragments: the ring holds kRing / kStage stages
constexpr int kNStage = kRing / kStage;
static_assert(kStage * kNStage == kRing && kNStage >= 2, "the ring is a whole number (>= 2) of stages");
constexpr int kTokWG = 128;           // tokens per workgroup
constexpr int kKV = 100;              // K/V row stride (floats): 4*25 -> conflict-free b128 row reads and b32 column reads
constexpr int kLdsK = 0;
constexpr int kLdsV = kTokWG * kKV;                 // 12800
constexpr int kLdsRing = 2 * kTokWG * kKV + 16;     // 16 floats of slack: tile 6 of the last row reads 12 floats past it
constexpr int kLdsMisc = kLdsRing + kRing * kFrag;  // 38928 floats = 155,712 B
constexpr int kMiscB1 = 0;            // [2][800]: FFN hidden bias, double-buffered by layer parity (no barrier needed:
                                      // dozens of stream barriers lie between a buffer's write and its reads / reuse)
constexpr int kMiscFloats = 2 * kHid;               // 1600
constexpr int kLdsBytes = 160 * 1024;
static_assert((kLdsMisc + kMiscFloats) * 4 <= kLdsBytes, "LDS budget");
constexpr int kScratchFloats = 2 * kTokWG * kKV;    // prologue window arrays live in the K/V region

__device__ __forceinline__ f4 mfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
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
// last k-chunk of a K = 200 product: k = 192..199 packed into TWO MFMAs (b0: k = 192 + {0,4,1,5}[g], b1: k = 192 + {2,6,3,7}[g]);
// the A fragments of that chunk are packed to match (FragDesc.kmode 1)
template <int N>
__device__ __forceinline__ void mma_group2(f4* acc, const f4* a, const float b0, const float b1) {
#pragma unroll
    for (int u = 0; u < N; ++u) acc[u] = mfma(a[u].x, b0, acc[u]);
#pragma unroll
    for (int u = 0; u < N; ++u) acc[u] = mfma(a[u].y, b1, acc[u]);
}
// v = rows 192 + 4g + r of an accumulator-layout tile (g >= 2: zero padding).  v_permlane32_swap moves lanes 0..31 of the second
// operand into lanes 32..63 of the first: (x, y) -> lanes g = 0,1,2,3 hold rows 192, 196, 193, 197; (z, w) -> 194, 198, 195, 199
__device__ __forceinline__ void kpack(const f4 v, float& b0, float& b1) {
    // (scalars first: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 whatever the element — hipcc, ROCm 7.2)
    const float vx = v.x, vy = v.y, vz = v.z, vw = v.w;
    const auto r0 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, vx), __builtin_bit_cast(unsigned, vy), false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, vz), __builtin_bit_cast(unsigned, vw), false, false);
    b0 = __builtin_bit_cast(float, r0[0]);
    b1 = __builtin_bit_cast(float, r1[0]);
}
// v = rows 96 + 4 g + r of a head-dim tile: only rows 96 .. 99 (lane group 0) are real (head dim 100).  Returns the B operand of ONE
// MFMA that carries all four: lane group g holds row 96 + g (permlane16_swap: 16-lane rows 1, 3 of the first operand <-> rows 0, 2 of the
// second; then permlane32_swap as in kpack).  The A fragments of that k-chunk are packed to match (FragDesc.kmode 2: k = 96 + g).
__device__ __forceinline__ float kpack4(const f4 v) {
    const float vx = v.x, vy = v.y, vz = v.z, vw = v.w;
    const auto t1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, vx), __builtin_bit_cast(unsigned, vy), false, false);
    const auto t2 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, vz), __builtin_bit_cast(unsigned, vw), false, false);
    const auto r = __builtin_amdgcn_permlane32_swap(t1[0], t2[0], false, false);
    return __builtin_bit_cast(float, r[0]);
}
template <int N>
__device__ __forceinline__ void mma_group1(f4* acc, const f4* a, const float b0) {
#pragma unroll
    for (int u = 0; u < N; ++u) acc[u] = mfma(a[u].x, b0, acc[u]);
}
__device__ __forceinline__ f4 ldg4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ f4 lds4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ f4 zero4() { return f4{0.f, 0.f, 0.f, 0.f}; }
// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS [dst, dst + 1 KiB), no VGPR destination.
// Written as inline asm on purpose.  With the builtin (__builtin_amdgcn_global_load_lds) hipcc (ROCm 7.2) knows an LDS-DMA is in flight
// and from then on waits `s_waitcnt lgkmcnt(0)` — not a counted lgkmcnt(N) — before the MFMAs that consume ds_read results: every other
// MFMA group of every weight loop then stalls for the LDS round trip of the fragments it has just PREFETCHED for the next group (this
// kernel keeps a DMA in flight all the time).  The asm form is invisible to that bookkeeping; the protocol needs nothing from it: every
// wave drains its own DMAs with an explicit `s_waitcnt vmcnt(0)` in front of the stage barrier (WStream::advance).
// The destination is given as a FLOAT OFFSET into the kernel's one dynamic LDS array (which starts at __builtin_amdgcn_groupstaticsize():
// the kernel has no static LDS), not as a pointer: an addrspacecast of a generic pointer in this position trips an instruction verifier error.
__device__ __forceinline__ void dma_frag(const float* gsrc_lane, int lds_float_off_uniform) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_groupstaticsize() + 4u * (unsigned)lds_float_off_uniform);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc_lane), "s"(m0v) : "memory", "m0");
}

// sum over the 16 lanes of a DPP row (= the 16 tokens of a tile, lane & 15), result in every lane: four VALU adds with
// DPP operands (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror) instead of four LDS bpermutes
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}

// cos for the time encoder.  The argument w*dt+b reaches 2.7e6 rad where libm's cosf takes its slow Payne-Hanek path; here x/(2 pi) is
// formed as a two-float product (INV_HI + INV_LO = 1/(2 pi) to 2^-52), its fractional part u in [0, 0.5] is folded to [0, 0.25] and
// cos(2 pi u) evaluated by an even degree-12 minimax polynomial (|err| <= 6e-8 on the folded range); beyond 3e7 the product's rounding
// error would exceed 1e-7 turns and libm is called instead (tests/test_dygformer_gpu.py::test_large_timestamps_take_the_libm_cosine_path).
// erf for the exact GELU: Abramowitz & Stegun 7.1.26 (|err| <= 1.5e-7), branch-free.
__device__ __forceinline__ float cos_time_fast(float x) {      // |x| <= 3e7 (branch-free; cos_time checks)
    const float INV_HI = 0.15915493667125702f, INV_LO = 6.4206382432985265e-09f;
    const float p = x * INV_HI;
    const float e = fmaf(x, INV_HI, -p);
    const float q = fmaf(x, INV_LO, e);
    const float t = (p - rintf(p)) + q;
    float u = fabsf(t);
    u = u > 0.5f ? 1.0f - u : u;
    const bool flip = u > 0.25f;
    const float v = flip ? 0.5f - u : u;
    const float z = v * v;
    float r = fmaf(7.903536371318467f, z, -26.42625678337438f);
    r = fmaf(r, z, 60.24464137187666f);
    r = fmaf(r, z, -85.45681720669373f);
    r = fmaf(r, z, 64.93939402266829f);
    r = fmaf(r, z, -19.739208802178716f);
    r = fmaf(r, z, 1.0f);
    return flip ? -r : r;
}
__device__ __forceinline__ float cos_time(float x) { return fabsf(x) <= 3.0e7f ? cos_time_fast(x) : cosf(x); }
// the same operations on two arguments at once, written on 2-vectors so that hipcc emits packed fp32 instructions (v_pk_mul / v_pk_fma /
// v_pk_add_f32: two results in ~1.6 issue slots, tools/coissue_ubench.hip; VALU instructions take matrix-pipe time in this kernel)
using f2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 cos_time_fast2(f2 x) {
    const f2 INV_HI = {0.15915493667125702f, 0.15915493667125702f}, INV_LO = {6.4206382432985265e-09f, 6.4206382432985265e-09f};
    const f2 p = x * INV_HI;
    const f2 e = pk_fma(x, INV_HI, -p);
    const f2 q = pk_fma(x, INV_LO, e);
    const f2 rp = {rintf(p.x), rintf(p.y)};
    const f2 t = (p - rp) + q;
    f2 u = {fabsf(t.x), fabsf(t.y)};
    const f2 one_u = f2{1.0f, 1.0f} - u;
    u = f2{u.x > 0.5f ? one_u.x : u.x, u.y > 0.5f ? one_u.y : u.y};
    const bool fx = u.x > 0.25f, fy = u.y > 0.25f;
    const f2 half_u = f2{0.5f, 0.5f} - u;
    const f2 v = {fx ? half_u.x : u.x, fy ? half_u.y : u.y};
    const f2 z = v * v;
    auto c2 = [](float c) { return f2{c, c}; };
    f2 r = pk_fma(c2(7.903536371318467f), z, c2(-26.42625678337438f));
    r = pk_fma(r, z, c2(60.24464137187666f));
    r = pk_fma(r, z, c2(-85.45681720669373f));
    r = pk_fma(r, z, c2(64.93939402266829f));
    r = pk_fma(r, z, c2(-19.739208802178716f));
    r = pk_fma(r, z, c2(1.0f));
    return f2{fx ? -r.x : r.x, fy ? -r.y : r.y};
}
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __expf(-ax * ax);
    return copysignf(fmaf(-p * t, e, 1.0f), x);
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erf_as(v * 0.70710678118654752440f)); }

struct LayerP {
    const float* b1;      // [800]; every other per-layer vector travels in the weight stream
};

struct Args {
    const int64_t* indptr; const int32_t* nbr; const int32_t* eid; const double* ts;
    const int64_t *src, *dst; const double* times;
    const int32_t* hist_len; const int64_t* end_pos; const CallDims* cd;
    const float *node_feat, *edge_feat, *time_w, *time_b, *lut;
    const float* stream; int nstages;
    const float* projw;           // projection fragments in step order [node | time | edge | cooc chunks][4 tiles]
    int proj_frags;               // total projection fragments
    int slab_chunks;              // k-chunk slots per LDS HALF (a multiple of 4; two halves)
    int scr_floats;               // LDS floats reserved for the window arrays (the slab follows)
    int slab_in_ring;             // long windows (e.g. L = 2048): the slab borrows the weight ring, whose stream then opens after the prologue
    int tab_off, tab_slots, tab_bits;   // co-occurrence table per pair (LDS word offset, slots = 2^bits; 0: counts by scanning the rows)
    const float* bias_x;          // [208] projection biases in model-dim order
    const float* outfrag;         // output layer as fragments [ceil(Fn/16) tiles][13 k-chunks]
    LayerP layer[DYGNN_MAX_LAYERS];
    const float *outT, *outb;     // output layer: transposed [200][Fn], bias [Fn]
    float *out_src, *out_dst;
    float* tap_enc; float* tap_layer[DYGNN_MAX_LAYERS];
    unsigned long long* stamps;
    int64_t B, G, num_nodes;
    int64_t pair_stride;          // > 0 (two pairs per workgroup only): workgroup w holds pairs w and w + pair_stride — the positive and the
                                  // negative call of one edge: where their (src, t) agree the src side is projected once
    int Fn, Fe, Ft, P, L, NL, Tmax;
    int nchunk[4];
    float qscale;
    train::TrainOut tr;           // training forward only (k_dygformer_fused3<.., true>): the dense activations the backward pass reads
};

__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};      // load / LDS-DMA source for rows that do not exist

// ---- the shared weight stream ---------------------------------------------------------------------------------
struct WStream {
    const float* gsrc;    // stream base + lane*4 (per lane)
    int ring;             // LDS ring: float offset into the dynamic LDS array (wave-uniform)
    int wave, nstages;
    int nw;               // waves of the workgroup (8; 4 in the one-pair-per-workgroup kernels for small batches): they split a stage's 13 fragments
    int pos;              // ring slot of the next fragment
    int instage;          // fragments consumed of the current stage
    int issued;           // stages whose DMA this wave has issued
    __device__ __forceinline__ void issue(int s) {
        if (s < nstages) {
            const float* srcp = gsrc + (size_t)s * (kStage * kFrag);
            const int dst = ring + (s % kNStage) * (kStage * kFrag);
#pragma unroll
            for (int u = 0; u < (kStage + 3) / 4; ++u) {      // nw >= 4 waves split the stage's fragments
                const int f = wave + u * nw;
                if (f < kStage) dma_frag(srcp + f * kFrag, dst + f * kFrag);
            }
        }
    }
    __device__ __forceinline__ void open(const float* stream, int ring_, int lane, int wave_, int nstages_, int nw_ = 8) {
        gsrc = stream + lane * 4; ring = ring_; wave = wave_; nstages = nstages_; nw = nw_;
        pos = 0; instage = 0; issued = kNStage;
#pragma unroll
        for (int s = 0; s < kNStage; ++s) issue(s);
    }
    // n fragments consumed (or skipped).  Crossing a stage boundary: wait for own DMAs, barrier (every wave is done with
    // the finished stage, every stage issued before is now visible), then refill the freed ring quarter.
    // younger_stores (training forward): this wave has issued exactly that many global stores since its last DMA issue.  vmcnt retires in
    // issue order, so waiting until only those are outstanding proves the (older) DMAs landed without exposing the stores' latency.
    __device__ __forceinline__ void advance(int n, int younger_stores = 0) {
        pos += n;
        if (pos >= kRing) pos -= kRing;
        instage += n;
        if (instage >= kStage) {
            if (younger_stores == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (younger_stores == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's LDS-DMA has landed before anyone passes the barrier
            __syncthreads();
            do { instage -= kStage; issue(issued); ++issued; } while (instage >= kStage);
        }
    }
    __device__ __forceinline__ void fit(int n) { if (pos + n > kRing) advance(kRing - pos); }
    __device__ __forceinline__ void align26() {
        if (pos != 0 && pos != 26) advance(pos < 26 ? 26 - pos : kRing - pos);
    }
    // ring slot of the step after one of n fragments that starts at `pos` (same rule as advance + fit)
    __device__ __forceinline__ int next_pos(int n, int n_next) const {
        int p = pos + n;
        if (p >= kRing) p -= kRing;
        if (p + n_next > kRing) p = 0;
        return p;
    }
};

// Diagnostic build (-DDYGNN_STAMPS): every wave accumulates s_memtime ticks per phase category and the last four
// workgroups of the grid store them: taps.phase_cycles[wg][wave][cat]; cat 31 = total.
#ifdef DYGNN_STAMPS
#define TDECL unsigned long long tacc_[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tk_ = __builtin_amdgcn_s_memtime(); const unsigned long long tk0_ = tk_
#define TACC(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc_[i] += t_ - tk_; tk_ = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define TSTORE()                                                                                   \
    do {                                                                                           \
        if (a.stamps != nullptr && lane == 0 && blockIdx.x + 4 >= gridDim.x) {                     \
            unsigned long long* o_ = a.stamps + ((size_t)(blockIdx.x + 4 - gridDim.x) * 8 + wave) * 32;   \
            for (int i_ = 0; i_ < 24; ++i_) o_[i_] = tacc_[i_];                                    \
            o_[31] = tk_ - tk0_;                                                                   \
        }                                                                                          \
    } while (0)
#else
#define TDECL do { } while (0)
#define TACC(i) do { } while (0)
#define TSTORE() do { } while (0)
#endif
enum { T_WIN = 0, T_PROJ, T_LN, T_QKV, T_QKVBAR, T_ATTN, T_OPROJ, T_FFN, T_POOL, T_MISC, T_POOL1, T_POOL2, T_PNODE, T_PTIME, T_PEDGE, T_PCOOC,
       T_F_W1 = 16, T_F_GELU, T_F_ADV1, T_F_W2, T_F_ADV2 };      // FFN sub-phases

// LayerNorm of the register-resident X^T (two-pass, biased variance, eps 1e-5); gamma/beta from LDS
__device__ __forceinline__ void layernorm(f4 (&xn)[kNT], const f4 (&x)[kNT], const float* gamma, const float* beta, int g, float& mean_o, float& rstd_o) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kNT; ++i) s += (x[i].x + x[i].y) + (x[i].z + x[i].w);     // rows 200..207 are exact zeros
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / kD);
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < kNT; ++i) {
        if (i < 12 || g < 2) {            // rows 200..207 (tile 12, g >= 2) are padding
            const float d0 = x[i].x - mean, d1 = x[i].y - mean, d2 = x[i].z - mean, d3 = x[i].w - mean;
            v += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    const float rstd = 1.0f / sqrtf(v * (1.0f / kD) + 1e-5f);
    mean_o = mean; rstd_o = rstd;
#pragma unroll
    for (int i = 0; i < kNT; ++i) {
        const f4 gm = lds4(gamma + 16 * i + 4 * g), bt = lds4(beta + 16 * i + 4 * g);   // zero beyond 200
        xn[i].x = (x[i].x - mean) * rstd * gm.x + bt.x;
        xn[i].y = (x[i].y - mean) * rstd * gm.y + bt.y;
        xn[i].z = (x[i].z - mean) * rstd * gm.z + bt.z;
        xn[i].w = (x[i].w - mean) * rstd * gm.w + bt.w;
    }
}

// acc[NT] += W(NT tiles of one head's q, k or v) . xn : 13 stream steps of NT fragments [k-chunk][tile], each multiplied
// as sub-groups of 4 and NT - 4 tiles whose fragments are read one sub-group ahead (8 fragments live instead of 14)
template <int NT>
__device__ __forceinline__ void qkv_group(f4 (&acc)[NT], const f4 (&xn)[kNT], const float xk0, const float xk1, WStream& ws, const float* ringl, bool active) {
    constexpr int N2 = NT - 4;
    f4 fs[2][4];
    ws.fit(NT);
    if (active) {
#pragma unroll
        for (int u = 0; u < 4; ++u) fs[0][u] = lds4(ringl + (ws.pos + u) * kFrag);
    }
#pragma unroll
    for (int kc = 0; kc < kKC; ++kc) {
        if (active) {
#pragma unroll
            for (int u = 0; u < N2; ++u) fs[1][u] = lds4(ringl + (ws.pos + 4 + u) * kFrag);
            __builtin_amdgcn_sched_barrier(0);
            if (F3_KSKIP && kc == kKC - 1) mma_group2<4>(&acc[0], fs[0], xk0, xk1); else mma_group<4>(&acc[0], fs[0], xn[kc]);
            __builtin_amdgcn_sched_barrier(0);
            if (kc + 1 < kKC) {
                const int p1 = ws.next_pos(NT, NT);
#pragma unroll
                for (int u = 0; u < 4; ++u) fs[0][u] = lds4(ringl + (p1 + u) * kFrag);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (F3_KSKIP && kc == kKC - 1) mma_group2<N2>(&acc[4], fs[1], xk0, xk1); else mma_group<N2>(&acc[4], fs[1], xn[kc]);
            __builtin_amdgcn_sched_barrier(0);
        }
        ws.advance(NT);
        if (kc + 1 < kKC) ws.fit(NT);
    }
}


// ---- FFN blocks: one block = the 26 fragments of one product of one step, at ring position 0 or 26.
__device__ __forceinline__ void gelu_tiles(f4 (&t)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        t[u].x = gelu_erf(t[u].x); t[u].y = gelu_erf(t[u].y); t[u].z = gelu_erf(t[u].z); t[u].w = gelu_erf(t[u].w);   // DyGFormer.py:458
    }
}
// First product: h[2] (two 16-wide hidden tiles) = b1 + W1 . LN(x); 13 k-chunks of two fragments [k-chunk][tile] read one chunk ahead.
// pre != nullptr: the block's first two fragments were read by the caller BEFORE the stage barrier in front of the block (legal: a block's
// fragments were made visible by the barrier one block earlier; the barrier in front of it only frees the ring half behind it) — the block
// then starts on operands that are already in registers instead of exposing an LDS round trip to both waves of the SIMD at once
__device__ __forceinline__ void ffn_w1(f4 (&h)[2], const f4 (&xn)[kNT], const float xk0, const float xk1, const float* abuf, const float* b1p, const int g,
                                       const f4* pre = nullptr) {
    if (b1p != nullptr) { h[0] = lds4(b1p + 4 * g); h[1] = lds4(b1p + 16 + 4 * g); }
    else { h[0] = zero4(); h[1] = zero4(); }
    f4 sa[2][2];
    if (pre) { sa[0][0] = pre[0]; sa[0][1] = pre[1]; }
    else { sa[0][0] = lds4(abuf); sa[0][1] = lds4(abuf + kFrag); }
#pragma unroll
    for (int kc = 0; kc < kKC; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < kKC) {
            sa[cur ^ 1][0] = lds4(abuf + (size_t)(2 * (kc + 1)) * kFrag);
            sa[cur ^ 1][1] = lds4(abuf + (size_t)(2 * (kc + 1) + 1) * kFrag);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (F3_KSKIP && kc == kKC - 1) mma_group2<2>(h, sa[cur], xk0, xk1); else mma_group<2>(h, sa[cur], xn[kc]);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// Second product: acc (13 model-dim tiles) += W2[:, the two hidden tiles] . h; fragments [tile u][n-tile i] in sub-groups (4,3,3,3)
// read one sub-group ahead
__device__ __forceinline__ void ffn_w2(f4 (&acc)[kNT], const f4 (&h)[2], const float* bbuf, const f4* pre = nullptr) {
    f4 fs[2][4];
#pragma unroll
    for (int v = 0; v < 4; ++v) fs[0][v] = pre ? pre[v] : lds4(bbuf + (size_t)v * kFrag);
#pragma unroll
    for (int gi = 0; gi < 8; ++gi) {
        const int u = gi >> 2, q = gi & 3;
        const int i0 = q == 0 ? 0 : 4 + 3 * (q - 1), n = q == 0 ? 4 : 3;
        if (gi + 1 < 8) {
            const int u2 = (gi + 1) >> 2, q2 = (gi + 1) & 3;
            const int j0 = q2 == 0 ? 0 : 4 + 3 * (q2 - 1), n2 = q2 == 0 ? 4 : 3;
#pragma unroll
            for (int v = 0; v < 4; ++v) if (v < n2) fs[(gi + 1) & 1][v] = lds4(bbuf + (size_t)(u2 * 13 + j0 + v) * kFrag);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (n == 4) mma_group<4>(&acc[i0], fs[gi & 1], h[u]); else mma_group<3>(&acc[i0], fs[gi & 1], h[u]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int TPW>
__device__ __forceinline__ void tap_store(const f4 (&x)[kNT], float* base, int64_t b, int Tmax, int T, int tt, int c, int g) {
    if (base == nullptr) return;
    const int tok = 16 * tt + c;
    if (tok >= T) return;
#pragma unroll
    for (int i = 0; i < kNT; ++i) {
        const int n = 16 * i + 4 * g;
        if (n < kD) *reinterpret_cast<f4*>(base + ((size_t)b * Tmax + tok) * kD + n) = x[i];
    }
}

// training forward: the 13 register tiles of a token-owner wave (rows 16 i + 4 g + r of token c) as dense row `row` of a [M][200] buffer
__device__ __forceinline__ void store_rows(float* base, int64_t row, const f4 (&x)[kNT], int g, bool valid) {
    if (!valid) return;
    float* p = base + row * kD + 4 * g;
#pragma unroll
    for (int i = 0; i < kNT; ++i)
        if (i < 12 || g < 2) *reinterpret_cast<f4*>(p + 16 * i) = x[i];
}
__device__ __forceinline__ void load_rows(f4 (&x)[kNT], const float* base, int64_t row, int g, bool valid) {
#pragma unroll
    for (int i = 0; i < kNT; ++i) x[i] = (valid && (i < 12 || g < 2)) ? ldg4(base + row * kD + 4 * g + 16 * i) : zero4();
}
// x = xin + dropout(y + bias) (DyGFormer.py:456, :460), xin re-read from its dense rows, x also written to `out` (or not: nullptr).  Element
// (row, n = 16 i + 4 g + r) draws mask(site, row * 200 + n) (indices < 2^32: checked by the host).  One tile at a time — load, hash, add,
// store — so that no more than a tile's worth of temporaries is alive beside the two register sets.
__device__ __forceinline__ void residual_dropped(f4 (&x)[kNT], const float* xin, float* out, const f4 (&y)[kNT], const float* bias_lds, const train::Drop& dr,
                                                 uint32_t site, int64_t row, int g, bool valid) {
    const uint32_t sk = dr.site_key(site), e0 = (uint32_t)row * kD + 4 * g;
    const float* src = xin + row * kD + 4 * g;
    float* dst = out ? out + row * kD + 4 * g : nullptr;
    // (all thirteen row loads in flight first would expose one latency instead of thirteen, but next to the two live register sets it spills:
    //  measured 10 % slower)
    constexpr int RQ = 1;                                    // row tiles in flight ahead of the one being finished
    f4 rq[RQ];
#pragma unroll
    for (int u = 0; u < RQ; ++u) rq[u] = (valid && (u < 12 || g < 2)) ? ldg4(src + 16 * u) : zero4();
#pragma unroll
    for (int i = 0; i < kNT; ++i) {
        const bool on = valid && (i < 12 || g < 2);          // rows 200 .. 207 do not exist
        f4 v = rq[i % RQ];
        if (i + RQ < kNT) rq[i % RQ] = (valid && (i + RQ < 12 || g < 2)) ? ldg4(src + 16 * (i + RQ)) : zero4();
        const f4 bv = lds4(bias_lds + 16 * i);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (y[i][r] + bv[r]) * dr.mask32(sk, e0 + 16 * i + r);      // rows >= 200: y and the bias are zero
        if (on && dst) *reinterpret_cast<f4*>(dst + 16 * i) = v;
        x[i] = v;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ================================================================================================
// TR = false: inference.  TR = true: the training forward — dropout at the reference's four sites per layer and every
// activation the backward pass reads written to HBM as dense rows (a.tr); the residual stream is re-read from those rows after the
// attention and the FFN block instead of being kept in registers next to the separate accumulators the dropout needs.
// NW = waves per workgroup: 8 (two pairs of <= 64 tokens, or one of <= 128), or 4 = ONE pair of <= 64 tokens per workgroup, for calls of at most 256
// pairs (the reference's own 200-pair call: 100 eight-wave workgroups would leave 156 CUs idle and run two waves per SIMD on the rest; 200
// four-wave workgroups give every wave a matrix pipe of its own)
template <int TPW, bool TR, int NW = 8>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void k_dygformer_fused3(const Args a) {
    constexpr int NP = NW / TPW;                 // pairs per workgroup
    constexpr int PT = 64 * NW / NP;             // threads per pair
    constexpr int NTHR = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pi = wave / TPW, tt = wave % TPW;
    const int c = lane & 15, g = lane >> 4;
    const bool paired = NP == 2 && a.pair_stride > 0;
    const int64_t b = paired ? (int64_t)blockIdx.x + pi * a.pair_stride : (int64_t)blockIdx.x * NP + pi;
    const bool pair_ok = paired ? blockIdx.x < a.pair_stride && b < a.B : b < a.B;
    const int ptid = tid - pi * PT;

    TDECL;
    CallDims cd{};
    if (pair_ok) cd = a.cd[b / a.G];
    const int Ss = cd.S_s, Sd = cd.S_d, Ts = cd.T_s, T = cd.T;
    const int SsA = (Ss + 3) & ~3, SdA = (Sd + 3) & ~3, SA = SsA + SdA;
    const bool active = 16 * tt < T;             // wave-uniform: this token tile holds real tokens
    const int tokbase = pi * (16 * TPW);         // this pair's first K/V row
    // f4 (caller-side fusion, train_link_prediction.py:166 / evaluate_models_utils.py:62-63): the second pair of the workgroup is the
    // NEGATIVE call of the first pair's edge when it has the same source, the same time and the same padded source length.  Its token
    // tiles that hold source tokens only then take the node / edge / time rows of the residual stream from the first pair's same tile
    // instead of gathering and projecting them again (rows of one channel receive non-zero terms from that channel only, so the bits are
    // those of a separate call); the co-occurrence rows and the destination side are its own.  Anything else: the plain path.
    bool src_shared = false;
    if (NP == 2) {
        if (paired && pi == 1 && pair_ok) {
            const int64_t b0 = blockIdx.x;
            const CallDims cd0 = a.cd[b0 / a.G];
            src_shared = a.src[b0] == a.src[b] && __double_as_longlong(a.times[b0]) == __double_as_longlong(a.times[b]) && cd0.S_s == Ss &&
                         16 * (tt + 1) <= Ts;
        }
        src_shared = __builtin_amdgcn_readfirstlane((int)src_shared) != 0;
    }
    const bool donor = NP == 2 && paired && pi == 0 && 16 * (tt + 1) <= Ts;     // first-pair tiles a shared tile may copy from (checked by the taker)

    // ---- weights start moving at once: the first four stages of the layer stream into the ring, the first two halves of
    // projection fragments into the K/V region behind the window arrays (all of it lands during the window phase)
    WStream ws;
    if (!a.slab_in_ring) ws.open(a.stream, kLdsRing, lane, wave, a.nstages, NW);
    const float* ringl = lds + kLdsRing + lane * 4;
    // projection fragments: two LDS halves of `slab_chunks` k-chunk slots each (4 fragments per slot), refilled by LDS-DMA one half
    // ahead of the consumer (half q of the slot sequence lives in buffer q & 1)
    const int slab_off = a.slab_in_ring ? kLdsRing : a.scr_floats;
    const float* slabl = lds + slab_off + lane * 4;
    const int hc = a.slab_chunks;
    const int half_frags = 4 * hc;
    auto load_half = [&](int q) {
        const int f0 = q * half_frags;
        const int n = a.proj_frags - f0 < half_frags ? a.proj_frags - f0 : half_frags;      // <= 0 beyond the last half
        for (int f = wave; f < n; f += NW) dma_frag(a.projw + (size_t)(f0 + f) * kFrag + lane * 4, slab_off + ((q & 1) * half_frags + f) * kFrag);
    };
    load_half(0);
    load_half(1);
    float* tws = lds + kLdsMisc + kMiscFloats;      // time-encoder w | b
    for (int i = tid; i < 2 * a.Ft; i += NTHR) tws[i] = i < a.Ft ? a.time_w[i] : a.time_b[i - a.Ft];

    // ---- windows (pad_sequences, DyGFormer.py:228-245): per-pair arrays in the (still unused) K/V region.
    // src positions at [0, Ss), dst positions at [SsA, SsA + Sd); alignment gaps hold id -1 (matches nothing).
    int32_t* ids = reinterpret_cast<int32_t*>(lds) + pi * (a.scr_floats / NP);
    int32_t* eids = ids + SA;
    float* dts = reinterpret_cast<float*>(eids + SA);
    int32_t* c0 = reinterpret_cast<int32_t*>(dts + SA);
    int32_t* c1 = c0 + SA;
    if (pair_ok) {
        const double tq = a.times[b];
        for (int p = ptid; p < SA; p += PT) {
            const bool is_dst = p >= SsA;
            const int j = is_dst ? p - SsA : p;
            int32_t id = -1, e = 0;
            float dt = 0.f;
            if (j < (is_dst ? Sd : Ss)) {
                const int64_t q = is_dst ? a.B + b : b;
                const int32_t len = a.hist_len[q];
                const int32_t m = len < a.L - 1 ? len : a.L - 1;
                float tn = 0.f;
                id = 0;
                if (j == 0) {
                    const int64_t qid = is_dst ? a.dst[b] : a.src[b];
                    id = qid < 0 || qid >= a.num_nodes ? 0 : (int32_t)qid;      // a bad query id is the padding node, as in sampler.hip (never a fault)
                    tn = (float)tq;
                } else if (j <= m) {
                    const int64_t pos = a.end_pos[q] - m + (j - 1);
                    id = a.nbr[pos]; e = a.eid[pos]; tn = (float)a.ts[pos];
                }
                dt = (float)(tq - (double)tn);                      // DyGFormer.py:263
            }
            ids[p] = id; eids[p] = e; dts[p] = dt;
        }
    }
    // long windows: an open-addressing table [keys | counts] per pair behind the projection halves (a.tab_slots > 0), cleared here
    int32_t* tkeys = reinterpret_cast<int32_t*>(lds) + a.tab_off + pi * 2 * a.tab_slots;
    int32_t* tcnts = tkeys + a.tab_slots;
    for (int i = ptid; i < a.tab_slots; i += PT) { tkeys[i] = -2; tcnts[i] = 0; }
    __syncthreads();
    // ---- co-occurrence counts (DyGFormer.py:337-393)
    if (a.tab_slots > 0) {
        // Windows of hundreds of positions (L = 512: 1,024 positions per pair, a million comparisons by the scan below = 5 % of the kernel):
        // every position inserts its id into the table (linear probing; the slot's count word holds the source-side count in its low
        // half, the destination-side count in its high half), one barrier, every position reads its id's slot.  Exact integers.
        const uint32_t mask = (uint32_t)a.tab_slots - 1;
        const int shift = 32 - a.tab_bits;
        if (pair_ok) {
            for (int p = ptid; p < SA; p += PT) {
                const int32_t v = ids[p];
                if (v <= 0) continue;
                uint32_t sl = ((uint32_t)v * 2654435761u) >> shift;
                for (int it = 0; it < a.tab_slots; ++it, sl = (sl + 1) & mask) {
                    const int32_t old = atomicCAS(&tkeys[sl], -2, v);
                    if (old == -2 || old == v) { atomicAdd(&tcnts[sl], p >= SsA ? 0x10000 : 1); break; }
                }
            }
        }
        __syncthreads();
        if (pair_ok) {
            for (int p = ptid; p < SA; p += PT) {
                const int32_t v = ids[p];
                int32_t cs = 0, cdn = 0;
                if (v > 0) {
                    uint32_t sl = ((uint32_t)v * 2654435761u) >> shift;
                    for (int it = 0; it < a.tab_slots && tkeys[sl] != v; ++it) sl = (sl + 1) & mask;
                    const int32_t w = tcnts[sl];
                    cs = w & 0xffff; cdn = w >> 16;
                }
                c0[p] = cs; c1[p] = cdn;
            }
        }
    } else if (pair_ok) {
        // one thread per position, 4 ids per broadcast LDS read
        for (int p = ptid; p < SA; p += PT) {
            const int32_t v = ids[p];
            int32_t cs = 0, cdn = 0;
            for (int q = 0; q < SsA; q += 4) {
                const i4 w = *reinterpret_cast<const i4*>(ids + q);
                cs += (w.x == v) + (w.y == v) + (w.z == v) + (w.w == v);
            }
            for (int q = SsA; q < SA; q += 4) {
                const i4 w = *reinterpret_cast<const i4*>(ids + q);
                cdn += (w.x == v) + (w.y == v) + (w.z == v) + (w.w == v);
            }
            if (v <= 0) { cs = 0; cdn = 0; }       // padding node 0 (DyGFormer.py:389-391) and alignment gaps
            c0[p] = cs; c1[p] = cdn;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of the first two projection halves and of the first ring stages has landed
    __syncthreads();

    TACC(T_WIN);
    // ---- resident residual stream X^T: 13 tiles (rows 16i+4g+r) x token c of tile tt
    f4 x[kNT];
#pragma unroll
    for (int i = 0; i < kNT; ++i) x[i] = ldg4(a.bias_x + 16 * i + 4 * g);

    // ---- patch projection (DyGFormer.py:148-157): channel ch writes model rows 50ch..50ch+49 = tiles (50ch)/16 .. +3.
    // One step = one 16-wide k-chunk x 4 tiles; the fragments of `slab_chunks` steps sit in an LDS half (loaded by LDS-DMA one half
    // ahead, all waves use the same ones), so inside a half nothing synchronises and the B operand — gathered straight from the feature
    // tables, (pp, f) = (patch position, feature) of this lane's k advanced incrementally — runs four chunks ahead.  Channel order
    // node, time, edge, cooc: the edge gathers are issued before the time channel computes its cosines.
    {
        const int tok = 16 * tt + c;
        const bool tv = tok < T;
        const int pos0 = tv ? (tok < Ts ? tok * a.P : SsA + (tok - Ts) * a.P) : 0;
        const int P = a.P;
        // Round 3: the whole phase is written WITHOUT branches around loads.  hipcc counts the loads in flight (s_waitcnt vmcnt / lgkmcnt (N))
        // only while every path through the code issues the same loads: with the earlier form — gathers skipped for absent rows, the
        // fragments of a step read "if fresh", cursor rows re-read on a wrap — every step waited `vmcnt(1)` for a gather issued one step
        // before (queue depth 8 on paper) and `lgkmcnt(3)` for the fragment reads of the NEXT step just issued: the matrix pipe ran at 50 %
        // (profiles/r03_lastfm_phase.txt).  Now absent rows are read from a zero word, cursors advance by selects with the next position's
        // row read one step ahead, and the step count of every loop body is a template parameter.
        constexpr int GS = 4;              // steps per group = gathered operands in flight per lane
        static_assert(GS == 4, "the counted wait below is written as vmcnt(4)");
        using std::integral_constant;
        // ---- slot walk.  The channels' k-chunks occupy consecutive slots of the fragment sequence, every channel padded to whole groups
        // of GS slots (build_proj); a half holds hc (a multiple of GS) slots, so a group never straddles a half.
        int pj_half = 0, pj_slot = 0, pj_next = 2, pj_young = 0;      // pj_young: gathers this wave issued since its last LDS-DMA
        auto frag_ptr = [&]() -> const float* { return slabl + (size_t)((pj_half * hc + pj_slot) * 4) * kFrag; };
        // A group is done (its last fragment read is issued).  At the end of a half: this wave's DMAs of the NEXT half have landed — they are
        // older than its last GS gathers, vmcnt retires in issue order, so `vmcnt(GS)` proves it without waiting for the gathers in flight —
        // its own reads of the finished half are complete, one barrier, and the finished half's buffer is refilled two halves ahead.
        auto end_group = [&](bool counted) {
            pj_slot += GS;
            if (pj_slot == hc) {
                if (counted && pj_young >= GS) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
                __syncthreads();
                load_half(pj_next++);
                pj_young = 0; pj_half ^= 1; pj_slot = 0;
            }
        };
        auto run_idle = [&](int n) { for (int i = 0; i < (n + GS - 1) / GS; ++i) end_group(false); };     // a wave without work in this channel keeps the barriers
        // fragments of one step: 4 tiles
        auto read_frags = [&](f4 (&dst)[4], const float* fr) {
#pragma unroll
            for (int v = 0; v < 4; ++v) dst[v] = lds4(fr + v * kFrag);
        };
        // fragment reads of step u of an NS-step group: the next step's, or — on the last step of a full group, after the half protocol —
        // the first step's of the NEXT group into fa[0] (a channel's last group reads ahead in vain: the next channel starts FRESH)
        auto frags_ahead = [&](auto NSc, int u, const float* fr, f4 (&fa)[2][4], bool counted) {
            constexpr int NS = decltype(NSc)::value;
            if (u + 1 < NS) read_frags(fa[(u + 1) & 1], fr + (size_t)(u + 1) * 4 * kFrag);
            else {
                end_group(counted);
                if (NS == GS) read_frags(fa[0], frag_ptr());
            }
        };

        // ---- gathered channels (node, edge features): (pp, f) = patch position and feature of this lane's k, row = the table row of that
        // position, rown = the row of position pp + 1 (read from LDS one step ahead, every step: no branch)
        struct Cursor { int pp, f, row, rown; };
        auto row_at = [&](const int32_t* idx, int pp) -> int {
            const int32_t r = idx[pos0 + (pp < P ? pp : P - 1)];
            return (tv && pp < P) ? (r < 0 ? 0 : r) : -1;
        };
        auto cur_init = [&](Cursor& cu, const int32_t* idx) { cu.pp = 0; cu.f = 4 * g; cu.row = row_at(idx, 0); cu.rown = row_at(idx, 1); };
        auto gather = [&](const float* table, int F, Cursor& cu, const int32_t* idx) -> f4 {
            // DyGFormer.py:259-261; an absent position reads the zero word.  The select is arithmetic on the address (a ?: on the pointers
            // comes back as a branch around the address computation, which would end the scheduling region of the step)
            const uintptr_t pz = reinterpret_cast<uintptr_t>(g_zero16);
            const uintptr_t pt = reinterpret_cast<uintptr_t>(table + (size_t)(cu.row >= 0 ? cu.row : 0) * F + cu.f);
            const f4 v = ldg4(reinterpret_cast<const float*>(pz + ((pt - pz) & (cu.row >= 0 ? ~uintptr_t(0) : uintptr_t(0)))));
            ++pj_young;
            cu.f += 16;
            const bool wrap = cu.f >= F;
            cu.f = wrap ? cu.f - F : cu.f;
            cu.pp += wrap ? 1 : 0;
            cu.row = wrap ? cu.rown : cu.row;
            cu.rown = row_at(idx, cu.pp + 1);
            return v;
        };
        auto prefill = [&](f4 (&bq)[GS], Cursor& cu, const float* table, const int32_t* idx, int F) {
            cur_init(cu, idx);
#pragma unroll
            for (int u = 0; u < GS; ++u) bq[u] = gather(table, F, cu, idx);
        };
        auto g_group = [&](auto L0c, auto NSc, auto FRESHc, f4 (&fa)[2][4], f4 (&bq)[GS], Cursor& cu, const float* table, const int32_t* idx, int F) {
            constexpr int L0 = decltype(L0c)::value, NS = decltype(NSc)::value;
            const float* fr = frag_ptr();
            if (decltype(FRESHc)::value) read_frags(fa[0], fr);
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const f4 bcur = bq[u];
                bq[u] = gather(table, F, cu, idx);             // chunk + GS (zeros beyond the patch)
                frags_ahead(NSc, u, fr, fa, true);
                // operand loads of later steps stay issued ABOVE this step's MFMAs.  (Measured and not kept: the gather and the cursor arithmetic
                // scheduled into the shadow of the step's own MFMAs by sched_group_barrier, as the time channel does with its cosines —
                // node / edge channel 126 k -> 138 k cycles per 86 chunks at L = 512: the address arithmetic is short enough for the SIMD's
                // other wave to cover, and spreading it stretches the wave's MFMA block.)
                __builtin_amdgcn_sched_barrier(0);
                mma_group<4>(&x[L0], fa[u & 1], bcur);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto run_gathered = [&](auto L0c, f4 (&bq)[GS], Cursor& cu, int n, const float* table, const int32_t* idx, int F) {
            f4 fa[2][4];
            const int ng = n / GS, rem = n % GS;
            if (ng > 0) {
                g_group(L0c, integral_constant<int, GS>{}, std::true_type{}, fa, bq, cu, table, idx, F);
                for (int i = 1; i < ng; ++i) g_group(L0c, integral_constant<int, GS>{}, std::false_type{}, fa, bq, cu, table, idx, F);
            }
            if (rem == 1) g_group(L0c, integral_constant<int, 1>{}, std::true_type{}, fa, bq, cu, table, idx, F);
            else if (rem == 2) g_group(L0c, integral_constant<int, 2>{}, std::true_type{}, fa, bq, cu, table, idx, F);
            else if (rem == 3) g_group(L0c, integral_constant<int, 3>{}, std::true_type{}, fa, bq, cu, table, idx, F);
        };

        // ---- time encoding (modules.py:27-39, DyGFormer.py:263-266): the cursor runs one chunk ahead of the MFMAs; (valid, dt) of the next patch
        // position and the next chunk's w / b are read one step ahead like the gather rows.  What the channel costs beyond its MFMAs is the
        // instruction count of the cosines: on this chip a VALU instruction does not issue in the shadow of an fp32 MFMA — not of the same wave,
        // not of the SIMD's other wave (tools/coissue_ubench.hip: 16 MFMAs + 64 v_fma_f32 take the SUM of their times, 1 or 2 waves per SIMD) —
        // so interleaving them (sched_group_barrier) bought nothing; the cosines are evaluated two at a time on packed fp32 instructions
        struct TCur { int pp, f; float dt; bool ok; int32_t idn; float dn; f4 w, bb; };      // idn, dn: id and dt of position pp + 1 as read from LDS; w, bb: encoder weights / biases of features f .. f+3
        auto tpos_at = [&](int pp, float& dt, bool& ok) {
            const int q = pos0 + (pp < P ? pp : P - 1);
            const int32_t id = ids[q];
            const float d = dts[q];
            ok = tv && pp < P && id > 0;                                                             // DyGFormer.py:266
            dt = ok ? d : 0.f;
        };
        // the reads of the next position are issued here and USED by the next call: nothing in a step waits for an LDS read of its own
        auto t_advance = [&](TCur& tc) {
            tc.f += 16;
            const bool wrap = tc.f >= a.Ft;
            tc.f = wrap ? tc.f - a.Ft : tc.f;
            const bool okn = tv && tc.pp + 1 < P && tc.idn > 0;
            tc.dt = wrap ? (okn ? tc.dn : 0.f) : tc.dt;
            tc.ok = wrap ? okn : tc.ok;
            tc.pp += wrap ? 1 : 0;
            const int q = pos0 + (tc.pp + 1 < P ? tc.pp + 1 : P - 1);
            tc.idn = ids[q];
            tc.dn = dts[q];
            tc.w = lds4(tws + tc.f);
            tc.bb = lds4(tws + a.Ft + tc.f);
        };
        auto t_finish = [&](const bool ok, const f4 arg, f4 cs) -> f4 {      // rare: an argument beyond the fast cosine's range takes libm's
            if (!(fabsf(arg.x) <= 3.0e7f && fabsf(arg.y) <= 3.0e7f && fabsf(arg.z) <= 3.0e7f && fabsf(arg.w) <= 3.0e7f)) {
                cs.x = cos_time(arg.x); cs.y = cos_time(arg.y); cs.z = cos_time(arg.z); cs.w = cos_time(arg.w);
            }
            return ok ? cs : zero4();
        };
        auto t_group = [&](auto L0c, auto NSc, auto FRESHc, f4 (&fa)[2][4], f4& bnx, TCur& tc) {
            constexpr int L0 = decltype(L0c)::value, NS = decltype(NSc)::value;
            const float* fr = frag_ptr();
            if (decltype(FRESHc)::value) read_frags(fa[0], fr);
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const f4 bcur = bnx;
                const f4 w = tc.w, bb = tc.bb;                  // of the next chunk (read during the previous step)
                frags_ahead(NSc, u, fr, fa, false);
                __builtin_amdgcn_sched_barrier(0);
                const f2 dt2 = {tc.dt, tc.dt};
                const f2 a01 = pk_fma(dt2, f2{w.x, w.y}, f2{bb.x, bb.y}), a23 = pk_fma(dt2, f2{w.z, w.w}, f2{bb.z, bb.w});
                const f2 c01 = cos_time_fast2(a01), c23 = cos_time_fast2(a23);
                const f4 arg = {a01.x, a01.y, a23.x, a23.y}, cs = {c01.x, c01.y, c23.x, c23.y};
                const bool okc = tc.ok;
                t_advance(tc);                                 // the cursor arithmetic and the next position's (valid, dt) reads: same region
                mma_group<4>(&x[L0], fa[u & 1], bcur);
                __builtin_amdgcn_sched_barrier(0);
                bnx = t_finish(okc, arg, cs);
            }
        };
        auto run_time = [&](auto L0c, int n) {
            f4 fa[2][4];
            TCur tc{0, 4 * g, 0.f, false, 0, 0.f, zero4(), zero4()};
            tpos_at(0, tc.dt, tc.ok);
            { const int q = pos0 + (1 < P ? 1 : P - 1); tc.idn = ids[q]; tc.dn = dts[q]; }
            f4 bnx;
            {   // chunk 0 (not overlapped)
                const f4 w = lds4(tws + tc.f), bb = lds4(tws + a.Ft + tc.f);
                f4 arg;
                arg.x = fmaf(tc.dt, w.x, bb.x); arg.y = fmaf(tc.dt, w.y, bb.y); arg.z = fmaf(tc.dt, w.z, bb.z); arg.w = fmaf(tc.dt, w.w, bb.w);
                f4 cs;
                cs.x = cos_time_fast(arg.x); cs.y = cos_time_fast(arg.y); cs.z = cos_time_fast(arg.z); cs.w = cos_time_fast(arg.w);
                bnx = t_finish(tc.ok, arg, cs);
                t_advance(tc);
            }
            const int ng = n / GS, rem = n % GS;
            if (ng > 0) {
                t_group(L0c, integral_constant<int, GS>{}, std::true_type{}, fa, bnx, tc);
                for (int i = 1; i < ng; ++i) t_group(L0c, integral_constant<int, GS>{}, std::false_type{}, fa, bnx, tc);
            }
            if (rem == 1) t_group(L0c, integral_constant<int, 1>{}, std::true_type{}, fa, bnx, tc);
            else if (rem == 2) t_group(L0c, integral_constant<int, 2>{}, std::true_type{}, fa, bnx, tc);
            else if (rem == 3) t_group(L0c, integral_constant<int, 3>{}, std::true_type{}, fa, bnx, tc);
        };

        // ---- co-occurrence features (DyGFormer.py:395-415): k = 50*pp + j is not 4-aligned per position, so every element finds its own
        // (pp, j); k/50 by multiply-shift (exact for k < 12000).  The two LUT rows' values of a chunk are loaded two steps ahead (L2 round trips).
        struct CQ { f4 u, v; };
        int kco = 4 * g;
        auto cooc_issue = [&]() -> CQ {
            CQ r;
#pragma unroll
            for (int t = 0; t < 4; t += 2) {             // k and 50 are even: the pair (k, k + 1) lies inside one position, its LUT address is 8-byte aligned
                const int k = kco + t;
                const int pp = (k * 1311) >> 16;
                const bool ok = tv && pp < P;
                const int q = pos0 + (pp < P ? pp : P - 1);
                const int j = k - pp * kC;
                const uintptr_t pz = reinterpret_cast<uintptr_t>(g_zero16), m = ok ? ~uintptr_t(0) : uintptr_t(0);
                const uintptr_t p0 = reinterpret_cast<uintptr_t>(a.lut + (size_t)c0[q] * kC + j), p1 = reinterpret_cast<uintptr_t>(a.lut + (size_t)c1[q] * kC + j);
                const f2 v0 = *reinterpret_cast<const f2*>(pz + ((p0 - pz) & m));                   // DyGFormer.py:409-411
                const f2 v1 = *reinterpret_cast<const f2*>(pz + ((p1 - pz) & m));
                r.u[t] = v0.x; r.u[t + 1] = v0.y; r.v[t] = v1.x; r.v[t + 1] = v1.y;
            }
            kco += 16;
            return r;
        };
        auto c_group = [&](auto L0c, auto NSc, auto FRESHc, f4 (&fa)[2][4], CQ (&cq)[2]) {
            constexpr int L0 = decltype(L0c)::value, NS = decltype(NSc)::value;
            const float* fr = frag_ptr();
            if (decltype(FRESHc)::value) read_frags(fa[0], fr);
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const f4 bcur = cq[u & 1].u + cq[u & 1].v;
                cq[u & 1] = cooc_issue();                       // chunk + 2
                frags_ahead(NSc, u, fr, fa, false);
                __builtin_amdgcn_sched_barrier(0);
                mma_group<4>(&x[L0], fa[u & 1], bcur);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto run_cooc = [&](auto L0c, int n) {
            f4 fa[2][4];
            CQ cq[2];
            cq[0] = cooc_issue();
            cq[1] = cooc_issue();
            const int ng = n / GS, rem = n % GS;
            if (ng > 0) {
                c_group(L0c, integral_constant<int, GS>{}, std::true_type{}, fa, cq);
                for (int i = 1; i < ng; ++i) c_group(L0c, integral_constant<int, GS>{}, std::false_type{}, fa, cq);
            }
            if (rem == 1) c_group(L0c, integral_constant<int, 1>{}, std::true_type{}, fa, cq);
            else if (rem == 2) c_group(L0c, integral_constant<int, 2>{}, std::true_type{}, fa, cq);
            else if (rem == 3) c_group(L0c, integral_constant<int, 3>{}, std::true_type{}, fa, cq);
        };

        // Channel order node, time, edge, cooc: the edge gathers are issued before the time channel computes its cosines.  A wave whose
        // tile is empty, or a source tile shared with the first pair (f4), only keeps the barriers of the channel.
        const bool work = active && !src_shared;
        f4 bq[GS];
        Cursor cu;
        if (work) prefill(bq, cu, a.node_feat, ids, a.Fn);
        TACC(T_PROJ);
        if (work) run_gathered(integral_constant<int, 0>{}, bq, cu, a.nchunk[0], a.node_feat, ids, a.Fn); else run_idle(a.nchunk[0]);
        if (work) prefill(bq, cu, a.edge_feat, eids, a.Fe);            // in flight while the time channel runs
        TACC(T_PNODE);
        if (work) run_time(integral_constant<int, 6>{}, a.nchunk[2]); else run_idle(a.nchunk[2]);
        TACC(T_PTIME);
        if (work) run_gathered(integral_constant<int, 3>{}, bq, cu, a.nchunk[1], a.edge_feat, eids, a.Fe); else run_idle(a.nchunk[1]);
        TACC(T_PEDGE);
        if (active) run_cooc(integral_constant<int, 9>{}, a.nchunk[3]); else run_idle(a.nchunk[3]);
        TACC(T_PCOOC);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // gathers issued past the end of a channel
    }
    TACC(T_PROJ);
    __syncthreads();     // everyone is done with the window arrays and the slab
    if (NP == 2 && paired) {
        // f4: source tiles of the second pair take rows 0 .. 149 (node, edge, time channels: tiles 0 .. 8 whole, tile 9 rows 144 .. 149) of the
        // first pair's same tile through the (now free) K/V region; [donor tile tt][x tile i][lane] float4
        f4* xch = reinterpret_cast<f4*>(lds);
        if (donor) {
#pragma unroll
            for (int i = 0; i < 10; ++i) xch[(tt * 10 + i) * 64 + lane] = x[i];
        }
        __syncthreads();
        if (src_shared) {
#pragma unroll
            for (int i = 0; i < 9; ++i) x[i] = xch[(tt * 10 + i) * 64 + lane];
            const f4 v = xch[(tt * 10 + 9) * 64 + lane];       // tile 9 = rows 144 + 4 g + r: time channel up to row 149
            if (g == 0) x[9] = v;
            else if (g == 1) { x[9].x = v.x; x[9].y = v.y; }
        }
        __syncthreads();
    }
    // K, V and the slack behind them: rows of absent tokens are read as MFMA operands and must be finite
    for (int i = tid; i < kLdsRing / 4; i += NTHR) reinterpret_cast<f4*>(lds)[i] = zero4();
    tap_store<TPW>(x, a.tap_enc, b, a.Tmax, T, tt, c, g);

    float* Kb = lds + kLdsK;
    float* Vb = lds + kLdsV;
    float* misc = lds + kLdsMisc;
    const int64_t trow = b * T + 16 * tt + c;                  // training: this lane's dense activation row
    const bool tokv = pair_ok && 16 * tt + c < T;

    if (a.slab_in_ring) {        // the ring was the projection slab until now: start the layer stream (one exposed DMA latency).  Outside the
        ws.open(a.stream, kLdsRing, lane, wave, a.nstages, NW);  // layer loop: inside it the compiler kept the eight DMA addresses live (spilled)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (int l = 0; l < a.NL; ++l) {
        const LayerP& W = a.layer[l];
        TACC(T_MISC);
        float* b1s = misc + kMiscB1 + (l & 1) * kHid;
        for (int i = tid; i < kHid; i += NTHR) b1s[i] = W.b1[i];
        if (l == 0) __syncthreads();     // the re-zeroing of K/V above is complete before the first K/V rows are written

        f4 xn[kNT];
        float ln_mean = 0.f, ln_rstd = 0.f;
        if constexpr (TR) { if (l == 0) store_rows(a.tr.X[0], trow, x, g, tokv); }      // X[l + 1] leaves with the FFN's residual add
        ws.fit(2);                       // LN0 gamma, beta: two vector fragments
        if (active) layernorm(xn, x, lds + kLdsRing + ws.pos * kFrag, lds + kLdsRing + (ws.pos + 1) * kFrag, g, ln_mean, ln_rstd);
        ws.advance(2);
        f4 ao[kNT];                      // training: the out-projection sum of both heads (dropout applies to the finished sum); unused otherwise
        if constexpr (TR) {
            store_rows(a.tr.layer[l].xn0, trow, xn, g, tokv);
            if (tokv && g == 0) { a.tr.layer[l].m0[trow] = ln_mean; a.tr.layer[l].r0[trow] = ln_rstd; }
#pragma unroll
            for (int i = 0; i < kNT; ++i) ao[i] = zero4();
        }
        auto& xo = [&]() -> f4 (&)[kNT] { if constexpr (TR) return ao; else return x; }();      // where the out-projection accumulates
        float xk0 = 0.f, xk1 = 0.f;      // LN(x) rows 192..199 as the two packed B operands of the last k-chunk
        if (F3_KSKIP) kpack(xn[kKC - 1], xk0, xk1);
        TACC(T_LN);

#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            // ================= Q, K, V of head h =================
            f4 qa[7];
            {
                ws.fit(1);               // bias fragment: rows 100h .. 100h+99 of the q bias; elements 100 .. 107: the k and v bias of rows 96 .. 99
                const float* bq = lds + kLdsRing + ws.pos * kFrag + 4 * g;
#pragma unroll
                for (int j = 0; j < 7; ++j) qa[j] = lds4(bq + 16 * j);
                ws.advance(1);
                qkv_group(qa, xn, xk0, xk1, ws, ringl, active);
                // Tile 6 is the COMBINED tile of the head (FragDesc kmode 8): lane group 0 holds rows 96 .. 99 of Q^T, group 1 those of K^T, group 2
                // those of V^T (the K and V parts below run 6 tiles instead of 7: 9.5 % of the layer's QKV MFMAs).  Every wave passed a stream
                // barrier since its last read of the previous head's K / V (the out-projection and this group lie in between): their rows
                // can be written.
                if (active && (g == 1 || g == 2)) *reinterpret_cast<f4*>((g == 2 ? Vb : Kb) + (tokbase + 16 * tt + c) * kKV + 96) = qa[6];
                if constexpr (TR) {
                    if (tokv && (g == 1 || g == 2)) *reinterpret_cast<f4*>(a.tr.layer[l].qkv + trow * (3 * kD) + g * kD + kHD * h + 96) = qa[6];
                }
                if (g != 0) qa[6] = zero4();          // rows 100 .. 111 of Q^T do not exist
                if constexpr (TR) {
                    if (tokv) {
                        float* qp = a.tr.layer[l].qkv + trow * (3 * kD) + kHD * h + 4 * g;
#pragma unroll
                        for (int j = 0; j < 7; ++j)
                            if (j < 6 || g == 0) *reinterpret_cast<f4*>(qp + 16 * j) = qa[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 7; ++j) qa[j] = qa[j] * a.qscale;
            }
#pragma unroll 1
            for (int kv = 0; kv < 2; ++kv) {
                f4 acc[6];               // rows 0 .. 95 of K^T / V^T (rows 96 .. 99 came out of the combined tile above)
                ws.fit(1);
                const float* bk = lds + kLdsRing + ws.pos * kFrag + 4 * g;
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[j] = lds4(bk + 16 * j);
                ws.advance(1);
                qkv_group(acc, xn, xk0, xk1, ws, ringl, active);
                if (active) {
                    float* row = (kv ? Vb : Kb) + (tokbase + 16 * tt + c) * kKV + 4 * g;
#pragma unroll
                    for (int j = 0; j < 6; ++j) *reinterpret_cast<f4*>(row + 16 * j) = acc[j];
                }
                if constexpr (TR) {
                    if (tokv) {
                        float* kp = a.tr.layer[l].qkv + trow * (3 * kD) + (kv + 1) * kD + kHD * h + 4 * g;
#pragma unroll
                        for (int j = 0; j < 6; ++j) *reinterpret_cast<f4*>(kp + 16 * j) = acc[j];
                    }
                }
            }
            TACC(T_QKV);
            __syncthreads();
            TACC(T_QKVBAR);

            // ================= attention of head h for this wave's 16 queries =================
            f4 oa[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) oa[j] = zero4();
            if (active) {
                f4 sa[TPW];
#pragma unroll
                for (int kt = 0; kt < TPW; ++kt) sa[kt] = zero4();
                const float q6 = F3_KSKIP ? kpack4(qa[6]) : 0.f;       // rows 96 .. 99 of Q^T for the one-MFMA last d-chunk
                // S^T[key][query] = sum_d K[key][d] * Q^T[d][query]; key tiles in chunks of 4
#pragma unroll
                for (int kh = 0; kh < TPW / 4; ++kh) {
                    const float* kbase = Kb + (tokbase + 64 * kh + c) * kKV + 4 * g;
                    f4 kf[2][4];
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) kf[0][kt] = lds4(kbase + 16 * kt * kKV);
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        if (j + 1 < 6 || (!F3_KSKIP && j + 1 < 7)) {
#pragma unroll
                            for (int kt = 0; kt < 4; ++kt) kf[(j + 1) & 1][kt] = lds4(kbase + 16 * kt * kKV + 16 * (j + 1));
                        } else if (j + 1 == 6) {     // d = 96 .. 99 in ONE MFMA: lane group g reads K[key][96 + g] (kbase points at column 4 g)
#pragma unroll
                            for (int kt = 0; kt < 4; ++kt) kf[0][kt].x = kbase[16 * kt * kKV + 96 - 3 * g];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (F3_KSKIP && j == 6) mma_group1<4>(&sa[4 * kh], kf[0], q6); else mma_group<4>(&sa[4 * kh], kf[j & 1], qa[j]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // softmax over keys (rows 16kt + 4g + r); keys >= T do not exist
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < TPW; ++kt)
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
                for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sa[kt][r] = __expf(sa[kt][r] - mx); sum += sa[kt][r]; }
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                const float inv = 1.0f / sum;
#pragma unroll
                for (int kt = 0; kt < TPW; ++kt) sa[kt] *= inv;
                if constexpr (TR) {
                    // probabilities of query 16 tt + c over keys 16 kt + 4 g + r: row (b H + h) T + query of the [B H][T][T] buffers; the
                    // dropout of nn.MultiheadAttention acts on them (mask index = offset in that buffer)
                    const int64_t prow = ((b * 2 + h) * (int64_t)T + 16 * tt + c) * T;
                    const uint32_t sk = a.tr.dr.site_key((uint32_t)(4 * l + 0));
                    float* const Pp = a.tr.layer[l].P + prow;
                    float* const Pdp = a.tr.layer[l].Pd + prow;
                    const bool vec = (T & 3) == 0;           // rows of T floats: float4 stores need T % 4 == 0 (wave-uniform)
#pragma unroll
                    for (int kt = 0; kt < TPW; ++kt) {
                        const int key0 = 16 * kt + 4 * g;
                        const f4 pv = sa[kt];
#pragma unroll
                        for (int r = 0; r < 4; ++r) sa[kt][r] *= a.tr.dr.mask32(sk, (uint32_t)prow + key0 + r);
                        if (vec) {
                            if (tokv && key0 < T) { *reinterpret_cast<f4*>(Pp + key0) = pv; *reinterpret_cast<f4*>(Pdp + key0) = sa[kt]; }
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (tokv && key0 + r < T) { Pp[key0 + r] = pv[r]; Pdp[key0 + r] = sa[kt][r]; }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // O^T[d][query] = sum_key V[key][d] * P^T[key][query]  (rows >= 100 are junk x zero weight columns)
                {
                    // V tiles as the A operand: sub-steps (key tile kt, d-tiles 0..3 | 4..6), read one sub-step ahead (8 fragments live, not 14)
                    auto load_v = [&](f4 (&va)[4], int kt, int j0, int n) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (j < n) {
                                const float* vp = Vb + (tokbase + 16 * kt + 4 * g) * kKV + 16 * (j0 + j) + c;
                                va[j].x = vp[0]; va[j].y = vp[kKV]; va[j].z = vp[2 * kKV]; va[j].w = vp[3 * kKV];
                            }
                        }
                    };
                    f4 va[2][4];
                    load_v(va[0], 0, 0, 4);
#pragma unroll
                    for (int st = 0; st < 2 * TPW; ++st) {
                        const int kt = st >> 1, half = st & 1;
                        if (st + 1 < 2 * TPW) load_v(va[(st + 1) & 1], (st + 1) >> 1, ((st + 1) & 1) ? 4 : 0, ((st + 1) & 1) ? 3 : 4);
                        __builtin_amdgcn_sched_barrier(0);
                        if (half == 0) mma_group<4>(&oa[0], va[st & 1], sa[kt]); else mma_group<3>(&oa[4], va[st & 1], sa[kt]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if constexpr (TR) {
                if (tokv) {
                    float* op = a.tr.layer[l].oa + trow * kD + kHD * h + 4 * g;
#pragma unroll
                    for (int j = 0; j < 7; ++j)
                        if (j < 6 || g == 0) *reinterpret_cast<f4*>(op + 16 * j) = oa[j];
                }
            }
            TACC(T_ATTN);
            // ================= out-projection, accumulated straight into the residual: x^T += Wo[:, head h] . O^T =================
            // 7 steps (d-chunk j) of 13 fragments (n-tile i), sub-groups (4,3,3,3) read one ahead
            ws.fit(13);
            {
                f4 fs[2][4];
                float o6 = 0.f;              // rows 96 .. 99 of O^T for the one-MFMA last d-chunk
                if (active) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) fs[0][v] = lds4(ringl + (ws.pos + v) * kFrag);
                    if (F3_KSKIP) o6 = kpack4(oa[6]);
                }
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const int pcur = ws.pos;
                    const int pnext = ws.next_pos(13, 13);
                    if (active) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int gi = 4 * j + q;
                            const int i0 = q == 0 ? 0 : 4 + 3 * (q - 1), n = q == 0 ? 4 : 3;
                            if (q + 1 < 4) {
                                const int j0 = 4 + 3 * q;
#pragma unroll
                                for (int v = 0; v < 3; ++v) fs[(gi + 1) & 1][v] = lds4(ringl + (pcur + j0 + v) * kFrag);
                            } else if (j + 1 < 7) {
#pragma unroll
                                for (int v = 0; v < 4; ++v) fs[(gi + 1) & 1][v] = lds4(ringl + (pnext + v) * kFrag);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            if (F3_KSKIP && j == 6) { if (n == 4) mma_group1<4>(&xo[i0], fs[gi & 1], o6); else mma_group1<3>(&xo[i0], fs[gi & 1], o6); }
                            else if (n == 4) mma_group<4>(&xo[i0], fs[gi & 1], oa[j]); else mma_group<3>(&xo[i0], fs[gi & 1], oa[j]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    ws.advance(13);
                    if (j + 1 < 7) ws.fit(13);
                }
            }
            TACC(T_OPROJ);
        }
        // out-projection bias (one vector fragment)
        ws.fit(1);
        {
            const float* bo = lds + kLdsRing + ws.pos * kFrag + 4 * g;
            if constexpr (TR) {          // x1 = x + dropout(Wo O + bo), DyGFormer.py:456; x re-read from the rows stored at the layer's start
                residual_dropped(x, a.tr.X[l], a.tr.layer[l].x1, ao, bo, a.tr.dr, (uint32_t)(4 * l + 1), trow, g, tokv);
            } else {
#pragma unroll
                for (int i = 0; i < kNT; ++i) x[i] = x[i] + lds4(bo + 16 * i);
            }
        }
        ws.advance(1);

        TACC(T_OPROJ);
        // ================= LN1 + FFN: 25 steps of two hidden tiles; W1 fragments [k-chunk][tile], W2 [tile][n-tile] =================
        ws.fit(2);
        if (active) layernorm(xn, x, lds + kLdsRing + ws.pos * kFrag, lds + kLdsRing + (ws.pos + 1) * kFrag, g, ln_mean, ln_rstd);
        ws.advance(2);
        if (F3_KSKIP) kpack(xn[kKC - 1], xk0, xk1);
        auto& f2 = xo;                   // where the second FFN product accumulates
        if constexpr (TR) {
            store_rows(a.tr.layer[l].xn1, trow, xn, g, tokv);
            if (tokv && g == 0) { a.tr.layer[l].m1[trow] = ln_mean; a.tr.layer[l].r1[trow] = ln_rstd; }
#pragma unroll
            for (int i = 0; i < kNT; ++i) ao[i] = zero4();
        }
        const uint32_t sk2 = TR ? a.tr.dr.site_key((uint32_t)(4 * l + 2)) : 0u;
        TACC(T_LN);
        ws.align26();
        // W1(p) | W2(p) per step; W2 accumulates straight into the residual registers (no separate FFN accumulator: 52 VGPRs fewer)
        constexpr bool XB = F3_XBAR && !TR;      // (the training forward has no registers to spare)
        f4 pre1[2];                      // first fragments of the next W1 block, read across the stage barrier in front of it
        if (XB && active) { pre1[0] = lds4(ringl + ws.pos * kFrag); pre1[1] = lds4(ringl + (ws.pos + 1) * kFrag); }
#pragma unroll 1
        for (int p = 0; p < 25; ++p) {
            f4 h[2];
            if (active) {
                ffn_w1(h, xn, xk0, xk1, ringl + ws.pos * kFrag, b1s + 32 * p, g, XB ? pre1 : nullptr);
                TACC(T_F_W1);
                if constexpr (TR) {
                    if (tokv) {
                        float* hp = a.tr.layer[l].hpre + trow * kHid + 32 * p + 4 * g;
                        *reinterpret_cast<f4*>(hp) = h[0]; *reinterpret_cast<f4*>(hp + 16) = h[1];
                    }
                }
                gelu_tiles(h);
                if constexpr (TR) {      // dropout on the activation (DyGFormer.py:458): element (row, hidden unit n) draws mask(site, row * 800 + n)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[u][r] *= a.tr.dr.mask32(sk2, (uint32_t)trow * kHid + 32 * p + 16 * u + 4 * g + r);
                    if (tokv) {
                        float* hp = a.tr.layer[l].hact + trow * kHid + 32 * p + 4 * g;
                        *reinterpret_cast<f4*>(hp) = h[0]; *reinterpret_cast<f4*>(hp + 16) = h[1];
                    }
                }
                TACC(T_F_GELU);
            }
            f4 pre2[4];                  // first fragments of this step's W2 block (blocks sit at ring positions 0 / 26)
            if (XB && F3_XBAR > 1 && active) {
                const float* nb = ringl + (ws.pos == 0 ? 26 : 0) * kFrag;
#pragma unroll
                for (int v = 0; v < 4; ++v) pre2[v] = lds4(nb + (size_t)v * kFrag);
            }
            ws.advance(26, (TR && active) ? 4 : 0);      // training: the four hpre / hact stores of this step stay in flight through the W2 block
            TACC(T_F_ADV1);
            if (active) {
                ffn_w2(f2, h, ringl + ws.pos * kFrag, (XB && F3_XBAR > 1) ? pre2 : nullptr);
                if (XB && p + 1 < 25) {
                    const float* nb = ringl + (ws.pos == 0 ? 26 : 0) * kFrag;       // the W1 block of the next step
                    pre1[0] = lds4(nb); pre1[1] = lds4(nb + kFrag);
                }
            }
            TACC(T_F_W2);
            ws.advance(26);
            TACC(T_F_ADV2);
        }
        ws.fit(1);
        {
            const float* b2 = lds + kLdsRing + ws.pos * kFrag + 4 * g;
            if constexpr (TR) {          // x_{l+1} = x1 + dropout(W2 h + b2), DyGFormer.py:460; x1 re-read from its rows (not kept through the FFN)
                residual_dropped(x, a.tr.layer[l].x1, a.tr.X[l + 1], f2, b2, a.tr.dr, (uint32_t)(4 * l + 3), trow, g, tokv);
            } else {
#pragma unroll
                for (int i = 0; i < kNT; ++i) x[i] = x[i] + lds4(b2 + 16 * i);
            }
        }
        ws.advance(1);
        TACC(T_FFN);
        tap_store<TPW>(x, a.tap_layer[l], b, a.Tmax, T, tt, c, g);
    }

    TACC(T_MISC);
    // ================= per-side mean over tokens + output layer (DyGFormer.py:181-192) =================
    __syncthreads();        // K/V are dead: reuse as scratch
    {
        float* pool = Kb;                        // [wave][side][208]
        const int tok = 16 * tt + c;
        const bool in_src = tok < Ts, in_dst = tok >= Ts && tok < T;
#pragma unroll
        for (int i = 0; i < kNT; ++i) {
            f4 vs = in_src ? x[i] : zero4();
            f4 vd = in_dst ? x[i] : zero4();
#pragma unroll
            for (int r = 0; r < 4; ++r) { vs[r] = row_sum16(vs[r]); vd[r] = row_sum16(vd[r]); }
            if (c == 0) {
                *reinterpret_cast<f4*>(pool + (wave * 2 + 0) * kDP + 16 * i + 4 * g) = vs;
                *reinterpret_cast<f4*>(pool + (wave * 2 + 1) * kDP + 16 * i + 4 * g) = vd;
            }
        }
        TACC(T_POOL1);
        __syncthreads();
        TACC(T_POOL2);
        // mean[col][208], col = 2*pair + side (4 columns of the 16-wide B operand are used; the rest multiply zeros)
        float* mean = Vb;
        const int Td = T - Ts;
        for (int i = ptid; i < 2 * kDP; i += PT) {
            const int side = i / kDP, n = i % kDP;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < TPW; ++w) s += pool[((pi * TPW + w) * 2 + side) * kDP + n];
            const float mv = n < kD ? s / (float)(side ? Td : Ts) : 0.f;
            mean[pi * 2 * kDP + i] = mv;
            if constexpr (TR) { if (pair_ok && n < kD) a.tr.pooled[((int64_t)side * a.B + b) * kD + n] = mv; }
        }
        __syncthreads();
        // output layer on the matrix cores: out^T[j][col] = sum_k W[j][k] mean[col][k] + b[j]; wave w owns output tiles w, w+8, ...
        // (each fragment is used by one wave only, so they come straight from global memory, all 13 of a tile in flight)
        const int ntile = (a.Fn + 15) >> 4;
        for (int jt = wave; jt < ntile; jt += NW) {
            f4 fa[kKC];
#pragma unroll
            for (int kc = 0; kc < kKC; ++kc) fa[kc] = ldg4(a.outfrag + ((size_t)jt * kKC + kc) * kFrag + lane * 4);
            const int j0 = 16 * jt + 4 * g;
            f4 acc0 = j0 < a.Fn ? ldg4(a.outb + j0) : zero4(), acc1 = zero4();
#pragma unroll
            for (int kc = 0; kc < kKC; ++kc) {
                const f4 bm = c < 2 * NP ? lds4(mean + c * kDP + 16 * kc + 4 * g) : zero4();
                if (kc & 1) { acc1 = mfma(fa[kc].x, bm.x, acc1); acc1 = mfma(fa[kc].y, bm.y, acc1); acc1 = mfma(fa[kc].z, bm.z, acc1); acc1 = mfma(fa[kc].w, bm.w, acc1); }
                else { acc0 = mfma(fa[kc].x, bm.x, acc0); acc0 = mfma(fa[kc].y, bm.y, acc0); acc0 = mfma(fa[kc].z, bm.z, acc0); acc0 = mfma(fa[kc].w, bm.w, acc0); }
            }
            const int64_t bo_ = paired ? (int64_t)blockIdx.x + (c >> 1) * a.pair_stride : (int64_t)blockIdx.x * NP + (c >> 1);
            if (c < 2 * NP && bo_ < a.B && j0 < a.Fn)
                *reinterpret_cast<f4*>(((c & 1) ? a.out_dst : a.out_src) + bo_ * a.Fn + j0) = acc0 + acc1;
        }
    }
    TACC(T_POOL);
    TSTORE();
}

// ================================================================================================
// Backward of one encoder layer's FFN block (DyGFormer.py:457-460 reversed), token-owner like the forward: a workgroup = 8 waves = 128
// dense token rows, wave w owns 16 rows x all 200 channels.  In: dX = d loss / d x_{l+1} [M][200].  Per 32-unit hidden step p the two
// activation-gradient products run register to register from the layer's BACKWARD stream (W2^T then W1^T fragments of the same ring):
//     dhact^T = W2[:, step]^T . dF2^T,   dhpre = dhact o mask2 o gelu'(hpre),   dxn1^T += W1[step, :]^T . dhpre^T
// with dF2 = dX o mask3; then LayerNorm-1 backward against the stored statistics, dX <- dX + LN1'(dxn1) in place, and the LN weight /
// bias gradients (row sums by DPP, the eight waves meet in LDS, one atomic per channel and workgroup).  dF2 and dhpre are written as dense
// rows for the grouped weight-gradient launch (k_dw_grouped, dygformer_train.hip), which also sums the bias gradients.
struct FfnBwdArgs {
    const float* stream; int nstages;
    int64_t M;
    float* dX;                                   // [M][200] in: d x_{l+1}; out: d x1
    const float *hpre, *x1, *m1, *r1;            // forward activations (dense rows)
    float *dF2, *dH;                             // [M][200], [M][800]
    float *dgamma, *dbeta;                       // LN1 (accumulated)
    train::Drop dr; uint32_t site_act, site_out;
    unsigned long long* stamps;                  // diagnostic build only
};
__device__ __forceinline__ float gelu_grad(float v) {                       // d/dv [v Phi(v)] = Phi(v) + v phi(v)
    const float cdf = 0.5f * (1.0f + erf_as(v * 0.70710678118654752440f));
    return fmaf(v * 0.39894228040143267794f, __expf(-0.5f * v * v), cdf);
}
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void k_ffn_bwd(const FfnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * (16 * NW) + 16 * wave, row = row0 + c;
    const bool active = row0 < a.M, valid = row < a.M;
    TDECL;
    WStream ws;
    ws.open(a.stream, kLdsRing, lane, wave, a.nstages, NW);
    const float* ringl = lds + kLdsRing + lane * 4;
    // dF2^T = (dX o mask3)^T: the B operand of every W2^T product of the layer
    f4 d2[kNT];
    {
        const uint32_t sk = a.dr.site_key(a.site_out), e0 = (uint32_t)row * kD + 4 * g;
        const float* src = a.dX + row * kD + 4 * g;
        float* dst = a.dF2 + row * kD + 4 * g;
#pragma unroll
        for (int i = 0; i < kNT; ++i) {
            const bool on = valid && (i < 12 || g < 2);
            f4 v = on ? ldg4(src + 16 * i) : zero4();
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= a.dr.mask32(sk, e0 + 16 * i + r);
            if (on) *reinterpret_cast<f4*>(dst + 16 * i) = v;
            d2[i] = v;
        }
    }
    float dk0 = 0.f, dk1 = 0.f;
    if (F3_KSKIP) kpack(d2[kKC - 1], dk0, dk1);
    f4 dxn[kNT];
#pragma unroll
    for (int i = 0; i < kNT; ++i) dxn[i] = zero4();
    const uint32_t sk2 = a.dr.site_key(a.site_act);
    const float* hrow = a.hpre + row * kHid + 4 * g;
    float* dhrow = a.dH + row * kHid + 4 * g;
    f4 hp[2];
    hp[0] = valid ? ldg4(hrow) : zero4();
    hp[1] = valid ? ldg4(hrow + 16) : zero4();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the first ring stages have landed
    __syncthreads();
    TACC(0);
#pragma unroll 1
    for (int p = 0; p < 25; ++p) {
        f4 dh[2];
        f4 hn[2];
        if (active) {
            // the next step's hidden pre-activations fly through this step (issued first: the two dhpre stores below are then the two
            // youngest vector-memory operations at the stage barrier)
            const bool more = valid && p + 1 < 25;
            hn[0] = more ? ldg4(hrow + 32 * (p + 1)) : zero4();
            hn[1] = more ? ldg4(hrow + 32 * (p + 1) + 16) : zero4();
            ffn_w1(dh, d2, dk0, dk1, ringl + ws.pos * kFrag, nullptr, g);
            TACC(1);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dh[u][r] *= a.dr.mask32(sk2, (uint32_t)row * kHid + 32 * p + 16 * u + 4 * g + r) * gelu_grad(hp[u][r]);
            // take the prefetched rows NOW, before the stores below are issued: vmcnt retires in order and the compiler does not see the ring's
            // DMAs, so a wait for these loads placed behind the stores would also sit out the stores' whole round trip
            hp[0] = hn[0]; hp[1] = hn[1];
            asm volatile("" : "+v"(hp[0]), "+v"(hp[1]));
        }
        TACC(2);
        ws.advance(26);
        TACC(3);
        // the dhpre rows leave AFTER the stage barrier's DMA issue: at the next barrier they are the two youngest operations and stay in flight
        // (vmcnt(2) proves the older DMAs landed), and they have both blocks of the next step to retire before a full drain
        if (valid) { *reinterpret_cast<f4*>(dhrow + 32 * p) = dh[0]; *reinterpret_cast<f4*>(dhrow + 32 * p + 16) = dh[1]; }
        if (active) ffn_w2(dxn, dh, ringl + ws.pos * kFrag);
        TACC(4);
        ws.advance(26, active ? 2 : 0);
        TACC(5);
    }
    // LayerNorm-1 backward (x1 rows and their statistics from the forward): dx1 = dX + rstd (gy - mean(gy) - xhat mean(gy xhat)), gy = dxn gamma
    ws.fit(1);
    const float* gam = lds + kLdsRing + ws.pos * kFrag + 4 * g;
    float* red = lds;                            // [8 waves][2][208] partial sums of dgamma / dbeta (the K/V region is unused here)
    {
        const float mean = valid ? a.m1[row] : 0.f, rstd = valid ? a.r1[row] : 0.f;
        const float* xr = a.x1 + row * kD + 4 * g;
        f4 xh[kNT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < kNT; ++i) {
            const bool on = valid && (i < 12 || g < 2);
            const f4 xv = on ? ldg4(xr + 16 * i) : zero4();
            const f4 gm = lds4(gam + 16 * i);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xh[i][r] = on ? (xv[r] - mean) * rstd : 0.f;
                const float gy = dxn[i][r] * gm[r];
                s1 += gy; s2 = fmaf(gy, xh[i][r], s2);
            }
        }
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        const float m1v = s1 * (1.0f / kD), m2v = s2 * (1.0f / kD);
        float* dxr = a.dX + row * kD + 4 * g;
#pragma unroll
        for (int i = 0; i < kNT; ++i) {
            const bool on = valid && (i < 12 || g < 2);
            const f4 gm = lds4(gam + 16 * i);
            f4 pg, pb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pg[r] = row_sum16(dxn[i][r] * xh[i][r]);
                pb[r] = row_sum16(dxn[i][r]);
            }
            if (c == 0) {
                *reinterpret_cast<f4*>(red + (wave * 2 + 0) * kDP + 16 * i + 4 * g) = pg;
                *reinterpret_cast<f4*>(red + (wave * 2 + 1) * kDP + 16 * i + 4 * g) = pb;
            }
            if (on) {
                f4 v = ldg4(dxr + 16 * i);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += rstd * (dxn[i][r] * gm[r] - m1v - xh[i][r] * m2v);
                *reinterpret_cast<f4*>(dxr + 16 * i) = v;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA of this workgroup is left in flight
    __syncthreads();
    TACC(6);
    for (int i = tid; i < 2 * kDP; i += 64 * NW) {
        const int which = i / kDP, n = i % kDP;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[(w * 2 + which) * kDP + n];
        if (n < kD) atomicAdd((which ? a.dbeta : a.dgamma) + n, t);
    }
    TACC(7);
    TSTORE();
}

// ================================================================================================
// Backward of one encoder layer's attention block (DyGFormer.py:440-456 reversed; nn.MultiheadAttention with dropout on the probabilities),
// token-owner like the forward: a workgroup = 8 waves = NP pairs x TPW token tiles, wave = 16 tokens x all channels.
// In: dX = d loss / d x1 [M][200] (after k_ffn_bwd).  dAo = dX o mask1 is written as rows (the out-projection's weight-gradient operand) and,
// per head h, streamed through the layer's backward ring:
//     dOa^T = Wo[:, h]^T . dAo^T                                                                    (same shape as a Q/K/V product)
//     phase A, this wave's tokens as QUERIES (K, V of the pair in LDS):  dPd^T = V . dOa^T,  dS^T = P^T o (dPd^T o mask0 - D),  dQ^T = K^T . dS^T
//     phase B, this wave's tokens as KEYS (Q, dOa of the pair in LDS over K, V):  dPd = dOa . V^T,  dS = P o (dPd o mask0 - D),
//              dV^T = dOa^T . Pd,  dK^T = Q^T . dS            — tiles [query rows][own key columns] are again MFMA B operands, so the sums over
//              the pair's queries need no cross-wave exchange beyond D (one float per query, through LDS); P and Pd are re-read from the
//              forward's [B H][T][T] buffers in either orientation
//     dxn0^T += Wq[h]^T . dQ^T + Wv[h]^T . dV^T + Wk[h]^T . dK^T                                     (same shape as the out-projection)
// then LayerNorm-0 backward, dX <- dX + LN0'(dxn0) in place = d loss / d x_l.  dQ | dK | dV leave as rows [M][600] for the grouped
// weight-gradient launch.  scale = 1/sqrt(head dim) multiplies dS once (it serves both dQ and dK: S = scale q.k).
template <int TPW>
__device__ __forceinline__ void s_like(f4 (&sa)[TPW], const float* base, const f4 (&q)[7], int c, int g) {       // sa[kt] += rows(16 kt ..)(base) . q   (forward: S^T = K Q^T)
    const float q6 = F3_KSKIP ? kpack4(q[6]) : 0.f;
#pragma unroll
    for (int kh = 0; kh < TPW / 4; ++kh) {
        const float* kbase = base + (64 * kh + c) * kKV + 4 * g;
        f4 kf[2][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) kf[0][kt] = lds4(kbase + 16 * kt * kKV);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            if (j + 1 < 6 || (!F3_KSKIP && j + 1 < 7)) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) kf[(j + 1) & 1][kt] = lds4(kbase + 16 * kt * kKV + 16 * (j + 1));
            } else if (j + 1 == 6) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) kf[0][kt].x = kbase[16 * kt * kKV + 96 - 3 * g];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (F3_KSKIP && j == 6) mma_group1<4>(&sa[4 * kh], kf[0], q6); else mma_group<4>(&sa[4 * kh], kf[j & 1], q[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
template <int TPW>
__device__ __forceinline__ void pv_like(f4 (&oa)[7], const float* base, const f4 (&p)[TPW], int c, int g) {     // oa += rows(base)^T . p   (forward: O^T = V^T P^T)
    auto load_v = [&](f4 (&va)[4], int kt, int j0, int n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < n) {
                const float* vp = base + (16 * kt + 4 * g) * kKV + 16 * (j0 + j) + c;
                va[j].x = vp[0]; va[j].y = vp[kKV]; va[j].z = vp[2 * kKV]; va[j].w = vp[3 * kKV];
            }
        }
    };
    f4 va[2][4];
    load_v(va[0], 0, 0, 4);
#pragma unroll
    for (int st = 0; st < 2 * TPW; ++st) {
        const int kt = st >> 1, half = st & 1;
        if (st + 1 < 2 * TPW) load_v(va[(st + 1) & 1], (st + 1) >> 1, ((st + 1) & 1) ? 4 : 0, ((st + 1) & 1) ? 3 : 4);
        __builtin_amdgcn_sched_barrier(0);
        if (half == 0) mma_group<4>(&oa[0], va[st & 1], p[kt]); else mma_group<3>(&oa[4], va[st & 1], p[kt]);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// acc (13 model-dim tiles) += W^T(7 d-chunks x 13 tiles, from the ring) . t (7 head-dim tiles)   (forward: the out-projection)
__device__ __forceinline__ void proj_t(f4 (&acc)[kNT], const f4 (&t)[7], WStream& ws, const float* ringl, bool active) {
    ws.fit(13);
    f4 fs[2][4];
    float t6 = 0.f;
    if (active) {
#pragma unroll
        for (int v = 0; v < 4; ++v) fs[0][v] = lds4(ringl + (ws.pos + v) * kFrag);
        if (F3_KSKIP) t6 = kpack4(t[6]);
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int pcur = ws.pos;
        const int pnext = ws.next_pos(13, 13);
        if (active) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gi = 4 * j + q;
                const int i0 = q == 0 ? 0 : 4 + 3 * (q - 1), n = q == 0 ? 4 : 3;
                if (q + 1 < 4) {
                    const int j0 = 4 + 3 * q;
#pragma unroll
                    for (int v = 0; v < 3; ++v) fs[(gi + 1) & 1][v] = lds4(ringl + (pcur + j0 + v) * kFrag);
                } else if (j + 1 < 7) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) fs[(gi + 1) & 1][v] = lds4(ringl + (pnext + v) * kFrag);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (F3_KSKIP && j == 6) { if (n == 4) mma_group1<4>(&acc[i0], fs[gi & 1], t6); else mma_group1<3>(&acc[i0], fs[gi & 1], t6); }
                else if (n == 4) mma_group<4>(&acc[i0], fs[gi & 1], t[j]); else mma_group<3>(&acc[i0], fs[gi & 1], t[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        ws.advance(13);
        if (j + 1 < 7) ws.fit(13);
    }
}
// LayerNorm backward of a token-owner wave against stored statistics (shared by the two backward kernels): dX rows += rstd (gy - mean(gy) -
// xhat mean(gy xhat)), gy = dxn gamma; the workgroup's sums of dxn xhat / dxn over its tokens go to `red` [8 waves][2][208]
__device__ __forceinline__ void ln_backward(const f4 (&dxn)[kNT], const float* xrows, const float* mean_p, const float* rstd_p, const float* gam, float* dXrows,
                                            int64_t row, bool valid, float* red, int wave, int c, int g) {
    const float mean = valid ? mean_p[row] : 0.f, rstd = valid ? rstd_p[row] : 0.f;
    const float* xr = xrows + row * kD + 4 * g;
    f4 xh[kNT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < kNT; ++i) {
        const bool on = valid && (i < 12 || g < 2);
        const f4 xv = on ? ldg4(xr + 16 * i) : zero4();
        const f4 gm = lds4(gam + 16 * i);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xh[i][r] = on ? (xv[r] - mean) * rstd : 0.f;
            const float gy = dxn[i][r] * gm[r];
            s1 += gy; s2 = fmaf(gy, xh[i][r], s2);
        }
    }
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
    const float m1v = s1 * (1.0f / kD), m2v = s2 * (1.0f / kD);
    float* dxr = dXrows + row * kD + 4 * g;
    f4 dxv[kNT];                                 // the incoming gradient rows: all loads in flight before the tile-by-tile pass
#pragma unroll
    for (int i = 0; i < kNT; ++i) dxv[i] = (valid && (i < 12 || g < 2)) ? ldg4(dxr + 16 * i) : zero4();
#pragma unroll
    for (int i = 0; i < kNT; ++i) {
        const bool on = valid && (i < 12 || g < 2);
        const f4 gm = lds4(gam + 16 * i);
        f4 pg, pb;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pg[r] = row_sum16(dxn[i][r] * xh[i][r]);
            pb[r] = row_sum16(dxn[i][r]);
        }
        if (c == 0) {
            *reinterpret_cast<f4*>(red + (wave * 2 + 0) * kDP + 16 * i + 4 * g) = pg;
            *reinterpret_cast<f4*>(red + (wave * 2 + 1) * kDP + 16 * i + 4 * g) = pb;
        }
        if (on) {
            f4 v = dxv[i];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rstd * (dxn[i][r] * gm[r] - m1v - xh[i][r] * m2v);
            *reinterpret_cast<f4*>(dxr + 16 * i) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
struct AttnBwdArgs {
    const float* stream; int nstages;
    int64_t B; int T;
    float* dX;                                   // [M][200] in: d x1; out: d x_l
    const float *X, *m0, *r0;                    // layer input rows and LN0 statistics
    const float *qkv, *P, *Pd;                   // forward activations
    float *dAo, *dQKV;                           // [M][200], [M][600]
    float *dgamma, *dbeta;                       // LN0 (accumulated)
    train::Drop dr; uint32_t site_p, site_ao;
    float qscale;
    unsigned long long* stamps;                  // diagnostic build only
};
template <int TPW, int NW = 8>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void k_attn_bwd(const AttnBwdArgs a) {
    constexpr int NP = NW / TPW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pi = wave / TPW, tt = wave % TPW;
    const int c = lane & 15, g = lane >> 4;
    const int64_t b = (int64_t)blockIdx.x * NP + pi;
    const int T = a.T;
    const bool pair_ok = b < a.B;
    const bool active = pair_ok && 16 * tt < T, valid = pair_ok && 16 * tt + c < T;
    const int tok = 16 * tt + c;
    const int64_t row = b * T + tok;
    const int tokbase = pi * (16 * TPW);
    TDECL;
    WStream ws;
    ws.open(a.stream, kLdsRing, lane, wave, a.nstages, NW);
    const float* ringl = lds + kLdsRing + lane * 4;
    float* Kb = lds + kLdsK;
    float* Vb = lds + kLdsV;
    float* Dq = lds + kLdsMisc;                  // [128] D of every query of the workgroup
    for (int i = tid; i < kLdsRing / 4; i += 64 * NW) reinterpret_cast<f4*>(lds)[i] = zero4();      // rows of absent tokens are MFMA operands: finite
    for (int i = tid; i < kTokWG; i += 64 * NW) Dq[i] = 0.f;
    // dAo = dX o mask1 (DyGFormer.py:456), as rows: the operand of the out-projection's weight gradient and of the dOa products below
    {
        const uint32_t sk = a.dr.site_key(a.site_ao), e0 = (uint32_t)row * kD + 4 * g;
        const float* src = a.dX + row * kD + 4 * g;
        float* dst = a.dAo + row * kD + 4 * g;
#pragma unroll
        for (int i = 0; i < kNT; ++i) {
            if (valid && (i < 12 || g < 2)) {
                f4 v = ldg4(src + 16 * i);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= a.dr.mask32(sk, e0 + 16 * i + r);
                *reinterpret_cast<f4*>(dst + 16 * i) = v;
            }
        }
    }
    f4 dxn[kNT];
#pragma unroll
    for (int i = 0; i < kNT; ++i) dxn[i] = zero4();
    const uint32_t skp = a.dr.site_key(a.site_p);
    const bool vec = (T & 3) == 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the first ring stages have landed (and this lane's dAo row is written)
    __syncthreads();
    TACC(0);
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        const float* qrow = a.qkv + row * (3 * kD) + kHD * h + 4 * g;
        float* drow = a.dQKV + row * (3 * kD) + kHD * h + 4 * g;
        const int64_t pbase = (b * 2 + h) * (int64_t)T * T;
        // ---- K and V rows of the pair go to LDS by DMA (no registers) and this wave's Q rows (needed in phase B) into registers NOW: they land
        // while the Wo^T product runs.  (Every wave passed stream barriers since the previous head's last reads of the K/V region.)
        {
            constexpr int NCH = 16 * TPW * kKV * 4 / 1024;             // 1-KiB pieces of a pair's K (or V) block: rows are contiguous in LDS
            for (int q = tt; q < NCH; q += TPW) {
                const int o = 1024 * q + 16 * lane, r = o / (4 * kKV), cb = (o - r * 4 * kKV) >> 2;
                const bool on = pair_ok && r < T;
                const float* src = a.qkv + (b * T + r) * (3 * kD) + kHD * h + cb;
                dma_frag(on ? src + kD : g_zero16, kLdsK + tokbase * kKV + 256 * q);
                dma_frag(on ? src + 2 * kD : g_zero16, kLdsV + tokbase * kKV + 256 * q);
            }
        }
        f4 qrows[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) qrows[j] = (valid && (j < 6 || g == 0)) ? ldg4(qrow + 16 * j) : zero4();
        // ---- dOa^T = Wo[:, h]^T . dAo^T
        f4 doa[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) doa[j] = zero4();
        {
            f4 dA[kNT];
            load_rows(dA, a.dAo, row, g, valid);
            float k0 = 0.f, k1 = 0.f;
            if (F3_KSKIP) kpack(dA[kKC - 1], k0, k1);
            qkv_group(doa, dA, k0, k1, ws, ringl, active);
        }
        TACC(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of K and V have landed
        __syncthreads();
        TACC(2);
        // ---- phase A: this wave's tokens as queries
        f4 dq[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) dq[j] = zero4();
        if (active) {
            f4 ds[TPW], pt[TPW];
#pragma unroll
            for (int kt = 0; kt < TPW; ++kt) ds[kt] = zero4();
            s_like<TPW>(ds, Vb + tokbase * kKV, doa, c, g);                    // dPd^T[key][query]
            const float* Pq = a.P + pbase + (int64_t)tok * T;
            float D = 0.f;
#pragma unroll
            for (int kt = 0; kt < TPW; ++kt) {
                const int key0 = 16 * kt + 4 * g;
                if (vec) pt[kt] = (valid && key0 < T) ? ldg4(Pq + key0) : zero4();
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pt[kt][r] = (valid && key0 + r < T) ? Pq[key0 + r] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ds[kt][r] *= a.dr.mask32(skp, (uint32_t)(pbase + (int64_t)tok * T) + key0 + r);       // dP = dPd o mask0
                    D = fmaf(ds[kt][r], pt[kt][r], D);
                }
            }
            D += __shfl_xor(D, 16, 64);
            D += __shfl_xor(D, 32, 64);
#pragma unroll
            for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[kt][r] = pt[kt][r] * (ds[kt][r] - D) * a.qscale;
            if (g == 0) Dq[tokbase + tok] = D;
            pv_like<TPW>(dq, Kb + tokbase * kKV, ds, c, g);                    // dQ^T = K^T . dS^T (scaled)
        }
        f4 vt[7];                                    // this wave's V^T tiles for phase B, from its own rows while they are still in LDS
#pragma unroll
        for (int j = 0; j < 7; ++j) vt[j] = (active && (j < 6 || g == 0)) ? lds4(Vb + (tokbase + tok) * kKV + 4 * g + 16 * j) : zero4();
        TACC(3);
        __syncthreads();                             // every wave is done with K and V
        // ---- Q (unscaled, from the registers loaded above) and dOa rows over K and V; this wave's own V rows (its tokens as keys) were read
        // back from LDS before the barrier: the exchange touches no global memory
        if (active) {
#pragma unroll
            for (int j = 0; j < 7; ++j)
                if (j < 6 || g == 0) {
                    *reinterpret_cast<f4*>(Kb + (tokbase + tok) * kKV + 4 * g + 16 * j) = qrows[j];
                    *reinterpret_cast<f4*>(Vb + (tokbase + tok) * kKV + 4 * g + 16 * j) = doa[j];
                }
        }
        __syncthreads();
        TACC(4);
        // dxn0 += Wq[h]^T . dQ^T while the exchange settles (stream order: Wo^T, Wq^T, Wv^T, Wk^T)
        proj_t(dxn, dq, ws, ringl, active);
        // the row stores of dQ (and of dV, dK below) come AFTER the loads that follow their computation: a load behind a store waits for the
        // store's round trip (in-order vmcnt)
        if (valid) {
#pragma unroll
            for (int j = 0; j < 7; ++j)
                if (j < 6 || g == 0) *reinterpret_cast<f4*>(drow + 16 * j) = dq[j];
        }
        TACC(5);
        // ---- phase B: this wave's tokens as keys; tiles [query rows 16 qt + 4 g + r][own key column c]
        f4 dv[7], dk[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) { dv[j] = zero4(); dk[j] = zero4(); }
        {
            f4 ds2[TPW], pd2[TPW];
#pragma unroll
            for (int qt = 0; qt < TPW; ++qt) { ds2[qt] = zero4(); pd2[qt] = zero4(); }
            if (active) {
                s_like<TPW>(ds2, Vb + tokbase * kKV, vt, c, g);                // dPd[query][key] = dOa[query] . V[key]
#pragma unroll
                for (int qt = 0; qt < TPW; ++qt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = 16 * qt + 4 * g + r;
                        const bool on = valid && q < T;
                        const int64_t off = pbase + (int64_t)q * T + tok;
                        const float pv = on ? a.P[off] : 0.f;
                        pd2[qt][r] = on ? a.Pd[off] : 0.f;
                        const float dp = ds2[qt][r] * a.dr.mask32(skp, (uint32_t)off);
                        ds2[qt][r] = pv * (dp - Dq[tokbase + q]) * a.qscale;
                    }
                pv_like<TPW>(dv, Vb + tokbase * kKV, pd2, c, g);               // dV^T = dOa^T . Pd
            }
            TACC(6);
            proj_t(dxn, dv, ws, ringl, active);
            TACC(7);
            if (active) pv_like<TPW>(dk, Kb + tokbase * kKV, ds2, c, g);       // dK^T = Q^T . dS (scaled)
        }
        TACC(8);
        proj_t(dxn, dk, ws, ringl, active);
        if (valid) {
#pragma unroll
            for (int j = 0; j < 7; ++j)
                if (j < 6 || g == 0) { *reinterpret_cast<f4*>(drow + 2 * kD + 16 * j) = dv[j]; *reinterpret_cast<f4*>(drow + kD + 16 * j) = dk[j]; }
        }
        TACC(9);
    }
    // ---- LayerNorm-0 backward
    ws.fit(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                 // every wave is done with the K/V region: it now holds the dgamma / dbeta partial sums
    ln_backward(dxn, a.X, a.m0, a.r0, lds + kLdsRing + ws.pos * kFrag + 4 * g, a.dX, row, valid, lds, wave, c, g);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 2 * kDP; i += 64 * NW) {
        const int which = i / kDP, n = i % kDP;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += lds[(w * 2 + which) * kDP + n];
        if (n < kD) atomicAdd((which ? a.dbeta : a.dgamma) + n, t);
    }
    TACC(10);
    TSTORE();
}

// ================================================================================================
// packing: the stream is described on the host as a list of fragment descriptors in consumption order (with the pad
// fragments the ring rule asks for), uploaded, and materialised by one kernel.
// ================================================================================================
struct FragDesc {
    const float* src;     // nullptr = pad fragment (zeros)
    int ld;               // matrix fragment: row stride; -1: vector fragment, element e = src[c0 + e] for e < rmax
    int r0, rmax;         // element (c,g,t): row = r0 + c, valid iff 0 <= row < rmax
    int c0, cmax;         //                  col = c0 + 4g + t, valid iff col < cmax
    int kmode;            // 1: last chunk of a K = 200 product, 8 real k in two MFMAs: t < 2: col = c0 + {0,4,1,5}[g] + 2t, t >= 2: zero (mma_group2)
                          // 2: last chunk of a head-dim (100) contraction, 4 real k in one MFMA: t = 0: col = c0 + g, t >= 1: zero (mma_group1)
                          // +4: transposed source, element (row, col) = src[col * ld + row] (the backward stream: W^T fragments of the same tensors)
                          // +8: the COMBINED last head-dim tile of a head's q | k | v (src = in_proj base, r0 = 100 h + 96): rows 4 G .. 4 G + 3 of the tile
                          //     are rows r0 .. r0 + 3 of row block G (q, k, v at G = 0, 1, 2: row = r0 + 200 G + (c & 3)), G = 3 is padding.  The three
                          //     parts' tiles 6 hold 4 real rows each (head dim 100 = 6 tiles + 4): one MFMA tile carries all twelve.
                          //     Vector fragment (+8): elements 100 .. 107 = src[c0 + 200 + 96 ..], src[c0 + 400 + 96 ..] (the k and v bias of those rows)
};
__device__ __forceinline__ float frag_element(const FragDesc& d, int e) {       // e = 4 * lane + t of the fragment
    if (d.src == nullptr) return 0.f;
    const int t = e & 3, lane = (e >> 2) & 63;
    const int c = lane & 15, g = lane >> 4;
    if (d.ld < 0) {
        if (e < d.rmax) return d.src[d.c0 + e];
        if ((d.kmode & 8) && e >= 100 && e < 108) return d.src[d.c0 + kD * ((e - 100) / 4 + 1) + 96 + ((e - 100) & 3)];
        return 0.f;
    }
    int row = d.r0 + c;
    bool rok = row >= 0 && row < d.rmax;
    if (d.kmode & 8) { row = d.r0 + kD * (c >> 2) + (c & 3); rok = (c >> 2) < 3; }
    int col = d.c0 + 4 * g + t;
    if ((d.kmode & 3) == 1) col = t < 2 ? d.c0 + (g & 1) * 4 + (g >> 1) + 2 * t : d.cmax;
    if ((d.kmode & 3) == 2) col = t == 0 ? d.c0 + g : d.cmax;
    if (!rok || col >= d.cmax) return 0.f;
    return (d.kmode & 4) ? d.src[(size_t)col * d.ld + row] : d.src[(size_t)row * d.ld + col];
}

__global__ void k_pack_stream(const FragDesc* __restrict__ desc, int64_t nfrag, float* __restrict__ dst) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nfrag * kFrag) return;
    dst[idx] = frag_element(desc[idx >> 8], (int)(idx & 255));
}

// every fragment stream of the packed buffer in ONE launch (the in-place refresh after an optimizer step): the descriptor table is one
// array, `r` maps its ranges to their destinations
struct PackRanges { int n; int64_t start[4 + 2 * DYGNN_MAX_LAYERS]; float* dst[3 + 2 * DYGNN_MAX_LAYERS]; };
__global__ void k_pack_ranges(const FragDesc* __restrict__ desc, const PackRanges r) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t f = idx >> 8;
    if (f >= r.start[r.n]) return;
    int q = 0;
    while (q + 1 < r.n && f >= r.start[q + 1]) ++q;
    r.dst[q][(f - r.start[q]) * kFrag + (idx & 255)] = frag_element(desc[f], (int)(idx & 255));
}
// the four projection biases in model-dim order [208]
__global__ void k_pack_bias4(const float* __restrict__ b0, const float* __restrict__ b1, const float* __restrict__ b2, const float* __restrict__ b3, float* __restrict__ dst) {
    const int i = threadIdx.x;
    if (i >= kDP) return;
    const int ch = i / kC, j = i % kC;
    dst[i] = i < kD ? (ch == 0 ? b0 : ch == 1 ? b1 : ch == 2 ? b2 : b3)[j] : 0.f;
}

__global__ void k_pack_vec3(const float* __restrict__ src, int n_valid, int src_off, float* __restrict__ dst, int dst_off, int n_total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    dst[dst_off + i] = i < n_valid ? src[src_off + i] : 0.f;
}

// host mirror of WStream's position rule
struct StreamBuilder {
    std::vector<FragDesc> frags;
    int pos = 0;
    void pad(int n) { for (int i = 0; i < n; ++i) frags.push_back(FragDesc{nullptr, 0, 0, 0, 0, 0, 0}); pos = (pos + n) % kRing; }
    void fit(int n) { if (pos + n > kRing) pad(kRing - pos); }
    void align26() { if (pos != 0 && pos != 26) pad(pos < 26 ? 26 - pos : kRing - pos); }
    void put(const float* src, int ld, int r0, int rmax, int c0, int cmax, int kmode = 0) { frags.push_back(FragDesc{src, ld, r0, rmax, c0, cmax, kmode}); pos = (pos + 1) % kRing; }
    void put_vec(const float* src, int off, int n) { put(src, -1, 0, n, off, 0); }     // floats [0, n) of the fragment = src[off ..]
};

static void build_stream(const Dims& d, const dygnn_dygformer_weights* w, StreamBuilder& sb, int (&nchunk)[4]) {
    const int K[4] = {d.P * d.Fn, d.P * d.Fe, d.P * d.Ft, d.P * d.C};
    for (int ch = 0; ch < 4; ++ch) nchunk[ch] = (K[ch] + 15) / 16;
    for (int l = 0; l < d.NL; ++l) {
        const dygnn_encoder_layer_weights& L = w->layers[l];
        sb.fit(2);
        sb.put_vec(L.norm0_weight, 0, kD);
        sb.put_vec(L.norm0_bias, 0, kD);
        for (int h = 0; h < 2; ++h) {
            for (int part = 0; part < 3; ++part) {           // q, k, v row blocks of in_proj
                // q: 7 tiles, the seventh = the COMBINED tile (rows 96 .. 99 of q, k and v: FragDesc kmode 8); k, v: tiles 0 .. 5
                const int nt = part == 0 ? 7 : 6;
                sb.fit(1);
                if (part == 0) sb.frags.push_back(FragDesc{L.in_proj_bias, -1, 0, kHD, kHD * h, 0, 8}), sb.pos = (sb.pos + 1) % kRing;
                else sb.put_vec(L.in_proj_bias, part * kD + kHD * h, kHD);      // head rows of the bias (elements 0 .. 95 are read)
                sb.fit(nt);
                for (int kc = 0; kc < kKC; ++kc) {
                    const int ks = (F3_KSKIP && kc == kKC - 1) ? 1 : 0;
                    for (int j = 0; j < 6; ++j)
                        sb.put(L.in_proj_weight ? L.in_proj_weight + (size_t)part * kD * kD : nullptr, kD, kHD * h + 16 * j, kHD * (h + 1), 16 * kc, kD, ks);
                    if (part == 0) sb.put(L.in_proj_weight, kD, kHD * h + 96, 3 * kD, 16 * kc, kD, ks | 8);
                    if (kc + 1 < kKC) sb.fit(nt);
                }
            }
            sb.fit(13);
            for (int j = 0; j < 7; ++j) {                    // out-projection: [d-chunk j][n-tile i], columns of head h
                for (int i = 0; i < kNT; ++i) sb.put(L.out_proj_weight, kD, 16 * i, kD, kHD * h + 16 * j, kHD * (h + 1), (F3_KSKIP && j == 6) ? 2 : 0);
                if (j + 1 < 7) sb.fit(13);
            }
        }
        sb.fit(1);
        sb.put_vec(L.out_proj_bias, 0, kD);
        sb.fit(2);
        sb.put_vec(L.norm1_weight, 0, kD);
        sb.put_vec(L.norm1_bias, 0, kD);
        sb.align26();
        auto put_w1 = [&](int p) {
            for (int kc = 0; kc < kKC; ++kc)
                for (int u = 0; u < 2; ++u) sb.put(L.ffn0_weight, kD, 16 * (2 * p + u), kHid, 16 * kc, kD, F3_KSKIP && kc == kKC - 1);
        };
        auto put_w2 = [&](int p) {
            for (int u = 0; u < 2; ++u)
                for (int i = 0; i < kNT; ++i) sb.put(L.ffn1_weight, kHid, 16 * i, kD, 16 * (2 * p + u), kHid);
        };
        for (int p = 0; p < 25; ++p) { put_w1(p); put_w2(p); }
        sb.fit(1);
        sb.put_vec(L.ffn1_bias, 0, kD);
    }
}

// backward stream of layer l's FFN block (k_ffn_bwd): per hidden step p the W2^T block [k-chunk over channels][hidden tile u] and the
// W1^T block [hidden chunk u][channel tile i] — the transposes of the forward's two blocks, cut from the same tensors — then LN1's gamma
static void build_bwd_ffn(const dygnn_encoder_layer_weights& L, StreamBuilder& sb) {
    for (int p = 0; p < 25; ++p) {
        for (int kc = 0; kc < kKC; ++kc)
            for (int u = 0; u < 2; ++u) sb.put(L.ffn1_weight, kHid, 16 * (2 * p + u), kHid, 16 * kc, kD, ((F3_KSKIP && kc == kKC - 1) ? 1 : 0) | 4);
        for (int u = 0; u < 2; ++u)
            for (int i = 0; i < kNT; ++i) sb.put(L.ffn0_weight, kD, 16 * i, kD, 16 * (2 * p + u), kHid, 4);
    }
    sb.fit(1);
    sb.put_vec(L.norm1_weight, 0, kD);
}
constexpr int64_t kBwdFfnFrags = 25 * 52 + 1;
// backward stream of layer l's attention block (k_attn_bwd): per head Wo[:, h]^T in the shape of a Q/K/V product ([channel chunk][7 head-dim
// tiles]), then Wq[h]^T, Wv[h]^T, Wk[h]^T in the shape of the out-projection ([head-dim chunk][13 channel tiles]); then LN0's gamma
static void build_bwd_attn(const dygnn_encoder_layer_weights& L, StreamBuilder& sb) {
    for (int h = 0; h < 2; ++h) {
        sb.fit(7);
        for (int kc = 0; kc < kKC; ++kc) {
            for (int j = 0; j < 7; ++j)
                sb.put(L.out_proj_weight, kD, kHD * h + 16 * j, kHD * (h + 1), 16 * kc, kD, ((F3_KSKIP && kc == kKC - 1) ? 1 : 0) | 4);
            if (kc + 1 < kKC) sb.fit(7);
        }
        const int order[3] = {0, 2, 1};              // q, v, k
        for (int o = 0; o < 3; ++o) {
            const float* Wp = L.in_proj_weight ? L.in_proj_weight + (size_t)order[o] * kD * kD : nullptr;
            sb.fit(13);
            for (int j = 0; j < 7; ++j) {
                for (int i = 0; i < kNT; ++i) sb.put(Wp, kD, 16 * i, kD, kHD * h + 16 * j, kHD * (h + 1), ((F3_KSKIP && j == 6) ? 2 : 0) | 4);
                if (j + 1 < 7) sb.fit(13);
            }
        }
    }
    sb.fit(1);
    sb.put_vec(L.norm0_weight, 0, kD);
}
static int64_t bwd_attn_frags() {
    static float dummy;
    dygnn_encoder_layer_weights lw{};
    lw.in_proj_weight = lw.out_proj_weight = lw.norm0_weight = &dummy;
    StreamBuilder sb;
    build_bwd_attn(lw, sb);
    return (int64_t)sb.frags.size();
}

// projection fragments in step order (channels node, time, edge, cooc; 4 tiles per k-chunk slot), staged through two LDS halves
static int proj_slots(int nchunk) { return (nchunk + 3) / 4 * 4; }
static void build_proj(const Dims& d, const dygnn_dygformer_weights* w, StreamBuilder& sb) {
    const float* pw[4] = {w->proj_node_w, w->proj_edge_w, w->proj_time_w, w->proj_cooc_w};
    const int K[4] = {d.P * d.Fn, d.P * d.Fe, d.P * d.Ft, d.P * d.C};
    const int order[4] = {0, 2, 1, 3};
    for (int o = 0; o < 4; ++o) {
        const int ch = order[o], t0 = (kC * ch) / 16;
        const int n = (K[ch] + 15) / 16;
        for (int kc = 0; kc < n; ++kc)
            for (int u = 0; u < 4; ++u) sb.put(pw[ch], K[ch], 16 * (t0 + u) - kC * ch, kC, 16 * kc, K[ch]);
        sb.pad(4 * (proj_slots(n) - n));           // every channel occupies whole groups of four slots (the kernel's loop bodies are groups)
    }
}

// fragments that do not travel through the ring (read by one wave each): the output layer [tile][k-chunk]
static void build_aux(const Dims& d, const dygnn_dygformer_weights* w, StreamBuilder& sb) {
    const int ntile = (d.Fn + 15) / 16;
    for (int jt = 0; jt < ntile; ++jt)
        for (int kc = 0; kc < kKC; ++kc) sb.put(w->output_w, kD, 16 * jt, d.Fn, 16 * kc, kD);
}

struct PackLayout3 {       // float offsets relative to PackedLayout.fused3
    size_t bias_x;
    size_t stream; int64_t nfrag; int nstages;     // ring stream: nfrag fragments, padded to whole stages (+ one of slack)
    size_t aux; int64_t naux;                      // output-layer fragments
    size_t proj; int64_t nproj;                    // projection fragments
    int scr_floats, slab_chunks;                   // LDS split of the K/V region during the prologue
    int np, slab_in_ring;                          // pairs per workgroup (0: shape unsupported); slab placed in the weight ring
    int tab_off, tab_slots, tab_bits;              // co-occurrence table (long windows): LDS word offset, slots per pair
    size_t bwd[DYGNN_MAX_LAYERS]; int bwd_nstages;  // per layer: the backward stream of its FFN block (training only)
    size_t bwa[DYGNN_MAX_LAYERS]; int bwa_nstages; int64_t bwa_frags;      // ... and of its attention block
    size_t desc;           // FragDesc table (device copy), 8-byte aligned
    size_t total;
};

static int64_t stream_frags(const Dims& d) {
    // fragment count of build_stream without touching weights: run the builder with null sources
    dygnn_dygformer_weights w{};
    static float dummy;
    w.proj_node_w = w.proj_edge_w = w.proj_time_w = w.proj_cooc_w = &dummy;
    dygnn_encoder_layer_weights lw{};
    lw.in_proj_weight = lw.out_proj_weight = lw.ffn0_weight = lw.ffn1_weight = &dummy;
    lw.in_proj_bias = lw.out_proj_bias = lw.ffn1_bias = lw.norm0_weight = lw.norm0_bias = lw.norm1_weight = lw.norm1_bias = &dummy;
    for (int l = 0; l < d.NL; ++l) w.layers[l] = lw;
    StreamBuilder sb;
    int nchunk[4];
    build_stream(d, &w, sb, nchunk);
    return (int64_t)sb.frags.size();
}

static PackLayout3 make_layout3(const Dims& d) {
    PackLayout3 f;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 63) & ~size_t(63); return r; };
    f.bias_x = take(kDP);
    f.nfrag = stream_frags(d);
    f.nstages = (int)((f.nfrag + kStage - 1) / kStage);
    f.stream = take((size_t)(f.nstages + 1) * kStage * kFrag);
    f.naux = (int64_t)((d.Fn + 15) / 16) * kKC;
    f.aux = take((size_t)f.naux * kFrag);
    const int K[4] = {d.P * d.Fn, d.P * d.Fe, d.P * d.Ft, d.P * d.C};
    f.nproj = 0;
    for (int ch = 0; ch < 4; ++ch) f.nproj += 4 * (int64_t)proj_slots((K[ch] + 15) / 16);
    f.proj = take((size_t)f.nproj * kFrag);
    f.bwd_nstages = (int)((kBwdFfnFrags + kStage - 1) / kStage);
    for (int l = 0; l < d.NL; ++l) f.bwd[l] = take((size_t)(f.bwd_nstages + 1) * kStage * kFrag);
    f.bwa_frags = bwd_attn_frags();
    f.bwa_nstages = (int)((f.bwa_frags + kStage - 1) / kStage);
    for (int l = 0; l < d.NL; ++l) f.bwa[l] = take((size_t)(f.bwa_nstages + 1) * kStage * kFrag);
    f.desc = take(((size_t)(f.nfrag + f.naux + f.nproj + d.NL * (kBwdFfnFrags + f.bwa_frags)) * sizeof(FragDesc) + 3) / 4);
    // prologue LDS split: pairs per workgroup, window arrays (5 x 2 sides x Smax ints per pair), projection slab
    const int per_pair = 5 * 2 * ((d.Smax + 3) & ~3);
    f.np = 0; f.slab_in_ring = 0; f.scr_floats = 0; f.slab_chunks = 0; f.tab_off = 0; f.tab_slots = 0; f.tab_bits = 0;
    if (d.Tmax <= 64 && 2 * per_pair + 8 * 4 * kFrag <= kScratchFloats) f.np = 2;
    else if (d.Tmax <= 128 && per_pair + 8 * 4 * kFrag <= kScratchFloats) f.np = 1;
    else if (d.Tmax <= 128 && per_pair <= kScratchFloats) { f.np = 1; f.slab_in_ring = 1; }
    if (f.np) {
        f.scr_floats = f.np * per_pair;
        // two halves of slab_chunks slots each, whole groups of four slots
        f.slab_chunks = (f.slab_in_ring ? kRing / 4 : (kScratchFloats - f.scr_floats) / (4 * kFrag)) / 8 * 4;
        // what the window arrays and the two halves leave of the K/V region: a co-occurrence table of >= 2 x positions slots per pair, for
        // windows long enough that two barriers cost less than the all-pairs scan
        f.tab_off = f.scr_floats + (f.slab_in_ring ? 0 : 2 * f.slab_chunks * 4 * kFrag);
        const int positions = 2 * ((d.Smax + 3) & ~3), words = (kScratchFloats - f.tab_off) / f.np;
        int bits = 0;
        while ((2 << (bits + 1)) <= words) ++bits;          // slots = 2^bits, two words per slot
        if (positions >= 512 && (1 << bits) >= 2 * positions) { f.tab_slots = 1 << bits; f.tab_bits = bits; }
    }
    f.total = o;
    return f;
}

bool supported(const Dims& d) {
    if (!(d.C == kC && d.H == 2 && d.Fn % 4 == 0 && d.Fe % 4 == 0 && d.Ft % 4 == 0 && d.Fn >= 16 && d.Fe >= 16 && d.Ft >= 16 &&
          d.Fn <= 512 && d.NL <= DYGNN_MAX_LAYERS && d.Tmax <= 128 && (kLdsMisc + kMiscFloats + 2 * d.Ft) * 4 <= kLdsBytes)) return false;
    // k/50 multiply-shift range; the window arrays must fit the K/V region (make_layout3 decides how)
    if (!(d.P * kC < 12000)) return false;
    const int per_pair = 5 * 2 * ((d.Smax + 3) & ~3);
    return per_pair <= kScratchFloats;
}

size_t packed_floats(const Dims& d) { return supported(d) ? make_layout3(d).total : 0; }

static int pack_vec(const float* src, int n_valid, int src_off, float* dst, int dst_off, int n_total, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_vec3, dim3((n_total + 255) / 256), dim3(256), 0, s, src, n_valid, src_off, dst, dst_off, n_total);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

// reuse_desc: the weights changed IN PLACE since the last full pack into this buffer (same addresses): the fragment descriptor table that
// pack left in the buffer is still right, so only the gather kernels run — no host work, no synchronisation (one optimizer step = one repack)
int pack(const Dims& d, const PackedLayout& pl, const dygnn_dygformer_weights* w, float* packed, hipStream_t s, bool reuse_desc) {
    const PackLayout3 f = make_layout3(d);
    float* base = packed + pl.fused3;
    if (!reuse_desc) DYGNN_HIP(hipMemsetAsync(base, 0, f.total * sizeof(float), s));
    hipLaunchKernelGGL(k_pack_bias4, dim3(1), dim3(256), 0, s, w->proj_node_b, w->proj_edge_b, w->proj_time_b, w->proj_cooc_b, base + f.bias_x);
    DYGNN_LAUNCH_CHECK();
    if (reuse_desc) {
        // table order (as laid down by the full pack below): stream | aux | proj | NL x FFN backward | NL x attention backward
        PackRanges r{};
        int64_t o = 0;
        auto range = [&](int64_t nfr, float* dst) { r.start[r.n] = o; r.dst[r.n] = dst; ++r.n; o += nfr; };
        range(f.nfrag, base + f.stream); range(f.naux, base + f.aux); range(f.nproj, base + f.proj);
        for (int l = 0; l < d.NL; ++l) range(kBwdFfnFrags, base + f.bwd[l]);
        for (int l = 0; l < d.NL; ++l) range(f.bwa_frags, base + f.bwa[l]);
        r.start[r.n] = o;
        hipLaunchKernelGGL(k_pack_ranges, dim3((unsigned)ceil_div(o * kFrag, 256)), dim3(256), 0, s, reinterpret_cast<const FragDesc*>(base + f.desc), r);
        DYGNN_LAUNCH_CHECK();
        return DYGNN_OK;
    }
    StreamBuilder sb;
    int nchunk[4];
    build_stream(d, w, sb, nchunk);
    if ((int64_t)sb.frags.size() != f.nfrag) { set_error("pack: stream builder mismatch"); return DYGNN_E_INVALID; }
    StreamBuilder aux;
    build_aux(d, w, aux);
    if ((int64_t)aux.frags.size() != f.naux) { set_error("pack: aux builder mismatch"); return DYGNN_E_INVALID; }
    FragDesc* ddesc = reinterpret_cast<FragDesc*>(base + f.desc);
    DYGNN_HIP(hipMemcpyAsync(ddesc, sb.frags.data(), sb.frags.size() * sizeof(FragDesc), hipMemcpyHostToDevice, s));
    DYGNN_HIP(hipMemcpyAsync(ddesc + f.nfrag, aux.frags.data(), aux.frags.size() * sizeof(FragDesc), hipMemcpyHostToDevice, s));
    StreamBuilder pj;
    build_proj(d, w, pj);
    if ((int64_t)pj.frags.size() != f.nproj) { set_error("pack: projection builder mismatch"); return DYGNN_E_INVALID; }
    DYGNN_HIP(hipMemcpyAsync(ddesc + f.nfrag + f.naux, pj.frags.data(), pj.frags.size() * sizeof(FragDesc), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pack_stream, dim3((unsigned)ceil_div(f.nproj * kFrag, 256)), dim3(256), 0, s, ddesc + f.nfrag + f.naux, f.nproj,
                       base + f.proj);
    DYGNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pack_stream, dim3((unsigned)ceil_div(f.nfrag * kFrag, 256)), dim3(256), 0, s, ddesc, f.nfrag, base + f.stream);
    DYGNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pack_stream, dim3((unsigned)ceil_div(f.naux * kFrag, 256)), dim3(256), 0, s, ddesc + f.nfrag, f.naux, base + f.aux);
    DYGNN_LAUNCH_CHECK();
    std::vector<FragDesc> bw;
    for (int l = 0; l < d.NL; ++l) {
        StreamBuilder sbb;
        build_bwd_ffn(w->layers[l], sbb);
        if ((int64_t)sbb.frags.size() != kBwdFfnFrags) { set_error("pack: backward stream builder mismatch"); return DYGNN_E_INVALID; }
        bw.insert(bw.end(), sbb.frags.begin(), sbb.frags.end());
    }
    for (int l = 0; l < d.NL; ++l) {
        StreamBuilder sba;
        build_bwd_attn(w->layers[l], sba);
        if ((int64_t)sba.frags.size() != f.bwa_frags) { set_error("pack: attention backward stream builder mismatch"); return DYGNN_E_INVALID; }
        bw.insert(bw.end(), sba.frags.begin(), sba.frags.end());
    }
    FragDesc* bdesc = ddesc + f.nfrag + f.naux + f.nproj;
    DYGNN_HIP(hipMemcpyAsync(bdesc, bw.data(), bw.size() * sizeof(FragDesc), hipMemcpyHostToDevice, s));
    for (int l = 0; l < d.NL; ++l) {
        hipLaunchKernelGGL(k_pack_stream, dim3((unsigned)ceil_div(kBwdFfnFrags * kFrag, 256)), dim3(256), 0, s, bdesc + l * kBwdFfnFrags, kBwdFfnFrags, base + f.bwd[l]);
        DYGNN_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_pack_stream, dim3((unsigned)ceil_div(f.bwa_frags * kFrag, 256)), dim3(256), 0, s, bdesc + d.NL * kBwdFfnFrags + l * f.bwa_frags, f.bwa_frags,
                           base + f.bwa[l]);
        DYGNN_LAUNCH_CHECK();
    }
    DYGNN_HIP(hipStreamSynchronize(s));     // the descriptor tables are copied from this call's host vectors
    return DYGNN_OK;
}

}  // namespace v3

int window_lengths_device(const Dims& d, const dygnn_csr* csr, const int64_t* src, const int64_t* dst, const double* times,
                          int64_t B, int64_t G, char* ws, const WorkspaceLayout& wl, hipStream_t s);   // dygformer_generic.hip

bool fused3_supported(const Dims& d) { return v3::supported(d); }
size_t fused3_packed_floats(const Dims& d) { return v3::packed_floats(d); }
int pack_fused3(const Dims& d, const PackedLayout& pl, const dygnn_dygformer_weights* w, float* packed, hipStream_t s, bool reuse_desc) {
    return v3::pack(d, pl, w, packed, s, reuse_desc);
}

// calls of at most this many pairs run one pair per four-wave workgroup (see k_dygformer_fused3): one round on the 256 CUs
constexpr int64_t kSmallBatchPairs = 256;
static bool small_off() { static const bool off = [] { const char* e = getenv("DYGNN_SMALL_BATCH_KERNELS"); return e && e[0] == '0'; }(); return off; }

static int fused3_args(const Dims& d, const PackedLayout& pl, const dygnn_dygformer_weights* w, const float* packed, const dygnn_csr* csr,
                       const float* node_feat, const float* edge_feat, const int64_t* src, const int64_t* dst, const double* times, int64_t B, int64_t G,
                       float* out_src, float* out_dst, char* ws, const WorkspaceLayout& wl, const dygnn_dygformer_taps* taps, v3::Args& a, v3::PackLayout3& f) {
    using namespace v3;
    f = make_layout3(d);
    const float* base = packed + pl.fused3;
    a.indptr = csr->indptr; a.nbr = csr->nbr; a.eid = csr->eid; a.ts = csr->ts;
    a.src = src; a.dst = dst; a.times = times;
    a.hist_len = reinterpret_cast<const int32_t*>(ws + wl.hist_len);
    a.end_pos = reinterpret_cast<const int64_t*>(ws + wl.end_pos);
    a.cd = reinterpret_cast<const CallDims*>(ws + wl.dims);
    a.node_feat = node_feat; a.edge_feat = edge_feat; a.time_w = w->time_w; a.time_b = w->time_b; a.lut = packed + pl.lut;
    a.stream = base + f.stream; a.nstages = f.nstages;
    a.bias_x = base + f.bias_x;
    for (int l = 0; l < d.NL; ++l) {
        a.layer[l].b1 = w->layers[l].ffn0_bias;
        a.tap_layer[l] = taps ? taps->layer_out[l] : nullptr;
    }
    a.outfrag = base + f.aux;
    a.projw = base + f.proj; a.proj_frags = (int)f.nproj; a.slab_chunks = f.slab_chunks; a.scr_floats = f.scr_floats;
    a.slab_in_ring = f.slab_in_ring;
    a.tab_off = f.tab_off; a.tab_slots = f.tab_slots; a.tab_bits = f.tab_bits;
    a.outT = packed + pl.outputT; a.outb = w->output_b;
    a.out_src = out_src; a.out_dst = out_dst;
    a.tap_enc = taps ? taps->encoder_input : nullptr;
    a.stamps = taps ? reinterpret_cast<unsigned long long*>(taps->phase_cycles) : nullptr;
    a.B = B; a.G = G; a.num_nodes = csr->num_nodes; a.Fn = d.Fn; a.Fe = d.Fe; a.Ft = d.Ft; a.P = d.P; a.L = d.L; a.NL = d.NL; a.Tmax = d.Tmax;
    const int K[4] = {d.P * d.Fn, d.P * d.Fe, d.P * d.Ft, d.P * d.C};
    for (int ch = 0; ch < 4; ++ch) a.nchunk[ch] = (K[ch] + 15) / 16;
    a.qscale = (float)sqrt(1.0 / (double)d.hd);
    return DYGNN_OK;
}

int forward_fused3(const Dims& d, const PackedLayout& pl, const dygnn_dygformer_weights* w, const float* packed,
                   const dygnn_csr* csr, const float* node_feat, const float* edge_feat, const int64_t* src,
                   const int64_t* dst, const double* times, int64_t B, int64_t G, int64_t pair_stride, float* out_src, float* out_dst, char* ws,
                   const WorkspaceLayout& wl, const dygnn_dygformer_taps* taps, hipStream_t s) {
    using namespace v3;
    if (!supported(d)) { set_error("fused kernel: unsupported shape"); return DYGNN_E_UNSUPPORTED; }
    if (int rc = window_lengths_device(d, csr, src, dst, times, B, G, ws, wl, s)) return rc;
    Args a{};
    PackLayout3 f;
    if (int rc = fused3_args(d, pl, w, packed, csr, node_feat, edge_feat, src, dst, times, B, G, out_src, out_dst, ws, wl, taps, a, f)) return rc;
    if (taps && taps->seq_lens) DYGNN_HIP(hipMemcpyAsync(taps->seq_lens, ws + wl.dims + 2 * sizeof(int32_t), 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    // per device and cheap: set on every call (a process may drive several GPUs, or call from several threads)
    DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dygformer_fused3<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dygformer_fused3<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    if (taps && taps->ev_kernel_start) DYGNN_HIP(hipEventRecord(static_cast<hipEvent_t>(taps->ev_kernel_start), s));
    a.pair_stride = (f.np == 2 && pair_stride > 0) ? pair_stride : 0;      // one pair per workgroup (128 tokens): nothing to share inside a workgroup
    if (f.np == 2 && a.pair_stride == 0 && B <= kSmallBatchPairs && !small_off()) {      // a small call: one pair per four-wave workgroup
        DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dygformer_fused3<4, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        hipLaunchKernelGGL((k_dygformer_fused3<4, false, 4>), dim3((unsigned)B), dim3(256), kLdsBytes, s, a);
    } else if (f.np == 2) hipLaunchKernelGGL((k_dygformer_fused3<4, false>), dim3((unsigned)(a.pair_stride ? a.pair_stride : (B + 1) / 2)), dim3(512), kLdsBytes, s, a);
    else hipLaunchKernelGGL((k_dygformer_fused3<8, false>), dim3((unsigned)B), dim3(512), kLdsBytes, s, a);
    DYGNN_LAUNCH_CHECK();
    if (taps && taps->ev_kernel_stop) DYGNN_HIP(hipEventRecord(static_cast<hipEvent_t>(taps->ev_kernel_stop), s));
    return DYGNN_OK;
}

// FFN block of layer l, backward (k_ffn_bwd); the caller's buffers are the dense rows of dygformer_train.hip's Plan
int ffn_backward_fused3(const Dims& d, const PackedLayout& pl, const float* packed, int l, int64_t M, float* dX, const float* hpre, const float* x1,
                        const float* m1, const float* r1, float* dF2, float* dH, float* dgamma, float* dbeta, const train::Drop& dr, hipStream_t s) {
    using namespace v3;
    if (!supported(d)) { set_error("fused FFN backward: unsupported shape"); return DYGNN_E_UNSUPPORTED; }
    const PackLayout3 f = make_layout3(d);
    FfnBwdArgs a{};
    a.stream = packed + pl.fused3 + f.bwd[l]; a.nstages = f.bwd_nstages;
    a.M = M; a.dX = dX; a.hpre = hpre; a.x1 = x1; a.m1 = m1; a.r1 = r1; a.dF2 = dF2; a.dH = dH; a.dgamma = dgamma; a.dbeta = dbeta;
    a.dr = dr; a.site_act = (uint32_t)(4 * l + 2); a.site_out = (uint32_t)(4 * l + 3);
#ifdef DYGNN_STAMPS
    if (const char* sp = getenv("DYGNN_STAMPS_FFN")) a.stamps = reinterpret_cast<unsigned long long*>(strtoull(sp, nullptr, 0));
#endif
    if (M <= (int64_t)kSmallBatchPairs * 64 && !small_off()) {
        DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ffn_bwd<4>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        hipLaunchKernelGGL(k_ffn_bwd<4>, dim3((unsigned)ceil_div(M, (int64_t)64)), dim3(256), kLdsBytes, s, a);
    } else {
        DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ffn_bwd<8>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        hipLaunchKernelGGL(k_ffn_bwd<8>, dim3((unsigned)ceil_div(M, (int64_t)kTokWG)), dim3(512), kLdsBytes, s, a);
    }
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

// Attention block of layer l, backward (k_attn_bwd); B pairs of T tokens each (one group: the training path's dense layout)
int attn_backward_fused3(const Dims& d, const PackedLayout& pl, const float* packed, int l, int64_t B, int T, float* dX, const float* X, const float* m0,
                         const float* r0, const float* qkv, const float* P, const float* Pd, float* dAo, float* dQKV, float* dgamma, float* dbeta,
                         const train::Drop& dr, hipStream_t s) {
    using namespace v3;
    if (!supported(d) || T > 128) { set_error("fused attention backward: unsupported shape"); return DYGNN_E_UNSUPPORTED; }
    const PackLayout3 f = make_layout3(d);
    AttnBwdArgs a{};
    a.stream = packed + pl.fused3 + f.bwa[l]; a.nstages = f.bwa_nstages;
    a.B = B; a.T = T; a.dX = dX; a.X = X; a.m0 = m0; a.r0 = r0; a.qkv = qkv; a.P = P; a.Pd = Pd; a.dAo = dAo; a.dQKV = dQKV; a.dgamma = dgamma; a.dbeta = dbeta;
    a.dr = dr; a.site_p = (uint32_t)(4 * l + 0); a.site_ao = (uint32_t)(4 * l + 1);
    a.qscale = (float)sqrt(1.0 / (double)d.hd);
#ifdef DYGNN_STAMPS
    if (const char* sp = getenv("DYGNN_STAMPS_ATTN")) a.stamps = reinterpret_cast<unsigned long long*>(strtoull(sp, nullptr, 0));
#endif
    DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd<4>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd<8>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    if (T <= 64 && B <= kSmallBatchPairs && !small_off()) {
        DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        hipLaunchKernelGGL((k_attn_bwd<4, 4>), dim3((unsigned)B), dim3(256), kLdsBytes, s, a);
    } else if (T <= 64) hipLaunchKernelGGL(k_attn_bwd<4>, dim3((unsigned)((B + 1) / 2)), dim3(512), kLdsBytes, s, a);
    else hipLaunchKernelGGL(k_attn_bwd<8>, dim3((unsigned)B), dim3(512), kLdsBytes, s, a);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

// Training forward through the fused kernel (dygformer_train.hip calls this when the shape is supported): one group of B pairs whose
// window lengths (hist_len / end_pos / dims at the head of `ws`, layout `wl`) the caller has already computed; `lut` = the co-occurrence
// table of the CURRENT weights; `packed` holds the fragment stream of the current weights (dygnn_dygformer_pack / _repack).
int forward_fused3_train(const Dims& d, const PackedLayout& pl, const dygnn_dygformer_weights* w, const float* packed, const dygnn_csr* csr,
                         const float* node_feat, const float* edge_feat, const int64_t* src, const int64_t* dst, const double* times, int64_t B,
                         const float* lut, float* out_src, float* out_dst, char* ws, const WorkspaceLayout& wl, const train::TrainOut& tr, hipStream_t s) {
    using namespace v3;
    if (!supported(d)) { set_error("fused training forward: unsupported shape"); return DYGNN_E_UNSUPPORTED; }
    Args a{};
    PackLayout3 f;
    if (int rc = fused3_args(d, pl, w, packed, csr, node_feat, edge_feat, src, dst, times, B, B, out_src, out_dst, ws, wl, nullptr, a, f)) return rc;
    a.lut = lut;
    a.tr = tr;
    a.pair_stride = 0;
#ifdef DYGNN_STAMPS
    if (const char* sp = getenv("DYGNN_STAMPS_PTR")) a.stamps = reinterpret_cast<unsigned long long*>(strtoull(sp, nullptr, 0));   // diagnostic build: [4][8][32] device words
#endif
    DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dygformer_fused3<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dygformer_fused3<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    if (f.np == 2 && B <= kSmallBatchPairs && !small_off()) {
        DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dygformer_fused3<4, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        hipLaunchKernelGGL((k_dygformer_fused3<4, true, 4>), dim3((unsigned)B), dim3(256), kLdsBytes, s, a);
    } else if (f.np == 2) hipLaunchKernelGGL((k_dygformer_fused3<4, true>), dim3((unsigned)((B + 1) / 2)), dim3(512), kLdsBytes, s, a);
    else hipLaunchKernelGGL((k_dygformer_fused3<8, true>), dim3((unsigned)B), dim3(512), kLdsBytes, s, a);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

}  // namespace dygnn
