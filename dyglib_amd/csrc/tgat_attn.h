// Attention of ONE node over its k neighbours' input rows, two waves per node (shared by k_tgat_attn_pair, tgat.hip, and by k_tgat_post,
// tgat_chain.hip, which runs it on its own rows and keeps z in LDS).  gfx950 only.
#pragma once
#include "common.h"

namespace dygnn {
namespace attn {

using af4 = __attribute__((ext_vector_type(4))) float;

__device__ __attribute__((noinline)) float cos_libm(float x) { return cosf(x); }
__device__ __forceinline__ float cos_time_t(float x) {      // same range reduction + polynomial as dygformer_fused3.hip
    if (!(fabsf(x) <= 3.0e7f)) return cos_libm(x);
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


// LDS floats the two-waves-per-node attention needs for `waves` waves (waves / 2 nodes at a time): partial scores [waves/2][2][H][KC],
// time staging [waves][KC][TW], probabilities [waves][H][KC]
__host__ __device__ inline int pair_smem_floats(int waves, int H, int KC, int Ft) {
    const int TW = 4 * ((Ft / 4 + 1) / 2);
    return waves * H * KC + waves * KC * TW + waves * H * KC;
}

// Wave `hf` (0 / 1) of node slot `slot` owns the float4 columns 2*lane + hf of the 444-wide input rows, so a lane keeps KC float4 (80 VGPRs
// at k = 20) instead of 2*KC; the kernel is bound by gather latency, so occupancy is what pays.  Phases: (A) the pair's cosines, spread over
// the lanes, into LDS -- BEFORE any gather is in flight, so the out-of-line libm fallback of cos_time_t has nothing live to spill; (B) all k
// gathers back to back; (C) partial scores per half -> LDS -> WORKGROUP BARRIER (every wave of the workgroup must call this function) -> both
// waves run the same softmax on lanes (h, j); (D) weighted sum of the cached rows, stored at zrow + h * zhs (head stride): the node's z
// [H][Dkv] in global memory or in LDS.  A dead slot (live = false: i is a stand-in) stores nothing, or zeros with zero_dead.
// smem: pair_smem_floats(waves, ...) floats; wave = index of the calling wave among `waves`.
template <int KC, bool FULL = false>      // FULL: k == KC, no idle row slots (their index clamps and zero fills drop out)
__device__ __forceinline__ void pair_node(const float* __restrict__ qk, const float* __restrict__ h_lower, const float* __restrict__ node_feat,
                                          const float* __restrict__ edge_feat, const int32_t* __restrict__ lower_ids, const int32_t* __restrict__ nbr_eid,
                                          const float* __restrict__ nbr_dt, const float* __restrict__ tw, const float* __restrict__ tb, int64_t n, int k, int Fn,
                                          int Fe, int Ft, int H, float scale, const int32_t* __restrict__ lower_map, int64_t i, bool live, bool zero_dead,
                                          int wave, int waves, int lane, float* smem, float* zrow, int zhs) {
    const int hf = wave & 1, slot = wave >> 1;
    const int Dkv = Fn + Fe + Ft, D4 = Dkv >> 2, T0 = (Fn + Fe) >> 2, NT4 = Ft >> 2;
    const int par = (hf - T0) & 1;                          // parity, inside the time block, of the time columns this half owns
    const int ntc = (NT4 - par + 1) >> 1;                   // how many of them
    const int TW = 4 * ((NT4 + 1) >> 1);                    // floats per row of the time staging area
    float* part = smem;                                                                     // [waves/2 slots][2 halves][H][KC]
    float* tf = part + waves * H * KC + (size_t)wave * KC * TW;                             // [waves][KC][TW]
    float* pw = part + waves * H * KC + (size_t)waves * KC * TW + (size_t)wave * H * KC;    // [waves][H][KC]
    const int64_t r0 = i * k;
    // (A) time encodings of this half's columns for all k rows
    if (lane < 4 * ntc) {
        const int f = 4 * (2 * (lane >> 2) + par) + (lane & 3);
        const float w = tw[f], b = tb[f];
#pragma unroll 1
        for (int j = 0; j < k; ++j) tf[j * TW + lane] = cos_time_t(fmaf(nbr_dt[r0 + j], w, b));
    }
    // (B) gathers
    const int col = 2 * lane + hf, kk = 4 * col;
    const bool vcol = col < D4;
    const int cls = !vcol ? 3 : kk < Fn ? 0 : kk < Fn + Fe ? 1 : 2;
    const af4 zero = af4{0.f, 0.f, 0.f, 0.f};
    af4 xs[KC];
    if (cls <= 1) {
        const float* bp = cls == 0 ? (h_lower ? h_lower : node_feat) + kk : edge_feat + (kk - Fn);
        const size_t st = cls == 0 ? (size_t)Fn : (size_t)Fe;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const int64_t r = r0 + ((FULL || j < k) ? j : 0);
            const int64_t nrow = h_lower ? (lower_map ? (int64_t)lower_map[n + r] : n + r) : (int64_t)lower_ids[n + r];
            const int64_t erow = nbr_eid[r];
            xs[j] = *reinterpret_cast<const af4*>(bp + (cls == 0 ? nrow : erow) * st);
        }
    } else {
#pragma unroll
        for (int j = 0; j < KC; ++j) xs[j] = zero;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (cls == 2) {
        const int lt = (col - T0) >> 1;
#pragma unroll
        for (int j = 0; j < KC; ++j) xs[j] = *reinterpret_cast<const af4*>(tf + ((FULL || j < k) ? j : 0) * TW + 4 * lt);
    }
    if constexpr (!FULL) {
#pragma unroll
        for (int j = 0; j < KC; ++j)
            if (j >= k) xs[j] = zero;                       // idle row slots contribute exact zeros
    }
    // (C) partial scores of this half.  The KC sums over the 64 lanes are a reduce-scatter, not KC all-reduces: across lane bit 5 a lane
    // keeps half of its values and sends the other half (one exchange adds TWO rows' partial sums), across bit 4 again, then the 16 lanes
    // of a row of lanes finish the KC/4 values they are left with: 35 cross-lane exchanges for 20 rows instead of 120 (this kernel is
    // bound by instruction issue, not by its gathers: profiles/r02_tgat_attn_notes.md).  The sums are formed in the same order for every
    // node, whatever the batch.
    constexpr int N1 = KC / 2, N2 = (N1 + 1) / 2;
    static_assert(KC % 2 == 0, "KC must be even");
    const bool up32 = (lane & 32) != 0, up16 = (lane & 16) != 0;      // where the folds below leave which rows
    auto dot4 = [](const af4 a, const af4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); };
    for (int h = 0; h < H; ++h) {
        const af4 q = vcol ? *reinterpret_cast<const af4*>(qk + ((size_t)i * H + h) * Dkv + kk) : zero;
        // rows p, p + N2 (even / odd row of lanes) and p + N1, p + N1 + N2 (upper half) meet in one value per lane.  v_permlane32_swap(a, b)
        // moves the upper half of a and the lower half of b into each other's place: a' + b' = (a summed across lane bit 5) in the lower
        // lanes, (b summed) in the upper ones -- one swap and one add for two rows, no select, no LDS; v_permlane16_swap the same across bit 4.
        float s2[N2];
        auto fold32 = [](float a, float b) {
            const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
            const unsigned r0 = r[0], r1 = r[1];
            return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
        };
        auto fold16 = [](float a, float b) {
            const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
            const unsigned r0 = r[0], r1 = r[1];
            return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
        };
#pragma unroll
        for (int p = 0; p < N2; ++p) {
            const bool two = p + N2 < N1;                   // (compile time: the odd one out when N1 is odd)
            const float t0 = fold32(dot4(q, xs[p]), dot4(q, xs[p + N1]));
            const float t1 = two ? fold32(dot4(q, xs[two ? p + N2 : p]), dot4(q, xs[two ? p + N1 + N2 : p])) : 0.f;
            s2[p] = fold16(t0, t1);
        }
        // ... and the 16 lanes of a row of lanes: rotations inside the row (DPP row_ror 8, 4, 2, 1: every lane ends with the sum), no LDS
#pragma unroll
        for (int p = 0; p < N2; ++p) {
            float v = s2[p];
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
            s2[p] = v;
        }
        if ((lane & 15) == 0) {                             // lane 16 r of row r = 2 (upper half) + (odd row) holds rows p + N2 * odd + N1 * upper
#pragma unroll
            for (int p = 0; p < N2; ++p) {
                const int jh = p + (up16 ? N2 : 0);
                if (jh < N1) part[((slot * 2 + hf) * H + h) * KC + jh + (up32 ? N1 : 0)] = s2[p];
            }
        }
    }
    __syncthreads();
    {   // softmax over the k neighbours: lane = 32 * head + row (modules.py:173 scale, :176-184 mask)
        const int h = lane >> 5, j = lane & 31;
        const bool on = h < H && j < k;
        float s = -INFINITY;
        if (on) {
            s = (part[((slot * 2 + 0) * H + h) * KC + j] + part[((slot * 2 + 1) * H + h) * KC + j]) * scale;
            if (lower_ids[n + r0 + j] == 0) s = -1e10f;
        }
        float mx = s;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float e = on ? expf(s - mx) : 0.f;
        float sum = e;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (on) pw[h * KC + j] = e * (1.0f / sum);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // (D) z_ih = sum_j p_ijh x_ij for this half's columns
    for (int h = 0; h < H; ++h) {
        af4 za = zero;
#pragma unroll
        for (int j = 0; j < KC; ++j) {                 // no early exit: a break keeps xs[] from being promoted to registers
            const float p = (FULL || j < k) ? pw[h * KC + j] : 0.f;
            za.x = fmaf(p, xs[j].x, za.x); za.y = fmaf(p, xs[j].y, za.y); za.z = fmaf(p, xs[j].z, za.z); za.w = fmaf(p, xs[j].w, za.w);
        }
        if (vcol && (live || zero_dead)) *reinterpret_cast<af4*>(zrow + (size_t)h * zhs + kk) = live ? za : zero;
    }
}

}  // namespace attn
}  // namespace dygnn
