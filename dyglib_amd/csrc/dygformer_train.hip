// Training path of the DyGFormer hot path: a forward that keeps what the backward pass needs and the
// backward pass itself, so train_link_prediction.py:229-257 (forward, loss.backward(), optimizer.step()) runs on hand-written
// HIP kernels.  models/DyGFormer.py:68-194 and :418-461 in train mode: dropout on the attention probabilities
// (nn.MultiheadAttention(dropout=...), :429), on the attention output and on the FFN (:456-460).
//
// Design: activations live in HBM (dense [B*T][.] rows, T = tokens per pair of THIS call), every product is one general
// fp32-MFMA GEMM kernel (k_mm: any transposition, strided batches for the per-(pair, head) attention products), the rest is
// row-wise / element-wise kernels.  Dropout masks are not stored: both passes draw them from a counter-based hash of
// (seed, site, element).  Gradients are produced for every parameter; the feature tables get none (the reference keeps
// them as constants, models/DyGFormer.py:28-29).
// This is the first, unfused version (correctness + a working training loop); inference uses dygformer_fused3.hip.
#include <cstdlib>
#include "dygformer_layout.h"
#include "gemm.h"
#include "dropout.h"

namespace dygnn {

int window_lengths_device(const Dims& d, const dygnn_csr* csr, const int64_t* src, const int64_t* dst, const double* times,
                          int64_t B, int64_t G, char* ws, const WorkspaceLayout& wl, hipStream_t s);   // dygformer_generic.hip
namespace train { struct TrainOut; }
bool fused3_supported(const Dims& d);                                                                   // dygformer_fused3.hip
namespace train { struct Drop; }
int attn_backward_fused3(const Dims& d, const PackedLayout& pl, const float* packed, int l, int64_t B, int T, float* dX, const float* X, const float* m0,
                         const float* r0, const float* qkv, const float* P, const float* Pd, float* dAo, float* dQKV, float* dgamma, float* dbeta,
                         const train::Drop& dr, hipStream_t s);
int ffn_backward_fused3(const Dims& d, const PackedLayout& pl, const float* packed, int l, int64_t M, float* dX, const float* hpre, const float* x1,
                        const float* m1, const float* r1, float* dF2, float* dH, float* dgamma, float* dbeta, const train::Drop& dr, hipStream_t s);
int forward_fused3_train(const Dims& d, const PackedLayout& pl, const dygnn_dygformer_weights* w, const float* packed, const dygnn_csr* csr,
                         const float* node_feat, const float* edge_feat, const int64_t* src, const int64_t* dst, const double* times, int64_t B,
                         const float* lut, float* out_src, float* out_dst, char* ws, const WorkspaceLayout& wl, const train::TrainOut& tr, hipStream_t s);

namespace train {

using f4 = __attribute__((ext_vector_type(4))) float;

// ------------------------------------------------------------------------------------------------
// C = alpha * op(A) . op(B) (+ bias[n]) (+ beta * C), batched: z = zb * H + zh selects A + zb*sAb + zh*sAh etc.
//   op(A)[m][k] = transA ? A[k*lda + m] : A[m*lda + k] ;  op(B)[k][n] = transB ? B[n*ldb + k] : B[k*ldb + n]
// 64x64 tile per workgroup (4 waves x 16 rows x 64 columns), K in steps of 16 through LDS, v_mfma_f32_16x16x4_f32.
// ------------------------------------------------------------------------------------------------
struct MM {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc, transA, transB;
    float alpha, beta;
    int H; int64_t sAb, sAh, sBb, sBh, sCb, sCh;
    int relu;               // epilogue max(v, 0) (not combined with split-K)
    int ksplit, kchunk;     // split-K: blockIdx.z = batch * ksplit + part; part sums k in [part*kchunk, +kchunk) and adds atomically
    float* colsum;          // transA products only: colsum[m] += sum_k op(A)[m][k] (the bias gradient next to a weight gradient), or null
    const int32_t* m_dev;   // optional device-side row count (<= M): workgroups whose rows all lie beyond it exit at once
    int vecC;               // C rows may be written as aligned float4 (pointer, ldc and batch strides multiples of 4 floats)
    const int32_t* a_rows;  // optional (k-contiguous A, LDS-DMA kernel only): row m of op(A) is A[a_rows[m]] — the product gathers its rows itself
};

__global__ __launch_bounds__(256) void k_mm(const MM p) {
    __shared__ float As[16][64 + 16];      // row stride 80 = 16 mod 64 banks: the four k-rows a wave reads at once do not collide
    __shared__ float Bs[16][64 + 16];
    const int z = blockIdx.z / p.ksplit, ks = blockIdx.z % p.ksplit, zb = z / p.H, zh = z % p.H;
    const float* A = p.A + zb * p.sAb + zh * p.sAh;
    const float* Bm = p.B + zb * p.sBb + zh * p.sBh;
    float* C = p.C + zb * p.sCb + zh * p.sCh;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    if (p.m_dev && m0 >= *p.m_dev) return;
    const int kbeg = ks * p.kchunk, kend = kbeg + p.kchunk < p.K ? kbeg + p.kchunk : p.K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    f4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + 256 * e;
            int m, k;
            if (p.transA) { m = idx & 63; k = idx >> 6; } else { m = idx >> 4; k = idx & 15; }
            float v = 0.f;
            if (m0 + m < p.M && k0 + k < kend) v = p.transA ? A[(size_t)(k0 + k) * p.lda + m0 + m] : A[(size_t)(m0 + m) * p.lda + k0 + k];
            As[k][m] = v;
            int n, kb;
            if (p.transB) { n = idx >> 4; kb = idx & 15; } else { n = idx & 63; kb = idx >> 6; }
            float w = 0.f;
            if (n0 + n < p.N && k0 + kb < kend) w = p.transB ? Bm[(size_t)(n0 + n) * p.ldb + k0 + kb] : Bm[(size_t)(k0 + kb) * p.ldb + n0 + n];
            Bs[kb][n] = w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float a = As[4 * kk + g][16 * wave + c];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bs[4 * kk + g][16 * nt + c], acc[nt], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + 16 * nt + c;
        if (n >= p.N) continue;
        const float bv = (p.bias && ks == 0) ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * wave + 4 * g + r;
            if (m >= p.M) continue;
            float v = p.alpha * acc[nt][r] + bv;
            if (p.ksplit > 1) { atomicAdd(&C[(size_t)m * p.ldc + n], v); continue; }
            if (p.beta != 0.f) v += p.beta * C[(size_t)m * p.ldc + n];
            if (p.relu) v = fmaxf(v, 0.f);
            C[(size_t)m * p.ldc + n] = v;
        }
    }
}

// The large products: 128 x (32*WN) tile per workgroup, 4 waves as 2 x 2, each wave 4 x WN accumulator tiles; operands go
// through LDS in k-major layout ([16 k][tile + 4]) with float4 global loads when pointers / leading dimensions allow
// (vecA / vecB), so one k-step of 16 costs a wave 8 or 6 LDS reads per 16 or 8 MFMAs instead of k_mm's 5 per 4.
template <int WN, int WM>
__global__ __launch_bounds__(256) void k_mm_big(const MM p, const int vecA, const int vecB) {
    constexpr int BM = 32 * WM, BN = 32 * WN, NB = BN / 64, NA = BM / 64;      // tile BM x BN; wave = WM x WN accumulator tiles
    __shared__ float As[2][16][BM + 16];      // double buffered: the global loads of k-step t+1 fly while step t multiplies
    __shared__ float Bs[2][16][BN + 16];
    const int z = blockIdx.z / p.ksplit, ks = blockIdx.z % p.ksplit, zb = z / p.H, zh = z % p.H;
    const float* A = p.A + zb * p.sAb + zh * p.sAh;
    const float* Bm = p.B + zb * p.sBb + zh * p.sBh;
    float* C = p.C + zb * p.sCb + zh * p.sCh;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    if (p.m_dev && m0 >= *p.m_dev) return;        // uniform over the workgroup, before any barrier
    const int kbeg = ks * p.kchunk, kend = kbeg + p.kchunk < p.K ? kbeg + p.kchunk : p.K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = wave >> 1, wx = wave & 1;
    const int c = lane & 15, g = lane >> 4;
    f4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    auto ldA = [&](int m, int k) -> float {      // op(A)[m][k], zero outside
        if (m >= p.M || k >= kend) return 0.f;
        return p.transA ? A[(size_t)k * p.lda + m] : A[(size_t)m * p.lda + k];
    };
    auto ldB = [&](int k, int n) -> float {
        if (n >= p.N || k >= kend) return 0.f;
        return p.transB ? Bm[(size_t)n * p.ldb + k] : Bm[(size_t)k * p.ldb + n];
    };
    f4 ra[NA], rb[NB];
    auto fetch = [&](int k0) {                    // this thread's share of the tiles of k-step k0 -> registers
        if (p.transA) {                           // stored [K][M]: m contiguous; thread -> (k, 4 consecutive m)
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                const int k = (tid / (BM / 4)) + (1024 / BM) * e, m = (tid % (BM / 4)) * 4;
                if (vecA && m0 + m + 3 < p.M && k0 + k < kend) ra[e] = *reinterpret_cast<const f4*>(A + (size_t)(k0 + k) * p.lda + m0 + m);
                else ra[e] = f4{ldA(m0 + m, k0 + k), ldA(m0 + m + 1, k0 + k), ldA(m0 + m + 2, k0 + k), ldA(m0 + m + 3, k0 + k)};
            }
        } else {                                  // stored [M][K]: k contiguous; thread -> (m, 4 consecutive k)
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                const int m = (tid >> 2) + 64 * e, k = (tid & 3) * 4;
                if (vecA && m0 + m < p.M && k0 + k + 3 < kend) ra[e] = *reinterpret_cast<const f4*>(A + (size_t)(m0 + m) * p.lda + k0 + k);
                else ra[e] = f4{ldA(m0 + m, k0 + k), ldA(m0 + m, k0 + k + 1), ldA(m0 + m, k0 + k + 2), ldA(m0 + m, k0 + k + 3)};
            }
        }
        if (!p.transB) {                          // stored [K][N]: n contiguous
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int k = (tid / (BN / 4)) + (1024 / BN) * e, n = (tid % (BN / 4)) * 4;
                if (vecB && n0 + n + 3 < p.N && k0 + k < kend) rb[e] = *reinterpret_cast<const f4*>(Bm + (size_t)(k0 + k) * p.ldb + n0 + n);
                else rb[e] = f4{ldB(k0 + k, n0 + n), ldB(k0 + k, n0 + n + 1), ldB(k0 + k, n0 + n + 2), ldB(k0 + k, n0 + n + 3)};
            }
        } else {                                  // stored [N][K]: k contiguous
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int n = (tid >> 2) + 64 * e, k = (tid & 3) * 4;
                if (vecB && n0 + n < p.N && k0 + k + 3 < kend) rb[e] = *reinterpret_cast<const f4*>(Bm + (size_t)(n0 + n) * p.ldb + k0 + k);
                else rb[e] = f4{ldB(k0 + k, n0 + n), ldB(k0 + k + 1, n0 + n), ldB(k0 + k + 2, n0 + n), ldB(k0 + k + 3, n0 + n)};
            }
        }
    };
    auto stash = [&](int buf) {                   // registers -> LDS tile `buf` (k-major)
        // LDS column of element (k, m) = (m + 16 * (k / 4)) mod BM: the four k-groups a wave writes at once (rows k, k+4, k+8,
        // k+12 of the same m) would otherwise share their banks (row stride = 16 mod 64 -> 4 rows = 0 mod 64): 4-way conflicts
        if (p.transA) {
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                const int k = (tid / (BM / 4)) + (1024 / BM) * e, m = (tid % (BM / 4)) * 4;
                *reinterpret_cast<f4*>(&As[buf][k][(m + 16 * (k >> 2)) & (BM - 1)]) = ra[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                const int m = (tid >> 2) + 64 * e, k = (tid & 3) * 4, mr = (m + 16 * (k >> 2)) & (BM - 1);
                As[buf][k][mr] = ra[e].x; As[buf][k + 1][mr] = ra[e].y; As[buf][k + 2][mr] = ra[e].z; As[buf][k + 3][mr] = ra[e].w;
            }
        }
        if (!p.transB) {
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int k = (tid / (BN / 4)) + (1024 / BN) * e, n = (tid % (BN / 4)) * 4;
                *reinterpret_cast<f4*>(&Bs[buf][k][(n + 16 * (k >> 2)) & (BN - 1)]) = rb[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int n = (tid >> 2) + 64 * e, k = (tid & 3) * 4, nr = (n + 16 * (k >> 2)) & (BN - 1);
                Bs[buf][k][nr] = rb[e].x; Bs[buf][k + 1][nr] = rb[e].y; Bs[buf][k + 2][nr] = rb[e].z; Bs[buf][k + 3][nr] = rb[e].w;
            }
        }
    };
    fetch(kbeg);
    stash(0);
    __syncthreads();
    int buf = 0;
    static_assert(BM == 64, "the column-sum side product assumes 64-row tiles (wave w sums k rows 4w .. 4w+3 of column `lane`)");
    const bool want_colsum = p.colsum != nullptr && blockIdx.y == 0;
    float csum = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        const bool more = k0 + 16 < kend;
        if (more) fetch(k0 + 16);
        if (want_colsum) {                        // rows beyond kend / M were stored as zeros
            const int col = (lane + 16 * wave) & (BM - 1);
            csum += (As[buf][4 * wave][col] + As[buf][4 * wave + 1][col]) + (As[buf][4 * wave + 2][col] + As[buf][4 * wave + 3][col]);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float a[WM], b[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) a[i] = As[buf][4 * kk + g][(16 * WM * wy + 16 * i + c + 16 * kk) & (BM - 1)];
#pragma unroll
            for (int j = 0; j < WN; ++j) b[j] = Bs[buf][4 * kk + g][(16 * WN * wx + 16 * j + c + 16 * kk) & (BN - 1)];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);                 // the other buffer was last read one iteration ago, before the barrier below
        __syncthreads();
        buf ^= 1;
    }
    if (want_colsum && m0 + lane < p.M) atomicAdd(&p.colsum[m0 + lane], csum);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = n0 + 16 * WN * wx + 16 * j + c;
        if (n >= p.N) continue;
        const float bv = (p.bias && ks == 0) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * WM * wy + 16 * i + 4 * g + r;
                if (m >= p.M) continue;
                float v = p.alpha * acc[i][j][r] + bv;
                if (p.ksplit > 1) { atomicAdd(&C[(size_t)m * p.ldc + n], v); continue; }
                if (p.beta != 0.f) v += p.beta * C[(size_t)m * p.ldc + n];
                if (p.relu) v = fmaxf(v, 0.f);
                C[(size_t)m * p.ldc + n] = v;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// The same 64 x 64 tile product with both operands fed by LDS-DMA (global_load_lds: no VGPR round trip) into a ring of NS
// stages of one 16-wide k-step each: NS - 1 steps of loads are in flight per workgroup instead of one, which is what the
// latency-bound k-loop of k_mm_big lacks (its second register stage cost occupancy instead).  One barrier per k-step.
// Tile of an operand in a stage: four 1-KiB blocks of 16 rows x 16 k, block w written by wave w with one DMA instruction
// (lane L lands at byte 16 L of the block).  Lane -> element mapping per orientation, chosen so that the MFMA operand
// reads are conflict-free in the PERMUTED k order (MFMA step r takes k = 4 g + r from both operands):
//   k-contiguous operand ([row][k] in memory): lane L loads the float4 (row = L & 15, k = 4 (L >> 4) ..+3)
//       -> lane (c, g) reads the granule at float 4 (16 g + c) with one ds_read_b128
//   row-contiguous operand ([k][row] in memory): lane L loads (k = 4 ((L >> 2) & 3) + (L >> 4), rows 4 (L & 3) ..+3)
//       -> element (row c, k = 4 g + r) sits at float 4 (16 r + 4 g + (c >> 2)) + (c & 3)     (banks 16 g + c per half-wave)
// Rows / k beyond the matrix load from a 16-byte page of zeros (DMA cannot zero-fill), so tails need no special path as long
// as a float4 never straddles a bound (K % 4 == 0, and M % 4 == 0 / N % 4 == 0 for a row-contiguous A / B: checked by mm()).
// ------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

template <int KA, int KB, int WM, int WN, int NS>      // KA / KB = 1: the operand is k-contiguous in memory (A: !transA, B: transB); tile 32 WM x 32 WN
__global__ __launch_bounds__(256) void k_mm_dma(const MM p) {
    constexpr int BM = 32 * WM, BN = 32 * WN, STAGE = 16 * (BM + BN);      // floats per stage: A tile then B tile
    constexpr int LOADS = (WM + WN) / 2;                                    // DMA instructions per wave and stage
    __shared__ __attribute__((aligned(16))) float ring[NS * STAGE];
    const int z = blockIdx.z / p.ksplit, ks = blockIdx.z % p.ksplit, zb = z / p.H, zh = z % p.H;
    const float* A = p.A + zb * p.sAb + zh * p.sAh;
    const float* Bm = p.B + zb * p.sBb + zh * p.sBh;
    float* C = p.C + zb * p.sCb + zh * p.sCh;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    if (p.m_dev && m0 >= *p.m_dev) return;
    const int kbeg = ks * p.kchunk, kend = kbeg + p.kchunk < p.K ? kbeg + p.kchunk : p.K;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wy = wave >> 1, wx = wave & 1, c = lane & 15, g = lane >> 4;
    f4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    // this lane's source coordinates inside a 16-row block (wave w loads blocks w, w + 4, ...)
    const int rA = KA ? (lane & 15) : 4 * (lane & 3), kA = KA ? 4 * (lane >> 4) : 4 * ((lane >> 2) & 3) + (lane >> 4);
    const int rB = KB ? (lane & 15) : 4 * (lane & 3), kB = KB ? 4 * (lane >> 4) : 4 * ((lane >> 2) & 3) + (lane >> 4);
    const int nsteps = (kend - kbeg + 15) >> 4;
    // rows of op(A) this lane loads (k-contiguous A): with a row table the product gathers them (rows beyond the live count read zeros: the
    // table's entries there are not written)
    int64_t arow[WM / 2];
#pragma unroll
    for (int u = 0; u < WM / 2; ++u) {
        const int r = m0 + 16 * (wave + 4 * u) + rA;
        const int mlim = (p.a_rows && p.m_dev) ? (*p.m_dev < p.M ? *p.m_dev : p.M) : p.M;
        arow[u] = r < mlim ? ((KA && p.a_rows) ? (int64_t)p.a_rows[r] : (int64_t)r) : -1;
    }
    auto issue = [&](int t) {                 // k-step t -> ring slot t % NS (beyond the last step: zeros, which keeps vmcnt uniform)
        const int k0 = kbeg + 16 * t;
        float* slot = ring + (t % NS) * STAGE;
#pragma unroll
        for (int u = 0; u < WM / 2; ++u) {
            const int blk = wave + 4 * u, r = m0 + 16 * blk + rA;
            const float* src = (t < nsteps && arow[u] >= 0 && k0 + kA < kend) ? (KA ? A + (size_t)arow[u] * p.lda + k0 + kA : A + (size_t)(k0 + kA) * p.lda + r) : g_zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(slot + 256 * blk), 16, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < WN / 2; ++u) {
            const int blk = wave + 4 * u, r = n0 + 16 * blk + rB;
            const float* src = (t < nsteps && r < p.N && k0 + kB < kend) ? (KB ? Bm + (size_t)r * p.ldb + k0 + kB : Bm + (size_t)(k0 + kB) * p.ldb + r) : g_zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(slot + 16 * BM + 256 * blk), 16, 0, 0);
        }
    };
    const bool want_colsum = p.colsum != nullptr && blockIdx.y == 0;      // only with a row-contiguous A (weight gradients), 64-row tiles
    float csum = 0.f;
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue(t);
    for (int t = 0; t < nsteps; ++t) {
        // this wave's loads of step t have landed; then a BARE barrier (everybody's have, and everybody is done with step t - 1):
        // __syncthreads() would add a fence = s_waitcnt vmcnt(0), i.e. wait for the whole ring and undo the pipelining
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(LOADS * (NS - 2)) : "memory");
        issue(t + NS - 1);                                                      // into the slot step t - 1 used
        const float* as = ring + (t % NS) * STAGE;
        const float* bs = as + 16 * BM;
        if (!KA && WM == 2 && want_colsum) {  // element (row = lane, k) of the A tile: block lane >> 4, row c; this wave sums k = 4 wave .. +3
            const float* q = as + 256 * g + 4 * (16 * wave + (c >> 2)) + (c & 3);
            csum += (q[0] + q[16]) + (q[32] + q[48]);
        }
        // Permuted k order: MFMA step r multiplies k = 4 g + r of both operands.  A k-contiguous operand then needs ONE conflict-free
        // ds_read_b128 per tile (lane (c, g) takes the whole granule of row c; ds_read_b32 of that layout would be 2-way conflicted,
        // its banks being (a/4) mod 32 per half-wave); a row-contiguous operand reads element (row c, k = 4 g + r) at float
        // 4 (16 r + 4 g + (c >> 2)) + (c & 3): banks 16 g + c inside a half-wave, conflict-free.
        float a[WM][4], b[WN][4];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            if constexpr (KA) {
                const f4 v = *reinterpret_cast<const f4*>(as + 256 * (WM * wy + i) + 4 * (16 * g + c));
                a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) a[i][r] = as[256 * (WM * wy + i) + 4 * (16 * r + 4 * g + (c >> 2)) + (c & 3)];
            }
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            if constexpr (KB) {
                const f4 v = *reinterpret_cast<const f4*>(bs + 256 * (WN * wx + j) + 4 * (16 * g + c));
                b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) b[j][r] = bs[256 * (WN * wx + j) + 4 * (16 * r + 4 * g + (c >> 2)) + (c & 3)];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][r], b[j][r], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the trailing zero-page loads must not outlive the workgroup's LDS
    if (WM == 2 && want_colsum && m0 + lane < p.M) atomicAdd(&p.colsum[m0 + lane], csum);
    if (WM == 2 && WN == 2 && p.vecC && p.ksplit == 1 && p.beta == 0.f) {
        // coalesced epilogue: the tile goes through LDS (the ring is free now) and leaves as float4 rows -- 16 lanes write one 256-byte
        // row segment, 4 store instructions per wave instead of 16 scalar ones that each touch four rows
        asm volatile("s_barrier" ::: "memory");                   // every wave is done reading the ring
        constexpr int LD = 64 + 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nl = 32 * wx + 16 * j + c;
            const float bv = (p.bias && n0 + nl < p.N) ? p.bias[n0 + nl] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = p.alpha * acc[i][j][r] + bv;
                    if (p.relu) v = fmaxf(v, 0.f);
                    ring[(32 * wy + 16 * i + 4 * g + r) * LD + nl] = v;
                }
        }
        __syncthreads();
        const int col = 4 * (tid & 15);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = 16 * ps + (tid >> 4), m = m0 + row, n = n0 + col;
            if (m >= p.M || n >= p.N) continue;
            const f4 v = *reinterpret_cast<const f4*>(ring + row * LD + col);
            float* dst = C + (size_t)m * p.ldc + n;
            if (n + 3 < p.N) *reinterpret_cast<f4*>(dst) = v;
            else {
                dst[0] = v.x;
                if (n + 1 < p.N) dst[1] = v.y;
                if (n + 2 < p.N) dst[2] = v.z;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = n0 + 16 * WN * wx + 16 * j + c;
        if (n >= p.N) continue;
        const float bv = (p.bias && ks == 0) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * WM * wy + 16 * i + 4 * g + r;
                if (m >= p.M) continue;
                float v = p.alpha * acc[i][j][r] + bv;
                if (p.ksplit > 1) { atomicAdd(&C[(size_t)m * p.ldc + n], v); continue; }
                if (p.beta != 0.f) v += p.beta * C[(size_t)m * p.ldc + n];
                if (p.relu) v = fmaxf(v, 0.f);
                C[(size_t)m * p.ldc + n] = v;
            }
    }
}

template <int WM, int WN, int NS>
static void launch_mm_dma(hipStream_t s, const MM& p, bool tA, bool tB, int batch) {
    const dim3 grid((unsigned)ceil_div(p.M, 32 * WM), (unsigned)ceil_div(p.N, 32 * WN), (unsigned)batch);
    if (!tA && tB) hipLaunchKernelGGL((k_mm_dma<1, 1, WM, WN, NS>), grid, dim3(256), 0, s, p);
    else if (!tA && !tB) hipLaunchKernelGGL((k_mm_dma<1, 0, WM, WN, NS>), grid, dim3(256), 0, s, p);
    else if (tA && !tB) hipLaunchKernelGGL((k_mm_dma<0, 0, WM, WN, NS>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_mm_dma<0, 1, WM, WN, NS>), grid, dim3(256), 0, s, p);
}

static int colsum(hipStream_t s, const float* A, int lda, int64_t M, int N, float* out, bool accumulate = false);

int mm(hipStream_t s, const float* A, int lda, bool tA, const float* B, int ldb, bool tB, float* C, int ldc, int M, int N, int K, const float* bias, float alpha,
       float beta, int batch, int H, int64_t sAb, int64_t sAh, int64_t sBb, int64_t sBh, int64_t sCb, int64_t sCh, bool relu, bool c_is_zero, float* colsum_out, const int32_t* m_dev, bool a_kpad,
       const int32_t* a_rows) {
    if (M <= 0 || N <= 0 || batch <= 0) return DYGNN_OK;
    MM p{A, B, C, bias, M, N, K, lda, ldb, ldc, tA ? 1 : 0, tB ? 1 : 0, alpha, beta, H, sAb, sAh, sBb, sBh, sCb, sCh, relu ? 1 : 0, 1, K, nullptr, m_dev,
         ((reinterpret_cast<uintptr_t>(C) & 15) == 0 && ldc % 4 == 0 && sCb % 4 == 0 && sCh % 4 == 0) ? 1 : 0, a_rows};
    if (colsum_out) {                 // rides along inside the 64-row-tile kernel; anything else gets the stand-alone reduction
        if (tA && batch == 1 && M >= 48 && N >= 48) p.colsum = colsum_out;
        else if (int rc = colsum(s, A, lda, K, M, colsum_out)) return rc;
    }
    // weight gradients: small output, K = all rows of the call -> split K over workgroups, partial sums meet by atomicAdd
    if (batch == 1 && K >= 2048 && !relu) {
        p.kchunk = 256;
        p.ksplit = (K + p.kchunk - 1) / p.kchunk;
        if (beta == 0.f && !c_is_zero) DYGNN_HIP(hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), (size_t)M, s));
        p.beta = 0.f;
    }
    batch *= p.ksplit;
    if (M >= 48 && N >= 48) {
        const int vecA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0 && lda % 4 == 0 && sAb % 4 == 0 && sAh % 4 == 0) ? 1 : 0;
        const int vecB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0 && ldb % 4 == 0 && sBb % 4 == 0 && sBh % 4 == 0) ? 1 : 0;
        // Tile choice, measured on the TGAT / TGN / training shapes (M = 50 .. 270,000, N = 136 .. 800, K = 136 .. 25,600 split):
        //  * 64-row tiles always: the k-loop is latency-bound (one global -> LDS hop per 16-wide k-step), so what pays is more resident
        //    workgroups per CU, not more MFMAs per LDS read; 128-row tiles were slower at every size (TGAT -7 %, training -5 %);
        //  * 64-column tiles unless 128-column tiles pad clearly less (N = 272 pads to 320 instead of 384): TGN +38 %, TGAT +14 %.
        // both operands by LDS-DMA when no float4 can straddle a bound (see k_mm_dma)
        const char* dma_env = getenv("DYGNN_MM_DMA");
        // (Measured and not kept, round 2: the 32 x 208-per-wave register tile of k_dw_grouped applied to the tall activation x weight products
        //  of TGAT — M ~ 10^5, N and K of 136 .. 444: 26 MFMAs per 15 LDS operand reads — ran them in the SAME time as these 64 x 64 tiles
        //  (16.5 vs 16.1 ms over a profile): with K this short a workgroup lives for 9 .. 28 k-steps and its ring fill and tile write-out,
        //  not the k-loop's MFMA density, set the pace.)
        // a_kpad: A is k-contiguous with rows padded to a multiple of 4 floats (zeros behind K): its last float4 stays inside the row
        const bool kok = K % 4 == 0 || (a_kpad && !tA && !tB && lda >= ((K + 3) & ~3));
        if (vecA && vecB && kok && (!tA || M % 4 == 0) && (tB || N % 4 == 0) && !(dma_env && dma_env[0] == '0')) {
            // measured: 64 x 64 tiles with a 4-stage ring; 3 stages tie, 6 / 8 stages and 128 x 128 tiles (3 stages) lose occupancy and are slower
            launch_mm_dma<2, 2, 4>(s, p, tA, tB, batch);
            DYGNN_LAUNCH_CHECK();
            return DYGNN_OK;
        }
        if (a_rows) { set_error("mm: a row table needs the LDS-DMA kernel (aligned k-contiguous operands, K %% 4 == 0)"); return DYGNN_E_UNSUPPORTED; }
        const double w64 = (double)(ceil_div(N, 64) * 64) / N, w128 = (double)(ceil_div(N, 128) * 128) / N;
        if (N <= 64 || w64 * 0.95 < w128)
            hipLaunchKernelGGL((k_mm_big<2, 2>), dim3((unsigned)ceil_div(M, 64), (unsigned)ceil_div(N, 64), (unsigned)batch), dim3(256), 0, s, p, vecA, vecB);
        else
            hipLaunchKernelGGL((k_mm_big<4, 2>), dim3((unsigned)ceil_div(M, 64), (unsigned)ceil_div(N, 128), (unsigned)batch), dim3(256), 0, s, p, vecA, vecB);
        DYGNN_LAUNCH_CHECK();
        return DYGNN_OK;
    }
    if (a_rows) { set_error("mm: a row table needs at least 48 rows and columns"); return DYGNN_E_UNSUPPORTED; }
    hipLaunchKernelGGL(k_mm, dim3((unsigned)ceil_div(M, 64), (unsigned)ceil_div(N, 64), (unsigned)batch), dim3(256), 0, s, p);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}


// ------------------------------------------------------------------------------------------------
// Weight gradients, all of them in ONE launch: C_p[m][n] += sum_k A_p[k][m] * B_p[k][n] for a list of problems p that share K (= all
// token rows of the call).  Both operands are k-major ([token][feature] activations / activation gradients), the outputs are small
// (<= 800 x 200) and K is long, so the work is cut as (problem, 128-row group of m, n-chunk of <= 208 columns, K-split) and every
// workgroup runs its whole K-range with 26 accumulator tiles per wave (32 rows x 208 columns): 26 MFMAs per 15 LDS operand reads,
// against 4 per 4 in the general kernel, and the partial tiles meet by float atomics ONCE per (workgroup, tile) — with all problems of
// both layers in one grid a K-split of ~11 fills the chip, where the general path split every product 100 ways (400 MB of atomics
// per step at the chip-wide 1.3 TB/s atomic rate).  Operands arrive by LDS-DMA in the row-contiguous block layout of k_mm_dma
// (conflict-free ds_read_b32 in the permuted k order); A blocks are private to their wave, the 13 B blocks are shared.  The column
// sums of A (the bias gradient that belongs to the weight gradient) fall out of the A operand registers.  The finished tiles leave
// through a wave-private LDS image so that every atomic wave-instruction adds 256 contiguous bytes (MI355X_MICROARCH.md, global float
// atomics: 64 lanes in 64 rows run 17x slower).
// ------------------------------------------------------------------------------------------------
struct DwProblem { const float* A; const float* B; float* C; float* colsum; int lda, ldb, ldc, M, N, ncw; };
constexpr int kDwMaxProblems = 4 * DYGNN_MAX_LAYERS + 4, kDwMaxItems = 32 * DYGNN_MAX_LAYERS + 16, kDwNS = 3, kDwStage = 22 * 256, kDwLdsBytes = kDwNS * kDwStage * 4;
static_assert(kDwMaxProblems <= (1 << 20), "item code: problem index above bit 12");
struct DwArgs {
    DwProblem prob[kDwMaxProblems];
    unsigned int item[kDwMaxItems];        // problem << 12 | m-group << 6 | n-chunk (up to kDwMaxProblems = 36 problems: a 16-bit code held 16)
    int nitems, K, kchunk;
};
__device__ __forceinline__ void dw_dma(const float* gsrc_lane, int lds_float_off_uniform) {      // see v3::dma_frag (dygformer_fused3.hip)
    const unsigned m0v = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_groupstaticsize() + 4u * (unsigned)lds_float_off_uniform);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc_lane), "s"(m0v) : "memory", "m0");
}
__global__ __launch_bounds__(256, 2) void k_dw_grouped(const DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float dw_lds[];
    const int it = blockIdx.x % a.nitems, ks = blockIdx.x / a.nitems;
    const unsigned code = a.item[it];
    const DwProblem& P = a.prob[code >> 12];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int m0 = 128 * (int)((code >> 6) & 63) + 32 * wave, n0 = P.ncw * (int)(code & 63);
    const int ncols = P.N - n0 < P.ncw ? P.N - n0 : P.ncw;
    const bool act = m0 < P.M;
    const int kbeg = ks * a.kchunk, kend = kbeg + a.kchunk < a.K ? kbeg + a.kchunk : a.K;
    const int nst = (kend - kbeg + 15) >> 4;
    // this lane's piece of a 16 x 16 block: k = kL, rows / columns 4 (lane & 3) .. +3.  Rows / columns outside the matrix read the zero
    // page with stride 0, so the per-step address update is one add and nothing is loaded from the argument block inside the loop
    const int kL = 4 * ((lane >> 2) & 3) + (lane >> 4), rL = 4 * (lane & 3);
    const int lda = P.lda, ldb = P.ldb;
    const float* pa[2]; const float* pb[4];
    size_t sa[2], sb[4];
    int lb[4];                                           // LDS block of this wave's B loads (21 = dummy)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = m0 + 16 * i + rL;
        const bool ok = row < P.M;
        pa[i] = ok ? P.A + (size_t)(kbeg + kL) * lda + row : g_zero_page;
        sa[i] = ok ? (size_t)16 * lda : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = wave + 4 * u;                      // blocks 0 .. 12; u = 3 exists for wave 0 only
        const int col = 16 * j + rL;
        const bool ok = j < 13 && col < ncols;
        lb[u] = j < 13 ? 8 + j : 21;
        pb[u] = ok ? P.B + (size_t)(kbeg + kL) * ldb + n0 + col : g_zero_page;
        sb[u] = ok ? (size_t)16 * ldb : 0;
    }
    const int nfull = (kend - kbeg) >> 4;                // steps whose 16 k all exist
    int slot = 0;                                        // ring slot of the next issue, floats
    auto issue = [&](int t) {                            // k-step t -> next ring slot; six DMAs per wave whatever t (uniform vmcnt)
        if (t < nfull) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { dw_dma(pa[i], slot + 256 * (2 * wave + i)); pa[i] += sa[i]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { dw_dma(pb[u], slot + 256 * lb[u]); pb[u] += sb[u]; }
        } else {                                         // the K tail of the last split, and the two issues beyond the end: zeros
            const bool kin = t < nst && kbeg + 16 * t + kL < kend;
#pragma unroll
            for (int i = 0; i < 2; ++i) dw_dma(kin ? pa[i] : g_zero_page, slot + 256 * (2 * wave + i));
#pragma unroll
            for (int u = 0; u < 4; ++u) dw_dma(kin ? pb[u] : g_zero_page, slot + 256 * lb[u]);
        }
        slot = slot == (kDwNS - 1) * kDwStage ? 0 : slot + kDwStage;
    };
    f4 acc[2][13];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 13; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    float csum[2] = {0.f, 0.f};
    issue(0);
    issue(1);
    const int roff = 4 * (4 * g + (c >> 2)) + (c & 3);   // element (row c, k = 4 g + r) of a block sits at float 64 r + roff
    int rslot = 0;                                       // ring slot of the step being multiplied
    for (int t = 0; t < nst; ++t) {
        asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");      // this wave's step-t loads have landed; everybody is done with step t - 1
        issue(t + 2);
        if (act) {
            const float* as = dw_lds + rslot + 512 * wave + roff;
            const float* bs = dw_lds + rslot + 2048 + roff;
            float av[2][2], bv[2][13];
            av[0][0] = as[0]; av[0][1] = as[256];
#pragma unroll
            for (int j = 0; j < 13; ++j) bv[0][j] = bs[256 * j];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r + 1 < 4) {
                    av[(r + 1) & 1][0] = as[64 * (r + 1)]; av[(r + 1) & 1][1] = as[256 + 64 * (r + 1)];
#pragma unroll
                    for (int j = 0; j < 13; ++j) bv[(r + 1) & 1][j] = bs[256 * j + 64 * (r + 1)];
                }
                __builtin_amdgcn_sched_barrier(0);
                csum[0] += av[r & 1][0]; csum[1] += av[r & 1][1];
#pragma unroll
                for (int j = 0; j < 13; ++j) {
                    acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r & 1][0], bv[r & 1][j], acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r & 1][1], bv[r & 1][j], acc[1][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        rslot = rslot == (kDwNS - 1) * kDwStage ? 0 : rslot + kDwStage;
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");          // the trailing zero-page loads have landed, the ring is free
    if (!act) return;
    if (P.colsum != nullptr && n0 == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float v = csum[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int m = m0 + 16 * i + c;
            if (g == 0 && m < P.M) atomicAdd(P.colsum + m, v);
        }
    }
    float* img = dw_lds + wave * (16 * 208);             // wave-private [16][ncols]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 13; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * j + c < ncols) img[(4 * g + r) * ncols + 16 * j + c] = acc[i][j][r];
        int row = 0, n = lane;
        for (int f = lane; f < 16 * ncols; f += 64, n += 64) {
            while (n >= ncols) { n -= ncols; ++row; }
            const int m = m0 + 16 * i + row;
            if (m < P.M) atomicAdd(P.C + (size_t)m * P.ldc + n0 + n, img[f]);
        }
    }
}

struct DwList {
    DwArgs args{};
    int nprob = 0;
    bool ok = true;
    void add(const float* A, int lda, int M, const float* B, int ldb, int N, float* C, int ldc, float* colsum) {
        if (!try_add(A, lda, M, B, ldb, N, C, ldc, colsum)) ok = false;
    }
    // M need not be a multiple of 4 when A's rows are padded to one (lda >= round_up(M, 4)): the float4 that holds the last columns then stays
    // inside the row, and outputs beyond M are never written
    bool try_add(const float* A, int lda, int M, const float* B, int ldb, int N, float* C, int ldc, float* colsum) {
        const bool aligned = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
                             (M % 4 == 0 || lda >= ((M + 3) & ~3)) && N % 4 == 0;
        const int nch = (N + 207) / 208, ncw = (((N + nch - 1) / nch) + 3) & ~3;
        const int mg = (M + 127) / 128;
        if (!aligned || nprob >= kDwMaxProblems || mg > 64 || nch > 64 || args.nitems + mg * nch > kDwMaxItems) return false;
        args.prob[nprob] = DwProblem{A, B, C, colsum, lda, ldb, ldc, M, N, ncw};
        for (int x = 0; x < mg; ++x)
            for (int y = 0; y < nch; ++y) args.item[args.nitems++] = (unsigned)nprob << 12 | (unsigned)x << 6 | (unsigned)y;
        ++nprob;
        return true;
    }
    int launch(hipStream_t s, int K) {
        if (nprob == 0 || K <= 0) return DYGNN_OK;
        // K-split: about two workgroups per CU over the whole grid, K-ranges of whole 16-wide steps
        int ksplit = 512 / args.nitems;
        if (ksplit < 1) ksplit = 1;
        int kchunk = (((K + ksplit - 1) / ksplit) + 15) & ~15;
        if (kchunk < 32) kchunk = 32;            // (a short K — the link predictor's 200 rows — is still worth splitting: each 16-wide step is a DMA round trip)
        ksplit = (K + kchunk - 1) / kchunk;
        args.K = K; args.kchunk = kchunk;
        DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dw_grouped), hipFuncAttributeMaxDynamicSharedMemorySize, kDwLdsBytes));
        hipLaunchKernelGGL(k_dw_grouped, dim3((unsigned)(args.nitems * ksplit)), dim3(256), kDwLdsBytes, s, args);
        DYGNN_LAUNCH_CHECK();
        return DYGNN_OK;
    }
};

int dw_grouped(hipStream_t s, int K, const DwPair* pairs, int npairs) {
    DwList dw;
    for (int i = 0; i < npairs; ++i) dw.add(pairs[i].A, pairs[i].lda, pairs[i].M, pairs[i].B, pairs[i].ldb, pairs[i].N, pairs[i].C, pairs[i].ldc, pairs[i].colsum);
    if (!dw.ok) { set_error("dw_grouped: operands are not 16-byte aligned / too many problems"); return DYGNN_E_UNSUPPORTED; }
    return dw.launch(s, K);
}

// out[n] += sum_m A[m][n]   (bias gradients): workgroup = 64 columns x a chunk of 256 rows, one atomic per column and workgroup
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ A, int lda, int64_t M, int N, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), r0 = threadIdx.x >> 6;
    const int64_t mlo = (int64_t)blockIdx.y * 256, mhi = mlo + 256 < M ? mlo + 256 : M;
    float s = 0.f;
    if (n < N)
        for (int64_t m = mlo + r0; m < mhi; m += 4) s += A[m * lda + n];
    red[r0][threadIdx.x & 63] = s;
    __syncthreads();
    if (r0 == 0 && n < N) atomicAdd(&out[n], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}
static int colsum(hipStream_t s, const float* A, int lda, int64_t M, int N, float* out, bool accumulate) {
    (void)accumulate;        // gradient buffers arrive zeroed (dygnn_dygformer_backward contract): every call accumulates
    hipLaunchKernelGGL(k_colsum, dim3((unsigned)ceil_div(N, 64), (unsigned)ceil_div(M, 256)), dim3(256), 0, s, A, lda, M, N, out);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

// ---- dropout: counter-based, identical in forward and backward (dropout.h) ----------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- LayerNorm rows (eps 1e-5, biased variance; models/DyGFormer.py:438-439) ---------------------------------------
__global__ __launch_bounds__(256) void k_ln_fwd(const float* __restrict__ X, const float* __restrict__ gamma, const float* __restrict__ beta, int64_t M,
                                                  int D, float* __restrict__ Y, float* __restrict__ mean_o, float* __restrict__ rstd_o) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* x = X + row * D;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) s += x[k];
    const float mean = wave_sum(s) / (float)D;
    float v = 0.f;
    for (int k = lane; k < D; k += 64) { const float d = x[k] - mean; v = fmaf(d, d, v); }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)D + 1e-5f);
    for (int k = lane; k < D; k += 64) Y[row * D + k] = (x[k] - mean) * rstd * gamma[k] + beta[k];
    if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
}
// dX[row] += LN'(dY) ; dgamma += sum dY*xhat ; dbeta += sum dY.  A lane owns columns lane, lane+64, ...: its partial sums over the
// 16 rows of its wave stay in registers, the four waves meet in LDS, one global atomic per column and workgroup.  D <= 256.
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ mean_i,
                                                  const float* __restrict__ rstd_i, const float* __restrict__ gamma, int64_t M, int D,
                                                  float* __restrict__ dX, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    extern __shared__ float part[];            // [4 waves][2][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int RB = 4;                      // rows of one wave in flight together: their loads and reductions overlap
    for (int rr = 0; rr < 16; rr += RB) {      // 64 rows per workgroup, 16 per wave
        float xh[RB][4], dy[RB][4], s1[RB], s2[RB], rstd[RB];
        int64_t row[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            row[u] = (int64_t)blockIdx.x * 64 + (rr + u) * 4 + wave;
            const bool on = row[u] < M;
            const float mean = on ? mean_i[row[u]] : 0.f;
            rstd[u] = on ? rstd_i[row[u]] : 0.f;
            s1[u] = 0.f; s2[u] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = lane + 64 * q;
                xh[u][q] = 0.f; dy[u][q] = 0.f;
                if (on && k < D) {
                    xh[u][q] = (X[row[u] * D + k] - mean) * rstd[u]; dy[u][q] = dY[row[u] * D + k];
                    const float gy = dy[u][q] * gamma[k];
                    s1[u] += gy; s2[u] = fmaf(gy, xh[u][q], s2[u]);
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < RB; ++u) { s1[u] += __shfl_xor(s1[u], o, 64); s2[u] += __shfl_xor(s2[u], o, 64); }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            if (row[u] >= M) continue;
            const float m1 = s1[u] / (float)D, m2 = s2[u] / (float)D;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = lane + 64 * q;
                if (k < D) {
                    dX[row[u] * D + k] += rstd[u] * (dy[u][q] * gamma[k] - m1 - xh[u][q] * m2);
                    pg[q] = fmaf(dy[u][q], xh[u][q], pg[q]); pb[q] += dy[u][q];
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = lane + 64 * q;
        if (k < D) { part[(wave * 2 + 0) * D + k] = pg[q]; part[(wave * 2 + 1) * D + k] = pb[q]; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * D; k += 256) {
        const int which = k / D, col = k % D;
        const float t = (part[(0 * 2 + which) * D + col] + part[(1 * 2 + which) * D + col]) + (part[(2 * 2 + which) * D + col] + part[(3 * 2 + which) * D + col]);
        atomicAdd(which ? &dbeta[col] : &dgamma[col], t);
    }
}

// ---- softmax over the keys of one (pair, head, query) row + dropout on the probabilities ---------------------------
__global__ __launch_bounds__(256) void k_softmax_fwd(const float* __restrict__ S, int64_t rows, int T, Drop dr, uint32_t site,
                                                       float* __restrict__ P, float* __restrict__ Pd) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* s = S + row * T;
    float mx = -INFINITY;
    for (int j = lane; j < T; j += 64) mx = fmaxf(mx, s[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int j = lane; j < T; j += 64) sum += expf(s[j] - mx);
    const float inv = 1.0f / wave_sum(sum);
    for (int j = lane; j < T; j += 64) {
        const float pv = expf(s[j] - mx) * inv;
        P[row * T + j] = pv;
        Pd[row * T + j] = pv * dr.mask(site, (uint64_t)row * T + j);
    }
}
// dS = P o (dP - sum_j dP_j P_j), dP = dPd * mask      (in place: dPd -> dS)
__global__ __launch_bounds__(256) void k_softmax_bwd(float* __restrict__ dPd, const float* __restrict__ P, int64_t rows, int T, Drop dr, uint32_t site) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float dot = 0.f;
    for (int j = lane; j < T; j += 64) dot = fmaf(dPd[row * T + j] * dr.mask(site, (uint64_t)row * T + j), P[row * T + j], dot);
    dot = wave_sum(dot);
    for (int j = lane; j < T; j += 64) {
        const float dp = dPd[row * T + j] * dr.mask(site, (uint64_t)row * T + j);
        dPd[row * T + j] = P[row * T + j] * (dp - dot);
    }
}

// ---- element-wise pieces ------------------------------------------------------------------------------------------------
__global__ void k_gelu_drop_fwd(const float* __restrict__ Hpre, int64_t n, Drop dr, uint32_t site, float* __restrict__ Hact) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = Hpre[i];
    Hact[i] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)) * dr.mask(site, (uint64_t)i);       // F.gelu, DyGFormer.py:458
}
__global__ void k_gelu_drop_bwd(float* __restrict__ dH, const float* __restrict__ Hpre, int64_t n, Drop dr, uint32_t site) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = Hpre[i];
    const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
    dH[i] = dH[i] * dr.mask(site, (uint64_t)i) * (cdf + v * pdf);
}
// Xout = Xin + dropout(Y)
__global__ void k_drop_add_fwd(const float* __restrict__ Xin, const float* __restrict__ Y, int64_t n, Drop dr, uint32_t site, float* __restrict__ Xout) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) Xout[i] = Xin[i] + Y[i] * dr.mask(site, (uint64_t)i);
}
// dY = dXout * mask
__global__ void k_drop_bwd(const float* __restrict__ dXout, int64_t n, Drop dr, uint32_t site, float* __restrict__ dY) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dY[i] = dXout[i] * dr.mask(site, (uint64_t)i);
}

// ---- embedding inputs: windows, co-occurrence counts, patch matrices -----------------------------------------------------
struct EmbedArgs {
    const int64_t* indptr; const int32_t* nbr; const int32_t* eid; const double* ts;
    const int64_t *src, *dst; const double* times; const int32_t* hist_len; const int64_t* end_pos;
    const float *node_feat, *edge_feat, *time_w, *time_b, *lut;
    int64_t B; int Ss, Sd, Ts, T, P, L, Fn, Fe, Ft, C;
    int32_t *ids, *c0, *c1; float* dts;          // meta [B][S]
    float *Pn, *Pe, *Pt, *Pc;                    // patch matrices [B*T][P*F]
    int64_t num_nodes;                           // rows of the CSR: query ids outside [0, num_nodes) are the padding node
};
__global__ __launch_bounds__(256) void k_embed_inputs(const EmbedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = a.Ss + a.Sd;
    int32_t* ids = reinterpret_cast<int32_t*>(smem);
    int32_t* eids = ids + S;
    float* dts = reinterpret_cast<float*>(eids + S);
    int32_t* c0 = reinterpret_cast<int32_t*>(dts + S);
    int32_t* c1 = c0 + S;
    const int64_t b = blockIdx.x;
    const double t = a.times[b];
    for (int p = threadIdx.x; p < S; p += 256) {           // pad_sequences, DyGFormer.py:228-245
        const bool is_dst = p >= a.Ss;
        const int j = is_dst ? p - a.Ss : p;
        const int64_t q = is_dst ? a.B + b : b;
        const int32_t len = a.hist_len[q];
        const int32_t m = len < a.L - 1 ? len : a.L - 1;
        int32_t id = 0, e = 0;
        float tn = 0.f;
        if (j == 0) { const int64_t qid = is_dst ? a.dst[b] : a.src[b]; id = qid < 0 || qid >= a.num_nodes ? 0 : (int32_t)qid; tn = (float)t; }
        else if (j <= m) { const int64_t pos = a.end_pos[q] - m + (j - 1); id = a.nbr[pos]; e = a.eid[pos]; tn = (float)a.ts[pos]; }
        ids[p] = id; eids[p] = e; dts[p] = (float)(t - (double)tn);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < S; p += 256) {           // count_nodes_appearances, DyGFormer.py:337-393
        const int32_t v = ids[p];
        int32_t cs = 0, cdn = 0;
        for (int q = 0; q < a.Ss; ++q) cs += (ids[q] == v);
        for (int q = a.Ss; q < S; ++q) cdn += (ids[q] == v);
        if (v == 0) { cs = 0; cdn = 0; }
        c0[p] = cs; c1[p] = cdn;
        if (blockIdx.y == 0) { a.ids[b * S + p] = v; a.dts[b * S + p] = dts[p]; a.c0[b * S + p] = cs; a.c1[b * S + p] = cdn; }
    }
    __syncthreads();
    // the patch matrices: the workgroups (b, 0 .. gridDim.y-1) of a pair share its tokens (each rebuilt the 128-position window
    // above, which is cheap); node / edge rows move as float4
    const int tok0 = (int)((int64_t)a.T * blockIdx.y / gridDim.y), ntok = (int)((int64_t)a.T * (blockIdx.y + 1) / gridDim.y) - tok0;
    const int Kn = a.P * a.Fn, Ke = a.P * a.Fe, Kt = a.P * a.Ft, Kc = a.P * a.C;
    const int Kn4 = Kn >> 2, KV = Kn4 + (Ke >> 2), KS = Kt + Kc;
    for (int idx = threadIdx.x; idx < ntok * KV; idx += 256) {
        const int tok = tok0 + idx / KV, k4 = idx % KV;
        const int p0 = tok < a.Ts ? tok * a.P : a.Ss + (tok - a.Ts) * a.P;
        const int64_t row = b * a.T + tok;
        if (k4 < Kn4) {
            const int k = 4 * k4, pp = p0 + k / a.Fn;
            *reinterpret_cast<f4*>(a.Pn + row * Kn + k) = *reinterpret_cast<const f4*>(a.node_feat + (size_t)ids[pp] * a.Fn + k % a.Fn);
        } else {
            const int k = 4 * (k4 - Kn4), pp = p0 + k / a.Fe;
            *reinterpret_cast<f4*>(a.Pe + row * Ke + k) = *reinterpret_cast<const f4*>(a.edge_feat + (size_t)eids[pp] * a.Fe + k % a.Fe);
        }
    }
    for (int idx = threadIdx.x; idx < ntok * KS; idx += 256) {
        const int tok = tok0 + idx / KS;
        int k = idx % KS;
        const int p0 = tok < a.Ts ? tok * a.P : a.Ss + (tok - a.Ts) * a.P;
        const int64_t row = b * a.T + tok;
        if (k < Kt) {
            const int pp = p0 + k / a.Ft, f = k % a.Ft;
            a.Pt[row * Kt + k] = ids[pp] == 0 ? 0.f : cosf(fmaf(dts[pp], a.time_w[f], a.time_b[f]));                 // modules.py:37, DyGFormer.py:266
            continue;
        }
        k -= Kt;
        { const int pp = p0 + k / a.C, f = k % a.C; a.Pc[row * Kc + k] = a.lut[(size_t)c0[pp] * a.C + f] + a.lut[(size_t)c1[pp] * a.C + f]; }
    }
}
// f(c) = W1 relu(W0 c + b0) + b1 for c = 0 .. rows-1 (DyGFormer.py:332-335); hidden activations kept for the backward pass
__global__ void k_lut_fwd(const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ w1, const float* __restrict__ b1,
                          int rows, int C, float* __restrict__ lut, float* __restrict__ hid) {
    const int c = blockIdx.x;
    extern __shared__ float h[];
    for (int j = threadIdx.x; j < C; j += blockDim.x) { h[j] = fmaxf(fmaf(w0[j], (float)c, b0[j]), 0.f); hid[c * C + j] = h[j]; }
    __syncthreads();
    for (int j = threadIdx.x; j < C; j += blockDim.x) {
        float acc = b1[j];
        for (int k = 0; k < C; ++k) acc = fmaf(w1[j * C + k], h[k], acc);
        lut[c * C + j] = acc;
    }
}
// sin of the time encoder's argument (up to ~3e6 rad): the range reduction of the forward's cosine (dygformer_fused3.hip cos_time) a quarter
// turn on: x / (2 pi) as a two-float product, the fraction folded to [0, 0.25], cos(2 pi u) by an even polynomial; libm's sinf takes its slow
// Payne-Hanek path for such arguments and is kept only beyond the product's accuracy
__device__ __forceinline__ float sin_time(float x) {
    if (!(fabsf(x) <= 3.0e7f)) return sinf(x);
    const float INV_HI = 0.15915493667125702f, INV_LO = 6.4206382432985265e-09f;
    const float p = x * INV_HI;
    const float e = fmaf(x, INV_HI, -p);
    const float q = fmaf(x, INV_LO, e);
    float t = ((p - rintf(p)) + q) - 0.25f;               // sin(2 pi t') = cos(2 pi (t' - 1/4))
    t -= rintf(t);
    float u = fabsf(t);
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
// time-encoder gradients from dPt [M][P*Ft]: pre = w dt + b, d cos = -sin(pre).  One workgroup per pair, thread = column k = pp * Ft + f of
// the pair's rows (coalesced), its sums over the T tokens in registers; the P columns of a feature meet in LDS, one atomic per feature.
__global__ __launch_bounds__(256) void k_time_bwd(const float* __restrict__ dPt, const int32_t* __restrict__ ids, const float* __restrict__ dts,
                                                    const float* __restrict__ tw, const float* __restrict__ tb, int64_t B, int Ss, int Sd, int Ts, int T,
                                                    int P, int Ft, float* __restrict__ dw, float* __restrict__ db) {
    extern __shared__ float part[];            // [2][Ft]
    for (int k = threadIdx.x; k < 2 * Ft; k += 256) part[k] = 0.f;
    __syncthreads();
    const int S = Ss + Sd, Kt = P * Ft;
    const int64_t b = blockIdx.x;
    for (int k = threadIdx.x; k < Kt; k += 256) {
        const int pp = k / Ft, f = k - pp * Ft;
        const float w = tw[f], bb = tb[f];
        float gw = 0.f, gb = 0.f;
        for (int tok = 0; tok < T; ++tok) {
            const int pos = (tok < Ts ? tok * P : Ss + (tok - Ts) * P) + pp;
            if (ids[b * S + pos] == 0) continue;
            const float dt = dts[b * S + pos];
            const float gsin = -sin_time(fmaf(dt, w, bb)) * dPt[(b * T + tok) * Kt + k];
            gw = fmaf(gsin, dt, gw); gb += gsin;
        }
        atomicAdd(&part[f], gw);
        atomicAdd(&part[Ft + f], gb);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < Ft; k += 256) { atomicAdd(&dw[k], part[k]); atomicAdd(&db[k], part[Ft + k]); }
}
// dlut[count][j] += dPc over both count channels (feature = lut[c0] + lut[c1], DyGFormer.py:409-411).
// One workgroup handles kPairsPerWg pairs.  Thread (grp, j) owns column j of its own LDS copy [grp][count < 16][C] and walks
// every 5th (token, position) of the pair, so the hot counts (0, 1, 2, ...) are summed without atomics; the five copies are
// then folded into the global table with one atomic per (count, column) and workgroup.  Counts >= 16 go straight to global.
constexpr int kCoocRows = 16, kCoocGroups = 5, kPairsPerWg = 1;      // one pair per workgroup: the walk is a chain of load latencies, so what pays is more workgroups
__global__ __launch_bounds__(256) void k_cooc_bwd(const float* __restrict__ dPc, const int32_t* __restrict__ c0, const int32_t* __restrict__ c1, int64_t B,
                                                    int Ss, int Sd, int Ts, int T, int P, int C, float* __restrict__ dlut) {
    extern __shared__ float acc[];             // [5][16][C]
    const int S = Ss + Sd, Kc = P * C;
    for (int i = threadIdx.x; i < kCoocGroups * kCoocRows * C; i += 256) acc[i] = 0.f;
    __syncthreads();
    const int grp = threadIdx.x / C, j = threadIdx.x % C;
    if (grp < kCoocGroups) {
        float* mine = acc + (size_t)grp * kCoocRows * C;
        for (int64_t b = (int64_t)blockIdx.x * kPairsPerWg; b < B && b < (int64_t)(blockIdx.x + 1) * kPairsPerWg; ++b) {
            for (int q = grp; q < T * P; q += kCoocGroups) {          // q = token * P + position inside the patch
                const int tok = q / P, pin = q - tok * P;
                const int pp = (tok < Ts ? tok * P : Ss + (tok - Ts) * P) + pin;
                const float gv = dPc[(b * T + tok) * Kc + pin * C + j];
                const int32_t a0 = c0[b * S + pp], a1 = c1[b * S + pp];
                if (a0 < kCoocRows) mine[a0 * C + j] += gv; else atomicAdd(&dlut[(size_t)a0 * C + j], gv);
                if (a1 < kCoocRows) mine[a1 * C + j] += gv; else atomicAdd(&dlut[(size_t)a1 * C + j], gv);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kCoocRows * C; i += 256) {
        float t = 0.f;
#pragma unroll
        for (int gq = 0; gq < kCoocGroups; ++gq) t += acc[(size_t)gq * kCoocRows * C + i];
        if (t != 0.f) atomicAdd(&dlut[i], t);
    }
}
// gradients of the co-occurrence MLP from dlut: dh[c][k] = (hid[c][k] > 0) sum_j w1[j][k] dlut[c][j]  (one workgroup per count)
__global__ void k_lut_bwd_hidden(const float* __restrict__ dlut, const float* __restrict__ hid, const float* __restrict__ w1, int C, float* __restrict__ dh) {
    const int c = blockIdx.x;
    for (int k = threadIdx.x; k < C; k += blockDim.x) {
        float v = 0.f;
        if (hid[c * C + k] > 0.f)
            for (int j = 0; j < C; ++j) v = fmaf(w1[j * C + k], dlut[c * C + j], v);
        dh[c * C + k] = v;
    }
}
// dw1[j][k] = sum_c dlut[c][j] hid[c][k] ; db1[j] = sum_c dlut[c][j] ; dw0[k] = sum_c dh[c][k] * c ; db0[k] = sum_c dh[c][k]
// (three [rows][C] tables, rows <= 2 Smax + 1: staged in LDS once per workgroup, the sums then run from LDS instead of as chains of global loads)
constexpr int kLutChunk = 128;                 // table rows staged per pass
__global__ __launch_bounds__(256) void k_lut_bwd(const float* __restrict__ dlut, const float* __restrict__ hid, const float* __restrict__ dh, int rows, int C,
                                                   float* __restrict__ dw0, float* __restrict__ db0, float* __restrict__ dw1, float* __restrict__ db1) {
    extern __shared__ float lsm[];             // dlut | hid | dh, each [kLutChunk][C]
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int nmax = kLutChunk * C;
    const float *sl = lsm, *sh = lsm + nmax, *sd = lsm + 2 * nmax;
    float acc = 0.f, acc2 = 0.f;
    for (int r0 = 0; r0 < rows; r0 += kLutChunk) {
        const int nr = rows - r0 < kLutChunk ? rows - r0 : kLutChunk, n = nr * C;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) { lsm[i] = dlut[r0 * C + i]; lsm[nmax + i] = hid[r0 * C + i]; lsm[2 * nmax + i] = dh[r0 * C + i]; }
        __syncthreads();
        if (idx < C * C) {
            const int j = idx / C, k = idx % C;
            for (int c = 0; c < nr; ++c) acc = fmaf(sl[c * C + j], sh[c * C + k], acc);
        } else if (idx < C * C + C) {
            const int j = idx - C * C;
            for (int c = 0; c < nr; ++c) acc += sl[c * C + j];
        } else if (idx < C * C + 2 * C) {
            const int k = idx - C * C - C;
            for (int c = 0; c < nr; ++c) { const float v = sd[c * C + k]; acc = fmaf(v, (float)(r0 + c), acc); acc2 += v; }
        }
    }
    if (idx < C * C) dw1[idx] = acc;
    else if (idx < C * C + C) db1[idx - C * C] = acc;
    else if (idx < C * C + 2 * C) { dw0[idx - C * C - C] = acc; db0[idx - C * C - C] = acc2; }
}

// pooled[side][b][:] = mean over the side's tokens (DyGFormer.py:181-187)
__global__ __launch_bounds__(256) void k_pool_fwd(const float* __restrict__ X, int64_t B, int Ts, int T, int D, float* __restrict__ pooled) {
    const int64_t b = blockIdx.x;
    for (int i = threadIdx.x; i < 2 * D; i += 256) {
        const int side = i / D, n = i % D;
        const int t0 = side ? Ts : 0, t1 = side ? T : Ts;
        float s = 0.f;
        for (int t = t0; t < t1; ++t) s += X[(b * T + t) * D + n];
        pooled[((int64_t)side * B + b) * D + n] = s / (float)(t1 - t0);
    }
}
__global__ void k_pool_bwd(const float* __restrict__ dpooled, int64_t B, int Ts, int T, int D, float* __restrict__ dX) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T * D) return;
    const int n = (int)(i % D);
    const int64_t row = i / D, b = row / T;
    const int t = (int)(row % T);
    const int side = t >= Ts;
    dX[i] = dpooled[((int64_t)side * B + b) * D + n] / (float)(side ? T - Ts : Ts);
}

// dX0 [M][4 C] -> four channel blocks [4][M][Cp], Cp = round_up(C, 4), zeros behind C: every block is then a 16-byte-aligned operand of the
// grouped weight-gradient launch (projection weights) and of the LDS-DMA GEMM (time / co-occurrence encoder inputs)
__global__ void k_split_channels(const float* __restrict__ dX, int64_t M, int C, int Cp, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * M * Cp) return;
    const int j = (int)(i % Cp);
    const int64_t row = (i / Cp) % M;
    const int ch = (int)(i / ((int64_t)Cp * M));
    out[i] = j < C ? dX[row * (4 * C) + ch * C + j] : 0.f;
}

// ---- workspace -----------------------------------------------------------------------------------------------------------
struct Plan {
    WorkspaceLayout wl;       // prefix compatible with window_lengths_device (dims, hist_len, end_pos)
    size_t ids, dts, c0, c1, lut, hid, Pn, Pe, Pt, Pc, X[DYGNN_MAX_LAYERS + 1];
    struct L { size_t xn0, m0, r0, qkv, S, P, Pd, oa, ao, x1, xn1, m1, r1, hpre, hact, f2, dF2, dH, dAo, dQKV; } layer[DYGNN_MAX_LAYERS];
    size_t pooled, out, dX, dA, dB, dXc, dpool, dPt, dPc, dlut, total;
};
static Plan make_plan(const Dims& d, int64_t B) {
    Plan p{};
    p.wl = make_workspace_layout(d, B);
    size_t o = (p.wl.end_pos + (size_t)2 * B * sizeof(int64_t) + 255) & ~size_t(255);
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~size_t(255); return r; };
    const size_t S = 2 * (size_t)d.Smax, M = (size_t)B * d.Tmax, F = sizeof(float);
    p.ids = take(B * S * 4); p.dts = take(B * S * 4); p.c0 = take(B * S * 4); p.c1 = take(B * S * 4);
    p.lut = take((size_t)(S + 1) * d.C * F); p.hid = take((size_t)(S + 1) * d.C * F);
    p.Pn = take(M * d.P * d.Fn * F); p.Pe = take(M * d.P * d.Fe * F); p.Pt = take(M * d.P * d.Ft * F); p.Pc = take(M * d.P * d.C * F);
    for (int l = 0; l <= d.NL; ++l) p.X[l] = take(M * d.D * F);
    for (int l = 0; l < d.NL; ++l) {
        auto& L = p.layer[l];
        L.xn0 = take(M * d.D * F); L.m0 = take(M * F); L.r0 = take(M * F); L.qkv = take(M * 3 * d.D * F);
        L.S = take((size_t)B * d.H * d.Tmax * d.Tmax * F); L.P = take((size_t)B * d.H * d.Tmax * d.Tmax * F); L.Pd = take((size_t)B * d.H * d.Tmax * d.Tmax * F);
        L.oa = take(M * d.D * F); L.ao = take(M * d.D * F); L.x1 = take(M * d.D * F); L.xn1 = take(M * d.D * F); L.m1 = take(M * F); L.r1 = take(M * F);
        L.hpre = take(M * 4 * d.D * F); L.hact = take(M * 4 * d.D * F); L.f2 = take(M * d.D * F);
    }
    p.pooled = take((size_t)2 * B * d.D * F); p.out = take((size_t)2 * B * d.Fn * F);
    p.dX = take(M * d.D * F); p.dA = take(M * d.D * F); p.dB = take(M * d.D * F); p.dXc = take(4 * M * ((d.C + 3) & ~3) * F);
    // operands of the weight gradients stay alive until the ONE grouped launch at the end of the backward pass (k_dw_grouped): per layer
    for (int l = 0; l < d.NL; ++l) {
        auto& L = p.layer[l];
        L.dF2 = take(M * d.D * F); L.dH = take(M * 4 * d.D * F); L.dAo = take(M * d.D * F); L.dQKV = take(M * 3 * d.D * F);
    }
    p.dpool = take((size_t)2 * B * d.D * F); p.dPt = take(M * d.P * d.Ft * F); p.dPc = take(M * d.P * d.C * F); p.dlut = take((size_t)(S + 1) * d.C * F);
    p.total = o;
    return p;
}

static int supported(const Dims& d) {
    if (d.C <= 0 || d.H <= 0 || d.D % d.H != 0) { set_error("train: bad dims"); return DYGNN_E_INVALID; }
    if (d.D > 256) { set_error("train: model dim > 256 not supported"); return DYGNN_E_UNSUPPORTED; }
    if (d.C * kCoocGroups > 256) { set_error("train: channel_embedding_dim > 51 not supported"); return DYGNN_E_UNSUPPORTED; }
    if ((size_t)5 * 2 * d.Smax * 4 > 60 * 1024) { set_error("train: max_input_sequence_length too large for the embedding kernel"); return DYGNN_E_UNSUPPORTED; }
    return DYGNN_OK;
}

#define EW(kernel, n, ...)                                                                                          \
    do {                                                                                                            \
        hipLaunchKernelGGL(kernel, dim3((unsigned)ceil_div((int64_t)(n), 256)), dim3(256), 0, s, __VA_ARGS__);      \
        DYGNN_LAUNCH_CHECK();                                                                                       \
    } while (0)

}  // namespace train
}  // namespace dygnn

using namespace dygnn;
using namespace dygnn::train;

extern "C" size_t dygnn_dygformer_train_workspace_bytes(const dygnn_dygformer_config* cfg, int64_t batch) {
    if (check_config(cfg) != DYGNN_OK || batch < 0) return 0;
    const Dims d = make_dims(*cfg);
    if (supported(d) != DYGNN_OK) return 0;
    return make_plan(d, batch).total;
}

extern "C" int dygnn_dygformer_train_forward(const dygnn_dygformer_config* cfg, const dygnn_dygformer_weights* w, const dygnn_csr* csr,
                                             const float* node_feat, const float* edge_feat, const int64_t* src, const int64_t* dst,
                                             const double* times, int64_t batch, float dropout_p, uint64_t seed, float* out_src, float* out_dst,
                                             void* workspace, size_t workspace_bytes, int32_t* seq_lens_host, const void* packed, dygnn_stream_t stream) {
    if (int rc = check_config(cfg)) return rc;
    const Dims d = make_dims(*cfg);
    if (int rc = supported(d)) return rc;
    DYGNN_REQUIRE(w && csr && csr->indptr && node_feat && edge_feat, "train_forward: null pointer");
    DYGNN_REQUIRE(batch > 0 && src && dst && times && out_src && out_dst && workspace && seq_lens_host, "train_forward: bad arguments");
    DYGNN_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "train_forward: dropout must be in [0, 1)");
    const Plan p = make_plan(d, batch);
    if (workspace_bytes < p.total) { set_error("train_forward: workspace too small (%zu < %zu bytes)", workspace_bytes, p.total); return DYGNN_E_WORKSPACE; }
    hipStream_t s = as_stream(stream);
    char* ws = static_cast<char*>(workspace);
    auto F32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    auto I32 = [&](size_t off) { return reinterpret_cast<int32_t*>(ws + off); };
    const int64_t B = batch;
    // window lengths -> this call's padded lengths (one host sync: they size every product below)
    if (int rc = window_lengths_device(d, csr, src, dst, times, B, B, ws, p.wl, s)) return rc;
    // the padded lengths size every product below.  seq_lens_host = {0, 0}: read them back here (one host synchronisation of this
    // stream); non-zero: the caller already knows them (e.g. from dygnn_window_lengths on a side stream) and the call stays asynchronous
    if (seq_lens_host[0] <= 0 || seq_lens_host[1] <= 0) {
        CallDims cd;
        DYGNN_HIP(hipMemcpyAsync(&cd, ws + p.wl.dims, sizeof(CallDims), hipMemcpyDeviceToHost, s));
        DYGNN_HIP(hipStreamSynchronize(s));
        seq_lens_host[0] = cd.S_s; seq_lens_host[1] = cd.S_d;
    }
    const int Ss = seq_lens_host[0], Sd = seq_lens_host[1];
    DYGNN_REQUIRE(Ss % d.P == 0 && Sd % d.P == 0 && Ss <= d.Smax && Sd <= d.Smax, "train_forward: bad sequence lengths");
    const int Ts = Ss / d.P, T = (Ss + Sd) / d.P, S = Ss + Sd;
    const int64_t M = B * T;
    const int D = d.D, C = d.C, H = d.H, hd = d.hd;
    const Drop dr = make_drop(dropout_p, seed);
    // co-occurrence LUT of this step's weights
    hipLaunchKernelGGL(k_lut_fwd, dim3((unsigned)(S + 1)), dim3(64), C * sizeof(float), s, w->cooc_w0, w->cooc_b0, w->cooc_w1, w->cooc_b1, S + 1, C,
                       F32(p.lut), F32(p.hid));
    DYGNN_LAUNCH_CHECK();
    EmbedArgs ea{csr->indptr, csr->nbr, csr->eid, csr->ts, src, dst, times, reinterpret_cast<const int32_t*>(ws + p.wl.hist_len),
                 reinterpret_cast<const int64_t*>(ws + p.wl.end_pos), node_feat, edge_feat, w->time_w, w->time_b, F32(p.lut), B, Ss, Sd, Ts, T, d.P, d.L,
                 d.Fn, d.Fe, d.Ft, C, I32(p.ids), I32(p.c0), I32(p.c1), F32(p.dts), F32(p.Pn), F32(p.Pe), F32(p.Pt), F32(p.Pc), csr->num_nodes};
    hipLaunchKernelGGL(k_embed_inputs, dim3((unsigned)B, (unsigned)(T >= 8 ? 8 : 1)), dim3(256), (size_t)5 * S * 4, s, ea);
    DYGNN_LAUNCH_CHECK();
    // The fused forward (dygformer_fused3.hip, TR = true): one kernel from the windows to the embeddings that also writes every activation
    // the backward pass reads into this Plan's buffers.  `packed` = the kernel-ready copy of the CURRENT weights (dygnn_dygformer_pack /
    // dygnn_dygformer_repack); NULL, or a shape the fused kernel does not take: the product-by-product path below.
    if (packed != nullptr && fused3_supported(d) && (uint64_t)M * 4 * D < (1ull << 32) && (uint64_t)B * H * T * T < (1ull << 32) && !getenv("DYGNN_TRAIN_UNFUSED")) {
        TrainOut tr{};
        for (int l = 0; l <= d.NL; ++l) tr.X[l] = F32(p.X[l]);
        for (int l = 0; l < d.NL; ++l) {
            const auto& L = p.layer[l];
            tr.layer[l] = TrainOut::L{F32(L.xn0), F32(L.m0), F32(L.r0), F32(L.qkv), F32(L.P), F32(L.Pd), F32(L.oa), F32(L.x1), F32(L.xn1), F32(L.m1), F32(L.r1),
                                      F32(L.hpre), F32(L.hact)};
        }
        tr.pooled = F32(p.pooled);
        tr.dr = dr;
        return forward_fused3_train(d, make_packed_layout(d), w, static_cast<const float*>(packed), csr, node_feat, edge_feat, src, dst, times, B, F32(p.lut),
                                    out_src, out_dst, ws, p.wl, tr, s);
    }
    // projections (DyGFormer.py:148-157): X0[:, 50ch : 50ch+50] = P_ch . W_ch^T + b_ch
    const float* PW[4] = {w->proj_node_w, w->proj_edge_w, w->proj_time_w, w->proj_cooc_w};
    const float* PB[4] = {w->proj_node_b, w->proj_edge_b, w->proj_time_b, w->proj_cooc_b};
    const size_t PM[4] = {p.Pn, p.Pe, p.Pt, p.Pc};
    const int PK[4] = {d.P * d.Fn, d.P * d.Fe, d.P * d.Ft, d.P * C};
    for (int ch = 0; ch < 4; ++ch)
        if (int rc = mm(s, F32(PM[ch]), PK[ch], false, PW[ch], PK[ch], true, F32(p.X[0]) + ch * C, D, (int)M, C, PK[ch], PB[ch])) return rc;
    const float scale = (float)sqrt(1.0 / (double)hd);
    for (int l = 0; l < d.NL; ++l) {
        const dygnn_encoder_layer_weights& Lw = w->layers[l];
        const auto& L = p.layer[l];
        const float* Xin = F32(p.X[l]);
        hipLaunchKernelGGL(k_ln_fwd, dim3((unsigned)ceil_div(M, 4)), dim3(256), 0, s, Xin, Lw.norm0_weight, Lw.norm0_bias, M, D, F32(L.xn0), F32(L.m0), F32(L.r0));
        DYGNN_LAUNCH_CHECK();
        if (int rc = mm(s, F32(L.xn0), D, false, Lw.in_proj_weight, D, true, F32(L.qkv), 3 * D, (int)M, 3 * D, D, Lw.in_proj_bias)) return rc;
        // S_bh = scale * Q_bh K_bh^T ; batch z = b*H + h
        if (int rc = mm(s, F32(L.qkv), 3 * D, false, F32(L.qkv) + D, 3 * D, true, F32(L.S), T, T, T, hd, nullptr, scale, 0.f, (int)(B * H), H,
                        (int64_t)T * 3 * D, hd, (int64_t)T * 3 * D, hd, (int64_t)H * T * T, (int64_t)T * T)) return rc;
        hipLaunchKernelGGL(k_softmax_fwd, dim3((unsigned)ceil_div(B * H * T, 4)), dim3(256), 0, s, F32(L.S), B * H * T, T, dr, (uint32_t)(4 * l + 0), F32(L.P), F32(L.Pd));
        DYGNN_LAUNCH_CHECK();
        // Oa_bh = Pd_bh V_bh
        if (int rc = mm(s, F32(L.Pd), T, false, F32(L.qkv) + 2 * D, 3 * D, false, F32(L.oa), D, T, hd, T, nullptr, 1.f, 0.f, (int)(B * H), H,
                        (int64_t)H * T * T, (int64_t)T * T, (int64_t)T * 3 * D, hd, (int64_t)T * D, hd)) return rc;
        if (int rc = mm(s, F32(L.oa), D, false, Lw.out_proj_weight, D, true, F32(L.ao), D, (int)M, D, D, Lw.out_proj_bias)) return rc;
        EW(k_drop_add_fwd, M * D, Xin, F32(L.ao), M * D, dr, (uint32_t)(4 * l + 1), F32(L.x1));                                         // DyGFormer.py:456
        hipLaunchKernelGGL(k_ln_fwd, dim3((unsigned)ceil_div(M, 4)), dim3(256), 0, s, F32(L.x1), Lw.norm1_weight, Lw.norm1_bias, M, D, F32(L.xn1), F32(L.m1), F32(L.r1));
        DYGNN_LAUNCH_CHECK();
        if (int rc = mm(s, F32(L.xn1), D, false, Lw.ffn0_weight, D, true, F32(L.hpre), 4 * D, (int)M, 4 * D, D, Lw.ffn0_bias)) return rc;
        EW(k_gelu_drop_fwd, M * 4 * D, F32(L.hpre), M * 4 * D, dr, (uint32_t)(4 * l + 2), F32(L.hact));                                 // :458
        if (int rc = mm(s, F32(L.hact), 4 * D, false, Lw.ffn1_weight, 4 * D, true, F32(L.f2), D, (int)M, D, 4 * D, Lw.ffn1_bias)) return rc;
        EW(k_drop_add_fwd, M * D, F32(L.x1), F32(L.f2), M * D, dr, (uint32_t)(4 * l + 3), F32(p.X[l + 1]));                             // :460
    }
    hipLaunchKernelGGL(k_pool_fwd, dim3((unsigned)B), dim3(256), 0, s, F32(p.X[d.NL]), B, Ts, T, D, F32(p.pooled));
    DYGNN_LAUNCH_CHECK();
    if (int rc = mm(s, F32(p.pooled), D, false, w->output_w, D, true, out_src, d.Fn, (int)B, d.Fn, D, w->output_b)) return rc;
    if (int rc = mm(s, F32(p.pooled) + B * D, D, false, w->output_w, D, true, out_dst, d.Fn, (int)B, d.Fn, D, w->output_b)) return rc;
    return DYGNN_OK;
}

extern "C" int dygnn_dygformer_backward(const dygnn_dygformer_config* cfg, const dygnn_dygformer_weights* w, const dygnn_dygformer_weights* grads,
                                        const float* grad_out_src, const float* grad_out_dst, int64_t batch, float dropout_p, uint64_t seed,
                                        const int32_t* seq_lens_host, void* workspace, size_t workspace_bytes, const void* packed, dygnn_stream_t stream) {
    if (int rc = check_config(cfg)) return rc;
    const Dims d = make_dims(*cfg);
    if (int rc = supported(d)) return rc;
    DYGNN_REQUIRE(w && grads && grad_out_src && grad_out_dst && workspace && seq_lens_host && batch > 0, "backward: bad arguments");
    const Plan p = make_plan(d, batch);
    if (workspace_bytes < p.total) { set_error("backward: workspace too small (%zu < %zu bytes)", workspace_bytes, p.total); return DYGNN_E_WORKSPACE; }
    hipStream_t s = as_stream(stream);
    char* ws = static_cast<char*>(workspace);
    auto F32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    auto I32 = [&](size_t off) { return reinterpret_cast<int32_t*>(ws + off); };
    auto G = [](const float* q) { return const_cast<float*>(q); };        // the grads struct reuses the weights layout; its buffers are written
    const int64_t B = batch;
    const int Ss = seq_lens_host[0], Sd = seq_lens_host[1], S = Ss + Sd;
    DYGNN_REQUIRE(Ss > 0 && Sd > 0 && Ss % d.P == 0 && Sd % d.P == 0 && S <= 2 * d.Smax, "backward: bad sequence lengths");
    const int Ts = Ss / d.P, T = S / d.P;
    const int64_t M = B * T;
    const int D = d.D, C = d.C, H = d.H, hd = d.hd, Fn = d.Fn;
    const Drop dr = make_drop(dropout_p, seed);
    const float scale = (float)sqrt(1.0 / (double)hd);
    // output layer: out = pooled W^T + b over both sides
    if (int rc = mm(s, grad_out_src, Fn, true, F32(p.pooled), D, false, G(grads->output_w), D, Fn, D, (int)B)) return rc;
    if (int rc = mm(s, grad_out_dst, Fn, true, F32(p.pooled) + B * D, D, false, G(grads->output_w), D, Fn, D, (int)B, nullptr, 1.f, 1.f)) return rc;
    if (int rc = colsum(s, grad_out_src, Fn, B, Fn, G(grads->output_b))) return rc;
    if (int rc = colsum(s, grad_out_dst, Fn, B, Fn, G(grads->output_b), true)) return rc;
    if (int rc = mm(s, grad_out_src, Fn, false, w->output_w, D, false, F32(p.dpool), D, (int)B, D, Fn)) return rc;
    if (int rc = mm(s, grad_out_dst, Fn, false, w->output_w, D, false, F32(p.dpool) + B * D, D, (int)B, D, Fn)) return rc;
    float* dX = F32(p.dX);
    EW(k_pool_bwd, M * D, F32(p.dpool), B, Ts, T, D, dX);
    DwList dw;
    // `packed` = the kernel-ready copy of the weights the forward ran with (it holds the backward fragment streams too): the FFN blocks run fused
    const bool fused = packed != nullptr && fused3_supported(d) && (uint64_t)M * 4 * D < (1ull << 32) && !getenv("DYGNN_TRAIN_UNFUSED");
    for (int l = d.NL - 1; l >= 0; --l) {
        const dygnn_encoder_layer_weights& Lw = w->layers[l];
        const dygnn_encoder_layer_weights& Lg = grads->layers[l];
        const auto& L = p.layer[l];
        float* dA = F32(p.dA);          // [M][D] scratch
        float* dBf = F32(p.dB);         // [M][D] scratch
        float* dF2 = F32(L.dF2);        // [M][D]   operands of this layer's weight gradients: alive until the grouped launch below
        float* dH = F32(L.dH);          // [M][4D]
        float* dAo = F32(L.dAo);        // [M][D]
        float* dQKV = F32(L.dQKV);      // [M][3D]
        // X_{l+1} = X1 + drop(F2), F2 = Hact W2^T + b2
        dw.add(dF2, D, D, F32(L.hact), 4 * D, 4 * D, G(Lg.ffn1_weight), 4 * D, G(Lg.ffn1_bias));                               // dW2 [D][4D], db2
        dw.add(dH, 4 * D, 4 * D, F32(L.xn1), D, D, G(Lg.ffn0_weight), D, G(Lg.ffn0_bias));                                     // dW1 [4D][D], db1
        if (fused) {      // the whole block in one kernel (dygformer_fused3.hip: k_ffn_bwd): dF2, dHpre written for the grouped launch, dX <- dX1 in place
            if (int rc = ffn_backward_fused3(d, make_packed_layout(d), static_cast<const float*>(packed), l, M, dX, F32(L.hpre), F32(L.x1), F32(L.m1), F32(L.r1), dF2, dH,
                                             G(Lg.norm1_weight), G(Lg.norm1_bias), dr, s)) return rc;
        } else {
            EW(k_drop_bwd, M * D, dX, M * D, dr, (uint32_t)(4 * l + 3), dF2);                                                  // dF2
            if (int rc = mm(s, dF2, D, false, Lw.ffn1_weight, 4 * D, false, dH, 4 * D, (int)M, 4 * D, D)) return rc;           // dHact
            EW(k_gelu_drop_bwd, M * 4 * D, dH, F32(L.hpre), M * 4 * D, dr, (uint32_t)(4 * l + 2));                             // dHpre
            if (int rc = mm(s, dH, 4 * D, false, Lw.ffn0_weight, D, false, dBf, D, (int)M, D, 4 * D)) return rc;                // dxn1
            hipLaunchKernelGGL(k_ln_bwd, dim3((unsigned)ceil_div(M, 64)), dim3(256), 8 * D * sizeof(float), s, dBf, F32(L.x1), F32(L.m1), F32(L.r1), Lw.norm1_weight, M, D,
                               dX, G(Lg.norm1_weight), G(Lg.norm1_bias));                                                      // dX is now dX1
            DYGNN_LAUNCH_CHECK();
        }
        // X1 = Xin + drop(Ao), Ao = Oa Wo^T + bo ; Oa_bh = Pd_bh V_bh ; S_bh = scale Q_bh K_bh^T ; [Q | K | V] = LN0(Xin) Win^T + bin
        dw.add(dAo, D, D, F32(L.oa), D, D, G(Lg.out_proj_weight), D, G(Lg.out_proj_bias));                                     // dWo [D][D], dbo
        dw.add(dQKV, 3 * D, 3 * D, F32(L.xn0), D, D, G(Lg.in_proj_weight), D, G(Lg.in_proj_bias));                              // dWin [3D][D], dbin
        if (fused && T <= 128 && H == 2) {      // the whole block in one kernel (dygformer_fused3.hip: k_attn_bwd): dAo, dQKV written for the grouped launch, dX <- dX_l in place
            if (int rc = attn_backward_fused3(d, make_packed_layout(d), static_cast<const float*>(packed), l, B, T, dX, F32(p.X[l]), F32(L.m0), F32(L.r0), F32(L.qkv), F32(L.P),
                                              F32(L.Pd), dAo, dQKV, G(Lg.norm0_weight), G(Lg.norm0_bias), dr, s)) return rc;
            continue;
        }
        EW(k_drop_bwd, M * D, dX, M * D, dr, (uint32_t)(4 * l + 1), dAo);                                                      // dAo
        if (int rc = mm(s, dAo, D, false, Lw.out_proj_weight, D, false, dBf, D, (int)M, D, D)) return rc;                       // dOa
        // dV_bh = Pd^T dOa_bh
        if (int rc = mm(s, F32(L.Pd), T, true, dBf, D, false, dQKV + 2 * D, 3 * D, T, hd, T, nullptr, 1.f, 0.f, (int)(B * H), H, (int64_t)H * T * T, (int64_t)T * T,
                        (int64_t)T * D, hd, (int64_t)T * 3 * D, hd)) return rc;
        // dPd_bh = dOa_bh V_bh^T  (into the S buffer)
        if (int rc = mm(s, dBf, D, false, F32(L.qkv) + 2 * D, 3 * D, true, F32(L.S), T, T, T, hd, nullptr, 1.f, 0.f, (int)(B * H), H, (int64_t)T * D, hd,
                        (int64_t)T * 3 * D, hd, (int64_t)H * T * T, (int64_t)T * T)) return rc;
        hipLaunchKernelGGL(k_softmax_bwd, dim3((unsigned)ceil_div(B * H * T, 4)), dim3(256), 0, s, F32(L.S), F32(L.P), B * H * T, T, dr, (uint32_t)(4 * l + 0));
        DYGNN_LAUNCH_CHECK();
        // dQ_bh = scale dS K_bh ; dK_bh = scale dS^T Q_bh
        if (int rc = mm(s, F32(L.S), T, false, F32(L.qkv) + D, 3 * D, false, dQKV, 3 * D, T, hd, T, nullptr, scale, 0.f, (int)(B * H), H, (int64_t)H * T * T,
                        (int64_t)T * T, (int64_t)T * 3 * D, hd, (int64_t)T * 3 * D, hd)) return rc;
        if (int rc = mm(s, F32(L.S), T, true, F32(L.qkv), 3 * D, false, dQKV + D, 3 * D, T, hd, T, nullptr, scale, 0.f, (int)(B * H), H, (int64_t)H * T * T,
                        (int64_t)T * T, (int64_t)T * 3 * D, hd, (int64_t)T * 3 * D, hd)) return rc;
        if (int rc = mm(s, dQKV, 3 * D, false, Lw.in_proj_weight, D, false, dA, D, (int)M, D, 3 * D)) return rc;                // dxn0
        hipLaunchKernelGGL(k_ln_bwd, dim3((unsigned)ceil_div(M, 64)), dim3(256), 8 * D * sizeof(float), s, dA, F32(p.X[l]), F32(L.m0), F32(L.r0), Lw.norm0_weight, M, D,
                           dX, G(Lg.norm0_weight), G(Lg.norm0_bias));                                                          // dX is now dX_l
        DYGNN_LAUNCH_CHECK();
    }
    // projections: X0[:, ch] = P_ch W_ch^T + b_ch.  dX0 is split into channel blocks (aligned operands); the four projection weight gradients
    // join the grouped launch where their shapes allow, the rest takes the general path
    const float* PW[4] = {w->proj_node_w, w->proj_edge_w, w->proj_time_w, w->proj_cooc_w};
    float* GW[4] = {G(grads->proj_node_w), G(grads->proj_edge_w), G(grads->proj_time_w), G(grads->proj_cooc_w)};
    float* GB[4] = {G(grads->proj_node_b), G(grads->proj_edge_b), G(grads->proj_time_b), G(grads->proj_cooc_b)};
    const size_t PM[4] = {p.Pn, p.Pe, p.Pt, p.Pc};
    const int PK[4] = {d.P * d.Fn, d.P * d.Fe, d.P * d.Ft, d.P * C};
    const int Cp = (C + 3) & ~3;
    float* dXc = F32(p.dXc);
    EW(k_split_channels, 4 * M * Cp, dX, M, C, Cp, dXc);
    bool grouped[4];
    for (int ch = 0; ch < 4; ++ch) grouped[ch] = dw.try_add(dXc + (size_t)ch * M * Cp, Cp, C, F32(PM[ch]), PK[ch], PK[ch], GW[ch], PK[ch], GB[ch]);
    // every weight gradient of the encoder layers and projections (and its bias gradient) in one grouped split-K launch
    if (!dw.ok) { set_error("backward: weight-gradient operands are not 16-byte aligned / too many problems"); return DYGNN_E_UNSUPPORTED; }
    if (int rc = dw.launch(s, (int)M)) return rc;
    for (int ch = 0; ch < 4; ++ch) {
        if (grouped[ch]) continue;
        if (int rc = mm(s, dX + ch * C, D, true, F32(PM[ch]), PK[ch], false, GW[ch], PK[ch], C, PK[ch], (int)M, nullptr, 1.f, 0.f, 1, 1, 0, 0, 0, 0, 0, 0, false, true, GB[ch])) return rc;     // dW_ch [C][K]
    }
    // time encoder
    if (int rc = mm(s, dXc + (size_t)2 * M * Cp, Cp, false, PW[2], PK[2], false, F32(p.dPt), PK[2], (int)M, PK[2], C, nullptr, 1.f, 0.f, 1, 1, 0, 0, 0, 0, 0, 0, false, false,
                    nullptr, nullptr, true)) return rc;
    hipLaunchKernelGGL(k_time_bwd, dim3((unsigned)B), dim3(256), 2 * d.Ft * sizeof(float), s, F32(p.dPt), I32(p.ids), F32(p.dts), w->time_w, w->time_b, B, Ss, Sd, Ts, T,
                       d.P, d.Ft, G(grads->time_w), G(grads->time_b));
    DYGNN_LAUNCH_CHECK();
    // co-occurrence encoder
    if (int rc = mm(s, dXc + (size_t)3 * M * Cp, Cp, false, PW[3], PK[3], false, F32(p.dPc), PK[3], (int)M, PK[3], C, nullptr, 1.f, 0.f, 1, 1, 0, 0, 0, 0, 0, 0, false, false,
                    nullptr, nullptr, true)) return rc;
    DYGNN_HIP(hipMemsetAsync(F32(p.dlut), 0, (size_t)(S + 1) * C * sizeof(float), s));
    hipLaunchKernelGGL(k_cooc_bwd, dim3((unsigned)ceil_div(B, kPairsPerWg)), dim3(256), (size_t)kCoocGroups * kCoocRows * C * sizeof(float), s, F32(p.dPc),
                       I32(p.c0), I32(p.c1), B, Ss, Sd, Ts, T, d.P, C, F32(p.dlut));
    DYGNN_LAUNCH_CHECK();
    float* dh = F32(p.dPc);        // dPc is dead now: reuse its head for dh [S+1][C]
    hipLaunchKernelGGL(k_lut_bwd_hidden, dim3((unsigned)(S + 1)), dim3(64), 0, s, F32(p.dlut), F32(p.hid), w->cooc_w1, C, dh);
    DYGNN_LAUNCH_CHECK();
    DYGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lut_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k_lut_bwd, dim3((unsigned)ceil_div(C * C + 2 * C, 256)), dim3(256), (size_t)3 * kLutChunk * C * sizeof(float), s, F32(p.dlut), F32(p.hid), dh, S + 1, C, G(grads->cooc_w0),
                       G(grads->cooc_b0), G(grads->cooc_w1), G(grads->cooc_b1));
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}
