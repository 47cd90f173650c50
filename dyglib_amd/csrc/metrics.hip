// Link-prediction / node-classification metrics on the device (reference utils/metrics.py:5-34, which hands the scores
// to scikit-learn's average_precision_score and roc_auc_score on the host; evaluate_models_utils.py:139-150 calls it once
// per batch, :245-249 once per evaluation for node classification).
//
// Both scores are rank statistics, so no sort is needed.  With P positives and N negatives in a group:
//   average precision = (1/P) * sum over positives i of  #{positives j : s_j >= s_i} / #{all j : s_j >= s_i}
//     (scikit-learn sums (R_k - R_{k-1}) * P_k over the distinct thresholds; all positives of one tie group share P_k);
//   ROC AUC           = sum over positives i of (2 * #{negatives j : s_j < s_i} + #{negatives j : s_j == s_i}) / (2 P N)
//     (the trapezoid rule over the distinct thresholds = Mann-Whitney U with ties counted one half).
// The counts are exact integers (scores are compared as the float32 values the caller holds, like np.diff on the float32
// array in scikit-learn); the only floating-point work is one division per positive and a fixed-order float64 sum, so a
// result is reproducible from run to run.  Binary cross-entropy (torch.nn.BCELoss, mean reduction, logs clamped at -100;
// evaluate_models_utils.py:145) is produced by the same pass.
//
// Kernel 1: grid (ceil(n/256), groups).  A thread owns one sample and scans the whole group through LDS tiles (all lanes
// read the same LDS address = broadcast).  Kernel 2: one wave per group adds the per-block partials in a fixed order.
#include "common.h"

namespace dygnn {
namespace {

constexpr int kBlock = 256;
constexpr int kTile = 1024;

struct Partial {
    double ap;
    double loss;
    unsigned long long auc2;
    unsigned long long npos;
};

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ inline unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

__global__ __launch_bounds__(kBlock) void k_rank_counts(const float* __restrict__ predicts, const float* __restrict__ labels,
                                                        int64_t n, Partial* __restrict__ partials) {
    __shared__ float s_tile[kTile];
    __shared__ uint32_t y_tile[kTile];
    __shared__ double red_ap[kBlock / kWave], red_loss[kBlock / kWave];
    __shared__ unsigned long long red_auc[kBlock / kWave];

    const int64_t g = blockIdx.y;
    const float* s = predicts + g * n;
    const float* y = labels + g * n;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < n;
    const float si = live ? s[i] : 0.f;
    const bool pos_i = live && y[i] != 0.f;

    uint32_t all_ge = 0, pos_ge = 0, all_eq = 0, pos_eq = 0, npos = 0;
    for (int64_t base = 0; base < n; base += kTile) {
        const int m = (int)((n - base < kTile) ? (n - base) : kTile);
        __syncthreads();
        for (int j = threadIdx.x; j < m; j += kBlock) {
            s_tile[j] = s[base + j];
            y_tile[j] = y[base + j] != 0.f ? 1u : 0u;
        }
        __syncthreads();
        for (int j = 0; j < m; ++j) {
            const float sj = s_tile[j];
            const uint32_t yj = y_tile[j];
            const uint32_t ge = sj >= si ? 1u : 0u, eq = sj == si ? 1u : 0u;
            all_ge += ge;
            pos_ge += ge & yj;
            all_eq += eq;
            pos_eq += eq & yj;
            npos += yj;
        }
    }

    double ap = 0.0, loss = 0.0;
    unsigned long long auc2 = 0;
    if (live) {
        // torch.nn.BCELoss: -(y log p + (1-y) log(1-p)), each log clamped at -100, in float32
        const float yi = y[i];
        const float lp = fmaxf(logf(si), -100.f), lq = fmaxf(logf(1.f - si), -100.f);
        loss = (double)(-(yi * lp + (1.f - yi) * lq));
    }
    if (pos_i) {
        ap = (double)pos_ge / (double)all_ge;                    // all_ge >= 1: the sample itself
        const uint32_t nneg = (uint32_t)n - npos;
        const uint32_t neg_ge = all_ge - pos_ge, neg_eq = all_eq - pos_eq;
        auc2 = 2ull * (nneg - neg_ge) + neg_eq;
    }
    ap = wave_sum(ap);
    loss = wave_sum(loss);
    auc2 = wave_sum(auc2);
    const int w = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) {
        red_ap[w] = ap;
        red_loss[w] = loss;
        red_auc[w] = auc2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        Partial p{0.0, 0.0, 0ull, npos};
        for (int k = 0; k < kBlock / kWave; ++k) {
            p.ap += red_ap[k];
            p.loss += red_loss[k];
            p.auc2 += red_auc[k];
        }
        partials[g * gridDim.x + blockIdx.x] = p;
    }
}

__global__ __launch_bounds__(kWave) void k_metrics_finalize(const Partial* __restrict__ partials, int nblocks, int64_t n,
                                                            double* __restrict__ average_precision, double* __restrict__ roc_auc,
                                                            double* __restrict__ bce_loss, int32_t* __restrict__ status) {
    const int64_t g = blockIdx.x;
    const Partial* p = partials + g * nblocks;
    double ap = 0.0, loss = 0.0;
    unsigned long long auc2 = 0;
    for (int b = threadIdx.x; b < nblocks; b += kWave) {
        ap += p[b].ap;
        loss += p[b].loss;
        auc2 += p[b].auc2;
    }
    ap = wave_sum(ap);
    loss = wave_sum(loss);
    auc2 = wave_sum(auc2);
    if (threadIdx.x == 0) {
        const unsigned long long npos = p[0].npos, nneg = (unsigned long long)n - npos;
        const bool one_class = npos == 0 || nneg == 0;
        if (status) status[g] = one_class ? 1 : 0;
        // one class only: roc_auc_score raises ("Only one class present in y_true"); the wrapper does the same from `status`
        if (average_precision) average_precision[g] = npos ? ap / (double)npos : 0.0;
        if (roc_auc) roc_auc[g] = one_class ? __longlong_as_double(0x7ff8000000000000ll) : (double)auc2 / (2.0 * (double)npos * (double)nneg);
        if (bce_loss) bce_loss[g] = loss / (double)n;
    }
}

}  // namespace
}  // namespace dygnn

using namespace dygnn;

extern "C" size_t dygnn_link_metrics_workspace_bytes(int64_t group_size, int64_t n_groups) {
    if (group_size <= 0 || n_groups <= 0) return 0;
    return (size_t)(ceil_div(group_size, kBlock) * n_groups) * sizeof(Partial);
}

extern "C" int dygnn_link_metrics(const float* predicts, const float* labels, int64_t group_size, int64_t n_groups,
                                  double* average_precision, double* roc_auc, double* bce_loss, int32_t* status,
                                  void* workspace, size_t workspace_bytes, dygnn_stream_t stream) {
    DYGNN_REQUIRE(group_size > 0 && n_groups >= 0, "link_metrics: group_size must be positive");
    DYGNN_REQUIRE(group_size < (1ll << 31), "link_metrics: at most 2^31 - 1 samples per group");
    DYGNN_REQUIRE(n_groups <= 65535, "link_metrics: at most 65535 groups per call");
    if (n_groups == 0) return DYGNN_OK;
    DYGNN_REQUIRE(predicts && labels && workspace, "link_metrics: null pointer");
    if (workspace_bytes < dygnn_link_metrics_workspace_bytes(group_size, n_groups)) {
        set_error("link_metrics: workspace too small");
        return DYGNN_E_WORKSPACE;
    }
    const int nblocks = (int)ceil_div(group_size, kBlock);
    Partial* partials = static_cast<Partial*>(workspace);
    hipLaunchKernelGGL(k_rank_counts, dim3((unsigned)nblocks, (unsigned)n_groups), dim3(kBlock), 0, as_stream(stream), predicts,
                       labels, group_size, partials);
    DYGNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_metrics_finalize, dim3((unsigned)n_groups), dim3(kWave), 0, as_stream(stream), partials, nblocks,
                       group_size, average_precision, roc_auc, bce_loss, status);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}
