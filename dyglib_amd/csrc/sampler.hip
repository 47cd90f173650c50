// Temporal neighbour lookup kernels (HBM/latency-bound integer work, bit-exact).
//
// One 64-lane wavefront per query.  The lower bound of find_neighbors_before (reference
// utils/utils.py:139-141: np.searchsorted(times[node], t), side='left' => strictly-earlier
// prefix) is a 64-ary search: every step the wave probes 64 evenly spaced timestamps of the row
// and one ballot+popcount picks the sub-range, so a row of degree d costs ceil(log64(d)) dependent
// memory round trips instead of log2(d) — rows up to 64 entries (the common case) take ONE
// coalesced 512-byte read.  The most-recent-k tail [i-k, i) of the CSR row is then copied with
// consecutive lanes on consecutive entries (coalesced int32/int32/f64 reads, coalesced writes).
#include "common.h"

namespace dygnn {

// first index p in [lo, hi) with ts[p] >= t, or hi.  All 64 lanes must call (uniform lo/hi/t).
__device__ __forceinline__ int64_t wave_lower_bound(const double* __restrict__ ts, int64_t lo, int64_t hi, double t, int lane) {
    while (hi - lo > kWave) {
        const int64_t step = (hi - lo + kWave - 1) / kWave;
        const int64_t p = lo + (int64_t)lane * step;
        const bool pred = (p < hi) && (ts[p] < t);
        const int c = __popcll(__ballot(pred));      // rows ascend => pred is true exactly for lanes < c
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

struct CsrView {
    const int64_t* indptr;
    const int32_t* nbr;
    const int32_t* eid;
    const double* ts;
    int64_t num_nodes;
};

__device__ __forceinline__ void query_row(const CsrView& g, int64_t node, int64_t& row_lo, int64_t& row_hi) {
    // ids are trusted by the reference (an out-of-range id is an IndexError there); here an
    // out-of-range id is clamped to the empty padding row instead of faulting the GPU.
    if (node < 0 || node >= g.num_nodes) node = 0;
    row_lo = g.indptr[node];
    row_hi = g.indptr[node + 1];
}

__global__ __launch_bounds__(256) void k_find_before(CsrView g, const int64_t* __restrict__ nodes,
                                                       const double* __restrict__ times, int64_t n, int32_t clampL,
                                                       int32_t* __restrict__ hist_len, int64_t* __restrict__ end_pos,
                                                       int32_t* __restrict__ max_window) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= n) return;   // wave-uniform
    int64_t lo, hi;
    query_row(g, nodes[q], lo, hi);
    const int64_t i = wave_lower_bound(g.ts, lo, hi, times[q], lane);
    if (lane == 0) {
        const int32_t len = (int32_t)(i - lo);
        if (hist_len) hist_len[q] = len;
        if (end_pos) end_pos[q] = i;
        if (max_window) {
            // the maximum is monotone: read it first and skip the atomic when this query cannot raise it (after the first
            // few waves almost none can) — 2 M same-address atomics cost 20 ms, the plain reads hit L2
            const int32_t v = len < clampL ? len : clampL;
            if (v > __atomic_load_n(max_window, __ATOMIC_RELAXED)) atomicMax(max_window, v);
        }
    }
}

__global__ __launch_bounds__(256) void k_sample_recent(CsrView g, const int64_t* __restrict__ nodes,
                                                         const double* __restrict__ times, int64_t n, int32_t k,
                                                         int64_t* __restrict__ out_nbr, int64_t* __restrict__ out_eid,
                                                         float* __restrict__ out_ts) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= n) return;
    int64_t lo, hi;
    query_row(g, nodes[q], lo, hi);
    const int64_t i = wave_lower_bound(g.ts, lo, hi, times[q], lane);
    const int64_t len = i - lo;
    const int32_t m = (int32_t)(len < k ? len : k);     // utils/utils.py:202-204
    const int32_t pad = k - m;                          // right-aligned, utils/utils.py:207-209
    int64_t* on = out_nbr + q * k;
    int64_t* oe = out_eid + q * k;
    float* ot = out_ts + q * k;
    for (int32_t j = lane; j < k; j += kWave) {
        if (j < pad) {
            on[j] = 0; oe[j] = 0; ot[j] = 0.0f;
        } else {
            const int64_t p = i - m + (j - pad);
            on[j] = g.nbr[p]; oe[j] = g.eid[p]; ot[j] = (float)g.ts[p];   // f64 -> f32 on store, utils.py:167
        }
    }
}

// ---- four queries per wave (large batches) ----------------------------------------------------------------------------------
// One wave per query leaves a CU with 32 queries in flight, each a chain of two or three dependent memory round trips: the kernel is
// bound by that latency, not by bytes.  With 16 lanes per query a wave carries FOUR searches (the rows of a 16-ary search are still one
// 128-byte line per probe step), so 128 queries are in flight per CU; the sub-group's ballot bits are cut out of the wave's 64-bit
// ballot.  A search step divides the range by 16 instead of 64 (one more round trip on rows beyond 1024 entries), which the four-fold
// overlap more than pays for once the batch is large enough to fill the chip (dispatch: n >= 16384).
constexpr int kSub = 16;

// first index p in [lo, hi) with ts[p] >= t, or hi, for the query of this lane's 16-lane sub-group.  All 64 lanes must call; `on`
// = the sub-group has a query (inactive sub-groups pass lo == hi).
__device__ __forceinline__ int64_t sub_lower_bound(const double* __restrict__ ts, int64_t lo, int64_t hi, double t, int sl, int sg) {
    while (__any(hi - lo > kSub)) {
        const bool wide = hi - lo > kSub;                                  // uniform inside a sub-group
        const int64_t step = wide ? (hi - lo + kSub - 1) / kSub : 1;
        const int64_t p = lo + (int64_t)sl * step;
        const bool pred = wide && (p < hi) && (ts[p] < t);
        const int c = __popc((unsigned)((__ballot(pred) >> (kSub * sg)) & 0xFFFFull));   // rows ascend => true exactly for sub-lanes < c
        if (wide) {
            if (c == 0) hi = lo;                                           // everything is >= t: the answer is lo
            else {
                const int64_t nlo = lo + (int64_t)(c - 1) * step + 1, nhi = lo + (int64_t)c * step;
                hi = nhi < hi ? nhi : hi;
                lo = nlo;
            }
        }
    }
    const int64_t p = lo + sl;
    const bool pred = (p < hi) && (ts[p] < t);
    return lo + __popc((unsigned)((__ballot(pred) >> (kSub * sg)) & 0xFFFFull));
}

__global__ __launch_bounds__(256) void k_sample_recent_sub(CsrView g, const int64_t* __restrict__ nodes, const double* __restrict__ times, int64_t n,
                                                             int32_t k, int64_t* __restrict__ out_nbr, int64_t* __restrict__ out_eid,
                                                             float* __restrict__ out_ts) {
    const int lane = threadIdx.x & 63, sg = lane / kSub, sl = lane % kSub;
    const int64_t q = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (kWave / kSub) + sg;
    const bool on = q < n;
    int64_t lo = 0, hi = 0;
    double t = 0.0;
    if (on) { query_row(g, nodes[q], lo, hi); t = times[q]; }
    const int64_t i = sub_lower_bound(g.ts, lo, hi, t, sl, sg);
    if (!on) return;
    const int64_t len = i - lo;
    const int32_t m = (int32_t)(len < k ? len : k);     // utils/utils.py:202-204
    const int32_t pad = k - m;                          // right-aligned, utils/utils.py:207-209
    int64_t* on_ = out_nbr + q * k;
    int64_t* oe = out_eid + q * k;
    float* ot = out_ts + q * k;
    for (int32_t j = sl; j < k; j += kSub) {
        if (j < pad) {
            on_[j] = 0; oe[j] = 0; ot[j] = 0.0f;
        } else {
            const int64_t p = i - m + (j - pad);
            on_[j] = g.nbr[p]; oe[j] = g.eid[p]; ot[j] = (float)g.ts[p];   // f64 -> f32 on store, utils.py:167
        }
    }
}

__global__ __launch_bounds__(256) void k_find_before_sub(CsrView g, const int64_t* __restrict__ nodes, const double* __restrict__ times, int64_t n,
                                                           int32_t clampL, int32_t* __restrict__ hist_len, int64_t* __restrict__ end_pos,
                                                           int32_t* __restrict__ max_window) {
    const int lane = threadIdx.x & 63, sg = lane / kSub, sl = lane % kSub;
    const int64_t q = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (kWave / kSub) + sg;
    const bool on = q < n;
    int64_t lo = 0, hi = 0;
    double t = 0.0;
    if (on) { query_row(g, nodes[q], lo, hi); t = times[q]; }
    const int64_t row_lo = lo;
    const int64_t i = sub_lower_bound(g.ts, lo, hi, t, sl, sg);
    if (on && sl == 0) {
        const int32_t len = (int32_t)(i - row_lo);
        if (hist_len) hist_len[q] = len;
        if (end_pos) end_pos[q] = i;
        if (max_window) {
            const int32_t v = len < clampL ? len : clampL;
            if (v > __atomic_load_n(max_window, __ATOMIC_RELAXED)) atomicMax(max_window, v);
        }
    }
}

// gather of host-drawn samples (utils/utils.py:192-199): one thread per output slot
__global__ __launch_bounds__(256) void k_gather_selected(CsrView g, const int64_t* __restrict__ nodes, const int32_t* __restrict__ sel,
                                                           int64_t total, int32_t k, int64_t* __restrict__ out_nbr,
                                                           int64_t* __restrict__ out_eid, float* __restrict__ out_ts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int32_t j = sel[i];
    int64_t nb = 0, ed = 0;
    float t = 0.0f;
    if (j >= 0) {
        int64_t lo, hi;
        query_row(g, nodes[i / k], lo, hi);
        const int64_t p = lo + j;
        if (p < hi) { nb = g.nbr[p]; ed = g.eid[p]; t = (float)g.ts[p]; }
    }
    out_nbr[i] = nb; out_eid[i] = ed; out_ts[i] = t;
}

__global__ __launch_bounds__(256) void k_window_fill(CsrView g, const int64_t* __restrict__ nodes,
                                                       const double* __restrict__ times, int64_t n, int32_t L, int32_t S,
                                                       const int32_t* __restrict__ hist_len, const int64_t* __restrict__ end_pos,
                                                       int64_t* __restrict__ out_ids, int64_t* __restrict__ out_eids,
                                                       float* __restrict__ out_ts) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= n) return;
    const int32_t len = hist_len[q];
    const int32_t m = len < L - 1 ? len : L - 1;        // models/DyGFormer.py:214-218
    const int64_t first = end_pos[q] - m;
    int64_t* oi = out_ids + q * S;
    int64_t* oe = out_eids + q * S;
    float* ot = out_ts + q * S;
    for (int32_t j = lane; j < S; j += kWave) {
        if (j == 0) {
            oi[0] = nodes[q]; oe[0] = 0; ot[0] = (float)times[q];            // models/DyGFormer.py:235-237
        } else if (j <= m) {
            const int64_t p = first + (j - 1);
            oi[j] = g.nbr[p]; oe[j] = g.eid[p]; ot[j] = (float)g.ts[p];      // :240-242
        } else {
            oi[j] = 0; oe[j] = 0; ot[j] = 0.0f;
        }
    }
}

static int check_csr(const dygnn_csr* c) {
    DYGNN_REQUIRE(c != nullptr, "csr is NULL");
    DYGNN_REQUIRE(c->num_nodes >= 1 && c->num_entries >= 0 && c->indptr, "csr: bad header");
    DYGNN_REQUIRE(c->num_entries == 0 || (c->nbr && c->eid && c->ts), "csr: null payload");
    return DYGNN_OK;
}

constexpr int64_t kSubMinQueries = 16384;     // below: one wave per query (shortest chain); at and above: four queries per wave
static CsrView view(const dygnn_csr* c) { return CsrView{c->indptr, c->nbr, c->eid, c->ts, c->num_nodes}; }

}  // namespace dygnn

using namespace dygnn;

extern "C" int dygnn_find_neighbors_before(const dygnn_csr* csr, const int64_t* nodes, const double* times, int64_t n,
                                           int32_t* hist_len, int64_t* end_pos, dygnn_stream_t stream) {
    if (int rc = check_csr(csr)) return rc;
    DYGNN_REQUIRE(n >= 0 && (n == 0 || (nodes && times)), "find_neighbors_before: bad arguments");
    if (n == 0) return DYGNN_OK;
    const int wpb = 4;
    if (n >= kSubMinQueries)
        hipLaunchKernelGGL(k_find_before_sub, dim3((unsigned)ceil_div(n, wpb * (kWave / kSub))), dim3(wpb * kWave), 0, as_stream(stream), view(csr),
                           nodes, times, n, INT32_MAX, hist_len, end_pos, (int32_t*)nullptr);
    else
    hipLaunchKernelGGL(k_find_before, dim3((unsigned)ceil_div(n, wpb)), dim3(wpb * kWave), 0, as_stream(stream), view(csr),
                       nodes, times, n, INT32_MAX, hist_len, end_pos, (int32_t*)nullptr);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

extern "C" int dygnn_sample_recent(const dygnn_csr* csr, const int64_t* nodes, const double* times, int64_t n, int32_t k,
                                   int64_t* out_nbr, int64_t* out_eid, float* out_ts, dygnn_stream_t stream) {
    if (int rc = check_csr(csr)) return rc;
    // utils/utils.py:157
    DYGNN_REQUIRE(k > 0, "Number of sampled neighbors for each node should be greater than 0!");
    DYGNN_REQUIRE(n >= 0 && (n == 0 || (nodes && times && out_nbr && out_eid && out_ts)), "sample_recent: bad arguments");
    if (n == 0) return DYGNN_OK;
    const int wpb = 4;
    if (n >= kSubMinQueries)
        hipLaunchKernelGGL(k_sample_recent_sub, dim3((unsigned)ceil_div(n, wpb * (kWave / kSub))), dim3(wpb * kWave), 0, as_stream(stream), view(csr),
                           nodes, times, n, k, out_nbr, out_eid, out_ts);
    else
    hipLaunchKernelGGL(k_sample_recent, dim3((unsigned)ceil_div(n, wpb)), dim3(wpb * kWave), 0, as_stream(stream), view(csr),
                       nodes, times, n, k, out_nbr, out_eid, out_ts);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

extern "C" int dygnn_gather_selected(const dygnn_csr* csr, const int64_t* nodes, const int32_t* sel, int64_t n, int32_t k,
                                     int64_t* out_nbr, int64_t* out_eid, float* out_ts, dygnn_stream_t stream) {
    if (int rc = check_csr(csr)) return rc;
    DYGNN_REQUIRE(k > 0, "Number of sampled neighbors for each node should be greater than 0!");     // utils/utils.py:157
    DYGNN_REQUIRE(n >= 0 && (n == 0 || (nodes && sel && out_nbr && out_eid && out_ts)), "gather_selected: bad arguments");
    if (n == 0) return DYGNN_OK;
    const int64_t total = n * k;
    hipLaunchKernelGGL(k_gather_selected, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream), view(csr), nodes, sel,
                       total, k, out_nbr, out_eid, out_ts);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

extern "C" int dygnn_window_lengths(const dygnn_csr* csr, const int64_t* nodes, const double* times, int64_t n,
                                    int32_t L, int32_t* hist_len, int64_t* end_pos, int32_t* max_window,
                                    dygnn_stream_t stream) {
    if (int rc = check_csr(csr)) return rc;
    // models/DyGFormer.py:209
    DYGNN_REQUIRE(L - 1 > 0, "Maximal number of neighbors for each node should be greater than 1!");
    DYGNN_REQUIRE(n >= 0 && max_window && (n == 0 || (nodes && times)), "window_lengths: bad arguments");
    DYGNN_HIP(hipMemsetAsync(max_window, 0, sizeof(int32_t), as_stream(stream)));
    if (n == 0) return DYGNN_OK;
    const int wpb = 4;
    if (n >= kSubMinQueries)
        hipLaunchKernelGGL(k_find_before_sub, dim3((unsigned)ceil_div(n, wpb * (kWave / kSub))), dim3(wpb * kWave), 0, as_stream(stream), view(csr),
                           nodes, times, n, L - 1, hist_len, end_pos, max_window);
    else
    hipLaunchKernelGGL(k_find_before, dim3((unsigned)ceil_div(n, wpb)), dim3(wpb * kWave), 0, as_stream(stream), view(csr),
                       nodes, times, n, L - 1, hist_len, end_pos, max_window);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}

extern "C" int dygnn_window_fill(const dygnn_csr* csr, const int64_t* nodes, const double* times, int64_t n, int32_t L,
                                 int32_t S, const int32_t* hist_len, const int64_t* end_pos, int64_t* out_ids,
                                 int64_t* out_eids, float* out_ts, dygnn_stream_t stream) {
    if (int rc = check_csr(csr)) return rc;
    DYGNN_REQUIRE(L - 1 > 0, "Maximal number of neighbors for each node should be greater than 1!");
    DYGNN_REQUIRE(S >= 1 && n >= 0, "window_fill: bad sizes");
    DYGNN_REQUIRE(n == 0 || (nodes && times && hist_len && end_pos && out_ids && out_eids && out_ts), "window_fill: null pointer");
    if (n == 0) return DYGNN_OK;
    const int wpb = 4;
    hipLaunchKernelGGL(k_window_fill, dim3((unsigned)ceil_div(n, wpb)), dim3(wpb * kWave), 0, as_stream(stream), view(csr),
                       nodes, times, n, L, S, hist_len, end_pos, out_ids, out_eids, out_ts);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}
