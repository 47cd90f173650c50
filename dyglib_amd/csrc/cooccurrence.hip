// Neighbour co-occurrence counts (reference models/DyGFormer.py:337-393), integer-exact.
//
// The reference loops over the batch in Python: two np.unique calls, two dicts and two
// Tensor.apply_ lambdas per pair (17.7 % of its CPU time).  Here one workgroup
// owns one (src,dst) pair: both padded id rows are staged once in LDS, every lane owns one
// position and sweeps both rows with wave-uniform (broadcast, conflict-free) LDS reads, so one
// wave-instruction advances 64 positions' counts at a time.  (A ballot+popcount per position
// yields ONE position's count per instruction and is 64x less efficient; ballots are used where
// they pay — the 64-ary row search in sampler.hip.)
#include "common.h"

namespace dygnn {

__global__ __launch_bounds__(256) void k_cooccurrence(const int64_t* __restrict__ src_ids, const int64_t* __restrict__ dst_ids,
                                                        int32_t S_s, int32_t S_d, float* __restrict__ cnt_src,
                                                        float* __restrict__ cnt_dst) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int64_t* row = reinterpret_cast<int64_t*>(smem);           // [S_s + S_d]: src row then dst row
    const int64_t b = blockIdx.x;
    const int32_t S = S_s + S_d;
    for (int32_t p = threadIdx.x; p < S; p += blockDim.x)
        row[p] = p < S_s ? src_ids[b * S_s + p] : dst_ids[b * S_d + (p - S_s)];
    __syncthreads();
    for (int32_t p = threadIdx.x; p < S; p += blockDim.x) {
        const int64_t v = row[p];
        int32_t c_src = 0, c_dst = 0;
        for (int32_t q = 0; q < S_s; ++q) c_src += (row[q] == v);
        for (int32_t q = S_s; q < S; ++q) c_dst += (row[q] == v);
        if (v == 0) { c_src = 0; c_dst = 0; }                  // models/DyGFormer.py:389-391
        float* out = p < S_s ? cnt_src + (b * S_s + p) * 2 : cnt_dst + (b * S_d + (p - S_s)) * 2;
        out[0] = (float)c_src;                                  // [count in src row, count in dst row] (:374, :380)
        out[1] = (float)c_dst;
    }
}

}  // namespace dygnn

using namespace dygnn;

extern "C" int dygnn_cooccurrence(const int64_t* src_ids, const int64_t* dst_ids, int64_t n, int32_t S_s, int32_t S_d,
                                  float* cnt_src, float* cnt_dst, dygnn_stream_t stream) {
    DYGNN_REQUIRE(n >= 0 && S_s >= 1 && S_d >= 1, "cooccurrence: bad sizes");
    DYGNN_REQUIRE(n == 0 || (src_ids && dst_ids && cnt_src && cnt_dst), "cooccurrence: null pointer");
    const size_t lds = (size_t)(S_s + S_d) * sizeof(int64_t);
    DYGNN_REQUIRE(lds <= 64 * 1024, "cooccurrence: rows too long for one workgroup (S_src+S_dst=%d)", S_s + S_d);
    if (n == 0) return DYGNN_OK;
    hipLaunchKernelGGL(k_cooccurrence, dim3((unsigned)n), dim3(256), lds, as_stream(stream), src_ids, dst_ids, S_s, S_d,
                       cnt_src, cnt_dst);
    DYGNN_LAUNCH_CHECK();
    return DYGNN_OK;
}
