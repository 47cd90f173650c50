// General fp32-MFMA GEMM of the library (defined in dygformer_train.hip):
//   C = [relu] alpha * op(A) . op(B) (+ bias[n]) (+ beta * C), op(A)[m][k] = tA ? A[k*lda + m] : A[m*lda + k], op(B)[k][n] = tB ? B[n*ldb + k] : B[k*ldb + n]
// Large-K products are split over workgroups and summed with atomics into C (zeroed first unless c_is_zero says it already is).
// `batch` products in one launch: batch index z = zb * H + zh adds zb*sXb + zh*sXh floats to the three base pointers.
#pragma once
#include "common.h"

namespace dygnn {
namespace train {
int mm(hipStream_t s, const float* A, int lda, bool tA, const float* B, int ldb, bool tB, float* C, int ldc, int M, int N, int K,
       const float* bias = nullptr, float alpha = 1.f, float beta = 0.f, int batch = 1, int H = 1, int64_t sAb = 0, int64_t sAh = 0,
       int64_t sBb = 0, int64_t sBh = 0, int64_t sCb = 0, int64_t sCh = 0, bool relu = false, bool c_is_zero = false,
       float* colsum_out = nullptr,       // tA only: colsum_out[m] += sum_k op(A)[m][k] (accumulates: zero it first)
       const int32_t* m_dev = nullptr);   // device-side count of live rows (<= M): row tiles beyond it are skipped on the device
}  // namespace train
}  // namespace dygnn
