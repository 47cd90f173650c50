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
       const int32_t* m_dev = nullptr,    // device-side count of live rows (<= M): row tiles beyond it are skipped on the device
       bool a_kpad = false,               // A [M][lda] has zeros behind its K columns up to a multiple of 4 (K itself need not be one)
       const int32_t* a_rows = nullptr);  // !tA only: row m of the product's A is A[a_rows[m]] (device table; the product gathers its rows: no staging copy)
// Weight-gradient-shaped products in one grouped split-K launch (k_dw_grouped): C_p[m][n] += sum_k A_p[k][m] * B_p[k][n] for every problem p, all
// over the same K rows; colsum_p[m] += sum_k A_p[k][m] where given.  C and colsum ACCUMULATE (atomics): zero them first.  A, B 16-byte aligned,
// lda, ldb, M, N multiples of 4.
struct DwPair { const float* A; int lda, M; const float* B; int ldb, N; float* C; int ldc; float* colsum; };
int dw_grouped(hipStream_t s, int K, const DwPair* pairs, int npairs);
}  // namespace train
}  // namespace dygnn
