// Internal (non-ABI) entry points shared between the translation units of libs2s_mi355x.
#pragma once
#include "common.h"

namespace s2s {

int gemm(hipStream_t st, const void* A, long lda, const void* W, const float* bias,
         const float* residual, void* C, long ldc, int M, int N, int K, int ab_dtype, int c_dtype,
         int act, bool allow_splitk);
int layernorm(hipStream_t st, const float* x, const float* gamma, const float* beta, void* y, int rows,
              int cols, float eps, int y_dtype);
int rmsnorm(hipStream_t st, const float* x, const float* gamma, void* y, int rows, int cols, float eps,
            int y_dtype);

}  // namespace s2s
