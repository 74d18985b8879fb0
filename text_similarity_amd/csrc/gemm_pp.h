// Host-side entry points of gemm_pp.hip (large-K "ping-pong" projection GEMM + residual/LayerNorm row kernel).
#pragma once
#include "common.h"

namespace tsim {

enum { PP_EPI_BIAS = 0, PP_EPI_GELU = 1, PP_EPI_F32 = 2 };   // bf16 out | bf16 out after GELU(erf) | fp32 out

// N % 256 == 0, K % 64 == 0, K >= 128.  Row counts: X, out must be allocated for ceil(M/256)*256 rows.
bool gemm_pp_supported(int N, int K);
int gemm_pp(int epi, const bf16_t *X, const bf16_t *W, const float *bias, void *out, int M, int N, int K, hipStream_t st);

// out = LayerNorm(y + res) * gamma + beta over rows of H (256, 512, 768, 1024) features.
int res_ln_rows(const float *y, const bf16_t *res, const float *gamma, const float *beta, float eps, bf16_t *out, int M,
                int H, hipStream_t st);

}  // namespace tsim
