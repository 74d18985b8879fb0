// Host-side entry points of gemm_pp.hip (large-K "ping-pong" projection GEMM + residual/LayerNorm row kernel).
#pragma once
#include "common.h"

namespace tsim {

// bf16 out | bf16 out after GELU(erf) | fp32 out | MXFP8 out after GELU (bytes + block scales; MX operands only)
enum { PP_EPI_BIAS = 0, PP_EPI_GELU = 1, PP_EPI_F32 = 2, PP_EPI_GELU_MX = 3 };

// N % 128 == 0, K % 64 == 0, K >= 128.  Row counts: X, out must be allocated for ceil(M/256)*256 rows.
bool gemm_pp_supported(int N, int K);
// w_packed: W is the re-laid copy made by pack_w(W, Wp, N, K * 2, gemm_pp_tile_width(N)) — contiguous 1-KiB DMA pieces.
int gemm_pp_tile_width(int N);
int pack_w(const void *W, void *Wp, int N, int kbytes, int BN, hipStream_t st);
int gemm_pp(int epi, const bf16_t *X, const bf16_t *W, int w_packed, const float *bias, void *out, int M, int N, int K,
            hipStream_t st);

// MXFP8 operands (OCP MX: e4m3 elements, one E8M0 scale byte per 32 elements along K; scale arrays [rows, K/32]):
// N % 256 == 0, K % 128 == 0, K >= 256.  v_mfma_scale_f32_32x32x64_f8f6f4, twice the bf16 MFMA rate.
bool gemm_pp_mx_supported(int N, int K);
int gemm_pp_mx(int epi, const uint8_t *Xq, const uint8_t *Xs, const uint8_t *Wq, int w_packed, const uint8_t *Ws,
               const float *bias, void *out, uint8_t *out_scales, int M, int N, int K, hipStream_t st);
// bf16 [rows, K] -> MXFP8 bytes [rows, K] + scales [rows, K/32]
int quant_mx(const bf16_t *x, int64_t rows, int K, uint8_t *q, uint8_t *scales, hipStream_t st);

// out = LayerNorm(y + res) * gamma + beta over rows of H (256, 512, 768, 1024) features.
// q / q_scales non-null: the row is also written as MXFP8 (the next projection's operand), fused.
int res_ln_rows(const float *y, const bf16_t *res, const float *gamma, const float *beta, float eps, bf16_t *out,
                uint8_t *q, uint8_t *q_scales, int M, int H, hipStream_t st);

}  // namespace tsim
