// Producer / consumer 384 -> 128 row GEMM (row_gemm_k384.hip), launched by dg_row_gemm (row_gemm.hip).
#pragma once

#include "common.h"

namespace dg {

// y [R,128] = LN?(a [R,384] . B + bias (+ relu) + residual?); `packed` from dg_row_gemm_pack (fp16 hi + lo, K = 384);
// gamma != NULL: LayerNorm epilogue, writes mean / rstd [R] and, if pre_ln != NULL, the pre-LayerNorm sum
// afmt 3 (DG_DTYPE_F32_H32): a = the hi plane, alo = the lo plane (fp16 each), ascale the inverse row scales: float32 class.
// afmt 2 (DG_DTYPE_F32_H24): a holds the top 24 bits of every float32, three bytes per element.
// ascale == NULL: a is float32.  ascale != NULL (DG_DTYPE_F32_H16): a is one fp16 plane with the inverse row scales ascale [R].
int launch_row_gemm_k384(const void* a, const float* ascale, const void* packed, float* y, int64_t R, const float* bias, int relu,
                         const float* residual, const float* gamma, const float* beta, float* mean, float* rstd,
                         float* pre_ln, float eps, hipStream_t stream, int afmt = 0, const void* alo = nullptr);
// launches a problem that is still waiting for its carrier (pair.h)
int flush_row_gemm_k384(hipStream_t stream);

}  // namespace dg
