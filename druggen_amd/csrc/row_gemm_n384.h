// Producer / consumer 128 -> 384 row GEMM (row_gemm_n384.hip), launched by dg_row_gemm (row_gemm.hip).
#pragma once

#include "common.h"

namespace dg {

// ReLU bit-mask words of a launch over R rows (one word per lane, consumer wave and 32-row stage)
size_t row_gemm_n384_mask_words(int64_t R);
// y [R,384] = epi(a [R,128] . B): bias, ReLU (+ bit mask out), bit mask in; `packed` from dg_row_gemm_pack (fp16 hi + lo).
// yscale == NULL: y is float32.  yscale != NULL (DG_DTYPE_F32_H16): y is ONE fp16 plane, row r scaled so that its largest
// magnitude lies in [2^14, 2^15), and yscale[r] the inverse (power-of-two) scale.
// yfmt 3 (DG_DTYPE_F32_H32): y = the hi plane, ylo = the lo plane (hi + lo = the row-scaled float32 to 22 bits), yscale as for H16.
// yfmt 2 (DG_DTYPE_F32_H24): y holds the top 24 bits of every float32, three bytes per element.
int launch_row_gemm_n384(const float* a, const void* packed, void* y, float* yscale, int64_t R, const float* bias, int relu,
                         unsigned* relu_bits, const unsigned* mask_bits, hipStream_t stream, int yfmt = 0, void* ylo = nullptr);
// launches a problem that is still waiting for its carrier (pair.h)
int flush_row_gemm_n384(hipStream_t stream);

}  // namespace dg
