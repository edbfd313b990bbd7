// Producer / consumer weight-gradient kernel (wgrad_stream.hip), launched by dg_linear_wgrad (linear_wgrad.hip).
#pragma once

#include "common.h"

namespace dg {

bool wgrad_stream_supported(int N, int K);
// workgroups (= split-K partials) the launch uses for R rows.  may_wait: the caller's reduce is deferred
// (dg_linear_wgrad_batch), so inside dg_launch_pair_begin / _end a node-level launch may wait for a carrier (pair.h) --
// the SAME value must be passed to launch_wgrad_stream
int wgrad_stream_blocks(int64_t R, int N, int K, bool may_wait = false);
// part_w [blocks][N][K], part_b [blocks][N] (nullable): partial sums, reduced by the caller in a fixed order
// dy1 / dy2 (N = 384, K = 128 only): dy's three 128-column blocks given as three [R,128] matrices
// hfmt 2 (DG_DTYPE_F32_H24): the 384-wide operand holds the top 24 bits of every float32, three bytes per element.
// hfmt 1, hscale != NULL (DG_DTYPE_F32_H16): the 384-wide operand (dy for N = 384, x for K = 384) is one fp16 plane with the inverse row
// scales hscale [R]; the other operand is float32
int launch_wgrad_stream(const void* dy, const void* x, float* part_w, float* part_b, int64_t R, int N, int K, int blocks,
                        hipStream_t stream, const float* dy1 = nullptr, const float* dy2 = nullptr, bool may_wait = false,
                        const float* hscale = nullptr, int hfmt = 0);
// launches a problem that is still waiting for its carrier (pair.h)
int flush_wgrad_stream(hipStream_t stream);

}  // namespace dg
