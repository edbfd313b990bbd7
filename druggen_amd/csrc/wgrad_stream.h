// Producer / consumer weight-gradient kernel (wgrad_stream.hip), launched by dg_linear_wgrad (linear_wgrad.hip).
#pragma once

#include "common.h"

namespace dg {

bool wgrad_stream_supported(int N, int K);
// workgroups (= split-K partials) the launch uses for R rows
int wgrad_stream_blocks(int64_t R, int N, int K);
// part_w [blocks][N][K], part_b [blocks][N] (nullable): partial sums, reduced by the caller in a fixed order
// dy1 / dy2 (N = 384, K = 128 only): dy's three 128-column blocks given as three [R,128] matrices
int launch_wgrad_stream(const float* dy, const float* x, float* part_w, float* part_b, int64_t R, int N, int K, int blocks,
                        hipStream_t stream, const float* dy1 = nullptr, const float* dy2 = nullptr);

}  // namespace dg
