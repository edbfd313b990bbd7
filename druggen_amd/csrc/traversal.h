// Traversal direction of the edge-level streaming kernels.  MI355X has a 256 MB memory-side cache: when a launch over
// B N^2 rows ends, the last ~256 MB it touched are still there -- a quarter of what the next launch reads, if that launch
// STARTS with those rows.  Every launch whose results do not depend on the order of its rows (row GEMMs: each output row is a
// function of its input row) therefore walks the rows in the direction opposite to its predecessor's: ascending, descending,
// ascending ...  Kernels whose results depend on the order (split-K partial sums: weight gradients, LayerNorm dgamma / dbeta)
// always ascend and say so.  All of them assign row tiles to workgroups round-robin, so "where the launch is" means the same
// window of rows for all.  Per host thread; DG_TRAVERSAL=forward switches the alternation off (A/B measurements).
#pragma once

#include "common.h"

namespace dg {

// direction of an order-independent launch over R rows: 1 = descending.  Launches below the edge level always ascend and
// leave the state alone.
int take_direction(int64_t R);
// an edge-level launch that always ascends
void note_forward(int64_t R);

}  // namespace dg
