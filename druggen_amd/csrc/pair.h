// Riding launches.  A step issues every feed-forward / weight-gradient kernel twice per encoder block: once over the
// B N^2 edge rows (hundreds of microseconds) and once over the B N node rows (10-25 us, launch-latency-bound: the kernel's
// prologue, one or two stages per workgroup, its tail).  Between dg_launch_pair_begin() and dg_launch_pair_end() a
// node-level launch of one of the producer / consumer kernels (row_gemm_n384.hip, row_gemm_k384.hip, wgrad_stream.hip) is
// not issued but kept -- one per kernel -- and rides in the NEXT launch of the same kernel: that launch runs two problems,
// the workgroups split in proportion to their stages.  The caller guarantees that nothing reads a rider's output before
// its carrier is launched (functional.py: the paired feed-forward nodes issue node, edge, node, edge ...).
// LIFETIME CONTRACT: a waiting launch is a set of raw device pointers.  Every buffer handed to a launch inside the region
// must stay allocated (not returned to a stream-ordered / caching allocator) until dg_launch_pair_end() has returned; the
// Python binding references all operands of such launches until then (functional._pair_hold).  A ProfScope around a call
// that only parks a rider records an empty span: the rider's time is part of its carrier's span.
#pragma once

#include "common.h"

namespace dg {

// set between dg_launch_pair_begin() and dg_launch_pair_end(), per host thread (runtime.hip)
bool pair_mode();
// launches at most this many rows wait for a carrier
constexpr int64_t kRiderMaxRows = 65536;

// workgroups of a launch over st0 (+ st1) stages with at most maxb workgroups: nb0 for problem 0, nb1 for problem 1
// (>= 1 each when st1 > 0, never more workgroups than stages)
inline void pair_split(int64_t st0, int64_t st1, int maxb, int* nb0, int* nb1) {
    if (st1 <= 0) {
        *nb0 = static_cast<int>(st0 < maxb ? st0 : maxb);
        *nb1 = 0;
        return;
    }
    if (st0 + st1 <= maxb) {
        *nb0 = static_cast<int>(st0);
        *nb1 = static_cast<int>(st1);
        return;
    }
    int64_t b1 = (st1 * maxb + (st0 + st1) / 2) / (st0 + st1);
    b1 = b1 < 1 ? 1 : b1;
    b1 = b1 > st1 ? st1 : b1;
    int64_t b0 = maxb - b1;
    if (b0 > st0) {
        b0 = st0;
        b1 = maxb - b0 < st1 ? maxb - b0 : st1;
    }
    *nb0 = static_cast<int>(b0);
    *nb1 = static_cast<int>(b1);
}

}  // namespace dg
