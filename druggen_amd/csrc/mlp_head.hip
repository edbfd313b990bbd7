// Tail of the Discriminator head (reference src/model/models.py:173-178, 207): after Linear(N dim, 64) the head is
//     a1 = act(z1) ; a2 = act(a1 W2^T + b2) ; a3 = act(a2 W3^T + b3) ; out = a3 W4^T + b4       64 -> 32 -> 16 -> 1
// over B (or 2B) rows -- on the BLAS + ATen that is 6 launches forward, 13 backward and ~20 in the second order of the
// gradient penalty, each a few microseconds of launch for a few thousand multiply-adds.  Three kernels cover all of it for the
// piecewise-linear activations (ReLU, LeakyReLU(0.01): act'' = 0, so the second order is the same chain with the forward's
// activation pattern as a mask):
//   head_chain   rows through the three layers.  Forward: activations from the pre-activations.  Second order (`m1` given):
//                u1 = t . act'(a1), u2 = (u1 W2^T) . act'(a2), u3 = (u2 W3^T) . act'(a3), u_out = u3 W4^T  -- the adjoints of
//                g2 W2, g3 W3, g_out W4 and g_out when t is the adjoint of the first backward's result.
//   head_bwd     g3 = (g_out W4) . act'(a3), g2 = (g3 W3) . act'(a2), g1 = (g2 W2) . act'(a1): gradient of z1.
//   head_wgrad   dW4 = l4^T r4, dW3 = l3^T r3, dW2 = l2^T r2 and the column sums of l4, l3, l2 -- first order with
//                (g_out, a3), (g3, a2), (g2, a1); second order with (g_out, u3), (g3, u2), (g2, u1).  Fixed summation
//                order: bit-reproducible.
// HBM traffic is nothing (B x 113 floats); what these kernels remove is launches.
#include "common.h"

namespace dg {
namespace {

constexpr int kW1 = 64, kW2 = 32, kW3 = 16;

enum HeadAct { kHeadRelu = 0, kHeadLeaky = 1 };

template <int ACT>
__device__ __forceinline__ float head_act(float x) {
    return ACT == kHeadRelu ? fmaxf(x, 0.f) : (x > 0.f ? x : 0.01f * x);
}
// derivative through the OUTPUT a = act(x) (a > 0 exactly when x > 0 for both activations)
template <int ACT>
__device__ __forceinline__ float head_dact(float a) {
    return a > 0.f ? 1.f : (ACT == kHeadRelu ? 0.f : 0.01f);
}

struct HeadWeights {
    const float* w2;      // [32,64]
    const float* b2;      // [32] or null
    const float* w3;      // [16,32]
    const float* b3;
    const float* w4;      // [1,16]
    const float* b4;
};

// A workgroup of 256 threads takes 8 rows: every layer is one multiply-add chain per (row, output) thread with the layer's input
// row in LDS (all lanes of a row read the same word: broadcast) and the weight read along the output index (transposed copies
// in the forward, padded to odd strides: conflict-free).
constexpr int kRows = 8;
constexpr int kS2 = kW2 + 1, kS3 = kW3 + 1;      // strides of the transposed weights

template <int ACT, bool MASKED>
__global__ __launch_bounds__(256) void head_chain_kernel(const float* __restrict__ in, const float* __restrict__ m1,
                                                         const float* __restrict__ m2, const float* __restrict__ m3,
                                                         const HeadWeights w, float* __restrict__ o1, float* __restrict__ o2,
                                                         float* __restrict__ o3, float* __restrict__ o4, int64_t R) {
    __shared__ float w2t[kW1 * kS2], w3t[kW2 * kS3], w4s[kW3], b2s[kW2], b3s[kW3];
    __shared__ float a1s[kRows * kW1], a2s[kRows * kW2], a3s[kRows * kW3];
    const int t = threadIdx.x;
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * kRows;
#pragma unroll
    for (int n = 0; n < kW2 * kW1 / 256; ++n) {
        const int idx = t + 256 * n;
        w2t[(idx % kW1) * kS2 + idx / kW1] = w.w2[idx];
    }
#pragma unroll
    for (int n = 0; n < kW3 * kW2 / 256; ++n) {
        const int idx = t + 256 * n;
        w3t[(idx % kW2) * kS3 + idx / kW2] = w.w3[idx];
    }
    if (t < kW3) {
        w4s[t] = w.w4[t];
        b3s[t] = w.b3 ? w.b3[t] : 0.f;
    }
    if (t < kW2) b2s[t] = w.b2 ? w.b2[t] : 0.f;
#pragma unroll
    for (int n = 0; n < kRows * kW1 / 256; ++n) {
        const int e = t + 256 * n;
        const int64_t r = r0 + e / kW1;
        float a = 0.f;
        if (r < R) {
            const float v = in[r * kW1 + e % kW1];
            a = MASKED ? v * head_dact<ACT>(m1[r * kW1 + e % kW1]) : head_act<ACT>(v);
            o1[r * kW1 + e % kW1] = a;
        }
        a1s[e] = a;
    }
    __syncthreads();
    {
        const int row = t / kW2, i = t % kW2;
        float acc[4] = {MASKED ? 0.f : b2s[i], 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < kW1; ++j) acc[j & 3] = fmaf(a1s[row * kW1 + j], w2t[j * kS2 + i], acc[j & 3]);
        const float z = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        const int64_t r = r0 + row;
        float a = 0.f;
        if (r < R) {
            a = MASKED ? z * head_dact<ACT>(m2[r * kW2 + i]) : head_act<ACT>(z);
            o2[r * kW2 + i] = a;
        }
        a2s[t] = a;
    }
    __syncthreads();
    if (t < kRows * kW3) {
        const int row = t / kW3, k = t % kW3;
        float acc[4] = {MASKED ? 0.f : b3s[k], 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < kW2; ++i) acc[i & 3] = fmaf(a2s[row * kW2 + i], w3t[i * kS3 + k], acc[i & 3]);
        const float z = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        const int64_t r = r0 + row;
        float a = 0.f;
        if (r < R) {
            a = MASKED ? z * head_dact<ACT>(m3[r * kW3 + k]) : head_act<ACT>(z);
            o3[r * kW3 + k] = a;
        }
        a3s[t] = a;
    }
    __syncthreads();
    if (t < kRows && r0 + t < R) {
        float out = (MASKED || !w.b4) ? 0.f : w.b4[0];
#pragma unroll
        for (int k = 0; k < kW3; ++k) out = fmaf(a3s[t * kW3 + k], w4s[k], out);
        o4[r0 + t] = out;
    }
}

template <int ACT>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ a1,
                                                       const float* __restrict__ a2, const float* __restrict__ a3,
                                                       const HeadWeights w, float* __restrict__ g3o, float* __restrict__ g2o,
                                                       float* __restrict__ g1o, int64_t R) {
    __shared__ float w2s[kW2 * kW1], w3s[kW3 * kW2], g3s[kRows * kW3], g2s[kRows * kW2];
    const int t = threadIdx.x;
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * kRows;
#pragma unroll
    for (int n = 0; n < kW2 * kW1 / 256; ++n) w2s[t + 256 * n] = w.w2[t + 256 * n];
#pragma unroll
    for (int n = 0; n < kW3 * kW2 / 256; ++n) w3s[t + 256 * n] = w.w3[t + 256 * n];
    if (t < kRows * kW3) {
        const int row = t / kW3, k = t % kW3;
        const int64_t r = r0 + row;
        float g = 0.f;
        if (r < R) {
            g = g_out[r] * w.w4[k] * head_dact<ACT>(a3[r * kW3 + k]);
            g3o[r * kW3 + k] = g;
        }
        g3s[t] = g;
    }
    __syncthreads();
    {
        const int row = t / kW2, i = t % kW2;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < kW3; ++k) acc[k & 3] = fmaf(g3s[row * kW3 + k], w3s[k * kW2 + i], acc[k & 3]);
        const int64_t r = r0 + row;
        float g = 0.f;
        if (r < R) {
            g = ((acc[0] + acc[1]) + (acc[2] + acc[3])) * head_dact<ACT>(a2[r * kW2 + i]);
            g2o[r * kW2 + i] = g;
        }
        g2s[t] = g;
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < kRows * kW1 / 256; ++n) {
        const int e = t + 256 * n;
        const int row = e / kW1, j = e % kW1;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < kW2; ++i) acc[i & 3] = fmaf(g2s[row * kW2 + i], w2s[i * kW1 + j], acc[i & 3]);
        const int64_t r = r0 + row;
        if (r < R) g1o[r * kW1 + j] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) * head_dact<ACT>(a1[r * kW1 + j]);
    }
}

// (dW2 [32,64] | dW3 [16,32] | dW4 [16] | db2 [32] | db3 [16] | db4 [1]) in blocks of 64 result elements; the 1024 threads of
// a workgroup are 16 row groups x 64 elements.  The rows are summed as TWO halves in the same order, half = 8 interleaved
// groups: the discriminator loss runs D(real) and D(fake) as one batch whose upstream gradients are -1/B and +1/B, and where
// the reference's two separate backward passes cancel exactly (e.g. the last bias: sum(-1/B) + sum(+1/B) = 0, and AdamW turns
// any residue into a full-size step) the two halves here do too.
constexpr int kNW2 = kW2 * kW1, kNW3 = kW3 * kW2, kNW = kNW2 + kNW3 + kW3, kNB = kW2 + kW3 + 1;
constexpr int kGroups = 16;

__global__ __launch_bounds__(1024) void head_wgrad_kernel(const float* __restrict__ l4, const float* __restrict__ r4,
                                                          const float* __restrict__ l3, const float* __restrict__ r3,
                                                          const float* __restrict__ l2, const float* __restrict__ r2,
                                                          float* __restrict__ dw4, float* __restrict__ db4, float* __restrict__ dw3,
                                                          float* __restrict__ db3, float* __restrict__ dw2, float* __restrict__ db2,
                                                          int64_t R, int with_bias) {
    __shared__ float part[kGroups][64];
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + e;
    const bool live = idx < kNW + (with_bias ? kNB : 0);
    const float* lp = l4;
    const float* rp = nullptr;
    int ls = 1, rs = 0;
    float* out = dw4;
    if (idx < kNW2) {
        lp = l2 + idx / kW1, ls = kW2, rp = r2 + idx % kW1, rs = kW1, out = dw2 + idx;
    } else if (idx < kNW2 + kNW3) {
        const int t = idx - kNW2;
        lp = l3 + t / kW2, ls = kW3, rp = r3 + t % kW2, rs = kW2, out = dw3 + t;
    } else if (idx < kNW) {
        const int t = idx - kNW2 - kNW3;
        lp = l4, ls = 1, rp = r4 + t, rs = kW3, out = dw4 + t;
    } else if (live) {
        const int t = idx - kNW;
        if (t < kW2) lp = l2 + t, ls = kW2, out = db2 + t;
        else if (t < kW2 + kW3) lp = l3 + (t - kW2), ls = kW3, out = db3 + (t - kW2);
        else lp = l4, ls = 1, out = db4;
    }
    const int64_t H = R / 2;
    const int half = g >> 3, sub = g & 7;
    const int64_t lo = half ? H : 0, hi = half ? R : H;
    float acc = 0.f;
    if (live) {
        if (rp) {
            for (int64_t r = lo + sub; r < hi; r += 8) acc = fmaf(lp[r * ls], rp[r * rs], acc);
        } else {
            for (int64_t r = lo + sub; r < hi; r += 8) acc += lp[r * ls];
        }
    }
    part[g][e] = acc;
    __syncthreads();
    if (g == 0 && live) {
        float h0 = part[0][e], h1 = part[8][e];
#pragma unroll
        for (int s_ = 1; s_ < 8; ++s_) {
            h0 += part[s_][e];
            h1 += part[8 + s_][e];
        }
        *out = h0 + h1;
    }
}

int head_check(const char* who, int64_t R, int act) {
    if (R < 0) return fail(DG_E_SHAPE, "%s: negative row count", who);
    if (act != kHeadRelu && act != kHeadLeaky) return fail(DG_E_ARG, "%s: activation %d (0 relu, 1 leaky relu 0.01)", who, act);
    return 0;
}

}  // namespace
}  // namespace dg

using namespace dg;

/* see include/druggen_hip.h */
extern "C" int dg_head_chain(const float* in, const float* m1, const float* m2, const float* m3, const float* w2, const float* b2,
                             const float* w3, const float* b3, const float* w4, const float* b4, float* o1, float* o2, float* o3,
                             float* o4, int64_t R, int act, dg_stream_t stream_) {
    if (!in || !w2 || !w3 || !w4 || !o1 || !o2 || !o3 || !o4) return fail(DG_E_ARG, "dg_head_chain: null pointer");
    if ((m1 != nullptr) != (m2 != nullptr) || (m1 != nullptr) != (m3 != nullptr))
        return fail(DG_E_ARG, "dg_head_chain: m1, m2, m3 are given together (second order) or not at all (forward)");
    if (int st = head_check("dg_head_chain", R, act)) return st;
    if (R == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const HeadWeights w{w2, m1 ? nullptr : b2, w3, m1 ? nullptr : b3, w4, m1 ? nullptr : b4};
    const dim3 grid(static_cast<unsigned>((R + kRows - 1) / kRows));
#define LAUNCH(A, M) hipLaunchKernelGGL((head_chain_kernel<A, M>), grid, dim3(256), 0, stream, in, m1, m2, m3, w, o1, o2, o3, o4, R)
    if (m1) { if (act == kHeadRelu) LAUNCH(kHeadRelu, true); else LAUNCH(kHeadLeaky, true); }
    else { if (act == kHeadRelu) LAUNCH(kHeadRelu, false); else LAUNCH(kHeadLeaky, false); }
#undef LAUNCH
    return check_launch("dg_head_chain");
}

extern "C" int dg_head_bwd(const float* g_out, const float* a1, const float* a2, const float* a3, const float* w2, const float* w3,
                           const float* w4, float* g3, float* g2, float* g1, int64_t R, int act, dg_stream_t stream_) {
    if (!g_out || !a1 || !a2 || !a3 || !w2 || !w3 || !w4 || !g3 || !g2 || !g1) return fail(DG_E_ARG, "dg_head_bwd: null pointer");
    if (int st = head_check("dg_head_bwd", R, act)) return st;
    if (R == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const HeadWeights w{w2, nullptr, w3, nullptr, w4, nullptr};
    const dim3 grid(static_cast<unsigned>((R + kRows - 1) / kRows));
    if (act == kHeadRelu) hipLaunchKernelGGL((head_bwd_kernel<kHeadRelu>), grid, dim3(256), 0, stream, g_out, a1, a2, a3, w, g3, g2, g1, R);
    else hipLaunchKernelGGL((head_bwd_kernel<kHeadLeaky>), grid, dim3(256), 0, stream, g_out, a1, a2, a3, w, g3, g2, g1, R);
    return check_launch("dg_head_bwd");
}

extern "C" int dg_head_wgrad(const float* l4, const float* r4, const float* l3, const float* r3, const float* l2, const float* r2,
                             float* dw4, float* db4, float* dw3, float* db3, float* dw2, float* db2, int64_t R,
                             dg_stream_t stream_) {
    if (!l4 || !r4 || !l3 || !r3 || !l2 || !r2 || !dw4 || !dw3 || !dw2) return fail(DG_E_ARG, "dg_head_wgrad: null pointer");
    if ((db4 != nullptr) != (db3 != nullptr) || (db4 != nullptr) != (db2 != nullptr))
        return fail(DG_E_ARG, "dg_head_wgrad: the three bias gradients are requested together or not at all");
    if (R < 0) return fail(DG_E_SHAPE, "dg_head_wgrad: negative row count");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int n = kNW + (db4 ? kNB : 0);
    hipLaunchKernelGGL(head_wgrad_kernel, dim3((n + 63) / 64), dim3(1024), 0, stream, l4, r4, l3, r3, l2, r2, dw4, db4, dw3, db3,
                       dw2, db2, R, db4 ? 1 : 0);
    return check_launch("dg_head_wgrad");
}
