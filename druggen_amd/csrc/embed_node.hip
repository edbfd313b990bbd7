// Node embedding of Generator / Discriminator (reference src/model/models.py:52-56, 154-158, applied at :91 / :196):
//     a1 = act(z W1^T + b1) [R,64] ;  a2 = act(a1 W2^T + b2) [R,128]          Linear(E,64) - act - Linear(64,128) - act
// over the R = B N node rows.  On the BLAS + ATen: 4 launches forward, ~8 backward, ~8 in the gradient penalty's second order,
// for 9 k multiply-adds per row.  Two kernels for the piecewise-linear activations (ReLU, LeakyReLU(0.01): act'' = 0):
//   embed_node_chain   both layers for 32 rows per workgroup.  Forward: activations from the pre-activations.  Second order
//                      (`m1`, `m2` = the forward's a1, a2; `in` = t, the adjoint of the first backward's dz):
//                      u1 = (t W1^T) . act'(a1), u2 = (u1 W2^T) . act'(a2) -- the adjoints of g2 W2 and of the upstream gradient.
//   embed_node_bwd     g2 = g . act'(a2), g1 = (g2 W2) . act'(a1), dz = g1 W1 (optional).
// The parameter gradients (g2^T a1, g1^T z and their second-order twins g2^T u1, g1^T t) stay dg_linear_wgrad's.
#include "common.h"

namespace dg {
namespace {

constexpr int kH = 64, kC = 128, kEP = 16;      // hidden width, output width, padded input width
constexpr int kRowsN = 32;
constexpr int kS1 = kH + 1, kS2 = kC + 1;       // strides of the transposed weights in LDS

enum NodeAct { kNodeRelu = 0, kNodeLeaky = 1 };
template <int ACT>
__device__ __forceinline__ float node_act(float x) {
    return ACT == kNodeRelu ? fmaxf(x, 0.f) : (x > 0.f ? x : 0.01f * x);
}
template <int ACT>
__device__ __forceinline__ float node_dact(float a) {      // through the OUTPUT a = act(x)
    return a > 0.f ? 1.f : (ACT == kNodeRelu ? 0.f : 0.01f);
}

template <int ACT, bool MASKED>
__global__ __launch_bounds__(256) void embed_node_chain_kernel(const float* __restrict__ in, const float* __restrict__ m1,
                                                               const float* __restrict__ m2, const float* __restrict__ w1,
                                                               const float* __restrict__ b1, const float* __restrict__ w2,
                                                               const float* __restrict__ b2, float* __restrict__ o1,
                                                               float* __restrict__ o2, int64_t R, int E) {
    __shared__ float w1t[kEP * kS1];            // [e][j]
    __shared__ float w2t[kH * kS2];             // [j][i]
    __shared__ float zs[kRowsN * kEP];
    __shared__ __attribute__((aligned(16))) float a1s[kRowsN * kH];
    const int t = threadIdx.x;
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * kRowsN;
    for (int idx = t; idx < kH * kEP; idx += 256) {
        const int j = idx / kEP, e = idx % kEP;
        w1t[e * kS1 + j] = e < E ? w1[j * E + e] : 0.f;
    }
    {
        float4 wv[kC * kH / 1024];      // all eight 16-byte loads of the thread in flight together
#pragma unroll
        for (int n = 0; n < kC * kH / 1024; ++n) wv[n] = ld4(w2 + 4 * (t + 256 * n));      // w2 [128][64]
#pragma unroll
        for (int n = 0; n < kC * kH / 1024; ++n) {
            const int idx = 4 * (t + 256 * n);
            const int i = idx / kH, j = idx % kH;
            w2t[j * kS2 + i] = wv[n].x;
            w2t[(j + 1) * kS2 + i] = wv[n].y;
            w2t[(j + 2) * kS2 + i] = wv[n].z;
            w2t[(j + 3) * kS2 + i] = wv[n].w;
        }
    }
    for (int idx = t; idx < kRowsN * kEP; idx += 256) {
        const int64_t r = r0 + idx / kEP;
        const int e = idx % kEP;
        zs[idx] = (r < R && e < E) ? in[r * E + e] : 0.f;
    }
    __syncthreads();
    // layer 1: thread (j = t % 64, row group t / 64) -> rows rg, rg + 4, ...
    {
        const int j = t % kH, rg = t / kH;
        float wj[kEP];
#pragma unroll
        for (int e = 0; e < kEP; ++e) wj[e] = w1t[e * kS1 + j];
        const float bj = (MASKED || !b1) ? 0.f : b1[j];
#pragma unroll
        for (int n = 0; n < kRowsN / 4; ++n) {
            const int row = rg + 4 * n;
            float acc = bj;
#pragma unroll
            for (int e = 0; e < kEP; ++e) acc = fmaf(zs[row * kEP + e], wj[e], acc);
            const int64_t r = r0 + row;
            float a = 0.f;
            if (r < R) {
                a = MASKED ? acc * node_dact<ACT>(m1[r * kH + j]) : node_act<ACT>(acc);
                o1[r * kH + j] = a;
            }
            a1s[row * kH + j] = a;
        }
    }
    __syncthreads();
    // layer 2: thread (i = t % 128, row half t / 128) -> rows rh, rh + 2, ...: 16 accumulators, a1 rows as float4 broadcasts
    {
        const int i = t % kC, rh = t / kC;
        float acc[kRowsN / 2];
        const float bi = (MASKED || !b2) ? 0.f : b2[i];
#pragma unroll
        for (int n = 0; n < kRowsN / 2; ++n) acc[n] = bi;
#pragma unroll 4
        for (int j = 0; j < kH; j += 4) {
            const float wa = w2t[j * kS2 + i], wb = w2t[(j + 1) * kS2 + i], wc = w2t[(j + 2) * kS2 + i], wd = w2t[(j + 3) * kS2 + i];
#pragma unroll
            for (int n = 0; n < kRowsN / 2; ++n) {
                const float4 a = *reinterpret_cast<const float4*>(&a1s[(rh + 2 * n) * kH + j]);
                acc[n] = fmaf(a.x, wa, acc[n]);
                acc[n] = fmaf(a.y, wb, acc[n]);
                acc[n] = fmaf(a.z, wc, acc[n]);
                acc[n] = fmaf(a.w, wd, acc[n]);
            }
        }
#pragma unroll
        for (int n = 0; n < kRowsN / 2; ++n) {
            const int64_t r = r0 + rh + 2 * n;
            if (r < R) o2[r * kC + i] = MASKED ? acc[n] * node_dact<ACT>(m2[r * kC + i]) : node_act<ACT>(acc[n]);
        }
    }
}

template <int ACT>
__global__ __launch_bounds__(256) void embed_node_bwd_kernel(const float* __restrict__ g, const float* __restrict__ a1,
                                                             const float* __restrict__ a2, const float* __restrict__ w1,
                                                             const float* __restrict__ w2, float* __restrict__ g2o,
                                                             float* __restrict__ g1o, float* __restrict__ dzo, int64_t R, int E) {
    __shared__ __attribute__((aligned(16))) float w2s[kC * kH];              // [i][j] (natural)
    __shared__ float w1s[kH * kEP];             // [j][e]
    __shared__ __attribute__((aligned(16))) float g2s[kRowsN * kC];
    __shared__ __attribute__((aligned(16))) float g1s[kRowsN * kH];
    const int t = threadIdx.x;
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * kRowsN;
    {
        float4 wv[kC * kH / 1024];
#pragma unroll
        for (int n = 0; n < kC * kH / 1024; ++n) wv[n] = ld4(w2 + 4 * (t + 256 * n));
#pragma unroll
        for (int n = 0; n < kC * kH / 1024; ++n) *reinterpret_cast<float4*>(&w2s[4 * (t + 256 * n)]) = wv[n];
    }
    for (int idx = t; idx < kH * kEP; idx += 256) {
        const int j = idx / kEP, e = idx % kEP;
        w1s[idx] = e < E ? w1[j * E + e] : 0.f;
    }
    {
        float4 gv[kRowsN * kC / 1024], av[kRowsN * kC / 1024];      // 4 + 4 loads in flight
#pragma unroll
        for (int n = 0; n < kRowsN * kC / 1024; ++n) {
            const int idx = 4 * (t + 256 * n);
            const int64_t r = r0 + idx / kC;
            gv[n] = r < R ? ld4(g + r * kC + idx % kC) : f4(0.f);
            av[n] = r < R ? ld4(a2 + r * kC + idx % kC) : f4(0.f);
        }
#pragma unroll
        for (int n = 0; n < kRowsN * kC / 1024; ++n) {
            const int idx = 4 * (t + 256 * n);
            const int64_t r = r0 + idx / kC;
            const float4 v = make_float4(gv[n].x * node_dact<ACT>(av[n].x), gv[n].y * node_dact<ACT>(av[n].y),
                                         gv[n].z * node_dact<ACT>(av[n].z), gv[n].w * node_dact<ACT>(av[n].w));
            if (r < R) st4(g2o + r * kC + idx % kC, v);
            *reinterpret_cast<float4*>(&g2s[idx]) = r < R ? v : f4(0.f);
        }
    }
    __syncthreads();
    // g1[row][j] = (sum_i g2[row][i] w2[i][j]) act'(a1): thread (j = t % 64, row group t / 64), 8 rows each
    {
        const int j = t % kH, rg = t / kH;
        float acc[kRowsN / 4];
#pragma unroll
        for (int n = 0; n < kRowsN / 4; ++n) acc[n] = 0.f;
#pragma unroll 4
        for (int i = 0; i < kC; i += 4) {
            const float wa = w2s[i * kH + j], wb = w2s[(i + 1) * kH + j], wc = w2s[(i + 2) * kH + j], wd = w2s[(i + 3) * kH + j];
#pragma unroll
            for (int n = 0; n < kRowsN / 4; ++n) {
                const float4 v = *reinterpret_cast<const float4*>(&g2s[(rg + 4 * n) * kC + i]);
                acc[n] = fmaf(v.x, wa, acc[n]);
                acc[n] = fmaf(v.y, wb, acc[n]);
                acc[n] = fmaf(v.z, wc, acc[n]);
                acc[n] = fmaf(v.w, wd, acc[n]);
            }
        }
#pragma unroll
        for (int n = 0; n < kRowsN / 4; ++n) {
            const int row = rg + 4 * n;
            const int64_t r = r0 + row;
            float v = 0.f;
            if (r < R) {
                v = acc[n] * node_dact<ACT>(a1[r * kH + j]);
                g1o[r * kH + j] = v;
            }
            g1s[row * kH + j] = v;
        }
    }
    if (!dzo) return;      // (uniform)
    __syncthreads();
    // dz[row][e] = sum_j g1[row][j] w1[j][e]: 32 x 16 (row, e) pairs
#pragma unroll
    for (int n = 0; n < kRowsN * kEP / 256; ++n) {
        const int idx = t + 256 * n;
        const int row = idx / kEP, e = idx % kEP;
        float acc = 0.f;
#pragma unroll 8
        for (int j = 0; j < kH; ++j) acc = fmaf(g1s[row * kH + j], w1s[j * kEP + e], acc);
        const int64_t r = r0 + row;
        if (r < R && e < E) dzo[r * E + e] = acc;
    }
}

int node_check(const char* who, int64_t R, int E, int act) {
    if (R < 0 || E < 1 || E > kEP) return fail(DG_E_SHAPE, "%s: unsupported shape R=%lld E=%d (1 <= E <= 16)", who, static_cast<long long>(R), E);
    if (act != kNodeRelu && act != kNodeLeaky) return fail(DG_E_ARG, "%s: activation %d (0 relu, 1 leaky relu 0.01)", who, act);
    return 0;
}

}  // namespace
}  // namespace dg

using namespace dg;

/* see include/druggen_hip.h */
extern "C" int dg_embed_node_chain(const float* in, const float* m1, const float* m2, const float* w1, const float* b1,
                                   const float* w2, const float* b2, float* o1, float* o2, int64_t R, int E, int act,
                                   dg_stream_t stream_) {
    if (!in || !w1 || !w2 || !o1 || !o2) return fail(DG_E_ARG, "dg_embed_node_chain: null pointer");
    if ((m1 != nullptr) != (m2 != nullptr))
        return fail(DG_E_ARG, "dg_embed_node_chain: m1 and m2 are given together (second order) or not at all (forward)");
    if (int st = node_check("dg_embed_node_chain", R, E, act)) return st;
    if (R == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const dim3 grid(static_cast<unsigned>((R + kRowsN - 1) / kRowsN));
#define LAUNCH(A, M) hipLaunchKernelGGL((embed_node_chain_kernel<A, M>), grid, dim3(256), 0, stream, in, m1, m2, w1, b1, w2, b2, o1, o2, R, E)
    if (m1) { if (act == kNodeRelu) LAUNCH(kNodeRelu, true); else LAUNCH(kNodeLeaky, true); }
    else { if (act == kNodeRelu) LAUNCH(kNodeRelu, false); else LAUNCH(kNodeLeaky, false); }
#undef LAUNCH
    return check_launch("dg_embed_node_chain");
}

extern "C" int dg_embed_node_bwd(const float* g, const float* a1, const float* a2, const float* w1, const float* w2, float* g2,
                                 float* g1, float* dz, int64_t R, int E, int act, dg_stream_t stream_) {
    if (!g || !a1 || !a2 || !w1 || !w2 || !g2 || !g1) return fail(DG_E_ARG, "dg_embed_node_bwd: null pointer");
    if (int st = node_check("dg_embed_node_bwd", R, E, act)) return st;
    if (R == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const dim3 grid(static_cast<unsigned>((R + kRowsN - 1) / kRowsN));
    if (act == kNodeRelu) hipLaunchKernelGGL((embed_node_bwd_kernel<kNodeRelu>), grid, dim3(256), 0, stream, g, a1, a2, w1, w2, g2, g1, dz, R, E);
    else hipLaunchKernelGGL((embed_node_bwd_kernel<kNodeLeaky>), grid, dim3(256), 0, stream, g, a1, a2, w1, w2, g2, g1, dz, R, E);
    return check_launch("dg_embed_node_bwd");
}
