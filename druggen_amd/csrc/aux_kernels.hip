// The steps either side of the GAN-step hot path:
//   * batch densify  -- reference src/data/utils.py:128-137 (PyG to_dense_adj + label2onehot)
//   * AdamW update   -- reference train.py:213-214,368,384 (torch.optim.AdamW defaults)
//   * argmax decode  -- reference inference.py:197-198 (torch.max(.., -1)[1])
// All HBM/latency-bound byte and elementwise work: one coalesced pass each.
#include "bf16.h"

namespace dg {
namespace {

// ---- densify --------------------------------------------------------------------------------
// labels[b,i,j] += attr for every COO edge (u -> v), b = u / N, i = u % N, j = v % N: the reference
// pads every graph to N = max_num_nodes nodes (utils.py:133), to_dense_adj scatter-ADDs duplicates and
// indexes (batch[u], u - ptr[batch[u]], v - ptr[batch[v]]): an edge that leaves its graph is written into the
// SOURCE graph's matrix at column v mod N (oracle/aux_oracle.py::to_dense_adj, tests/test_hip_aux.py).
__global__ void densify_scatter_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                       const int64_t* __restrict__ attr, int64_t n_edges, int B, int N,
                                       int* __restrict__ labels) {
    const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const int64_t u = src[e], v = dst[e];
    const int64_t b = u / N;
    if (u < 0 || b >= B || v < 0 || v >= static_cast<int64_t>(B) * N) return;   // node ids outside the batch
    atomicAdd(labels + (b * N + u % N) * N + v % N, static_cast<int>(attr[e]));
}

// one-hot expand: a[r, c] = (labels[r] == c); labels outside [0, E) give an all-zero row, which is
// what scatter_ on an out-of-range index would refuse -- reported through `bad`.
__global__ void onehot_kernel(const int* __restrict__ labels, int64_t rows, int E, float* __restrict__ a,
                              int* __restrict__ bad) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= rows * E) return;
    const int64_t r = idx / E;
    const int c = static_cast<int>(idx - r * E);
    const int lab = labels[r];
    if (c == 0 && (lab < 0 || lab >= E)) atomicAdd(bad, 1);
    a[idx] = lab == c ? 1.0f : 0.0f;
}

// ---- AdamW over a flat buffer ---------------------------------------------------------------
// torch.optim.AdamW single-tensor formulas: p *= 1 - lr*wd;  m = b1 m + (1-b1) g;
// v = b2 v + (1-b2) g^2;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, float bc1, float bc2_sqrt) {
    const int64_t i4 = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    const float step = lr / bc1;
    if (i4 + 3 < n) {
        float4 pp = ld4(p + i4), gg = ld4(g + i4), mm = ld4(m + i4), vv = ld4(v + i4);
        float* pe = &pp.x; float* ge = &gg.x; float* me = &mm.x; float* ve = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pe[k] *= 1.0f - lr * weight_decay;
            me[k] = beta1 * me[k] + (1.0f - beta1) * ge[k];
            ve[k] = beta2 * ve[k] + (1.0f - beta2) * ge[k] * ge[k];
            pe[k] -= step * me[k] / (sqrtf(ve[k]) / bc2_sqrt + eps);
        }
        st4(p + i4, pp);
        st4(m + i4, mm);
        st4(v + i4, vv);
    } else {
        for (int64_t i = i4; i < n; ++i) {
            float pp = p[i] * (1.0f - lr * weight_decay);
            const float gg = g[i];
            const float mm = beta1 * m[i] + (1.0f - beta1) * gg;
            const float vv = beta2 * v[i] + (1.0f - beta2) * gg * gg;
            pp -= step * mm / (sqrtf(vv) / bc2_sqrt + eps);
            p[i] = pp;
            m[i] = mm;
            v[i] = vv;
        }
    }
}

// Graph-replay-safe variant: the step count lives in device memory, so a captured launch computes
// fresh bias corrections on every replay (host-side scalars would be frozen into the graph).
__global__ void adamw_bump_kernel(int* step) { *step += 1; }

__global__ void adamw_devstep_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                     float* __restrict__ v, int64_t n, float lr, float beta1, float beta2, float eps,
                                     float weight_decay, const int* __restrict__ step) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float t = static_cast<float>(*step);
    const float bc1 = 1.0f - powf(beta1, t);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, t));
    float pp = p[i] * (1.0f - lr * weight_decay);
    const float gg = g[i];
    const float mm = beta1 * m[i] + (1.0f - beta1) * gg;
    const float vv = beta2 * v[i] + (1.0f - beta2) * gg * gg;
    pp -= (lr / bc1) * mm / (sqrtf(vv) / bc2_sqrt + eps);
    p[i] = pp;
    m[i] = mm;
    v[i] = vv;
}

// ---- argmax decode --------------------------------------------------------------------------
// out[r] = index of the first maximum of logits[r, 0..E) (NaN counts as maximal, like torch.max)
__global__ void argmax_kernel(const float* __restrict__ logits, int64_t rows, int E, unsigned char* __restrict__ out) {
    const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* x = logits + r * E;
    float best = x[0];
    int arg = 0;
    for (int c = 1; c < E; ++c) {
        const float v = x[c];
        if (v > best || (v != v && best == best)) {
            best = v;
            arg = c;
        }
    }
    out[r] = static_cast<unsigned char>(arg);
}

// ---- skinny readout: nn.Linear(128 -> N <= 16) over edge / node rows (reference models.py:67-68,100-101) ----
// y[r][n] = sum_k x[r][k] w[n][k] + b[n]: a coalesced stream over x (32 lanes x 4 columns = one 128-wide row, two rows
// per wave instruction), N x 4 multiply-adds per lane against weights held in registers, a 32-lane DPP sum per output.
// x may be float32 or bfloat16 (the activation dtype); the logits are float32 in every mode (no `.float()` copy).
__device__ __forceinline__ float half_wave_sum32(float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));   // row_half_mirror
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, true));   // row_mirror
    const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0));
    const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
    const float lo2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32));
    const float hi2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
    return (threadIdx.x & 32) ? lo2 + hi2 : lo + hi;
}
template <typename T, int NMAX>
__global__ __launch_bounds__(256) void skinny_linear_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, float* __restrict__ y,
                                                              int64_t R, int N) {
    const int lane = threadIdx.x & 63, half = lane >> 5, c4 = lane & 31;
    float4 wr[NMAX];
    float br[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
        wr[n] = n < N ? ld4(w + n * 128 + 4 * c4) : f4(0.f);
        br[n] = (n < N && b) ? b[n] : 0.f;
    }
    const int64_t wave = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
    for (int64_t r0 = wave * 2; r0 < R; r0 += nwaves * 2) {
        const int64_t row = r0 + half;
        const bool ok = row < R;
        const float4 xv = ok ? ld4_stream(x + row * 128 + 4 * c4) : f4(0.f);
        float out = 0.f;      // lane n of each half keeps output n
#pragma unroll
        for (int n = 0; n < NMAX; ++n) {
            const float s = half_wave_sum32((xv.x * wr[n].x + xv.y * wr[n].y) + (xv.z * wr[n].z + xv.w * wr[n].w));
            if (c4 == n) out = s + br[n];
        }
        if (ok && c4 < N) y[row * N + c4] = out;
    }
}
// dx[r][k] = sum_n dy[r][n] w[n][k]  (the readout's input gradient; dy float32, dx in the activation dtype)
template <typename T, int NMAX>
__global__ __launch_bounds__(256) void skinny_linear_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                T* __restrict__ dx, int64_t R, int N) {
    const int lane = threadIdx.x & 63, half = lane >> 5, c4 = lane & 31;
    float4 wr[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) wr[n] = n < N ? ld4(w + n * 128 + 4 * c4) : f4(0.f);
    const int64_t wave = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
    for (int64_t r0 = wave * 2; r0 < R; r0 += nwaves * 2) {
        const int64_t row = r0 + half;
        if (row >= R) continue;
        float4 acc = f4(0.f);
#pragma unroll
        for (int n = 0; n < NMAX; ++n)
            if (n < N) acc = fma4(f4(dy[row * N + n]), wr[n], acc);      // one address per half-wave: broadcast load
        st4_stream(dx + row * 128 + 4 * c4, acc);
    }
}

}  // namespace
}  // namespace dg

using namespace dg;

extern "C" int dg_skinny_linear_fwd(const void* x, const float* w, const float* b, float* y, int64_t R, int N, int K,
                                    int dtype, dg_stream_t stream_) {
    if (!x || !w || !y) return fail(DG_E_ARG, "dg_skinny_linear_fwd: null pointer");
    if (!dtype_ok(dtype)) return fail(DG_E_ARG, "dg_skinny_linear_fwd: unknown dtype %d", dtype);
    if (R < 0 || K != 128 || N < 1 || N > 16) return fail(DG_E_SHAPE, "dg_skinny_linear_fwd: unsupported N=%d K=%d (N <= 16, K == 128)", N, K);
    if (R == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t blocks = (R / 2 + 3) / 4 + 1;
    const int grid = static_cast<int>(blocks < 2048 ? blocks : 2048);
#define LAUNCH(T, NM) hipLaunchKernelGGL((skinny_linear_fwd_kernel<T, NM>), dim3(grid), dim3(256), 0, stream, static_cast<const T*>(x), w, b, y, R, N)
    if (dtype == DG_DTYPE_BF16) { if (N <= 8) LAUNCH(bf16_t, 8); else LAUNCH(bf16_t, 16); }
    else { if (N <= 8) LAUNCH(float, 8); else LAUNCH(float, 16); }
#undef LAUNCH
    return check_launch("dg_skinny_linear_fwd");
}

extern "C" int dg_skinny_linear_dgrad(const float* dy, const float* w, void* dx, int64_t R, int N, int K, int dtype,
                                      dg_stream_t stream_) {
    if (!dy || !w || !dx) return fail(DG_E_ARG, "dg_skinny_linear_dgrad: null pointer");
    if (!dtype_ok(dtype)) return fail(DG_E_ARG, "dg_skinny_linear_dgrad: unknown dtype %d", dtype);
    if (R < 0 || K != 128 || N < 1 || N > 16) return fail(DG_E_SHAPE, "dg_skinny_linear_dgrad: unsupported N=%d K=%d (N <= 16, K == 128)", N, K);
    if (R == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t blocks = (R / 2 + 3) / 4 + 1;
    const int grid = static_cast<int>(blocks < 2048 ? blocks : 2048);
#define LAUNCH(T, NM) hipLaunchKernelGGL((skinny_linear_dgrad_kernel<T, NM>), dim3(grid), dim3(256), 0, stream, dy, w, static_cast<T*>(dx), R, N)
    if (dtype == DG_DTYPE_BF16) { if (N <= 8) LAUNCH(bf16_t, 8); else LAUNCH(bf16_t, 16); }
    else { if (N <= 8) LAUNCH(float, 8); else LAUNCH(float, 16); }
#undef LAUNCH
    return check_launch("dg_skinny_linear_dgrad");
}


extern "C" int dg_densify(const int64_t* edge_src, const int64_t* edge_dst, const int64_t* edge_attr, int64_t n_edges,
                          int B, int N, int E, int* labels, float* a, int* bad_count, dg_stream_t stream_) {
    if ((n_edges > 0 && (!edge_src || !edge_dst || !edge_attr)) || !labels || !a || !bad_count)
        return fail(DG_E_ARG, "dg_densify: null pointer");
    if (B < 0 || N < 1 || E < 1 || n_edges < 0) return fail(DG_E_SHAPE, "dg_densify: bad sizes");
    if (B == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t rows = static_cast<int64_t>(B) * N * N;
    hipError_t err = hipMemsetAsync(labels, 0, rows * sizeof(int), stream);
    if (err == hipSuccess) err = hipMemsetAsync(bad_count, 0, sizeof(int), stream);
    if (err != hipSuccess) return fail(static_cast<int>(err), "dg_densify: %s", hipGetErrorString(err));
    if (n_edges > 0)
        hipLaunchKernelGGL(densify_scatter_kernel, dim3(static_cast<unsigned>((n_edges + 255) / 256)), dim3(256), 0,
                           stream, edge_src, edge_dst, edge_attr, n_edges, B, N, labels);
    hipLaunchKernelGGL(onehot_kernel, dim3(static_cast<unsigned>((rows * E + 255) / 256)), dim3(256), 0, stream, labels,
                       rows, E, a, bad_count);
    return check_launch("dg_densify");
}

extern "C" int dg_adamw_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int64_t step,
                             dg_stream_t stream_) {
    if (!param || !grad || !exp_avg || !exp_avg_sq) return fail(DG_E_ARG, "dg_adamw_flat: null pointer");
    if (n < 0 || step < 1) return fail(DG_E_ARG, "dg_adamw_flat: n >= 0 and step >= 1 required");
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow(static_cast<double>(beta1), static_cast<double>(step));
    const double bc2 = 1.0 - pow(static_cast<double>(beta2), static_cast<double>(step));
    const int64_t threads = (n + 3) / 4;
    hipLaunchKernelGGL(adamw_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                       weight_decay, static_cast<float>(bc1), static_cast<float>(sqrt(bc2)));
    return check_launch("dg_adamw_flat");
}

extern "C" int dg_adamw_flat_devstep(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                     float lr, float beta1, float beta2, float eps, float weight_decay,
                                     int* step_counter, dg_stream_t stream_) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step_counter)
        return fail(DG_E_ARG, "dg_adamw_flat_devstep: null pointer");
    if (n < 0) return fail(DG_E_ARG, "dg_adamw_flat_devstep: n >= 0 required");
    if (n == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(adamw_bump_kernel, dim3(1), dim3(1), 0, stream, step_counter);
    hipLaunchKernelGGL(adamw_devstep_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, param,
                       grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step_counter);
    return check_launch("dg_adamw_flat_devstep");
}

extern "C" int dg_argmax_decode(const float* logits, int64_t rows, int E, unsigned char* out, dg_stream_t stream_) {
    if (!logits || !out) return fail(DG_E_ARG, "dg_argmax_decode: null pointer");
    if (rows < 0 || E < 1 || E > 255) return fail(DG_E_SHAPE, "dg_argmax_decode: need 1 <= classes <= 255");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(argmax_kernel, dim3(static_cast<unsigned>((rows + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), logits, rows, E, out);
    return check_launch("dg_argmax_decode");
}
