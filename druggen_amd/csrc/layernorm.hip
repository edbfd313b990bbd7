// Residual + LayerNorm of DrugGEN's Encoder_Block (reference
// src/model/layers.py:185-192):  y = LN(a + r) * gamma + beta over the last dim.
//
// HBM-bound row kernels.  A row of C floats is owned by a group of G lanes
// (G = 8/16/32/64, each lane holds QPL float4), so a wave covers 64/G rows per
// pass with 16-byte loads; mean/variance and the backward row sums are
// butterflies over the group's lanes.  The column sums for dgamma / dbeta are
// reduced in a fixed order (per-block partials in the workspace, then one
// finishing kernel), so results are bit-reproducible.
#include "bf16.h"
#include "traversal.h"

namespace dg {
namespace {

// sum over the G lanes of a row group (G is a power of two, 8 <= G <= 64), result in every lane: DPP inside the 16-lane rows
// (quad_perm x 2, row_half_mirror, row_mirror), v_permlane16_swap / v_permlane32_swap across them -- no LDS round trip (a
// __shfl_xor butterfly was 3 - 6 dependent ds_bpermute per sum, four sums per row in the second-order kernel)
template <int CTRL>
__device__ __forceinline__ float ln_dpp_add(float x) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true);
    return x + __int_as_float(moved);
}
template <int G>
__device__ __forceinline__ float group_sum(float x) {
    static_assert(G == 8 || G == 16 || G == 32 || G == 64, "row group");
    x = ln_dpp_add<0xB1>(x);       // quad_perm [1,0,3,2]
    x = ln_dpp_add<0x4E>(x);       // quad_perm [2,3,0,1]
    x = ln_dpp_add<0x141>(x);      // row_half_mirror: 8-lane totals
    if (G >= 16) x = ln_dpp_add<0x140>(x);      // row_mirror: 16-lane totals
    if (G >= 32) x = xor_step<false>(x, 16);
    if (G >= 64) x = xor_step<false>(x, 32);
    return x;
}

__device__ __forceinline__ float hsum(float4 a) { return (a.x + a.y) + (a.z + a.w); }

// The QPL four-channel slots of a lane.  PAIR (bf16 rows, QPL = 2): the lane's two slots are ADJACENT -- eight consecutive
// channels = one 16-byte access (an 8-byte bf16 access per lane moved half of what the memory pipe takes per instruction:
// 4.2 TB/s of the three passes of an edge-level LayerNorm backward at B = 2048).
typedef unsigned ln_u32x4 __attribute__((ext_vector_type(4)));
template <bool PAIR, typename T, int QPL>
__device__ __forceinline__ void ld_slots(const T* p, const int (&coff)[QPL], float4 (&out)[QPL]) {
    if constexpr (PAIR) {
        static_assert(QPL == 2, "paired slots");
        const ln_u32x4 w = *reinterpret_cast<const ln_u32x4*>(p + coff[0]);
        out[0] = unpack4_bf16(u32x2_t{w[0], w[1]});
        out[1] = unpack4_bf16(u32x2_t{w[2], w[3]});
    } else {
#pragma unroll
        for (int t = 0; t < QPL; ++t) out[t] = ld4(p + coff[t]);
    }
}
template <bool PAIR, typename T, int QPL>
__device__ __forceinline__ void st_slots(T* p, const int (&coff)[QPL], const bool (&cok)[QPL], const float4 (&v)[QPL]) {
    if constexpr (PAIR) {
        const u32x2_t a = pack4_bf16(v[0]), b = pack4_bf16(v[1]);
        *reinterpret_cast<ln_u32x4*>(p + coff[0]) = ln_u32x4{a[0], a[1], b[0], b[1]};
    } else {
#pragma unroll
        for (int t = 0; t < QPL; ++t)
            if (cok[t]) st4(p + coff[t], v[t]);
    }
}

constexpr int kBlock = 256;

struct RowMap {
    int64_t row;
    int lane_in_group;
    bool ok;
};

template <int G>
__device__ __forceinline__ RowMap map_row(int64_t pass, int64_t R) {
    constexpr int RPB = kBlock / G;  // rows per block pass
    RowMap m;
    m.lane_in_group = threadIdx.x % G;
    m.row = pass * RPB + threadIdx.x / G;
    m.ok = m.row < R;
    if (!m.ok) m.row = R - 1;  // clamp: loads stay in bounds, stores are masked
    return m;
}

// ------------------------------------------------------------------ forward --
template <typename T, int G, int QPL>
__global__ __launch_bounds__(kBlock) void ln_fwd_kernel(const T* __restrict__ a, const T* __restrict__ r,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, T* __restrict__ y,
                                                      float* __restrict__ mean, float* __restrict__ rstd, int64_t R,
                                                      int C, float eps) {
    constexpr int RPB = kBlock / G;
    const int64_t passes = (R + RPB - 1) / RPB;
    const float invC = 1.0f / static_cast<float>(C);
    for (int64_t pass = blockIdx.x; pass < passes; pass += gridDim.x) {
        const RowMap m = map_row<G>(pass, R);
        float4 z[QPL];
        bool cok[QPL];
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < QPL; ++t) {
            const int c = (m.lane_in_group + t * G) * 4;
            cok[t] = c < C;
            const int cc = cok[t] ? c : 0;
            z[t] = ld4(a + m.row * C + cc);
            if (r) z[t] += ld4(r + m.row * C + cc);
            if (!cok[t]) z[t] = f4(0.f);
            s += hsum(z[t]);
        }
        const float mu = group_sum<G>(s) * invC;
        float v = 0.f;
#pragma unroll
        for (int t = 0; t < QPL; ++t) {
            float4 d = z[t] - f4(mu);
            if (!cok[t]) d = f4(0.f);
            v += hsum(d * d);
            z[t] = d;
        }
        const float rs = rsqrtf(group_sum<G>(v) * invC + eps);
#pragma unroll
        for (int t = 0; t < QPL; ++t) {
            const int c = (m.lane_in_group + t * G) * 4;
            if (cok[t] && m.ok) st4(y + m.row * C + c, fma4(rs * z[t], ld4(gamma + c), ld4(beta + c)));
        }
        if (m.ok && m.lane_in_group == 0) {
            mean[m.row] = mu;
            rstd[m.row] = rs;
        }
    }
}

// ----------------------------------------------------------------- backward --
// dz = rstd (u - mean(u) - xhat mean(u xhat)), u = gamma dy
// dgamma = sum_rows dy xhat, dbeta = sum_rows dy  (block partials -> part[])
template <typename T, int G, int QPL, bool PAIR = false>
__global__ __launch_bounds__(kBlock) void ln_bwd_kernel(const T* __restrict__ a, const T* __restrict__ r,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const T* __restrict__ dy, T* __restrict__ dz,
                                                      float* __restrict__ part, int64_t R, int C,
                                                      const T* __restrict__ dz_add) {
    constexpr int RPB = kBlock / G;
    __shared__ float4 red[2][QPL][kBlock];
    const int64_t passes = (R + RPB - 1) / RPB;
    const float invC = 1.0f / static_cast<float>(C);
    float4 gacc[QPL], bacc[QPL], gam[QPL];
    bool cok[QPL];
    int coff[QPL];
    const int lig = threadIdx.x % G;
#pragma unroll
    for (int t = 0; t < QPL; ++t) {
        const int c = PAIR ? (lig * 2 + t) * 4 : (lig + t * G) * 4;
        cok[t] = c < C;
        coff[t] = cok[t] ? c : 0;
        gam[t] = ld4(gamma + coff[t]);
        gacc[t] = bacc[t] = f4(0.f);
    }
    for (int64_t pass = blockIdx.x; pass < passes; pass += gridDim.x) {
        const RowMap m = map_row<G>(pass, R);
        const float mu = mean[m.row], rs = rstd[m.row];
        float4 xh[QPL], u[QPL], zin[QPL], rin[QPL], gin[QPL];
        ld_slots<PAIR>(a + m.row * C, coff, zin);
        if (r) ld_slots<PAIR>(r + m.row * C, coff, rin);
        ld_slots<PAIR>(dy + m.row * C, coff, gin);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < QPL; ++t) {
            float4 z = zin[t];
            if (r) z += rin[t];
            float4 g = gin[t];
            if (!cok[t] || !m.ok) g = f4(0.f);
            xh[t] = rs * (z - f4(mu));
            if (!cok[t]) xh[t] = f4(0.f);
            gacc[t] = fma4(g, xh[t], gacc[t]);
            bacc[t] += g;
            u[t] = g * gam[t];
            s1 += hsum(u[t]);
            s2 += hsum(u[t] * xh[t]);
        }
        const float c1 = group_sum<G>(s1) * invC, c2 = group_sum<G>(s2) * invC;
        float4 out[QPL], extra[QPL];
        if (dz_add) ld_slots<PAIR>(dz_add + m.row * C, coff, extra);   // second gradient source of the pre-LN sum
#pragma unroll
        for (int t = 0; t < QPL; ++t) {
            out[t] = rs * (u[t] - f4(c1) - c2 * xh[t]);
            if (dz_add) {
                // rounded product, THEN the add: the sum must carry the bits of "this kernel without dz_add, then one elementwise
                // add" (what the autograd engine does with two gradient sources; tests/test_hip_step.py holds the two paths to
                // bit equality) -- fused into an fma by the compiler it differed in the last place
                asm volatile("" : "+v"(out[t].x), "+v"(out[t].y), "+v"(out[t].z), "+v"(out[t].w));
                out[t] += extra[t];
            }
        }
        if (m.ok) st_slots<PAIR>(dz + m.row * C, coff, cok, out);
    }
    // block partial: sum over the RPB row groups in a fixed order
#pragma unroll
    for (int t = 0; t < QPL; ++t) {
        red[0][t][threadIdx.x] = gacc[t];
        red[1][t][threadIdx.x] = bacc[t];
    }
    __syncthreads();
    if (threadIdx.x < G) {
#pragma unroll
        for (int t = 0; t < QPL; ++t) {
            float4 gs = f4(0.f), bs = f4(0.f);
            for (int g = 0; g < RPB; ++g) {
                gs += red[0][t][g * G + threadIdx.x];
                bs += red[1][t][g * G + threadIdx.x];
            }
            if (cok[t]) {
                st4(part + (static_cast<size_t>(blockIdx.x) * 2 + 0) * C + coff[t], gs);
                st4(part + (static_cast<size_t>(blockIdx.x) * 2 + 1) * C + coff[t], bs);
            }
        }
    }
}

// out[k][c] = sum_blocks part[block][k][c]  (k < K).  One block per (k, 32-column chunk):
// 32 row groups stride over the partials, then a fixed-order LDS sum (bit-reproducible).
__global__ __launch_bounds__(1024) void ln_finish_kernel(const float* __restrict__ part, int nblocks, int K, int C,
                                                       float* __restrict__ out0, float* __restrict__ out1) {
    __shared__ float red[32][33];
    const int k = blockIdx.y;
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    const int g = threadIdx.x >> 5;
    float s = 0.f;
    if (c < C) {
        // eight independent loads in flight per thread (the launch is all latency: 1 024 partials = 32 per thread), summed in
        // a fixed order
        const float* src = part + static_cast<size_t>(k) * C + c;
        const size_t step = static_cast<size_t>(K) * C;
        int b = g;
        for (; b + 7 * 32 < nblocks; b += 8 * 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[static_cast<size_t>(b + 32 * u) * step];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; b < nblocks; b += 32) s += src[static_cast<size_t>(b) * step];
    }
    red[g][threadIdx.x & 31] = s;
    __syncthreads();
    if (g == 0 && c < C) {
        float t = 0.f;
        for (int i = 0; i < 32; ++i) t += red[i][threadIdx.x];
        float* out = k == 0 ? out0 : out1;
        if (out) out[c] = t;
    }
}

// ------------------------------------------------------- backward of backward --
// tests/kernel_math.py::ln_bwd2:
//   xdot = rstd (tz - mean(tz) - xhat mean(tz xhat))
//   gdy = gamma xdot ; ggamma = sum_rows dy xdot
//   gz = -rstd (xhat mean(xdot u) + xdot mean(u xhat) + w mean(tz xhat)),  u = gamma dy,
//   w = rstd (u - mean(u) - xhat mean(u xhat))
template <typename T, int G, int QPL, bool PAIR = false>
__global__ __launch_bounds__(kBlock) void ln_bwd2_kernel(const T* __restrict__ a, const T* __restrict__ r,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const T* __restrict__ dy,
                                                       const T* __restrict__ tz, T* __restrict__ gz,
                                                       T* __restrict__ gdy, float* __restrict__ part, int64_t R,
                                                       int C) {
    constexpr int RPB = kBlock / G;
    __shared__ float4 red[QPL][kBlock];
    const int64_t passes = (R + RPB - 1) / RPB;
    const float invC = 1.0f / static_cast<float>(C);
    float4 gacc[QPL], gam[QPL];
    bool cok[QPL];
    int coff[QPL];
    const int lig = threadIdx.x % G;
#pragma unroll
    for (int t = 0; t < QPL; ++t) {
        const int c = PAIR ? (lig * 2 + t) * 4 : (lig + t * G) * 4;
        cok[t] = c < C;
        coff[t] = cok[t] ? c : 0;
        gam[t] = ld4(gamma + coff[t]);
        gacc[t] = f4(0.f);
    }
    for (int64_t pass = blockIdx.x; pass < passes; pass += gridDim.x) {
        const RowMap m = map_row<G>(pass, R);
        const float mu = mean[m.row], rs = rstd[m.row];
        float4 xh[QPL], u[QPL], tt[QPL], g[QPL], zin[QPL], rin[QPL];
        ld_slots<PAIR>(a + m.row * C, coff, zin);
        if (r) ld_slots<PAIR>(r + m.row * C, coff, rin);
        ld_slots<PAIR>(dy + m.row * C, coff, g);
        ld_slots<PAIR>(tz + m.row * C, coff, tt);
        float su = 0.f, sux = 0.f, st = 0.f, stx = 0.f;
#pragma unroll
        for (int t = 0; t < QPL; ++t) {
            float4 z = zin[t];
            if (r) z += rin[t];
            if (!cok[t] || !m.ok) {
                g[t] = f4(0.f);
                tt[t] = f4(0.f);
            }
            xh[t] = rs * (z - f4(mu));
            if (!cok[t]) xh[t] = f4(0.f);
            u[t] = g[t] * gam[t];
            su += hsum(u[t]);
            sux += hsum(u[t] * xh[t]);
            st += hsum(tt[t]);
            stx += hsum(tt[t] * xh[t]);
        }
        const float u1 = group_sum<G>(su) * invC, c2 = group_sum<G>(sux) * invC;
        const float t1 = group_sum<G>(st) * invC, t2 = group_sum<G>(stx) * invC;
        float sxu = 0.f;
#pragma unroll
        for (int t = 0; t < QPL; ++t) {
            tt[t] = rs * (tt[t] - f4(t1) - t2 * xh[t]);  // xdot
            if (!cok[t]) tt[t] = f4(0.f);
            sxu += hsum(tt[t] * u[t]);
            gacc[t] = fma4(g[t], tt[t], gacc[t]);
        }
        const float s1 = group_sum<G>(sxu) * invC;
        float4 o_gdy[QPL], o_gz[QPL];
#pragma unroll
        for (int t = 0; t < QPL; ++t) {
            const float4 w = rs * (u[t] - f4(u1) - c2 * xh[t]);
            o_gdy[t] = gam[t] * tt[t];
            o_gz[t] = (-rs) * (s1 * xh[t] + c2 * tt[t] + t2 * w);
        }
        if (m.ok) {
            st_slots<PAIR>(gdy + m.row * C, coff, cok, o_gdy);
            st_slots<PAIR>(gz + m.row * C, coff, cok, o_gz);
        }
    }
#pragma unroll
    for (int t = 0; t < QPL; ++t) red[t][threadIdx.x] = gacc[t];
    __syncthreads();
    if (threadIdx.x < G) {
#pragma unroll
        for (int t = 0; t < QPL; ++t) {
            float4 gs = f4(0.f);
            for (int g2 = 0; g2 < RPB; ++g2) gs += red[t][g2 * G + threadIdx.x];
            if (cok[t]) st4(part + static_cast<size_t>(blockIdx.x) * C + coff[t], gs);
        }
    }
}

// ----------------------------------------------------------------- dispatch --
struct LnGeom {
    int G, QPL;
};

bool ln_geometry(int C, LnGeom* g) {
    if (C < 4 || (C & 3)) return false;
    const int quads = C / 4;
    int G = 8;
    while (G < 64 && G < quads) G <<= 1;
    const int qpl = (quads + G - 1) / G;
    if (qpl > 4) return false;  // C <= 1024
    g->G = G;
    g->QPL = qpl == 3 ? 4 : qpl;
    return true;
}

int ln_grid(int64_t R, int G) {
    const int64_t rpb = kBlock / G;
    int64_t passes = (R + rpb - 1) / rpb;
    const int64_t cap = 1024;  // ~4 blocks per CU (fewer dgamma/dbeta partials to finish), grid-stride beyond
    return static_cast<int>(passes < cap ? (passes < 1 ? 1 : passes) : cap);
}

#define DG_FOR_LN(M) M(8, 1) M(16, 1) M(32, 1) M(64, 1) M(64, 2) M(64, 4)

template <typename... P>
bool aligned16(P... p) {
    return ((reinterpret_cast<uintptr_t>(p) | ...) & 15u) == 0;
}

}  // namespace

bool reduce_batch_try_add(const float* part, int S, long long n_floats, float* out);      // linear_wgrad.hip

// out_k[c] = sum_b part[b][k][c] in a fixed order (shared with the LayerNorm-backward epilogue of row_gemm.hip)
void launch_ln_finish(const float* part, int nblocks, int K, int C, float* out0, float* out1, hipStream_t stream) {
    hipLaunchKernelGGL(ln_finish_kernel, dim3((C + 31) / 32, K), dim3(1024), 0, stream, part, nblocks, K, C, out0, out1);
}
}  // namespace dg

using namespace dg;

extern "C" size_t dg_ln_workspace_bytes(int64_t R, int C) {
    LnGeom g;
    if (R < 1 || !ln_geometry(C, &g)) return 0;
    return static_cast<size_t>(ln_grid(R, g.G)) * 2 * C * sizeof(float);
}

extern "C" int dg_ln_residual_fwd(const void* a, const void* r, const float* gamma, const float* beta, void* y,
                                  float* mean, float* rstd, int64_t R, int C, float eps, int dtype, dg_stream_t stream_) {
    if (!a || !gamma || !beta || !y || !mean || !rstd) return fail(DG_E_ARG, "dg_ln_residual_fwd: null pointer");
    if (!dtype_ok(dtype)) return fail(DG_E_ARG, "dg_ln_residual_fwd: unknown dtype %d", dtype);
    LnGeom g;
    if (R < 0 || !ln_geometry(C, &g)) return fail(DG_E_SHAPE, "dg_ln_residual_fwd: unsupported C=%d (C%%4==0, C<=1024)", C);
    if (R == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int grid = ln_grid(R, g.G);
    ProfScope prof(DG_K_LN_FWD, stream);
    note_forward(R);
#define LAUNCH_T(T, GG, QQ)                                                                                      \
    hipLaunchKernelGGL((ln_fwd_kernel<T, GG, QQ>), dim3(grid), dim3(kBlock), 0, stream, static_cast<const T*>(a), \
                       static_cast<const T*>(r), gamma, beta, static_cast<T*>(y), mean, rstd, R, C, eps);
#define LAUNCH(GG, QQ)                                          \
    if (g.G == GG && g.QPL == QQ) {                             \
        if (dtype == DG_DTYPE_BF16) { LAUNCH_T(bf16_t, GG, QQ) } \
        else { LAUNCH_T(float, GG, QQ) }                        \
    }
    DG_FOR_LN(LAUNCH)
#undef LAUNCH
#undef LAUNCH_T
    return check_launch("dg_ln_residual_fwd");
}

extern "C" int dg_ln_residual_bwd_add(const void* a, const void* r, const float* gamma, const float* mean,
                                      const float* rstd, const void* dy, const void* dz_add, void* dz,
                                      float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, int64_t R,
                                      int C, int dtype, dg_stream_t stream_);

extern "C" int dg_ln_residual_bwd(const void* a, const void* r, const float* gamma, const float* mean,
                                  const float* rstd, const void* dy, void* dz, float* dgamma, float* dbeta,
                                  void* workspace, size_t workspace_bytes, int64_t R, int C, int dtype,
                                  dg_stream_t stream_) {
    return dg_ln_residual_bwd_add(a, r, gamma, mean, rstd, dy, nullptr, dz, dgamma, dbeta, workspace, workspace_bytes,
                                  R, C, dtype, stream_);
}

extern "C" int dg_ln_residual_bwd_add(const void* a, const void* r, const float* gamma, const float* mean,
                                      const float* rstd, const void* dy, const void* dz_add, void* dz,
                                      float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, int64_t R,
                                      int C, int dtype, dg_stream_t stream_) {
    if (!a || !gamma || !mean || !rstd || !dy || !dz || !workspace)
        return fail(DG_E_ARG, "dg_ln_residual_bwd: null pointer");
    if (!dtype_ok(dtype)) return fail(DG_E_ARG, "dg_ln_residual_bwd: unknown dtype %d", dtype);
    LnGeom g;
    if (R < 1 || !ln_geometry(C, &g)) return fail(DG_E_SHAPE, "dg_ln_residual_bwd: unsupported R=%lld C=%d", (long long)R, C);
    if (workspace_bytes < dg_ln_workspace_bytes(R, C)) return fail(DG_E_WORKSPACE, "dg_ln_residual_bwd: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // bf16 rows of 128 channels: 16 lanes per row, eight consecutive channels = one 16-byte access per lane (ld_slots<PAIR>)
    const bool pair = dtype == DG_DTYPE_BF16 && C == 128 && aligned16(a, r, dy, dz, dz_add);
    const int grid = ln_grid(R, pair ? 16 : g.G);
    float* part = static_cast<float*>(workspace);
    ProfScope prof(DG_K_LN_BWD, stream);
    note_forward(R);
#define LAUNCH_T(T, GG, QQ)                                                                                      \
    hipLaunchKernelGGL((ln_bwd_kernel<T, GG, QQ>), dim3(grid), dim3(kBlock), 0, stream, static_cast<const T*>(a), \
                       static_cast<const T*>(r), gamma, mean, rstd, static_cast<const T*>(dy), static_cast<T*>(dz), \
                       part, R, C, static_cast<const T*>(dz_add));
#define LAUNCH(GG, QQ)                                          \
    if (!pair && g.G == GG && g.QPL == QQ) {                    \
        if (dtype == DG_DTYPE_BF16) { LAUNCH_T(bf16_t, GG, QQ) } \
        else { LAUNCH_T(float, GG, QQ) }                        \
    }
    DG_FOR_LN(LAUNCH)
#undef LAUNCH
#undef LAUNCH_T
    if (pair)
        hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, 16, 2, true>), dim3(grid), dim3(kBlock), 0, stream, static_cast<const bf16_t*>(a),
                           static_cast<const bf16_t*>(r), gamma, mean, rstd, static_cast<const bf16_t*>(dy), static_cast<bf16_t*>(dz),
                           part, R, C, static_cast<const bf16_t*>(dz_add));
    // inside dg_linear_wgrad_batch_begin / _end the reduction joins that batch's single reduce launch (dgamma and dbeta
    // adjacent in memory: out[2][C]); the partials must then stay untouched until _end
    if ((dgamma || dbeta) && !(dgamma && dbeta == dgamma + C && reduce_batch_try_add(part, grid, 2 * static_cast<long long>(C), dgamma)))
        hipLaunchKernelGGL(ln_finish_kernel, dim3((C + 31) / 32, 2), dim3(1024), 0, stream, part, grid, 2, C, dgamma, dbeta);
    return check_launch("dg_ln_residual_bwd");
}

extern "C" int dg_ln_residual_bwd2(const void* a, const void* r, const float* gamma, const float* mean,
                                   const float* rstd, const void* dy, const void* tz, void* gz, void* gdy,
                                   float* ggamma, void* workspace, size_t workspace_bytes, int64_t R, int C,
                                   int dtype, dg_stream_t stream_) {
    if (!a || !gamma || !mean || !rstd || !dy || !tz || !gz || !gdy || !workspace)
        return fail(DG_E_ARG, "dg_ln_residual_bwd2: null pointer");
    if (!dtype_ok(dtype)) return fail(DG_E_ARG, "dg_ln_residual_bwd2: unknown dtype %d", dtype);
    LnGeom g;
    if (R < 1 || !ln_geometry(C, &g)) return fail(DG_E_SHAPE, "dg_ln_residual_bwd2: unsupported R=%lld C=%d", (long long)R, C);
    if (workspace_bytes < dg_ln_workspace_bytes(R, C)) return fail(DG_E_WORKSPACE, "dg_ln_residual_bwd2: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool pair = dtype == DG_DTYPE_BF16 && C == 128 && aligned16(a, r, dy, tz, gz, gdy);
    const int grid = ln_grid(R, pair ? 16 : g.G);
    float* part = static_cast<float*>(workspace);
    ProfScope prof(DG_K_LN_BWD2, stream);
    note_forward(R);
#define LAUNCH_T(T, GG, QQ)                                                                                       \
    hipLaunchKernelGGL((ln_bwd2_kernel<T, GG, QQ>), dim3(grid), dim3(kBlock), 0, stream, static_cast<const T*>(a), \
                       static_cast<const T*>(r), gamma, mean, rstd, static_cast<const T*>(dy),                    \
                       static_cast<const T*>(tz), static_cast<T*>(gz), static_cast<T*>(gdy), part, R, C);
#define LAUNCH(GG, QQ)                                          \
    if (!pair && g.G == GG && g.QPL == QQ) {                    \
        if (dtype == DG_DTYPE_BF16) { LAUNCH_T(bf16_t, GG, QQ) } \
        else { LAUNCH_T(float, GG, QQ) }                        \
    }
    DG_FOR_LN(LAUNCH)
#undef LAUNCH
#undef LAUNCH_T
    if (pair)
        hipLaunchKernelGGL((ln_bwd2_kernel<bf16_t, 16, 2, true>), dim3(grid), dim3(kBlock), 0, stream, static_cast<const bf16_t*>(a),
                           static_cast<const bf16_t*>(r), gamma, mean, rstd, static_cast<const bf16_t*>(dy),
                           static_cast<const bf16_t*>(tz), static_cast<bf16_t*>(gz), static_cast<bf16_t*>(gdy), part, R, C);
    if (ggamma)
        hipLaunchKernelGGL(ln_finish_kernel, dim3((C + 31) / 32, 1), dim3(1024), 0, stream, part, grid, 1, C, ggamma,
                           static_cast<float*>(nullptr));
    return check_launch("dg_ln_residual_bwd2");
}
