// fp32-MFMA "row GEMM" for the per-edge / per-node dense layers of DrugGEN's
// encoder (reference src/model/layers.py: MHA projections :111-116,127,135 and
// MLP.fc1/fc2 :50-53), with the elementwise neighbours of those GEMMs fused in:
//
//     Y[R,N] = epilogue( prologue(A)[R,K] . B )        R = B*N*N rows (518 400 at configs[1])
//
//   forward   B[k][n] = W[n][k]    (y = x W^T + b)          -> pack mode 0
//   dgrad     B[k][n] = W[k][n]    (dx = dy W)              -> pack mode 1
//   prologue: A, or A * (mask > 0)            (ReLU backward folded into the operand load)
//   epilogue: + bias, ReLU, * (mask > 0), + residual, LayerNorm(gamma, beta) -> y (+ mean, rstd)
//
// MI355X mapping
//   * v_mfma_f32_32x32x2_f32 (exact fp32).  A workgroup (4 waves) owns a 64-row tile; wave w owns
//     the 32-column slab w of a 128-column chunk for both 32-row halves (2 accumulators).
//   * The weight operand never touches LDS: a tiny pack kernel re-orders W once per weight
//     version into MFMA *fragment order*, so each wave streams its B fragments for a 128x128
//     weight block as 16 perfectly coalesced float4 loads straight into 64 VGPRs (L2-resident).
//     The contraction index is permuted (lanes 0-31 take k = 4q+j, lanes 32-63 take k = 64+4q+j)
//     so that both operands are read 16 bytes at a time.
//   * The activation tile goes HBM -> VGPR (16 B/lane, coalesced rows) -> LDS with a 528-byte
//     row pitch: ds_read_b128 of the A fragments is bank-conflict free.
//   * LayerNorm epilogue: the 64x128 result tile is exchanged through LDS so that each wave
//     normalises whole rows (32 lanes x float4, 5-step butterflies) and stores full 512 B rows.
#include "common.h"

namespace dg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTR = 64;          // rows per workgroup tile
constexpr int kPitch = 132;      // LDS row pitch in floats for a 128-wide tile (528 B)

// ---------------------------------------------------------------- weight packing --
// P[((t*KC + c)*16 + q)*64 + lane] (float4), t = 32-column tile of the output, c = 128-wide
// chunk of the contraction, lane = (n = lane & 31, h = lane >> 5):
//   mode 0 (forward): { W[32t+n][128c + 64h + 4q + j] }_j           W: [Nout, Kin]
//   mode 1 (dgrad)  : { W[128c + 64h + 4q + j][32t+n] }_j           W: [Kcontract, Nout']
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ p, int rows, int cols, int mode,
                                   int n_tiles, int k_chunks) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // one float4 each
    const int total = n_tiles * k_chunks * 16 * 64;
    if (idx >= total) return;
    const int lane = idx & 63, q = (idx >> 6) & 15;
    const int c = (idx >> 10) % k_chunks, t = (idx >> 10) / k_chunks;
    const int n = 32 * t + (lane & 31);
    const int k0 = 128 * c + 64 * (lane >> 5) + 4 * q;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + j;
        if (mode == 0)
            v[j] = (n < rows && k < cols) ? w[static_cast<size_t>(n) * cols + k] : 0.f;
        else
            v[j] = (k < rows && n < cols) ? w[static_cast<size_t>(k) * cols + n] : 0.f;
    }
    st4(p + static_cast<size_t>(idx) * 4, make_float4(v[0], v[1], v[2], v[3]));
}

struct Epilogue {
    const float* bias;      // [N] or null
    const float* mask;      // [R,N] or null: y *= (mask > 0)
    const float* residual;  // [R,N] or null
    const float* gamma;     // LayerNorm (needs N == 128) or null
    const float* beta;
    float* mean;
    float* rstd;
    float* pre;             // optional [R,N]: the pre-LayerNorm sum, saved for the backward
    float eps;
    int relu;
};

// 32-lane butterfly (lanes l and l^m for m = 1..16 stay inside one half-wave)
__device__ __forceinline__ float half_sum(float x) {
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) x += __shfl_xor(x, m, 64);
    return x;
}

// KC = K/128 contraction chunks, NC = N/128 output chunks; at most one of them > 1.
template <int KC, int NC>
__global__ __launch_bounds__(256, 2) void row_gemm_kernel(const float* __restrict__ a, const float* __restrict__ amask,
                                                      const float* __restrict__ packed, float* __restrict__ y,
                                                      int64_t R, Epilogue ep) {
    constexpr int K = KC * 128, N = NC * 128;
    static_assert(KC == 1 || NC == 1, "one of the two dimensions must be a single chunk");
    __shared__ __attribute__((aligned(16))) float tile[kTR * kPitch];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, col = lane & 31;
    const int64_t tiles = (R + kTR - 1) / kTR;

    for (int64_t tix = blockIdx.x; tix < tiles; tix += gridDim.x) {
        const int64_t r0 = tix * kTR;
        f32x16 acc[2];
#pragma unroll 1
        for (int nc = 0; nc < NC; ++nc) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
#pragma unroll 1
            for (int kc = 0; kc < KC; ++kc) {
                // B fragments for (n tile 4nc+w, k chunk kc): 16 coalesced float4 per lane
                float4 bf[16];
                const float* pb = packed + (static_cast<size_t>((4 * nc + w) * KC + kc) * 16) * 64 * 4 + lane * 4;
#pragma unroll
                for (int q = 0; q < 16; ++q) bf[q] = ld4(pb + q * 256);
                if (nc == 0 || KC > 1) {
                    // stage A[r0 .. r0+64, 128kc .. 128kc+128) -> LDS (pitch 132), prologue applied
                    __syncthreads();   // previous users of `tile` are done
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int chunk = it * 256 + threadIdx.x;   // 2048 float4 per tile
                        const int rr = chunk >> 5, cc = (chunk & 31) * 4;
                        const int64_t row = r0 + rr;
                        float4 v = f4(0.f);
                        if (row < R) {
                            v = ld4(a + row * K + 128 * kc + cc);
                            if (amask) {
                                const float4 mk = ld4(amask + row * K + 128 * kc + cc);
                                v.x = mk.x > 0.f ? v.x : 0.f;
                                v.y = mk.y > 0.f ? v.y : 0.f;
                                v.z = mk.z > 0.f ? v.z : 0.f;
                                v.w = mk.w > 0.f ? v.w : 0.f;
                            }
                        }
                        st4(tile + rr * kPitch + cc, v);
                    }
                    __syncthreads();
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float4 a0 = ld4(tile + col * kPitch + 64 * half + 4 * q);
                    const float4 a1 = ld4(tile + (32 + col) * kPitch + 64 * half + 4 * q);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bf[q].x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bf[q].x, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bf[q].y, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bf[q].y, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bf[q].z, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bf[q].z, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bf[q].w, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bf[q].w, acc[1], 0, 0, 0);
                }
            }
            // ---- epilogue for output chunk nc: this lane holds column n, 2 x 16 rows
            const int n = 128 * nc + 32 * w + col;
            const float bias = ep.bias ? ep.bias[n] : 0.f;
            if (ep.gamma == nullptr) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int64_t row = r0 + 32 * m + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                        if (row < R) {
                            float v = acc[m][reg] + bias;
                            if (ep.relu) v = fmaxf(v, 0.f);
                            if (ep.mask) v = ep.mask[row * N + n] > 0.f ? v : 0.f;
                            if (ep.residual) v += ep.residual[row * N + n];
                            y[row * N + n] = v;
                        }
                    }
            } else {
                // LayerNorm epilogue (N == 128): exchange through LDS, then whole rows per half-wave
                __syncthreads();   // every wave finished reading A fragments from `tile`
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int rr = 32 * m + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                        float v = acc[m][reg] + bias;
                        if (ep.relu) v = fmaxf(v, 0.f);
                        tile[rr * kPitch + 32 * w + col] = v;
                    }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int rr = w * 16 + it * 2 + half;       // wave w: rows 16w .. 16w+15
                    const int64_t row = r0 + rr;
                    const bool ok = row < R;
                    float4 v = ld4(tile + rr * kPitch + col * 4);
                    if (ep.residual && ok) v += ld4(ep.residual + row * N + col * 4);
                    if (ep.pre && ok) st4(ep.pre + row * N + col * 4, v);
                    const float mu = half_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 128.0f);
                    const float4 d = v - f4(mu);
                    const float var = half_sum((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * (1.0f / 128.0f);
                    const float rs = rsqrtf(var + ep.eps);
                    if (ok) {
                        st4(y + row * N + col * 4, fma4(rs * d, ld4(ep.gamma + col * 4), ld4(ep.beta + col * 4)));
                        if (col == 0) {
                            ep.mean[row] = mu;
                            ep.rstd[row] = rs;
                        }
                    }
                }
            }
        }
    }
}

}  // namespace
}  // namespace dg

using namespace dg;

extern "C" size_t dg_row_gemm_packed_floats(int n_out, int k_contract) {
    if (n_out < 1 || k_contract < 1) return 0;
    const size_t nt = (n_out + 31) / 32, kc = (k_contract + 127) / 128;
    return nt * kc * 16 * 64 * 4;
}

extern "C" int dg_row_gemm_pack(const float* w, float* packed, int rows, int cols, int mode, dg_stream_t stream_) {
    if (!w || !packed) return fail(DG_E_ARG, "dg_row_gemm_pack: null pointer");
    if (mode != 0 && mode != 1) return fail(DG_E_ARG, "dg_row_gemm_pack: mode must be 0 (forward) or 1 (dgrad)");
    const int n_out = mode == 0 ? rows : cols, k = mode == 0 ? cols : rows;
    const int nt = (n_out + 31) / 32, kc = (k + 127) / 128;
    const int total = nt * kc * 16 * 64;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream_), w,
                       packed, rows, cols, mode, nt, kc);
    return check_launch("dg_row_gemm_pack");
}

extern "C" int dg_row_gemm(const float* a, const float* a_mask, const float* packed, float* y, int64_t R, int K, int N,
                           const float* bias, int relu, const float* out_mask, const float* residual,
                           const float* gamma, const float* beta, float* mean, float* rstd, float* pre_ln,
                           float eps, dg_stream_t stream_) {
    if (!a || !packed || !y) return fail(DG_E_ARG, "dg_row_gemm: null pointer");
    if (R < 0 || !((K == 128 && (N == 128 || N == 384)) || (K == 384 && N == 128)))
        return fail(DG_E_SHAPE, "dg_row_gemm: unsupported K=%d N=%d (supported: 128x128, 128x384, 384x128)", K, N);
    if (gamma && (N != 128 || !beta || !mean || !rstd || out_mask))
        return fail(DG_E_ARG, "dg_row_gemm: LayerNorm epilogue needs N == 128, beta, mean, rstd and no output mask");
    if (R == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Epilogue ep{bias, out_mask, residual, gamma, beta, mean, rstd, pre_ln, eps, relu};
    const int64_t tiles = (R + kTR - 1) / kTR;
    const int grid = static_cast<int>(tiles < 2048 ? tiles : 2048);
    ProfScope prof(DG_K_ROW_GEMM, stream);
    const int kc = K / 128, nc = N / 128;
#define LAUNCH(KC_, NC_)          \
    if (kc == KC_ && nc == NC_)   \
        hipLaunchKernelGGL((row_gemm_kernel<KC_, NC_>), dim3(grid), dim3(256), 0, stream, a, a_mask, packed, y, R, ep);
    LAUNCH(1, 1)
    LAUNCH(1, 3)
    LAUNCH(3, 1)
#undef LAUNCH
    return check_launch("dg_row_gemm");
}
