// bf16 "row GEMM" for the per-edge / per-node dense layers (reference src/model/layers.py: MHA
// projections :111-116,127,135 and MLP.fc1/fc2 :50-53) in the bf16 configuration (BASELINE
// configs[2]): activations bf16 in HBM, ONE v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate,
// fp32 bias / ReLU / residual / LayerNorm arithmetic, bf16 result.
//
//     Y[R,N] = epilogue( A[R,K] . W'^T ),      (K, N) in {(128,128), (128,384), (384,128)}
//     epilogue(v) = LN?( relu?(v + bias) * mask? + residual? )
//
// MI355X mapping (differences to the fp32 / bf16x6 kernels of row_gemm.hip)
//   * the A tile needs no split: 64 rows x K bf16 go HBM -> LDS by LDS-DMA, double buffered, the
//     swizzle applied on the source address (gemm_bf16.h);
//   * 8 waves; wave (c = w & 3, g = w >> 2) owns tile rows [32 g, 32 g + 32) and output channels
//     32 (c + 4 nc) .. + 31 of every 128-channel chunk nc.  The product is "swapped" (weights are the
//     MFMA A operand, activations the B operand), so a lane ends up with 4 consecutive channels of ONE
//     row: accumulators leave through an fp32 exchange tile as 16-byte slots;
//   * weight fragments (P32 order) stay in VGPRs for the whole persistent kernel: 32 KC NC registers;
//   * every result row is finalised by one half-wave from the exchange tile: bias, ReLU (+ 1 bit per
//     element out / mask in), residual (8-byte bf16 loads issued before the MFMA phase), LayerNorm with
//     DPP row sums, then 8-byte-per-lane stores = whole 256-byte rows.
// Algorithmic bytes per launch: 2 R (K + N (1 + [residual] + [pre])).
#include "gemm_bf16.h"

namespace dg {

// fp32 implementations (row_gemm.hip)
size_t row_gemm_f32_packed_floats(int n_out, int k_contract);
int row_gemm_f32_pack(const float* w, float* packed, int rows, int cols, int mode, dg_stream_t stream,
                      const float* w1 = nullptr, const float* w2 = nullptr);
int row_gemm_f32_lin3(const float* a, const float* packed, float* y0, float* y1, float* y2, int64_t R, const float* b0,
                      const float* b1, const float* b2, dg_stream_t stream);
int row_gemm_f32_sum3(const float* a0, const float* a1, const float* a2, const float* packed, float* y, int64_t R,
                      const float* residual, dg_stream_t stream);
size_t row_gemm_f32_mask_words(int64_t R, int K, int N);
int row_gemm_f32_pack_batch(const void* table, int n, int max_rows_cols, dg_stream_t stream);
size_t row_gemm_f32_ln_bwd_workspace_bytes();
int row_gemm_f32_ln_bwd(const float* a, const float* packed, float* dz, int64_t R, int K, const float* residual,
                        const float* pre, const float* mean, const float* rstd, const float* gamma, float* dgamma,
                        float* dbeta, void* workspace, size_t workspace_bytes, dg_stream_t stream);
int row_gemm_f32_ln_in(const float* dy, const float* pre, const float* mean, const float* rstd, const float* gamma,
                       const float* packed, float* dz, float* y, float* dgamma, float* dbeta, void* workspace,
                       size_t workspace_bytes, int64_t R, dg_stream_t stream);
int row_gemm_f32(const float* a, const float* packed, float* y, int64_t R, int K, int N, const float* bias, int relu,
                 unsigned* relu_bits_out, const unsigned* mask_bits, const float* residual, const float* gamma,
                 const float* beta, float* mean, float* rstd, float* pre_ln, float eps, dg_stream_t stream,
                 const float* ascale = nullptr, float* yscale = nullptr, int afmt = 0, int yfmt = 0, const void* alo = nullptr,
                 void* ylo = nullptr);

namespace {

// ---------------------------------------------------------------- weight packing --
// mb_size 32: P32, 16: P16 (see gemm_bf16.h).  One thread per (m-block, k-step, lane).
__global__ void pack_bf16_kernel(const float* __restrict__ w, bf16x8* __restrict__ p, int rows, int cols, int mode,
                                 int mb_size) {
    const int M = mode == 0 ? rows : cols, K = mode == 0 ? cols : rows;
    const int kstep = 512 / mb_size;                 // contraction elements per MFMA
    const int MB = M / mb_size, KS = K / kstep;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= MB * KS * 64) return;
    const int lane = idx & 63, ks = (idx >> 6) % KS, mb = (idx >> 6) / KS;
    const int m = mb * mb_size + (lane & (mb_size - 1));
    const int k0 = ks * kstep + 8 * (mb_size == 32 ? lane >> 5 : lane >> 4);
    bf16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        const float v = mode == 0 ? w[static_cast<size_t>(m) * cols + k] : w[static_cast<size_t>(k) * cols + m];
        out[j] = static_cast<__bf16>(v);
    }
    p[idx] = out;
}

// every stale bf16 pack of a network in one launch: blockIdx.y walks a device table of { w, packed, rows, cols, mode }
__global__ void pack_bf16_batch_kernel(const long long* __restrict__ table, int mb_size) {
    const long long* e = table + 7 * static_cast<size_t>(blockIdx.y);      // (entries 5, 6: fp32 stacks only)
    const float* w = reinterpret_cast<const float*>(e[0]);
    bf16x8* p = reinterpret_cast<bf16x8*>(e[1]);
    const int rows = static_cast<int>(e[2]), cols = static_cast<int>(e[3]), mode = static_cast<int>(e[4]);
    const int M = mode == 0 ? rows : cols, K = mode == 0 ? cols : rows;
    const int kstep = 512 / mb_size;
    const int MB = M / mb_size, KS = K / kstep;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= MB * KS * 64) return;
    const int lane = idx & 63, ks = (idx >> 6) % KS, mb = (idx >> 6) / KS;
    const int m = mb * mb_size + (lane & (mb_size - 1));
    const int k0 = ks * kstep + 8 * (mb_size == 32 ? lane >> 5 : lane >> 4);
    bf16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        const float v = mode == 0 ? w[static_cast<size_t>(m) * cols + k] : w[static_cast<size_t>(k) * cols + m];
        out[j] = static_cast<__bf16>(v);
    }
    p[idx] = out;
}

struct EpiB {
    const float* bias;          // [N] or null
    const unsigned* mask_bits;  // ReLU mask written by a relu launch of the same (K, N) geometry
    unsigned* relu_bits;        // optional output of the ReLU epilogue
    const bf16_t* residual;     // [R,N] or null
    const float* gamma;         // LayerNorm (N == 128) or null
    const float* beta;
    float* mean;
    float* rstd;
    bf16_t* pre;                // optional [R,N]: pre-LayerNorm sum
    float eps;
    int relu;
};

template <int KC, int NC>
__global__ __launch_bounds__(512, (KC * NC == 1 ? 4 : 2)) void row_gemm_bf16_kernel(const bf16_t* __restrict__ a,
                                                                                   const bf16x8* __restrict__ packed,
                                                                                   bf16_t* __restrict__ y, int64_t R,
                                                                                   EpiB ep) {
    constexpr int K = 128 * KC, N = 128 * NC, KS = KC * 8;
    constexpr int ABYTES = kRowsPerTile * K * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* abuf = smem;                       // [2][64][K] bf16
    char* xch = smem + 2 * ABYTES;           // [64][N] fp32 exchange tile
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = w & 3, g = w >> 2;
    const int half = lane >> 5, col = lane & 31;
    const int64_t tiles = (R + kRowsPerTile - 1) / kRowsPerTile;

    bf16x8 wf[NC][KS];
#pragma unroll
    for (int nc = 0; nc < NC; ++nc)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[nc][ks] = packed[(static_cast<size_t>(c + 4 * nc) * KS + ks) * 64 + lane];
    float4 bias4[NC], gam = f4(0.f), bet = f4(0.f);
#pragma unroll
    for (int nc = 0; nc < NC; ++nc) bias4[nc] = ep.bias ? ld4(ep.bias + 128 * nc + 4 * col) : f4(0.f);
    if (ep.gamma) {
        gam = ld4(ep.gamma + 4 * col);
        bet = ld4(ep.beta + 4 * col);
    }
    wait_all_vmem_visible();

    int64_t tix = blockIdx.x;
    if (tix < tiles) dma_tile_bf16<K, 8>(a, tix * kRowsPerTile, R, abuf, w, lane);
    int buf = 0;
    for (; tix < tiles; tix += gridDim.x, buf ^= 1) {
        const int64_t r0 = tix * kRowsPerTile;
        wait_all_vmem();
        __syncthreads();      // tile `tix` landed for every wave; exchange tile and the other buffer are free
        // rows finalised by this wave: 8 w + 2 it + half.  Residual rows are requested before the MFMA
        // phase; the bit masks too.
        u32x2_t res[4][NC];
        if (ep.residual) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                int64_t row = r0 + 8 * w + 2 * it + half;
                if (row > R - 1) row = R - 1;
#pragma unroll
                for (int nc = 0; nc < NC; ++nc)
                    res[it][nc] = *reinterpret_cast<const u32x2_t*>(ep.residual + row * N + 128 * nc + 4 * col);
            }
        }
        constexpr int BW = (16 * NC + 31) / 32;       // bit-mask words per lane and tile
        unsigned mbits[BW];
        const size_t bix = (static_cast<size_t>(tix) * 512 + threadIdx.x) * BW;
        if (ep.mask_bits) {
#pragma unroll
            for (int i = 0; i < BW; ++i) mbits[i] = ep.mask_bits[bix + i];
        }
        if (tix + gridDim.x < tiles)
            dma_tile_bf16<K, 8>(a, (tix + gridDim.x) * kRowsPerTile, R, abuf + (buf ^ 1) * ABYTES, w, lane);

        // ---- MFMA phase: acc[nc][m = channel][n = row] += W'[m][k] x[n][k]
        f32x16 acc[NC];
#pragma unroll
        for (int nc = 0; nc < NC; ++nc)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nc][i] = 0.f;
        const char* at = abuf + buf * ABYTES;
        const int arow = 32 * g + col;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(at + tile_off(arow, 16 * ks + 8 * half, K));
#pragma unroll
            for (int nc = 0; nc < NC; ++nc)
                acc[nc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nc][ks], xf, acc[nc], 0, 0, 0);
        }
        // accumulators -> exchange tile: lane holds channels 32 (c + 4 nc) + 8 q + 4 half + {0..3} of row arow
#pragma unroll
        for (int nc = 0; nc < NC; ++nc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = 32 * (c + 4 * nc) + 8 * q + 4 * half;
                *reinterpret_cast<float4*>(xch + xch_off(arow, ch, N)) =
                    make_float4(acc[nc][4 * q], acc[nc][4 * q + 1], acc[nc][4 * q + 2], acc[nc][4 * q + 3]);
            }
        __syncthreads();
        // ---- row phase
        unsigned newbits[BW];
#pragma unroll
        for (int i = 0; i < BW; ++i) newbits[i] = 0u;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rr = 8 * w + 2 * it + half;
            const int64_t row = r0 + rr;
            const bool ok = row < R;
            float4 v[NC];
#pragma unroll
            for (int nc = 0; nc < NC; ++nc) {
                v[nc] = *reinterpret_cast<const float4*>(xch + xch_off(rr, 128 * nc + 4 * col, N)) + bias4[nc];
                const int b0 = (it * NC + nc) * 4;
                if (ep.relu) {
                    newbits[b0 >> 5] |= ((v[nc].x > 0.f ? 1u : 0u) | (v[nc].y > 0.f ? 2u : 0u) | (v[nc].z > 0.f ? 4u : 0u) |
                                         (v[nc].w > 0.f ? 8u : 0u))
                                        << (b0 & 31);
                    v[nc] = max4(v[nc], f4(0.f));
                }
                if (ep.mask_bits) {
                    const unsigned mb = mbits[b0 >> 5] >> (b0 & 31);
                    v[nc] = make_float4(mb & 1u ? v[nc].x : 0.f, mb & 2u ? v[nc].y : 0.f, mb & 4u ? v[nc].z : 0.f,
                                        mb & 8u ? v[nc].w : 0.f);
                }
                if (ep.residual) v[nc] += unpack4_bf16(res[it][nc]);
            }
            if (NC == 1 && ep.gamma) {
                if (ep.pre && ok) st4(ep.pre + row * N + 4 * col, v[0]);
                const float mu = half_wave_sum((v[0].x + v[0].y) + (v[0].z + v[0].w)) * (1.0f / 128.0f);
                const float4 d = v[0] - f4(mu);
                const float var = half_wave_sum((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * (1.0f / 128.0f);
                const float rs = rsqrtf(var + ep.eps);
                v[0] = fma4(rs * d, gam, bet);
                if (ok && col == 0) {
                    ep.mean[row] = mu;
                    ep.rstd[row] = rs;
                }
            }
            if (ok) {
#pragma unroll
                for (int nc = 0; nc < NC; ++nc) st4(y + row * N + 128 * nc + 4 * col, v[nc]);
            }
        }
        if (ep.relu_bits) {
#pragma unroll
            for (int i = 0; i < BW; ++i) ep.relu_bits[bix + i] = newbits[i];
        }
    }
}

bool bf16_shape_ok(int K, int N) { return (K == 128 && (N == 128 || N == 384)) || (K == 384 && N == 128); }

}  // namespace

size_t row_gemm_bf16_mask_words(int64_t R, int K, int N) {
    if (!bf16_shape_ok(K, N) || R < 1) return 0;
    const size_t tiles = static_cast<size_t>((R + kRowsPerTile - 1) / kRowsPerTile);
    return tiles * 512 * ((16 * (N / 128) + 31) / 32);
}

int pack_bf16(const float* w, void* packed, int rows, int cols, int mode, int mb_size, hipStream_t stream) {
    const int M = mode == 0 ? rows : cols, K = mode == 0 ? cols : rows;
    if (M % mb_size || K % (512 / mb_size)) return fail(DG_E_SHAPE, "bf16 pack: %d x %d is not a multiple of the MFMA tile", M, K);
    const int total = (M / mb_size) * (K / (512 / mb_size)) * 64;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w,
                       static_cast<bf16x8*>(packed), rows, cols, mode, mb_size);
    return check_launch("bf16 pack");
}

int row_gemm_bf16(const bf16_t* a, const void* packed, bf16_t* y, int64_t R, int K, int N, const float* bias, int relu,
                  unsigned* relu_bits_out, const unsigned* mask_bits, const bf16_t* residual, const float* gamma,
                  const float* beta, float* mean, float* rstd, bf16_t* pre_ln, float eps, hipStream_t stream) {
    if (!bf16_shape_ok(K, N))
        return fail(DG_E_SHAPE, "dg_row_gemm(bf16): unsupported K=%d N=%d (supported: 128x128, 128x384, 384x128)", K, N);
    if (gamma && (N != 128 || !beta || !mean || !rstd))
        return fail(DG_E_ARG, "dg_row_gemm: LayerNorm epilogue needs N == 128, beta, mean and rstd");
    if (R == 0) return 0;
    EpiB ep{bias, mask_bits, relu_bits_out, residual, gamma, beta, mean, rstd, pre_ln, eps, relu};
    const int64_t tiles = (R + kRowsPerTile - 1) / kRowsPerTile;
    const bf16x8* pk = static_cast<const bf16x8*>(packed);
    ProfScope prof(R < edge_rows() ? DG_K_ROW_GEMM : (K == 384 ? DG_K_ROW_GEMM_E_K384 : (N == 384 ? DG_K_ROW_GEMM_E_N384 : DG_K_ROW_GEMM_E_128)), stream);
#define LAUNCH(KC_, NC_, PER_CU_)                                                                                  \
    {                                                                                                              \
        constexpr int lds = 2 * kRowsPerTile * 128 * KC_ * 2 + kRowsPerTile * 128 * NC_ * 4;                       \
        DG_OPT_IN_LDS((&row_gemm_bf16_kernel<KC_, NC_>), lds);                                                     \
        const int grid = static_cast<int>(tiles < 256 * PER_CU_ ? tiles : 256 * PER_CU_);                          \
        hipLaunchKernelGGL((row_gemm_bf16_kernel<KC_, NC_>), dim3(grid), dim3(512), lds, stream, a, pk, y, R, ep); \
    }
    if (K == 128 && N == 128) LAUNCH(1, 1, 2)
    else if (K == 128) LAUNCH(1, 3, 1)
    else LAUNCH(3, 1, 1)
#undef LAUNCH
    return check_launch("dg_row_gemm(bf16)");
}

}  // namespace dg

using namespace dg;

// ---- public entry points: dispatch on dtype -------------------------------------------------------
extern "C" size_t dg_row_gemm_packed_bytes(int n_out, int k_contract, int dtype) {
    if (n_out < 1 || k_contract < 1) return 0;
    dtype = act_dtype(dtype);
    if (dtype == DG_DTYPE_BF16) return static_cast<size_t>(n_out) * k_contract * 2;
    return row_gemm_f32_packed_floats(n_out, k_contract) * sizeof(float);
}

extern "C" int dg_row_gemm_pack(const float* w, void* packed, int rows, int cols, int mode, int dtype,
                                dg_stream_t stream_) {
    if (!w || !packed) return fail(DG_E_ARG, "dg_row_gemm_pack: null pointer");
    if (mode != 0 && mode != 1) return fail(DG_E_ARG, "dg_row_gemm_pack: mode must be 0 (forward) or 1 (dgrad)");
    dtype = act_dtype(dtype);
    if (dtype == DG_DTYPE_BF16) return pack_bf16(w, packed, rows, cols, mode, 32, static_cast<hipStream_t>(stream_));
    if (dtype != DG_DTYPE_F32) return fail(DG_E_ARG, "dg_row_gemm_pack: unknown dtype %d", dtype);
    return row_gemm_f32_pack(w, static_cast<float*>(packed), rows, cols, mode, stream_);
}

extern "C" int dg_row_gemm_pack3(const float* w0, const float* w1, const float* w2, void* packed, int cols, int mode,
                                 int dtype, dg_stream_t stream_) {
    if (!w0 || !w1 || !w2 || !packed) return fail(DG_E_ARG, "dg_row_gemm_pack3: null pointer");
    if (mode != 0 && mode != 1) return fail(DG_E_ARG, "dg_row_gemm_pack3: mode must be 0 (forward) or 1 (dgrad)");
    if (dtype != DG_DTYPE_F32) return fail(DG_E_SHAPE, "dg_row_gemm_pack3: float32 activations only");
    if (cols != 128) return fail(DG_E_SHAPE, "dg_row_gemm_pack3: three [128,128] weights (got cols = %d)", cols);
    return row_gemm_f32_pack(w0, static_cast<float*>(packed), 384, cols, mode, stream_, w1, w2);
}

extern "C" int dg_row_gemm_lin3(const void* a, const void* packed, void* y0, void* y1, void* y2, int64_t R, const float* b0,
                                const float* b1, const float* b2, int dtype, dg_stream_t stream_) {
    if (dtype != DG_DTYPE_F32) return fail(DG_E_SHAPE, "dg_row_gemm_lin3: float32 activations only");
    return row_gemm_f32_lin3(static_cast<const float*>(a), static_cast<const float*>(packed), static_cast<float*>(y0),
                             static_cast<float*>(y1), static_cast<float*>(y2), R, b0, b1, b2, stream_);
}

extern "C" int dg_row_gemm_sum3(const void* a0, const void* a1, const void* a2, const void* packed, void* y, int64_t R,
                                const void* residual, int dtype, dg_stream_t stream_) {
    if (dtype != DG_DTYPE_F32) return fail(DG_E_SHAPE, "dg_row_gemm_sum3: float32 activations only");
    return row_gemm_f32_sum3(static_cast<const float*>(a0), static_cast<const float*>(a1), static_cast<const float*>(a2),
                             static_cast<const float*>(packed), static_cast<float*>(y), R,
                             static_cast<const float*>(residual), stream_);
}

extern "C" size_t dg_row_gemm_mask_words(int64_t R, int K, int N, int dtype) {
    return act_dtype(dtype) == DG_DTYPE_BF16 ? row_gemm_bf16_mask_words(R, K, N) : row_gemm_f32_mask_words(R, K, N);
}

extern "C" int dg_row_gemm(const void* a, const void* packed, void* y, int64_t R, int K, int N, const float* bias,
                           int relu, unsigned* relu_bits_out, const unsigned* mask_bits, const void* residual,
                           const float* gamma, const float* beta, float* mean, float* rstd, void* pre_ln, float eps,
                           int dtype, dg_stream_t stream_) {
    if (!a || !packed || !y) return fail(DG_E_ARG, "dg_row_gemm: null pointer");
    if (R < 0) return fail(DG_E_SHAPE, "dg_row_gemm: negative row count");
    if (dtype == DG_DTYPE_BF16)
        return row_gemm_bf16(static_cast<const bf16_t*>(a), packed, static_cast<bf16_t*>(y), R, K, N, bias, relu,
                             relu_bits_out, mask_bits, static_cast<const bf16_t*>(residual), gamma, beta, mean, rstd,
                             static_cast<bf16_t*>(pre_ln), eps, static_cast<hipStream_t>(stream_));
    if (act_dtype(dtype) != DG_DTYPE_F32) return fail(DG_E_ARG, "dg_row_gemm: unknown dtype %d", dtype);
    // DG_DTYPE_F32_H16: the 384-wide operand (a for K = 384, y for N = 384) is an fp16 plane + inverse row scales
    const float* ascale = nullptr;
    float* yscale = nullptr;
    const void* alo = nullptr;
    void* ylo = nullptr;
    const size_t hoff = hidden_scale_offset(R, 384);
    if (dtype == DG_DTYPE_F32_H16 && K == 384) ascale = reinterpret_cast<const float*>(static_cast<const char*>(a) + hoff);
    if (dtype == DG_DTYPE_F32_H16 && N == 384) yscale = reinterpret_cast<float*>(static_cast<char*>(y) + hoff);
    if (dtype == DG_DTYPE_F32_H32 && K == 384) {      // hi plane | lo plane | inverse row scales
        alo = static_cast<const char*>(a) + hoff;
        ascale = reinterpret_cast<const float*>(static_cast<const char*>(a) + 2 * hoff);
    }
    if (dtype == DG_DTYPE_F32_H32 && N == 384) {
        ylo = static_cast<char*>(y) + hoff;
        yscale = reinterpret_cast<float*>(static_cast<char*>(y) + 2 * hoff);
    }
    return row_gemm_f32(static_cast<const float*>(a), static_cast<const float*>(packed), static_cast<float*>(y), R, K, N,
                        bias, relu, relu_bits_out, mask_bits, static_cast<const float*>(residual), gamma, beta, mean, rstd,
                        static_cast<float*>(pre_ln), eps, stream_, ascale, yscale, K == 384 ? hidden_fmt(dtype) : 0,
                        N == 384 ? hidden_fmt(dtype) : 0, alo, ylo);
}

extern "C" size_t dg_hidden_scale_offset(int64_t R, int H) { return R < 0 || H < 1 ? 0 : hidden_scale_offset(R, H); }

extern "C" size_t dg_hidden_bytes(int64_t R, int H, int dtype) {
    if (R < 0 || H < 1) return 0;
    if (dtype == DG_DTYPE_F32_H16) return hidden_scale_offset(R, H) + static_cast<size_t>(R) * 4;
    if (dtype == DG_DTYPE_F32_H24) return static_cast<size_t>(R) * H * 3;
    if (dtype == DG_DTYPE_F32_H32) return 2 * hidden_scale_offset(R, H) + static_cast<size_t>(R) * 4;
    return static_cast<size_t>(R) * H * dtype_size(dtype);
}

extern "C" size_t dg_row_gemm_ln_bwd_workspace_bytes(int dtype) {
    return dtype == DG_DTYPE_F32 ? row_gemm_f32_ln_bwd_workspace_bytes() : 0;
}

extern "C" int dg_row_gemm_ln_bwd(const void* a, const void* packed, void* dz, int64_t R, int K, const void* residual,
                                  const void* ln_pre, const float* ln_mean, const float* ln_rstd, const float* ln_gamma,
                                  float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, int dtype,
                                  dg_stream_t stream_) {
    if (dtype != DG_DTYPE_F32)
        return fail(DG_E_ARG, "dg_row_gemm_ln_bwd: float32 only (the bf16 configuration fuses this LayerNorm backward elsewhere)");
    return row_gemm_f32_ln_bwd(static_cast<const float*>(a), static_cast<const float*>(packed), static_cast<float*>(dz), R, K,
                               static_cast<const float*>(residual), static_cast<const float*>(ln_pre), ln_mean, ln_rstd,
                               ln_gamma, dgamma, dbeta, workspace, workspace_bytes, stream_);
}

extern "C" int dg_row_gemm_ln_bwd_in(const void* dy, const void* ln_pre, const float* ln_mean, const float* ln_rstd,
                                     const float* ln_gamma, const void* packed, void* dz, void* y, float* dgamma,
                                     float* dbeta, void* workspace, size_t workspace_bytes, int64_t R, int K, int N,
                                     int dtype, dg_stream_t stream_) {
    if (dtype != DG_DTYPE_F32) return fail(DG_E_ARG, "dg_row_gemm_ln_bwd_in: float32 only");
    if (K != 128 || N != 128) return fail(DG_E_SHAPE, "dg_row_gemm_ln_bwd_in: unsupported K=%d N=%d (K = N = 128)", K, N);
    return row_gemm_f32_ln_in(static_cast<const float*>(dy), static_cast<const float*>(ln_pre), ln_mean, ln_rstd, ln_gamma,
                              static_cast<const float*>(packed), static_cast<float*>(dz), static_cast<float*>(y), dgamma,
                              dbeta, workspace, workspace_bytes, R, stream_);
}

extern "C" int dg_row_gemm_pack_batch(const void* table, int n, int max_dim, int dtype, dg_stream_t stream_) {
    if (!table) return fail(DG_E_ARG, "dg_row_gemm_pack_batch: null pointer");
    if (max_dim < 1 || max_dim > 4096) return fail(DG_E_SHAPE, "dg_row_gemm_pack_batch: max_dim %d", max_dim);
    if (dtype == DG_DTYPE_BF16) {      // entries must be multiples of the MFMA tile (as dg_row_gemm_pack checks one by one)
        if (n < 1) return 0;
        const int total = (max_dim / 32) * (max_dim / 16) * 64;
        hipLaunchKernelGGL(pack_bf16_batch_kernel, dim3((total + 255) / 256, n), dim3(256), 0,
                           static_cast<hipStream_t>(stream_), static_cast<const long long*>(table), 32);
        return check_launch("dg_row_gemm_pack_batch(bf16)");
    }
    if (dtype != DG_DTYPE_F32) return fail(DG_E_ARG, "dg_row_gemm_pack_batch: unknown dtype %d", dtype);
    return row_gemm_f32_pack_batch(table, n, max_dim, stream_);
}
