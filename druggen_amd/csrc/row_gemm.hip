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
    const unsigned* mask_bits;  // per (tile, wave, lane) bit mask written by a ReLU forward of the SAME geometry
    unsigned* relu_bits;        // optional output of the ReLU epilogue: bit (16*m + reg) = (v > 0)
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

// Persistent row-GEMM workgroup.
//   KC  = K/128 contraction chunks (B fragments for all of them stay in VGPRs for the whole kernel)
//   NG  = N/128 output chunks, one 4-wave group each (all groups share the A tile)
//   MG  = row groups of waves: the TR-row tile is split into MG slabs of MT = TR/32/MG 32-row blocks
//   TR  = rows per tile
//   EXCH = epilogue through an LDS exchange tile (row-wise float4 residual loads / stores,
//          optional LayerNorm; NG == MG == 1); otherwise direct stores from the accumulator layout,
//          software-pipelined: the stores of tile t-1 are issued inside the MFMA phase of tile t.
// A tiles arrive by LDS-DMA into a double buffer; the LDS image is the global image with the
// 16-byte chunk index XOR-ed by (row & 15) inside every 512-byte segment (applied on the per-lane
// SOURCE address, the DMA destination is lane-linear), which makes the ds_read_b128 fragment
// reads bank-conflict free without padding.
template <int KC, int NG, int MG, int TR, bool EXCH, bool PIPE, int MINW, bool XPIPE = false>
__global__ __launch_bounds__(NG * MG * 256, MINW) void row_gemm_kernel(const float* __restrict__ a,
                                                                const float* __restrict__ packed,
                                                                float* __restrict__ y, int64_t R, Epilogue ep) {
    constexpr int K = KC * 128, N = NG * 128, MT = TR / 32 / MG, WAVES = NG * MG * 4;
    constexpr int SLOTS = TR * K / 4;            // 16-byte slots per tile
    constexpr int STEPS = KC * 16;               // MFMA steps (4 MFMAs per 32-row block each)
    static_assert(!EXCH || (NG == 1 && MG == 1), "exchange epilogue needs the whole tile in one group");
    static_assert(TR % (32 * MG) == 0, "tile rows must split evenly over the row groups");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);   // [2][TR*K] (+ [TR][128] exchange tile when XPIPE)
    float* xtile = lds + 2 * TR * K;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wq = w & 3, g = (w >> 2) % NG, mg = (w >> 2) / NG;
    const int half = lane >> 5, col = lane & 31;
    const int64_t tiles = (R + TR - 1) / TR;
    const int n = 128 * g + 32 * wq + col;
    const int row_base = 32 * MT * mg;           // first tile row of this wave's slab

    float4 bf[KC][16];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int q = 0; q < 16; ++q)
            bf[kc][q] = ld4(packed + (static_cast<size_t>((4 * g + wq) * KC + kc) * 16 + q) * 256 + lane * 4);
    const float bias = ep.bias ? ep.bias[n] : 0.f;

    auto dma_tile = [&](int64_t tile, int buf) {
        const int64_t r0 = tile * TR;
        const unsigned dst = lds_byte_address(lds + buf * (TR * K));
        for (int ii = w; ii < SLOTS / 64; ii += WAVES) {
            const int L = ii * 64 + lane;
            const int row = L / (K / 4), cs = L % (K / 4);
            const int src = (cs & ~31) | ((cs & 31) ^ (row & 15));
            if (r0 + row < R) dma16_async(a + (r0 + row) * K + src * 4, dst + ii * 1024);
        }
    };
    // direct epilogue of one accumulator register (tile rows r0.., this lane's column n)
    auto store_reg = [&](const f32x16 (&acc)[MT], int64_t r0, unsigned bits, int m, int reg, unsigned& newbits) {
        const int64_t row = r0 + row_base + 32 * m + (reg & 3) + 8 * (reg >> 2) + 4 * half;
        float v = acc[m][reg] + bias;
        if (ep.relu) {
            newbits |= (v > 0.f ? 1u : 0u) << (16 * m + reg);
            v = fmaxf(v, 0.f);
        }
        if (ep.mask_bits) v = (bits >> (16 * m + reg)) & 1u ? v : 0.f;
        if (row < R) y[row * N + n] = v;
    };

    // exchange-epilogue helpers (EXCH): the wave's accumulators go to a [TR][128] LDS tile, then every
    // half-wave finalises whole rows (residual add, optional LayerNorm, 512 B stores)
    auto exch_write = [&](float* ex, const f32x16 (&acc)[MT], int m, int reg) {
        const int rr = 32 * m + (reg & 3) + 8 * (reg >> 2) + 4 * half;
        float v = acc[m][reg] + bias;
        if (ep.relu) v = fmaxf(v, 0.f);
        ex[rr * 128 + 32 * wq + col] = v;
    };
    auto exch_row = [&](const float* ex, int64_t r0, int it, float4 res) {
        const int rr = wq * (TR / 4) + it * 2 + half;
        const int64_t row = r0 + rr;
        const bool ok = row < R;
        float4 v = ld4(ex + rr * 128 + col * 4);
        if (ep.residual) v += res;
        if (ep.gamma == nullptr) {
            if (ok) st4(y + row * N + col * 4, v);
            return;
        }
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
    };
    auto load_res = [&](int64_t r0, int it) {
        int64_t row = r0 + wq * (TR / 4) + it * 2 + half;
        if (row >= R) row = R - 1;
        return ep.residual ? ld4(ep.residual + row * N + col * 4) : f4(0.f);
    };
    float4 resP[TR / 8], resN[TR / 8];   // XPIPE: residual rows of the previous / current tile

    wait_all_vmem_visible();   // B fragments and bias are in registers (and the compiler knows it)
    int64_t tix = blockIdx.x;
    if (tix < tiles) dma_tile(tix, 0);
    wait_all_vmem();
    __syncthreads();
    int buf = 0;
    f32x16 accP[MT];            // accumulators of the previous tile, stored during this tile's MFMA phase
    int64_t r0P = 0, tixP = -1;
    for (; tix < tiles; tix += gridDim.x, buf ^= 1) {
        const int64_t r0 = tix * TR;
        unsigned bitsP = 0, newbits = 0;
        if (!EXCH && PIPE && tixP >= 0 && ep.mask_bits) {
            bitsP = ep.mask_bits[(tixP * WAVES + w) * 64 + lane];
            wait_all_vmem_visible();      // issued before this tile's DMA: does not wait for it
        }
        if (EXCH && XPIPE) {   // issued before the DMA: the end-of-phase vmcnt(0) covers them
#pragma unroll
            for (int it = 0; it < TR / 8; ++it) resN[it] = load_res(r0, it);
        }
        if (tix + gridDim.x < tiles) dma_tile(tix + gridDim.x, buf ^ 1);
        const float* at = lds + buf * (TR * K);
        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
        // A fragments are read one step ahead of the MFMAs that consume them
        float4 nxt[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
            nxt[m] = ld4(at + (row_base + 32 * m + col) * K + (((16 * half) ^ (col & 15)) << 2));
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int kc = st / 16, q = st % 16;
            float4 af[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) af[m] = nxt[m];
            if (st + 1 < STEPS) {
                const int kc2 = (st + 1) / 16, q2 = (st + 1) % 16;
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    nxt[m] = ld4(at + (row_base + 32 * m + col) * K + kc2 * 128 + (((16 * half + q2) ^ (col & 15)) << 2));
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].x, bf[kc][q].x, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].y, bf[kc][q].y, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].z, bf[kc][q].z, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].w, bf[kc][q].w, acc[m], 0, 0, 0);
            if (EXCH && XPIPE && tixP >= 0) {
                // previous tile's exchange epilogue rides inside this tile's MFMA phase:
                // steps 0..3 scatter the old accumulators into the exchange tile, a barrier, then one
                // row pair per step is finalised; everything is issued in the first half of the phase
                constexpr int WR = 16 * MT / 4;
                if (st < 4) {
#pragma unroll
                    for (int i = 0; i < WR; ++i) exch_write(xtile, accP, (st * WR + i) / 16, (st * WR + i) % 16);
                    if (st == 3) __syncthreads();
                } else if (st - 4 < TR / 8) {
                    exch_row(xtile, r0P, st - 4, resP[st - 4]);
                }
            }
            if (!EXCH && PIPE) {
                // previous tile's stores ride in the shadow of this tile's MFMAs (first steps only, so
                // they have drained by the time the DMA wait below needs vmcnt == 0)
                constexpr int PER = (16 * MT + 7) / 8;          // registers stored per step, 8 steps
                if (st < 8 && tixP >= 0) {
#pragma unroll
                    for (int i = 0; i < PER; ++i) {
                        const int r = st * PER + i;
                        if (r < 16 * MT) store_reg(accP, r0P, bitsP, r / 16, r % 16, newbits);
                    }
                }
            }
        }
        // the next tile's DMA had the whole MFMA phase to land; the stores above were issued early
        wait_all_vmem();
        if (!EXCH && PIPE) {
            if (ep.relu_bits && tixP >= 0) ep.relu_bits[(tixP * WAVES + w) * 64 + lane] = newbits;
#pragma unroll
            for (int m = 0; m < MT; ++m) accP[m] = acc[m];
            r0P = r0;
            tixP = tix;
            __syncthreads();   // every wave: DMA(t+1) landed, tile t fully read
        } else if (!EXCH) {
            // many waves per SIMD: the other waves' MFMA phases hide this epilogue
            const unsigned bits = ep.mask_bits ? ep.mask_bits[(tix * WAVES + w) * 64 + lane] : 0u;
            unsigned nb = 0;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) store_reg(acc, r0, bits, m, reg, nb);
            if (ep.relu_bits) ep.relu_bits[(tix * WAVES + w) * 64 + lane] = nb;
            __syncthreads();
        } else if (XPIPE) {
#pragma unroll
            for (int m = 0; m < MT; ++m) accP[m] = acc[m];
#pragma unroll
            for (int it = 0; it < TR / 8; ++it) resP[it] = resN[it];
            r0P = r0;
            tixP = tix;
            __syncthreads();   // DMA(t+1) landed everywhere, tile t read, exchange tile consumed
        } else {
            float* ex = lds + buf * (TR * K);   // consumed A buffer becomes the [TR][128] exchange tile
            __syncthreads();                    // all waves finished their fragment reads
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int rr = 32 * m + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                    float v = acc[m][reg] + bias;
                    if (ep.relu) v = fmaxf(v, 0.f);
                    ex[rr * 128 + 32 * wq + col] = v;
                }
            __syncthreads();
            float4 res[TR / 8];
            if (ep.residual) {
#pragma unroll
                for (int it = 0; it < TR / 8; ++it) {
                    int64_t row = r0 + wq * (TR / 4) + it * 2 + half;
                    if (row >= R) row = R - 1;
                    res[it] = ld4(ep.residual + row * N + col * 4);
                }
            }
#pragma unroll
            for (int it = 0; it < TR / 8; ++it) {
                const int rr = wq * (TR / 4) + it * 2 + half;
                const int64_t row = r0 + rr;
                const bool ok = row < R;
                float4 v = ld4(ex + rr * 128 + col * 4);
                if (ep.residual) v += res[it];
                if (ep.gamma == nullptr) {
                    if (ok) st4(y + row * N + col * 4, v);
                    continue;
                }
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
            __syncthreads();   // exchange tile consumed before the next DMA overwrites it
        }
    }
    if (EXCH && XPIPE && tixP >= 0) {   // drain: exchange epilogue of the last tile (block-uniform)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) exch_write(xtile, accP, m, reg);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < TR / 8; ++it) exch_row(xtile, r0P, it, resP[it]);
    }
    if (!EXCH && PIPE && tixP >= 0) {   // drain: epilogue of the last tile
        unsigned bitsP = ep.mask_bits ? ep.mask_bits[(tixP * WAVES + w) * 64 + lane] : 0u, newbits = 0;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) store_reg(accP, r0P, bitsP, m, reg, newbits);
        if (ep.relu_bits) ep.relu_bits[(tixP * WAVES + w) * 64 + lane] = newbits;
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

extern "C" size_t dg_row_gemm_mask_words(int64_t R, int K, int N) {
    // geometry of the direct-epilogue kernels (see the launch table in dg_row_gemm)
    if (K == 128 && N == 384) return static_cast<size_t>((R + 31) / 32) * 12 * 64;
    if (K == 128 && N == 128) return static_cast<size_t>((R + 63) / 64) * 8 * 64;   // upper bound over variants
    return 0;
}

extern "C" int dg_row_gemm(const float* a, const float* packed, float* y, int64_t R, int K, int N, const float* bias,
                           int relu, unsigned* relu_bits_out, const unsigned* mask_bits, const float* residual,
                           const float* gamma, const float* beta, float* mean, float* rstd, float* pre_ln,
                           float eps, dg_stream_t stream_) {
    if (!a || !packed || !y) return fail(DG_E_ARG, "dg_row_gemm: null pointer");
    if (R < 0 || !((K == 128 && (N == 128 || N == 384)) || (K == 384 && N == 128)))
        return fail(DG_E_SHAPE, "dg_row_gemm: unsupported K=%d N=%d (supported: 128x128, 128x384, 384x128)", K, N);
    if (gamma && (N != 128 || !beta || !mean || !rstd))
        return fail(DG_E_ARG, "dg_row_gemm: LayerNorm epilogue needs N == 128, beta, mean and rstd");
    const bool exch = gamma || residual;
    if (exch && (mask_bits || relu_bits_out))
        return fail(DG_E_ARG, "dg_row_gemm: bit masks cannot be combined with residual / LayerNorm epilogues");
    if ((mask_bits || relu_bits_out) && K != 128)
        return fail(DG_E_ARG, "dg_row_gemm: bit masks need K == 128");
    if (exch && N != 128) return fail(DG_E_ARG, "dg_row_gemm: residual / LayerNorm epilogues need N == 128");
    if (R == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Epilogue ep{bias, mask_bits, relu_bits_out, residual, gamma, beta, mean, rstd, pre_ln, eps, relu};
    ProfScope prof(DG_K_ROW_GEMM, stream);
#define LAUNCH(KC_, NG_, MG_, TR_, EX_, PIPE_, MINW_, PER_CU_) LAUNCHX(KC_, NG_, MG_, TR_, EX_, PIPE_, MINW_, PER_CU_, false)
#define LAUNCHX(KC_, NG_, MG_, TR_, EX_, PIPE_, MINW_, PER_CU_, XP_)                                               \
    {                                                                                                              \
        constexpr int lds_bytes = 2 * TR_ * KC_ * 128 * 4 + (XP_ ? TR_ * 128 * 4 : 0);                             \
        static const hipError_t attr = hipFuncSetAttribute(                                                        \
            reinterpret_cast<const void*>(&row_gemm_kernel<KC_, NG_, MG_, TR_, EX_, PIPE_, MINW_, XP_>),           \
            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);                                                \
        (void)attr;                                                                                                \
        const int64_t tiles = (R + TR_ - 1) / TR_;                                                                 \
        const int grid = static_cast<int>(tiles < 256 * PER_CU_ ? tiles : 256 * PER_CU_);                          \
        hipLaunchKernelGGL((row_gemm_kernel<KC_, NG_, MG_, TR_, EX_, PIPE_, MINW_, XP_>), dim3(grid),              \
                           dim3(NG_ * MG_ * 256), lds_bytes, stream, a, packed, y, R, ep);                         \
    }
    static const int variant = getenv("DG_GEMM_VARIANT") ? atoi(getenv("DG_GEMM_VARIANT")) : 0;
    if (K == 128 && N == 128) {
        if (exch) LAUNCH(1, 1, 1, 64, true, false, 1, 2)
        else if (variant == 1) LAUNCH(1, 1, 2, 64, false, false, 4, 2)   /* 8 waves, 4 waves/SIMD */
        else LAUNCH(1, 1, 1, 64, false, true, 1, 2)
    } else if (K == 128 && N == 384) {
        LAUNCH(1, 3, 1, 32, false, true, 1, 1)
    } else {
        if (!exch) LAUNCH(3, 1, 1, 32, false, true, 1, 1)             /* plain: pipelined direct stores */
        else if (variant == 2) LAUNCH(3, 1, 1, 32, true, false, 1, 1)  /* unpipelined exchange (A/B testing) */
        else LAUNCHX(3, 1, 1, 32, true, false, 1, 1, true)
    }
#undef LAUNCH
#undef LAUNCHX
    return check_launch("dg_row_gemm");
}
