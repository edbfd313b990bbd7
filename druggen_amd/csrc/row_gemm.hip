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
#include "row_gemm_n384.h"
#include "row_gemm_k384.h"
#include "traversal.h"

#include <cstring>
#include <type_traits>

// Developer builds only (-DDG_DBG=<bits>, loaded through DG_LIB; scripts/gemm_phases.py, scripts/gemm_variants.py;
// results in profiles/r02_gemm_*_phases.txt).  16: s_memtime phase stamps of workgroup 17 written to ep.rstd;
// K = 384 kernel ablations: 1 = no MFMAs, 2 = raw LDS writes instead of the fp16 split, 8 = no global fetch; K = 128
// kernels (direct epilogue): 1 = no MFMAs, 4 = no result stores, 8 = no global fetch (profiles/r04_gemm_ablation.txt).
// The shipped library is built with DG_DBG = 0: every such block folds away.
#ifndef DG_DBG
#define DG_DBG 0
#endif
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
    // LayerNorm-BACKWARD epilogue (LNB kernels): v = a B + residual is the gradient of a LayerNorm output; the row
    // leaves as dz = rstd (v gamma - mean(v gamma) - xhat mean(v gamma xhat)), xhat = (lnb_pre - mean) rstd, with
    // `gamma`, `mean`, `rstd` above read as that LayerNorm's saved parameters / statistics.  Per half-wave partial sums
    // of dgamma = sum_rows v xhat and dbeta = sum_rows v go to lnb_part[(block * 8 + half-wave)][2][128].
    const float* lnb_pre = nullptr;
    float* lnb_part = nullptr;
    // LayerNorm-backward PROLOGUE (LNA kernel): the A operand is dz = LayerNormBackward(a; lna_pre, mean, rstd, gamma),
    // computed by the producer waves on the rows they stream (a is that LayerNorm's OUTPUT gradient); dz is also
    // written to lna_dz [R,128] (the residual path and the weight gradient read it later), dgamma / dbeta partials go
    // to lnb_part in the layout above (half-waves of the four producer waves).
    const float* lna_pre = nullptr;
    float* lna_dz = nullptr;
    // Three Linears in one launch (q / k / v of an attention block, reference layers.py:111-113 and their backward):
    //   128 -> 384, S3 kernels: output slabs 4..7 / 8..11 go to y_alt[0] / y_alt[1] (row pitch 128, like y), their bias
    //   comes from bias_alt[0] / bias_alt[1];
    //   384 -> 128, A3 kernels: k-chunks 1 / 2 of the A operand are the [R,128] matrices a_alt[0] / a_alt[1].
    float* y_alt[2] = {nullptr, nullptr};
    const float* bias_alt[2] = {nullptr, nullptr};
    const float* a_alt[2] = {nullptr, nullptr};
    int reverse = 0;      // 128 -> 128 kernels without LayerNorm-backward stages: tiles in descending order (traversal.h)
};

// Sum over the 32 lanes of a half-wave, result in every lane.  DPP adds inside each 16-lane row
// (quad xor 1, quad xor 2, half-row mirror, row mirror), then the two row totals of the half-wave are
// read as scalars: no LDS-crossbar round trips (a __shfl_xor butterfly is five dependent ds_bpermute).
template <int CTRL>
__device__ __forceinline__ float dpp_sum_step(float x) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true);
    return x + __int_as_float(moved);
}
__device__ __forceinline__ float half_sum(float x) {
    x = dpp_sum_step<0xB1>(x);    // quad_perm [1,0,3,2]
    x = dpp_sum_step<0x4E>(x);    // quad_perm [2,3,0,1]
    x = dpp_sum_step<0x141>(x);   // row_half_mirror
    x = dpp_sum_step<0x140>(x);   // row_mirror: every lane holds its 16-lane row total
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
    return (threadIdx.x & 32) ? r2 + r3 : r0 + r1;
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


// =====================================================================================================
// bf16x6 row GEMM: fp32 operands split three ways into bf16 (a = a1 + a2 + a3, 8 mantissa bits each),
// the six cross products with i + j <= 4 run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
// The dropped terms are <= 2^-23 |a b|, and every kept product is exact, so the result is as accurate
// as an fp32 FMA chain (measured against fp64: rms 1.9e-8 / max 1.0e-7 of sum|a b| at K = 128, versus
// 2.4e-8 / 1.6e-7 for v_mfma_f32_32x32x2_f32 -- scripts/ubench/mfma_bf16_layout.hip and
// tests/test_hip_kernels.py) while the MFMA time drops to 6/16 of the fp32 pipe's: the GEMMs stop being
// MFMA-bound and run at the HBM roof of their activation streams.
//
//   * packed weights: for output slab t (32 columns), k-step ks (16 k) and plane p the B fragment of
//     lane (n = lane & 31, kg = lane >> 5) is the 8 bf16 { B[ks*16 + 8 kg + j][32 t + n] }_j.
//   * workgroup = 4 waves, wave w = slab w of a 128-column group (blockIdx.y), 64-row tiles, 2 workgroups
//     per CU.  A tile chunks (64 x 128 fp32) go HBM -> registers (prefetched one chunk ahead) -> split ->
//     three bf16 planes in LDS (272-byte row pitch: conflict-free ds_read_b128 fragments).
//   * K = 384 is three chunks accumulated into the same accumulators; the B fragments of a chunk (96
//     VGPRs) are re-read from L2 while the next chunk is being split.
//   * N = 384 is three column groups (gridDim.y): the groups of one tile sequence share an XCD
//     (gridDim.x is a multiple of 8), so two of the three reads of an A tile hit L2.
// LDS-DMA with a scalar chunk base and a per-lane 32-bit byte offset (3 instructions per 1 KiB piece)
__device__ __forceinline__ void dma16_saddr(const float* base, unsigned lane_off, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(lane_off), "s"(base), "s"(lds_addr)
                 : "memory", "m0");
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int kX6Pitch = 272;                 // bytes per LDS row of one bf16 plane (128 k): conflict-free b128 reads
constexpr int kX6Plane = kTR * kX6Pitch;      // one plane of a 64-row chunk
constexpr int kX6Buf = 3 * kX6Plane;          // three planes (also holds the 64 x 128 fp32 exchange tile)
constexpr int kX6Lds = 2 * kX6Buf;            // double-buffered
// fp16 two-plane variant (K = 128 kernels): two planes, then the 64 inverse row scales of the chunk
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int kH3Plane = kTR * kX6Pitch;      // same row pitch as the bf16 planes
constexpr int kH3Rs = 2 * kH3Plane;           // byte offset of float inv_row_scale[64] (past the 32 KiB exchange tile)
constexpr int kH3Buf = kH3Rs + kTR * 4;
constexpr int kH3Lds = 2 * kH3Buf;
static_assert(kH3Rs >= kTR * 128 * 4, "the fp32 exchange tile must not reach the row scales");

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = static_cast<__bf16>(x);
    const float r1 = x - static_cast<float>(h);
    m = static_cast<__bf16>(r1);
    l = static_cast<__bf16>(r1 - static_cast<float>(m));
}

__global__ void pack_weight_x6_kernel(const float* __restrict__ w, bf16x8* __restrict__ p, int rows, int cols, int mode,
                                      int n_tiles, int k_steps) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // one (slab, k-step, lane)
    if (idx >= n_tiles * k_steps * 64) return;
    const int lane = idx & 63, ks = (idx >> 6) % k_steps, t = (idx >> 6) / k_steps;
    const int n = 32 * t + (lane & 31);
    bf16x8 out[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = ks * 16 + 8 * (lane >> 5) + j;
        float v;
        if (mode == 0) v = (n < rows && k < cols) ? w[static_cast<size_t>(n) * cols + k] : 0.f;
        else v = (k < rows && n < cols) ? w[static_cast<size_t>(k) * cols + n] : 0.f;
        __bf16 h, m, l;
        split3(v, h, m, l);
        out[0][j] = h;
        out[1][j] = m;
        out[2][j] = l;
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) p[(static_cast<size_t>(t * k_steps + ks) * 3 + pl) * 64 + lane] = out[pl];
}

// ---- fp16 x 3: power-of-two scaled two-plane split -------------------------------------------------------
// A row a (and a weight column w) is multiplied by a power of two that brings its largest magnitude into
// [2^14, 2^15) -- exact -- and split as hi = fp16(x), lo = fp16(x - hi): the residual is exact in fp32 and
// hi + lo carries 22 significand bits of every element that is within 2^-18 of the row maximum (smaller
// elements lose bits only below 2^-39 of the maximum).  a.w ~= hi.hi + hi.lo + lo.hi with fp32
// accumulation (the dropped lo.lo term is 2^-22 relative): three MFMAs per product instead of the six of
// the bf16 three-plane split, two LDS planes instead of three, and the same fp32-class error (measured
// against fp64 in tests/test_hip_kernels.py).  The epilogue multiplies by the inverse scales.
__device__ __forceinline__ unsigned scale_exponent(float absmax) {   // biased exponent, clamped away from 0
    const unsigned e = __float_as_uint(absmax) >> 23;
    return e < 15u ? 15u : e;
}
__device__ __forceinline__ float scale_of(unsigned e) { return __uint_as_float((268u - e) << 23); }      // 2^(14 - (e - 127))
__device__ __forceinline__ float inv_scale_of(unsigned e) { return __uint_as_float((e - 14u) << 23); }
__device__ __forceinline__ float comp(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
// unsigned max steps (the ordering of |float| bit patterns): 0 is the identity, so the DPP move folds into v_max_u32
template <int CTRL>
__device__ __forceinline__ unsigned dpp_umax_step(unsigned x) {
    const unsigned moved = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, 0xF, 0xF, true));
    return x > moved ? x : moved;
}
// Producer side of the fp16x3 kernels: thread `pt` holds PFN float4s, the i-th belonging to tile row (pt + STRIDE i) >> 5
// at float4 column (pt & 31) (the 32 lanes of a half-wave hold one 128-wide row).  Row maximum -> power-of-two scale
// -> hi / lo fp16 planes + the inverse scale, written to the LDS image `pl`.  The work is staged across groups of
// eight float4s so that the eight independent dependency chains (DPP reduction, conversions) interleave instead of
// running one after the other: the producers' VALU latency is the critical path of these kernels.
template <int PFN, int STRIDE>
__device__ __forceinline__ void split_write_h3(const float4 (&set)[PFN], char* pl, int pt) {
    constexpr int G = PFN > 8 ? 2 : 4;   // the 16-float4 producers (two waves) are short of registers
    static_assert(PFN % G == 0 && STRIDE % 32 == 0, "whole groups, whole rows");
    char* const prow = pl + (pt >> 5) * kX6Pitch + (pt & 31) * 8;
    char* const prs = pl + kH3Rs + (pt >> 5) * 4;
#pragma unroll
    for (int g0 = 0; g0 < PFN; g0 += G) {
        unsigned m[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {   // |.| maximum of the four components: two VOP3 instructions with abs modifiers
            const float4& v = set[g0 + j];
            float t, u;
            asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(t) : "v"(v.x), "v"(v.y), "v"(v.z));
            asm("v_max_f32_e64 %0, |%1|, %2" : "=v"(u) : "v"(v.w), "v"(t));
            m[j] = __float_as_uint(u);
        }
#pragma unroll
        for (int j = 0; j < G; ++j) m[j] = dpp_umax_step<0xB1>(m[j]);
#pragma unroll
        for (int j = 0; j < G; ++j) m[j] = dpp_umax_step<0x4E>(m[j]);
#pragma unroll
        for (int j = 0; j < G; ++j) m[j] = dpp_umax_step<0x141>(m[j]);
#pragma unroll
        for (int j = 0; j < G; ++j) m[j] = dpp_umax_step<0x140>(m[j]);
        float sc[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const auto r = __builtin_amdgcn_permlane16_swap(m[j], m[j], false, false);
            const unsigned x = r[0] > r[1] ? r[0] : r[1];
            const unsigned e = max(x >> 23, 15u);
            m[j] = e;
            sc[j] = scale_of(e);
        }
        f32x2 xa[G], xb[G];
        f16x2 ha[G], hb[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const float4& v = set[g0 + j];
            xa[j] = f32x2{v.x, v.y} * sc[j];
            xb[j] = f32x2{v.z, v.w} * sc[j];
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            ha[j] = __builtin_convertvector(xa[j], f16x2);
            hb[j] = __builtin_convertvector(xb[j], f16x2);
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            xa[j] -= __builtin_convertvector(ha[j], f32x2);
            xb[j] -= __builtin_convertvector(hb[j], f32x2);
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            // row (pt >> 5) + (STRIDE / 32) (g0 + j): one base address per call, the rest are store-offset immediates
            const f16x2 la = __builtin_convertvector(xa[j], f16x2), lb = __builtin_convertvector(xb[j], f16x2);
            const u32x2 h = {__builtin_bit_cast(unsigned, ha[j]), __builtin_bit_cast(unsigned, hb[j])};
            const u32x2 l = {__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
            *reinterpret_cast<u32x2*>(prow + 0 * kH3Plane + (STRIDE / 32) * (g0 + j) * kX6Pitch) = h;
            *reinterpret_cast<u32x2*>(prow + 1 * kH3Plane + (STRIDE / 32) * (g0 + j) * kX6Pitch) = l;
        }
        if ((pt & 31) == 0) {   // one divergent block per group (branches between the stages would serialise the chains)
#pragma unroll
            for (int j = 0; j < G; ++j)
                *reinterpret_cast<float*>(prs + (STRIDE / 32) * (g0 + j) * 4) = inv_scale_of(m[j]);
        }
    }
}

// packed fp16x3 weights: [slab][k-step][plane 0/1][lane] x 8 fp16, then float inv_col_scale[32 * slabs] (one scale
// per output column over the whole contraction).  One 512-thread workgroup per (32-column slab, 128-wide k
// chunk): thread = (k-step, lane) keeps its 8 values in registers while the column maxima are reduced through LDS.
// (w1, w2 non-null: the matrix is the vertical stack [w; w1; w2] of three [128, cols] weights -- q / k / v of an
// attention block packed as ONE 128 -> 384 forward operand or ONE 384 -> 128 input-gradient operand)
__device__ __forceinline__ void pack_weight_h3_block(const float* __restrict__ w, const float* __restrict__ w1,
                                                     const float* __restrict__ w2, f16x8* __restrict__ p, int rows, int cols,
                                                     int mode, int n_tiles, int t, int kcb, int kchunks) {
    __shared__ unsigned part[16][32];
    const int k_steps = 8 * kchunks;
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int c = lane & 31, n = 32 * t + c;
    const int n_out = mode == 0 ? rows : cols, kdim = mode == 0 ? cols : rows;
    auto at = [&](int k) -> float {
        if (n >= n_out || k >= kdim) return 0.f;
        int r = mode == 0 ? n : k;
        const int c = mode == 0 ? k : n;
        const float* ws = w;
        if (w1) {
            ws = r < 128 ? w : (r < 256 ? w1 : w2);
            r &= 127;
        }
        return ws[static_cast<size_t>(r) * cols + c];
    };
    float v[8];
    unsigned mx = 0;
    for (int kc = 0; kc < kchunks; ++kc)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = at(kc * 128 + ks * 16 + 8 * (lane >> 5) + j);
            if (kc == kcb) v[j] = x;
            mx = max(mx, __float_as_uint(x) & 0x7FFFFFFFu);
        }
    part[2 * ks + (lane >> 5)][c] = mx;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = max(mx, part[i][c]);
    const unsigned e = scale_exponent(__uint_as_float(mx));
    const float sc = scale_of(e);
    f16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float xs = v[j] * sc;
        hi[j] = static_cast<_Float16>(xs);
        lo[j] = static_cast<_Float16>(xs - static_cast<float>(hi[j]));
    }
    const size_t slot = static_cast<size_t>(t) * k_steps + 8 * kcb + ks;
    p[(slot * 2 + 0) * 64 + lane] = hi;
    p[(slot * 2 + 1) * 64 + lane] = lo;
    if (kcb == 0 && ks == 0 && lane < 32)
        reinterpret_cast<float*>(p + static_cast<size_t>(n_tiles) * k_steps * 2 * 64)[n] = inv_scale_of(e);
}
__global__ __launch_bounds__(512) void pack_weight_h3_kernel(const float* __restrict__ w, const float* __restrict__ w1,
                                                             const float* __restrict__ w2, f16x8* __restrict__ p, int rows,
                                                             int cols, int mode, int n_tiles) {
    pack_weight_h3_block(w, w1, w2, p, rows, cols, mode, n_tiles, blockIdx.x, blockIdx.y, gridDim.y);
}
// Every weight of a network in ONE launch (after an optimizer step): blockIdx.z walks a device table of
// { w, packed, rows, cols, mode, w1, w2 } (int64 x 7; w1 = w2 = 0 unless the entry is a stack of three weights); blocks
// outside an entry's (slab, k-chunk) grid leave at once.
__global__ __launch_bounds__(512) void pack_weight_h3_batch_kernel(const long long* __restrict__ table) {
    const long long* e = table + 7 * static_cast<size_t>(blockIdx.z);
    const float* w = reinterpret_cast<const float*>(e[0]);
    f16x8* p = reinterpret_cast<f16x8*>(e[1]);
    const int rows = static_cast<int>(e[2]), cols = static_cast<int>(e[3]), mode = static_cast<int>(e[4]);
    const int n_out = mode == 0 ? rows : cols, k = mode == 0 ? cols : rows;
    const int nt = (n_out + 31) / 32, kc = (k + 127) / 128;
    if (static_cast<int>(blockIdx.x) >= nt || static_cast<int>(blockIdx.y) >= kc) return;      // block-uniform
    pack_weight_h3_block(w, reinterpret_cast<const float*>(e[5]), reinterpret_cast<const float*>(e[6]), p, rows, cols, mode,
                         nt, blockIdx.x, blockIdx.y, kc);
}

// exact three-way split of a float4 into bf16 planes by truncation: h = top 16 bits of x,
// m = top 16 bits of (x - h), l = top 16 bits of (x - h - m); every remainder is exact and
// h + m + l covers all 24 significand bits.  v_perm_b32 packs the high halves of a pair.
__device__ __forceinline__ void split4(const float4& v, u32x2& h, u32x2& m, u32x2& l) {
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const unsigned a0 = __float_as_uint(x[2 * i]), a1 = __float_as_uint(x[2 * i + 1]);
        h[i] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
        const float r0 = x[2 * i] - __uint_as_float(a0 & 0xFFFF0000u);
        const float r1 = x[2 * i + 1] - __uint_as_float(a1 & 0xFFFF0000u);
        const unsigned b0 = __float_as_uint(r0), b1 = __float_as_uint(r1);
        m[i] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        const float s0 = r0 - __uint_as_float(b0 & 0xFFFF0000u);
        const float s1 = r1 - __uint_as_float(b1 & 0xFFFF0000u);
        l[i] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    }
}

// Workgroup program: 8 waves, one workgroup per CU, persistent over 64-row tiles.
//   waves 4..7  producers: stream A chunks (64 rows x 128 k fp32) HBM -> registers (three chunks deep,
//               96 KiB in flight per CU) -> split -> three bf16 planes in LDS (double-buffered);
//   waves 0..3  consumers: wave w = 32-column slab w of the current 128-column group: 6 MFMAs per
//               (k-step, 32-row block) on fragments read from the planes, then the epilogue.
// One barrier per chunk separates "planes[c] written" from "planes[c] read".  K = 384: three chunks per
// tile accumulate into the same registers; N = 384: the three column groups of a tile run back to back on
// the same planes.  The B fragments of the current (group, chunk) live in 96 VGPRs; when they change
// per unit they are double-buffered (the next unit's arrive from L2 during this unit's MFMAs).
template <int KC, int NG, bool EXCH, int NC = 4, bool LNB = false, bool LNA = false, bool S3 = false>
__global__ __launch_bounds__(512, 2) void row_gemm_h3_kernel(const float* __restrict__ a, const f16x8* __restrict__ packed,
                                                             float* __restrict__ y, int64_t R, Epilogue ep) {
    // NC = 6 (N = 384 only): six consumer waves, each with two resident 32-column slabs (w and w + 6: 128 VGPRs
    // of fp16 fragments), two producer waves: no B-fragment stream from L2, the A tile is read once and every
    // A fragment read from LDS feeds two slabs.
    constexpr int NM = 8 - NC, PFN = 2048 / (64 * NM);   // producer waves, float4 per producer thread and chunk
    constexpr int K = KC * 128, KS = KC * 8, N = NC == 6 ? 384 : 128 * NG, UPT = KC * NG;
    constexpr int SLABS = NC == 6 ? 12 : 4 * NG;   // 32-column slabs of the whole output (bit-mask layout)
    constexpr int NS = NC == 6 ? 2 : 1;            // slabs per consumer wave
    static_assert(NC == 4 || (NC == 6 && KC == 1 && NG == 1 && !EXCH), "6 consumers: resident-B 128 -> 384 only");
    static_assert(KC == 1, "fp16x3: the row scale covers the whole contraction, K = 128 only");
    static_assert(!LNB || (EXCH && NC == 4 && NG == 1), "LayerNorm-backward epilogue: 128 -> 128 exchange kernel");
    static_assert(!LNA || (!EXCH && !LNB && NC == 4 && NG == 1), "LayerNorm-backward prologue: plain 128 -> 128 kernel");
    static_assert(!S3 || (NG == 3 && NC == 4 && !EXCH), "three outputs: the streaming-B 128 -> 384 kernel (one column group per Linear)");
    constexpr int DEPTH = LNA ? 2 : 3;      // register sets of A chunks the producers keep in flight
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* lds = smem_raw;                              // 2 x { planes[2][64 rows][272 B], inv_row_scale[64] }
    float* xb = reinterpret_cast<float*>(smem_raw + kH3Lds);   // direct epilogue: one 32 x 32 fp32 tile per consumer wave
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, col = lane & 31;
    const int64_t tiles = (R + kTR - 1) / kTR;
    if (static_cast<int64_t>(blockIdx.x) >= tiles) return;
    const int64_t my_tiles = (tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int64_t nchunks = my_tiles * KC;
    // Workgroups walk the three k-chunks / column groups of a tile in rotated orders (blockIdx % 3): at any
    // moment a third of the CUs streams each group's B fragments from L2 instead of all CUs the same 96 KiB.
    const int rot = static_cast<int>(blockIdx.x % 3);
    const int64_t padded = (nchunks + DEPTH - 1) / DEPTH * DEPTH;      // the producers run whole groups of DEPTH iterations
    // tiles round-robin over the workgroups, ascending or (ep.reverse, traversal.h) descending
    auto tile_at = [&](int64_t ti) {
        const int64_t t = blockIdx.x + ti * gridDim.x;
        if constexpr (LNB || LNA || S3 || NC != 4) return t;      // (partial sums per workgroup / node-level kernels: always ascending)
        else return ep.reverse ? tiles - 1 - t : t;
    };
    auto tile_of = [&](int64_t chunk) { return tile_at(chunk / KC); };

    if (w >= NC) {
        // ------------------------------------------------------------------ producers
        const int pt = threadIdx.x - 64 * NC;
        if constexpr (LNA) {
            // ---- LayerNorm backward on the way in.  Thread pt holds float4 column (pt & 31) of tile rows
            // (pt >> 5) + 8 i: the 32 lanes of a half-wave own whole rows, so both row means are DPP sums.  Two register
            // sets of dy rows; ONE set of pre-LayerNorm rows and row statistics, requested as soon as the previous
            // chunk's rows are finished (a chunk period ahead of their use: with four [R,128] streams per tile that
            // period is ~2x the plain kernel's).  Addresses past the end (last tile, padded iterations) are CLAMPED,
            // never branched around: a clamped row recomputes the dz of the row it was clamped to and stores the same
            // values again; only the dgamma / dbeta sums are masked.
            constexpr int K_ = 128;
            float4 dyr[2][PFN], prr[PFN];
            float st;      // lane c < 8: mean of this half-wave's row c ; 8 <= c < 16: rstd of row c - 8
            const float4 gam = ld4(ep.gamma + (pt & 31) * 4);
            float4 dgam = f4(0.f), dbet = f4(0.f);
            const unsigned voff0 = static_cast<unsigned>(pt >> 5) * (K_ * 4) + static_cast<unsigned>(pt & 31) * 16;
            auto offsets = [&](int64_t chunk, int64_t& r0, unsigned& lim) {
                r0 = tile_of(chunk) * kTR;
                const int64_t last = R - 1 - r0;   // >= 0
                lim = last >= kTR - 1 ? 0xFFFFFFFFu
                                      : static_cast<unsigned>(last) * (K_ * 4) + static_cast<unsigned>(pt & 31) * 16;
            };
            auto fetch_dy = [&](auto s_tag, int64_t chunk) {
                constexpr int s = decltype(s_tag)::value;
                if (chunk > nchunks - 1) chunk = nchunks - 1;
                int64_t r0;
                unsigned lim;
                offsets(chunk, r0, lim);
                const char* ba = reinterpret_cast<const char*>(a) + r0 * (K_ * 4);
#pragma unroll
                for (int i = 0; i < PFN; ++i) {
                    unsigned off = voff0 + i * (2 * NM * K_ * 4);
                    off = off < lim ? off : lim;
                    dyr[s][i] = ld4(reinterpret_cast<const float*>(ba + off));
                }
            };
            auto fetch_pre = [&](int64_t chunk) {
                if (chunk > nchunks - 1) chunk = nchunks - 1;
                int64_t r0;
                unsigned lim;
                offsets(chunk, r0, lim);
                const char* bp = reinterpret_cast<const char*>(ep.lna_pre) + r0 * (K_ * 4);
#pragma unroll
                for (int i = 0; i < PFN; ++i) {
                    unsigned off = voff0 + i * (2 * NM * K_ * 4);
                    off = off < lim ? off : lim;
                    prr[i] = ld4(reinterpret_cast<const float*>(bp + off));
                }
                int64_t row = r0 + (pt >> 5) + 8 * (pt & 7);
                if (row > R - 1) row = R - 1;
                st = (pt & 8) ? ep.rstd[row] : ep.mean[row];
            };
            auto write2 = [&](auto s_tag, int64_t chunk) {
                constexpr int s = decltype(s_tag)::value;
                const float live = chunk < nchunks ? 1.f : 0.f;
                const int buf = static_cast<int>(chunk & 1);
                if (chunk > nchunks - 1) chunk = nchunks - 1;
                int64_t r0;
                unsigned lim;
                offsets(chunk, r0, lim);
                char* bz = reinterpret_cast<char*>(ep.lna_dz) + r0 * (K_ * 4);
                const int64_t last = R - 1 - r0;
                // 32-bit row test from a lane term that is opaque per chunk (as loop invariants the eight 64-bit row
                // numbers of a thread are hoisted and spilled)
                const unsigned lastu = static_cast<unsigned>(last < kTR - 1 ? last : kTR - 1);
                unsigned rowt = static_cast<unsigned>(pt) >> 5;
                asm volatile("" : "+v"(rowt));
                const int sti = __float_as_int(st);
                // four rows at a time -- LayerNorm backward, dz store, split into the LDS planes -- fenced: left alone,
                // the scheduler interleaves all eight rows' chains and spills
#pragma unroll
                for (int i0 = 0; i0 < PFN; i0 += 4) {
                    float4 dzs[4];
#pragma unroll
                    for (int i = i0; i < i0 + 4; ++i) {
                        // statistics of row i of this half-wave: scalar reads of both half-waves' lanes, then a select
                        const int mu0 = __builtin_amdgcn_readlane(sti, i), mu1 = __builtin_amdgcn_readlane(sti, 32 + i);
                        const int rs0 = __builtin_amdgcn_readlane(sti, 8 + i), rs1 = __builtin_amdgcn_readlane(sti, 40 + i);
                        const float mu = __int_as_float(half ? mu1 : mu0), rs = __int_as_float(half ? rs1 : rs0);
                        const float4 v = dyr[s][i];
                        const float4 xh = rs * (prr[i] - f4(mu));
                        const float4 u = v * gam;
                        const float c1 = half_sum((u.x + u.y) + (u.z + u.w)) * (1.0f / 128.0f);
                        const float c2 = half_sum((u.x * xh.x + u.y * xh.y) + (u.z * xh.z + u.w * xh.w)) * (1.0f / 128.0f);
                        const float4 dz = rs * (u - f4(c1) - c2 * xh);
                        const float ok = rowt + 8u * i <= lastu ? live : 0.f;
                        const float4 vm = ok * v;
                        dgam = fma4(vm, xh, dgam);
                        dbet += vm;
                        dzs[i - i0] = dz;
                        unsigned off = voff0 + i * (2 * NM * K_ * 4);
                        off = off < lim ? off : lim;
                        st4(reinterpret_cast<float*>(bz + off), dz);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    split_write_h3<4, 64 * NM>(dzs, lds + buf * kH3Buf, pt + i0 * (64 * NM));
                    __builtin_amdgcn_sched_barrier(0);
                }
                fetch_pre(chunk + 1);      // the pre rows / statistics of the next chunk
            };
            constexpr std::integral_constant<int, 0> S0{};
            constexpr std::integral_constant<int, 1> S1{};
            fetch_dy(S0, 0);
            fetch_pre(0);
            fetch_dy(S1, 1);
            write2(S0, 0);
            fetch_dy(S0, 2);
            __syncthreads();
            for (int64_t c = 0; c < padded; c += 2) {
                write2(S1, c + 1);
                fetch_dy(S1, c + 3);
                __syncthreads();
                write2(S0, c + 2);
                fetch_dy(S0, c + 4);
                __syncthreads();
            }
            float* pp = ep.lnb_part + (static_cast<size_t>(blockIdx.x) * 8 + 2 * (w - NC) + half) * 256 + (pt & 31) * 4;
            st4(pp, dgam);
            st4(pp + 128, dbet);
            return;
        }
        float4 pf[3][PFN];
        // Straight-line on purpose (chunk indices are clamped instead of guarded): with branches around
        // the loads hipcc can no longer count how many younger loads may stay in flight and drains
        // the whole queue (vmcnt(0)) before every split, which serialises the stream with HBM latency.
        // uniform 64-bit tile base + a 32-bit per-lane byte offset; rows past the end of a partial tile are clamped
        // on the offset (row-major: the row term dominates the comparison)
        const unsigned voff0 = static_cast<unsigned>(pt >> 5) * (K * 4) + static_cast<unsigned>(pt & 31) * 16;
        auto fetch = [&](float4 (&set)[PFN], int64_t chunk) {
            if (chunk > nchunks - 1) chunk = nchunks - 1;
            const int64_t r0 = tile_of(chunk) * kTR;
            const char* base = reinterpret_cast<const char*>(a) + r0 * (K * 4);
            const int64_t last = R - 1 - r0;   // >= 0
            const unsigned lim = last >= kTR - 1 ? 0xFFFFFFFFu
                                                 : static_cast<unsigned>(last) * (K * 4) + static_cast<unsigned>(pt & 31) * 16;
#pragma unroll
            for (int i = 0; i < PFN; ++i) {
                unsigned off = voff0 + i * (2 * NM * K * 4);   // 64 NM threads = 2 NM rows per step
                off = off < lim ? off : lim;
                if (DG_DBG & 8) set[i] = f4(static_cast<float>(off & 7));
                else set[i] = ld4(reinterpret_cast<const float*>(base + off));
            }
        };
        auto write = [&](const float4 (&set)[PFN], int64_t chunk) {   // chunks past the end land in the idle buffer
            split_write_h3<PFN, 64 * NM>(set, lds + (chunk & 1) * kH3Buf, pt);
        };
        auto end_of_iteration = [&](int64_t c) {
            if (EXCH && c % KC == KC - 1 && c < nchunks) {   // the consumers' exchange epilogue: two more barriers
                __syncthreads();
                __syncthreads();
            }
            __syncthreads();
        };
#if DG_DBG & 16
        unsigned long long tl = __builtin_amdgcn_s_memtime(), tW = 0, tF = 0, tB = 0, t00 = tl;
#define GSTAMP(x) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); x += t_ - tl; tl = t_; }
#else
#define GSTAMP(x)
#endif
        fetch(pf[0], 0);
        fetch(pf[1], 1);
        fetch(pf[2], 2);
        write(pf[0], 0);
        fetch(pf[0], 3);
        __syncthreads();
        for (int64_t c = 0; c < padded; c += 3) {
            // iteration c writes chunk c + 1 (the consumers are on chunk c) and refills its register set
            write(pf[1], c + 1); GSTAMP(tW)
            fetch(pf[1], c + 4); GSTAMP(tF)
            end_of_iteration(c); GSTAMP(tB)
            write(pf[2], c + 2); GSTAMP(tW)
            fetch(pf[2], c + 5); GSTAMP(tF)
            end_of_iteration(c + 1); GSTAMP(tB)
            write(pf[0], c + 3); GSTAMP(tW)
            fetch(pf[0], c + 6); GSTAMP(tF)
            end_of_iteration(c + 2); GSTAMP(tB)
        }
#if DG_DBG & 16
        if (lane == 0 && blockIdx.x == 17 && ep.rstd) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(ep.rstd) + 8 * w;
            o[0] = tW; o[1] = tF; o[2] = tB; o[3] = 0; o[4] = __builtin_amdgcn_s_memtime() - t00; o[5] = my_tiles;
        }
#endif
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const unsigned relu_sel = ep.relu ? 0xFFFFFFFFu : 0u;
    // B fragments of the current unit in two halves (k-steps 0-3 and 4-7, 32 VGPRs each).  When the
    // unit changes (K = 384 or N = 384) the halves form a ring: half 1 of this unit is requested from L2
    // at the start of the unit, half 0 of the next unit after the first four k-steps.
    f16x8 bset[2][NS][2][4];
    auto slab_of = [&](int g, int s) { return NC == 6 ? w + 6 * s : 4 * g + w; };
    auto load_b = [&](f16x8 (&bfr)[NS][2][4], int g, int kc, int h) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const f16x8* wp = packed + static_cast<size_t>(slab_of(g, s)) * KS * 2 * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int p = 0; p < 2; ++p) bfr[s][p][ks] = wp[((kc * 8 + 4 * h + ks) * 2 + p) * 64];
        }
    };
    const float* inv_cs = reinterpret_cast<const float*>(packed + static_cast<size_t>(SLABS) * KS * 2 * 64);
    float bias_g[NG * NS], cs_g[NG * NS];   // NS > 1 only with NG == 1
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int n = 32 * slab_of(g, s) + col;
            if (S3) {      // slab / 4 selects the Linear, n & 127 its output channel
                const float* bp = n < 128 ? ep.bias : ep.bias_alt[(n >> 7) - 1];
                bias_g[g * NS + s] = bp ? bp[n & 127] : 0.f;
            } else {
                bias_g[g * NS + s] = ep.bias ? ep.bias[n] : 0.f;
            }
            cs_g[g * NS + s] = inv_cs[n];
        }
    float4 res[8];   // EXCH: residual rows of the tile, requested before its MFMAs
    float4 lpre[LNB ? 4 : 1];      // LNB: rows of the saved pre-LayerNorm sum (four at a time)
    float4 dgam = f4(0.f), dbet = f4(0.f);      // LNB: this lane's four columns, summed over every row it finishes
    f32x16 acc[NS][2];
#if DG_DBG & 16
    unsigned long long tl = 0, tM = 0, tE = 0, tB = 0, t00 = 0, tC = 0, tS = 0;
#endif
    // one unit = one (chunk, column group): MFMAs on planes[chunk & 1] with `bfr`, epilogue when the
    // contraction is complete; `bnext` receives the B fragments of the following unit meanwhile
    auto unit = [&](int64_t ti, int pos_kc, int pos_g) {   // positions within the tile; values are rotated
        const int kc = pos_kc, kcv = KC > 1 ? (pos_kc + rot) % KC : 0;
        const int g = NG > 1 ? (pos_g + rot) % NG : 0;
        const int64_t chunk = ti * KC + pos_kc;
        const int64_t tix = tile_at(ti);
        const int64_t r0 = tix * kTR;
        if (UPT > 1) load_b(bset[1], g, kcv, 1);
        // ReLU mask words of this unit's slabs, requested before the MFMA phase and unconditionally (a null mask reads a
        // valid dummy word): a load inside the epilogue would be awaited with vmcnt(0) -- i.e. behind the stores
        // the previous slab has just issued -- once per slab, whether or not a mask is present
        constexpr bool BITS_IN = !EXCH && (NG == 3 || NC == 6);
        unsigned mask_in[NS];
        if (BITS_IN) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const size_t bix = (static_cast<size_t>(tix) * SLABS + slab_of(g, s)) * 64 + lane;
                const unsigned* mp = ep.mask_bits ? ep.mask_bits + bix : reinterpret_cast<const unsigned*>(packed) + lane;
                mask_in[s] = *mp;
            }
        }
        if (kc == 0) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[s][m][i] = 0.f;
        }
        // rows of this wave in the exchange epilogue: w * 16 + it * 2 + half, clamped to the last row of a partial tile
        // in 32 bits: uniform 64-bit tile base + a per-lane byte offset derived HERE from lane terms that are opaque per
        // tile (as loop invariants the per-row pointers are hoisted out of the tile loop and spilled)
        unsigned colv = static_cast<unsigned>(col), halfv = static_cast<unsigned>(half);
        if (EXCH) asm volatile("" : "+v"(colv), "+v"(halfv));
        const unsigned lastr = static_cast<unsigned>(R - 1 - r0 < kTR - 1 ? R - 1 - r0 : kTR - 1);
        auto row_off = [&](int it) {
            unsigned rr = static_cast<unsigned>(w * 16 + it * 2) + halfv;
            rr = rr < lastr ? rr : lastr;
            return rr * 512u + colv * 16u;
        };
        if (EXCH && kc == KC - 1 && ep.residual) {
            const char* rbase = reinterpret_cast<const char*>(ep.residual) + r0 * 512;
#pragma unroll
            for (int it = 0; it < 8; ++it) res[it] = ld4(reinterpret_cast<const float*>(rbase + row_off(it)));
        }
        const char* pl = lds + (chunk & 1) * kH3Buf;
        // fragments of step ks + 1 are requested before the MFMAs of step ks; consecutive MFMAs alternate
        // between the two row blocks (independent accumulators)
        f16x8 af[2][2][2];
        auto frags = [&](int ks, f16x8 (&dst)[2][2]) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    dst[m][p] = *reinterpret_cast<const f16x8*>(pl + p * kH3Plane + (32 * m + col) * kX6Pitch +
                                                                (ks * 16 + 8 * half) * 2);
        };
        // inverse scales of this lane's accumulator rows: 32 m + 8 q + 4 half + (0..3), times the column's
        auto row_scales = [&](float4 (&rs)[4], int m, float cs) {
            const float* rsp = reinterpret_cast<const float*>(pl + kH3Rs);
#pragma unroll
            for (int q = 0; q < 4; ++q) rs[q] = cs * ld4(rsp + 32 * m + 8 * q + 4 * half);
        };
        frags(0, af[0]);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks + 1 < 8) frags(ks + 1, af[(ks + 1) & 1]);
            if (UPT > 1 && ks == 4) {   // half 0 is consumed: request the next unit's (same tile or next)
                const int un = (pos_kc * NG + pos_g + 1) % UPT;   // next position: same tile, or the next tile's first
                load_b(bset[0], NG > 1 ? (un + rot) % NG : 0, KC > 1 ? (un + rot) % KC : 0, 0);
            }
            const f16x8(&f)[2][2] = af[ks & 1];
            const f16x8(&bfr)[NS][2][4] = bset[ks >> 2];
            constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};   // lo.hi, hi.lo, hi.hi: smallest terms first
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        if (DG_DBG & 1) asm volatile("" ::"v"(f[m][TA[t]]), "v"(bfr[s][TB[t]][ks & 3]));
                        else acc[s][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[m][TA[t]], bfr[s][TB[t]][ks & 3], acc[s][m], 0, 0, 0);
                    }
        }
        GSTAMP(tM)
        // The ReLU mask words were requested before the MFMA phase and have long arrived, but hipcc placed the wait for
        // mask word s at its first use -- behind the stores of slab s - 1 -- as vmcnt(0): the wave sat out an HBM write
        // round trip per tile.  Waiting HERE (nothing is outstanding but last tile's stores, issued a whole MFMA phase
        // ago) tells the compiler's scoreboard that every earlier load is back: the slabs' stores then stream without a
        // wait between them (128 -> 384: 310 -> 296 us at R = 518 400, A/B in one call).
        if (BITS_IN) wait_all_vmem_visible();
        if (!EXCH && kc == KC - 1) {
            // direct epilogue from the accumulator layout, one (slab, 32-row block) at a time
            constexpr bool BITS = NG == 3 || NC == 6;   // ReLU bit masks in / out: only the fc1-shaped launches use them
            const bool full = r0 + kTR <= R;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int slab = slab_of(g, s);
                const int n = 32 * slab + col;
                const float bias = bias_g[g * NS + s];
                const size_t bix = (static_cast<size_t>(tix) * SLABS + slab) * 64 + lane;
                unsigned bits = 0xFFFFFFFFu, newbits = 0;
                if (BITS) bits = ep.mask_bits ? mask_in[s] : 0xFFFFFFFFu;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    float out[16];
                    float4 rs[4];
                    row_scales(rs, m, cs_g[g * NS + s]);
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        float v = fmaf(acc[s][m][reg], comp(rs[reg >> 2], reg & 3), bias);
                        if (BITS) {
                            newbits |= (v > 0.f ? 1u : 0u) << (16 * m + reg);
                            v = __uint_as_float((__float_as_uint(fmaxf(v, 0.f)) & relu_sel) | (__float_as_uint(v) & ~relu_sel));
                            v = (bits >> (16 * m + reg)) & 1u ? v : 0.f;
                        } else if (ep.relu) {
                            v = fmaxf(v, 0.f);
                        }
                        out[reg] = v;
                    }
                    GSTAMP(tC)
                    // Leave as whole 128-byte row segments: the 32 x 32 block goes through a wave-private 4 KiB LDS tile
                    // (no barrier: one wave, LDS operations stay in order) and out as four 16-byte-per-lane stores, each
                    // covering 8 rows -- instead of sixteen 4-byte-per-lane stores, which cost the consumers as much
                    // time as their MFMAs (profiles/r02_gemm_n384_phases.txt).
                    {
                        float* xw = xb + w * (32 * 32);
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) xw[((reg & 3) + 8 * (reg >> 2) + 4 * half) * 32 + col] = out[reg];
                        __builtin_amdgcn_wave_barrier();   // compiler-level ordering only: other lanes' writes are read below
                        float4 rowv[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) rowv[i] = ld4(xw + (8 * i + (lane >> 3)) * 32 + 4 * (lane & 7));
                        __builtin_amdgcn_wave_barrier();
                        constexpr int LDY = S3 ? 128 : N;      // three [R,128] outputs, or one [R,N]
                        float* yb = S3 ? (slab < 4 ? y : ep.y_alt[(slab >> 2) - 1]) : y;
                        float* yr = yb + (r0 + 32 * m + (lane >> 3)) * LDY + 32 * (S3 ? (slab & 3) : slab) + 4 * (lane & 7);
                        if (DG_DBG & 4) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(rowv[i].x), "v"(rowv[i].y), "v"(rowv[i].z), "v"(rowv[i].w));
                        } else if (full) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) st4(yr + static_cast<size_t>(8 * i) * LDY, rowv[i]);
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (r0 + 32 * m + 8 * i + (lane >> 3) < R) st4(yr + static_cast<size_t>(8 * i) * LDY, rowv[i]);
                        }
                    }
                    GSTAMP(tS)
                }
                if (BITS && ep.relu_bits) ep.relu_bits[bix] = newbits;
            }
        }
        if (EXCH && kc == KC - 1) {
            // exchange through the consumed planes so that each half-wave finalises whole 512-byte rows
            float* ex = reinterpret_cast<float*>(lds + (chunk & 1) * kH3Buf);
            const float bias = bias_g[0];
            float4 rs[2][4];
            row_scales(rs[0], 0, cs_g[0]);
            row_scales(rs[1], 1, cs_g[0]);
            if (LNB) {      // rows of the saved pre-LayerNorm sum: requested here (the fragment registers are free now),
                            // they arrive during the exchange
                const char* pbase = reinterpret_cast<const char*>(ep.lnb_pre) + r0 * 512;
#pragma unroll
                for (int it = 0; it < 4; ++it) lpre[LNB ? it : 0] = ld4(reinterpret_cast<const float*>(pbase + row_off(it)));
            }
            __syncthreads();   // every consumer has finished its fragment reads
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int rr = 32 * m + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                    float v = fmaf(acc[0][m][reg], comp(rs[m][reg >> 2], reg & 3), bias);
                    if (ep.relu) v = fmaxf(v, 0.f);
                    ex[rr * 128 + 32 * w + col] = v;
                }
            __syncthreads();
            // rows of this wave: w * 16 + it * 2 + half.  `CHECK` only for the tail tile (uniform branch)
            auto finish_rows_lnb = [&](auto check_tag) {
                constexpr bool CHECK = decltype(check_tag)::value;
                float* yrow = reinterpret_cast<float*>(reinterpret_cast<char*>(y + r0 * 128) + ((w * 16 + halfv) * 512u + colv * 16u));
                const float4 gam = ld4(reinterpret_cast<const float*>(reinterpret_cast<const char*>(ep.gamma) + colv * 16u));
                // two groups of four rows (register budget): the second group's rows of the saved pre-LayerNorm sum are
                // requested before the first group is finished
#pragma unroll
                for (int hg = 0; hg < 2; ++hg) {
                    float4 pn[4];
                    if (hg == 0) {
                        const char* pbase = reinterpret_cast<const char*>(ep.lnb_pre) + r0 * 512;
#pragma unroll
                        for (int j = 0; j < 4; ++j) pn[j] = ld4(reinterpret_cast<const float*>(pbase + row_off(4 + j)));
                    }
                    float mu4[4], rs4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {      // one address per half-wave: broadcast loads
                        unsigned rr = static_cast<unsigned>(w * 16 + (4 * hg + j) * 2) + halfv;
                        rr = rr < lastr ? rr : lastr;
                        mu4[j] = ep.mean[r0 + rr];
                        rs4[j] = ep.rstd[r0 + rr];
                    }
                    float4 yv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int it = 4 * hg + j, rr = w * 16 + it * 2 + half;
                        float4 v = ld4(reinterpret_cast<const float*>(reinterpret_cast<const char*>(ex) + ((w * 16 + it * 2 + halfv) * 512u + colv * 16u)));
                        if (ep.residual) v += res[it];
                        if (CHECK && static_cast<unsigned>(w * 16 + it * 2) + halfv > lastr) v = f4(0.f);
                        const float4 xh = rs4[j] * (lpre[LNB ? j : 0] - f4(mu4[j]));
                        const float4 u = v * gam;
                        const float c1 = half_sum((u.x + u.y) + (u.z + u.w)) * (1.0f / 128.0f);
                        const float c2 = half_sum((u.x * xh.x + u.y * xh.y) + (u.z * xh.z + u.w * xh.w)) * (1.0f / 128.0f);
                        yv[j] = rs4[j] * (u - f4(c1) - c2 * xh);
                        dgam = fma4(v, xh, dgam);
                        dbet += v;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int it = 4 * hg + j, rr = w * 16 + it * 2 + half;
                        if (!CHECK || static_cast<unsigned>(w * 16 + it * 2) + halfv <= lastr) st4(yrow + it * 256, yv[j]);
                        if (hg == 0) lpre[LNB ? j : 0] = pn[j];
                    }
                }
            };
            auto finish_rows = [&](auto check_tag) {
                constexpr bool CHECK = decltype(check_tag)::value;
                float* yrow = y + (r0 + w * 16 + half) * 128 + col * 4;
                float* prow = ep.pre ? ep.pre + (r0 + w * 16 + half) * 128 + col * 4 : nullptr;
                float4 gam = f4(0.f), bet = f4(0.f);
                if (ep.gamma) {
                    gam = ld4(ep.gamma + col * 4);
                    bet = ld4(ep.beta + col * 4);
                }
                // all results of the tile are computed into distinct registers first and stored afterwards:
                // a store-data register that is rewritten while its store is in flight costs a vmcnt(0)
                float4 yv[8], pv[8];
                float mu8[8], rs8[8];
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int rr = w * 16 + it * 2 + half;
                    float4 v = ld4(ex + rr * 128 + col * 4);
                    if (ep.residual) v += res[it];
                    pv[it] = v;
                    if (ep.gamma) {
                        const float mu = half_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 128.0f);
                        const float4 d = v - f4(mu);
                        const float var = half_sum((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * (1.0f / 128.0f);
                        const float rs = rsqrtf(var + ep.eps);
                        v = fma4(rs * d, gam, bet);
                        mu8[it] = mu;
                        rs8[it] = rs;
                    }
                    yv[it] = v;
                }
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int rr = w * 16 + it * 2 + half;
                    if (!CHECK || r0 + rr < R) {
                        st4(yrow + it * 256, yv[it]);
                        if (prow && ep.gamma) st4(prow + it * 256, pv[it]);
                    }
                }
                if (ep.gamma && col == 0) {   // one divergent block per tile for the row statistics
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int rr = w * 16 + it * 2 + half;
                        if (!CHECK || r0 + rr < R) {
                            ep.mean[r0 + rr] = mu8[it];
                            ep.rstd[r0 + rr] = rs8[it];
                        }
                    }
                }
            };
            if constexpr (LNB) {
                if (r0 + kTR <= R) finish_rows_lnb(std::false_type{});
                else finish_rows_lnb(std::true_type{});
            } else {
                if (r0 + kTR <= R) finish_rows(std::false_type{});
                else finish_rows(std::true_type{});
            }
        }
        GSTAMP(tE)
        if (pos_g == NG - 1) __syncthreads();   // end of this chunk's iteration
        GSTAMP(tB)
    };
    load_b(bset[0], NG > 1 ? rot : 0, KC > 1 ? rot : 0, 0);
    if (UPT == 1) load_b(bset[1], 0, 0, 1);
    __syncthreads();   // chunk 0 is in planes[0]
#if DG_DBG & 16
    tl = t00 = __builtin_amdgcn_s_memtime();
#endif
    for (int64_t ti = 0; ti < my_tiles; ++ti) {
        if constexpr (UPT == 1) {
            unit(ti, 0, 0);
        } else {   // rolled on purpose: three inlined copies let hipcc hoist loads across units and spill
#pragma unroll 1
            for (int u = 0; u < UPT; ++u) unit(ti, KC > 1 ? u : 0, NG > 1 ? u : 0);
        }
    }
    for (int64_t c = nchunks; c < padded; ++c) __syncthreads();   // match the producers' padded iterations
    if (LNB) {      // per half-wave partials of dgamma / dbeta (reduced in a fixed order by ln_finish)
        float* pp = ep.lnb_part + (static_cast<size_t>(blockIdx.x) * 8 + 2 * w + half) * 256 + col * 4;
        st4(pp, dgam);
        st4(pp + 128, dbet);
    }
#if DG_DBG & 16
    if (lane == 0 && blockIdx.x == 17 && ep.rstd) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(ep.rstd) + 8 * w;
        o[0] = tM; o[1] = tE; o[2] = tB; o[3] = 0; o[4] = __builtin_amdgcn_s_memtime() - t00; o[5] = my_tiles;
        o[6] = tC; o[7] = tS;
    }
#endif
#undef GSTAMP
}


// ------------------------------------------------------------------------------------------------------
// K = 384 -> N = 128 (fc2 forward, fc1 input gradient): the contraction is three 128-wide chunks whose B
// fragments cannot all be resident (3 x 96 VGPRs), and re-reading 96 KiB from L2 for every 64-row unit
// bounds the MFMA phase (every CU pulls the same bytes: ~15-19 TB/s in aggregate).  This program
//   * walks two tiles at a time, position major -- (t0,0) (t1,0) (t0,1) (t1,1) (t0,2) (t1,2) -- so one set
//     of B fragments serves two units (two accumulator sets);
//   * refills the fragments by hand: the load of a k-step's registers is issued right after their last
//     MFMA in the second unit of a position, the first unit of the next position waits per k-step with the
//     exact in-order count (3 (7 - ks) younger loads) -- the consumers issue no other VMEM operation;
//   * keeps every global access in waves 4..7: A chunks HBM -> registers (three deep) -> split -> bf16
//     planes, and the finished tile's epilogue (exchange tile in LDS -> residual / LayerNorm -> 512-byte
//     row stores).  Their loop is one straight line per tile pair so that hipcc can count loads in flight.
// Barriers: B(c) ends every chunk; A(c) precedes the exchange-tile write of a chunk that completes a tile.
__device__ __forceinline__ void load_b128_async(f16x8& dst, const f16x8* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_b_refill(f16x8& b0, f16x8& b1) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b0), "+v"(b1) : "n"(N));
}

template <bool EXCH, bool A3 = false>
__global__ __launch_bounds__(512, 2) void row_gemm_h3_k384_kernel(const float* __restrict__ a, const f16x8* __restrict__ packed,
                                                                  float* __restrict__ y, int64_t R, Epilogue ep) {
    constexpr int KC = 3, K = 384, KS = 24;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* lds = smem_raw;                                          // 2 x { planes[2][64 rows][272 B], inv_row_scale[64] }
    float* ex = reinterpret_cast<float*>(smem_raw + 2 * kH3Buf);   // exchange tile [64][128] fp32
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, col = lane & 31;
    const int64_t tiles = (R + kTR - 1) / kTR;
    if (static_cast<int64_t>(blockIdx.x) >= tiles) return;
    const int ntl = static_cast<int>((tiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
    const int npairs = ntl / 2, nchunks = 3 * ntl;
    const int rot = static_cast<int>(blockIdx.x % 3);   // rotated chunk order per workgroup (spreads the L2 stream)
    auto tile_row0 = [&](int ti) { return (blockIdx.x + static_cast<int64_t>(ti) * gridDim.x) * kTR; };
    auto chunk_map = [&](int c, int& ti, int& pos) {
        if (c < 6 * npairs) {
            const int p = c / 6, r = c - 6 * p;
            pos = r >> 1;
            ti = 2 * p + (r & 1);
        } else {
            pos = c - 6 * npairs;
            ti = ntl - 1;
        }
    };

    if (w >= 4) {
        // ---------------------------------------------------------------------- movers
        // (they are the critical path: their VALU work wins the issue arbitration against the consumer on the SIMD)
        __builtin_amdgcn_s_setprio(3);
        const int pw = w - 4, pt = threadIdx.x - 256;
        float4 pf[3][8];
        // loads of a chunk: uniform 64-bit base (tile, k chunk) + a 32-bit per-lane byte offset; rows past the end
        // of a partial tile are clamped on the offset (row-major, so the row term dominates the comparison)
        // A3: the three k-chunks are three separate [R,128] matrices (row pitch 512 B) instead of column blocks of one [R,384]
        constexpr int APITCH = A3 ? 512 : K * 4;
        const unsigned voff0 = static_cast<unsigned>(pt >> 5) * APITCH + static_cast<unsigned>(pt & 31) * 16;
        auto fetch = [&](float4 (&set)[8], int c) {
            if (c > nchunks - 1) c = nchunks - 1;
            int ti, pos;
            chunk_map(c, ti, pos);
            const int64_t r0 = tile_row0(ti);
            const int kc = (pos + rot) % KC;
            const char* base = A3 ? reinterpret_cast<const char*>(kc == 0 ? a : ep.a_alt[kc - 1]) + r0 * 512
                                  : reinterpret_cast<const char*>(a) + (r0 * K + kc * 128) * 4;
            const int64_t last = R - 1 - r0;   // >= 0
            const unsigned lim = last >= kTR - 1 ? 0xFFFFFFFFu : static_cast<unsigned>(last) * APITCH + static_cast<unsigned>(pt & 31) * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                unsigned off = voff0 + i * (8 * APITCH);
                off = off < lim ? off : lim;
                if (DG_DBG & 8) set[i] = f4(static_cast<float>(off & 7));
                else set[i] = ld4(reinterpret_cast<const float*>(base + off));
            }
        };
        auto write = [&](const float4 (&set)[8], int c) {   // chunks past the end land in the idle buffer
            // the scale of a row is per 128-wide chunk: each chunk has its own accumulation
            if (DG_DBG & 2) {
                char* pl = lds + (c & 1) * kH3Buf;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int L = pt + 256 * i, r = L >> 5, c4 = L & 31;
                    *reinterpret_cast<float4*>(pl + (i & 1) * kH3Plane + r * kX6Pitch + (c4 >> 1) * 16) = set[i];
                }
            } else {
                split_write_h3<8, 256>(set, lds + (c & 1) * kH3Buf, pt);
            }
        };
        float4 res[8];   // residual rows of the tile being finished: pw * 16 + it * 2 + half
        auto fetch_residual = [&](int ti) {
            if (!EXCH || !ep.residual) return;
            const int64_t r0 = tile_row0(ti);
            const char* base = reinterpret_cast<const char*>(ep.residual) + r0 * 512;
            const int64_t last = R - 1 - r0;
            const unsigned lim = last >= kTR - 1 ? 0xFFFFFFFFu : static_cast<unsigned>(last) * 512 + col * 16;
            const unsigned roff0 = static_cast<unsigned>(pw * 16 + half) * 512 + col * 16;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                unsigned off = roff0 + it * 1024;
                off = off < lim ? off : lim;
                res[it] = ld4(reinterpret_cast<const float*>(base + off));
            }
        };
        auto store_tile = [&](int ti) {   // exchange tile -> residual / LayerNorm -> global rows
            const int64_t r0 = tile_row0(ti);
            const bool full = r0 + kTR <= R;
            float* yrow = y + (r0 + pw * 16 + half) * 128 + col * 4;
            float4 gam = f4(0.f), bet = f4(0.f);
            if (EXCH && ep.gamma) {
                gam = ld4(ep.gamma + col * 4);
                bet = ld4(ep.beta + col * 4);
            }
            // two groups of four rows (register budget); inside a group every result has its own registers and
            // the stores follow the arithmetic
#pragma unroll
            for (int hg = 0; hg < 2; ++hg) {
                float4 yv[4], pv[4];
                float mu4[4], rs4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int it = 4 * hg + j, rr = pw * 16 + it * 2 + half;
                    float4 v = ld4(ex + rr * 128 + col * 4);
                    if (EXCH && ep.residual) v += res[it];
                    pv[j] = v;
                    if (EXCH && ep.gamma) {
                        const float mu = half_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 128.0f);
                        const float4 d = v - f4(mu);
                        const float var = half_sum((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * (1.0f / 128.0f);
                        const float rs = rsqrtf(var + ep.eps);
                        v = fma4(rs * d, gam, bet);
                        mu4[j] = mu;
                        rs4[j] = rs;
                    }
                    yv[j] = v;
                }
                if (full) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int it = 4 * hg + j;
                        st4(yrow + it * 256, yv[j]);
                        if (EXCH && ep.pre && ep.gamma) st4(ep.pre + (r0 + pw * 16 + it * 2 + half) * 128 + col * 4, pv[j]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int it = 4 * hg + j, rr = pw * 16 + it * 2 + half;
                        if (r0 + rr < R) {
                            st4(yrow + it * 256, yv[j]);
                            if (EXCH && ep.pre && ep.gamma) st4(ep.pre + (r0 + rr) * 128 + col * 4, pv[j]);
                        }
                    }
                }
                if (EXCH && ep.gamma && col == 0) {   // one divergent block per group for the row statistics
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int rr = pw * 16 + (4 * hg + j) * 2 + half;
                        if (r0 + rr < R) {
                            ep.mean[r0 + rr] = mu4[j];
                            ep.rstd[r0 + rr] = rs4[j];
                        }
                    }
                }
            }
        };
#if DG_DBG & 16
        unsigned long long tl = __builtin_amdgcn_s_memtime(), tW = 0, tF = 0, tB = 0, tS = 0, t00 = tl;
#define STAMP(x) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); x += t_ - tl; tl = t_; }
#else
#define STAMP(x)
#endif
#define SYNC() { __syncthreads(); STAMP(tB) }
#define STORE_TILE(t) { store_tile(t); STAMP(tS) }
#define WRITE(a_, b_) { write(a_, b_); STAMP(tW) }
#define FETCH(a_, b_) { fetch(a_, b_); STAMP(tF) }
        FETCH(pf[0], 0);
        FETCH(pf[1], 1);
        FETCH(pf[2], 2);
        WRITE(pf[0], 0);
        FETCH(pf[0], 3);
        SYNC();   // chunk 0 is in planes[0]
        int c = 0;
        for (int p = 0; p < npairs; ++p, c += 6) {
            const int t0 = 2 * p, t1 = 2 * p + 1;
            WRITE(pf[1], c + 1); FETCH(pf[1], c + 4); SYNC();                              // (t0, 0)
            WRITE(pf[2], c + 2); FETCH(pf[2], c + 5); SYNC();                              // (t1, 0)
            WRITE(pf[0], c + 3); FETCH(pf[0], c + 6); SYNC();                              // (t0, 1)
            WRITE(pf[1], c + 4); FETCH(pf[1], c + 7); fetch_residual(t0); SYNC();          // (t1, 1)
            WRITE(pf[2], c + 5); FETCH(pf[2], c + 8); SYNC(); SYNC();             // (t0, 2): A, B
            STORE_TILE(t0);
            fetch_residual(t1);
            WRITE(pf[0], c + 6); FETCH(pf[0], c + 9); SYNC(); SYNC();             // (t1, 2): A, B
            STORE_TILE(t1);
        }
        if (ntl & 1) {
            const int t = ntl - 1;
            WRITE(pf[1], c + 1); FETCH(pf[1], c + 4); SYNC();                              // (t, 0)
            WRITE(pf[2], c + 2); FETCH(pf[2], c + 5); fetch_residual(t); SYNC();           // (t, 1)
            WRITE(pf[0], c + 3); SYNC(); SYNC();                                  // (t, 2): A, B
            STORE_TILE(t);
        }
#if DG_DBG & 16
        if (lane == 0 && blockIdx.x == 17 && ep.rstd) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(ep.rstd) + 8 * w;
            o[0] = tW; o[1] = tF; o[2] = tB; o[3] = tS; o[4] = __builtin_amdgcn_s_memtime() - t00; o[5] = ntl;
        }
#endif
#undef SYNC
#undef STORE_TILE
#undef WRITE
#undef FETCH
        return;
    }

    // -------------------------------------------------------------------------- consumers
    f16x8 bfr[2][8];
    auto b_ptr = [&](int pos) { return packed + (static_cast<size_t>(w) * KS + ((pos + rot) % KC) * 8) * 2 * 64 + lane; };
    {
        const f16x8* b0 = b_ptr(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int p = 0; p < 2; ++p) bfr[p][ks] = b0[(ks * 2 + p) * 64];
    }
    const float bias = ep.bias ? ep.bias[32 * w + col] : 0.f;
    const float cs = reinterpret_cast<const float*>(packed + static_cast<size_t>(4) * KS * 2 * 64)[32 * w + col];
    f32x16 accs[2][2];
    int chunk = 0;
    __syncthreads();   // chunk 0 is in planes[0]
#if DG_DBG & 16
    unsigned long long tl = __builtin_amdgcn_s_memtime(), tM = 0, tFo = 0, tE = 0, tB = 0, t00 = tl;
#endif
    auto body = [&](int pos, f32x16 (&acc)[2], auto second_tag) {
        constexpr bool SECOND = decltype(second_tag)::value;
        if (pos == 0) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
        }
        const f16x8* bnext = b_ptr((pos + 1) % KC);
        const char* pl = lds + (chunk & 1) * kH3Buf;
        f16x8 af[2][2][2];
        auto frags = [&](int ks, f16x8 (&dst)[2][2]) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    dst[m][p] = *reinterpret_cast<const f16x8*>(pl + p * kH3Plane + (32 * m + col) * kX6Pitch +
                                                                (ks * 16 + 8 * half) * 2);
        };
        f32x16 part[2];   // this chunk's products; folded into `acc` with the chunk's inverse row scales
        frags(0, af[0]);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks + 1 < 8) frags(ks + 1, af[(ks + 1) & 1]);
            if (!SECOND) {
                switch (ks) {
                    case 0: wait_b_refill<14>(bfr[0][ks], bfr[1][ks]); break;
                    case 1: wait_b_refill<12>(bfr[0][ks], bfr[1][ks]); break;
                    case 2: wait_b_refill<10>(bfr[0][ks], bfr[1][ks]); break;
                    case 3: wait_b_refill<8>(bfr[0][ks], bfr[1][ks]); break;
                    case 4: wait_b_refill<6>(bfr[0][ks], bfr[1][ks]); break;
                    case 5: wait_b_refill<4>(bfr[0][ks], bfr[1][ks]); break;
                    case 6: wait_b_refill<2>(bfr[0][ks], bfr[1][ks]); break;
                    default: wait_b_refill<0>(bfr[0][ks], bfr[1][ks]); break;
                }
            }
            const f16x8(&f)[2][2] = af[ks & 1];
            constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};   // lo.hi, hi.lo, hi.hi: smallest terms first
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    if (DG_DBG & 1) {
                        if (ks == 0 && t == 0) for (int i = 0; i < 16; ++i) part[m][i] = f[m][0][0] * (float)bfr[0][ks][i & 7];
                    } else if (ks == 0 && t == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        part[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[m][TA[t]], bfr[TB[t]][ks], zero, 0, 0, 0);
                    } else {
                        part[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[m][TA[t]], bfr[TB[t]][ks], part[m], 0, 0, 0);
                    }
                }
            if (SECOND) {
#pragma unroll
                for (int p = 0; p < 2; ++p) load_b128_async(bfr[p][ks], bnext + (ks * 2 + p) * 64);
            }
        }
        STAMP(tM)
        {
            const float* rsp = reinterpret_cast<const float*>(pl + kH3Rs);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // inverse row scale of the chunk TIMES this lane's inverse column scale: folding with the row scale alone
                    // overflows for rows above ~2^90 although the result is representable (tests: max 2^120 row)
                    const float4 r4 = cs * ld4(rsp + 32 * m + 8 * q + 4 * half);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[m][4 * q + i] = fmaf(part[m][4 * q + i], comp(r4, i), acc[m][4 * q + i]);
                }
        }
        STAMP(tFo)
        if (pos == KC - 1) {
            __syncthreads();   // A: the movers have finished with the previous exchange tile
            STAMP(tB)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int rr = 32 * m + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                    float v = acc[m][reg] + bias;
                    if (ep.relu) v = fmaxf(v, 0.f);
                    ex[rr * 128 + 32 * w + col] = v;
                }
        }
        STAMP(tE)
        __syncthreads();   // B
        STAMP(tB)
        ++chunk;
    };
    for (int p = 0; p < npairs; ++p) {
#pragma unroll 1
        for (int pos = 0; pos < KC; ++pos) {
            body(pos, accs[0], std::false_type{});
            body(pos, accs[1], std::true_type{});
        }
    }
    if (ntl & 1) {
#pragma unroll 1
        for (int pos = 0; pos < KC; ++pos) {
            body(pos, accs[0], std::false_type{});
            if (pos + 1 < KC) {   // no partner tile: refill between the units
                const f16x8* bnext = b_ptr(pos + 1);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                    for (int q = 0; q < 2; ++q) load_b128_async(bfr[q][ks], bnext + (ks * 2 + q) * 64);
            }
        }
    }
#if DG_DBG & 16
    if (lane == 0 && blockIdx.x == 17 && ep.rstd) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(ep.rstd) + 8 * w;
        o[0] = tM; o[1] = tFo; o[2] = tE; o[3] = tB; o[4] = __builtin_amdgcn_s_memtime() - t00; o[5] = ntl;
    }
#endif
#undef STAMP
}

}  // namespace
}  // namespace dg

namespace dg {

// The float32 row GEMMs run the fp16 hi + lo arithmetic (three MFMA products); the v_mfma_f32_32x32x2_f32 kernels of round 1
// are no longer launched.
static constexpr bool use_x6() { return true; }

size_t row_gemm_f32_packed_floats(int n_out, int k_contract) {
    if (n_out < 1 || k_contract < 1) return 0;
    const size_t nt = (n_out + 31) / 32, kc = (k_contract + 127) / 128;
    return nt * kc * 8 * 2 * 64 * 4 + nt * 32;   // fp16x3: [slab][k-step][plane][lane] x 8 fp16, inv_col_scale
}

int row_gemm_f32_pack(const float* w, float* packed, int rows, int cols, int mode, dg_stream_t stream_,
                      const float* w1, const float* w2) {
    if (!w || !packed) return fail(DG_E_ARG, "dg_row_gemm_pack: null pointer");
    if ((w1 || w2) && (!w1 || !w2 || rows != 384 || !use_x6()))
        return fail(DG_E_ARG, "dg_row_gemm_pack3: needs three [128, cols] weights (and the fp16 hi+lo row GEMM)");
    if (mode != 0 && mode != 1) return fail(DG_E_ARG, "dg_row_gemm_pack: mode must be 0 (forward) or 1 (dgrad)");
    const int n_out = mode == 0 ? rows : cols, k = mode == 0 ? cols : rows;
    const int nt = (n_out + 31) / 32, kc = (k + 127) / 128;
    hipLaunchKernelGGL(pack_weight_h3_kernel, dim3(nt, kc), dim3(512), 0, static_cast<hipStream_t>(stream_), w, w1, w2,
                       reinterpret_cast<f16x8*>(packed), rows, cols, mode, nt);
    return check_launch("dg_row_gemm_pack");
}

int row_gemm_f32_pack_batch(const void* table, int n, int max_rows_cols, dg_stream_t stream_) {
    if (!table) return fail(DG_E_ARG, "dg_row_gemm_pack_batch: null pointer");
    if (n < 1) return 0;
    if (!use_x6()) return fail(DG_E_ARG, "dg_row_gemm_pack_batch: needs the fp16 hi+lo row GEMM (DG_ROW_GEMM=mfma32 is set)");
    const int nt = (max_rows_cols + 31) / 32, kc = (max_rows_cols + 127) / 128;
    hipLaunchKernelGGL(pack_weight_h3_batch_kernel, dim3(nt, kc, n), dim3(512), 0, static_cast<hipStream_t>(stream_),
                       static_cast<const long long*>(table));
    return check_launch("dg_row_gemm_pack_batch");
}

size_t row_gemm_f32_mask_words(int64_t R, int K, int N) {
    // geometry of the direct-epilogue kernels (see the launch table in dg_row_gemm)
    if (K == 128 && N == 384) return static_cast<size_t>((R + 31) / 32) * 12 * 64;      // (>= row_gemm_n384_mask_words(R))
    if (K == 128 && N == 128) return static_cast<size_t>((R + 63) / 64) * 8 * 64;   // upper bound over variants
    return 0;
}

int row_gemm_f32(const float* a, const float* packed, float* y, int64_t R, int K, int N, const float* bias,
                           int relu, unsigned* relu_bits_out, const unsigned* mask_bits, const float* residual,
                           const float* gamma, const float* beta, float* mean, float* rstd, float* pre_ln,
                           float eps, dg_stream_t stream_, const float* ascale, float* yscale, int afmt, int yfmt,
                           const void* alo, void* ylo) {
    if (!a || !packed || !y) return fail(DG_E_ARG, "dg_row_gemm: null pointer");
    if ((afmt && K != 384) || (yfmt && N != 384)) return fail(DG_E_ARG, "dg_row_gemm: narrow hidden operands are the 384-wide ones");
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
    if (use_x6() && (mask_bits || relu_bits_out) && N != 384)
        return fail(DG_E_ARG, "dg_row_gemm: bit masks need the 128 -> 384 shape");
    if (R == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Epilogue ep{bias, mask_bits, relu_bits_out, residual, gamma, beta, mean, rstd, pre_ln, eps, relu};
    ProfScope prof(R < edge_rows() ? DG_K_ROW_GEMM : (K == 384 ? DG_K_ROW_GEMM_E_K384 : (N == 384 ? DG_K_ROW_GEMM_E_N384 : DG_K_ROW_GEMM_E_128)), stream);
    if (use_x6()) {
        const int ng = N / 128;
        const int64_t tiles = (R + kTR - 1) / kTR;
        int seqs = 256;                             // one 8-wave workgroup per CU
        if (tiles < seqs) seqs = static_cast<int>(tiles);
        (void)ng;
#define LAUNCH6(KC_, NG_, EX_)                                                                                     \
    {                                                                                                              \
        constexpr int lds_ = kH3Lds + (EX_ ? 0 : 4 * 32 * 32 * 4);                                                \
        DG_OPT_IN_LDS((&row_gemm_h3_kernel<KC_, NG_, EX_>), lds_);                                                 \
        hipLaunchKernelGGL((row_gemm_h3_kernel<KC_, NG_, EX_>), dim3(seqs), dim3(512), lds_, stream, a,            \
                           reinterpret_cast<const f16x8*>(packed), y, R, ep);                                      \
    }
        // 128 -> 384: the producer / consumer kernel (row_gemm_n384.hip)
        if (K == 128 && N == 384) {
            if (int st = launch_row_gemm_n384(a, packed, y, yscale, R, bias, relu, relu_bits_out, mask_bits, stream, yfmt, ylo)) return st;
            return check_launch("dg_row_gemm");
        }
        // 384 -> 128: the producer / consumer kernel (row_gemm_k384.hip)
        if (K == 384) {
            if (int st = launch_row_gemm_k384(a, ascale, packed, y, R, bias, relu, residual, gamma, beta, mean, rstd, pre_ln, eps, stream, afmt, alo))
                return st;
            return check_launch("dg_row_gemm");
        }
        if (K == 128 && exch) {
            ep.reverse = take_direction(R);
            LAUNCH6(1, 1, true)
        } else {
            ep.reverse = take_direction(R);
            LAUNCH6(1, 1, false)
        }
#undef LAUNCH6
    }
    return check_launch("dg_row_gemm");
}


void launch_ln_finish(const float* part, int nblocks, int K, int C, float* out0, float* out1, hipStream_t stream);

size_t row_gemm_f32_ln_bwd_workspace_bytes() { return static_cast<size_t>(256) * 8 * 256 * sizeof(float); }

// dz = LayerNormBackward(a B + residual; pre, mean, rstd, gamma), dgamma, dbeta: see Epilogue::lnb_pre
int row_gemm_f32_ln_bwd(const float* a, const float* packed, float* dz, int64_t R, int K, const float* residual,
                        const float* pre, const float* mean, const float* rstd, const float* gamma, float* dgamma,
                        float* dbeta, void* workspace, size_t workspace_bytes, dg_stream_t stream_) {
    if (!a || !packed || !dz || !pre || !mean || !rstd || !gamma || !workspace)
        return fail(DG_E_ARG, "dg_row_gemm_ln_bwd: null pointer");
    // (384 -> 128: the epilogue of that kernel runs in its producer waves, which already hold three A chunks in
    // registers: built, 593 us against 302 + 161 us for the two launches at R = 518 400 -- not kept)
    if (R < 0 || K != 128) return fail(DG_E_SHAPE, "dg_row_gemm_ln_bwd: unsupported K=%d (K = N = 128)", K);
    if (!use_x6()) return fail(DG_E_ARG, "dg_row_gemm_ln_bwd: needs the fp16 hi+lo row GEMM (DG_ROW_GEMM=mfma32 is set)");
    if (workspace_bytes < row_gemm_f32_ln_bwd_workspace_bytes()) return fail(DG_E_WORKSPACE, "dg_row_gemm_ln_bwd: workspace too small");
    if (R == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Epilogue ep{nullptr, nullptr, nullptr, residual, gamma, nullptr, const_cast<float*>(mean), const_cast<float*>(rstd), nullptr, 0.f, 0};
    ep.lnb_pre = pre;
    ep.lnb_part = static_cast<float*>(workspace);
    const int64_t tiles = (R + kTR - 1) / kTR;
    const int seqs = static_cast<int>(tiles < 256 ? tiles : 256);
    {
        ProfScope prof(R < edge_rows() ? DG_K_ROW_GEMM : DG_K_ROW_GEMM_E_128, stream);
        note_forward(R);      // (dgamma / dbeta partial sums: the order of the rows matters)
        DG_OPT_IN_LDS((&row_gemm_h3_kernel<1, 1, true, 4, true>), kH3Lds);
        hipLaunchKernelGGL((row_gemm_h3_kernel<1, 1, true, 4, true>), dim3(seqs), dim3(512), kH3Lds, stream, a,
                           reinterpret_cast<const f16x8*>(packed), dz, R, ep);
    }
    if (dgamma || dbeta) launch_ln_finish(ep.lnb_part, seqs * 8, 2, 128, dgamma, dbeta, stream);
    return check_launch("dg_row_gemm_ln_bwd");
}

bool reduce_batch_try_add(const float* part, int S, long long n_floats, float* out);      // linear_wgrad.hip

// y = dz B with dz = LayerNormBackward(dy; pre, mean, rstd, gamma) computed (and written) on the way in: see
// Epilogue::lna_pre.  K = N = 128.
int row_gemm_f32_ln_in(const float* dy, const float* pre, const float* mean, const float* rstd, const float* gamma,
                       const float* packed, float* dz, float* y, float* dgamma, float* dbeta, void* workspace,
                       size_t workspace_bytes, int64_t R, dg_stream_t stream_) {
    if (!dy || !pre || !mean || !rstd || !gamma || !packed || !dz || !y || !workspace)
        return fail(DG_E_ARG, "dg_row_gemm_ln_bwd_in: null pointer");
    if (R < 0) return fail(DG_E_SHAPE, "dg_row_gemm_ln_bwd_in: R = %lld", static_cast<long long>(R));
    if (!use_x6()) return fail(DG_E_ARG, "dg_row_gemm_ln_bwd_in: needs the fp16 hi+lo row GEMM (DG_ROW_GEMM=mfma32 is set)");
    if (workspace_bytes < row_gemm_f32_ln_bwd_workspace_bytes()) return fail(DG_E_WORKSPACE, "dg_row_gemm_ln_bwd_in: workspace too small");
    if (R == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Epilogue ep{nullptr, nullptr, nullptr, nullptr, gamma, nullptr, const_cast<float*>(mean), const_cast<float*>(rstd), nullptr, 0.f, 0};
    ep.lna_pre = pre;
    ep.lna_dz = dz;
    ep.lnb_part = static_cast<float*>(workspace);
    const int64_t tiles = (R + kTR - 1) / kTR;
    const int seqs = static_cast<int>(tiles < 256 ? tiles : 256);
    {
        ProfScope prof(R < edge_rows() ? DG_K_ROW_GEMM : DG_K_ROW_GEMM_E_128, stream);
        note_forward(R);      // (dgamma / dbeta partial sums: the order of the rows matters)
        constexpr int lds = kH3Lds + 4 * 32 * 32 * 4;
        DG_OPT_IN_LDS((&row_gemm_h3_kernel<1, 1, false, 4, false, true>), lds);
        hipLaunchKernelGGL((row_gemm_h3_kernel<1, 1, false, 4, false, true>), dim3(seqs), dim3(512), lds, stream, dy,
                           reinterpret_cast<const f16x8*>(packed), y, R, ep);
    }
    // inside dg_linear_wgrad_batch_begin / _end the reduction joins that batch's launch (dgamma, dbeta adjacent)
    if ((dgamma || dbeta) && !(dgamma && dbeta == dgamma + 128 && reduce_batch_try_add(ep.lnb_part, seqs * 8, 256, dgamma)))
        launch_ln_finish(ep.lnb_part, seqs * 8, 2, 128, dgamma, dbeta, stream);
    return check_launch("dg_row_gemm_ln_bwd_in");
}


// Three Linears that share their input in ONE launch: y_i = a W_i^T + b_i (q / k / v of an attention block, reference
// layers.py:111-113; also their transposed use in the second order).  `packed`: dg_row_gemm_pack3(mode 0).
int row_gemm_f32_lin3(const float* a, const float* packed, float* y0, float* y1, float* y2, int64_t R, const float* b0,
                      const float* b1, const float* b2, dg_stream_t stream_) {
    if (!a || !packed || !y0 || !y1 || !y2) return fail(DG_E_ARG, "dg_row_gemm_lin3: null pointer");
    if (R < 0) return fail(DG_E_SHAPE, "dg_row_gemm_lin3: negative row count");
    if (!use_x6()) return fail(DG_E_ARG, "dg_row_gemm_lin3: needs the fp16 hi+lo row GEMM (DG_ROW_GEMM=mfma32 is set)");
    if (R == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Epilogue ep{b0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0};
    ep.y_alt[0] = y1;
    ep.y_alt[1] = y2;
    ep.bias_alt[0] = b1;
    ep.bias_alt[1] = b2;
    const int64_t tiles = (R + kTR - 1) / kTR;
    const int seqs = static_cast<int>(tiles < 256 ? tiles : 256);
    // profiler / roofline: an edge-level launch of this kernel is a 128 -> 384 launch
    ProfScope prof(R < edge_rows() ? DG_K_ROW_GEMM : DG_K_ROW_GEMM_E_N384, stream);
    // the column-group kernel (B fragments streamed per group: these launches are node-level, one or two tiles per
    // workgroup, where residency buys nothing; the resident-B instance with three outputs spills 168 B / lane)
    constexpr int lds3 = kH3Lds + 4 * 32 * 32 * 4;
    DG_OPT_IN_LDS((&row_gemm_h3_kernel<1, 3, false, 4, false, false, true>), lds3);
    hipLaunchKernelGGL((row_gemm_h3_kernel<1, 3, false, 4, false, false, true>), dim3(seqs), dim3(512), lds3, stream, a,
                       reinterpret_cast<const f16x8*>(packed), y0, R, ep);
    return check_launch("dg_row_gemm_lin3");
}

// y = a0 B0 + a1 B1 + a2 B2 (+ residual): the input gradient of those three Linears in ONE launch -- a 384 -> 128
// contraction whose k-chunks are three separate [R,128] matrices.  `packed`: dg_row_gemm_pack3(mode 1).
int row_gemm_f32_sum3(const float* a0, const float* a1, const float* a2, const float* packed, float* y, int64_t R,
                      const float* residual, dg_stream_t stream_) {
    if (!a0 || !a1 || !a2 || !packed || !y) return fail(DG_E_ARG, "dg_row_gemm_sum3: null pointer");
    if (R < 0) return fail(DG_E_SHAPE, "dg_row_gemm_sum3: negative row count");
    if (!use_x6()) return fail(DG_E_ARG, "dg_row_gemm_sum3: needs the fp16 hi+lo row GEMM (DG_ROW_GEMM=mfma32 is set)");
    if (R == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Epilogue ep{nullptr, nullptr, nullptr, residual, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0};
    ep.a_alt[0] = a1;
    ep.a_alt[1] = a2;
    const int64_t tiles = (R + kTR - 1) / kTR;
    const int seqs = static_cast<int>(tiles < 256 ? tiles : 256);
    ProfScope prof(R < edge_rows() ? DG_K_ROW_GEMM : DG_K_ROW_GEMM_E_K384, stream);
    constexpr int lds384 = 2 * kH3Buf + kTR * 128 * 4;
    if (residual) {
        DG_OPT_IN_LDS((&row_gemm_h3_k384_kernel<true, true>), lds384);
        hipLaunchKernelGGL((row_gemm_h3_k384_kernel<true, true>), dim3(seqs), dim3(512), lds384, stream, a0,
                           reinterpret_cast<const f16x8*>(packed), y, R, ep);
    } else {
        DG_OPT_IN_LDS((&row_gemm_h3_k384_kernel<false, true>), lds384);
        hipLaunchKernelGGL((row_gemm_h3_k384_kernel<false, true>), dim3(seqs), dim3(512), lds384, stream, a0,
                           reinterpret_cast<const f16x8*>(packed), y, R, ep);
    }
    return check_launch("dg_row_gemm_sum3");
}

}  // namespace dg
