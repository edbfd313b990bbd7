// Fused feed-forward half of an Encoder_Block in the bf16 configuration -- reference
// src/model/layers.py:191-192 with MLP.forward (:40-54):
//
//     y = LayerNorm( x + fc2( relu( fc1(x) ) ) ) * gamma + beta            C = 128, hidden H = 384
//
// The [R, 384] hidden tensor never exists in HBM (R = B N^2 = 518 400 rows at configs[1], 4.1 M at
// configs[2]): per 64-row tile it is produced by MFMA into LDS (48 KB of bf16) and consumed from
// there; the backward recomputes it.  Saved for the backward: the pre-LayerNorm sum (bf16), mean /
// rstd (fp32) and ONE BIT per hidden element (the ReLU mask).
//
//   forward   (1 kernel)  : x -> [fc1 + b1 + ReLU -> H tile] -> [fc2 -> exchange tile] -> + b2 + x -> LN -> y
//                           HBM: read x, write y, pre (+ bits): 3 x 256 B per row, vs 13 x 256 B unfused
//   backward  (3 kernels) : dx-kernel   dy, pre -> dz = LN'(dy) -> dh = (dz W2) * mask -> dx = dz + dh W1
//                           dW2-kernel  x -> h (recomputed, stored TRANSPOSED in LDS), dW2 += dz^T h
//                           dW1-kernel  dz -> dh (recomputed, transposed), dW1 += dh^T x, db1 += sum dh
//                           (one kernel cannot hold both [128 x 384] fp32 weight-gradient accumulators,
//                           192 registers per lane, next to the resident weight fragments)
//
// MI355X mapping: 8 waves, one workgroup per CU, persistent over 64-row tiles.  All weight operands
// are MFMA fragments resident in VGPRs for the whole kernel (gemm_bf16.h: P16 / P32 orders); x / dz
// tiles arrive by LDS-DMA, double buffered.  128 -> 384 products use v_mfma_f32_16x16x32_bf16 (wave w
// owns hidden channels [48 w, 48 w + 48) for all 64 rows: no fragment is duplicated across waves),
// 384 -> 128 products v_mfma_f32_32x32x16_bf16 (wave = 32 output channels x 32 rows).  Products whose
// result is consumed row-wise are "swapped" (weights = A operand: a lane gets 4 consecutive channels of
// one row, 8-byte LDS stores); products that feed a weight gradient are not (a lane gets 4 consecutive
// ROWS of one channel = the transposed tile the contraction over rows wants, read back as 16-byte
// fragments).  Weight-gradient partials are reduced afterwards in a fixed order (bit-reproducible).
#include "gemm_bf16.h"

#include <cstring>

namespace dg {

int pack_bf16(const float* w, void* packed, int rows, int cols, int mode, int mb_size, hipStream_t stream);
void launch_splitk_reduce(const float* part, int S, int64_t n4, float* out, hipStream_t stream);

namespace {

constexpr int kC = 128, kH = 384;
constexpr int kSec = 24 * 4 * 64;                 // bf16x8 entries per packed section (96 KB)
// sections of the FFN pack
constexpr int kP16W1 = 0;                         // P16 of W1   [384][128]  (fc1 forward; h recompute)
constexpr int kP16W2 = kSec;                      // P16 of W2   [128][384]  (fc2 forward)
constexpr int kP16W2T = 2 * kSec;                 // P16 of W2^T [384][128]  (dh = dz W2)
constexpr int kP16W1T = 3 * kSec;                 // P16 of W1^T [128][384]  (dx = dh W1)
constexpr int kXBytes = kRowsPerTile * kC * 2;    // 16 KB
constexpr int kHBytes = kRowsPerTile * kH * 2;    // 48 KB
constexpr int kZBytes = kRowsPerTile * kC * 4;    // 32 KB
constexpr int kTPitch = 144;                      // bytes per channel row of a transposed [384][64] tile (128 + 16:
                                                  // 32 consecutive channels hit 16 distinct 16-byte slots twice)
constexpr int kTBytes = kH * kTPitch;             // 54 KB

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// 64 x 128 tile -> 128 -> 384 product, one 16-row block `rb` of the tile at a time (the caller's loop over
// rb stays ROLLED: only 3 accumulators are live, and hipcc cannot hoist all 16 fragment reads of a tile).
//   SWAPPED = true : weights are the A operand; lane = row 16 rb + (lane & 15),
//                    hidden channels 48 w + 16 i + 4 (lane >> 4) + reg
//   SWAPPED = false: tile rows are the A operand; lane = hidden channel 48 w + 16 i + (lane & 15),
//                    rows 16 rb + 4 (lane >> 4) + reg
__device__ __forceinline__ void up_frags(const char* tile, int rb, bf16x8 (&xf)[4], int lane) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        xf[ks] = *reinterpret_cast<const bf16x8*>(tile + tile_off(16 * rb + (lane & 15), 32 * ks + 8 * (lane >> 4), kC));
}
template <bool SWAPPED>
__device__ __forceinline__ void up_mfma(const bf16x8 (&xf)[4], const bf16x8 (&wf)[3][4], f32x4 (&acc)[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] = SWAPPED ? mfma16(wf[i][ks], xf[ks], acc[i]) : mfma16(xf[ks], wf[i][ks], acc[i]);
}
// The 128 -> 384 product of a 64-row tile as two rolled iterations over PAIRS of 16-row blocks: the fragments of
// the next block are requested before the MFMAs of the current one (two fragment sets, three accumulators live),
// `epi(rb, acc)` consumes a block's result.  The fences pin "request next, multiply current, finish current".
template <bool SWAPPED, bool PREFETCH, typename Epi>
__device__ __forceinline__ void gemm_up_tile(const char* tile, const bf16x8 (&wf)[3][4], int lane, Epi&& epi) {
    if constexpr (PREFETCH) {
        bf16x8 fa[4], fb[4];
        up_frags(tile, 0, fa, lane);
#pragma unroll 1
        for (int rb = 0; rb < 4; rb += 2) {
            f32x4 acc[3];
            up_frags(tile, rb + 1, fb, lane);
            up_mfma<SWAPPED>(fa, wf, acc);
            __builtin_amdgcn_sched_barrier(0);
            epi(rb, acc);
            if (rb + 2 < 4) up_frags(tile, rb + 2, fa, lane);
            up_mfma<SWAPPED>(fb, wf, acc);
            __builtin_amdgcn_sched_barrier(0);
            epi(rb + 1, acc);
        }
    } else {      // kernels at the register limit: one fragment set, the two waves of a SIMD overlap each other
#pragma unroll 1
        for (int rb = 0; rb < 4; ++rb) {
            bf16x8 fa[4];
            f32x4 acc[3];
            up_frags(tile, rb, fa, lane);
            up_mfma<SWAPPED>(fa, wf, acc);
            epi(rb, acc);
        }
    }
}
// 64 x 384 tile -> 384 -> 128 product on 16x16x32 tiles (swapped), one 16-row block per call: wave w owns
// output channels [16 w, 16 w + 16) -- 48 registers of weight fragments, nothing duplicated across waves
// (the 32x32x16 form above needs 96).  Result: lane = row 16 rb + (lane & 15), channels 16 w + 4 (lane >> 4) + reg.
__device__ __forceinline__ void down16_frags(const char* rowp, int rsw, int kq, int k4, bf16x8 (&hf)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int chunk = 4 * (4 * k4 + j) + kq;
        hf[j] = *reinterpret_cast<const bf16x8*>(rowp + (((chunk & ~15) | ((chunk & 15) ^ rsw)) << 4));
    }
}
template <bool FENCE = true>
__device__ __forceinline__ f32x4 gemm_down_block16(const char* tile, int rb, const bf16x8 (&wf)[12], int lane) {
    f32x4 acc_a = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
    const int row = 16 * rb + (lane & 15);
    const char* rowp = tile + row * (kH * 2);
    const int rsw = row & 15, kq = lane >> 4;
    bf16x8 h0[4], h1[4];     // two groups of four k-steps in flight (32 registers, transient)
    down16_frags(rowp, rsw, kq, 0, h0);
    down16_frags(rowp, rsw, kq, 1, h1);
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
        acc_a = mfma16(wf[j], h0[j], acc_a);
        acc_b = mfma16(wf[j + 1], h0[j + 1], acc_b);
    }
    if (FENCE) __builtin_amdgcn_sched_barrier(0);
    down16_frags(rowp, rsw, kq, 2, h0);
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
        acc_a = mfma16(wf[4 + j], h1[j], acc_a);
        acc_b = mfma16(wf[4 + j + 1], h1[j + 1], acc_b);
    }
    if (FENCE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
        acc_a = mfma16(wf[8 + j], h0[j], acc_a);
        acc_b = mfma16(wf[8 + j + 1], h0[j + 1], acc_b);
    }
    if (FENCE) __builtin_amdgcn_sched_barrier(0);
    return acc_a + acc_b;
}

// ------------------------------------------------------------------------------------ forward --
// The first version of this kernel (PMC counters, phase ablation and s_memtime stamps: profiles/r02_ffn_fwd_phases.txt)
// was VALU-issue bound, not MFMA or HBM bound: ~740 vector instructions per wave and tile against 96 MFMAs -- swizzled
// LDS addresses recomputed for every fragment, ~25 instructions per 4 hidden values in the fc1 epilogue, a row phase of
// 4 x ~100 dependent instructions -- and its phases ran strictly one after the other in all 8 waves.  Here
//   * every LDS address is a per-lane offset computed ONCE before the tile loop plus an immediate (the XOR swizzle
//     only touches bits that are fixed per lane once the row block / k-group is an immediate);
//   * the row phase handles the wave's 8 rows in ONE pass: 8 lanes x 16 channels per row, so the two LayerNorm
//     reductions are 3 DPP steps inside a half-row (no readlane), and the results leave as 16-byte stores;
//   * outputs are stored unconditionally (the caller pads them to whole tiles): no per-row branches.
__device__ __forceinline__ float sum8(float x) {      // sum over the 8 lanes of a half-row, result in all 8
    x = dpp_add<0xB1>(x);    // quad_perm [1,0,3,2]
    x = dpp_add<0x4E>(x);    // quad_perm [2,3,0,1]
    x = dpp_add<0x141>(x);   // row_half_mirror: lane i <-> 7 - i, i.e. the other quad
    return x;
}
template <bool SAVE>
__global__ __launch_bounds__(512, 2) void ffn_fwd_bf16_v2_kernel(const bf16_t* __restrict__ x, const bf16x8* __restrict__ pk,
                                                                const float* __restrict__ b1, const float* __restrict__ b2,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                bf16_t* __restrict__ y, bf16_t* __restrict__ pre,
                                                                float* __restrict__ mean, float* __restrict__ rstd,
                                                                unsigned* __restrict__ relu_bits, int64_t R, float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xbuf = smem;                       // [2][64][128] bf16
    char* htile = smem + 2 * kXBytes;        // [64][384] bf16
    char* zt = htile + kHBytes;              // [64][128] fp32 exchange tile
    float4* gb = reinterpret_cast<float4*>(zt + kZBytes);   // gamma [32 x float4], beta [32 x float4]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int64_t tiles = (R + kRowsPerTile - 1) / kRowsPerTile;

    bf16x8 wf1[3][4], wf2[12];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf1[i][ks] = pk[kP16W1 + ((3 * w + i) * 4 + ks) * 64 + lane];
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) wf2[ks] = pk[kP16W2 + (w * 12 + ks) * 64 + lane];
    float4 b1v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) b1v[i] = ld4(b1 + 48 * w + 16 * i + 4 * kq);
    const float4 b2z = ld4(b2 + 16 * w + 4 * kq);
    if (threadIdx.x < 32) gb[threadIdx.x] = ld4(gamma + 4 * threadIdx.x);
    else if (threadIdx.x < 64) gb[threadIdx.x] = ld4(beta + 4 * (threadIdx.x - 32));
    // ---- per-lane LDS byte offsets (row blocks of 16 rows and k-groups enter as immediates)
    unsigned xf_off[4], hf_off[4], hw_off[3], zr_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        xf_off[j] = r16 * (kC * 2) + (((4 * j + kq) ^ r16) << 4);          // X fragment, k-step j
        hf_off[j] = r16 * (kH * 2) + (((4 * j + kq) ^ r16) << 4);          // H fragment, k-step 4 a + j (+ 256 a)
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int chunk = 6 * w + 2 * i + (kq >> 1);
        hw_off[i] = r16 * (kH * 2) + (((chunk & ~15) | ((chunk & 15) ^ r16)) << 4) + (kq & 1) * 8;
    }
    const int zslot = 4 * w + kq;
    const unsigned zw_off = r16 * (kC * 4) + (((zslot & ~7) | ((zslot & 7) ^ (r16 & 7))) << 4);
    const unsigned xr_off = r16 * (kC * 2) + (((2 * w + (kq >> 1)) ^ r16) << 4) + (kq & 1) * 8;
    const int r8 = lane >> 3, sub = lane & 7;      // row phase: row 8 w + r8, channels [16 sub, 16 sub + 16)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int slot = 4 * sub + q;
        zr_off[q] = (8 * w + r8) * (kC * 4) + (((slot & ~7) | ((slot & 7) ^ r8)) << 4);
    }
    wait_all_vmem_visible();

    int64_t tix = blockIdx.x;
    if (tix < tiles) dma_tile_bf16<kC, 8>(x, tix * kRowsPerTile, R, xbuf, w, lane);
    wait_all_vmem();
    int buf = 0;
    for (; tix < tiles; tix += gridDim.x, buf ^= 1) {
        const int64_t r0 = tix * kRowsPerTile;
        __syncthreads();      // x(t) landed everywhere (waited for before the previous stores); H / exchange tiles free
        if (tix + gridDim.x < tiles)
            dma_tile_bf16<kC, 8>(x, (tix + gridDim.x) * kRowsPerTile, R, xbuf + (buf ^ 1) * kXBytes, w, lane);
        const char* xt = xbuf + buf * kXBytes;
        // ---- fc1 + b1 + ReLU -> H tile (bf16), one mask bit per element.  Software pipeline over the four 16-row
        // blocks: the MFMAs of block nb + 1 are issued before the epilogue of block nb (two accumulator sets), so
        // the epilogue's vector instructions fill the MFMA issue gaps; the bias enters as the accumulators' initial
        // value; the ReLU bits are taken from the packed bf16 words (2 values per instruction).
        unsigned bits_lo = 0u, bits_hi = 0u;
        {
            bf16x8 fa[4], fb[4];
            f32x4 acc_a[3], acc_b[3];
            auto frags = [&](int nb, bf16x8 (&f)[4]) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const bf16x8*>(xt + nb * (16 * kC * 2) + xf_off[ks]);
            };
            auto mfmas = [&](const bf16x8 (&f)[4], f32x4 (&acc)[3]) {
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[i] = f32x4{b1v[i].x, b1v[i].y, b1v[i].z, b1v[i].w};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int i = 0; i < 3; ++i) acc[i] = mfma16(wf1[i][ks], f[ks], acc[i]);
            };
            auto finish = [&](int nb, const f32x4 (&acc)[3]) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const u32x2_t pv = pack4_bf16(max4(make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]), f4(0.f)));
                    // a positive bf16 half plus 0x7FFF carries into its bit 15: flags at bits 15 / 31 (values 0, 2 of the
                    // group) and, for the second word, shifted to 14 / 30 (values 1... see ffn_mask4 for the decoding)
                    const unsigned f0 = (pv[0] + 0x7FFF7FFFu) & 0x80008000u, f1 = (pv[1] + 0x7FFF7FFFu) & 0x80008000u;
                    const unsigned fl = f0 | (f1 >> 1);
                    const int g = nb * 3 + i;                    // 12 groups per lane and tile, 7 per word
                    if (g < 7) bits_lo |= fl >> (2 * g);
                    else bits_hi |= fl >> (2 * (g - 7));
                    *reinterpret_cast<u32x2_t*>(htile + nb * (16 * kH * 2) + hw_off[i]) = pv;
                }
            };
            frags(0, fa);
            frags(1, fb);
            mfmas(fa, acc_a);
            mfmas(fb, acc_b);
            frags(2, fa);
            finish(0, acc_a);
            mfmas(fa, acc_a);
            frags(3, fb);
            finish(1, acc_b);
            mfmas(fb, acc_b);
            finish(2, acc_a);
            finish(3, acc_b);
        }
        const unsigned long long bits = static_cast<unsigned long long>(bits_lo) | (static_cast<unsigned long long>(bits_hi) << 32);
        if (SAVE) {
            const size_t bix = (static_cast<size_t>(tix) * 512 + threadIdx.x) * 2;
            relu_bits[bix] = static_cast<unsigned>(bits);
            relu_bits[bix + 1] = static_cast<unsigned>(bits >> 32);
        }
        __syncthreads();
        // ---- fc2 + b2 + x -> exchange tile (wave w: output channels [16 w, 16 w + 16) of all 64 rows); four
        // accumulator chains per 16-row block (dependent 16x16x32 MFMAs need ~3 others in between)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const char* hb = htile + nb * (16 * kH * 2);
            f32x4 ac[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                bf16x8 hf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) hf[j] = *reinterpret_cast<const bf16x8*>(hb + 256 * a + hf_off[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) ac[j] = mfma16(wf2[4 * a + j], hf[j], ac[j]);
            }
            const f32x4 a4 = (ac[0] + ac[1]) + (ac[2] + ac[3]);
            const float4 xr = unpack4_bf16(*reinterpret_cast<const u32x2_t*>(xt + nb * (16 * kC * 2) + xr_off));
            *reinterpret_cast<float4*>(zt + nb * (16 * kC * 4) + zw_off) = make_float4(a4[0], a4[1], a4[2], a4[3]) + b2z + xr;
        }
        wait_all_vmem();       // the next tile's DMA (issued two MFMA phases ago) and this wave's older stores
        __syncthreads();
        // ---- LayerNorm of this wave's 8 rows in one pass, 16-byte stores
        {
            const int64_t row = r0 + 8 * w + r8;
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(zt + zr_off[q]);
            if (SAVE) {
                u32x4_t p0, p1;
                p0[0] = pack_bf16(v[0].x, v[0].y); p0[1] = pack_bf16(v[0].z, v[0].w);
                p0[2] = pack_bf16(v[1].x, v[1].y); p0[3] = pack_bf16(v[1].z, v[1].w);
                p1[0] = pack_bf16(v[2].x, v[2].y); p1[1] = pack_bf16(v[2].z, v[2].w);
                p1[2] = pack_bf16(v[3].x, v[3].y); p1[3] = pack_bf16(v[3].z, v[3].w);
                *reinterpret_cast<u32x4_t*>(pre + row * kC + 16 * sub) = p0;
                *reinterpret_cast<u32x4_t*>(pre + row * kC + 16 * sub + 8) = p1;
            }
            float4 t = (v[0] + v[1]) + (v[2] + v[3]);
            const float mu = sum8((t.x + t.y) + (t.z + t.w)) * (1.0f / 128.0f);
            float4 sq = f4(0.f);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = v[q] - f4(mu);
                sq = fma4(v[q], v[q], sq);
            }
            const float var = sum8((sq.x + sq.y) + (sq.z + sq.w)) * (1.0f / 128.0f);
            const float rs = rsqrtf(var + eps);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fma4(rs * v[q], gb[4 * sub + q], gb[32 + 4 * sub + q]);
            u32x4_t o0, o1;
            o0[0] = pack_bf16(v[0].x, v[0].y); o0[1] = pack_bf16(v[0].z, v[0].w);
            o0[2] = pack_bf16(v[1].x, v[1].y); o0[3] = pack_bf16(v[1].z, v[1].w);
            o1[0] = pack_bf16(v[2].x, v[2].y); o1[1] = pack_bf16(v[2].z, v[2].w);
            o1[2] = pack_bf16(v[3].x, v[3].y); o1[3] = pack_bf16(v[3].z, v[3].w);
            *reinterpret_cast<u32x4_t*>(y + row * kC + 16 * sub) = o0;
            *reinterpret_cast<u32x4_t*>(y + row * kC + 16 * sub + 8) = o1;
            if (sub == 0) {
                mean[row] = mu;
                rstd[row] = rs;
            }
        }
    }
}

// ---------------------------------------------------------------------------- backward: dx -----
// part[block][3][128]: dgamma, dbeta, db2 partial sums
__global__ __launch_bounds__(512, 2) void ffn_bwd_dx_bf16_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ pre,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const unsigned* __restrict__ relu_bits,
                                                                const float* __restrict__ gamma, const bf16x8* __restrict__ pk,
                                                                bf16_t* __restrict__ dz, bf16_t* __restrict__ dx,
                                                                float* __restrict__ part, int64_t R) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* dzt = smem;                        // [64][128] bf16
    char* dht = smem + kXBytes;              // [64][384] bf16
    char* xch = dht + kHBytes;               // [64][128] fp32; afterwards the partial-sum scratch
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, col = lane & 31;
    const int64_t tiles = (R + kRowsPerTile - 1) / kRowsPerTile;

    bf16x8 wfa[3][4], wfb[12];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wfa[i][ks] = pk[kP16W2T + ((3 * w + i) * 4 + ks) * 64 + lane];
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) wfb[ks] = pk[kP16W1T + (w * 12 + ks) * 64 + lane];
    const float4 gam = ld4(gamma + 4 * col);
    float4 dgam = f4(0.f), dbet = f4(0.f), db2 = f4(0.f);

    // rows of this wave: 8 w + 2 it + half; operands of the NEXT tile are requested during this tile's MFMAs
    // (the row statistics and the mask words too: loaded where they are used, each of the four row passes of a tile waited
    // for its own mean / rstd with nothing else to do -- four exposed memory latencies per tile, 1 block per CU)
    u32x2_t dyN[4], prN[4];
    float statN;      // lanes 0..7: mean of this wave's rows 8 w + lane, lanes 8..15: rstd of row 8 w + lane - 8
    unsigned bitN[2];
    auto fetch = [&](int64_t t) {
        const int64_t r0 = t * kRowsPerTile;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            int64_t row = r0 + 8 * w + 2 * it + half;
            if (row > R - 1) row = R - 1;
            dyN[it] = *reinterpret_cast<const u32x2_t*>(dy + row * kC + 4 * col);
            prN[it] = *reinterpret_cast<const u32x2_t*>(pre + row * kC + 4 * col);
        }
        {
            int64_t row = r0 + 8 * w + (lane & 7);
            if (row > R - 1) row = R - 1;
            statN = (lane & 8 ? rstd : mean)[row];
        }
        const size_t bix = (static_cast<size_t>(t) * 512 + threadIdx.x) * 2;
        bitN[0] = relu_bits[bix];
        bitN[1] = relu_bits[bix + 1];
    };
    int64_t tix = blockIdx.x;
    if (tix < tiles) fetch(tix);
    for (; tix < tiles; tix += gridDim.x) {
        const int64_t r0 = tix * kRowsPerTile;
        // per-iteration opaque copy of the lane id: every swizzled LDS offset below derives from it, so hipcc cannot keep
        // the loop-invariant offsets of all four phases in registers for the whole kernel (it did: 256 VGPRs + scratch)
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const int half = lo >> 5, col = lo & 31;
        __syncthreads();       // previous tile: dz tile and exchange tile fully consumed
        const unsigned long long bits = static_cast<unsigned long long>(bitN[0]) | (static_cast<unsigned long long>(bitN[1]) << 32);
        // ---- LayerNorm backward of this wave's rows -> dz (HBM + LDS tile)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rr = 8 * w + 2 * it + half;
            const int64_t row = r0 + rr;
            const bool ok = row < R;
            const int sti = __float_as_int(statN);
            const float mu = __int_as_float(half ? __builtin_amdgcn_readlane(sti, 2 * it + 1) : __builtin_amdgcn_readlane(sti, 2 * it));
            const float rs = __int_as_float(half ? __builtin_amdgcn_readlane(sti, 8 + 2 * it + 1) : __builtin_amdgcn_readlane(sti, 8 + 2 * it));
            float4 gy = unpack4_bf16(dyN[it]);
            if (!ok) gy = f4(0.f);
            const float4 xh = rs * (unpack4_bf16(prN[it]) - f4(mu));
            const float4 u = gy * gam;
            const float c1 = half_wave_sum((u.x + u.y) + (u.z + u.w)) * (1.0f / 128.0f);
            const float c2 = half_wave_sum((u.x * xh.x + u.y * xh.y) + (u.z * xh.z + u.w * xh.w)) * (1.0f / 128.0f);
            const float4 dzv = rs * (u - f4(c1) - c2 * xh);      // zero for rows past the end (gy = 0)
            dgam = fma4(gy, xh, dgam);
            dbet += gy;
            const u32x2_t dzp = pack4_bf16(dzv);
            db2 += unpack4_bf16(dzp);                              // db2 of the values the GEMMs see
            *reinterpret_cast<u32x2_t*>(dzt + tile_off(rr, 4 * col, kC)) = dzp;
            if (ok) *reinterpret_cast<u32x2_t*>(dz + row * kC + 4 * col) = dzp;
        }
        if (tix + gridDim.x < tiles) fetch(tix + gridDim.x);
        if (!dx) continue;     // block-uniform
        __syncthreads();
        // ---- dh = (dz W2) * mask -> DH tile
        gemm_up_tile<true, false>(dzt, wfa, lo, [&](int nb, const f32x4 (&acc)[3]) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int g = nb * 3 + i;                     // layout written by the forward kernel (12 groups, 7 per word)
                const unsigned t = g < 7 ? static_cast<unsigned>(bits) << (2 * g) : static_cast<unsigned>(bits >> 32) << (2 * (g - 7));
                const float4 v = make_float4(t & 0x8000u ? acc[i][0] : 0.f, t & 0x80000000u ? acc[i][1] : 0.f,
                                             t & 0x4000u ? acc[i][2] : 0.f, t & 0x40000000u ? acc[i][3] : 0.f);
                *reinterpret_cast<u32x2_t*>(dht + tile_off(16 * nb + (lo & 15), 48 * w + 16 * i + 4 * (lo >> 4), kH)) =
                    pack4_bf16(v);
            }
        });
        __syncthreads();
        // ---- dx = dz + dh W1
#pragma unroll 1
        for (int nb = 0; nb < 4; ++nb) {
            const f32x4 a4 = gemm_down_block16(dht, nb, wfb, lo);
            *reinterpret_cast<float4*>(xch + xch_off(16 * nb + (lo & 15), 16 * w + 4 * (lo >> 4), kC)) =
                make_float4(a4[0], a4[1], a4[2], a4[3]);
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rr = 8 * w + 2 * it + half;
            const int64_t row = r0 + rr;
            float4 v = *reinterpret_cast<const float4*>(xch + xch_off(rr, 4 * col, kC));
            v += unpack4_bf16(*reinterpret_cast<const u32x2_t*>(dzt + tile_off(rr, 4 * col, kC)));
            if (row < R) st4(dx + row * kC + 4 * col, v);
        }
    }
    // ---- workgroup partials of dgamma, dbeta, db2: 16 half-waves hold the same 4 channels per lane
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(xch);     // [3][16][32]
    red[(0 * 16 + 2 * w + half) * 32 + col] = dgam;
    red[(1 * 16 + 2 * w + half) * 32 + col] = dbet;
    red[(2 * 16 + 2 * w + half) * 32 + col] = db2;
    __syncthreads();
    if (threadIdx.x < 96) {
        const int k = threadIdx.x / 32, cc = threadIdx.x % 32;
        float4 s = f4(0.f);
        for (int hh = 0; hh < 16; ++hh) s += red[(k * 16 + hh) * 32 + cc];
        st4(part + (static_cast<size_t>(blockIdx.x) * 3 + k) * kC + 4 * cc, s);
    }
}

// ------------------------------------------------------------------- backward: weight gradients --
// MODE 1: T = h  = relu(x W1^T + b1)   (recomputed), dW2[c][j] += sum_r dz[r][c] h[r][j]; writes the mask
//                                       bits in THIS kernel's lane layout for MODE 2
// MODE 2: T = dh = (dz W2) * mask,      dW1[j][c] += sum_r dh[r][j] x[r][c], db1[j] += sum_r dh[r][j]
// partials: part_w[block][128 x 384 or 384 x 128], part_b[block][384] (MODE 2)
template <int MODE>
__global__ __launch_bounds__(512, 2) void ffn_bwd_dw_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dz,
                                                                const float* __restrict__ b1, const bf16x8* __restrict__ pk,
                                                                unsigned* __restrict__ bits_io, float* __restrict__ part_w,
                                                                float* __restrict__ part_b, int64_t R) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xbuf = smem;                        // [2][64][128] bf16
    char* zbuf = smem + 2 * kXBytes;          // [2][64][128] bf16
    char* tt = zbuf + 2 * kXBytes;            // transposed T tile [384][144 B]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, col = lane & 31;
    const int a = w & 3, b = w >> 2;          // weight-gradient tiles: 128-side block a, 384-side blocks 6 b .. 6 b + 5
    const int64_t tiles = (R + kRowsPerTile - 1) / kRowsPerTile;

    bf16x8 wf[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[i][ks] = pk[(MODE == 1 ? kP16W1 : kP16W2T) + ((3 * w + i) * 4 + ks) * 64 + lane];
    float bias[3] = {0.f, 0.f, 0.f};
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 3; ++i) bias[i] = b1[48 * w + 16 * i + (lane & 15)];
    }
    f32x16 wacc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) wacc[t][i] = 0.f;
    float bsum[3] = {0.f, 0.f, 0.f};
    wait_all_vmem_visible();

    auto stage = [&](int64_t t, int bufi) {
        const int64_t r0 = t * kRowsPerTile;
        if (r0 + kRowsPerTile > R) {          // tail tile (block-uniform): rows past the end must be ZERO operands
            for (int i = threadIdx.x; i < kXBytes / 16; i += 512) {
                *reinterpret_cast<float4*>(xbuf + bufi * kXBytes + i * 16) = f4(0.f);
                *reinterpret_cast<float4*>(zbuf + bufi * kXBytes + i * 16) = f4(0.f);
            }
            __syncthreads();
        }
        dma_tile_bf16<kC, 8>(x, r0, R, xbuf + bufi * kXBytes, w, lane);
        dma_tile_bf16<kC, 8>(dz, r0, R, zbuf + bufi * kXBytes, w, lane);
    };
    int64_t tix = blockIdx.x;
    if (tix < tiles) stage(tix, 0);
    int buf = 0;
    for (; tix < tiles; tix += gridDim.x, buf ^= 1) {
        wait_all_vmem();
        __syncthreads();      // tiles landed; previous T tile fully consumed
        if (tix + gridDim.x < tiles) stage(tix + gridDim.x, buf ^ 1);
        const char* xt = xbuf + buf * kXBytes;
        const char* zt = zbuf + buf * kXBytes;
        // ---- T tile, transposed: lane = hidden channel 48 w + 16 i + (lane & 15), rows 16 mb + 4 (lane >> 4) + reg
        const size_t bix = (static_cast<size_t>(tix) * 512 + threadIdx.x) * 2;
        unsigned long long bits = 0ull;
        if (MODE == 2) bits = static_cast<unsigned long long>(bits_io[bix]) | (static_cast<unsigned long long>(bits_io[bix + 1]) << 32);
        gemm_up_tile<false, false>(MODE == 1 ? xt : zt, wf, lane, [&](int mb, const f32x4 (&acc)[3]) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float4 v = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
                if (MODE == 1) {
                    v = v + f4(bias[i]);
                    const unsigned nib = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
                    bits |= static_cast<unsigned long long>(nib) << (mb * 12 + i * 4);
                    v = max4(v, f4(0.f));
                } else {
                    const unsigned m4 = static_cast<unsigned>(bits >> (mb * 12 + i * 4));
                    v = make_float4(m4 & 1u ? v.x : 0.f, m4 & 2u ? v.y : 0.f, m4 & 4u ? v.z : 0.f, m4 & 8u ? v.w : 0.f);
                }
                const u32x2_t pv = pack4_bf16(v);
                if (MODE == 2) {
                    const float4 r4 = unpack4_bf16(pv);
                    bsum[i] += (r4.x + r4.y) + (r4.z + r4.w);
                }
                *reinterpret_cast<u32x2_t*>(tt + (48 * w + 16 * i + (lane & 15)) * kTPitch + (16 * mb + 4 * (lane >> 4)) * 2) = pv;
            }
        });
        if (MODE == 1 && bits_io) {
            bits_io[bix] = static_cast<unsigned>(bits);
            bits_io[bix + 1] = static_cast<unsigned>(bits >> 32);
        }
        __syncthreads();
        // ---- weight gradient: contraction over the 64 rows of the tile, 16 per MFMA
        const char* ut = MODE == 1 ? zt : xt;        // row-major operand (gathered 2 bytes at a time)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4_t pku;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 16 * s + 8 * half + 2 * j;
                const unsigned lo = *reinterpret_cast<const unsigned short*>(ut + tile_off(r, 32 * a + col, kC));
                const unsigned hi = *reinterpret_cast<const unsigned short*>(ut + tile_off(r + 1, 32 * a + col, kC));
                pku[j] = lo | (hi << 16);
            }
            const bf16x8 uf = __builtin_bit_cast(bf16x8, pku);
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const bf16x8 tf = *reinterpret_cast<const bf16x8*>(tt + (32 * (6 * b + t) + col) * kTPitch + (16 * s + 8 * half) * 2);
                wacc[t] = MODE == 1 ? mfma32(uf, tf, wacc[t]) : mfma32(tf, uf, wacc[t]);
            }
            __builtin_amdgcn_sched_barrier(0);     // one 16-row step at a time: bounds the fragments in flight
        }
    }
    // ---- partials
    float* pw = part_w + static_cast<size_t>(blockIdx.x) * kC * kH;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = (reg & 3) + 8 * (reg >> 2) + 4 * half;
            if (MODE == 1) pw[(32 * a + m) * kH + 32 * (6 * b + t) + col] = wacc[t][reg];       // dW2 [128][384]
            else pw[(32 * (6 * b + t) + m) * kC + 32 * a + col] = wacc[t][reg];                 // dW1 [384][128]
        }
    if (MODE == 2 && part_b) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float s = bsum[i];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (lane < 16) part_b[static_cast<size_t>(blockIdx.x) * kH + 48 * w + 16 * i + lane] = s;
        }
    }
}

// out[i] = sum_s part[s * n + i] in a fixed order (small n)
__global__ __launch_bounds__(256) void ffn_reduce_small_kernel(const float* __restrict__ part, int S, int n,
                                                             float* __restrict__ o0, float* __restrict__ o1,
                                                             float* __restrict__ o2, int per_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll 8
    for (int p = 0; p < S; ++p) s += part[static_cast<size_t>(p) * n + i];
    float* out = i < per_out ? o0 : (i < 2 * per_out ? o1 : o2);
    if (out) out[i % per_out] = s;
}

int ffn_grid(int64_t R) {
    const int64_t tiles = (R + kRowsPerTile - 1) / kRowsPerTile;
    return static_cast<int>(tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256);
}

}  // namespace
}  // namespace dg

using namespace dg;

extern "C" size_t dg_ffn_bf16_packed_bytes(void) { return static_cast<size_t>(4) * kSec * 16; }

extern "C" int dg_ffn_bf16_pack(const float* w1, const float* w2, void* packed, dg_stream_t stream_) {
    if (!w1 || !w2 || !packed) return fail(DG_E_ARG, "dg_ffn_bf16_pack: null pointer");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    bf16x8* p = static_cast<bf16x8*>(packed);
    int st = pack_bf16(w1, p + kP16W1, kH, kC, 0, 16, stream);            // W1 [384,128]: W' = W1
    if (!st) st = pack_bf16(w2, p + kP16W2, kC, kH, 0, 16, stream);       // W2 [128,384]: W' = W2
    if (!st) st = pack_bf16(w2, p + kP16W2T, kC, kH, 1, 16, stream);      // W' = W2^T [384,128]
    if (!st) st = pack_bf16(w1, p + kP16W1T, kH, kC, 1, 16, stream);      // W' = W1^T [128,384]
    return st;
}

extern "C" int64_t dg_ffn_bf16_padded_rows(int64_t R) { return R < 1 ? 0 : (R + kRowsPerTile - 1) / kRowsPerTile * kRowsPerTile; }

extern "C" size_t dg_ffn_bf16_mask_words(int64_t R) {
    return R < 1 ? 0 : static_cast<size_t>((R + kRowsPerTile - 1) / kRowsPerTile) * 512 * 2;
}

extern "C" size_t dg_ffn_bf16_workspace_bytes(int64_t R) {
    if (R < 1) return 0;
    const size_t grid = static_cast<size_t>(ffn_grid(R));
    return grid * (static_cast<size_t>(kC) * kH + kH + 3 * kC) * sizeof(float);
}

extern "C" int dg_ffn_ln_fwd_bf16(const void* x, const void* packed, const float* b1, const float* b2, const float* gamma,
                                  const float* beta, void* y, void* pre, float* mean, float* rstd, unsigned* relu_bits,
                                  int64_t R, float eps, dg_stream_t stream_) {
    if (!x || !packed || !b1 || !b2 || !gamma || !beta || !y || !mean || !rstd)
        return fail(DG_E_ARG, "dg_ffn_ln_fwd_bf16: null pointer");      // pre, relu_bits: only needed for a backward
    if (R < 0) return fail(DG_E_SHAPE, "dg_ffn_ln_fwd_bf16: negative row count");
    if (R == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ProfScope prof(R < edge_rows() ? DG_K_FFN_NODE : DG_K_FFN, stream);
    // lean kernel: stores whole 64-row tiles -- y, pre_ln, mean, rstd must hold dg_ffn_bf16_padded_rows(R) rows
    constexpr int lds = 2 * kXBytes + kHBytes + kZBytes + 1024;
    const bool save = pre != nullptr && relu_bits != nullptr;
    if (save) {
        DG_OPT_IN_LDS((&ffn_fwd_bf16_v2_kernel<true>), lds);
        hipLaunchKernelGGL(ffn_fwd_bf16_v2_kernel<true>, dim3(ffn_grid(R)), dim3(512), lds, stream,
                           static_cast<const bf16_t*>(x), static_cast<const bf16x8*>(packed), b1, b2, gamma, beta,
                           static_cast<bf16_t*>(y), static_cast<bf16_t*>(pre), mean, rstd, relu_bits, R, eps);
    } else {
        DG_OPT_IN_LDS((&ffn_fwd_bf16_v2_kernel<false>), lds);
        hipLaunchKernelGGL(ffn_fwd_bf16_v2_kernel<false>, dim3(ffn_grid(R)), dim3(512), lds, stream,
                           static_cast<const bf16_t*>(x), static_cast<const bf16x8*>(packed), b1, b2, gamma, beta,
                           static_cast<bf16_t*>(y), static_cast<bf16_t*>(pre), mean, rstd, relu_bits, R, eps);
    }
    return check_launch("dg_ffn_ln_fwd_bf16");
}

extern "C" int dg_ffn_ln_bwd_bf16(const void* x, const void* pre, const float* mean, const float* rstd,
                                  const unsigned* relu_bits, const float* gamma, const void* packed, const float* b1,
                                  const void* dy, void* dz, void* dx, float* dgamma, float* dbeta, float* dw1, float* db1,
                                  float* dw2, float* db2, unsigned* bits_scratch, void* workspace, size_t workspace_bytes,
                                  int64_t R, dg_stream_t stream_) {
    if (!x || !pre || !mean || !rstd || !relu_bits || !gamma || !packed || !b1 || !dy || !dz || !workspace)
        return fail(DG_E_ARG, "dg_ffn_ln_bwd_bf16: null pointer");
    if (dw1 && (!dw2 || !db1 || !db2 || !bits_scratch))
        return fail(DG_E_ARG, "dg_ffn_ln_bwd_bf16: weight gradients need dw1, db1, dw2, db2 and bits_scratch together");
    if (R < 1) return fail(DG_E_SHAPE, "dg_ffn_ln_bwd_bf16: empty input");
    if (workspace_bytes < dg_ffn_bf16_workspace_bytes(R)) return fail(DG_E_WORKSPACE, "dg_ffn_ln_bwd_bf16: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int grid = ffn_grid(R);
    float* part_w = static_cast<float*>(workspace);
    float* part_b = part_w + static_cast<size_t>(grid) * kC * kH;
    float* part_ln = part_b + static_cast<size_t>(grid) * kH;
    const bf16x8* pk = static_cast<const bf16x8*>(packed);
    {
        constexpr int lds = kXBytes + kHBytes + kZBytes;
        DG_OPT_IN_LDS((&ffn_bwd_dx_bf16_kernel), lds);
        ProfScope prof(R < edge_rows() ? DG_K_FFN_NODE : DG_K_FFN, stream);
        hipLaunchKernelGGL(ffn_bwd_dx_bf16_kernel, dim3(grid), dim3(512), lds, stream, static_cast<const bf16_t*>(dy),
                           static_cast<const bf16_t*>(pre), mean, rstd, relu_bits, gamma, pk, static_cast<bf16_t*>(dz),
                           static_cast<bf16_t*>(dx), part_ln, R);
    }
    hipLaunchKernelGGL(ffn_reduce_small_kernel, dim3((3 * kC + 255) / 256), dim3(256), 0, stream, part_ln, grid, 3 * kC,
                       dgamma, dbeta, db2, kC);
    if (dw1) {
        constexpr int lds = 4 * kXBytes + kTBytes;
        DG_OPT_IN_LDS((&ffn_bwd_dw_bf16_kernel<1>), lds);
        DG_OPT_IN_LDS((&ffn_bwd_dw_bf16_kernel<2>), lds);
        {
            ProfScope prof(R < edge_rows() ? DG_K_FFN_WGRAD_NODE : DG_K_FFN_WGRAD, stream);
            hipLaunchKernelGGL((ffn_bwd_dw_bf16_kernel<1>), dim3(grid), dim3(512), lds, stream, static_cast<const bf16_t*>(x),
                               static_cast<const bf16_t*>(dz), b1, pk, bits_scratch, part_w, static_cast<float*>(nullptr), R);
        }
        launch_splitk_reduce(part_w, grid, static_cast<int64_t>(kC) * kH / 4, dw2, stream);
        {
            ProfScope prof(R < edge_rows() ? DG_K_FFN_WGRAD_NODE : DG_K_FFN_WGRAD, stream);
            hipLaunchKernelGGL((ffn_bwd_dw_bf16_kernel<2>), dim3(grid), dim3(512), lds, stream, static_cast<const bf16_t*>(x),
                               static_cast<const bf16_t*>(dz), b1, pk, bits_scratch, part_w, part_b, R);
        }
        launch_splitk_reduce(part_w, grid, static_cast<int64_t>(kC) * kH / 4, dw1, stream);
        hipLaunchKernelGGL(ffn_reduce_small_kernel, dim3((kH + 255) / 256), dim3(256), 0, stream, part_b, grid, kH, db1,
                           static_cast<float*>(nullptr), static_cast<float*>(nullptr), kH);
    }
    return check_launch("dg_ffn_ln_bwd_bf16");
}
