// Backward of the edge embedding + symmetrisation (reference src/model/models.py:57-61,92-94 / 159-163,197-199) for the
// bf16 configuration and the piecewise-linear activations (relu, leaky -- the reference's defaults):
//
//     h = act(W1 a + b1)  [64]      f = act(W2 h + b2)  [128]      out_ij = (f_ij + f_ji) / 2
//     given g = d loss / d out:     gs_ij = (g_ij + g_ji) / 2
//     dpre2 = gs * act'(pre2)       dW2 += dpre2^T h      db2 += sum dpre2      dh = dpre2 W2
//     dpre1 = dh * act'(pre1)       dW1 += dpre1^T a      db1 += sum dpre1      da = dpre1 W1
//
// The general kernel (embed_sym.hip: 32 atom pairs per tile, fp32-class bf16x3 products, gathers of single rows) runs at
// 0.05-0.1 of the HBM roof with bf16 activations (2.0-2.5 ms per launch at B = 2048); this one streams whole row blocks
// (1.1-1.3 ms; VALU-issue bound: ~450 vector instructions per wave and tile, memory is not the limit -- reading g[b,i,:,:]
// twice instead of the strided g[b,:,i,:] does not change the time):
//   * tile = one (molecule b, atom i): the N rows g[b,i,:,:] (contiguous) AND the N rows g[b,:,i,:] (stride N C) arrive
//     by LDS-DMA, double buffered; their sum is taken in registers after ds_read_b64_tr_b16 brings both into the
//     per-channel layout (lane = channel 16 w + (lane & 15), rows 16 mb + 4 (lane >> 4) + r) that the unswapped
//     pre2 = h W2^T product delivers;
//   * h is recomputed on the VALU (E <= 8 inputs) into an LDS tile; dpre2 (bf16) is at the same time the A operand of the
//     weight-gradient MFMAs (accumulator-layout trick, see attn_half.hip) and, through an LDS tile, the operand of
//     dh = dpre2 W2; dW2 [128,64] stays in registers for the kernel's lifetime (16 per lane);
//   * dW1 / db1 / db2 are per-lane VALU accumulators, da = dpre1 W1 one more small MFMA product.
// One bf16 MFMA per product (the bf16 configuration's arithmetic), fp32 accumulation.  HBM traffic: g twice (the second
// read of a row comes from the L2 / MALL when it is lucky), a once, da once.
#include "gemm_bf16.h"

namespace dg {

void launch_splitk_reduce(const float* part, int S, int64_t n4, float* out, hipStream_t stream);

namespace {

constexpr int kC = 128, kH = 64, kEP = 8;
constexpr int kHalf = kH * kEP + kH;           // dW1 [64][8] + db1 [64] of one wave half
constexpr int kSmall = 2 * kHalf + kC;         // per-block partials: two halves (waves 0..3 / 4..7), then db2 [128]

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 cat8(u32x2_t lo, u32x2_t hi) {
    const u32x4_t t = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, t);
}
// swizzle of the [rows][128] tiles (256-byte pitch): see bswz in attn_half.hip
__device__ __forceinline__ int swz16(int row) {
    const int r = row & 15;
    const int p = (r & 3) | ((((r >> 2) ^ (r >> 3)) & 1) << 2);
    return (p << 1) | (r >> 3);
}
__device__ __forceinline__ void lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// ds_read_b64_tr_b16 (compiler builtin: tracked by hipcc's scoreboard)
__device__ __forceinline__ u32x2_t tr_read(unsigned addr) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>(static_cast<size_t>(addr)));
    return __builtin_bit_cast(u32x2_t, v);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 up2(unsigned w) { return f32x2{lo_bf16(w), hi_bf16(w)}; }

struct EmbBwdArgs {
    const float* a;        // [B,N,N,E]
    const float* w1;       // [64,E]
    const float* b1;
    const float* w2;       // [128,64]
    const float* b2;
    const bf16_t* g;       // [B,N,N,128]
    float* da;             // [B,N,N,E] or null
    float* part_w2;        // [grid][128 * 64]
    float* part_small;     // [grid][kSmall]
    int B, N, E;
    float slope;           // act'(x) for x <= 0: 0 (relu) or 0.01 (leaky)
};

template <int MB, bool DA>
__global__ __launch_bounds__(512, 2) void embed_bwd_bf16_kernel(const EmbBwdArgs p) {
    constexpr int ROWS = 16 * MB;
    constexpr int GB = ROWS * 256;            // one [ROWS][128] bf16 tile
    constexpr int HB = ROWS * 128;            // one [ROWS][64] bf16 tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [buf][g_ij | g_ji] tiles, then the dpre2 tile, the h and dpre1 tiles, the a rows
    char* dp2 = smem + 4 * GB;                // [ROWS][128] bf16
    char* ht = dp2 + GB;                      // [ROWS][64] bf16
    char* dp1 = ht + HB;                      // [2][ROWS][64] bf16: dpre1 of this tile / of the previous one (da runs one tile late)
    char* hl = dp1 + 2 * HB;                  // [ROWS][64] bf16: h - bf16(h) (only the SIGN of pre2 needs it, see below)
    float* abuf = reinterpret_cast<float*>(hl + HB);       // [2][ROWS][8]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int N = p.N, E = p.E;

    // ---- weights as bf16 MFMA fragments, straight from the fp32 parameters
    // pre2 = h W2^T: B operand, this wave's 16 output channels, K = 64.  The ReLU mask act'(pre2) multiplies every
    // gradient of this kernel: with single bf16 products ~0.1 % of the signs come out different from the forward's
    // (|pre2| below the bf16 rounding of h and W2) and each flip is an O(1) error of its element (measured 3.6e-2 on every
    // output).  pre2 is therefore recomputed from hi + lo splits of h and W2 (three products, 18 small MFMAs per tile).
    bf16x8 w2f[2], w2l[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float wv = p.w2[(16 * w + r16) * kH + 32 * ks + 8 * kq + j];
            w2f[ks][j] = static_cast<__bf16>(wv);
            w2l[ks][j] = static_cast<__bf16>(wv - static_cast<float>(w2f[ks][j]));
        }
    // dh = dpre2 W2: wave w takes the hidden units [16 (w & 3), +16) of the row blocks of its half (waves 0..3: blocks
    // [0, MBA), waves 4..7: [MBA, MB)): W'[u][c] = W2[c][u], K = 128
    const int nbh = w & 3;
    bf16x8 w2t[4];
    bf16x8 w1t[2];         // da = dpre1 W1 (waves 4..): W'[e][u] = W1[u][e], rows e >= E are zero, K = 64
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) w2t[ks][j] = static_cast<__bf16>(p.w2[(32 * ks + 8 * kq + j) * kH + 16 * nbh + r16]);
    if (DA && w >= 4) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                w1t[ks][j] = static_cast<__bf16>(r16 < E ? p.w1[(32 * ks + 8 * kq + j) * E + r16] : 0.f);
    }
    const float b2c = p.b2[16 * w + r16];
    // layer 1 on the VALU: lane = hidden unit, wave = row residue (rows w, w + 8, ...)
    float w1u[kEP];
#pragma unroll
    for (int e = 0; e < kEP; ++e) w1u[e] = e < E ? p.w1[lane * E + e] : 0.f;
    const float b1u = p.b1[lane];

    // zero the tile rows the DMA never writes (both buffers, both tiles) and the a rows
    for (int idx = threadIdx.x; idx < (ROWS - N) * 16 * 4; idx += 512) {
        const int bufi = idx / ((ROWS - N) * 16), rem = idx % ((ROWS - N) * 16);
        *reinterpret_cast<float4*>(smem + bufi * GB + N * 256 + rem * 16) = f4(0.f);
    }
    for (int idx = threadIdx.x; idx < 2 * ROWS * kEP; idx += 512) abuf[idx] = 0.f;

    // per-lane LDS offsets
    unsigned hf_off[2];           // A fragment of the h / dpre1 tiles: row r16, k-step ks (row blocks add 16 * 128)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) hf_off[ks] = r16 * 128 + (((4 * ks + kq) ^ (r16 & 7)) << 4);
    unsigned df_off[4];           // A fragment of the dpre2 tile
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) df_off[ks] = r16 * 256 + (((4 * ks + kq) ^ swz16(r16)) << 4);
    const int trr = 4 * kq + (r16 >> 2);
    // transposing-read source of this lane in a [rows][128] tile, channel block 0 (block n: ^ (n << 5)), row block 0
    const unsigned trg = lds_byte_address(smem) + trr * 256 + (((((r16 >> 1) & 1) ^ swz16(trr)) & 15) << 4) + (r16 & 1) * 8;
    // ... in a [rows][64] tile (h, 128-byte pitch, chunk ^ (row & 7)), unit block 0 (block n: ^ (n << 5))
    const unsigned trh = lds_byte_address(ht) + trr * 128 + (((((r16 >> 1) & 1) ^ trr) & 7) << 4) + (r16 & 1) * 8;
    unsigned w2_base[4], w1_base[4];      // 2-byte stores of dpre2 / dpre1 from the per-channel layout, per r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        w2_base[r] = (4 * kq + r) * 256 + (((2 * w + (r16 >> 3)) ^ swz16(4 * kq + r)) << 4) + (r16 & 7) * 2;
        w1_base[r] = (4 * kq + r) * 128 + ((((2 * nbh + (r16 >> 3)) ^ (4 * kq + r)) & 7) << 4) + (r16 & 7) * 2;
    }

    f32x4 accW2[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) accW2[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float db2 = 0.f, db1 = 0.f, dw1[kEP];
#pragma unroll
    for (int e = 0; e < kEP; ++e) dw1[e] = 0.f;

    const long long tiles = static_cast<long long>(p.B) * N;
    long long t = tiles * blockIdx.x / gridDim.x;
    const long long t_end = tiles * (blockIdx.x + 1) / gridDim.x;
    if (t >= t_end) return;
    auto dma_tile = [&](long long tile, int bufi) {
        const int b = static_cast<int>(tile / N), i = static_cast<int>(tile % N);
        const unsigned d1 = lds_byte_address(smem + bufi * 2 * GB), d2 = d1 + GB;
        const bf16_t* g1 = p.g + static_cast<size_t>(tile) * N * kC;                       // rows j of g[b,i,:,:]
        const bf16_t* g2 = p.g + (static_cast<size_t>(b) * N * N + i) * kC;                // rows j of g[b,:,i,:], stride N C
#pragma unroll
        for (int ii0 = 0; ii0 < 4 * MB; ii0 += 8) {
            const int ii = ii0 + w;
            const int L = ii * 64 + lane;
            const int row = L >> 4, cpos = L & 15;
            if (ii < 4 * MB && row < N) {
                const int sc = (cpos ^ swz16(row)) << 3;
                dma16_async(reinterpret_cast<const float*>(g1 + static_cast<size_t>(row) * kC + sc), d1 + ii * 1024);
                dma16_async(reinterpret_cast<const float*>(g2 + static_cast<size_t>(row) * N * kC + sc), d2 + ii * 1024);
            }
        }
    };
    __syncthreads();
    dma_tile(t, 0);
    {   // a rows of the first tile
        const int idx = threadIdx.x;
        if (idx < N * E) abuf[(idx / E) * kEP + idx % E] = p.a[static_cast<size_t>(t) * N * E + idx];
    }
    wait_all_vmem_visible();
    const long long t_first = t;
    // da[row][e] = sum_u dpre1[row][u] W1[u][e] for row block mb of `tile`, from a dpre1 tile
    auto da_block = [&](const char* src, int mb, long long tile) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) o = mfma16(*reinterpret_cast<const bf16x8*>(src + mb * 2048 + hf_off[ks]), w1t[ks], o);
        if (r16 < E) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * mb + 4 * kq + r;
                if (row < N) p.da[(static_cast<size_t>(tile) * N + row) * E + r16] = o[r];
            }
        }
    };
    int buf = 0;
    for (;; buf ^= 1, ++t) {
        const bool more = t + 1 < t_end;
        lds_barrier();            // tile landed for every wave; dp2 / h tiles and this tile's dpre1 buffer are free
        float a_next = 0.f;
        if (more) {
            dma_tile(t + 1, buf ^ 1);
            if (threadIdx.x < N * E) a_next = p.a[static_cast<size_t>(t + 1) * N * E + threadIdx.x];
        }
        const float* at = abuf + buf * ROWS * kEP;
        // ---- layer 1: h[row][unit] for rows w, w + 8, ... (lane = unit), bf16 -> h tile
#pragma unroll
        for (int rr = 0; rr < 2 * MB; ++rr) {
            const int row = w + 8 * rr;
            const float4 a0 = *reinterpret_cast<const float4*>(at + row * kEP), a1 = *reinterpret_cast<const float4*>(at + row * kEP + 4);
            float s = b1u;
            s = fmaf(w1u[0], a0.x, s); s = fmaf(w1u[1], a0.y, s); s = fmaf(w1u[2], a0.z, s); s = fmaf(w1u[3], a0.w, s);
            s = fmaf(w1u[4], a1.x, s); s = fmaf(w1u[5], a1.y, s); s = fmaf(w1u[6], a1.z, s); s = fmaf(w1u[7], a1.w, s);
            s = s > 0.f ? s : p.slope * s;
            const unsigned ho = row * 128 + ((((lane >> 3) ^ row) & 7) << 4) + (lane & 7) * 2;
            const __bf16 sh = static_cast<__bf16>(s);
            *reinterpret_cast<bf16_t*>(ht + ho) = sh;
            *reinterpret_cast<bf16_t*>(hl + ho) = static_cast<__bf16>(s - static_cast<float>(sh));
        }
        lds_barrier();
        // ---- pre2 = h W2^T + b2 (per-channel layout), gs = (g_ij + g_ji) / 2, dpre2 = gs act'(pre2)
        const unsigned gcur = trg + buf * (2 * GB);
        u32x2_t dA[MB];
        {
            f32x4 pre[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) pre[mb] = f32x4{b2c, b2c, b2c, b2c};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const bf16x8 hh = *reinterpret_cast<const bf16x8*>(ht + mb * (16 * 128) + hf_off[ks]);
                    const bf16x8 lo = *reinterpret_cast<const bf16x8*>(hl + mb * (16 * 128) + hf_off[ks]);
                    pre[mb] = mfma16(lo, w2f[ks], pre[mb]);
                    pre[mb] = mfma16(hh, w2l[ks], pre[mb]);
                    pre[mb] = mfma16(hh, w2f[ks], pre[mb]);
                }
            const unsigned ga = gcur ^ (w << 5);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const u32x2_t x1 = tr_read(ga + mb * 4096), x2 = tr_read(ga + GB + mb * 4096);
                f32x2 d0 = (up2(x1[0]) + up2(x2[0])) * f32x2{0.5f, 0.5f}, d1 = (up2(x1[1]) + up2(x2[1])) * f32x2{0.5f, 0.5f};
                d0[0] = pre[mb][0] > 0.f ? d0[0] : p.slope * d0[0];
                d0[1] = pre[mb][1] > 0.f ? d0[1] : p.slope * d0[1];
                d1[0] = pre[mb][2] > 0.f ? d1[0] : p.slope * d1[0];
                d1[1] = pre[mb][3] > 0.f ? d1[1] : p.slope * d1[1];
                dA[mb][0] = pack_bf16(d0[0], d0[1]);
                dA[mb][1] = pack_bf16(d1[0], d1[1]);
                const f32x2 q0 = up2(dA[mb][0]), q1 = up2(dA[mb][1]);      // db2 of the values the GEMMs see
                db2 += (q0[0] + q0[1]) + (q1[0] + q1[1]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<unsigned short*>(dp2 + w2_base[r] + mb * 4096) =
                        static_cast<unsigned short>(dA[mb][r >> 1] >> (16 * (r & 1)));
            }
        }
        // ---- dW2[c][k] += dpre2^T h   (A from registers, B = h by transposing reads; one opcode, zero-padded odd block)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const unsigned ha = trh ^ (n << 5);
            u32x2_t hb[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) hb[mb] = tr_read(ha + mb * 2048);
#pragma unroll
            for (int pp = 0; pp < MB / 2; ++pp)
                accW2[n] = mfma16(cat8(dA[2 * pp], dA[2 * pp + 1]), cat8(hb[2 * pp], hb[2 * pp + 1]), accW2[n]);
            if (MB & 1) accW2[n] = mfma16(cat8(dA[MB - 1], u32x2_t{0u, 0u}), cat8(hb[MB - 1], u32x2_t{0u, 0u}), accW2[n]);
        }
        lds_barrier();
        // ---- dh = dpre2 W2 for hidden units 16 nbh + r16 and this half's row blocks, dpre1 = dh act'(pre1), dW1 / db1,
        // dpre1 -> tile.  Waves 4.. also produce da of the PREVIOUS tile from the other dpre1 buffer (no extra barrier).
        {
            constexpr int MBA = (MB + 1) / 2;
            const int mb_lo = w < 4 ? 0 : MBA, mb_hi = w < 4 ? MBA : MB;
            char* dcur = dp1 + buf * HB;
            const unsigned ha = trh ^ (nbh << 5);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                if (mb < mb_lo || mb >= mb_hi) continue;      // wave-uniform
                f32x4 dh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    dh = mfma16(*reinterpret_cast<const bf16x8*>(dp2 + mb * 4096 + df_off[ks]), w2t[ks], dh);
                const u32x2_t hv = tr_read(ha + mb * 2048);        // h of (rows, this unit): act'(pre1) from its sign
                const f32x2 h0 = up2(hv[0]), h1 = up2(hv[1]);
                float d[4];
                d[0] = h0[0] > 0.f ? dh[0] : p.slope * dh[0];
                d[1] = h0[1] > 0.f ? dh[1] : p.slope * dh[1];
                d[2] = h1[0] > 0.f ? dh[2] : p.slope * dh[2];
                d[3] = h1[1] > 0.f ? dh[3] : p.slope * dh[3];
                const unsigned k0 = pack_bf16(d[0], d[1]), k1 = pack_bf16(d[2], d[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * mb + 4 * kq + r;
                    db1 += d[r];
                    const float4 a0 = *reinterpret_cast<const float4*>(at + row * kEP), a1 = *reinterpret_cast<const float4*>(at + row * kEP + 4);
                    dw1[0] = fmaf(d[r], a0.x, dw1[0]); dw1[1] = fmaf(d[r], a0.y, dw1[1]);
                    dw1[2] = fmaf(d[r], a0.z, dw1[2]); dw1[3] = fmaf(d[r], a0.w, dw1[3]);
                    dw1[4] = fmaf(d[r], a1.x, dw1[4]); dw1[5] = fmaf(d[r], a1.y, dw1[5]);
                    dw1[6] = fmaf(d[r], a1.z, dw1[6]); dw1[7] = fmaf(d[r], a1.w, dw1[7]);
                    if (DA)
                        *reinterpret_cast<unsigned short*>(dcur + w1_base[r] + mb * 2048) =
                            static_cast<unsigned short>((r < 2 ? k0 : k1) >> (16 * (r & 1)));
                }
            }
        }
        if (DA && w >= 4 && w - 4 < MB && t > t_first) da_block(dp1 + (buf ^ 1) * HB, w - 4, t - 1);
        // a rows of the next tile -> the other buffer; its DMA is waited for here, behind this tile's work
        wait_all_vmem_visible();
        if (more && threadIdx.x < N * E) abuf[(buf ^ 1) * ROWS * kEP + (threadIdx.x / E) * kEP + threadIdx.x % E] = a_next;
        if (!more) break;
    }
    if (DA) {       // da of the last tile
        lds_barrier();
        if (w >= 4 && w - 4 < MB) da_block(dp1 + buf * HB, w - 4, t);
    }
    // ---- per-workgroup partials
    float* pw = p.part_w2 + static_cast<size_t>(blockIdx.x) * kC * kH;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[(16 * w + 4 * kq + r) * kH + 16 * n + r16] = accW2[n][r];      // dW2[c][k]
    float* ps = p.part_small + static_cast<size_t>(blockIdx.x) * kSmall;
    const float db2_all = xor_sum<16>(db2);
    if (kq == 0) ps[2 * kHalf + 16 * w + r16] = db2_all;
    {
        float* ph = ps + (w >> 2) * kHalf;
        const float db1_all = xor_sum<16>(db1);
        if (kq == 0) ph[kH * kEP + 16 * nbh + r16] = db1_all;
#pragma unroll
        for (int e = 0; e < kEP; ++e) {
            const float v = xor_sum<16>(dw1[e]);
            if (kq == 0) ph[(16 * nbh + r16) * kEP + e] = v;
        }
    }
}

// dw1 [64][E], db1 [64], db2 [128] from the per-block partials (fixed order)
__global__ __launch_bounds__(256) void embed_small_reduce_kernel(const float* __restrict__ part, int S, int E,
                                                               float* __restrict__ dw1, float* __restrict__ db1,
                                                               float* __restrict__ db2) {
    const int i = blockIdx.x * 256 + threadIdx.x;      // over kHalf + kC outputs
    if (i >= kHalf + kC) return;
    float s = 0.f;
    if (i < kHalf) {
#pragma unroll 4
        for (int q = 0; q < S; ++q) s += part[static_cast<size_t>(q) * kSmall + i] + part[static_cast<size_t>(q) * kSmall + kHalf + i];
    } else {
#pragma unroll 8
        for (int q = 0; q < S; ++q) s += part[static_cast<size_t>(q) * kSmall + kHalf + i];
    }
    if (i < kH * kEP) {
        const int u = i / kEP, e = i % kEP;
        if (e < E) dw1[u * E + e] = s;
    } else if (i < kHalf) {
        db1[i - kH * kEP] = s;
    } else {
        db2[i - kHalf] = s;
    }
}

int emb_grid(long long tiles) { return static_cast<int>(tiles < 512 ? tiles : 512); }

}  // namespace

bool embed_bwd_bf16_ok(int N, int E, int H, int C, int act) {
    return N >= 1 && N <= 48 && E >= 1 && E <= kEP && H == kH && C == kC && (act == 0 || act == 1);
}

size_t embed_bwd_bf16_workspace_bytes(int B, int N) {
    const int grid = emb_grid(static_cast<long long>(B) * N);
    return static_cast<size_t>(grid) * (kC * kH + kSmall) * sizeof(float);
}

// a, w1, b1, w2 (RAW fp32 [128,64]), b2; g bf16; da may be null
int embed_bwd_bf16(const float* a, const float* w1, const float* b1, const float* w2, const float* b2, const bf16_t* g,
                   float* da, float* dw1, float* db1, float* dw2, float* db2, void* workspace, int B, int N, int E, int act,
                   hipStream_t stream) {
    EmbBwdArgs p;
    p.a = a; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.g = g; p.da = da;
    const int grid = emb_grid(static_cast<long long>(B) * N);
    p.part_w2 = static_cast<float*>(workspace);
    p.part_small = p.part_w2 + static_cast<size_t>(grid) * kC * kH;
    p.B = B; p.N = N; p.E = E; p.slope = act == 1 ? 0.01f : 0.f;
#define LAUNCH(MB_, DA_)                                                                                     \
    {                                                                                                        \
        constexpr int lds = 16 * MB_ * (5 * 256 + 4 * 128 + 2 * kEP * 4);                                    \
        DG_OPT_IN_LDS((&embed_bwd_bf16_kernel<MB_, DA_>), lds);                                               \
        hipLaunchKernelGGL((embed_bwd_bf16_kernel<MB_, DA_>), dim3(grid), dim3(512), lds, stream, p);         \
    }
#define LAUNCH_MB(MB_) { if (da) LAUNCH(MB_, true) else LAUNCH(MB_, false) }
    if (N <= 16) LAUNCH_MB(1) else if (N <= 32) LAUNCH_MB(2) else LAUNCH_MB(3)
#undef LAUNCH_MB
#undef LAUNCH
    launch_splitk_reduce(p.part_w2, grid, kC * kH / 4, dw2, stream);
    hipLaunchKernelGGL(embed_small_reduce_kernel, dim3((kHalf + kC + 255) / 256), dim3(256), 0, stream, p.part_small, grid, E, dw1,
                       db1, db2);
    return check_launch("dg_embed_sym_bwd(bf16)");
}

}  // namespace dg

using namespace dg;

extern "C" size_t dg_embed_sym_bwd_bf16_workspace_bytes(int B, int N) {
    return B < 1 || N < 1 ? 0 : embed_bwd_bf16_workspace_bytes(B, N);
}

extern "C" int dg_embed_sym_bwd_bf16(const float* a, const float* w1, const float* b1, const float* w2, const float* b2,
                                     const void* g, float* da, float* dw1, float* db1, float* dw2, float* db2,
                                     void* workspace, size_t workspace_bytes, int B, int N, int E, int H, int C, int act,
                                     dg_stream_t stream_) {
    if (!a || !w1 || !b1 || !w2 || !b2 || !g || !dw1 || !db1 || !dw2 || !db2 || !workspace)
        return fail(DG_E_ARG, "dg_embed_sym_bwd_bf16: null pointer");
    if (B < 1 || !embed_bwd_bf16_ok(N, E, H, C, act))
        return fail(DG_E_SHAPE, "dg_embed_sym_bwd_bf16: unsupported B=%d N=%d E=%d H=%d C=%d act=%d (need N<=48, E<=8, H=64, "
                                "C=128, relu or leaky)", B, N, E, H, C, act);
    if (workspace_bytes < embed_bwd_bf16_workspace_bytes(B, N)) return fail(DG_E_WORKSPACE, "dg_embed_sym_bwd_bf16: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ProfScope prof(DG_K_EMBED_SYM, stream);
    return embed_bwd_bf16(a, w1, b1, w2, b2, static_cast<const bf16_t*>(g), da, dw1, db1, dw2, db2, workspace, B, N, E, act, stream);
}
