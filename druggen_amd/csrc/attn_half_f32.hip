// Attention half of an Encoder_Block's edge branch, FORWARD, float32 rows, as ONE kernel (reference src/model/layers.py:
//   e  = y We^T + be                                  :114   (e projection of the edge rows)
//   sc = (q_i k_j / sqrt(d_k)) * (e + 1) * e          :119-125   -- the edge output of the attention, and
//   o_i = sum_j softmax_j(sc) v_j                     :130-134   -- the node output, per channel
//   y2 = LayerNorm(y + sc Woe^T + boe)                :135, 186-188 (out_e, residual, ln4)
// Unfused these are three edge-level launches (e projection, attention core, out_e + residual + LayerNorm): `e` and `sc`
// make a round trip through HBM between them and `y` is read twice -- 2.1 GB for 1.33 GB of results that must exist anyway
// (e, sc, the pre-LayerNorm sum and y2 are what the backward reads).  Here one workgroup walks whole row groups (b, i) -- the
// N <= 48 edge rows (b, i, :) that share a softmax -- in the producer / consumer form of row_gemm_k384.hip:
//
//   waves 8..11  producers: every edge-level global access.  y rows HBM -> registers (buffer loads, three row groups deep; the
//                registers stay until the group's residual add) -> fp16 hi / lo planes under a power-of-two row scale; the
//                score tile the consumers leave in LDS -> HBM (sc, and e) AND -> planes again (A operand of out_e); the
//                out_e tile + residual -> LayerNorm -> y2, the pre-LayerNorm sum, mean / rstd.
//   waves 0..7   consumers: wave w owns channels [16 w, 16 w + 16) of BOTH 128 x 128 weights (2 x 32 VGPRs of fragments).
//                Swapped product on v_mfma_f32_16x16x32_f16: a lane ends up with 4 consecutive channels of rows n, 16 + n,
//                32 + n -- the rows of one channel lie along a 16-lane DPP row, so the softmax over j is three in-lane steps
//                and four DPP steps per reduction.  k_j, v_j and q_i of the lane's rows / channels come from L2.
//   three barriers per row group: e projection + scores | score tile -> planes | out_e ; everything else overlaps.
// Arithmetic of both contractions as in row_gemm.hip (fp16 hi + lo, three products, fp32 accumulation, inverse scales).
// 48 < N <= 96 (BASELINE configs[4], N = 90): a row group takes TWO stages, j in [0, 48) and [48, N) -- the e projection, the
// scores, out_e and the LayerNorm are per edge row, only o_i couples the halves: an online softmax carries (max, sum, sum p v)
// of the first half through a 1.5 KB LDS stash and the second half finishes o_i.  k_j / v_j are fetched per stage there
// (two register sets do not fit next to the two weights' fragments).
#include "common.h"
#include "traversal.h"

#ifndef AH_DBG
#define AH_DBG 0      // ablation builds (scripts/build_variant.sh): 1 no MFMAs, 2 no score / softmax arithmetic, 4 no score tile
#endif                // -> HBM / planes, 8 no finish (residual, LayerNorm, stores), 16 no q / k / v loads, 32 no y split
namespace dg {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kNP = 48;                                   // rows of a stage (row groups are padded to 48 rows)
constexpr int kPlane = 16 * kNP * 16;                     // [k-step 4][k-quarter 4][row 48 (xor-swizzled)][16 B]
constexpr int kPlanes = 2 * kPlane + 256;                 // hi, lo, inverse row scales [48]
constexpr int kTile = kNP * 512;                          // fp32 tile [row 48][16-byte slot 32 (xor row & 7)]
constexpr int kOffPy = 0, kOffPs = kPlanes;
constexpr int kOffTe = 2 * kPlanes, kOffTs = kOffTe + kTile, kOffTo = kOffTs + kTile;
constexpr int kOffTab = kOffTo + kTile;                   // inv column scales + bias of both weights, gamma, beta: 6 x [128]
constexpr int kOffO = kOffTab + 6 * 512;                  // o_i [128]
constexpr int kOffRun = kOffO + 512;                      // two-stage row groups: max, sum, sum p v of the first half [3][128]
constexpr int kLds = kOffRun + 3 * 512;
constexpr int kCons = 8, kProd = 4;
constexpr float kNegBig = -3.0e38f;

template <int CTRL>
__device__ __forceinline__ unsigned umax_dpp(unsigned x) {
    const unsigned moved = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, 0xF, 0xF, true));
    return x > moved ? x : moved;
}
template <int CTRL>
__device__ __forceinline__ float sum_dpp(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ float max_dpp(float x) {
    return fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true)));
}
// over the 16 lanes of a DPP row, result in every lane (quad xor 1, quad xor 2, half-row mirror, row mirror)
__device__ __forceinline__ float row16_sum(float x) {
    x = sum_dpp<0xB1>(x);
    x = sum_dpp<0x4E>(x);
    x = sum_dpp<0x141>(x);
    return sum_dpp<0x140>(x);
}
__device__ __forceinline__ float row16_max(float x) {
    x = max_dpp<0xB1>(x);
    x = max_dpp<0x4E>(x);
    x = max_dpp<0x141>(x);
    return max_dpp<0x140>(x);
}
// sum over the 32 lanes of a half-wave, result in every lane: the two 16-lane row totals meet through one v_permlane16_swap
// (four v_readlane + their wait states before)
__device__ __forceinline__ float half_wave_total(float x) {
    x = row16_sum(x);
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// in-place MFMAs and the fence in front of the first vector read of their results: see row_gemm_k384.hip
__device__ __forceinline__ void mfma16(f32x4& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_first(f32x4& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_results_ready() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

struct HalfArgs {
    const float* y;       // [B,N,N,128]
    const float* q;       // [B,N,128]
    const float* k;
    const float* v;
    const f16x8* we;      // dg_row_gemm_pack(e.weight, mode 0)
    const float* be;
    const f16x8* woe;     // dg_row_gemm_pack(out_e.weight, mode 0)
    const float* boe;
    const float* gamma;
    const float* beta;
    float* e;             // [B,N,N,128] or null (no backward will follow)
    float* s;             // scores, [B,N,N,128] or null
    float* o;             // [B,N,128]
    float* y2;            // [B,N,N,128]
    float* pre;           // pre-LayerNorm sum or null
    float* mean;          // [B N N]
    float* rstd;
    int B, N;
    float alpha, eps;
    int reverse;
    int cs, spl;          // row groups per chunk, chunks per molecule (a workgroup keeps k, v of a chunk's molecule)
};

template <int H>      // stages per row group: 1 (N <= 48) or 2 (48 < N <= 96)
__global__ __launch_bounds__(64 * (kCons + kProd)) void attn_half_f32_fwd_kernel(const HalfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const tab = reinterpret_cast<float*>(smem + kOffTab);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = a.N;
    // Work units are chunks of a.cs consecutive row groups (b, i0 .. i0 + cs) of ONE molecule -- round-robin over the workgroups,
    // ascending or (a.reverse, traversal.h) descending -- so that k_j, v_j of the molecule are fetched once per chunk, not once per
    // row group (they were 56 KB of L2 traffic per 115 KB of HBM traffic).  Stages past the molecule's last row group (the
    // last chunk of a molecule may be short) and past the last chunk run empty.
    const int64_t total = static_cast<int64_t>(a.B) * a.spl;      // chunks
    const int bidx = blockIdx.x, nblk = gridDim.x;
    const int T = static_cast<int>((total - bidx + nblk - 1) / nblk) * a.cs * H;      // stages (>= cs H)
    const int TP = (T + 2) / 3 * 3;
    struct Where {
        int64_t g;      // row group b N + i (clamped to a valid one)
        int b;
        bool live, first;      // first: the stage opens a chunk
        int j0;         // first neighbour of the stage (0, or 48 in the second stage of a row group)
        bool last;      // the stage completes its row group
    };
    // stage cursor: steps through the workgroup's stages with adds and compares only (one 32-bit division per chunk: three
    // 64-bit divisions per stage cost the consumer waves ~300 scalar instructions of the ~1 100 they issued per stage)
    struct Cursor {
        int t, ii, jh, j, b, i0;
        int T, cs, spl, bidx, nblk, N, reverse, total;
        __device__ __forceinline__ void chunk() {
            int cid = bidx + j * nblk;
            if (cid > total - 1) cid = total - 1;      // (stages past the last chunk run empty on a valid one)
            if (reverse) cid = total - 1 - cid;
            b = cid / spl;
            i0 = (cid - b * spl) * cs;
        }
        __device__ __forceinline__ void start() {
            t = ii = jh = j = 0;
            chunk();
        }
        __device__ __forceinline__ void advance() {
            ++t;
            if (H == 2) {
                jh ^= 1;
                if (jh) return;
            }
            if (++ii == cs) {
                ii = 0;
                ++j;
                chunk();
            }
        }
        __device__ __forceinline__ Where here() const {
            int i = i0 + ii;
            const bool live = t < T && i < N;
            if (i > N - 1) i = N - 1;
            return Where{static_cast<int64_t>(b) * N + i, b, live, ii == 0 && jh == 0, H == 2 ? kNP * jh : 0, H == 1 || jh == 1};
        }
    };
    Cursor cur{0, 0, 0, 0, 0, 0, T, a.cs, a.spl, bidx, nblk, N, a.reverse, static_cast<int>(total)};
    cur.start();

    if (w >= kCons) {
        // ------------------------------------------------------------------------------------------ producers
        __builtin_amdgcn_s_setprio(3);
        const int pt = threadIdx.x - 64 * kCons;
        const int hw = pt >> 5, l32 = pt & 31;
        {
            const float* cse = reinterpret_cast<const float*>(a.we + 4 * 8 * 2 * 64);
            const float* cso = reinterpret_cast<const float*>(a.woe + 4 * 8 * 2 * 64);
            if (pt < 128) {
                tab[pt] = cse[pt];
                tab[128 + pt] = a.be ? a.be[pt] : 0.f;
                tab[256 + pt] = cso[pt];
                tab[384 + pt] = a.boe ? a.boe[pt] : 0.f;
                tab[512 + pt] = a.gamma[pt];
                tab[640 + pt] = a.beta[pt];
            }
        }
        // rows hw + 8 i (i < 6) of the stage, float4 column l32 (channels / k = 4 l32 .. + 3)
        const unsigned voff = static_cast<unsigned>(hw) * 512u + static_cast<unsigned>(l32) * 16u;
        // rows of a stage that exist: N (one stage per row group), or 48 / N - 48 (two)
        auto stage_rows = [&](const Where& wh) { return H == 1 ? N : (wh.j0 ? N - kNP : kNP); };
        float4 ys[3][6];
        auto fetch = [&](float4 (&set)[6], const Where& wh) {
            const int64_t g = wh.g * N + wh.j0;      // first edge row of the stage
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.y) + g * 128, 0, stage_rows(wh) * 512,
                                                                                  0x00020000);      // rows >= N read as zeros
#pragma unroll
            for (int i = 0; i < 6; ++i)
                set[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, i * 4096, 0));
        };
        // block b = c >> 1 of float4 column c = l32 (k-step c >> 3, quarter (c >> 1) & 3), half c & 1; row r of a block sits at r ^ (b & 7)
        const int blk = l32 >> 1;
        const unsigned wbase = static_cast<unsigned>(blk * (kNP * 16) + (l32 & 1) * 8);
        // fp32 tiles: row hw + 8 i sits at tile + i * 4096 + trow (row & 7 = hw for all six rows); planes: row hw + 8 i of block
        // blk at plane + i * 128 + prow.  ONE lane term each, made opaque per use: hipcc otherwise keeps every (buffer, row)
        // address of a stage in registers across the whole loop
        const unsigned trow0 = static_cast<unsigned>(hw * 512 + ((l32 ^ hw) * 16));
        const unsigned prow0 = wbase + static_cast<unsigned>((hw ^ (blk & 7)) * 16);
        auto split = [&](const float4 (&set)[6], char* pl, bool fence) {      // six rows -> hi / lo planes + inverse row scales
            if ((AH_DBG & 32) && fence) return;
            unsigned prow = prow0;
            asm volatile("" : "+v"(prow));
            char* pw_ = pl + prow;
            unsigned m[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float4& v = set[i];
                float t0, u;
                if (fence) {      // (volatile: the first use of prefetched loads stays behind the previous barrier)
                    asm volatile("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(t0) : "v"(v.x), "v"(v.y), "v"(v.z));
                } else {
                    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(t0) : "v"(v.x), "v"(v.y), "v"(v.z));
                }
                asm("v_max_f32_e64 %0, |%1|, %2" : "=v"(u) : "v"(v.w), "v"(t0));
                m[i] = __float_as_uint(u);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = umax_dpp<0xB1>(m[i]);
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = umax_dpp<0x4E>(m[i]);
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = umax_dpp<0x141>(m[i]);
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = umax_dpp<0x140>(m[i]);
            float sc[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const auto r = __builtin_amdgcn_permlane16_swap(m[i], m[i], false, false);
                const unsigned xm = r[0] > r[1] ? r[0] : r[1];
                unsigned e = xm >> 23;
                e = e < 15u ? 15u : e;
                m[i] = e;
                sc[i] = __uint_as_float((268u - e) << 23);      // 2^(14 - (e - 127)): the row maximum lands in [2^14, 2^15)
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float4& v = set[i];
                f32x2 xa = f32x2{v.x, v.y} * sc[i], xb = f32x2{v.z, v.w} * sc[i];
                const f16x2 ha = __builtin_convertvector(xa, f16x2), hb = __builtin_convertvector(xb, f16x2);
                xa -= __builtin_convertvector(ha, f32x2);
                xb -= __builtin_convertvector(hb, f32x2);
                const f16x2 la = __builtin_convertvector(xa, f16x2), lb = __builtin_convertvector(xb, f16x2);
                *reinterpret_cast<u32x2*>(pw_ + i * 128) = u32x2{__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
                *reinterpret_cast<u32x2*>(pw_ + kPlane + i * 128) = u32x2{__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
            }
            if (l32 == 0) {
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    *reinterpret_cast<float*>(pl + 2 * kPlane + (hw + 8 * i) * 4) = __uint_as_float((m[i] - 14u) << 23);
            }
        };
        // score tile of stage t -> planes (A operand of out_e): the only work between the two barriers the consumers wait at
        float4 sv[6];
        auto scores_split = [&]() {
            if (AH_DBG & 4) return;
            unsigned trow = trow0;
            asm volatile("" : "+v"(trow));
#pragma unroll
            for (int i = 0; i < 6; ++i) sv[i] = *reinterpret_cast<const float4*>(smem + kOffTs + trow + i * 4096);
            split(sv, smem + kOffPs, false);
        };
        // ... and, while the consumers run out_e, the scores (still in registers), the e tile and the node output o_i -> HBM
        auto scores_store = [&](const Where& wh) {
            if (AH_DBG & 4) return;
            const bool ok = wh.live;
            const int64_t g = wh.g * N + wh.j0;
            const int bytes = ok ? stage_rows(wh) * 512 : 0;      // 0: every store is dropped
            const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(a.s ? a.s + g * 128 : a.y2, 0, a.s ? bytes : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t re_ = __builtin_amdgcn_make_buffer_rsrc(a.e ? a.e + g * 128 : a.y2, 0, a.e ? bytes : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t ro_ = __builtin_amdgcn_make_buffer_rsrc(a.o + wh.g * 128, 0, ok && wh.last ? 512 : 0, 0x00020000);
            unsigned trow = trow0;
            asm volatile("" : "+v"(trow));
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float4 ev = *reinterpret_cast<const float4*>(smem + kOffTe + trow + i * 4096);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sv[i]), rs_, voff, i * 4096, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ev), re_, voff, i * 4096, 0);
            }
            const float4 ov = *reinterpret_cast<const float4*>(smem + kOffO + (pt & 31) * 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ov), ro_, pt < 32 ? static_cast<unsigned>(pt) * 16u : 0x7FFFFFF0u, 0, 0);
        };
        // out_e tile of stage t + residual (the y rows still in registers) -> LayerNorm -> HBM
        auto finish = [&](const float4 (&res)[6], const Where& wh) {
            if (AH_DBG & 8) return;
            const bool ok = wh.live;
            const int64_t g = wh.g * N + wh.j0;
            const int nrows = stage_rows(wh);
            const int bytes = ok ? nrows * 512 : 0;
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y2 + g * 128, 0, bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(a.pre ? a.pre + g * 128 : a.y2, 0, a.pre ? bytes : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(a.mean + g, 0, ok ? nrows * 4 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(a.rstd + g, 0, ok ? nrows * 4 : 0, 0x00020000);
            const float4 gam = ld4(tab + 512 + 4 * l32), bet = ld4(tab + 640 + 4 * l32);
            unsigned trow = trow0;
            asm volatile("" : "+v"(trow));
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int row = hw + 8 * i;
                float4 v = *reinterpret_cast<const float4*>(smem + kOffTo + trow + i * 4096);
                v += res[i];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rp, voff, i * 4096, 0);
                const float mu = half_wave_total((v.x + v.y) + (v.z + v.w)) * (1.0f / 128.0f);
                const float4 d = v - f4(mu);
                const float var = half_wave_total((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * (1.0f / 128.0f);
                const float rsd = rsqrtf(var + a.eps);
                v = fma4(rsd * d, gam, bet);
                const unsigned soff = l32 == 0 ? static_cast<unsigned>(row) * 4u : 0x7FFFFFF0u;      // one lane per row
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mu), rm, soff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rsd), rr, soff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, voff, i * 4096, 0);
            }
        };
        // stages t - 1 (finish), t (stores), t + 1, t + 2 (fetch) of the iteration
        Where wq[4];
        wq[0] = Where{0, 0, false, false, 0, true};
        wq[1] = cur.here();
        cur.advance();
        wq[2] = cur.here();
        cur.advance();
        wq[3] = cur.here();
        auto next_stage = [&]() {
            wq[0] = wq[1];
            wq[1] = wq[2];
            wq[2] = wq[3];
            cur.advance();
            wq[3] = cur.here();
        };
        fetch(ys[0], wq[1]);
        fetch(ys[1], wq[2]);
        split(ys[0], smem + kOffPy, true);
        __syncthreads();
        // stage t: | consumers: e projection + scores (t)      producers: finish (t - 1), fetch (t + 2)
        //          | producers: score tile (t) -> planes         consumers wait
        //          | consumers: out_e (t)                        producers: scores, e, o_i (t) -> HBM; y planes of stage t + 1
        for (int t = 0; t < TP; t += 3) {
            finish(ys[2], wq[0]);
            fetch(ys[2], wq[3]);
            __syncthreads();
            scores_split();
            __syncthreads();
            scores_store(wq[1]);
            split(ys[1], smem + kOffPy, true);
            next_stage();
            __syncthreads();

            finish(ys[0], wq[0]);
            fetch(ys[0], wq[3]);
            __syncthreads();
            scores_split();
            __syncthreads();
            scores_store(wq[1]);
            split(ys[2], smem + kOffPy, true);
            next_stage();
            __syncthreads();

            finish(ys[1], wq[0]);
            fetch(ys[1], wq[3]);
            __syncthreads();
            scores_split();
            __syncthreads();
            scores_store(wq[1]);
            split(ys[0], smem + kOffPy, true);
            next_stage();
            __syncthreads();
        }
        finish(ys[2], wq[0]);
        return;
    }

    // ---------------------------------------------------------------------------------------------- consumers
    const int n = lane & 15, kq = lane >> 4;
    // weight fragments (packed: 32-column slabs x 16-deep k-steps, lane = (column, k half)): channel 16 w + n,
    // k = 32 ks + 8 kq .. + 7  ->  slab (16 w + n) >> 5, k-step 2 ks + (kq >> 1), lane (kq & 1) * 32 + column
    f16x8 wfe[4][2], wfo[4][2];
    {
        const int ch = 16 * w + n;
        const int tslab = ch >> 5, col = ch & 31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const size_t idx = (static_cast<size_t>(tslab * 8 + 2 * ks + (kq >> 1)) * 2 + p) * 64 + (kq & 1) * 32 + col;
                wfe[ks][p] = a.we[idx];
                wfo[ks][p] = a.woe[idx];
            }
    }
    const int c0 = 16 * w + 4 * kq;      // this lane's four channels
    // fragment of lane (row n of block rb, quarter kq) in k-step ks: block 4 ks + kq, position (16 rb + n) ^ ((4 ks + kq) & 7)
    const unsigned xo_e = static_cast<unsigned>(kq * (kNP * 16) + ((n ^ kq) * 16));            // even k-steps
    const unsigned xo_o = static_cast<unsigned>(kq * (kNP * 16) + ((n ^ (4 + kq)) * 16));      // odd k-steps
    auto contract = [&](const char* pl, const f16x8 (&wf)[4][2], f32x4 (&acc)[3]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 xh[3], xl[3];
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const char* p0 = pl + ks * (4 * kNP * 16) + rb * 256 + ((ks & 1) ? xo_o : xo_e);
                xh[rb] = *reinterpret_cast<const f16x8*>(p0);
                xl[rb] = *reinterpret_cast<const f16x8*>(p0 + kPlane);
            }
            // lo.hi, hi.lo, hi.hi; consecutive MFMAs hit different accumulators
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) {
                    if (AH_DBG & 1) {
                        if (ks == 0 && term == 0) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
                        acc[rb][0] += static_cast<float>(xh[rb][0]) * static_cast<float>(wf[ks][0][0]);
                    } else if (ks == 0 && term == 0) mfma16_first(acc[rb], wf[ks][1], xh[rb]);
                    else mfma16(acc[rb], wf[ks][term == 0 ? 1 : 0], term == 1 ? xl[rb] : xh[rb]);
                }
        }
        mfma_results_ready();
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);      // weight fragments are in registers
    __syncthreads();                         // stage 0 is in the y planes, the tables are written
    const float4 cse = ld4(tab + c0), bev = ld4(tab + 128 + c0), cso = ld4(tab + 256 + c0), bov = ld4(tab + 384 + c0);
    // q_i of the stage, and -- once per chunk -- k_j / v_j of the lane's three rows (clamped: rows >= N are masked below),
    // requested one stage ahead: as soon as stage t's softmax has used its set
    float4 qa = f4(0.f), kk[3], vv[3];
    auto request_qkv = [&](const Where& wh) {
        if (AH_DBG & 16) return;
        qa = ld4(a.q + wh.g * 128 + c0);
        if (wh.first || H == 2) {      // (two stages per row group: the halves' k_j / v_j alternate, from L2)
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const int j = wh.j0 + 16 * rb + n;
                const int64_t jr = (wh.b * N + (j < N ? j : 0)) * 128 + c0;
                kk[rb] = ld4(a.k + jr);
                vv[rb] = ld4(a.v + jr);
            }
        }
    };
    Where wc = cur.here();
    request_qkv(wc);
    for (int t = 0; t < TP; ++t) {
        const bool live = wc.live;
        const int j0 = wc.j0;
        const bool last = wc.last;
        cur.advance();
        wc = cur.here();
        if (live) {
            f32x4 acc[3];
            contract(smem + kOffPy, wfe, acc);
            float4 sc[3];
            float4 m = f4(kNegBig);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const int row = 16 * rb + n;
                const float rs = *reinterpret_cast<const float*>(smem + kOffPy + 2 * kPlane + row * 4);
                const float4 ev = make_float4(fmaf(acc[rb][0], rs * cse.x, bev.x), fmaf(acc[rb][1], rs * cse.y, bev.y),
                                              fmaf(acc[rb][2], rs * cse.z, bev.z), fmaf(acc[rb][3], rs * cse.w, bev.w));
                const bool valid = j0 + row < N;
                sc[rb] = valid ? (a.alpha * qa) * kk[rb] * fma4(ev, ev, ev) : f4(0.f);
                if (valid) m = max4(m, sc[rb]);
                const unsigned lo = static_cast<unsigned>(row * 512 + (((4 * w + kq) ^ (row & 7)) * 16));
                *reinterpret_cast<float4*>(smem + kOffTe + lo) = ev;
                *reinterpret_cast<float4*>(smem + kOffTs + lo) = sc[rb];
            }
            if (!(AH_DBG & 2)) {
            m = make_float4(row16_max(m.x), row16_max(m.y), row16_max(m.z), row16_max(m.w));
            float4 l = f4(0.f), av = f4(0.f);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const float4 pe = (j0 + 16 * rb + n < N) ? exp4(sc[rb] - m) : f4(0.f);
                l += pe;
                av = fma4(pe, vv[rb], av);
            }
            l = make_float4(row16_sum(l.x), row16_sum(l.y), row16_sum(l.z), row16_sum(l.w));
            av = make_float4(row16_sum(av.x), row16_sum(av.y), row16_sum(av.z), row16_sum(av.w));
            if (H == 2) {      // online softmax over the two halves of the row group (the stash is this wave's own: no barrier)
                float4* run = reinterpret_cast<float4*>(smem + kOffRun) + (c0 >> 2);
                if (!last) {
                    if (n == 0) {
                        run[0] = m;
                        run[32] = l;
                        run[64] = av;
                    }
                } else {
                    const float4 m0 = run[0], l0 = run[32], av0 = run[64];
                    const float4 mm = max4(m, m0);
                    const float4 f0 = exp4(m0 - mm), f1 = exp4(m - mm);
                    l = fma4(l0, f0, l * f1);
                    av = fma4(av0, f0, av * f1);
                }
            }
            if (n == 0 && last)      // (v_rcp_f32: 1 ulp; an IEEE division is a 10-instruction sequence per component)
                *reinterpret_cast<float4*>(smem + kOffO + c0 * 4) =
                    make_float4(av.x * __builtin_amdgcn_rcpf(l.x), av.y * __builtin_amdgcn_rcpf(l.y), av.z * __builtin_amdgcn_rcpf(l.z),
                                av.w * __builtin_amdgcn_rcpf(l.w));
            }
        }
        request_qkv(wc);      // the NEXT stage's (also after an empty stage: the next one may open a chunk)
        __syncthreads();      // e / score tiles, o_i written
        __syncthreads();      // score planes written
        if (live) {
            f32x4 acc[3];
            contract(smem + kOffPs, wfo, acc);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const int row = 16 * rb + n;
                const float rs = *reinterpret_cast<const float*>(smem + kOffPs + 2 * kPlane + row * 4);
                const float4 ov = make_float4(fmaf(acc[rb][0], rs * cso.x, bov.x), fmaf(acc[rb][1], rs * cso.y, bov.y),
                                              fmaf(acc[rb][2], rs * cso.z, bov.z), fmaf(acc[rb][3], rs * cso.w, bov.w));
                *reinterpret_cast<float4*>(smem + kOffTo + row * 512 + (((4 * w + kq) ^ (row & 7)) * 16)) = ov;
            }
        }
        __syncthreads();      // out_e tile written, y planes of the next stage written
    }
}

}  // namespace
}  // namespace dg

using namespace dg;

/* Attention half of an edge branch, forward, float32: see include/druggen_hip.h. */
extern "C" int dg_attn_half_f32_fwd(const float* y, const float* q, const float* k, const float* v, const void* we_packed,
                                    const float* be, const void* woe_packed, const float* boe, const float* gamma,
                                    const float* beta, float* e, float* s, float* o, float* y2, float* pre_ln, float* mean,
                                    float* rstd, int B, int N, int C, float alpha, float eps, dg_stream_t stream_) {
    if (!y || !q || !k || !v || !we_packed || !woe_packed || !gamma || !beta || !o || !y2 || !mean || !rstd)
        return fail(DG_E_ARG, "dg_attn_half_f32_fwd: null pointer");
    if (B < 0 || C != 128 || N < 1 || N > 2 * kNP)
        return fail(DG_E_SHAPE, "dg_attn_half_f32_fwd: unsupported shape B=%d N=%d C=%d (C = 128, N <= 96)", B, N, C);
    if (B == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // chunks of row groups: enough of them to fill the chip, as long as possible (k, v are fetched once per chunk)
    const int want = (256 + B - 1) / B;                      // chunks per molecule wanted
    const int cs = (N + want - 1) / want;                    // row groups per chunk (>= 1)
    const int spl = (N + cs - 1) / cs;                       // chunks per molecule
    const int64_t chunks = static_cast<int64_t>(B) * spl;
    const int64_t rows = static_cast<int64_t>(B) * N * N;
    HalfArgs a{y, q, k, v, static_cast<const f16x8*>(we_packed), be, static_cast<const f16x8*>(woe_packed), boe, gamma, beta,
               e, s, o, y2, pre_ln, mean, rstd, B, N, alpha, eps, take_direction(rows), cs, spl};
    const int blocks = static_cast<int>(chunks < 256 ? chunks : 256);
    ProfScope prof(DG_K_ATTN_HALF_FWD, stream);
    if (N <= kNP) {
        DG_OPT_IN_LDS((&attn_half_f32_fwd_kernel<1>), kLds);
        hipLaunchKernelGGL(attn_half_f32_fwd_kernel<1>, dim3(blocks), dim3(64 * (kCons + kProd)), kLds, stream, a);
    } else {
        DG_OPT_IN_LDS((&attn_half_f32_fwd_kernel<2>), kLds);
        hipLaunchKernelGGL(attn_half_f32_fwd_kernel<2>, dim3(blocks), dim3(64 * (kCons + kProd)), kLds, stream, a);
    }
    return check_launch("dg_attn_half_f32_fwd");
}
