// Backward of the edge branch's attention half, FIRST PART, float32 rows, as ONE kernel: ln4 backward + ds = dz4 Woe (out_e
// input gradient) + attention-core backward (reference src/model/layers.py:119-135, 186-188, differentiated).  Given
// dy2 = d loss / d y2 and d_o = d loss / d o:
//   dz4 = LayerNormBackward(dy2; pre4, mean4, rstd4, gamma4)           (+ dgamma4 / dbeta4 partial sums)
//   ds  = dz4 Woe                                                      gradient of the scores through out_e
//   p = softmax_j(sc), sc = alpha q_i k_j (e + 1) e ;  dsc = p (d_o v_j - sum_j p d_o v_j) + ds
//   dv_j += p d_o ;  dq_i = alpha sum_j dsc k_j (e^2 + e) ;  dk_j += dsc alpha q_i (e^2 + e) ;  de = dsc alpha q_i k_j (2 e + 1)
// Unfused these are the LayerNorm-backward row GEMM (dz4 AND ds to HBM) and dg_attn_core_bwd (ds, e from HBM).  Here ds never
// leaves the accumulators: with the swapped MFMA product a lane holds ds for rows n, 16 + n, 32 + n x 4 channels -- exactly
// the layout the softmax backward wants (the rows of a channel along a 16-lane DPP row).  A workgroup walks whole molecules:
// dk_j, dv_j are accumulated over i in registers and written once per molecule, in a fixed order (bit-reproducible).
//   waves 8..11  producers: dy2 / pre4 rows HBM -> registers (a stage ahead) -> LayerNorm backward (half-wave per row) ->
//                dz4 to HBM and to fp16 hi / lo planes; e rows HBM -> registers -> LDS tile; the de tile and dq_i of the
//                previous stage LDS -> HBM.
//   waves 0..7   consumers: 36 MFMAs (ds = dz4 Woe), then the attention backward on the accumulator layout; de to an LDS tile,
//                dq_i to LDS, dk / dv in registers.
//   ONE barrier per row group: planes, e tile and de tile are double-buffered (155 KB of LDS).
// The second part -- dy = de We + dz4 (+ the previous block's LayerNorm backward) -- stays dg_row_gemm / dg_row_gemm_ln_bwd.
#include "common.h"
#include "traversal.h"

namespace dg {
bool reduce_batch_try_add(const float* part, int S, long long n_floats, float* out);      // linear_wgrad.hip
void launch_ln_finish(const float* part, int nblocks, int K, int C, float* out0, float* out1, hipStream_t stream);
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
constexpr int kOffPa = 0;                                 // planes of dz4 [2]
constexpr int kOffTe = 2 * kPlanes;                       // e tile [2]
constexpr int kOffTd = kOffTe + 2 * kTile;                // de tile [2]
constexpr int kOffTab = kOffTd + 2 * kTile;               // inverse column scales of the packed weight [128]
constexpr int kOffQ = kOffTab + 512;                      // dq_i [2][128]
constexpr int kOffRed = kOffQ + 2 * 512;                  // dgamma / dbeta of the eight half-waves [8][2][128]
constexpr int kLds = kOffRed + 8 * 2 * 512;
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
__device__ __forceinline__ float4 row16_sum4(float4 v) {
    return make_float4(row16_sum(v.x), row16_sum(v.y), row16_sum(v.z), row16_sum(v.w));
}
// sum over the 32 lanes of a half-wave, result in every lane
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

struct HalfBwdArgs {
    const float* dy2;     // [B,N,N,128]
    const float* pre;     // pre-LayerNorm sum of ln4
    const float* mean;
    const float* rstd;
    const float* gamma;
    const f16x8* woe;     // dg_row_gemm_pack(out_e.weight, mode 1)
    const float* e;       // [B,N,N,128]
    const float* q;       // [B,N,128]
    const float* k;
    const float* v;
    const float* d_o;     // [B,N,128]
    float* dz;            // [B,N,N,128]: LayerNorm input gradient
    float* de;            // [B,N,N,128]
    float* ds;            // [B,N,N,128]: dz4 Woe, written only by the DS instance (a second order will differentiate this pass)
    float* dq;            // [B,N,128]
    float* dk;
    float* dv;
    float* part;          // [workgroups][2][128]: partial sums of dgamma4 / dbeta4
    int B, N;
    float alpha;
};

struct Where {
    int64_t g;      // row group b N + i (always a valid one)
    int b, i;
    bool live;
};
// molecules round-robin over the workgroups; a stage is one row group (b, i), i ascending
struct Cursor {
    int t, i, m, T, N, bidx, nblk, nmol;
    __device__ __forceinline__ void advance() {
        ++t;
        if (++i == N) {
            i = 0;
            ++m;
        }
    }
    __device__ __forceinline__ Where here() const {
        const int mm = m < nmol ? m : nmol - 1;      // (stages past the last molecule run empty on a valid one)
        const int b = bidx + mm * nblk;
        return Where{static_cast<int64_t>(b) * N + i, b, i, t < T};
    }
};

template <bool DS>
__global__ __launch_bounds__(64 * (kCons + kProd)) void attn_half_f32_bwd1_kernel(const HalfBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const tab = reinterpret_cast<float*>(smem + kOffTab);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = a.N;
    const int bidx = blockIdx.x, nblk = gridDim.x;
    const int nmol = (a.B - bidx + nblk - 1) / nblk;      // >= 1
    const int T = nmol * N;
    const int TP = (T + 1) / 2 * 2;
    Cursor cur{0, 0, 0, T, N, bidx, nblk, nmol};
    const int rowbytes = N * 512;

    if (w >= kCons) {
        // ------------------------------------------------------------------------------------------ producers
        __builtin_amdgcn_s_setprio(3);
        const int pt = threadIdx.x - 64 * kCons;
        const int hw = pt >> 5, l32 = pt & 31;
        if (pt < 128) tab[pt] = reinterpret_cast<const float*>(a.woe + 4 * 8 * 2 * 64)[pt];
        // rows hw + 8 i (i < 6) of the stage, float4 column l32
        const unsigned voff = static_cast<unsigned>(hw) * 512u + static_cast<unsigned>(l32) * 16u;
        const int blk = l32 >> 1;
        const unsigned wbase = static_cast<unsigned>(blk * (kNP * 16) + (l32 & 1) * 8);
        const float4 gam = ld4(a.gamma + 4 * l32);
        float4 dgam = f4(0.f), dbet = f4(0.f);
        // fp32 tiles: row hw + 8 i of a tile sits at tile + i * 4096 + trow (row & 7 = hw for all six rows); planes: row
        // hw + 8 i of block blk at plane + i * 128 + prow.  ONE lane term each, made opaque per use: hipcc otherwise keeps the
        // ~18 (buffer, row) addresses of a stage in registers across the whole loop (29 spilled registers)
        const unsigned trow0 = static_cast<unsigned>(hw * 512 + ((l32 ^ hw) * 16));
        const unsigned prow0 = wbase + static_cast<unsigned>((hw ^ (blk & 7)) * 16);
        float4 es[6];               // e rows of the same stage: through registers into the LDS tile (an LDS-DMA would have to be
                                    // awaited with vmcnt before the barrier -- together with every store issued since)
        float4 dys[6], prs[6];      // ONE set: the rows of stage t + 2 are requested right after the LayerNorm backward of stage
        float stat;                 // t + 1 has used the registers (a stage ahead of their use); stat: lane i < 6 of a half-wave
                                    // holds the mean of row hw + 8 i, lane 8 + i its rstd
        auto prefetch = [&](float4 (&dyr)[6], float4 (&prr)[6], float& st, const Where& wh) {
            const __amdgpu_buffer_rsrc_t ree = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.e) + wh.g * N * 128, 0, rowbytes, 0x00020000);
#pragma unroll
            for (int i = 0; i < 6; ++i) es[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ree, voff, i * 4096, 0));
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy2) + wh.g * N * 128, 0, rowbytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.pre) + wh.g * N * 128, 0, rowbytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mean) + wh.g * N, 0, N * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.rstd) + wh.g * N, 0, N * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < 6; ++i) {      // (rows >= N read as zeros)
                dyr[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rd, voff, i * 4096, 0));
                prr[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rp, voff, i * 4096, 0));
            }
            const int ri = l32 & 7;      // row hw + 8 ri (ri < 6): lanes 0..5 fetch its mean, lanes 8..13 its rstd
            const unsigned so = ri < 6 ? static_cast<unsigned>(hw + 8 * ri) * 4u : 0x7FFFFFF0u;
            const unsigned mv = __builtin_amdgcn_raw_buffer_load_b32(rm, (l32 & 24) == 0 ? so : 0x7FFFFFF0u, 0, 0);
            const unsigned rv = __builtin_amdgcn_raw_buffer_load_b32(rr, (l32 & 24) == 8 ? so : 0x7FFFFFF0u, 0, 0);
            st = __uint_as_float(mv | rv);      // (the lane that does not take part read 0)
        };
        auto stage_e = [&](int buf) {      // the prefetched e rows -> LDS tile [buf]
            unsigned trow = trow0;
            asm volatile("" : "+v"(trow));
            char* tb = smem + kOffTe + buf * kTile + trow;
#pragma unroll
            for (int i = 0; i < 6; ++i) *reinterpret_cast<float4*>(tb + i * 4096) = es[i];
            __builtin_amdgcn_sched_barrier(0);
        };
        auto store_prev = [&](const Where& wh, int buf) {      // de tile and dq_i of a finished stage -> HBM
            const __amdgpu_buffer_rsrc_t re_ = __builtin_amdgcn_make_buffer_rsrc(a.de + wh.g * N * 128, 0, wh.live ? rowbytes : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rq_ = __builtin_amdgcn_make_buffer_rsrc(a.dq + wh.g * 128, 0, wh.live ? 512 : 0, 0x00020000);
            unsigned trow = trow0;
            asm volatile("" : "+v"(trow));
            const char* tb = smem + kOffTd + buf * kTile + trow;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const u32x4 dv_ = *reinterpret_cast<const u32x4*>(tb + i * 4096);
                __builtin_amdgcn_raw_buffer_store_b128(dv_, re_, voff, i * 4096, 0);
            }
            const u32x4 qv = *reinterpret_cast<const u32x4*>(smem + kOffQ + buf * 512 + (pt & 31) * 16);
            __builtin_amdgcn_raw_buffer_store_b128(qv, rq_, pt < 32 ? static_cast<unsigned>(pt) * 16u : 0x7FFFFFF0u, 0, 0);
            __builtin_amdgcn_sched_barrier(0);      // (the producer's steps one after the other: their temporaries do not add up)
        };
        // LayerNorm backward of the six rows: dz4 -> HBM and -> planes[buf] (fp16 hi / lo under a power-of-two row scale)
        auto ln_bwd_split = [&](const float4 (&dyr)[6], float4 (&prr)[6], float st, const Where& wh, int buf) {
            const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(a.dz + wh.g * N * 128, 0, wh.live ? rowbytes : 0, 0x00020000);
            char* pl = smem + kOffPa + buf * kPlanes;
            unsigned prow = prow0;
            asm volatile("" : "+v"(prow));
            char* pw_ = pl + prow;
            const float livef = wh.live ? 1.f : 0.f;
            float4 (&dzr)[6] = prr;      // dz4 takes the place of the pre-LayerNorm rows, row by row
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int src = (lane & 32) + i;      // this half-wave's lane i holds the row's mean, lane 8 + i its rstd
                const float mu = __int_as_float(__builtin_amdgcn_ds_bpermute(src * 4, __float_as_int(st)));
                const float rsd = __int_as_float(__builtin_amdgcn_ds_bpermute((src + 8) * 4, __float_as_int(st)));
                const float4 xh = rsd * (prr[i] - f4(mu));
                const float4 g = dyr[i];
                const float4 u = g * gam;
                const float c1 = half_wave_total((u.x + u.y) + (u.z + u.w)) * (1.0f / 128.0f);
                const float c2 = half_wave_total((u.x * xh.x + u.y * xh.y) + (u.z * xh.z + u.w * xh.w)) * (1.0f / 128.0f);
                dgam = fma4(livef * g, xh, dgam);      // (g = 0 in the padding rows)
                dbet += livef * g;
                dzr[i] = (hw + 8 * i < N) ? rsd * (u - f4(c1) - c2 * xh) : f4(0.f);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dzr[i]), rz, voff, i * 4096, 0);
                __builtin_amdgcn_sched_barrier(0);      // one row at a time: bounds the live temporaries
            }
            unsigned m[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float4& v = dzr[i];
                float t0, u;
                asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(t0) : "v"(v.x), "v"(v.y), "v"(v.z));
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
                const float4& v = dzr[i];
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
            __builtin_amdgcn_sched_barrier(0);      // (the next stage's loads are not hoisted above this: they reuse the registers)
        };
        // stages t - 1 (stores), t + 1 (LayerNorm backward, e tile), t + 2 (prefetch) of iteration t
        Where wprev{0, 0, 0, false};
        Where w0 = cur.here();
        cur.advance();
        Where w1 = cur.here();
        cur.advance();
        Where w2 = cur.here();
        prefetch(dys, prs, stat, w0);
        stage_e(0);
        ln_bwd_split(dys, prs, stat, w0, 0);
        prefetch(dys, prs, stat, w1);
        __syncthreads();
        for (int t = 0; t < TP; t += 2) {
            // iteration t: the consumers are on stage t (planes, e tile, de tile [t & 1])
            stage_e(1);                                   // stage t + 1
            store_prev(wprev, 1);                         // stage t - 1
            ln_bwd_split(dys, prs, stat, w1, 1);          // stage t + 1
            prefetch(dys, prs, stat, w2);                 // stage t + 2
            wprev = w0;
            w0 = w1;
            w1 = w2;
            cur.advance();
            w2 = cur.here();
            __syncthreads();
            stage_e(0);                                   // stage t + 2
            store_prev(wprev, 0);                         // stage t
            ln_bwd_split(dys, prs, stat, w1, 0);          // stage t + 2
            prefetch(dys, prs, stat, w2);                 // stage t + 3
            wprev = w0;
            w0 = w1;
            w1 = w2;
            cur.advance();
            w2 = cur.here();
            __syncthreads();
        }
        store_prev(wprev, 1);      // stage TP - 1
        // dgamma4 / dbeta4 of this workgroup: the eight half-waves summed in a fixed order
        float* red = reinterpret_cast<float*>(smem + kOffRed);
        *reinterpret_cast<float4*>(red + (hw * 2 + 0) * 128 + 4 * l32) = dgam;
        *reinterpret_cast<float4*>(red + (hw * 2 + 1) * 128 + 4 * l32) = dbet;
        __syncthreads();
        if (a.part) {      // thread pt: value pt of [2][128]
            float s = 0.f;
#pragma unroll
            for (int h = 0; h < 8; ++h) s += red[h * 256 + pt];
            a.part[static_cast<size_t>(bidx) * 256 + pt] = s;
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------- consumers
    const int n = lane & 15, kq = lane >> 4;
    f16x8 wf[4][2];
    {
        const int ch = 16 * w + n;
        const int tslab = ch >> 5, col = ch & 31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                wf[ks][p] = a.woe[(static_cast<size_t>(tslab * 8 + 2 * ks + (kq >> 1)) * 2 + p) * 64 + (kq & 1) * 32 + col];
    }
    const int c0 = 16 * w + 4 * kq;      // this lane's four channels
    // row 16 rb + n of an fp32 tile, this lane's slot: tile + rb * 8192 + tlane (row & 7 = n & 7 for the three row blocks)
    const unsigned tlane0 = static_cast<unsigned>(n * 512 + (((4 * w + kq) ^ (n & 7)) * 16));
    const unsigned xo_e = static_cast<unsigned>(kq * (kNP * 16) + ((n ^ kq) * 16));            // even k-steps
    const unsigned xo_o = static_cast<unsigned>(kq * (kNP * 16) + ((n ^ (4 + kq)) * 16));      // odd k-steps
    // q_i, d_o_i of the stage and -- at the first row group of a molecule -- k_j, v_j of the lane's three rows, one stage ahead
    float4 qa = f4(0.f), woi = f4(0.f), kk[3], vv[3];
    auto request = [&](const Where& wh) {
        qa = ld4(a.q + wh.g * 128 + c0);
        woi = ld4(a.d_o + wh.g * 128 + c0);
        if (wh.i == 0) {
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const int j = 16 * rb + n;
                const int64_t jr = (static_cast<int64_t>(wh.b) * N + (j < N ? j : 0)) * 128 + c0;
                kk[rb] = ld4(a.k + jr);      // (scaled by alpha at its first use: see `fresh`)
                vv[rb] = ld4(a.v + jr);
            }
        }
    };
    float4 dka[3], dva[3];
#pragma unroll
    for (int rb = 0; rb < 3; ++rb) dka[rb] = dva[rb] = f4(0.f);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // weight fragments are in registers
    Where wc = cur.here();
    request(wc);
    __syncthreads();                         // stage 0: planes, e tile, column scales
    for (int t = 0; t < TP; ++t) {
        const Where wh = wc;
        cur.advance();
        wc = cur.here();
        if (wh.live) {
            const char* pl = smem + kOffPa + (t & 1) * kPlanes;
            unsigned tlane = tlane0;
            asm volatile("" : "+v"(tlane));      // opaque per stage: the (buffer, row block) addresses are not kept across the loop
            const char* te = smem + kOffTe + (t & 1) * kTile + tlane;
            char* td = smem + kOffTd + (t & 1) * kTile + tlane;
            f32x4 acc[3];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                f16x8 xh[3], xl[3];
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) {
                    const char* p0 = pl + ks * (4 * kNP * 16) + rb * 256 + ((ks & 1) ? xo_o : xo_e);
                    xh[rb] = *reinterpret_cast<const f16x8*>(p0);
                    xl[rb] = *reinterpret_cast<const f16x8*>(p0 + kPlane);
                }
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int rb = 0; rb < 3; ++rb) {
                        if (ks == 0 && term == 0) mfma16_first(acc[rb], wf[ks][1], xh[rb]);
                        else mfma16(acc[rb], wf[ks][term == 0 ? 1 : 0], term == 1 ? xl[rb] : xh[rb]);
                    }
            }
            mfma_results_ready();
            if (wh.i == 0) {      // a new molecule's k_j: alpha folded in once (sc = q_i (alpha k_j) (e + 1) e)
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) kk[rb] = a.alpha * kk[rb];
            }
            // pass 1: scores, softmax statistics, sum_j p d_o v_j
            float4 pe[3];
            float4 m = f4(kNegBig);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const int row = 16 * rb + n;
                const float4 ev = *reinterpret_cast<const float4*>(te + rb * 8192);
                pe[rb] = qa * kk[rb] * fma4(ev, ev, ev);
                if (row < N) m = max4(m, pe[rb]);
            }
            m = make_float4(row16_max(m.x), row16_max(m.y), row16_max(m.z), row16_max(m.w));
            float4 l = f4(0.f), A = f4(0.f);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                pe[rb] = (16 * rb + n < N) ? exp4(pe[rb] - m) : f4(0.f);
                l += pe[rb];
                A = fma4(pe[rb], woi * vv[rb], A);
            }
            l = row16_sum4(l);
            A = row16_sum4(A);
            const float4 inv = make_float4(__builtin_amdgcn_rcpf(l.x), __builtin_amdgcn_rcpf(l.y), __builtin_amdgcn_rcpf(l.z),
                                           __builtin_amdgcn_rcpf(l.w));
            const float4 abar = A * inv;
            // pass 2: dsc, the k / v / q gradients, de
            float4 dqa = f4(0.f);
            __amdgpu_buffer_rsrc_t rds;
            if constexpr (DS) rds = __builtin_amdgcn_make_buffer_rsrc(a.ds + wh.g * N * 128, 0, rowbytes, 0x00020000);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const int row = 16 * rb + n;
                const float rs = *reinterpret_cast<const float*>(pl + 2 * kPlane + row * 4);
                const float4 cs = ld4(tab + c0);
                const float4 wss = make_float4(acc[rb][0] * (rs * cs.x), acc[rb][1] * (rs * cs.y), acc[rb][2] * (rs * cs.z),
                                               acc[rb][3] * (rs * cs.w));
                if constexpr (DS)      // (rows past N fall outside the buffer range: dropped)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, wss), rds,
                                                           static_cast<unsigned>(row * 512 + c0 * 4), 0, 0);
                const float4 ev = *reinterpret_cast<const float4*>(te + rb * 8192);
                const float4 p = pe[rb] * inv;
                float4 ds = fma4(p, woi * vv[rb] - abar, wss);
                if (row >= N) ds = f4(0.f);
                dva[rb] = fma4(p, woi, dva[rb]);
                const float4 dsg = ds * fma4(ev, ev, ev);
                dqa = fma4(dsg, kk[rb], dqa);      // (kk holds alpha k_j)
                dka[rb] = fma4(dsg, qa, dka[rb]);  // (alpha at the store)
                const float4 dev = ds * qa * kk[rb] * fma4(f4(2.f), ev, f4(1.f));
                *reinterpret_cast<float4*>(td + rb * 8192) = dev;
                __builtin_amdgcn_sched_barrier(0);      // one row block at a time: bounds the live temporaries
            }
            dqa = row16_sum4(dqa);
            if (n == 0) *reinterpret_cast<float4*>(smem + kOffQ + (t & 1) * 512 + c0 * 4) = dqa;
            if (wh.i == N - 1) {      // the molecule is complete: dk_j, dv_j of the lane's rows
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) {
                    const int j = 16 * rb + n;
                    if (j < N) {
                        const int64_t jr = (static_cast<int64_t>(wh.b) * N + j) * 128 + c0;
                        st4(a.dk + jr, a.alpha * dka[rb]);
                        st4(a.dv + jr, dva[rb]);
                    }
                    dka[rb] = dva[rb] = f4(0.f);
                }
            }
        }
        request(wc);      // the NEXT stage's operands
        __syncthreads();
    }
    __syncthreads();      // (the producers' dgamma / dbeta exchange)
}

}  // namespace
}  // namespace dg

using namespace dg;

extern "C" size_t dg_attn_half_f32_bwd1_workspace_bytes(int B) {
    const int blocks = B < 256 ? B : 256;
    return static_cast<size_t>(blocks > 0 ? blocks : 1) * 256 * sizeof(float);
}

/* First part of the float32 attention-half backward: see include/druggen_hip.h. */
extern "C" int dg_attn_half_f32_bwd1(const float* dy2, const float* pre4, const float* mean4, const float* rstd4,
                                     const float* gamma4, const void* woe_dgrad_packed, const float* e, const float* q,
                                     const float* k, const float* v, const float* d_o, float* dz4, float* ds, float* de,
                                     float* dq, float* dk, float* dv, float* dgamma4, float* dbeta4, void* workspace,
                                     size_t workspace_bytes, int B, int N, int C, float alpha, dg_stream_t stream_) {
    if (!dy2 || !pre4 || !mean4 || !rstd4 || !gamma4 || !woe_dgrad_packed || !e || !q || !k || !v || !d_o || !dz4 || !de || !dq ||
        !dk || !dv || !workspace)
        return fail(DG_E_ARG, "dg_attn_half_f32_bwd1: null pointer");
    if (B < 0 || C != 128 || N < 1 || N > kNP)
        return fail(DG_E_SHAPE, "dg_attn_half_f32_bwd1: unsupported shape B=%d N=%d C=%d (C = 128, N <= 48)", B, N, C);
    if (workspace_bytes < dg_attn_half_f32_bwd1_workspace_bytes(B))
        return fail(DG_E_WORKSPACE, "dg_attn_half_f32_bwd1: workspace too small");
    if (B == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int blocks = B < 256 ? B : 256;
    float* part = static_cast<float*>(workspace);
    HalfBwdArgs a{dy2, pre4, mean4, rstd4, gamma4, static_cast<const f16x8*>(woe_dgrad_packed), e, q, k, v, d_o,
                  dz4, de, ds, dq, dk, dv, (dgamma4 || dbeta4) ? part : nullptr, B, N, alpha};
    {
        ProfScope prof(DG_K_ATTN_HALF_BWD, stream);
        note_forward(static_cast<int64_t>(B) * N * N);      // (dgamma / dbeta partial sums, dk / dv accumulation: fixed order)
        if (ds) {
            DG_OPT_IN_LDS((&attn_half_f32_bwd1_kernel<true>), kLds);
            hipLaunchKernelGGL(attn_half_f32_bwd1_kernel<true>, dim3(blocks), dim3(64 * (kCons + kProd)), kLds, stream, a);
        } else {
            DG_OPT_IN_LDS((&attn_half_f32_bwd1_kernel<false>), kLds);
            hipLaunchKernelGGL(attn_half_f32_bwd1_kernel<false>, dim3(blocks), dim3(64 * (kCons + kProd)), kLds, stream, a);
        }
    }
    // inside dg_linear_wgrad_batch_begin / _end the reduction joins that batch's single launch (dgamma4 and dbeta4 adjacent)
    if ((dgamma4 || dbeta4) &&
        !(dgamma4 && dbeta4 == dgamma4 + 128 && reduce_batch_try_add(part, blocks, 256, dgamma4)))
        launch_ln_finish(part, blocks, 2, 128, dgamma4, dbeta4, stream);
    return check_launch("dg_attn_half_f32_bwd1");
}
