// 384 -> 128 row GEMM (fc2 of the feed-forward + residual + LayerNorm: y = LN(x + h W2^T + b2), reference
// src/model/layers.py:52-53,191-192; and dx = dz + dh W1 in the backward), fp32 rows, producer / consumer form -- the twin
// of row_gemm_n384.hip.
//
// Arithmetic as in row_gemm.hip's K = 384 kernel: fp16 hi + lo planes, one exact power-of-two scale per (row, 128-wide
// chunk) of the activation and per output column of the weight, three MFMA products; each chunk's products go to their own
// accumulator set, which is folded `acc += part * (row scale * column scale)` (row AND column inverse scale: folding with
// the row scale alone overflows for rows above ~2^90).  Same packed weight (dg_row_gemm_pack).
//
//   waves 8..11  producers: every global access.  16-row stages: the stage's 16 x 3 (row, chunk) pieces of 512 bytes are
//                streamed HBM -> registers (one half-wave per piece: both the row maximum of a chunk and, later, the
//                LayerNorm sums of a row are DPP reductions; buffer loads whose range ends at the last row; three stages =
//                72 KiB deep) -> hi / lo planes in LDS.  The finished output tile of the previous stage comes back from
//                LDS, gets its residual row, LayerNorm, and leaves as whole 512-byte rows (y, the pre-LayerNorm sum, the
//                row statistics) through range-checked buffer stores: no branch in the loop.
//   waves 0..7   consumers: wave w owns output channels [16 w, 16 w + 16) with its weight fragments resident (12 k-steps x
//                2 planes = 96 VGPRs).  Swapped product on v_mfma_f32_16x16x32_f16: a lane ends up with 4 consecutive
//                channels of ONE row.  Per stage 24 ds_read_b128, 36 MFMAs, three folds, + bias (ReLU), one 16-byte LDS
//                write.  No global memory operation.
//   one s_barrier per stage; planes and output tile double-buffered.
//
// H16 instances (DG_DTYPE_F32_H16, include/druggen_hip.h): the [R,384] operand arrives as ONE fp16 plane with one inverse
// power-of-two scale per row (what row_gemm_n384.hip's H16 instance writes).  The producers then only MOVE it: 16-byte pieces HBM
// -> registers -> LDS (row-major stage [16][768 B], the 16-byte slot index xor-ed with the row inside each 256-byte group:
// conflict-free for the writes of 16 consecutive lanes and for the fragment reads of 16 rows), 12 KiB per stage, six stages in
// flight.  The consumers run TWO products per k-step (x . w_hi, x . w_lo) on six independent chains and fold once.
#include "common.h"

#include <cstdlib>
#include "row_gemm_k384.h"
#include "pair.h"
#include "traversal.h"

#ifndef K3_DBG
#define K3_DBG 0      // ablation builds (scripts/build_variant.sh): 1 no MFMAs, 2 raw LDS writes instead of the split, 4 no tile
#endif                // finishing (residual, LayerNorm, stores), 8 no global fetch of A, 16 no fragment reads
namespace dg {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kSR = 16;                                  // rows per stage
constexpr int kPlane = 48 * 256;                         // [chunk 3][k-step 4][k-quarter 4][row 16 (xor-swizzled)][16 B]
constexpr int kStage = 2 * kPlane + 256;                 // hi, lo, inverse scales [16 rows][3 chunks] (+ pad)
constexpr int kOut = kSR * 32 * 16;                      // output tile: [row 16][16-byte slot 32 (xor row & 7)]
constexpr int kOffOut = 2 * kStage;
constexpr int kOffTab = kOffOut + 2 * kOut;              // gamma [128], beta [128]
constexpr int kLds = kOffTab + 2 * 128 * 4;
constexpr int kCons = 8, kProd = 4, kDepth = 3;

template <int CTRL>
__device__ __forceinline__ unsigned umax_dpp(unsigned x) {
    const unsigned moved = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, 0xF, 0xF, true));
    return x > moved ? x : moved;
}
template <int CTRL>
__device__ __forceinline__ float sum_dpp(float x) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true);
    return x + __int_as_float(moved);
}
// sum over the 32 lanes of a half-wave, result in every lane (DPP inside the 16-lane rows, two scalar reads across)
__device__ __forceinline__ float half_wave_total(float x, bool upper) {
    x = sum_dpp<0xB1>(x);
    x = sum_dpp<0x4E>(x);
    x = sum_dpp<0x141>(x);
    x = sum_dpp<0x140>(x);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
    return upper ? r2 + r3 : r0 + r1;
}

// acc += A . B on v_mfma_f32_16x16x32_f16, ALWAYS in place (result registers = accumulator input).  Through the builtin hipcc
// renamed the destination of some MFMAs and put them one slot behind the MFMA that produced their accumulator input; lanes
// 48..63 of that input were then still being written (a few wrong columns per launch, never the same ones: the result
// latency of this gfx950 opcode is longer than the hazard tables assume).  In-place chains are interlocked by the hardware;
// what the compiler no longer sees -- a vector read of a result -- is fenced by mfma_results_ready().
__device__ __forceinline__ void mfma16(f32x4& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// first MFMA of a chain: accumulator input = the constant 0 (no vector write of the accumulator in front of the chain)
__device__ __forceinline__ void mfma16_first(f32x4& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_results_ready() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

struct EpiK {
    const float* bias;        // [128] or null
    const float* residual;    // [R,128] or null
    const float* gamma;       // LayerNorm or null
    const float* beta;
    float* mean;
    float* rstd;
    float* pre;               // optional [R,128]: the pre-LayerNorm sum
    float eps;
    int relu;
    int reverse;              // stages in descending order (traversal.h)
};

// One problem of a launch; workgroups [0, nb0) run problem 0, the others problem 1 (a node-level GEMM riding in the
// edge-level launch of the same kernel: pair.h, row_gemm_n384.hip).
struct ProbK {
    const float* a;           // [R,384] float32, or (H16) [R,384] fp16, or (afmt 2) [R,384] 3-byte elements (DG_DTYPE_F32_H24)
    const float* ascale;      // H16 / H32: inverse row scales [R]
    const void* alo;          // H32: the lo plane [R,384] fp16
    int afmt;                 // 0 float32, 1 H16, 2 H24, 3 H32 (DG_DTYPE_F32_H32: hi + lo fp16 planes under one row scale)
    const f16x8* packed;
    float* y;
    int64_t R;
    EpiK ep;
};

// NP (fp16-plane operand only): 2 products per k-step (x . w_hi + x . w_lo); 1 (x . w_hi) is the DG_DH_PRODUCTS=1 experiment.
template <bool RES, bool LN, int FMT, int NP = 2>
__global__ __launch_bounds__(64 * (kCons + kProd)) void row_gemm_k384_kernel(const ProbK p0, const ProbK p1, const int nb0) {
    // H16: the operand arrives as fp16 plane(s) + row scales and is only MOVED by the producers; TWO: with its lo plane (H32: the
    // float32-class split done once by the writer instead of by every reader)
    constexpr bool H16 = FMT == 1 || FMT == 3, TWO = FMT == 3, H24 = FMT == 2;
    static_assert(NP == 2 || (NP == 1 && FMT == 1 && !LN), "single-product arithmetic: the backward's fp16-plane operand only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const tab = reinterpret_cast<float*>(smem + kOffTab);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool second = static_cast<int>(blockIdx.x) >= nb0;      // uniform
    const float* __restrict__ const a = second ? p1.a : p0.a;
    const float* __restrict__ const ascale = second ? p1.ascale : p0.ascale;
    const void* __restrict__ const alo = second ? p1.alo : p0.alo;
    const f16x8* __restrict__ const packed = second ? p1.packed : p0.packed;
    float* __restrict__ const y = second ? p1.y : p0.y;
    const int64_t R = second ? p1.R : p0.R;
    const EpiK ep = second ? p1.ep : p0.ep;
    const int bidx = second ? static_cast<int>(blockIdx.x) - nb0 : static_cast<int>(blockIdx.x);
    const int nblk = second ? static_cast<int>(gridDim.x) - nb0 : nb0;
    // stages round-robin over the problem's workgroups, ascending or (ep.reverse) descending: at any time the launch works on
    // one window of consecutive rows that moves through the matrix like the windows of its neighbours in the stream
    const int64_t total = (R + kSR - 1) / kSR;
    const int T = static_cast<int>((total - bidx + nblk - 1) / nblk);      // >= 1
    auto stage_of = [&](int t) {
        const int64_t st = bidx + static_cast<int64_t>(t) * nblk;
        return ep.reverse ? total - 1 - st : st;
    };
    constexpr int DEPTH = (H16 && !TWO) ? 2 * kDepth : kDepth;      // stages in flight (H16 stages are half the bytes: twice as many)
    const int TP = (T + DEPTH - 1) / DEPTH * DEPTH;

    if (w >= kCons) {
        // ------------------------------------------------------------------------------------------ producers
        __builtin_amdgcn_s_setprio(3);
        const int pt = threadIdx.x - 64 * kCons;
        const int hw = pt >> 5, l32 = pt & 31;
        const bool upper = (lane & 32) != 0;
        if (LN && pt < 128) {
            tab[pt] = ep.gamma[pt];
            tab[128 + pt] = ep.beta[pt];
        }
        constexpr int NPF = TWO ? 7 : (H16 ? 4 : 6);
        float4 pf[DEPTH][NPF];      // (H16: three pieces (+ three of the lo plane) + the stage's inverse row scales, kept in .x of the last)
        // half-wave hw streams rows 2 hw, 2 hw + 1 of the stage: six consecutive 512-byte pieces (row-major, chunk minor)
        const unsigned voff = static_cast<unsigned>(hw) * 3072u + static_cast<unsigned>(l32) * 16u;
        // H16: thread pt moves the 16-byte pieces pt, pt + 256, pt + 512 of the stage's 768 (= 16 rows x 48); piece q is slot
        // q % 48 of row q / 48 and goes to LDS slot (slot ^ (row & 15)) of that row (the xor stays inside a 16-slot group)
        unsigned hdst[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int q = pt + 256 * i, row = q / 48, slot = q % 48;
            hdst[i] = static_cast<unsigned>(row * 768 + ((slot & ~15) | ((slot & 15) ^ row)) * 16);
        }
        auto fetch = [&](float4 (&set)[NPF], int t) {
            if (t > T - 1) t = T - 1;
            if (H16) {
                const int64_t r0 = stage_of(t) * kSR;
                const int64_t left = (R - r0) * 768;
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    reinterpret_cast<_Float16*>(const_cast<float*>(a)) + r0 * 384, 0, static_cast<int>(left < (1 << 30) ? left : (1 << 30)),
                    0x00020000);
                const int64_t lefts = (R - r0) * 4;
                const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(ascale) + r0, 0, static_cast<int>(lefts < 64 ? lefts : 64), 0x00020000);
                if (K3_DBG & 8) return;
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    set[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<unsigned>(pt) * 16u, i * 4096, 0));
                if (TWO) {
                    const __amdgpu_buffer_rsrc_t rlo = __builtin_amdgcn_make_buffer_rsrc(
                        reinterpret_cast<_Float16*>(const_cast<void*>(alo)) + r0 * 384, 0, static_cast<int>(left < (1 << 30) ? left : (1 << 30)),
                        0x00020000);
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        set[3 + i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rlo, static_cast<unsigned>(pt) * 16u, i * 4096, 0));
                }
                // threads 0..15: the inverse scale of row pt (rows past the end read as 0: their products are never stored)
                set[NPF - 1].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsc, pt < 16 ? static_cast<unsigned>(pt) * 4u : 0x7FFFFFF0u, 0, 0));
                return;
            }
            const int64_t r0 = stage_of(t) * kSR;
            if (H24) {      // 12 bytes per lane and piece: four 3-byte elements, unpacked in split()
                const int64_t left = (R - r0) * 1152;
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    reinterpret_cast<char*>(const_cast<float*>(a)) + r0 * 1152, 0, static_cast<int>(left < (1 << 30) ? left : (1 << 30)),
                    0x00020000);
                if (K3_DBG & 8) return;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const dg_u32x3 w = __builtin_amdgcn_raw_buffer_load_b96(rsrc, static_cast<unsigned>(hw) * 2304u + static_cast<unsigned>(l32) * 12u,
                                                                             i * 384, 0);
                    set[i] = make_float4(__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), 0.f);
                }
                return;
            }
            const int64_t left = (R - r0) * 1536;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a) + r0 * 384, 0, static_cast<int>(left < (1 << 30) ? left : (1 << 30)), 0x00020000);
            if (K3_DBG & 8) return;
#pragma unroll
            for (int i = 0; i < 6; ++i)
                set[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, i * 512, 0));
        };
        // piece i of the thread: row 2 hw + i / 3, chunk i % 3; float4 column c = l32 of the chunk covers k = 4c .. 4c + 3:
        // block (chunk, c >> 1), half c & 1; row r of a block sits at position r ^ (block & 7)
        const int blk = l32 >> 1;
        auto split = [&](float4 (&set)[NPF], int t) {      // stage t -> planes[t & 1]
            char* const pl = smem + (t & 1) * kStage;
            if (H16) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    // (volatile: the first use of the loads stays behind the previous iteration's barrier)
                    asm volatile("" : "+v"(set[i].x), "+v"(set[i].y), "+v"(set[i].z), "+v"(set[i].w));
                    *reinterpret_cast<float4*>(pl + hdst[i]) = set[i];
                    if (TWO) *reinterpret_cast<float4*>(pl + kPlane + hdst[i]) = set[3 + i];
                }
                if (pt < 16) *reinterpret_cast<float*>(pl + 2 * kPlane + pt * 4) = set[NPF - 1].x;
                return;
            }
            if (K3_DBG & 2) {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int row = 2 * hw + i / 3, kc = i % 3;
                    const unsigned off = static_cast<unsigned>((kc * 16 + blk) * 256 + ((row ^ (blk & 7)) * 16) + (l32 & 1) * 8);
                    asm volatile("" : "+v"(set[i].x), "+v"(set[i].y), "+v"(set[i].z), "+v"(set[i].w));
                    *reinterpret_cast<u32x2*>(pl + off) = u32x2{__float_as_uint(set[i].x), __float_as_uint(set[i].y)};
                    *reinterpret_cast<u32x2*>(pl + kPlane + off) = u32x2{__float_as_uint(set[i].z), __float_as_uint(set[i].w)};
                }
                return;
            }
            if (H24) {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    asm volatile("" : "+v"(set[i].x), "+v"(set[i].y), "+v"(set[i].z));      // first use behind the previous barrier
                    set[i] = unpack_f24x4(dg_u32x3{__float_as_uint(set[i].x), __float_as_uint(set[i].y), __float_as_uint(set[i].z)});
                }
            }
            unsigned m[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float4& v = set[i];
                float t0, u;
                // (volatile: the first use of the loads stays behind the previous iteration's barrier)
                asm volatile("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(t0) : "v"(v.x), "v"(v.y), "v"(v.z));
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
                sc[i] = __uint_as_float((268u - e) << 23);      // 2^(14 - (e - 127)): the chunk's maximum lands in [2^14, 2^15)
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float4& v = set[i];
                f32x2 xa = f32x2{v.x, v.y} * sc[i], xb = f32x2{v.z, v.w} * sc[i];
                const f16x2 ha = __builtin_convertvector(xa, f16x2), hb = __builtin_convertvector(xb, f16x2);
                xa -= __builtin_convertvector(ha, f32x2);
                xb -= __builtin_convertvector(hb, f32x2);
                const f16x2 la = __builtin_convertvector(xa, f16x2), lb = __builtin_convertvector(xb, f16x2);
                const int row = 2 * hw + i / 3, kc = i % 3;
                const unsigned off = static_cast<unsigned>((kc * 16 + blk) * 256 + ((row ^ (blk & 7)) * 16) + (l32 & 1) * 8);
                *reinterpret_cast<u32x2*>(pl + off) = u32x2{__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
                *reinterpret_cast<u32x2*>(pl + kPlane + off) = u32x2{__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
            }
            if (l32 == 0) {
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    *reinterpret_cast<float*>(pl + 2 * kPlane + ((2 * hw + i / 3) * 3 + i % 3) * 4) = __uint_as_float((m[i] - 14u) << 23);
            }
        };
        // residual rows of stage t (requested one iteration before the tile is finished)
        float4 res[2];
        auto fetch_res = [&](int t) {
            if (!RES || (K3_DBG & 4)) return;
            if (t > T - 1) t = T - 1;
            if (t < 0) t = 0;
            const int64_t r0 = stage_of(t) * kSR;
            const int64_t left = (R - r0) * 512;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(ep.residual) + r0 * 128, 0, static_cast<int>(left < (1 << 30) ? left : (1 << 30)), 0x00020000);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                res[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                        rsrc, static_cast<unsigned>(hw) * 1024u + static_cast<unsigned>(l32) * 16u, j * 512, 0));
        };
        // finished tile of stage t: LDS -> (+ residual, LayerNorm) -> HBM.  Thread (hw, l32) owns channels 4 l32 .. of rows
        // 2 hw, 2 hw + 1; the tile stores a row's slots xor-ed with (row & 7).
        auto finish = [&](int t) {
            if (K3_DBG & 4) return;
            const bool ok = t >= 0 && t < T;
            const int tc = ok ? t : 0;
            const int64_t r0 = stage_of(tc) * kSR;
            const int64_t left = R - r0;
            const int rows = ok ? static_cast<int>(left < kSR ? left : kSR) : 0;      // 0: every store is dropped
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y + r0 * 128, 0, rows * 512, 0x00020000);
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((LN && ep.pre) ? ep.pre + r0 * 128 : y, 0,
                                                                               (LN && ep.pre) ? rows * 512 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(LN ? ep.mean + r0 : y, 0, LN ? rows * 4 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(LN ? ep.rstd + r0 : y, 0, LN ? rows * 4 : 0, 0x00020000);
            const char* ot = smem + kOffOut + (tc & 1) * kOut;
            float4 gam = f4(0.f), bet = f4(0.f);
            if (LN) {
                gam = ld4(tab + 4 * l32);
                bet = ld4(tab + 128 + 4 * l32);
            }
            const unsigned goff = static_cast<unsigned>(hw) * 1024u + static_cast<unsigned>(l32) * 16u;
            const unsigned soff = l32 == 0 ? static_cast<unsigned>(hw) * 8u : 0x7FFFFFF0u;      // one lane per row writes the statistics
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 2 * hw + j;
                float4 v = *reinterpret_cast<const float4*>(ot + row * 512 + ((l32 ^ (row & 7)) * 16));
                if (RES) v += res[j];
                if (LN) {
                    if (ep.pre) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rp, goff, j * 512, 0);
                    const float mu = half_wave_total((v.x + v.y) + (v.z + v.w), upper) * (1.0f / 128.0f);
                    const float4 d = v - f4(mu);
                    const float var = half_wave_total((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w), upper) * (1.0f / 128.0f);
                    const float rs = rsqrtf(var + ep.eps);
                    v = fma4(rs * d, gam, bet);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mu), rm, soff, j * 4, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rs), rr, soff, j * 4, 0);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, goff, j * 512, 0);
            }
        };
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) fetch(pf[u], u);
        split(pf[0], 0);
        fetch(pf[0], DEPTH);
        __syncthreads();
        for (int t = 0; t < TP; t += DEPTH) {
            // iteration t + u: the consumers are on stage t + u; planes of stage t + u + 1 are written, the tile of stage
            // t + u - 1 is finished with the residual rows requested an iteration ago, the residual rows of stage t + u are requested
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                split(pf[(u + 1) % DEPTH], t + u + 1);
                fetch(pf[(u + 1) % DEPTH], t + u + 1 + DEPTH);
                finish(t + u - 1);
                fetch_res(t + u);
                __syncthreads();
            }
        }
        finish(TP - 1);
        return;
    }

    // ---------------------------------------------------------------------------------------------- consumers
    const int n = lane & 15, kq = lane >> 4;
    // weight fragments from the packed operand (32-column slabs x 16-deep k-steps, lane = (column, k half)): channel
    // 16 w + n, k = 32 ks + 8 kq .. + 7  ->  slab (16 w + n) >> 5, k-step 2 ks + (kq >> 1), lane (kq & 1) * 32 + column
    f16x8 wf[12][NP == 1 ? 1 : 2];
    {
        const int ch = 16 * w + n;
        const int tslab = ch >> 5, col = ch & 31;
#pragma unroll
        for (int ks = 0; ks < 12; ++ks)
#pragma unroll
            for (int p = 0; p < (NP == 1 ? 1 : 2); ++p)
                wf[ks][p] = packed[(static_cast<size_t>(tslab * 24 + 2 * ks + (kq >> 1)) * 2 + p) * 64 + (kq & 1) * 32 + col];
    }
    const float* inv_cs = reinterpret_cast<const float*>(packed + static_cast<size_t>(4) * 24 * 2 * 64);
    const float4 cs = ld4(inv_cs + 16 * w + 4 * kq);
    const float4 bs = ep.bias ? ld4(ep.bias + 16 * w + 4 * kq) : f4(0.f);
    // activation fragment of lane (row n, quarter kq) in k-step ks of a chunk: block 4 ks + kq, position n ^ ((4 ks + kq) & 7)
    const unsigned xo_e = static_cast<unsigned>(kq * 256 + ((n ^ kq) * 16));            // even k-steps
    const unsigned xo_o = static_cast<unsigned>(kq * 256 + ((n ^ (4 + kq)) * 16));      // odd k-steps
    // H16: slot 4 ks + kq of chunk kc of row n sits at row n, slot 16 kc + ((4 ks + kq) ^ n)
    unsigned xo_h[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xo_h[ks] = static_cast<unsigned>(n * 768 + (((4 * ks + kq) ^ n) * 16));
    __builtin_amdgcn_s_waitcnt(0x0F70);      // weight fragments, scales and bias are in registers
    __syncthreads();                         // stage 0 is in planes[0], gamma / beta are written
    for (int t = 0; t < TP; ++t) {
        if (t < T) {
            const char* pl = smem + (t & 1) * kStage;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            if (TWO) {
                // float32 class: three products per k-step on three chains over the whole contraction, one fold (one scale per row)
                f32x4 q0, q1, q2;
#pragma unroll
                for (int kc = 0; kc < 3; ++kc)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const f16x8 xh = *reinterpret_cast<const f16x8*>(pl + kc * 256 + xo_h[ks]);
                        const f16x8 xl = *reinterpret_cast<const f16x8*>(pl + kPlane + kc * 256 + xo_h[ks]);
                        const f16x8 wh = wf[4 * kc + ks][0], wl = wf[4 * kc + ks][NP == 1 ? 0 : 1];
                        if (kc == 0 && ks == 0) {
                            mfma16_first(q0, wl, xh);
                            mfma16_first(q1, wh, xl);
                            mfma16_first(q2, wh, xh);
                        } else {
                            mfma16(q0, wl, xh);
                            mfma16(q1, wh, xl);
                            mfma16(q2, wh, xh);
                        }
                    }
                const float rs = *reinterpret_cast<const float*>(pl + 2 * kPlane + n * 4);
                mfma_results_ready();
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = ((q0[i] + q1[i]) + q2[i]) * (rs * (i == 0 ? cs.x : i == 1 ? cs.y : i == 2 ? cs.z : cs.w));
            } else if (H16) {
                // two products per k-step, six independent chains (chunk x weight plane), one fold with the row's inverse scale
                f32x4 pl0[3], ph0[3];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int kc = 0; kc < 3; ++kc) {
                        const f16x8 xh = (K3_DBG & 16) ? wf[ks][0] : *reinterpret_cast<const f16x8*>(pl + kc * 256 + xo_h[ks]);
                        const f16x8 wh = wf[4 * kc + ks][0], wl = wf[4 * kc + ks][NP == 1 ? 0 : 1];
                        if (K3_DBG & 1) {
                            if (ks == 0) pl0[kc] = ph0[kc] = f32x4{0.f, 0.f, 0.f, 0.f};
                            pl0[kc][0] += static_cast<float>(xh[0]) * static_cast<float>(wl[0]);
                        } else if (ks == 0) {
                            if (NP == 2) mfma16_first(pl0[kc], wl, xh);
                            else pl0[kc] = f32x4{0.f, 0.f, 0.f, 0.f};
                            mfma16_first(ph0[kc], wh, xh);
                        } else {
                            if (NP == 2) mfma16(pl0[kc], wl, xh);
                            mfma16(ph0[kc], wh, xh);
                        }
                    }
                const float rs = *reinterpret_cast<const float*>(pl + 2 * kPlane + n * 4);
                mfma_results_ready();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float lo = (pl0[0][i] + pl0[1][i]) + pl0[2][i], hi = (ph0[0][i] + ph0[1][i]) + ph0[2][i];
                    acc[i] = (lo + hi) * (rs * (i == 0 ? cs.x : i == 1 ? cs.y : i == 2 ? cs.z : cs.w));
                }
            }
#pragma unroll
            for (int kc = 0; kc < (H16 ? 0 : 3); ++kc) {
                // three accumulation chains per chunk (lo.hi, hi.lo, hi.hi): two independent MFMAs between dependent ones
                f32x4 p0, p1, p2;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const char* q0 = pl + kc * 4096 + ks * 1024 + ((ks & 1) ? xo_o : xo_e);
                    f16x8 xh, xl;
                    if (K3_DBG & 16) {
                        xh = wf[ks][0];
                        xl = wf[ks][NP == 1 ? 0 : 1];
                    } else {
                        xh = *reinterpret_cast<const f16x8*>(q0);
                        xl = *reinterpret_cast<const f16x8*>(q0 + kPlane);
                    }
                    const f16x8 wh = wf[4 * kc + ks][0], wl = wf[4 * kc + ks][NP == 1 ? 0 : 1];
                    if (K3_DBG & 1) {
                        if (ks == 0) p0 = p1 = p2 = f32x4{0.f, 0.f, 0.f, 0.f};
                        p0[0] += static_cast<float>(xh[0]) * static_cast<float>(wl[0]);
                        p1[0] += static_cast<float>(xl[0]) * static_cast<float>(wh[0]);
                    } else if (ks == 0) {
                        mfma16_first(p0, wl, xh);
                        mfma16_first(p1, wh, xl);
                        mfma16_first(p2, wh, xh);
                    } else {
                        mfma16(p0, wl, xh);
                        mfma16(p1, wh, xl);
                        mfma16(p2, wh, xh);
                    }
                }
                const float rs = *reinterpret_cast<const float*>(pl + 2 * kPlane + (n * 3 + kc) * 4);
                mfma_results_ready();
                acc[0] = fmaf((p0[0] + p1[0]) + p2[0], rs * cs.x, acc[0]);
                acc[1] = fmaf((p0[1] + p1[1]) + p2[1], rs * cs.y, acc[1]);
                acc[2] = fmaf((p0[2] + p1[2]) + p2[2], rs * cs.z, acc[2]);
                acc[3] = fmaf((p0[3] + p1[3]) + p2[3], rs * cs.w, acc[3]);
            }
            float4 v = make_float4(acc[0] + bs.x, acc[1] + bs.y, acc[2] + bs.z, acc[3] + bs.w);
            if (ep.relu) v = max4(v, f4(0.f));
            char* ot = smem + kOffOut + (t & 1) * kOut;
            *reinterpret_cast<float4*>(ot + n * 512 + (((4 * w + kq) ^ (n & 7)) * 16)) = v;
        }
        __syncthreads();
    }
}

}  // namespace

namespace {
struct Pending {
    bool valid = false;
    ProbK p;
};
thread_local Pending g_rider;

// the kernel variant a problem needs: residual and LayerNorm epilogues are template parameters
int variant(const ProbK& p) { return (p.ep.residual ? 1 : 0) + (p.ep.gamma ? 2 : 0) + 4 * p.afmt; }

int launch(const ProbK& p0, const ProbK* p1, hipStream_t stream) {
    const int64_t st0 = (p0.R + kSR - 1) / kSR, st1 = p1 ? (p1->R + kSR - 1) / kSR : 0;
    int nb0, nb1;
    pair_split(st0, st1, 256, &nb0, &nb1);
    const ProbK& q1 = p1 ? *p1 : p0;
#define DG_K384_LAUNCH(RES_, LN_, H16_)                                                                            \
    {                                                                                                              \
        DG_OPT_IN_LDS((&row_gemm_k384_kernel<RES_, LN_, H16_>), kLds);                                             \
        hipLaunchKernelGGL((row_gemm_k384_kernel<RES_, LN_, H16_>), dim3(nb0 + nb1), dim3(64 * (kCons + kProd)), kLds, stream, p0, \
                           q1, nb0);                                                                               \
    }
    // DG_DH_PRODUCTS=1: only the hi plane of the weights for the backward's fp16-plane operand (measured outside the parity bar:
    // the weights' rounding is the same for every row; kept for A/B runs)
    const bool dh_single = false;      // (one product for the backward's fp16-plane operand: 1.15e-3 on a golden, not offered)
    if (dh_single && (variant(p0) == 5 || variant(p0) == 4)) {      // fp16-plane operand, no LayerNorm: dx = dz + dh W1, t + vbar W2^T
        if (variant(p0) == 5) {
            DG_OPT_IN_LDS((&row_gemm_k384_kernel<true, false, 1, 1>), kLds);
            hipLaunchKernelGGL((row_gemm_k384_kernel<true, false, 1, 1>), dim3(nb0 + nb1), dim3(64 * (kCons + kProd)), kLds, stream, p0, q1, nb0);
        } else {
            DG_OPT_IN_LDS((&row_gemm_k384_kernel<false, false, 1, 1>), kLds);
            hipLaunchKernelGGL((row_gemm_k384_kernel<false, false, 1, 1>), dim3(nb0 + nb1), dim3(64 * (kCons + kProd)), kLds, stream, p0, q1, nb0);
        }
        return 0;
    }
    switch (variant(p0)) {
        case 3: DG_K384_LAUNCH(true, true, 0) break;
        case 1: DG_K384_LAUNCH(true, false, 0) break;
        case 2: DG_K384_LAUNCH(false, true, 0) break;
        case 7: DG_K384_LAUNCH(true, true, 1) break;
        case 5: DG_K384_LAUNCH(true, false, 1) break;
        case 6: DG_K384_LAUNCH(false, true, 1) break;
        case 4: DG_K384_LAUNCH(false, false, 1) break;
        case 11: DG_K384_LAUNCH(true, true, 2) break;
        case 9: DG_K384_LAUNCH(true, false, 2) break;
        case 10: DG_K384_LAUNCH(false, true, 2) break;
        case 8: DG_K384_LAUNCH(false, false, 2) break;
        case 15: DG_K384_LAUNCH(true, true, 3) break;
        case 13: DG_K384_LAUNCH(true, false, 3) break;
        case 14: DG_K384_LAUNCH(false, true, 3) break;
        case 12: DG_K384_LAUNCH(false, false, 3) break;
        default: DG_K384_LAUNCH(false, false, 0) break;
    }
#undef DG_K384_LAUNCH
    return 0;
}
}  // namespace

int flush_row_gemm_k384(hipStream_t stream) {
    if (!g_rider.valid) return 0;
    g_rider.valid = false;
    return launch(g_rider.p, nullptr, stream);
}

int launch_row_gemm_k384(const void* a, const float* ascale, const void* packed, float* y, int64_t R, const float* bias, int relu,
                         const float* residual, const float* gamma, const float* beta, float* mean, float* rstd,
                         float* pre_ln, float eps, hipStream_t stream, int afmt, const void* alo) {
    if ((afmt == 1 || afmt == 3) != (ascale != nullptr)) return fail(DG_E_ARG, "row_gemm_k384: row scales go with the fp16 planes (afmt 1, 3)");
    if ((afmt == 3) != (alo != nullptr)) return fail(DG_E_ARG, "row_gemm_k384: the lo plane goes with afmt 3");
    const ProbK p{static_cast<const float*>(a), ascale, alo, afmt, static_cast<const f16x8*>(packed), y, R,
                  EpiK{bias, residual, gamma, beta, mean, rstd, pre_ln, eps, relu, take_direction(R)}};
    if (pair_mode() && !g_rider.valid && R <= kRiderMaxRows) {      // waits for the next launch of this kernel
        g_rider.valid = true;
        g_rider.p = p;
        return 0;
    }
    if (g_rider.valid) {
        g_rider.valid = false;
        if (variant(g_rider.p) == variant(p)) return launch(p, &g_rider.p, stream);
        if (int st = launch(g_rider.p, nullptr, stream)) return st;      // another epilogue: on its own, first
    }
    return launch(p, nullptr, stream);
}

}  // namespace dg
