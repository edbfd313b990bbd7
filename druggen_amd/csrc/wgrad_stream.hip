// Weight gradient of nn.Linear over the edge rows, fp32 operands, producer / consumer form.
//
//     dW[n][k] = sum_r dy[r][n] * x[r][k]        db[n] = sum_r dy[r][n]
//
// (reference: `mm(dy.t(), x)` + `sum(dy, 0)` in every Linear backward of src/model/layers.py:50-53,111-116,127,135
// and inside the gradient penalty's double backward, src/model/loss.py:32-39).  Same arithmetic as SPLIT 2 of
// linear_wgrad.hip -- fp16 hi + lo operands under one running power-of-two scale per COLUMN of dy and of x, three
// v_mfma_f32_32x32x16_f16 per product, fp32 accumulation, exact un-scaling at the end -- but every element is
// converted ONCE per workgroup instead of once per wave that multiplies it (2.5x for the 384-wide shapes; the
// symmetric kernel issued 14.5 vector instructions per MFMA and was bound by that, profiles/r03_pmc_wgrad.txt):
//
//   waves 8..11  producers.  Producer p owns a quarter of the concatenated columns [dy | x] for the whole launch: it
//                streams its 8 x (64 / LPR) row stage HBM -> registers (16 B per lane, three stages deep: 96 KiB in
//                flight per CU), so that a lane ends up with 8 consecutive ROWS of 4 columns -- exactly the k-run of
//                an MFMA operand --, keeps the running scales of those columns, splits hi / lo and writes the two
//                fp16 planes to LDS in fragment order.  No accumulators: the registers go to the loads in flight.
//   waves 0..7   consumers.  Wave (wn, wk) owns TN x TK tiles of 32 x 32 of dW; per 16-row sub-step it reads its
//                (TN + TK) x 2 fragments with one ds_read_b128 each (lane-linear, conflict-free) and issues
//                3 TN TK MFMAs.  Nothing else in the loop.
//   one s_barrier per stage (double-buffered planes).
//
// A column whose values outgrow its scale (a handful of times per launch) is announced by its producer through a
// per-stage tag + ratio array in LDS; the consumers leave the hot loop, multiply the accumulators of that column by
// the exact power-of-two ratio and re-enter.
//
// HID instances (DG_DTYPE_F32_H16, include/druggen_hip.h): the 384-wide operand -- dy (HID = 1: dW1 = dh^T x) or x (HID = 2:
// dW2 = dz^T h) -- arrives as ONE fp16 plane with one inverse power-of-two scale per ROW (row_gemm_n384.hip's H16 output).  The
// contraction runs over the rows, so the row scale moves to the OTHER operand: sum_r dy[r][n] (s_r xh[r][k]) = sum_r (s_r dy[r][n])
// xh[r][k], exact (a power of two).  The three producers of the fp16 operand only transpose 8 rows x 4 columns of halves into
// fragment order (16 v_perm_b32, no maxima, no split, scale 1 for ever: |xh| < 2^15 by construction, 8-byte loads, six stages in
// flight); the producer of the float32 operand multiplies its rows by s_r in front of its usual running-scale split; the
// consumers run TWO products per tile pair (the fp16 operand has no lo plane).
#include "bf16.h"

#include <cstdlib>
#include "wgrad_stream.h"
#include "pair.h"

// Developer builds only (-DWS_DBG=<bits>, loaded through DG_LIB; scripts/build_variant.sh): ablations that time the
// kernel with a phase removed -- 1: no MFMAs, 2: raw stores instead of the hi / lo split, 4: no LDS writes, 8: no
// fragment reads, 16: no global loads.  The shipped library is built with WS_DBG = 0: every such block folds away.
// Measured at R = 518 400, 384 x 128 (profiles/r04_wgrad_ablation.txt): full 268-272 us; without MFMAs 211-215 (the
// same with the split, the LDS writes and the fragment reads removed as well: the load stream alone); without global
// loads 147; nothing but barriers + partial sums + the reduce launch 59.  Refilling each row pair's registers right
// after its split (loads issued ~60 % earlier) measured 279 vs 270 us: the stream is not starved for issue slots --
// the MFMA phase costs clock (1.5 GHz against 1.93 GHz for the 128 x 128 shape, SQ_BUSY_CU_CYCLES / duration).
#ifndef WS_DBG
#define WS_DBG 0
#endif
namespace dg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 2^(8 - floor(log2 m)) for a finite m > 0 (see linear_wgrad.hip: a column's scale moves again only for a value
// 64-128 times larger than the one that set it)
__device__ __forceinline__ float ws_scale_for(float m) {
    const int e = static_cast<int>((__float_as_uint(m) >> 23) & 255u);
    int be = 127 + 8 - (e - 127);
    be = be > 253 ? 253 : (be < 1 ? 1 : be);
    return __uint_as_float(static_cast<unsigned>(be) << 23);
}
__device__ __forceinline__ float ws_pow2_ratio(float num, float den) {      // num <= den, both powers of two
    const int d = static_cast<int>(__float_as_uint(num) >> 23) - static_cast<int>(__float_as_uint(den) >> 23) + 127;
    return d < 1 ? 0.f : __uint_as_float(static_cast<unsigned>(d) << 23);
}
__device__ __forceinline__ float ws_pow2_inv(float p) { return __uint_as_float((254u - (__float_as_uint(p) >> 23)) << 23); }

// hi = s rounded toward zero to fp16, lo = s - hi (exact in fp32) rounded toward zero, for the pair (v0, v1) * sc
__device__ __forceinline__ void ws_split2(float v0, float v1, float sc, unsigned& hw, unsigned& lw) {
    const float s0 = v0 * sc, s1 = v1 * sc;
    hw = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(s0, s1));
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hw), "v"(s0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hw), "v"(s1));
    lw = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l0, l1));
}
__device__ __forceinline__ float comp4(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

constexpr int kConsumers = 8, kProducers = 4, kDepth = 3;
constexpr int kStageBytes = 32768;                 // one stage of planes: SR rows x (N + K) columns x (hi + lo)
constexpr int kHdr = 2 * kStageBytes;              // tags [2][4] u32, then ratios [2][N + K], then final inverse scales [N + K]

// One problem of a launch; workgroups [0, nb0) run problem 0, the others problem 1 (a node-level weight gradient riding
// in the edge-level launch of the same shape: pair.h).  Each problem has its own partial sums.
struct ProbW {
    const float* dy;
    const float* dy1;
    const float* dy2;
    const float* x;
    const float* hscale;      // hfmt 1: inverse row scales [R] of the fp16 operand
    int hfmt;                 // storage of the 384-wide operand: 0 float32, 1 fp16 plane + row scales, 2 three-byte elements
    float* part_w;
    float* part_b;
    int64_t R;
};

template <int NT, int KT, int CN, int CK, int HID = 0>
__global__ __launch_bounds__(64 * (kConsumers + kProducers)) void wgrad_stream_kernel(const ProbW p0, const ProbW p1, const int nb0) {
    const bool second = static_cast<int>(blockIdx.x) >= nb0;      // uniform
    const float* __restrict__ const dy = second ? p1.dy : p0.dy;
    const float* __restrict__ const dy1 = second ? p1.dy1 : p0.dy1;
    const float* __restrict__ const dy2 = second ? p1.dy2 : p0.dy2;
    const float* __restrict__ const x = second ? p1.x : p0.x;
    const float* __restrict__ const hscale = second ? p1.hscale : p0.hscale;
    float* __restrict__ const part_w = second ? p1.part_w : p0.part_w;
    float* __restrict__ const part_b = second ? p1.part_b : p0.part_b;
    const int64_t R = second ? p1.R : p0.R;
    const int bidx = second ? static_cast<int>(blockIdx.x) - nb0 : static_cast<int>(blockIdx.x);
    const int nblk = second ? static_cast<int>(gridDim.x) - nb0 : nb0;
    constexpr int N = NT * 32, K = KT * 32, COLS = N + K, NTILES = NT + KT;
    constexpr int TN = NT / CN, TK = KT / CK;
    constexpr int CW = COLS / kProducers;          // columns per producer
    constexpr int LPR = CW / 4, RG = 64 / LPR;     // lanes per row, row groups per wave
    constexpr int SR = 8 * RG, SUB = SR / 16;      // rows per stage, 16-row MFMA sub-steps per stage
    static_assert(CN * CK == kConsumers && NT % CN == 0 && KT % CK == 0, "consumer grid");
    static_assert(SR * COLS * 4 == kStageBytes && (LPR == 32 || LPR == 16) && N % CW == 0, "stage geometry");
    // HID 3 / 4 (DG_DTYPE_F32_H24): dy / x holds the top 24 bits of every float32 (3 bytes per element): the producers of those
    // columns fetch 12 bytes per row instead of 16 and unpack; everything else is the float32 kernel
    // HID 5 (128 x 128 only): x is float32 in HBM but enters the products as ONE fp16 plane -- a weight gradient is a leaf of the
    // backward, the 2^-12 rounding of its activation operand is independent from element to element and averages over the rows
    // (the same arithmetic as dW2 = dz^T h_hi and dW1 = dh^T x of the feed-forward): two products, no lo plane of x
    static_assert(HID == 0 || (HID == 5 && N == 128 && K == 128) ||
                      (CW == 128 && (((HID == 1 || HID == 3) && N == 384) || ((HID == 2 || HID == 4) && K == 384))),
                  "narrow operand: the 384-wide one");
    constexpr bool HS = HID == 1 || HID == 2;                 // fp16 plane + row scales
    constexpr int DEPTH_ALL = HS ? 2 * kDepth : kDepth;       // iterations are padded to whole groups of this
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* const tags = reinterpret_cast<unsigned*>(smem + kHdr);                       // [2][4]
    float* const ratios = reinterpret_cast<float*>(smem + kHdr + 64);                     // [2][COLS]
    float* const fin = ratios + 2 * COLS;                                                 // [COLS] inverse scales
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // stages of this workgroup: a contiguous range, as even as possible
    const int64_t total = (R + SR - 1) / SR;
    const int64_t q = total / nblk, rem = total % nblk;
    const int64_t s_lo = bidx * q + (bidx < rem ? bidx : rem);
    const int T = static_cast<int>(q + (bidx < rem ? 1 : 0));      // >= 1 (workgroups of a problem <= its stages)
    const int TP = (T + DEPTH_ALL - 1) / DEPTH_ALL * DEPTH_ALL;          // iterations incl. padding (whole groups)

    if (w >= kConsumers) {
        // ------------------------------------------------------------------------------------------ producers
        __builtin_amdgcn_s_setprio(3);
        const int p = w - kConsumers;
        const bool is_dy = p * CW < N;
        if (HS && is_dy == (HID == 1)) {
            // ---- a 128-column chunk of the fp16 operand: transpose into fragment order, nothing else
            const _Float16* hsrc = reinterpret_cast<const _Float16*>(is_dy ? dy : x);
            const int coff = is_dy ? p * CW : p * CW - N;
            const int cq = lane & 31, rg = lane >> 5;
            const unsigned voff0 = static_cast<unsigned>((8 * rg) * 384 + coff + 4 * cq) * 2u;
            unsigned wa[4];
            {
                const int tile = (p * CW + 4 * cq) >> 5, kh = rg & 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = 4 * (cq & 7) + j, pc = c ^ (c >> 3);
                    wa[j] = static_cast<unsigned>(tile * 2048 + kh * 512 + pc * 16);
                }
            }
            if (lane == 0) {
                tags[p] = 0u;
                tags[4 + p] = 0u;
            }
            constexpr int D = 2 * kDepth;
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2 ph[D][8];
            float psc[D];
            float4 bsum = f4(0.f);
            const bool want_b = part_b && is_dy;
            auto fetch = [&](u32x2 (&set)[8], float& sv, int t) {
                if (t > T - 1) t = T - 1;
                const int64_t r0 = (s_lo + t) * SR;
                const int64_t left = (R - r0) * 768;
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<_Float16*>(hsrc) + r0 * 384, 0, static_cast<int>(left < (1 << 30) ? left : (1 << 30)), 0x00020000);
                if (want_b) {      // lane l: the inverse scale of row 8 (l >> 5) + (l & 7) of the stage
                    const int64_t lefts = (R - r0) * 4;
                    const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float*>(hscale) + r0, 0, static_cast<int>(lefts < SR * 4 ? lefts : SR * 4), 0x00020000);
                    sv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsc, static_cast<unsigned>(8 * rg + (lane & 7)) * 4u, 0, 0));
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) set[i] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff0, i * 768, 0);
            };
            auto process = [&](u32x2 (&set)[8], float sv, int t) {
                char* const st = smem + (t & 1) * kStageBytes;
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(set[i]));      // first use behind the previous barrier
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u32x4 hw;
#pragma unroll
                    for (int pr = 0; pr < 4; ++pr)
                        hw[pr] = __builtin_amdgcn_perm(set[2 * pr + 1][j >> 1], set[2 * pr][j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
                    if (!(WS_DBG & 4)) *reinterpret_cast<u32x4*>(st + wa[j]) = hw;
                }
                if (want_b && t < T) {      // db[n] = sum_r s_r xh[r][n]
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), i));
                        const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), 32 + i));
                        const float si = rg ? hi : lo;
                        const f16x8 hv = __builtin_bit_cast(f16x8, u32x4{set[i][0], set[i][1], 0u, 0u});
                        bsum.x = fmaf(static_cast<float>(hv[0]), si, bsum.x);
                        bsum.y = fmaf(static_cast<float>(hv[1]), si, bsum.y);
                        bsum.z = fmaf(static_cast<float>(hv[2]), si, bsum.z);
                        bsum.w = fmaf(static_cast<float>(hv[3]), si, bsum.w);
                    }
                }
            };
#pragma unroll
            for (int u = 0; u < D; ++u) fetch(ph[u], psc[u], u);
            process(ph[0], psc[0], 0);
            fetch(ph[0], psc[0], D);
            __syncthreads();
            for (int t = 1; t < TP + 1; t += D) {
#pragma unroll
                for (int u = 0; u < D; ++u) {
                    process(ph[(u + 1) % D], psc[(u + 1) % D], t + u);
                    fetch(ph[(u + 1) % D], psc[(u + 1) % D], t + u + D);
                    __syncthreads();
                }
            }
            if (rg == 0) st4(fin + p * CW + 4 * cq, f4(1.0f));      // the plane is exact: scale 1
            __syncthreads();
            if (want_b) {
                bsum.x = xor_step<false>(bsum.x, 32);
                bsum.y = xor_step<false>(bsum.y, 32);
                bsum.z = xor_step<false>(bsum.z, 32);
                bsum.w = xor_step<false>(bsum.w, 32);
                if (rg == 0) st4(part_b + static_cast<size_t>(bidx) * N + p * CW + 4 * cq, bsum);
            }
            return;
        }
        // dy1 / dy2 non-null (N = 384, CW = 128): the three 128-column blocks of dy are three separate [R,128] matrices
        // (dq, dk, dv of an attention block: one launch gives the stacked weight gradient of q / k / v)
        const bool three = dy1 != nullptr && is_dy;
        const float* src = three ? (p == 0 ? dy : (p == 1 ? dy1 : dy2)) : (is_dy ? dy : x);
        const int LD = three ? 128 : (is_dy ? N : K);
        const int coff = three ? 0 : (is_dy ? p * CW : p * CW - N);
        const int cq = lane % LPR, rg = lane / LPR;
        const bool f24 = (HID == 3 && is_dy) || (HID == 4 && !is_dy);      // this producer's columns are 3-byte elements
        const unsigned voff0 = static_cast<unsigned>((8 * rg) * LD + coff + 4 * cq) * (f24 ? 3u : 4u);
        const int rowb = LD * (f24 ? 3 : 4);
        // LDS byte offsets of this lane's four columns inside a stage (hi plane; lo plane = + 1024)
        unsigned wa[4];
        {
            const int tile = (p * CW + 4 * cq) >> 5, u = RG == 4 ? rg >> 1 : 0, kh = rg & 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 4 * (cq & 7) + j, pc = c ^ (c >> 3);
                wa[j] = static_cast<unsigned>((u * NTILES + tile) * 2048 + kh * 512 + pc * 16);
            }
        }
        if (lane == 0) {      // no stale tag may look like a stage of this launch
            tags[p] = 0u;
            tags[4 + p] = 0u;
        }
        float sc[4] = {8.507059e37f, 8.507059e37f, 8.507059e37f, 8.507059e37f};      // 2^126: nothing seen yet
        float4 bsum = f4(0.f);
        float4 pf[kDepth][8];
        float psc[kDepth];      // HID: lane l holds the fp16 operand's inverse scale of row 8 (l >> 5) + (l & 7) of the stage

        // Stage loads go through a buffer descriptor whose range ends at the last row of the launch: rows past the end
        // read as zeros (no clamping, no tail branch), the eight rows of a lane share ONE offset register (row i is a
        // scalar / immediate offset).  Straight-line on purpose: with branches around the loads hipcc can no longer
        // count how many younger loads may stay in flight and drains the queue (vmcnt(0)) before every split.
        auto fetch = [&](float4 (&set)[8], float& sv, int t) {
            if (t > T - 1) t = T - 1;
            const int64_t r0 = (s_lo + t) * SR;
            const int64_t left = (R - r0) * rowb;      // bytes from the stage's first row to the end of the matrix
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<char*>(const_cast<float*>(src)) + r0 * rowb, 0, static_cast<int>(left < (1 << 30) ? left : (1 << 30)),
                0x00020000);
            if (HID >= 3 && f24) {      // (wave-uniform) 12 bytes per row: unpacked in process()
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const dg_u32x3 w = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff0, i * rowb, 0);
                    set[i] = make_float4(__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), 0.f);
                }
                return;
            }
            if (HS) {      // (in front of the data loads: loads return in order)
                const int64_t lefts = (R - r0) * 4;
                const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(hscale) + r0, 0, static_cast<int>(lefts < SR * 4 ? lefts : SR * 4), 0x00020000);
                sv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsc, static_cast<unsigned>(8 * (lane >> 5) + (lane & 7)) * 4u, 0, 0));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (WS_DBG & 16) set[i] = f4(static_cast<float>(i + t));
                else set[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff0, i * rowb, 0));
            }
        };
        auto process = [&](float4 (&set)[8], float sv, int t) {
            const bool live = t < T;
            if (HID >= 3 && f24) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    asm volatile("" : "+v"(set[i].x), "+v"(set[i].y), "+v"(set[i].z));
                    set[i] = unpack_f24x4(dg_u32x3{__float_as_uint(set[i].x), __float_as_uint(set[i].y), __float_as_uint(set[i].z)});
                }
            }
            if (HS) {
                // the column sums of dy are sums of the UNSCALED rows; then the rows take the fp16 operand's row scales
                if (part_b && is_dy && live) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) bsum += set[i];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), i));
                    const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), 32 + i));
                    set[i] = (rg ? hi : lo) * set[i];
                }
            }
            // column maxima over the stage's rows: 8 in this lane, the rest in the lanes of the other row groups
            float m[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a, b2, c2;
                asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(a) : "v"(comp4(set[0], j)), "v"(comp4(set[1], j)), "v"(comp4(set[2], j)));
                asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(b2) : "v"(a), "v"(comp4(set[3], j)), "v"(comp4(set[4], j)));
                asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(c2) : "v"(b2), "v"(comp4(set[5], j)), "v"(comp4(set[6], j)));
                asm("v_max_f32_e64 %0, %1, |%2|" : "=v"(m[j]) : "v"(c2), "v"(comp4(set[7], j)));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                m[j] = xor_step<true>(m[j], 32);
                if (RG == 4) m[j] = xor_step<true>(m[j], 16);
            }
            bool grow = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) grow |= m[j] * sc[j] >= 32768.f;
            if (__builtin_expect(live && __builtin_amdgcn_ballot_w64(grow) != 0, 0)) {
                // some column of this wave outgrew its scale: new scales, the exact ratios go to the consumers
                float rp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float nsc = m[j] * sc[j] >= 32768.f ? ws_scale_for(m[j]) : sc[j];
                    rp[j] = ws_pow2_ratio(nsc, sc[j]);
                    sc[j] = nsc;
                }
                if (rg == 0) st4(ratios + (t & 1) * COLS + p * CW + 4 * cq, make_float4(rp[0], rp[1], rp[2], rp[3]));
                if (lane == 0) tags[(t & 1) * 4 + p] = static_cast<unsigned>(t) + 1u;
            }
            // split with the running scales -> two fp16 planes in fragment order (16 B = this lane's 8 rows of a column)
            char* const st = smem + (t & 1) * kStageBytes;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4 hw, lw;
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    unsigned h, l;
                    if (WS_DBG & 2) {
                        h = __float_as_uint(comp4(set[2 * pr], j));
                        l = __float_as_uint(comp4(set[2 * pr + 1], j));
                    } else {
                        ws_split2(comp4(set[2 * pr], j), comp4(set[2 * pr + 1], j), sc[j], h, l);
                    }
                    hw[pr] = h;
                    lw[pr] = l;
                }
                if (WS_DBG & 4) {
                    asm volatile("" ::"v"(hw), "v"(lw));
                } else {
                    *reinterpret_cast<u32x4*>(st + wa[j]) = hw;
                    if (!(HID == 5 && !is_dy)) *reinterpret_cast<u32x4*>(st + wa[j] + 1024) = lw;
                }
            }
            if (!HS && part_b && is_dy && live) {
#pragma unroll
                for (int i = 0; i < 8; ++i) bsum += set[i];
            }
        };
        fetch(pf[0], psc[0], 0);
        fetch(pf[1], psc[1], 1);
        fetch(pf[2], psc[2], 2);
        process(pf[0], psc[0], 0);
        fetch(pf[0], psc[0], 3);
        __syncthreads();
        for (int t = 1; t < TP + 1; t += 3) {
            // iteration t writes stage t (the consumers are on stage t - 1) and refills its register set
            process(pf[1], psc[1], t);
            fetch(pf[1], psc[1], t + 3);
            __syncthreads();
            process(pf[2], psc[2], t + 1);
            fetch(pf[2], psc[2], t + 4);
            __syncthreads();
            process(pf[0], psc[0], t + 2);
            fetch(pf[0], psc[0], t + 5);
            __syncthreads();
        }
        // final inverse scales for the consumers' un-scaling; column sums of dy
        if (rg == 0) {
            float4 inv = make_float4(ws_pow2_inv(sc[0]), ws_pow2_inv(sc[1]), ws_pow2_inv(sc[2]), ws_pow2_inv(sc[3]));
            st4(fin + p * CW + 4 * cq, inv);
        }
        __syncthreads();
        if (part_b && is_dy) {
            bsum.x = xor_step<false>(bsum.x, 32);
            bsum.y = xor_step<false>(bsum.y, 32);
            bsum.z = xor_step<false>(bsum.z, 32);
            bsum.w = xor_step<false>(bsum.w, 32);
            if (RG == 4) {
                bsum.x = xor_step<false>(bsum.x, 16);
                bsum.y = xor_step<false>(bsum.y, 16);
                bsum.z = xor_step<false>(bsum.z, 16);
                bsum.w = xor_step<false>(bsum.w, 16);
            }
            if (rg == 0) st4(part_b + static_cast<size_t>(bidx) * N + p * CW + 4 * cq, bsum);
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------- consumers
    const int wn = w % CN, wk = w / CN;
    const int half = lane >> 5, col = lane & 31;
    f32x16 acc[TN][TK];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    // fragment of lane (c, kh) inside a 1 KiB (tile, plane) block: [kh][c ^ (c >> 3)][16 B]
    const unsigned lane_off = static_cast<unsigned>(half * 512 + ((col ^ (col >> 3)) * 16));
    const unsigned off_y = lane_off + static_cast<unsigned>(wn * TN) * 2048u;
    const unsigned off_x = lane_off + static_cast<unsigned>(NT + wk * TK) * 2048u;
    // producers of this wave's tiles (tile -> column / CW)
    auto prod_y = [&](int i) { return ((wn * TN + i) * 32) / CW; };
    auto prod_x = [&](int j) { return (N + (wk * TK + j) * 32) / CW; };

    __syncthreads();      // stage 0 is in slot 0
    int t = 0;
    bool recheck = true;
    while (t < TP) {
        bool moved = false;
        for (; t < TP; ++t) {
            if (t < T) {
                const char* st = smem + (t & 1) * kStageBytes;
                f16x8 yh[TN], yl[TN], xh[TK], xl[TK];
                auto read_frags = [&](int u) {
                    if (WS_DBG & 8) {
#pragma unroll
                        for (int i = 0; i < TN; ++i) yh[i] = yl[i] = f16x8{};
#pragma unroll
                        for (int j = 0; j < TK; ++j) xh[j] = xl[j] = f16x8{};
                        return;
                    }
#pragma unroll
                    for (int i = 0; i < TN; ++i) {
                        yh[i] = *reinterpret_cast<const f16x8*>(st + off_y + (u * NTILES + i) * 2048);
                        if (HID != 1) yl[i] = *reinterpret_cast<const f16x8*>(st + off_y + (u * NTILES + i) * 2048 + 1024);
                    }
#pragma unroll
                    for (int j = 0; j < TK; ++j) {
                        xh[j] = *reinterpret_cast<const f16x8*>(st + off_x + (u * NTILES + j) * 2048);
                        if (HID != 2 && HID != 5) xl[j] = *reinterpret_cast<const f16x8*>(st + off_x + (u * NTILES + j) * 2048 + 1024);
                    }
                };
                // the stage's tags first, the first sub-step's fragments right behind them: the tag test then waits for
                // ONE LDS round trip with the fragment reads already in flight
                const u32x4 tg = *reinterpret_cast<const u32x4*>(tags + (t & 1) * 4);
                read_frags(0);
                if (recheck) {
                    const unsigned want = static_cast<unsigned>(t) + 1u;
                    const bool any = tg[0] == want || tg[1] == want || tg[2] == want || tg[3] == want;
                    if (__builtin_expect(__builtin_amdgcn_readfirstlane(any ? 1 : 0) != 0, 0)) {
                        moved = true;
                        break;
                    }
                }
                recheck = true;
#pragma unroll
                for (int u = 0; u < SUB; ++u) {
                    if (u > 0) read_frags(u);
                    // lo.hi, hi.lo, hi.hi: smallest terms first; consecutive MFMAs hit different accumulators
#pragma unroll
                    for (int part = 0; part < 3; ++part)
#pragma unroll
                        for (int i = 0; i < TN; ++i)
#pragma unroll
                            for (int j = 0; j < TK; ++j) {
                                if ((HID == 1 && part == 0) || ((HID == 2 || HID == 5) && part == 1)) continue;      // no lo plane of the fp16 operand
                                if (WS_DBG & 1) asm volatile("" ::"v"(yl[i]), "v"(yh[i]), "v"(xl[j]), "v"(xh[j]));
                                else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(part == 0 ? yl[i] : yh[i], part == 1 ? xl[j] : xh[j],
                                                                                       acc[i][j], 0, 0, 0);
                            }
                }
            }
            __syncthreads();
        }
        if (!moved) break;
        // rare: the producers of some of this wave's columns moved their scales at stage t
        {
            const unsigned want = static_cast<unsigned>(t) + 1u;
            const float* rt = ratios + (t & 1) * COLS;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                if (tags[(t & 1) * 4 + prod_y(i)] != want) continue;      // wave-uniform
                // dy columns are accumulator ROWS: row (reg & 3) + 8 (reg >> 2) + 4 half
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const float r = rt[(wn * TN + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half];
#pragma unroll
                    for (int j = 0; j < TK; ++j) acc[i][j][reg] *= r;
                }
            }
#pragma unroll
            for (int j = 0; j < TK; ++j) {
                if (tags[(t & 1) * 4 + prod_x(j)] != want) continue;
                const float r = rt[N + (wk * TK + j) * 32 + col];          // x columns are accumulator COLUMNS
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) acc[i][j][reg] *= r;
            }
            recheck = false;
        }
    }
    __syncthreads();      // final inverse scales are in LDS
    float* pw = part_w + static_cast<size_t>(bidx) * N * K;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) {
            const float ix = fin[N + (wk * TK + j) * 32 + col];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
                // two exact multiplications (each factor inside the fp32 range)
                const float val = acc[i][j][reg] * fin[(wn * TN + i) * 32 + row] * ix;
                pw[((wn * TN + i) * 32 + row) * K + (wk * TK + j) * 32 + col] = val;
            }
        }
}

}  // namespace

bool wgrad_stream_supported(int N, int K) {
    return (N == 128 && K == 128) || (N == 384 && K == 128) || (N == 128 && K == 384);
}

// rows per stage of the shape (16 for the 512-column shapes, 32 for 128 x 128)
static int stage_rows(int N, int K) { return 32768 / ((N + K) * 4); }


namespace {
struct Pending {
    bool valid = false;
    ProbW p;
    int N = 0, K = 0, blocks = 0;
};
thread_local Pending g_rider;

int launch(const ProbW& p0, int nb0, const ProbW* p1, int nb1, int N, int K, hipStream_t stream) {
    const int lds = kHdr + 64 + 3 * (N + K) * 4;
    const ProbW& q1 = p1 ? *p1 : p0;
#define DG_WS_LAUNCH(NT_, KT_, CN_, CK_, HID_)                                                                    \
    {                                                                                                            \
        DG_OPT_IN_LDS((&wgrad_stream_kernel<NT_, KT_, CN_, CK_, HID_>), lds);                                     \
        hipLaunchKernelGGL((wgrad_stream_kernel<NT_, KT_, CN_, CK_, HID_>), dim3(nb0 + nb1), dim3(64 * (kConsumers + kProducers)), \
                           lds, stream, p0, q1, nb0);                                                            \
    }
    const int fmt = p0.hfmt;      // 0 float32, 1 fp16 plane + row scales, 2 three-byte elements: the 384-wide operand
    if (fmt && p0.dy1) return fail(DG_E_ARG, "wgrad_stream: three dy matrices cannot be combined with a narrow operand");
    // DG_WGRAD128_PRODUCTS=2: the activation operand of the 128 x 128 weight gradients as ONE fp16 plane.  Measured (round 5): no
    // gain in the step (51.46 vs 51.49 ms) and 1.8e-3 on the c5_b2 golden (180 node rows: nothing averages) -- not the default.
    const bool x_single = false;      // (two products for the 128 x 128 shape measured outside the parity bar: not offered)
    if (N == 128 && K == 128 && !fmt && x_single) DG_WS_LAUNCH(4, 4, 4, 2, 5)
    else if (N == 128 && K == 128 && !fmt) DG_WS_LAUNCH(4, 4, 4, 2, 0)
    else if (N == 384 && K == 128 && !fmt) DG_WS_LAUNCH(12, 4, 4, 2, 0)
    else if (N == 128 && K == 384 && !fmt) DG_WS_LAUNCH(4, 12, 2, 4, 0)
    else if (N == 384 && K == 128 && fmt == 1) DG_WS_LAUNCH(12, 4, 4, 2, 1)
    else if (N == 128 && K == 384 && fmt == 1) DG_WS_LAUNCH(4, 12, 2, 4, 2)
    else if (N == 384 && K == 128 && fmt == 2) DG_WS_LAUNCH(12, 4, 4, 2, 3)
    else if (N == 128 && K == 384 && fmt == 2) DG_WS_LAUNCH(4, 12, 2, 4, 4)
    else return fail(DG_E_SHAPE, "wgrad_stream: unsupported shape N=%d K=%d", N, K);
#undef DG_WS_LAUNCH
    return 0;
}
}  // namespace

// Workgroups (= partial sums) of a launch, fixed when the call is made: the caller lays out the workspace and records the
// reduce with this count.  A launch that will wait for a carrier (pair.h) gets its share of a full grid next to an
// edge-level problem -- one workgroup per ~2 000 rows; the carrier then takes the rest of the 256.
int wgrad_stream_blocks(int64_t R, int N, int K, bool may_wait) {
    const int64_t stages = (R + stage_rows(N, K) - 1) / stage_rows(N, K);
    if (may_wait && pair_mode() && R <= kRiderMaxRows && !g_rider.valid) {
        int64_t b = (R + 1023) / 2048;
        b = b < 1 ? 1 : (b > 32 ? 32 : b);
        return static_cast<int>(b > stages ? stages : b);
    }
    // every workgroup writes an [N, K] partial: short launches use fewer workgroups (>= 4 stages each)
    int64_t blocks = stages / 4;
    blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
    if (g_rider.valid && g_rider.N == N && g_rider.K == K && blocks + g_rider.blocks > 256) blocks = 256 - g_rider.blocks;
    return static_cast<int>(blocks);
}

int flush_wgrad_stream(hipStream_t stream) {
    if (!g_rider.valid) return 0;
    g_rider.valid = false;
    return launch(g_rider.p, g_rider.blocks, nullptr, 0, g_rider.N, g_rider.K, stream);
}

int launch_wgrad_stream(const void* dy, const void* x, float* part_w, float* part_b, int64_t R, int N, int K, int blocks,
                        hipStream_t stream, const float* dy1, const float* dy2, bool may_wait, const float* hscale, int hfmt) {
    if ((dy1 || dy2) && (!dy1 || !dy2 || N != 384 || K != 128))
        return fail(DG_E_ARG, "wgrad_stream: three dy matrices need N = 384, K = 128");
    if (!wgrad_stream_supported(N, K)) return fail(DG_E_SHAPE, "wgrad_stream: unsupported shape N=%d K=%d", N, K);
    if (hfmt && N + K != 512) return fail(DG_E_SHAPE, "wgrad_stream: a narrow operand needs N = 384 or K = 384");
    if ((hfmt == 1) != (hscale != nullptr)) return fail(DG_E_ARG, "wgrad_stream: row scales go with the fp16 plane (hfmt 1)");
    const ProbW p{static_cast<const float*>(dy), dy1, dy2, static_cast<const float*>(x), hscale, hfmt, part_w, part_b, R};
    if (may_wait && pair_mode() && !g_rider.valid && R <= kRiderMaxRows) {      // waits for the next launch of this shape
        g_rider.valid = true;
        g_rider.p = p;
        g_rider.N = N;
        g_rider.K = K;
        g_rider.blocks = blocks;
        return 0;
    }
    if (g_rider.valid) {
        g_rider.valid = false;
        if (g_rider.N == N && g_rider.K == K && blocks + g_rider.blocks <= 256 && g_rider.p.hfmt == hfmt)
            return launch(p, blocks, &g_rider.p, g_rider.blocks, N, K, stream);
        if (int st = launch(g_rider.p, g_rider.blocks, nullptr, 0, g_rider.N, g_rider.K, stream)) return st;
    }
    return launch(p, blocks, nullptr, 0, N, K, stream);
}

}  // namespace dg
