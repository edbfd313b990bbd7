// 128 -> 384 row GEMM (fc1 of the feed-forward: y = relu(x W1^T + b1), reference src/model/layers.py:50-51, and its
// twin in the backward, dh = (dz W2) * relu-mask), fp32 rows, producer / consumer form.
//
// Same arithmetic as row_gemm.hip (fp16 hi + lo planes under an exact power-of-two scale per ROW of the activation
// chunk and per output COLUMN of the weight, three MFMA products, fp32 accumulation, inverse scales in the epilogue),
// same packed weight (dg_row_gemm_pack).  What differs is who touches global memory.  In the 6 + 2 wave kernel
// (row_gemm_h3_kernel<1,1,false,6>) every consumer wave ran its MFMA phase, then scaled / masked / transposed / stored
// its own 2 x 2 blocks, and two of the four SIMDs hosted two consumer waves in lock step: the kernel's phases added up
// instead of overlapping (profiles/r04_gemm_ablation.txt: no MFMAs 211 us, no stores 206, no fetch 211, none of the
// three 88 of 296).  Here a workgroup is 12 waves, 3 per SIMD:
//
//   waves 8..11  producers: every global access.  A rows HBM -> registers (16-byte buffer loads whose range ends at the
//                last row: no clamps, no tail branch; three stages deep) -> row maximum -> hi / lo planes in LDS; and
//                the FINISHED output tile of the previous stage LDS -> HBM as whole 1536-byte rows, 16 bytes per lane.
//   waves 0..7   consumers: wave w owns output channels [48 w, 48 w + 48) with its weight fragments resident (96
//                VGPRs, gathered once from the packed weight).  Swapped product on v_mfma_f32_16x16x32_f16 (weights =
//                A operand, activations = B operand): a lane ends up with 4 consecutive channels of ONE row.  Per
//                32-row stage: 16 ds_read_b128, 72 MFMAs, then scale + bias + ReLU (+ bit masks) and one 16-byte LDS
//                write per block into the output tile.  No global memory operation at all: the ReLU mask words travel
//                through LDS too (a consumer that stored its own word waited for that store, vmcnt(0), every stage).
//   one s_barrier per stage; planes and output tile double-buffered.
//
// H16 instance (DG_DTYPE_F32_H16, include/druggen_hip.h): the [R,384] result leaves as ONE fp16 plane -- every row scaled by the
// power of two that puts its largest magnitude into [2^14, 2^15), rounded to nearest -- plus one inverse scale per row (768 + 4
// bytes per row instead of 1536).  The producers do it on the way out of the output tile: a half-wave owns a whole row there,
// so the row maximum is the same four DPP steps + one v_permlane16_swap as on the way in.
#include "common.h"

#include <cstdlib>
#include "row_gemm_n384.h"
#include "pair.h"
#include "traversal.h"

#ifndef N3_SCHED
#define N3_SCHED 1    // consumers: fragment reads one k-step ahead, per row block (0: hipcc's order, A/B builds)
#endif
#ifndef N3_DBG
#define N3_DBG 0      // ablation builds (scripts/build_variant.sh, scripts/n384_variants.sh): 1 no MFMAs, 2 no epilogue arithmetic
#endif                // (raw accumulators to the tile), 4 no output stores, 8 no global fetch of A
namespace dg {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kSR = 32;                                  // rows per stage
constexpr int kPlane = 16 * 512;                         // [k-step 4][k-quarter 4][row 32 (xor-swizzled)][16 B]
constexpr int kStage = 2 * kPlane + 128;                 // hi, lo, inverse row scales [32]
constexpr int kOut = kSR * 96 * 16;                      // output tile: [row 32][16-byte slot 96 (xor row & 7)]
constexpr int kOffOut = 2 * kStage;
constexpr int kOffTab = kOffOut + 2 * kOut;              // inverse column scales [384], bias [384]
constexpr int kOffBitsIn = kOffTab + 2 * 384 * 4;        // ReLU mask words of a stage [2][8 waves][64 lanes] (with the planes)
constexpr int kOffBitsOut = kOffBitsIn + 2 * 2048;       // ... written by the consumers [2][8][64] (with the output tile)
constexpr int kLds = kOffBitsOut + 2 * 2048;
constexpr int kCons = 8, kProd = 4, kDepth = 3;

template <int CTRL>
__device__ __forceinline__ unsigned umax_dpp(unsigned x) {
    const unsigned moved = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, 0xF, 0xF, true));
    return x > moved ? x : moved;
}

// acc += A . B on v_mfma_f32_16x16x32_f16, ALWAYS in place, and the fence in front of the first vector read of a result:
// see row_gemm_k384.hip (hipcc's renamed destinations one slot behind the producing MFMA read partly written accumulators).
__device__ __forceinline__ void mfma16(f32x4& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// first MFMA of a chain: accumulator input = the constant 0 (no vector write of the accumulator in front of the chain)
__device__ __forceinline__ void mfma16_first(f32x4& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_results_ready() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

// packed fp16 pair { fp16(s0 - hi.lo), fp16(s1 - hi.hi) }: the lo plane of two scaled values whose hi plane is `hpk`
__device__ __forceinline__ unsigned lo_pair(unsigned hpk, float s0, float s1) {
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hpk), "v"(s0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hpk), "v"(s1));
    return d;
}

struct Epi {
    const float* bias;            // [384] or null
    const unsigned* mask_bits;    // [stages][8][64] words written by a launch with relu_bits of the SAME geometry, or null
    unsigned* relu_bits;          // optional output, same layout: bit (rb * 3 + cb) * 4 + i = (v > 0)
    int relu;
    int reverse;                  // stages in descending order (traversal.h)
};

// One problem of a launch.  A launch carries one or two: workgroups [0, nb0) run problem 0, the others problem 1 -- a
// node-level GEMM (R = B N rows) riding in the edge-level launch (R = B N^2) of the same kernel (pair.h).
struct Prob {
    const float* a;
    const f16x8* packed;
    float* y;             // [R,384] float32, or (H16) [R,384] fp16, or (yfmt 2) [R,384] 3-byte elements
    float* yscale;        // H16 / H32: inverse row scales [R]
    void* ylo;            // H32: the lo plane [R,384] fp16
    int yfmt;             // 0 float32, 1 H16, 2 H24 (DG_DTYPE_F32_H24: top 24 bits of every float32), 3 H32 (hi + lo fp16 planes)
    int64_t R;
    Epi ep;
};

// MODE: the consumers' epilogue.  0 generic (bias, optional ReLU, optional mask in, bits out); 1 the forward of fc1 -- bias + ReLU +
// bits out, no mask in; 2 its twin in the backward -- mask in only (dh = (dz W2) * m).  The generic form spends ~8 vector
// instructions per element on selects whose conditions are launch constants (and a wait state per v_cmp -> v_cndmask pair); the
// two hot forms need 3 and 2.
// NP: MFMA products per k-step.  3: w_hi.x_hi + w_lo.x_hi + w_hi.x_lo (float32 class).  2: without w_hi.x_lo -- the activation rows
// enter as ONE fp16 plane (2^-12 rounding per element, independent from element to element) -- for results that leave as an fp16
// plane anyway (FMT 1) in the BACKWARD (dh = (dz W2) * m and its second-order twin), which only travel through linear maps
// (DESIGN 3.16).  The weights keep both planes: their rounding would be the SAME for every row (measured with a single product:
// 1.15e-3 on the chembl_b4 golden, outside the bar; DG_DH_PRODUCTS=1 selects it for the record).
template <int FMT, int MODE, int NP = 3>
__global__ __launch_bounds__(64 * (kCons + kProd)) void row_gemm_n384_kernel(const Prob p0, const Prob p1, const int nb0) {
    constexpr bool H16 = FMT == 1 || FMT == 3, TWO = FMT == 3, H24 = FMT == 2;      // TWO: the lo plane of the split leaves too (H32)
    static_assert(NP == 3 || ((NP == 1 || NP == 2) && FMT == 1 && MODE == 2), "reduced products: fp16-plane results of the backward only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const tab = reinterpret_cast<float*>(smem + kOffTab);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool second = static_cast<int>(blockIdx.x) >= nb0;      // uniform
    const float* __restrict__ const a = second ? p1.a : p0.a;
    const f16x8* __restrict__ const packed = second ? p1.packed : p0.packed;
    float* __restrict__ const y = second ? p1.y : p0.y;
    float* __restrict__ const yscale = second ? p1.yscale : p0.yscale;
    void* __restrict__ const ylo = second ? p1.ylo : p0.ylo;
    const int64_t R = second ? p1.R : p0.R;
    const Epi ep = second ? p1.ep : p0.ep;
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
    const int TP = (T + kDepth - 1) / kDepth * kDepth;

    if (w >= kCons) {
        // ------------------------------------------------------------------------------------------ producers
        __builtin_amdgcn_s_setprio(3);
        const int pt = threadIdx.x - 64 * kCons;
        const int hw = pt >> 5, l32 = pt & 31;
        // tables for the consumers' epilogue (once)
        {
            const float* inv_cs = reinterpret_cast<const float*>(packed + static_cast<size_t>(12) * 8 * 2 * 64);
            for (int i = pt; i < 384; i += 64 * kProd) {
                tab[i] = inv_cs[i];
                tab[384 + i] = ep.bias ? ep.bias[i] : 0.f;
            }
        }
        float4 pf[kDepth][4];
        u32x2 mk[kDepth];      // this thread's two mask words of the stage (words 2 pt, 2 pt + 1 of its 512)
        const unsigned voff = static_cast<unsigned>(hw) * 512u + static_cast<unsigned>(l32) * 16u;
        auto fetch = [&](float4 (&set)[4], u32x2& mword, int t) {
            if (t > T - 1) t = T - 1;
            // (a null mask: a zero-sized range, every word reads as 0 and is replaced by all-ones in split())
            const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<unsigned*>(ep.mask_bits ? ep.mask_bits + stage_of(t) * 512 : reinterpret_cast<const unsigned*>(a)), 0,
                ep.mask_bits ? 2048 : 0, 0x00020000);
            mword = __builtin_amdgcn_raw_buffer_load_b64(rmask, static_cast<unsigned>(pt) * 8u, 0, 0);
            const int64_t r0 = stage_of(t) * kSR;
            const int64_t left = (R - r0) * 512;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a) + r0 * 128, 0, static_cast<int>(left < (1 << 30) ? left : (1 << 30)), 0x00020000);
            if (N3_DBG & 8) return;
#pragma unroll
            for (int i = 0; i < 4; ++i)      // rows hw + 8 i
                set[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, i * 4096, 0));
        };
        // float4 column c = l32 covers k = 4c .. 4c + 3: block b = c >> 1 (k-step c >> 3, quarter (c >> 1) & 3), half c & 1;
        // row r of block b sits at position r ^ (b & 7)
        const int blk = l32 >> 1;
        const unsigned wbase = static_cast<unsigned>(blk * 512 + (l32 & 1) * 8);
        auto split = [&](float4 (&set)[4], const u32x2& mword, int t) {      // stage t -> planes[t & 1]
            char* const pl = smem + (t & 1) * kStage;
            *reinterpret_cast<u32x2*>(smem + kOffBitsIn + (t & 1) * 2048 + pt * 8) = ep.mask_bits ? mword : u32x2{0xFFFFFFFFu, 0xFFFFFFFFu};
            unsigned m[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4& v = set[i];
                float t0, u;
                // (volatile: the first use of the loads stays behind the previous iteration's barrier)
                asm volatile("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(t0) : "v"(v.x), "v"(v.y), "v"(v.z));
                asm("v_max_f32_e64 %0, |%1|, %2" : "=v"(u) : "v"(v.w), "v"(t0));
                m[i] = __float_as_uint(u);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) m[i] = umax_dpp<0xB1>(m[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) m[i] = umax_dpp<0x4E>(m[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) m[i] = umax_dpp<0x141>(m[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) m[i] = umax_dpp<0x140>(m[i]);
            float sc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const auto r = __builtin_amdgcn_permlane16_swap(m[i], m[i], false, false);
                const unsigned xm = r[0] > r[1] ? r[0] : r[1];
                unsigned e = xm >> 23;
                e = e < 15u ? 15u : e;
                m[i] = e;
                sc[i] = __uint_as_float((268u - e) << 23);      // 2^(14 - (e - 127)): the row maximum lands in [2^14, 2^15)
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4& v = set[i];
                f32x2 xa = f32x2{v.x, v.y} * sc[i], xb = f32x2{v.z, v.w} * sc[i];
                const f16x2 ha = __builtin_convertvector(xa, f16x2), hb = __builtin_convertvector(xb, f16x2);
                xa -= __builtin_convertvector(ha, f32x2);
                xb -= __builtin_convertvector(hb, f32x2);
                const f16x2 la = __builtin_convertvector(xa, f16x2), lb = __builtin_convertvector(xb, f16x2);
                const int row = hw + 8 * i;
                const unsigned off = wbase + static_cast<unsigned>((row ^ (blk & 7)) * 16);
                *reinterpret_cast<u32x2*>(pl + off) = u32x2{__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
                if (NP == 3)
                    *reinterpret_cast<u32x2*>(pl + kPlane + off) = u32x2{__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
            }
            if (l32 == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<float*>(pl + 2 * kPlane + (hw + 8 * i) * 4) = __uint_as_float((m[i] - 14u) << 23);
            }
        };
        // finished output tile of stage t: LDS -> HBM, whole rows.  Thread (hw, l32) moves, for row hw + 8 k and piece q,
        // the 16 bytes at LDS slot 32 q + l32; their channel slot is that position ^ (row & 7) = ^ hw.
        const unsigned ooff = static_cast<unsigned>(hw) * 1536u + static_cast<unsigned>(l32) * 16u;
        const unsigned goff = static_cast<unsigned>(hw) * 1536u + static_cast<unsigned>(l32 ^ hw) * 16u;
        auto store_out = [&](int t) {
            if (N3_DBG & 4) return;
            const bool ok = t >= 0 && t < T;
            const int tc = ok ? t : 0;
            const int64_t r0 = stage_of(tc) * kSR;
            const char* ot = smem + kOffOut + (tc & 1) * kOut;
            if (H16) {
                // one fp16 plane + one inverse scale per row.  Eight lanes per row, six pairs of 16-byte tile slots (= 8
                // consecutive channels = one 16-byte store) per lane: thread pt -> row 2 (pt >> 4) + ((pt >> 3) & 1), pairs
                // (pt & 7) + 8 m.  A pair's lower-channel half sits at tile position 2 j + (row & 1): the even row of a
                // 16-lane group reads even positions while the odd row reads odd ones -- conflict-free ds_read_b128.
                const int64_t left = R - r0;
                const int rows = ok ? static_cast<int>(left < kSR ? left : kSR) : 0;      // 0: every store is dropped
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    reinterpret_cast<_Float16*>(y) + r0 * 384, 0, rows * 768, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(yscale + r0, 0, rows * 4, 0x00020000);
                const __amdgpu_buffer_rsrc_t rlo = __builtin_amdgcn_make_buffer_rsrc(
                    TWO ? reinterpret_cast<_Float16*>(ylo) + r0 * 384 : reinterpret_cast<_Float16*>(y), 0, TWO ? rows * 768 : 0, 0x00020000);
                const int g = pt >> 4, sub = (pt >> 3) & 1, l8 = pt & 7, row = 2 * g + sub;
                const unsigned lo_off = static_cast<unsigned>(row * 1536 + l8 * 32 + sub * 16);
                const unsigned hi_off = static_cast<unsigned>(row * 1536 + l8 * 32 + (1 - sub) * 16);
                const unsigned hoff = static_cast<unsigned>(row * 768 + ((l8 ^ (g & 3)) * 16));
                const unsigned soff = l8 == 0 ? static_cast<unsigned>(row) * 4u : 0x7FFFFFF0u;      // one lane per row
                float4 vl[6], vh[6];
#pragma unroll
                for (int mm = 0; mm < 6; ++mm) {
                    vl[mm] = *reinterpret_cast<const float4*>(ot + lo_off + mm * 256);
                    vh[mm] = *reinterpret_cast<const float4*>(ot + hi_off + mm * 256);
                }
                float mx = 0.f;
#pragma unroll
                for (int mm = 0; mm < 6; ++mm) {
                    float a0, a1;
                    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(a0) : "v"(vl[mm].x), "v"(vl[mm].y), "v"(vl[mm].z));
                    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(a1) : "v"(vl[mm].w), "v"(vh[mm].x), "v"(vh[mm].y));
                    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(a0) : "v"(a0), "v"(vh[mm].z), "v"(vh[mm].w));
                    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(a0), "v"(a1));
                }
                unsigned m = __float_as_uint(mx);
                m = umax_dpp<0xB1>(m);
                m = umax_dpp<0x4E>(m);
                m = umax_dpp<0x141>(m);      // the 8 lanes of the row
                unsigned e = m >> 23;
                e = e < 15u ? 15u : e;
                const float sc = __uint_as_float((268u - e) << 23);      // the row maximum lands in [2^14, 2^15)
                // All six packed vectors first, in registers of their own, then the stores back to back: a 16-byte store reads
                // its data registers over many cycles, and a VALU write into them right behind the store (hipcc reuses the
                // registers of the previous vector; it only pads stores WITHOUT an SGPR offset) corrupted the second dword of
                // lanes 12..15 of every DPP row (found by tests/test_hip_kernels.py::test_hidden_fp16_plane_*).
                u32x4 hq[6], lq[TWO ? 6 : 1];
#pragma unroll
                for (int mm = 0; mm < 6; ++mm) {
                    f32x2 x0 = f32x2{vl[mm].x, vl[mm].y} * sc, x1 = f32x2{vl[mm].z, vl[mm].w} * sc;
                    f32x2 x2 = f32x2{vh[mm].x, vh[mm].y} * sc, x3 = f32x2{vh[mm].z, vh[mm].w} * sc;
                    const f16x2 h0 = __builtin_convertvector(x0, f16x2), h1 = __builtin_convertvector(x1, f16x2);
                    const f16x2 h2 = __builtin_convertvector(x2, f16x2), h3 = __builtin_convertvector(x3, f16x2);
                    hq[mm] = u32x4{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1), __builtin_bit_cast(unsigned, h2),
                                   __builtin_bit_cast(unsigned, h3)};
                    if (TWO) {      // lo = fp16(x s - hi): the residual is exact in float32 (the split of row_gemm.hip, done once here);
                        // one v_fma_mix per element: fma(hi as fp16 operand, -1, x s) rounded to fp16 in the same instruction
                        lq[mm] = u32x4{lo_pair(__builtin_bit_cast(unsigned, h0), x0[0], x0[1]), lo_pair(__builtin_bit_cast(unsigned, h1), x1[0], x1[1]),
                                       lo_pair(__builtin_bit_cast(unsigned, h2), x2[0], x2[1]), lo_pair(__builtin_bit_cast(unsigned, h3), x3[0], x3[1])};
                    }
                }
                unsigned inv = (e - 14u) << 23, so = soff;
                asm volatile("" : "+v"(hq[0]), "+v"(hq[1]), "+v"(hq[2]), "+v"(hq[3]), "+v"(hq[4]), "+v"(hq[5]), "+v"(inv), "+v"(so));
                if (TWO) asm volatile("" : "+v"(lq[0]), "+v"(lq[1]), "+v"(lq[2]), "+v"(lq[3]), "+v"(lq[4]), "+v"(lq[5]));
#pragma unroll
                for (int mm = 0; mm < 6; ++mm) __builtin_amdgcn_raw_buffer_store_b128(hq[mm], rsrc, hoff, mm * 128, 0);
                if (TWO) {
#pragma unroll
                    for (int mm = 0; mm < 6; ++mm) __builtin_amdgcn_raw_buffer_store_b128(lq[mm], rlo, hoff, mm * 128, 0);
                }
                __builtin_amdgcn_raw_buffer_store_b32(inv, rsc, so, 0, 0);
                asm volatile("s_nop 15" ::: "memory");
            } else if (H24) {
                // three bytes per element: the float32 path with 12-byte stores (all packed vectors first: see the note above)
                const int64_t left = (R - r0) * 1152;
                const int bytes = ok ? static_cast<int>(left < kSR * 1152 ? left : kSR * 1152) : 0;
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(y) + r0 * 1152, 0, bytes, 0x00020000);
                const unsigned g24 = static_cast<unsigned>(hw) * 1152u + static_cast<unsigned>(l32 ^ hw) * 12u;
                dg_u32x3 pk[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const int k = i / 3, qq = i % 3;
                    const float4 v = *reinterpret_cast<const float4*>(ot + ooff + k * 8 * 1536 + qq * 512);
                    pk[i] = pack_f24x4(v.x, v.y, v.z, v.w);
                }
                asm volatile("" : "+v"(pk[0]), "+v"(pk[1]), "+v"(pk[2]), "+v"(pk[3]), "+v"(pk[4]), "+v"(pk[5]), "+v"(pk[6]), "+v"(pk[7]),
                             "+v"(pk[8]), "+v"(pk[9]), "+v"(pk[10]), "+v"(pk[11]));
#pragma unroll
                for (int i = 0; i < 12; ++i) __builtin_amdgcn_raw_buffer_store_b96(pk[i], rsrc, g24, (i / 3) * 8 * 1152 + (i % 3) * 384, 0);
                asm volatile("s_nop 15" ::: "memory");
            } else {
                const int64_t left = (R - r0) * 1536;
                const int bytes = ok ? static_cast<int>(left < kSR * 1536 ? left : kSR * 1536) : 0;      // 0: every store is dropped
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(y + r0 * 384, 0, bytes, 0x00020000);
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const int k = i / 3, qq = i % 3;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(ot + ooff + k * 8 * 1536 + qq * 512);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, goff, k * 8 * 1536 + qq * 512, 0);
                }
            }
            // the stage's ReLU bit words (written by the consumers next to the tile)
            const __amdgpu_buffer_rsrc_t rbits = __builtin_amdgcn_make_buffer_rsrc(
                ep.relu_bits ? ep.relu_bits + stage_of(tc) * 512 : reinterpret_cast<unsigned*>(y), 0, (ok && ep.relu_bits) ? 2048 : 0,
                0x00020000);
            const u32x2 bw = *reinterpret_cast<const u32x2*>(smem + kOffBitsOut + (tc & 1) * 2048 + pt * 8);
            __builtin_amdgcn_raw_buffer_store_b64(bw, rbits, static_cast<unsigned>(pt) * 8u, 0, 0);
        };
        fetch(pf[0], mk[0], 0);
        fetch(pf[1], mk[1], 1);
        fetch(pf[2], mk[2], 2);
        split(pf[0], mk[0], 0);
        fetch(pf[0], mk[0], 3);
        __syncthreads();
        for (int t = 0; t < TP; t += 3) {
            // iteration t: the consumers are on stage t; planes of stage t + 1 are written, the output tile of stage t - 1 leaves
            split(pf[1], mk[1], t + 1);
            fetch(pf[1], mk[1], t + 4);
            store_out(t - 1);
            __syncthreads();
            split(pf[2], mk[2], t + 2);
            fetch(pf[2], mk[2], t + 5);
            store_out(t);
            __syncthreads();
            split(pf[0], mk[0], t + 3);
            fetch(pf[0], mk[0], t + 6);
            store_out(t + 1);
            __syncthreads();
        }
        store_out(TP - 1);
        return;
    }

    // ---------------------------------------------------------------------------------------------- consumers
    const int n = lane & 15, kq = lane >> 4;
    // weight fragments from the packed operand (32-column slabs x 16-deep k-steps, lane = (column, k half)): channel
    // 48 w + 16 cb + n, k = 32 ks + 8 kq .. + 7  ->  slab t, k-step 2 ks + (kq >> 1), lane (kq & 1) * 32 + column
    f16x8 wf[3][4][NP == 1 ? 1 : 2];
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) {
        const int ch = 48 * w + 16 * cb + n;
        const int tslab = ch >> 5, col = ch & 31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int p = 0; p < (NP == 1 ? 1 : 2); ++p)
                wf[cb][ks][p] = packed[(static_cast<size_t>(tslab * 8 + 2 * ks + (kq >> 1)) * 2 + p) * 64 + (kq & 1) * 32 + col];
    }
    // activation fragment of lane (row n, quarter kq) in k-step ks: block 4 ks + kq, position (16 rb + n) ^ ((4 ks + kq) & 7)
    const unsigned xo_e = static_cast<unsigned>(kq * 512 + ((n ^ kq) * 16));            // even k-steps
    const unsigned xo_o = static_cast<unsigned>(kq * 512 + ((n ^ (4 + kq)) * 16));      // odd k-steps
    const unsigned relu_sel = ep.relu ? 0xFFFFFFFFu : 0u;
    __builtin_amdgcn_s_waitcnt(0x0F70);      // weight fragments are in registers
    __syncthreads();                         // stage 0 is in planes[0], the tables are written
    for (int t = 0; t < TP; ++t) {
        if (t < T) {
            const char* pl = smem + (t & 1) * kStage;
            const unsigned bits = *reinterpret_cast<const unsigned*>(smem + kOffBitsIn + (t & 1) * 2048 + (w * 64 + lane) * 4);
            f32x4 acc[2][3];
            // Fragment reads run one k-step AHEAD, per row block: the fragments of (k-step ks + 1, row block rb) are requested
            // right behind the MFMAs of (ks, rb) -- into the same registers, the MFMAs have read them at issue -- and arrive
            // while the other row block's MFMAs run.  (hipcc's own order was 4 reads, wait, 18 MFMAs, 4 reads, wait ...: every
            // k-step exposed one LDS round trip with the matrix pipe idle; N3_SCHED=0 keeps that order for A/B builds.)
            f16x8 xh[2], xl[2];
            auto read_frags = [&](int ks, int rb) {
                const char* p0 = pl + ks * 2048 + rb * 256 + ((ks & 1) ? xo_o : xo_e);
                xh[rb] = *reinterpret_cast<const f16x8*>(p0);
                if (NP == 3) xl[rb] = *reinterpret_cast<const f16x8*>(p0 + kPlane);
            };
            read_frags(0, 0);
            read_frags(0, 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    // w_lo.x_hi, w_hi.x_lo, w_hi.x_hi: smallest terms first; three accumulators in rotation
#pragma unroll
                    for (int term = (NP == 1 ? 2 : 0); term < 3; ++term)
#pragma unroll
                        for (int cb = 0; cb < 3; ++cb) {
                            if (NP != 3 && term == 1) continue;
                            constexpr int LO = NP == 1 ? 0 : 1;      // (index of the weights' lo plane; never read when NP == 1)
                            constexpr int FIRST = NP == 1 ? 2 : 0;
                            if (N3_DBG & 1) {
                                if (ks == 0 && term == FIRST) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                                acc[rb][cb][0] += static_cast<float>(xh[rb][0]) * static_cast<float>(wf[cb][ks][0][0]);
                            } else if (ks == 0 && term == FIRST) mfma16_first(acc[rb][cb], wf[cb][ks][term == 0 ? LO : 0], xh[rb]);
                            else mfma16(acc[rb][cb], wf[cb][ks][term == 0 ? LO : 0], term == 1 ? xl[rb] : xh[rb]);
                        }
#if N3_SCHED
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks < 3) read_frags(ks + 1, rb);
                    __builtin_amdgcn_sched_barrier(0);
#else
                    if (ks < 3 && rb == 1) {
                        read_frags(ks + 1, 0);
                        read_frags(ks + 1, 1);
                    }
#endif
                }
            }
            mfma_results_ready();
            // epilogue: lane (row n of block rb, channel group kq) holds channels 48 w + 16 cb + 4 kq + i of its row
            char* ot = smem + kOffOut + (t & 1) * kOut;
            unsigned newbits = 0;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int row = 16 * rb + n;
                const float rs = *reinterpret_cast<const float*>(pl + 2 * kPlane + row * 4);
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) {
                    const int c0 = 48 * w + 16 * cb + 4 * kq;
                    const float4 cs = ld4(tab + c0), bs = ld4(tab + 384 + c0);
                    if (N3_DBG & 2) {
                        const int slot = (c0 >> 2) ^ (row & 7);
                        *reinterpret_cast<float4*>(ot + row * 1536 + slot * 16) =
                            make_float4(acc[rb][cb][0], acc[rb][cb][1], acc[rb][cb][2], acc[rb][cb][3]);
                        continue;
                    }
                    float v[4] = {fmaf(acc[rb][cb][0], rs * cs.x, bs.x), fmaf(acc[rb][cb][1], rs * cs.y, bs.y),
                                  fmaf(acc[rb][cb][2], rs * cs.z, bs.z), fmaf(acc[rb][cb][3], rs * cs.w, bs.w)};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int bit = (rb * 3 + cb) * 4 + i;
                        if (MODE == 1) {
                            newbits |= v[i] > 0.f ? (1u << bit) : 0u;
                            v[i] = fmaxf(v[i], 0.f);
                        } else if (MODE == 2) {
                            // bit -> 0 / ~0 (one v_bfe_i32), and: no compare, no select
                            const int keep = __builtin_amdgcn_sbfe(static_cast<int>(bits), bit, 1);
                            v[i] = __uint_as_float(__float_as_uint(v[i]) & static_cast<unsigned>(keep));
                        } else {
                            newbits |= (v[i] > 0.f ? 1u : 0u) << bit;
                            v[i] = __uint_as_float((__float_as_uint(fmaxf(v[i], 0.f)) & relu_sel) | (__float_as_uint(v[i]) & ~relu_sel));
                            v[i] = (bits >> bit) & 1u ? v[i] : 0.f;
                        }
                    }
                    const int slot = (c0 >> 2) ^ (row & 7);
                    *reinterpret_cast<float4*>(ot + row * 1536 + slot * 16) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
            *reinterpret_cast<unsigned*>(smem + kOffBitsOut + (t & 1) * 2048 + (w * 64 + lane) * 4) = newbits;
        }
        __syncthreads();
    }
}

}  // namespace

size_t row_gemm_n384_mask_words(int64_t R) { return static_cast<size_t>((R + kSR - 1) / kSR) * kCons * 64; }

namespace {
struct Pending {
    bool valid = false;
    Prob p;
};
thread_local Pending g_rider;

// the epilogue form a problem can take (see MODE); a launch of two problems needs the same form and output storage for both
int mode_of(const Prob& p) {
    if (p.ep.relu && !p.ep.mask_bits) return 1;
    if (!p.ep.relu && p.ep.mask_bits && !p.ep.relu_bits && !p.ep.bias) return 2;
    return 0;
}
bool same_kernel(const Prob& a, const Prob& b) { return a.yfmt == b.yfmt && mode_of(a) == mode_of(b); }

int launch(const Prob& p0, const Prob* p1, hipStream_t stream) {
    const int64_t st0 = (p0.R + kSR - 1) / kSR, st1 = p1 ? (p1->R + kSR - 1) / kSR : 0;
    int nb0, nb1;
    pair_split(st0, st1, 256, &nb0, &nb1);
    const int mode = mode_of(p0);
#define DG_N384_LAUNCH(H16_, MODE_)                                                                                     \
    {                                                                                                                    \
        DG_OPT_IN_LDS((&row_gemm_n384_kernel<H16_, MODE_>), kLds);                                                       \
        hipLaunchKernelGGL((row_gemm_n384_kernel<H16_, MODE_>), dim3(nb0 + nb1), dim3(64 * (kCons + kProd)), kLds, stream, p0, \
                           p1 ? *p1 : p0, nb0);                                                                          \
    }
    // DG_DH_PRODUCTS=3: full three-product arithmetic for the backward's fp16-plane results too, =1: a single product (A/B, tests)
    const int dh_products = 2;
    if (p0.yfmt == 1) {
        if (mode == 1) DG_N384_LAUNCH(1, 1)
        else if (mode == 2 && dh_products == 2) {
            DG_OPT_IN_LDS((&row_gemm_n384_kernel<1, 2, 2>), kLds);
            hipLaunchKernelGGL((row_gemm_n384_kernel<1, 2, 2>), dim3(nb0 + nb1), dim3(64 * (kCons + kProd)), kLds, stream, p0,
                               p1 ? *p1 : p0, nb0);
        } else if (mode == 2 && dh_products == 1) {
            DG_OPT_IN_LDS((&row_gemm_n384_kernel<1, 2, 1>), kLds);
            hipLaunchKernelGGL((row_gemm_n384_kernel<1, 2, 1>), dim3(nb0 + nb1), dim3(64 * (kCons + kProd)), kLds, stream, p0,
                               p1 ? *p1 : p0, nb0);
        } else if (mode == 2) DG_N384_LAUNCH(1, 2)
        else DG_N384_LAUNCH(1, 0)
    } else if (p0.yfmt == 3) {
        if (mode == 1) DG_N384_LAUNCH(3, 1)
        else if (mode == 2) DG_N384_LAUNCH(3, 2)
        else DG_N384_LAUNCH(3, 0)
    } else if (p0.yfmt == 2) {
        if (mode == 1) DG_N384_LAUNCH(2, 1)
        else if (mode == 2) DG_N384_LAUNCH(2, 2)
        else DG_N384_LAUNCH(2, 0)
    } else {
        if (mode == 1) DG_N384_LAUNCH(0, 1)
        else if (mode == 2) DG_N384_LAUNCH(0, 2)
        else DG_N384_LAUNCH(0, 0)
    }
#undef DG_N384_LAUNCH
    return 0;
}
}  // namespace

int flush_row_gemm_n384(hipStream_t stream) {
    if (!g_rider.valid) return 0;
    g_rider.valid = false;
    return launch(g_rider.p, nullptr, stream);
}

int launch_row_gemm_n384(const float* a, const void* packed, void* y, float* yscale, int64_t R, const float* bias, int relu,
                         unsigned* relu_bits, const unsigned* mask_bits, hipStream_t stream, int yfmt, void* ylo) {
    if ((yfmt == 1 || yfmt == 3) != (yscale != nullptr)) return fail(DG_E_ARG, "row_gemm_n384: row scales go with the fp16 planes (yfmt 1, 3)");
    if ((yfmt == 3) != (ylo != nullptr)) return fail(DG_E_ARG, "row_gemm_n384: the lo plane goes with yfmt 3");
    const Prob p{a, static_cast<const f16x8*>(packed), static_cast<float*>(y), yscale, ylo, yfmt, R,
                 Epi{bias, mask_bits, relu_bits, relu, take_direction(R)}};
    if (pair_mode() && !g_rider.valid && R <= kRiderMaxRows) {      // waits for the next launch of this kernel
        g_rider.valid = true;
        g_rider.p = p;
        return 0;
    }
    if (g_rider.valid) {
        g_rider.valid = false;
        if (same_kernel(p, g_rider.p)) return launch(p, &g_rider.p, stream);
        if (int st = launch(g_rider.p, nullptr, stream)) return st;      // another output format / epilogue: on its own, first
    }
    return launch(p, nullptr, stream);
}

}  // namespace dg
