// Fused float32 feed-forward forward:  y = LayerNorm(x + fc2(relu(fc1 x)))  with the [R,384] hidden tensor ON CHIP
// (reference src/model/layers.py:50-53 `MLP.forward`, :191-192 the residual + ln5 / ln6 of `Encoder_Block`).
//
// The two-launch form (row_gemm_n384.hip writes h, row_gemm_k384.hip reads it back) moves 5.2 KB per row, 3.1 KB of it the
// float32-class hidden tensor.  Here h never leaves the CU: per row the launch reads x (512 B) and writes y, the pre-LayerNorm
// sum, and -- for the backward's weight gradient dW2 = dz^T h only -- the hi fp16 plane of h with one row scale (768 + 4 B,
// the DG_DTYPE_F32_H16 layout) and the ReLU bit mask in row_gemm_n384.hip's layout (the backward's dh = (dz W2) * m launch
// reads it unchanged).
//
// What made the fusion impossible on the producer / consumer kernels is the weights: W1 and W2 as fp16 hi + lo planes are
// 2 x 192 KB, more than the LDS and 3/4 of the register file.  So the roles are swapped: the ACTIVATIONS are stationary in
// registers and the WEIGHTS stream.
//   * a workgroup = 8 waves walks 128-row tiles (round-robin); wave w owns 16 rows of the tile for the whole pass:
//     x as MFMA B fragments (hi + lo fp16 under one power-of-two row scale, 32 VGPRs), the residual copy of x (32), the
//     hidden row block h [16 x 384] (96 VGPRs: float32 while fc1 runs, then hi + lo planes under ONE row scale -- the row
//     maximum is known only after the last fc1 block, which is why a wave keeps whole rows), the fc2 results (32).
//   * both weights, pre-packed into fragment order (dg_ffn_f32_pack), travel L2 -> LDS by LDS-DMA (global_load_lds_dwordx4,
//     no VGPRs) in 20 chunks per pass -- 12 x 16 KB of W1 (two 16-channel blocks each), 8 x 24 KB of W2 (one 16-channel
//     output block each) -- through three 24 KB buffers, two chunks ahead; one s_barrier per chunk.  Every wave reads every
//     fragment (conflict-free lane-linear ds_read_b128): 384 KB of LDS reads per wave and pass, the bound of the kernel
//     (LDS 24.6 k cycles per 128 rows against 18.4 k of MFMA issue per SIMD).  L2 -> LDS traffic is 1 KB per row.
//   * swapped products on v_mfma_f32_16x16x32_f16 (weights = A, activation rows = B): a lane ends with 4 consecutive
//     channels of ONE row.  The channel order inside a pair of 16-channel blocks is chosen at pack time so that the lane's
//     two results are 8 CONSECUTIVE channels: fc1's output is already fc2's B fragment (no transpose, no LDS round trip),
//     and every store is 16 / 32 contiguous bytes per lane.
//   * three products per k-step (w_lo.x_hi + w_hi.x_lo + w_hi.x_hi) on independent in-place chains, float32 class --
//     the arithmetic of row_gemm_n384.hip / row_gemm_k384.hip's H32 instances (one row scale over the whole contraction).
//   * the next tile's x rows arrive by LDS-DMA into a wave-private 8 KB region during the pass (fragment order: the
//     permutation is applied on the SOURCE address); waits are counted s_waitcnt vmcnt(N) with N derived from the fixed
//     issue order below (a store's acknowledgement is never waited for inside a pass).
#include "common.h"

#include "pair.h"
#include "traversal.h"

#ifndef FF_DBG
#define FF_DBG 0      // ablation builds (scripts/build_variant.sh; results wrong, timing only): 1 no MFMAs, 2 no global stores, 4 no
#endif                // weight-fragment reads, 8 no weight DMA, 16 no fc1 epilogue arithmetic, 32 no h split, 64 no x split, 128 no
                      // LayerNorm statistics, 256 no workgroup barriers in the pass, 512 no counted waits, 1024 / 2048 / 4096 no stores of the pre-LayerNorm sum / the h plane / y
                      // (build those with -DFF_SAFE_WAIT=1: the counted waits assume every store)
#ifndef FF_SAFE_WAIT
#define FF_SAFE_WAIT 0      // 1: every chunk wait is vmcnt(0) (debug builds: rules the counted waits out)
#endif
namespace dg {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifndef FF_AUX
#define FF_AUX 0      // cache-policy bits of the result stores (gfx950 buffer stores: 1 sc0, 2 nt, 16 sc1)
#endif
#define FF_STORE128(...) ff_store128(__VA_ARGS__)
#define FF_STORE32(...) ff_store32(__VA_ARGS__)

__device__ __forceinline__ void ff_store128(u32x4 data, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, int imm, int) {
    if (!(FF_DBG & 2)) __builtin_amdgcn_raw_buffer_store_b128(data, rsrc, voff, imm, FF_AUX);
}
__device__ __forceinline__ void ff_store32(unsigned data, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, int imm, int) {
    if (!(FF_DBG & 2)) __builtin_amdgcn_raw_buffer_store_b32(data, rsrc, voff, imm, FF_AUX);
}

#ifndef FF_PROF
#define FF_PROF 0      // developer builds: s_memtime around the sections of every iteration (1), around the y / pre-LN stores only (2), around
#endif                 // the h-plane stores only (3); totals of workgroup 0 per wave into mean[300000 ..] (scripts/ffn_f32_prof.py)
__device__ __forceinline__ unsigned long long ff_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
constexpr int kTile = 128;                       // rows per workgroup pass (16 per wave)
constexpr int kWaves = 8;
constexpr int kW1Chunk = 16 * 1024;              // one pair of 16-channel fc1 blocks: [block 2][k-step 4][plane 2] fragments of 1 KB
constexpr int kW2Chunk = 24 * 1024;              // one 16-channel fc2 output block: [k-step 12][plane 2] fragments
constexpr int kW1Bytes = 12 * kW1Chunk, kW2Bytes = 8 * kW2Chunk;
constexpr int kPackedBytes = kW1Bytes + kW2Bytes + (384 + 128) * 4;      // + inverse column scales of W1 [384], W2 [128]
constexpr int kOffW = 0;                         // three chunk buffers
constexpr int kOffX = 3 * kW2Chunk;              // [wave 8][fragment 8][lane 64][16 B]: the next tile's x rows
constexpr int kOffTab = kOffX + kWaves * 8192;   // cs1 [384], b1 [384], cs2 [128], b2 [128], gamma [128], beta [128]
constexpr int kOffBits = kOffTab + 1408 * 4;     // ReLU mask words of the pass: [stage 4][wave' 8][lane' 64]
constexpr int kLds = kOffBits + 4 * 512 * 4;
static_assert(kLds <= 160 * 1024, "LDS budget");

// acc += A . B on v_mfma_f32_16x16x32_f16, ALWAYS in place; the first MFMA of a chain takes the constant 0; a vector read of
// a result is fenced: see row_gemm_k384.hip (a renamed destination one slot behind its producer read a partly written
// accumulator on gfx950).
__device__ __forceinline__ void mfma16(f32x4& acc, const f16x8& a, const f16x8& b) {
    if (FF_DBG & 1) {
        asm volatile("" : "+v"(acc) : "v"(a), "v"(b));
        return;
    }
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_first(f32x4& acc, const f16x8& a, const f16x8& b) {
    if (FF_DBG & 1) {
        acc = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("" : "+v"(acc) : "v"(a), "v"(b));
        return;
    }
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
__device__ __forceinline__ unsigned scale_exponent(float absmax) {      // biased exponent, clamped away from 0
    const unsigned e = __float_as_uint(absmax) >> 23;
    return e < 15u ? 15u : e;
}
__device__ __forceinline__ float scale_of(unsigned e) { return __uint_as_float((268u - e) << 23); }      // 2^(14 - (e - 127))
__device__ __forceinline__ float inv_scale_of(unsigned e) { return __uint_as_float((e - 14u) << 23); }

// hi / lo fp16 planes of eight values under the scale sc: (x sc) = hi + lo up to 2^-22 of the row maximum
__device__ __forceinline__ void split8(const float (&v)[8], float sc, f16x8& hi, f16x8& lo) {
    u32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 s = f32x2{v[2 * i], v[2 * i + 1]} * sc;
        const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(s, f16x2));
        h[i] = hp;
        l[i] = lo_pair(hp, s[0], s[1]);
    }
    hi = __builtin_bit_cast(f16x8, h);
    lo = __builtin_bit_cast(f16x8, l);
}

// LDS-DMA, 16 bytes per lane, SGPR base + 32-bit lane offset + immediate.  The instruction offset applies to BOTH addresses:
// LDS[lds_base + IMM + lane * 16 ..] = *(sbase + voff + IMM).
// (common.h's dma16_async takes a 64-bit address per lane: twenty chunk addresses per pass are loop invariants that hipcc hoists
// and spills, and a scratch reload in front of a DMA waits vmcnt(0) -- the whole pipeline drained at every issue.)
template <int IMM>
__device__ __forceinline__ void dma16_s(const void* sbase, unsigned voff, unsigned lds_base) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"      // (an SGPR base fresh from v_readfirstlane)
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:%4\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_base), "n"(IMM)
        : "memory");
}

// workgroup barrier that leaves VMEM in flight: __syncthreads()'s fence would wait vmcnt(0) for pending stores; the LDS side
// (fragment reads, ds_or) is drained explicitly
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void pass_barrier() {
    if (FF_DBG & 256) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else wg_barrier();
}
// ablation builds: a "split" that only converts
__device__ __forceinline__ void fake_split8(const float (&v)[8], f16x8& hi, f16x8& lo) {
    u32x4 h;
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2 * i], v[2 * i + 1]}, f16x2));
    hi = __builtin_bit_cast(f16x8, h);
    lo = hi;
}

// s_waitcnt vmcnt(n) with n a constant after unrolling (the switch folds)
__device__ __forceinline__ void vm_wait(int n) {
    if (FF_DBG & 512) return;
#define DG_VMW(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
    switch (n) {
        DG_VMW(0) DG_VMW(1) DG_VMW(2) DG_VMW(3) DG_VMW(4) DG_VMW(5) DG_VMW(6) DG_VMW(7) DG_VMW(8) DG_VMW(9) DG_VMW(10) DG_VMW(11)
        DG_VMW(12) DG_VMW(13) DG_VMW(14) DG_VMW(15) DG_VMW(16) DG_VMW(17) DG_VMW(18) DG_VMW(19) DG_VMW(20) DG_VMW(21) DG_VMW(22)
        DG_VMW(23) DG_VMW(24) DG_VMW(25) DG_VMW(26) DG_VMW(27) DG_VMW(28) DG_VMW(29) DG_VMW(30) DG_VMW(31) DG_VMW(32)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef DG_VMW
}

struct FProb {
    const float* x;           // [R,128]
    const char* packed;       // dg_ffn_f32_pack
    const float* b1;          // [384]
    const float* b2;          // [128]
    const float* gamma;       // [128]
    const float* beta;
    float* y;                 // [R,128]
    _Float16* h;              // [R,384] fp16: the hi plane of h under one row scale (DG_DTYPE_F32_H16 buffer), KEEP only
    float* hscale;            // [R] inverse row scales of that plane
    unsigned* bits;           // ReLU mask words in row_gemm_n384.hip's layout ([stage of 32 rows][8][64]), KEEP only
    float* pre;               // [R,128] pre-LayerNorm sum, KEEP only
    float* mean;              // [R]
    float* rstd;
    int64_t R;
    float eps;
    int reverse;              // tiles in descending order (traversal.h)
};

// KEEP: the outputs only a backward reads (h plane, its scales, mask words, pre-LayerNorm sum) are written
template <bool KEEP>
__global__ __launch_bounds__(64 * kWaves) void ffn_fused_f32_kernel(const FProb p0, const FProb p1, const int nb0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool second = static_cast<int>(blockIdx.x) >= nb0;      // uniform
    const float* __restrict__ const x = second ? p1.x : p0.x;
    const char* __restrict__ const packed = second ? p1.packed : p0.packed;
    float* __restrict__ const y = second ? p1.y : p0.y;
    _Float16* __restrict__ const hplane = second ? p1.h : p0.h;
    float* __restrict__ const hscale = second ? p1.hscale : p0.hscale;
    unsigned* __restrict__ const gbits = second ? p1.bits : p0.bits;
    float* __restrict__ const pre = second ? p1.pre : p0.pre;
    float* __restrict__ const gmean = second ? p1.mean : p0.mean;
    float* __restrict__ const grstd = second ? p1.rstd : p0.rstd;
    const int64_t R = second ? p1.R : p0.R;
    const float eps = second ? p1.eps : p0.eps;
    const int reverse = second ? p1.reverse : p0.reverse;
    const int bidx = second ? static_cast<int>(blockIdx.x) - nb0 : static_cast<int>(blockIdx.x);
    const int nblk = second ? static_cast<int>(gridDim.x) - nb0 : nb0;
    const int64_t tiles = (R + kTile - 1) / kTile;
    const int T = static_cast<int>((tiles - bidx + nblk - 1) / nblk);      // >= 1: the launch has at most `tiles` workgroups
    auto tile_of = [&](int t) {
        const int64_t st = bidx + static_cast<int64_t>(t) * nblk;
        const int64_t c = st < tiles ? st : tiles - 1;      // (prefetch past the end: any valid tile)
        return reverse ? tiles - 1 - c : c;
    };
    const int n = lane & 15, kq = lane >> 4;
    float* const tab = reinterpret_cast<float*>(smem + kOffTab);
    const unsigned lds_w = lds_byte_address(smem + kOffW);
    const unsigned lds_x = lds_byte_address(smem + kOffX) + static_cast<unsigned>(w) * 8192u;

    // ---- streams: weight chunk c of a pass (0..11 fc1, 12..19 fc2) into buffer b; wave w copies its share of fragments
    const unsigned lane16 = static_cast<unsigned>(lane) * 16u;
    auto dma_chunk = [&](int c, int b) {
        c = c >= 20 ? c - 20 : c;
        const int per = c < 12 ? 2 : 3;
        if (FF_DBG & 8) return;
        // (uniform: the wave's first fragment of the chunk; the following ones are instruction offsets)
        const char* src = (c < 12 ? packed + c * kW1Chunk : packed + kW1Bytes + (c - 12) * kW2Chunk) + w * per * 1024;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_w + static_cast<unsigned>(b * kW2Chunk + w * per * 1024));
        dma16_s<0>(src, lane16, dst);
        dma16_s<1024>(src, lane16, dst);
        if (per == 3) dma16_s<2048>(src, lane16, dst);
    };
    // the 16 rows of this wave in tile `tile`, as B fragments in the wave's LDS region: fragment f = (ks, half) holds, for lane
    // (n, kq), x[row n][32 ks + 8 kq + 4 half .. + 3]; rows past the end read the last row (their results are dropped)
    auto dma_x = [&](int64_t tile, int f) {
        const int64_t r0t = tile * kTile + w * 16;
        const int64_t rb = r0t < R - 1 ? r0t : R - 1;      // uniform
        const int64_t rl = (r0t + n < R - 1 ? r0t + n : R - 1) - rb;
        const unsigned voff = static_cast<unsigned>(rl) * 512u + static_cast<unsigned>(kq) * 32u;
        // (the instruction offset (f >> 1) * 128 + (f & 1) * 16 moves the LDS address too: taken off the base)
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_x + static_cast<unsigned>(f * 1024 - ((f >> 1) * 128 + (f & 1) * 16)));
        const float* sb = x + rb * 128;
        switch (f) {      // (f is a constant after unrolling: instruction offset (f >> 1) * 128 + (f & 1) * 16)
            case 0: dma16_s<0>(sb, voff, dst); break;
            case 1: dma16_s<16>(sb, voff, dst); break;
            case 2: dma16_s<128>(sb, voff, dst); break;
            case 3: dma16_s<144>(sb, voff, dst); break;
            case 4: dma16_s<256>(sb, voff, dst); break;
            case 5: dma16_s<272>(sb, voff, dst); break;
            case 6: dma16_s<384>(sb, voff, dst); break;
            default: dma16_s<400>(sb, voff, dst); break;
        }
    };

    // ---- prologue: tables, mask words zeroed, chunks 0 and 1, the first tile's rows
    {
        const float* inv_cs = reinterpret_cast<const float*>(packed + kW1Bytes + kW2Bytes);
        const float* b1 = second ? p1.b1 : p0.b1;
        const float* b2 = second ? p1.b2 : p0.b2;
        const float* gamma = second ? p1.gamma : p0.gamma;
        const float* beta = second ? p1.beta : p0.beta;
        for (int i = threadIdx.x; i < 384; i += 64 * kWaves) {
            tab[i] = inv_cs[i];
            tab[384 + i] = b1[i];
        }
        if (threadIdx.x < 128) {
            const int i = threadIdx.x;
            tab[768 + i] = inv_cs[384 + i];
            tab[896 + i] = b2[i];
            tab[1024 + i] = gamma[i];
            tab[1152 + i] = beta[i];
        }
        *reinterpret_cast<u32x4*>(smem + kOffBits + threadIdx.x * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    dma_chunk(0, 0);
    dma_chunk(1, 1);
#pragma unroll
    for (int f = 0; f < 8; ++f) dma_x(tile_of(0), f);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();

    // Counted waits.  VMEM operations of a wave retire in order (vmcnt(N) = all but the youngest N are done), stores included: a
    // weight chunk requested behind a burst of stores is not confirmed before the burst is acknowledged.  The first version
    // issued the h plane (13 stores) and y / pre / statistics (18) in two bursts per pass and ran 436 us, 306 us with the stores
    // compiled out and the same 445 us with every wait a vmcnt(0): the store drain was ADDED to the compute time.  So the stores are
    // spread: the results of pass t leave during the fc1 iterations 0..7 of pass t + 1 (two per iteration; z, the mean and rstd
    // stay in registers across the pass boundary), the h plane during the fc2 iterations (one or two per iteration).
    // Issue order of iteration c of a pass: wait for chunk c, barrier, DMA of chunk c + 2 (2 instructions for an fc1 chunk, 3 for
    // an fc2 chunk), then post(c) further VMEM instructions, compute.  Behind the DMA of chunk c (issued first in iteration c - 2)
    // and in front of its wait there are post(c - 2) + the DMA of chunk c + 1 + post(c - 1).
    constexpr bool ST = !(FF_DBG & 2);
    auto post = [&](int c) {      // VMEM instructions of iteration c behind its weight DMA
        c = (c + 20) % 20;
        if (c < 8) return ST ? (KEEP ? 2 : 1) + (c == 0 ? 2 : 0) : 0;      // pre + y of a 16-byte column pair (+ mean, rstd)
        if (c < 12) return 0;
        const int ob = c - 12;
        return 1 + (ST && KEEP ? (ob < 4 ? 2 : 1) + (ob == 0 ? 2 : 0) : 0);      // rows fragment; h plane (+ scale, mask words)
    };
    auto younger = [&](int c) { return post(c - 2) + ((c + 1) % 20 < 12 ? 2 : 3) + post(c - 1); };

    int gb = 0;      // buffer of the chunk about to be consumed (uniform)
    // results of the previous pass that have not left yet: z = pre-LayerNorm sum (32 values per lane), its row statistics, the
    // rows they belong to (0 rows before the first pass: every store is dropped by its descriptor's range, the counts stay)
    float z[4][8] = {};
    float mu_p = 0.f, rstd_p = 0.f;
    int64_t r0_p = 0;
    int rows_p = 0;
    const unsigned yoff = static_cast<unsigned>(n) * 512u + static_cast<unsigned>(kq) * 32u;
    const unsigned soff = kq == 0 ? static_cast<unsigned>(n) * 4u : 0x7FFFFFF0u;      // one lane per row writes the statistics
    // (lane (n, kq) holds channels 32 p + 8 kq + 4 half .. + 3 of row n.  The W2 channel order 16 half + 4 kq -- 64 contiguous,
    // aligned bytes per row and store instruction instead of every second 16-byte piece of the line -- was measured SLOWER:
    // 460 vs 428 us at R = 518 400.)
    unsigned long long st_ticks = 0;
    auto store_results = [&](int i) {      // 16-byte column group i (p = i >> 1, half = i & 1) of the previous pass's rows
        const int p = i >> 1, hf = i & 1;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y + r0_p * 128, 0, rows_p * 512, 0x00020000);
        const float4 g = ld4(tab + 1024 + 32 * p + 8 * kq + 4 * hf), e = ld4(tab + 1152 + 32 * p + 8 * kq + 4 * hf);
        const u32x4 o = {__float_as_uint(fmaf((z[p][4 * hf] - mu_p) * rstd_p, g.x, e.x)), __float_as_uint(fmaf((z[p][4 * hf + 1] - mu_p) * rstd_p, g.y, e.y)),
                         __float_as_uint(fmaf((z[p][4 * hf + 2] - mu_p) * rstd_p, g.z, e.z)), __float_as_uint(fmaf((z[p][4 * hf + 3] - mu_p) * rstd_p, g.w, e.w))};
        unsigned long long ta = 0;
        if (FF_PROF == 2) ta = ff_now();
        if (KEEP) {
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(pre + r0_p * 128, 0, rows_p * 512, 0x00020000);
            if (!(FF_DBG & 1024))
                FF_STORE128(u32x4{__float_as_uint(z[p][4 * hf]), __float_as_uint(z[p][4 * hf + 1]), __float_as_uint(z[p][4 * hf + 2]),
                                  __float_as_uint(z[p][4 * hf + 3])}, rp, yoff, p * 128 + hf * 16, 0);
        }
        if (!(FF_DBG & 4096)) FF_STORE128(o, ry, yoff, p * 128 + hf * 16, 0);
        if (FF_PROF == 2) st_ticks += ff_now() - ta;
        if (i == 0) {
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(gmean + r0_p, 0, rows_p * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(grstd + r0_p, 0, rows_p * 4, 0x00020000);
            FF_STORE32(__float_as_uint(mu_p), rm, soff, 0, 0);
            FF_STORE32(__float_as_uint(rstd_p), rr, soff, 0, 0);
        }
    };
    unsigned long long prof[6] = {0, 0, 0, 0, 0, 0}, pt0 = 0, pt1 = 0, pt2 = 0;      // fc1: wait, issue, compute; fc2: the same
#pragma nounroll
    for (int t = 0; t < T; ++t) {
        const int64_t tile = tile_of(t);
        const int64_t r0 = tile * kTile + w * 16;      // first row of this wave
        const int64_t leftw = R - r0;
        const int rows = leftw <= 0 ? 0 : (leftw < 16 ? static_cast<int>(leftw) : 16);      // 0: every store is dropped

        // ---- x rows: LDS -> registers, row maximum, hi / lo planes (the float32 rows stay in the LDS region: fc2 reads its
        // residual operand from there, fragment by fragment, and refills each fragment with the next tile's)
        vm_wait(FF_SAFE_WAIT ? 0 : post(19) - 1);      // the last fragment was requested in iteration 19, in front of its h-plane store
        float xa[4][8];
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const float4 v = *reinterpret_cast<const float4*>(smem + kOffX + w * 8192 + f * 1024 + lane * 16);
            xa[f >> 1][(f & 1) * 4 + 0] = v.x;
            xa[f >> 1][(f & 1) * 4 + 1] = v.y;
            xa[f >> 1][(f & 1) * 4 + 2] = v.z;
            xa[f >> 1][(f & 1) * 4 + 3] = v.w;
        }
        float mx = 0.f;
        if (!(FF_DBG & 64)) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 8; ++i) mx = fmaxf(mx, fabsf(xa[ks][i]));
            mx = xor_max<16>(mx);      // the four lanes (kq) of row n
        }
        const unsigned ex = scale_exponent(mx);
        const float inv_sx = inv_scale_of(ex);
        f16x8 xh[4], xl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (FF_DBG & 64) fake_split8(xa[ks], xh[ks], xl[ks]);
            else split8(xa[ks], scale_of(ex), xh[ks], xl[ks]);
        }

        // ---- fc1: h = relu(x W1^T + b1), 12 pairs of 16-channel blocks; lane (n, kq) ends with channels 32 j + 8 kq .. + 7.
        // The epilogue of pair j - 1 (scales, bias, ReLU, mask bits: ~25 vector instructions per element quarter) is issued in
        // four pieces BETWEEN the MFMA groups of pair j: the matrix pipe runs them for free, and the first piece covers the LDS
        // round trip of the pair's first fragments.  Two accumulators per block (even / odd k-steps, terms smallest first) in
        // two sets that alternate from pair to pair.
        float hv[12][8];
        const int rb = w & 1, stg = w >> 1;      // row block / stage of this wave's rows in the mask layout (32-row stages)
        f32x4 acc1[2][4];                        // [set][A even, A odd, B even, B odd]
        float4 tca, tcb, tba, tbb;               // inverse column scales and bias of the pair whose epilogue is pending
        unsigned ma = 0, mb = 0;
        auto fc1_tables = [&](int j) {
            tca = ld4(tab + 32 * j + 8 * kq);
            tcb = ld4(tab + 32 * j + 8 * kq + 4);
            tba = ld4(tab + 384 + 32 * j + 8 * kq);
            tbb = ld4(tab + 384 + 32 * j + 8 * kq + 4);
        };
        auto fc1_epilogue = [&](int j, int i) {      // element quarter i of pair j (accumulator set j & 1)
            const f32x4(&a)[4] = acc1[j & 1];
            if (i == 0) asm volatile("s_nop 7" ::: "memory");      // (MFMA results of the pair -> first vector read: far away already)
            if (FF_DBG & 16) {
                hv[j][i] = a[0][i] + a[1][i];
                hv[j][4 + i] = a[2][i] + a[3][i];
                return;
            }
            const float ca = i == 0 ? tca.x : i == 1 ? tca.y : i == 2 ? tca.z : tca.w, cb = i == 0 ? tcb.x : i == 1 ? tcb.y : i == 2 ? tcb.z : tcb.w;
            const float ba = i == 0 ? tba.x : i == 1 ? tba.y : i == 2 ? tba.z : tba.w, bb = i == 0 ? tbb.x : i == 1 ? tbb.y : i == 2 ? tbb.z : tbb.w;
            const float va = fmaf(a[0][i] + a[1][i], inv_sx * ca, ba);
            const float vb = fmaf(a[2][i] + a[3][i], inv_sx * cb, bb);
            if (KEEP) {
                if (i == 0) ma = mb = 0;
                ma |= va > 0.f ? (1u << i) : 0u;
                mb |= vb > 0.f ? (1u << i) : 0u;
            }
            hv[j][i] = fmaxf(va, 0.f);
            hv[j][4 + i] = fmaxf(vb, 0.f);
            if (KEEP && i == 3) {
                // channel c = 32 j + 8 kq + 4 blk + i of row 16 rb + n of the stage: word (c / 48, lane' = ((c % 16) / 4) * 16 + n),
                // bit (rb * 3 + (c % 48) / 16) * 4 + i  (row_gemm_n384.hip)
                const int b16 = 2 * j + (kq >> 1);            // 16-channel block of the lane's eight channels
                const int wv = b16 / 3, cbk = b16 - 3 * wv;
                unsigned* const word = reinterpret_cast<unsigned*>(smem + kOffBits) + stg * 512 + wv * 64 + (2 * (kq & 1)) * 16 + n;
                const int sh = (rb * 3 + cbk) * 4;
                atomicOr(word, ma << sh);
                atomicOr(word + 16, mb << sh);
            }
        };
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            if (FF_PROF) pt0 = ff_now();
            vm_wait(FF_SAFE_WAIT ? 0 : younger(j));
            pass_barrier();
            if (FF_PROF) pt1 = ff_now();
            dma_chunk(j + 2, gb >= 1 ? gb - 1 : 2);      // buffer (gb + 2) % 3
            if (j < 8) store_results(j);                 // the previous pass's results, two stores per iteration
            if (FF_PROF) pt2 = ff_now();
            const char* wb = smem + kOffW + gb * kW2Chunk + lane * 16;
            // fragments one k-step ahead of their MFMAs (scheduling fences: hipcc otherwise requests all 16 at once, 64 VGPRs)
            f16x8 fr[2][4];
            auto read_frags = [&](int ks, f16x8 (&d)[4]) {
                if (FF_DBG & 4) {
                    d[0] = d[1] = d[2] = d[3] = xh[ks];
                    return;
                }
                d[0] = *reinterpret_cast<const f16x8*>(wb + (ks * 2 + 0) * 1024);            // block A hi, lo
                d[1] = *reinterpret_cast<const f16x8*>(wb + (ks * 2 + 1) * 1024);
                d[2] = *reinterpret_cast<const f16x8*>(wb + ((4 + ks) * 2 + 0) * 1024);      // block B hi, lo
                d[3] = *reinterpret_cast<const f16x8*>(wb + ((4 + ks) * 2 + 1) * 1024);
            };
            read_frags(0, fr[0]);
            f32x4(&a)[4] = acc1[j & 1];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < 3) read_frags(ks + 1, fr[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (j > 0) fc1_epilogue(j - 1, ks);
                __builtin_amdgcn_sched_barrier(0);
                const f16x8 &wha = fr[ks & 1][0], &wla = fr[ks & 1][1], &whb = fr[ks & 1][2], &wlb = fr[ks & 1][3];
                f32x4 &ca = a[ks & 1], &cb = a[2 + (ks & 1)];
                if (ks < 2) {
                    mfma16_first(ca, wla, xh[ks]);
                    mfma16_first(cb, wlb, xh[ks]);
                } else {
                    mfma16(ca, wla, xh[ks]);
                    mfma16(cb, wlb, xh[ks]);
                }
                mfma16(ca, wha, xl[ks]);
                mfma16(cb, whb, xl[ks]);
                mfma16(ca, wha, xh[ks]);
                mfma16(cb, whb, xh[ks]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (FF_PROF) {
                const unsigned long long pt3 = ff_now();
                prof[0] += pt1 - pt0;
                prof[1] += pt2 - pt1;
                prof[2] += pt3 - pt2;
            }
            gb = gb == 2 ? 0 : gb + 1;
            fc1_tables(j);      // (read behind the last quarter of pair j - 1, used from the next iteration on)
        }
        mfma_results_ready();
#pragma unroll
        for (int i = 0; i < 4; ++i) fc1_epilogue(11, i);

        // ---- h: ONE scale per row over all 384 channels, hi / lo planes; the hi plane leaves for the backward
        float mh = 0.f;
        if (!(FF_DBG & 32)) {
#pragma unroll
            for (int j = 0; j < 12; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) mh = fmaxf(mh, hv[j][i]);
            mh = xor_max<16>(mh);
        }
        const unsigned eh = scale_exponent(mh);
        const float inv_sh = inv_scale_of(eh);
        f16x8 hh[12], hl[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            if (FF_DBG & 32) fake_split8(hv[j], hh[j], hl[j]);
            else split8(hv[j], scale_of(eh), hh[j], hl[j]);
        }

        // ---- fc2: z = x + h W2^T + b2, 8 blocks of 16 output channels; lane (n, kq) ends with channels 32 p + 8 kq .. + 7.
        // Three accumulation chains per block (w_lo.h_hi, w_hi.h_lo, w_hi.h_hi) in two alternating sets; the epilogue of block
        // ob - 1 rides behind the second MFMA group of block ob.
        f32x4 acc2[2][3];
        float4 xres[2], tcs, tbs;
        auto fc2_epilogue = [&](int ob) {
            const int p = ob >> 1, blk = ob & 1;
            const f32x4(&q)[3] = acc2[ob & 1];
            const float4 xr = xres[ob & 1];
            const float csv[4] = {tcs.x, tcs.y, tcs.z, tcs.w}, bsv[4] = {tbs.x, tbs.y, tbs.z, tbs.w};
            const float xrv[4] = {xr.x, xr.y, xr.z, xr.w};
            asm volatile("s_nop 7" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) z[p][4 * blk + i] = fmaf((q[0][i] + q[1][i]) + q[2][i], inv_sh * csv[i], bsv[i]) + xrv[i];
        };
#pragma unroll
        for (int ob = 0; ob < 8; ++ob) {
            if (FF_PROF) pt0 = ff_now();
            vm_wait(FF_SAFE_WAIT ? 0 : younger(12 + ob));
            pass_barrier();
            if (FF_PROF) pt1 = ff_now();
            dma_chunk(12 + ob + 2, gb >= 1 ? gb - 1 : 2);
            if (KEEP && ob == 0) {
                // every wave's fc1 bits of the pass are in LDS (the barrier above): 2048 words leave, 16 bytes per thread, and
                // are zeroed for the next pass
                const int64_t stages = (R + 31) / 32 - tile * 4;
                const int nst = stages < 4 ? static_cast<int>(stages) : 4;
                const __amdgpu_buffer_rsrc_t rbits = __builtin_amdgcn_make_buffer_rsrc(gbits + tile * 2048, 0, nst * 2048, 0x00020000);
                u32x4* const src = reinterpret_cast<u32x4*>(smem + kOffBits + threadIdx.x * 16);
                const u32x4 v = *src;
                *src = u32x4{0u, 0u, 0u, 0u};
                FF_STORE128(v, rbits, static_cast<unsigned>(threadIdx.x) * 16u, 0, 0);
            }
            // residual operand: fragment ob of this tile's rows (channels 32 p + 8 kq + 4 blk .. of row n); its slot is refilled
            // with the same fragment of the next tile
            xres[ob & 1] = *reinterpret_cast<const float4*>(smem + kOffX + w * 8192 + ob * 1024 + lane * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_x(tile_of(t + 1), ob);
            if (KEEP) {      // the hi plane of h for the backward: twelve 16-byte stores, one or two per iteration (+ the row scale)
                const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(hplane + r0 * 384, 0, rows * 768, 0x00020000);
                const unsigned voff = static_cast<unsigned>(n) * 768u + static_cast<unsigned>(kq) * 16u;
                unsigned long long ta = 0;
                if (FF_PROF == 3) ta = ff_now();
                if (FF_DBG & 2048) {
                } else if (ob < 4) {
                    FF_STORE128(__builtin_bit_cast(u32x4, hh[2 * ob]), rh, voff, (2 * ob) * 64, 0);
                    FF_STORE128(__builtin_bit_cast(u32x4, hh[2 * ob + 1]), rh, voff, (2 * ob + 1) * 64, 0);
                } else {
                    FF_STORE128(__builtin_bit_cast(u32x4, hh[4 + ob]), rh, voff, (4 + ob) * 64, 0);
                }
                if (FF_PROF == 3) st_ticks += ff_now() - ta;
                if (ob == 0) {
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(hscale + r0, 0, rows * 4, 0x00020000);
                    FF_STORE32(__float_as_uint(inv_sh), rs, soff, 0, 0);
                }
            }
            if (FF_PROF) pt2 = ff_now();
            const char* wb = smem + kOffW + gb * kW2Chunk + lane * 16;
            f32x4(&q)[3] = acc2[ob & 1];
            f16x8 fr[3][2];
            auto read_frags = [&](int j, f16x8 (&d)[2]) {
                if (FF_DBG & 4) {
                    d[0] = d[1] = hh[j];
                    return;
                }
                d[0] = *reinterpret_cast<const f16x8*>(wb + (j * 2 + 0) * 1024);
                d[1] = *reinterpret_cast<const f16x8*>(wb + (j * 2 + 1) * 1024);
            };
            read_frags(0, fr[0]);
            read_frags(1, fr[1]);
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                if (j < 10) read_frags(j + 2, fr[(j + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0);
                if (j == 1 && ob > 0) fc2_epilogue(ob - 1);
                __builtin_amdgcn_sched_barrier(0);
                const f16x8 &wh = fr[j % 3][0], &wl = fr[j % 3][1];
                if (j == 0) {
                    mfma16_first(q[0], wl, hh[0]);
                    mfma16_first(q[1], wh, hl[0]);
                    mfma16_first(q[2], wh, hh[0]);
                } else {
                    mfma16(q[0], wl, hh[j]);
                    mfma16(q[1], wh, hl[j]);
                    mfma16(q[2], wh, hh[j]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (FF_PROF) {
                const unsigned long long pt3 = ff_now();
                prof[3] += pt1 - pt0;
                prof[4] += pt2 - pt1;
                prof[5] += pt3 - pt2;
            }
            gb = gb == 2 ? 0 : gb + 1;
            const int p = ob >> 1, blk = ob & 1;
            tcs = ld4(tab + 768 + 32 * p + 8 * kq + 4 * blk);
            tbs = ld4(tab + 896 + 32 * p + 8 * kq + 4 * blk);
        }
        mfma_results_ready();
        fc2_epilogue(7);

        // ---- LayerNorm statistics of the row (32 values per lane, four lanes per row); y leaves during the next pass
        float mu = 0.f;
        if (FF_DBG & 128) {
            rstd_p = 1.f;
        } else {
            float s1 = 0.f;
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int i = 0; i < 8; ++i) s1 += z[p][i];
            mu = xor_sum<16>(s1) * (1.0f / 128.0f);
            float s2 = 0.f;
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float d = z[p][i] - mu;
                    s2 = fmaf(d, d, s2);
                }
            rstd_p = rsqrtf(xor_sum<16>(s2) * (1.0f / 128.0f) + eps);
        }
        mu_p = mu;
        r0_p = r0;
        rows_p = rows;
    }
    // the last pass's results
#pragma unroll
    for (int i = 0; i < 8; ++i) store_results(i);
    // the DMAs issued for a pass that does not come must land before the LDS is handed to the next workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (FF_PROF && bidx == 0 && !second && lane == 0 && R > 300064)
        for (int i = 0; i < 6; ++i) gmean[300000 + w * 6 + i] = static_cast<float>(FF_PROF > 1 && i == 1 ? st_ticks : prof[i]);
}

// ---- pack: one wave per output channel.  W1 [384,128] (blocks 0..383), W2 [128,384] (blocks 384..511): the row is scaled by
// the power of two that puts its largest magnitude into [2^14, 2^15), split hi / lo, and scattered into the fragment order of
// the kernel above.  Channel c = 32 j + 8 g + 4 blk + i sits at MFMA row m = 4 g + i of block blk of pair j.
__global__ __launch_bounds__(64) void ffn_f32_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2, char* __restrict__ packed) {
    const int b = blockIdx.x, t = threadIdx.x;
    const bool first = b < 384;
    const int c = first ? b : b - 384;
    const int K = first ? 128 : 384;
    const float* row = first ? w1 + static_cast<size_t>(c) * 128 : w2 + static_cast<size_t>(c) * 384;
    const int per = K / 64;      // 2 or 6 consecutive k per thread
    float v[6];
    float mx = 0.f;
    for (int i = 0; i < per; ++i) {
        v[i] = row[t * per + i];
        mx = fmaxf(mx, fabsf(v[i]));
    }
    mx = xor_max<1>(mx);
    const unsigned e = scale_exponent(mx);
    const float sc = scale_of(e);
    const int pj = c >> 5, rem = c & 31, m = 4 * (rem >> 3) + (rem & 3), blk = (rem >> 2) & 1;
    for (int i = 0; i < per; ++i) {
        const int k = t * per + i;
        const int ks = k >> 5, kq = (k >> 3) & 3, tt = k & 7;
        const float xs = v[i] * sc;
        const _Float16 hi = static_cast<_Float16>(xs);
        const _Float16 lo = static_cast<_Float16>(xs - static_cast<float>(hi));
        size_t off;
        if (first) off = static_cast<size_t>(pj) * kW1Chunk + static_cast<size_t>((blk * 4 + ks) * 2) * 1024 + (kq * 16 + m) * 16 + tt * 2;
        else off = kW1Bytes + static_cast<size_t>(2 * pj + blk) * kW2Chunk + static_cast<size_t>(ks * 2) * 1024 + (kq * 16 + m) * 16 + tt * 2;
        *reinterpret_cast<_Float16*>(packed + off) = hi;
        *reinterpret_cast<_Float16*>(packed + off + 1024) = lo;
    }
    if (t == 0) reinterpret_cast<float*>(packed + kW1Bytes + kW2Bytes)[first ? c : 384 + c] = inv_scale_of(e);
}

int check_args(const dg_ffn_fwd_args* a, const char* who) {
    if (!a || !a->x || !a->w1_packed || !a->b1 || !a->b2 || !a->gamma || !a->beta || !a->y || !a->mean || !a->rstd)
        return fail(DG_E_ARG, "dg_ffn_ln_fwd_f32: null pointer (%s)", who);
    const bool keep = a->h != nullptr;
    if (keep != (a->relu_bits != nullptr) || keep != (a->pre_ln != nullptr))
        return fail(DG_E_ARG, "dg_ffn_ln_fwd_f32: h, relu_bits and pre_ln go together (%s)", who);
    if (a->R < 0) return fail(DG_E_SHAPE, "dg_ffn_ln_fwd_f32: negative row count (%s)", who);
    return 0;
}

FProb make_prob(const dg_ffn_fwd_args* a) {
    char* h = static_cast<char*>(a->h);
    return FProb{static_cast<const float*>(a->x), static_cast<const char*>(a->w1_packed), a->b1, a->b2, a->gamma, a->beta,
                 static_cast<float*>(a->y), reinterpret_cast<_Float16*>(h),
                 h ? reinterpret_cast<float*>(h + dg_hidden_scale_offset(a->R, 384)) : nullptr, a->relu_bits,
                 static_cast<float*>(a->pre_ln), a->mean, a->rstd, a->R, a->eps, take_direction(a->R)};
}

}  // namespace
}  // namespace dg

using namespace dg;

extern "C" size_t dg_ffn_f32_packed_bytes(void) { return kPackedBytes; }

extern "C" int dg_ffn_f32_pack(const float* w1, const float* w2, void* packed, dg_stream_t stream) {
    if (!w1 || !w2 || !packed) return fail(DG_E_ARG, "dg_ffn_f32_pack: null pointer");
    hipLaunchKernelGGL(ffn_f32_pack_kernel, dim3(512), dim3(64), 0, static_cast<hipStream_t>(stream), w1, w2, static_cast<char*>(packed));
    return check_launch("dg_ffn_f32_pack");
}

extern "C" int dg_ffn_ln_fwd_f32(const dg_ffn_fwd_args* node, const dg_ffn_fwd_args* edge, dg_stream_t stream_) {
    if (int st = check_args(edge, "edge")) return st;
    if (node)
        if (int st = check_args(node, "node")) return st;
    if (node && (node->h != nullptr) != (edge->h != nullptr))
        return fail(DG_E_ARG, "dg_ffn_ln_fwd_f32: both problems keep their backward outputs or neither does");
    if (node && node->R == 0) node = nullptr;
    if (edge->R == 0 && !node) return 0;
    if (edge->R == 0) {
        edge = node;
        node = nullptr;
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const FProb pe = make_prob(edge);
    const FProb pn = node ? make_prob(node) : pe;
    const int64_t t0 = (pe.R + kTile - 1) / kTile, t1 = node ? (pn.R + kTile - 1) / kTile : 0;
    int nb0, nb1;
    pair_split(t0, t1, 256, &nb0, &nb1);
    ProfScope prof(pe.R < edge_rows() ? DG_K_FFN_F32_NODE : DG_K_FFN_F32, stream);
    if (edge->h) {
        DG_OPT_IN_LDS((&ffn_fused_f32_kernel<true>), kLds);
        hipLaunchKernelGGL((ffn_fused_f32_kernel<true>), dim3(nb0 + nb1), dim3(64 * kWaves), kLds, stream, pe, pn, nb0);
    } else {
        DG_OPT_IN_LDS((&ffn_fused_f32_kernel<false>), kLds);
        hipLaunchKernelGGL((ffn_fused_f32_kernel<false>), dim3(nb0 + nb1), dim3(64 * kWaves), kLds, stream, pe, pn, nb0);
    }
    return check_launch("dg_ffn_ln_fwd_f32");
}
