// Weight/bias gradient of nn.Linear over the edge rows -- the contraction the
// reference runs as `mm(dy.t(), x)` + `sum(dy, 0)` in every Linear backward of
// src/model/layers.py (q/k/v/e/out_e/out_n, MLP.fc1/fc2) and again inside the
// gradient-penalty double backward (src/model/loss.py:32-39):
//
//     dW[n][k] = sum_r dy[r][n] * x[r][k]        db[n] = sum_r dy[r][n]
//
// with R = B*N*N (518 400 at configs[1]) and N,K in {64,128,384}: a tall-skinny
// "TN" GEMM whose contraction runs over the huge row dimension.  fp32 MFMA
// (v_mfma_f32_32x32x2_f32, exact fp32): the full [N,K] output lives in the
// accumulators of one workgroup (NT x KT tiles of 32x32 over WN x WK waves), the
// workgroup streams its share of rows HBM -> LDS with the async LDS-DMA
// (global_load_lds, 16 B/lane; a TR-row tile of a row-major matrix is one
// contiguous chunk, so the LDS image is lane-linear), double buffered, and
// feeds MFMA operands with conflict-free ds_read_b32 (lane&31 walks a row).
// Split-K partials are reduced by a second kernel in a fixed order.
#include "bf16.h"
#include "traversal.h"
#include "wgrad_stream.h"

#include <cstring>
#include <type_traits>

namespace dg {
void launch_ln_finish(const float* part, int nblocks, int K, int C, float* out0, float* out1, hipStream_t stream);      // layernorm.hip
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Copy `rows` x W floats (contiguous in global memory starting at g) into LDS at l.
// Rows >= valid_rows are skipped (their LDS bytes must have been zeroed before).
template <int THREADS>
__device__ __forceinline__ void stage_tile(const float* g, float* l, int W, int rows, int valid_rows) {
    const int chunks = rows * W / 4;          // 16-byte chunks
    const int valid = valid_rows * W / 4;
    const unsigned base = lds_byte_address(l);
    for (int c = threadIdx.x; c < chunks; c += THREADS) {
        // the LDS destination of the DMA is (wave-uniform base) + lane*16: c is lane-linear per wave
        if (c < valid) dma16_async(g + static_cast<size_t>(c) * 4, base + (c - (threadIdx.x & 63)) * 16);
    }
}

// exact three-way bf16 split of 8 fp32 values (see row_gemm.hip: h + m + l covers all 24 significand
// bits; the six cross products with i + j <= 4 on v_mfma_f32_32x32x16_bf16 are as accurate as fp32 FMAs)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 hp, mp, lp;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned a0 = __float_as_uint(x[2 * i]), a1 = __float_as_uint(x[2 * i + 1]);
        hp[i] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
        const float r0 = x[2 * i] - __uint_as_float(a0 & 0xFFFF0000u);
        const float r1 = x[2 * i + 1] - __uint_as_float(a1 & 0xFFFF0000u);
        const unsigned b0 = __float_as_uint(r0), b1 = __float_as_uint(r1);
        mp[i] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        const float s0 = r0 - __uint_as_float(b0 & 0xFFFF0000u);
        const float s1 = r1 - __uint_as_float(b1 & 0xFFFF0000u);
        lp[i] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    }
    h = __builtin_bit_cast(bf16x8, hp);
    m = __builtin_bit_cast(bf16x8, mp);
    l = __builtin_bit_cast(bf16x8, lp);
}

// fp16 hi + lo split of 8 scaled fp32 values: hi = s rounded toward zero to fp16 (11 significant bits), lo = s - hi
// (exact in fp32) rounded toward zero; hi_a hi_b + hi_a lo_b + lo_a hi_b on v_mfma_f32_32x32x16_f16 leaves a relative
// error of 2^-22 per product, fp32 accumulate (row_gemm.hip uses the same split for the forward / dgrad GEMMs).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// one pair of values: packed hi and lo words
__device__ __forceinline__ void split2_f16(float v0, float v1, float sc, unsigned& hw, unsigned& lw) {
    const float s0 = v0 * sc, s1 = v1 * sc;
    hw = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(s0, s1));      // hi = s rounded toward zero
    // lo = s - hi, exact in fp32: v_fma_mix_f32 reads the fp16 halves of `hw` directly (hipcc emits cvt + sub)
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hw), "v"(s0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hw), "v"(s1));
    lw = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l0, l1));
}
// 2^(8 - floor(log2 m)) for a finite m > 0: the scale that maps m into [2^8, 2^9).  A column's scale moves again
// only when a later value is 64-128 times larger than the one that set it (fp16 holds 2^16): on real gradients a
// tighter window (2^12: 4-8 times) had some column of nearly every 16-row step of a wave moving, and every move
// costs the wave ~2 steps.  Elements more than 2^11 below their column's running maximum lose relative (not
// absolute) accuracy: absolute error 2^-25 scaled units = 2^-33 of that maximum.
__device__ __forceinline__ float scale_for(float m) {
    const int e = static_cast<int>((__float_as_uint(m) >> 23) & 255u);      // biased exponent (0 for denormals)
    int be = 127 + 8 - (e - 127);
    be = be > 253 ? 253 : (be < 1 ? 1 : be);
    return __uint_as_float(static_cast<unsigned>(be) << 23);
}

// exact quotient / reciprocal of powers of two by exponent arithmetic (v_rcp_f32 is a 1-ulp approximation)
__device__ __forceinline__ float pow2_ratio(float num, float den) {      // num <= den
    const int d = static_cast<int>(__float_as_uint(num) >> 23) - static_cast<int>(__float_as_uint(den) >> 23) + 127;
    return d < 1 ? 0.f : __uint_as_float(static_cast<unsigned>(d) << 23);
}
__device__ __forceinline__ float pow2_inv(float p) {                      // p in [2^-126, 2^126]
    return __uint_as_float((254u - (__float_as_uint(p) >> 23)) << 23);
}

// SPLIT 0: fp32 MFMA (v_mfma_f32_32x32x2_f32).  SPLIT 1: operands split into three bf16 planes in registers, 6 MFMAs
// per 16-row step and tile pair (6/16 of the fp32 matrix time per flop).  SPLIT 2 (default for fp32): fp16 hi + lo,
// 3 MFMAs per step and tile pair.  fp16 has 5 exponent bits, so every COLUMN of dy and of x carries a running
// power-of-two scale: before a 16-row step is converted, the step's column maxima (8 values per lane + the partner
// lane) are compared with what the scale can hold; when a column outgrows it, the scale drops so that the new maximum
// lands in [2^8, 2^9) and the accumulators of that column are multiplied by the (exact) ratio -- a handful of times
// per launch.  Nothing overflows, the partial sums are un-scaled exactly at the end, and an element far below its
// column's running maximum keeps an ABSOLUTE error of 2^-25 scaled units, i.e. 2^-33 of that maximum.
template <typename T, int NT, int KT, int WN, int WK, int TR, bool MASK, int SPLIT = 0, int NBUF = 2>
__global__ __launch_bounds__(WN* WK * 64) void wgrad_kernel(const T* __restrict__ dy,
                                                           const T* __restrict__ dymask,
                                                           const T* __restrict__ x,
                                                           float* __restrict__ part_w, float* __restrict__ part_b,
                                                           int64_t R, int tiles_per_block) {
    constexpr int N = NT * 32, K = KT * 32, THREADS = WN * WK * 64;
    constexpr int TN = NT / WN, TK = KT / WK;   // tiles per wave
    constexpr bool BF = std::is_same<T, bf16_t>::value;
    constexpr int EPF = BF ? 2 : 1;             // elements per 4-byte word
    static_assert(NT % WN == 0 && KT % WK == 0, "wave grid must divide the tile grid");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    // buffers: [NBUF][TR*(N+K)] (dy tile, x tile) or [NBUF][TR*(2N+K)] (dy, x, mask tiles), elements of T.  NBUF > 2: a
    // ring with NBUF - 1 tiles in flight -- one 16-row tile per CU (32 KB) is 8 MB in flight on the whole chip, which at
    // ~2 us of loaded HBM latency caps the stream near 4 TB/s; the waits count the DMA instructions of the tiles behind
    // the one being consumed (the same number in every wave: the chunk counts are multiples of the block size).
    constexpr int BUF = TR * (N + K + (MASK ? N : 0));
    constexpr int W4 = 4 * EPF;                 // elements per 16-byte chunk
    constexpr int PER_TILE = (TR * N / W4) / THREADS * (MASK ? 2 : 1) + (TR * K / W4) / THREADS;
    static_assert(NBUF == 2 || ((TR * N / W4) % THREADS == 0 && (TR * K / W4) % THREADS == 0), "ring needs uniform DMA counts");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wave % WN, wk = wave / WN;
    const int64_t total_tiles = (R + TR - 1) / TR;
    const int64_t t_lo = static_cast<int64_t>(blockIdx.x) * tiles_per_block;
    int64_t t_hi = t_lo + tiles_per_block;
    if (t_hi > total_tiles) t_hi = total_tiles;

    f32x16 acc[TN][TK];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    constexpr int H = THREADS >= 2 * N ? 2 : 1;   // row halves for the bias column sums
    static_assert(THREADS >= N, "bias reduction needs one thread per column");
    float bacc = 0.f;   // bias partial: thread t owns column t % N, row part t / N
    // SPLIT 2: running scales of this lane's column in every dy / x tile of the wave (2^126: nothing seen yet)
    float scy[TN], scx[TK];
#pragma unroll
    for (int i = 0; i < TN; ++i) scy[i] = 8.507059e37f;
#pragma unroll
    for (int j = 0; j < TK; ++j) scx[j] = 8.507059e37f;

    auto issue = [&](int64_t t, int buf) {
        const int64_t r0 = t * TR;
        const int valid = static_cast<int>((R - r0) < TR ? (R - r0) : TR);
        T* ldy = lds + buf * BUF;
        T* lx = ldy + TR * N;
        if (valid < TR) {   // zero the whole buffer first (block-uniform branch)
            float* zf = reinterpret_cast<float*>(ldy);
            for (int c = threadIdx.x; c < BUF / EPF / 4; c += THREADS) st4(zf + c * 4, f4(0.f));
            __syncthreads();
        }
        // a TR-row tile of a row-major matrix is one contiguous chunk: staged as 4-byte words
        stage_tile<THREADS>(reinterpret_cast<const float*>(dy + r0 * N), reinterpret_cast<float*>(ldy), N / EPF, TR, valid);
        stage_tile<THREADS>(reinterpret_cast<const float*>(x + r0 * K), reinterpret_cast<float*>(lx), K / EPF, TR, valid);
        if (MASK)
            stage_tile<THREADS>(reinterpret_cast<const float*>(dymask + r0 * N), reinterpret_cast<float*>(lx + TR * K),
                                N / EPF, TR, valid);
    };

#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s)
        if (t_lo + s < t_hi) issue(t_lo + s, s);
    auto sync_tile = [&](int64_t t, int buf) {
        if (NBUF > 2 && t + NBUF - 2 < t_hi - 1) {      // the tiles behind this one are full tiles: leave them in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE * (NBUF - 2)) : "memory");
        } else {
            wait_all_vmem();
        }
        __syncthreads();                       // tile t landed for every wave; tile t-1 fully consumed
        if (t + NBUF - 1 < t_hi) issue(t + NBUF - 1, (buf + NBUF - 1) % NBUF);
    };
    auto bias_tile = [&](int buf) {
        const T* ldy = lds + buf * BUF;
        const T* lx = ldy + TR * N;
        if (threadIdx.x < N * H) {   // (column, row part)
            const int c = threadIdx.x % N, h = threadIdx.x / N;
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < TR / H; ++r) {
                const int o = (h * (TR / H) + r) * N + c;
                s += (MASK && !(ld1(lx + TR * K + o) > 0.f)) ? 0.f : ld1(ldy + o);
            }
            bacc += s;
        }
    };
    if constexpr (SPLIT == 2 && !BF) {
        // The accumulators are only touched by MFMAs inside the hot loop: a step whose column maxima outgrow the running
        // scales leaves the loop, the scales move and the accumulators are multiplied OUTSIDE it, and the loop is
        // re-entered at the same 16-row step (hipcc otherwise copies all 96 accumulator registers around the branch
        // in every iteration).
        constexpr int STEPS = TR / 16;
        const int half = lane >> 5, col = lane & 31;
        int64_t t = t_lo;
        int s16 = 0;
        bool need_sync = true, just_moved = false;
        float my[TN], mx[TK];
        while (t < t_hi) {
            bool moved = false;
            while (t < t_hi) {
                const int buf = static_cast<int>((t - t_lo) % NBUF);
                if (need_sync) {
                    sync_tile(t, buf);
                    need_sync = false;
                }
                const float* ldy = lds + buf * BUF + 16 * s16 * N;
                const float* lx = lds + buf * BUF + TR * N + 16 * s16 * K;
                const float* lm = lds + buf * BUF + TR * (N + K) + 16 * s16 * N;
                // phase A: this step's operands (every LDS read is issued before the first use: one latency, not one
                // per operand tile) and their column maxima (8 rows per lane + the partner lane)
                float vy[TN][8], vx[TK][8];
                float vm[MASK ? TN : 1][8];
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int o = (8 * half + j) * N + (wn * TN + i) * 32 + col;
                        vy[i][j] = ldy[o];
                        if (MASK) vm[MASK ? i : 0][j] = lm[o];
                    }
#pragma unroll
                for (int j2 = 0; j2 < TK; ++j2)
#pragma unroll
                    for (int j = 0; j < 8; ++j) vx[j2][j] = lx[(8 * half + j) * K + (wk * TK + j2) * 32 + col];
                __builtin_amdgcn_sched_barrier(0);
                bool grow = false;
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    float m = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (MASK) vy[i][j] = vm[MASK ? i : 0][j] > 0.f ? vy[i][j] : 0.f;
                        m = fmaxf(m, fabsf(vy[i][j]));
                    }
                    my[i] = xor_step<true>(m, 32);
                    grow |= my[i] * scy[i] >= 32768.f;
                }
#pragma unroll
                for (int j2 = 0; j2 < TK; ++j2) {
                    float m = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(vx[j2][j]));
                    mx[j2] = xor_step<true>(m, 32);
                    grow |= mx[j2] * scx[j2] >= 32768.f;
                }
                // wave-uniform, a few times per launch.  A step re-entered after its scales moved is not checked again
                // (an inf maximum cannot be scaled into range; it goes through as inf)
                if (__builtin_expect(!just_moved && __builtin_amdgcn_ballot_w64(grow) != 0, 0)) {
                    moved = true;
                    break;
                }
                just_moved = false;
                // phase B: split with the running scales, 3 MFMAs per tile pair (small terms first).  The waves of a
                // block run in lock step behind the tile barrier, so VALU and matrix work only overlap if they
                // alternate INSIDE a wave: operand tiles are split in the order y0, x0, x1.., y1.., and the split of
                // each one is interleaved (scheduling fences) with the MFMAs of the tile pairs the previous split
                // completed; the pairs of the last split run at the end.
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                u32x4 yh[TN], yl[TN], xh[TK], xl[TK];
                auto split_pair = [&](int job, int pr) {      // job: 0 = y0, 1..TK = x(job-1), TK+1.. = y(job-TK)
                    if (job == 0 || job > TK) {
                        const int i = job == 0 ? 0 : job - TK;
                        unsigned hw, lw;
                        split2_f16(vy[i][2 * pr], vy[i][2 * pr + 1], scy[i], hw, lw);
                        yh[i][pr] = hw;
                        yl[i][pr] = lw;
                    } else {
                        const int j2 = job - 1;
                        unsigned hw, lw;
                        split2_f16(vx[j2][2 * pr], vx[j2][2 * pr + 1], scx[j2], hw, lw);
                        xh[j2][pr] = hw;
                        xl[j2][pr] = lw;
                    }
                };
                auto mfma_part = [&](int i, int j2, int part) {
                    const f16x8 a = __builtin_bit_cast(f16x8, part == 0 ? yl[i] : yh[i]);
                    const f16x8 bq = __builtin_bit_cast(f16x8, part == 1 ? xl[j2] : xh[j2]);
                    acc[i][j2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq, acc[i][j2], 0, 0, 0);
                };
#pragma unroll
                for (int job = 0; job < TN + TK; ++job) {
                    if (job < 2) {
#pragma unroll
                        for (int pr = 0; pr < 4; ++pr) split_pair(job, pr);
                        continue;
                    }
                    // tile pairs completed by the previous split: x(j) -> (0, j); y(i) -> (i, 0..TK-1)
                    const int prev = job - 1;
                    const int ng = prev <= TK ? 1 : TK;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < 3 * ng; ++m) {
                        const int g = m / 3;
                        mfma_part(prev <= TK ? 0 : prev - TK, prev <= TK ? prev - 1 : g, m % 3);
#pragma unroll
                        for (int pr = 4 * m / (3 * ng); pr < 4 * (m + 1) / (3 * ng); ++pr) split_pair(job, pr);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                {
                    const int prev = TN + TK - 1;      // the last split: y(TN-1), or x(TK-1) when TN == 1
#pragma unroll
                    for (int m = 0; m < 3 * (prev <= TK ? 1 : TK); ++m)
                        mfma_part(prev <= TK ? 0 : prev - TK, prev <= TK ? prev - 1 : m / 3, m % 3);
                }
                if (++s16 == STEPS) {
                    if (part_b) bias_tile(buf);
                    s16 = 0;
                    ++t;
                    need_sync = true;
                }
            }
            if (!moved) break;
            just_moved = true;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const float nsc = my[i] * scy[i] >= 32768.f ? scale_for(my[i]) : scy[i];
                const float ratio = pow2_ratio(nsc, scy[i]);
                scy[i] = nsc;
                // dy columns are accumulator ROWS: row (reg & 3) + 8 (reg >> 2) + 4 half is held by lane `row`
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
                    const float rr = __int_as_float(__builtin_amdgcn_ds_bpermute(row * 4, __float_as_int(ratio)));
#pragma unroll
                    for (int j2 = 0; j2 < TK; ++j2) acc[i][j2][reg] *= rr;
                }
            }
#pragma unroll
            for (int j2 = 0; j2 < TK; ++j2) {
                const float nsc = mx[j2] * scx[j2] >= 32768.f ? scale_for(mx[j2]) : scx[j2];
                const float ratio = pow2_ratio(nsc, scx[j2]);      // x columns are accumulator COLUMNS: this lane's own
                scx[j2] = nsc;
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) acc[i][j2][reg] *= ratio;
            }
        }
    } else
    for (int64_t t = t_lo; t < t_hi; ++t) {
        const int buf = static_cast<int>((t - t_lo) % NBUF);
        sync_tile(t, buf);
        const T* ldy = lds + buf * BUF;
        const T* lx = ldy + TR * N;
        const int half = lane >> 5, col = lane & 31;
        if constexpr (BF) {
            // bf16 operands: the fragment of lane (column, half) is the 8 consecutive ROWS 16 s + 8 half + j of
            // its column -- eight 2-byte LDS reads (32 lanes cover 64 contiguous bytes: conflict free), one MFMA
            // per tile pair and 16-row step
            const unsigned short* uy = reinterpret_cast<const unsigned short*>(ldy);
            const unsigned short* ux = reinterpret_cast<const unsigned short*>(lx);
#pragma unroll
            for (int s16 = 0; s16 < TR / 16; ++s16) {
                bf16x8 af[TN], bfg[TK];
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    u32x4_t pk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int o0 = (16 * s16 + 8 * half + 2 * j) * N + (wn * TN + i) * 32 + col;
                        unsigned lo = uy[o0], hi = uy[o0 + N];
                        if (MASK) {
                            const unsigned short m0 = ux[TR * K + o0], m1 = ux[TR * K + o0 + N];
                            // keep where the saved activation is > 0 (positive, non-zero bf16)
                            lo = (m0 != 0 && !(m0 & 0x8000u)) ? lo : 0u;
                            hi = (m1 != 0 && !(m1 & 0x8000u)) ? hi : 0u;
                        }
                        pk[j] = lo | (hi << 16);
                    }
                    af[i] = __builtin_bit_cast(bf16x8, pk);
                }
#pragma unroll
                for (int j2 = 0; j2 < TK; ++j2) {
                    u32x4_t pk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int o0 = (16 * s16 + 8 * half + 2 * j) * K + (wk * TK + j2) * 32 + col;
                        pk[j] = static_cast<unsigned>(ux[o0]) | (static_cast<unsigned>(ux[o0 + K]) << 16);
                    }
                    bfg[j2] = __builtin_bit_cast(bf16x8, pk);
                }
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j2 = 0; j2 < TK; ++j2)
                        acc[i][j2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfg[j2], acc[i][j2], 0, 0, 0);
            }
        } else
        if constexpr (SPLIT == 1) {
#pragma unroll
            for (int s16 = 0; s16 < TR / 16; ++s16) {
                bf16x8 af[TN][3], bfg[TK][3];
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int o = (16 * s16 + 8 * half + j) * N + (wn * TN + i) * 32 + col;
                        v[j] = ldy[o];
                        if (MASK) v[j] = lx[TR * K + o] > 0.f ? v[j] : 0.f;
                    }
                    split8(v, af[i][0], af[i][1], af[i][2]);
                }
#pragma unroll
                for (int j2 = 0; j2 < TK; ++j2) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = lx[(16 * s16 + 8 * half + j) * K + (wk * TK + j2) * 32 + col];
                    split8(v, bfg[j2][0], bfg[j2][1], bfg[j2][2]);
                }
                constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};   // smallest terms first
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < TN; ++i)
#pragma unroll
                        for (int j2 = 0; j2 < TK; ++j2)
                            acc[i][j2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][TA[t]], bfg[j2][TB[t]], acc[i][j2], 0, 0, 0);
            }
        } else
#pragma unroll
        for (int ks = 0; ks < TR / 2; ++ks) {
            float a[TN], b[TK];
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int o = (2 * ks + half) * N + (wn * TN + i) * 32 + col;
                a[i] = ldy[o];
                if (MASK) a[i] = lx[TR * K + o] > 0.f ? a[i] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < TK; ++j) b[j] = lx[(2 * ks + half) * K + (wk * TK + j) * 32 + col];
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TK; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (part_b) bias_tile(buf);
    }
    // partial tile -> workspace
    float* pw = part_w + static_cast<size_t>(blockIdx.x) * N * K;
    const int half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
                float val = acc[i][j][reg];
                if constexpr (SPLIT == 2 && !BF) {      // un-scale: two exact multiplications (each factor within fp32 range)
                    const float iy = pow2_inv(scy[i]);
                    val = val * __int_as_float(__builtin_amdgcn_ds_bpermute(row * 4, __float_as_int(iy)));
                    val = val * pow2_inv(scx[j]);
                }
                pw[((wn * TN + i) * 32 + row) * K + (wk * TK + j) * 32 + col] = val;
            }
    if (part_b) {
        // combine the two row halves in a fixed order through LDS
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_raw);
        if (threadIdx.x < N * H) red[threadIdx.x] = bacc;
        __syncthreads();
        if (threadIdx.x < N)
            part_b[static_cast<size_t>(blockIdx.x) * N + threadIdx.x] =
                H == 2 ? red[threadIdx.x] + red[N + threadIdx.x] : red[threadIdx.x];
    }
}

// out[i] = sum_s part[s][i] in a fixed order.  Block = 16 partial-groups x 16 float4 columns:
// group g sums partials g, g+16, ... (independent 16 B loads, unrolled), then the 16 group
// sums are added in order through LDS.
// Two outputs in one launch (dW and db): blocks [0, blocks_a) reduce `part`, the rest `part_b`.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, int64_t n4,
                                                          float* __restrict__ out, int blocks_a,
                                                          const float* __restrict__ part_b, int64_t n4_b,
                                                          float* __restrict__ out_b) {
    __shared__ float4 red[16][17];
    const int g = threadIdx.x >> 4, c = threadIdx.x & 15;
    int64_t blk = blockIdx.x;
    if (blk >= blocks_a) {   // block-uniform
        blk -= blocks_a;
        part = part_b;
        n4 = n4_b;
        out = out_b;
    }
    const int64_t i = blk * 16 + c;
    float4 s = f4(0.f);
    if (i < n4) {
#pragma unroll 8
        for (int p = g; p < S; p += 16) s += ld4(part + (static_cast<size_t>(p) * n4 + i) * 4);
    }
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && i < n4) {
        float4 t = red[0][c];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][c];
        st4(out + i * 4, t);
    }
}

// The fixed-order reduce of up to 8 weight gradients in ONE launch (the projections of an attention block, fc1 + fc2 of
// a feed-forward block: their split-K kernels run back to back into separate workspaces): blockIdx.y = entry.
struct RedEntry {
    const float* part;
    const float* part_b;
    float* out;
    float* out_b;
    long long n4, n4_b;
    int S, blocks_a, blocks;
};
struct RedBatch {
    RedEntry e[8];
};
__global__ __launch_bounds__(256) void splitk_reduce_multi_kernel(const RedBatch b) {
    __shared__ float4 red[16][17];
    const RedEntry& en = b.e[blockIdx.y];
    if (static_cast<int>(blockIdx.x) >= en.blocks) return;      // block-uniform
    const int g = threadIdx.x >> 4, c = threadIdx.x & 15;
    int64_t blk = blockIdx.x;
    const float* part = en.part;
    float* out = en.out;
    long long n4 = en.n4;
    if (blk >= en.blocks_a) {
        blk -= en.blocks_a;
        part = en.part_b;
        n4 = en.n4_b;
        out = en.out_b;
    }
    const int64_t i = blk * 16 + c;
    float4 s = f4(0.f);
    if (i < n4) {
#pragma unroll 8
        for (int p = g; p < en.S; p += 16) s += ld4(part + (static_cast<size_t>(p) * n4 + i) * 4);
    }
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && i < n4) {
        float4 t = red[0][c];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][c];
        st4(out + i * 4, t);
    }
}

// ---- skinny outputs: N <= 16 rows of dW (readout_e / readout_n, reference models.py:67-68) -----
// dW[n][k] = sum_r dy[r][n] x[r][k] with N = 5 / 13: no MFMA tile to fill, the kernel is a coalesced
// stream over x (512 B rows) with N float4 accumulators per lane; dy[r][.] is a broadcast load.
// Thread = (row phase, k quad); row phases are combined through LDS in a fixed order.
template <typename TY, typename T, int NMAX>
__global__ __launch_bounds__(256) void skinny_wgrad_kernel(const TY* __restrict__ dy, const T* __restrict__ x,
                                                         float* __restrict__ part_w, float* __restrict__ part_b,
                                                         int64_t R, int N, int K, int64_t rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float4* red = reinterpret_cast<float4*>(smem_raw);   // [phases][NMAX][KQ]
    const int KQ = K / 4;
    const int phases = 256 / KQ;
    const int kq = threadIdx.x % KQ, rp = threadIdx.x / KQ;
    const int64_t r_lo = static_cast<int64_t>(blockIdx.x) * rows_per_block;
    int64_t r_hi = r_lo + rows_per_block;
    if (r_hi > R) r_hi = R;
    float4 acc[NMAX];
    float bsum[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
        acc[n] = f4(0.f);
        bsum[n] = 0.f;
    }
    if (rp < phases) {
        for (int64_t r = r_lo + rp; r < r_hi; r += phases) {
            const float4 xv = ld4(x + r * K + kq * 4);
            const TY* dr = dy + r * N;
#pragma unroll
            for (int n = 0; n < NMAX; ++n) {
                const float d = n < N ? ld1(dr + n) : 0.f;
                acc[n] = fma4(f4(d), xv, acc[n]);
                bsum[n] += d;
            }
        }
#pragma unroll
        for (int n = 0; n < NMAX; ++n) red[(rp * NMAX + n) * KQ + kq] = acc[n];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < N * KQ; idx += 256) {
        const int n = idx / KQ, q = idx % KQ;
        float4 t = f4(0.f);
        for (int p = 0; p < phases; ++p) t += red[(p * NMAX + n) * KQ + q];
        st4(part_w + (static_cast<size_t>(blockIdx.x) * N * K) + static_cast<size_t>(n) * K + q * 4, t);
    }
    if (part_b) {
        __syncthreads();
        float* rb = reinterpret_cast<float*>(smem_raw);   // [phases][NMAX]
        if (rp < phases && kq == 0)
#pragma unroll
            for (int n = 0; n < NMAX; ++n) rb[rp * NMAX + n] = bsum[n];
        __syncthreads();
        if (threadIdx.x < N) {
            float t = 0.f;
            for (int p = 0; p < phases; ++p) t += rb[p * NMAX + threadIdx.x];
            part_b[static_cast<size_t>(blockIdx.x) * N + threadIdx.x] = t;
        }
    }
}

constexpr int kSkinnyBlocks = 512;

bool skinny_ok(int N, int K) { return N >= 1 && N <= 16 && K >= 4 && K % 4 == 0 && K <= 1024 && 256 % (K / 4) == 0; }

// profiler key: edge-level launches by shape, everything else together
int wgrad_prof_key(int64_t R, int N, int K) {
    if (R < edge_rows()) return DG_K_LINEAR_WGRAD;
    if (N == 128 && K == 128) return DG_K_LINEAR_WGRAD_E_128;
    if (N == 384 && K == 128) return DG_K_LINEAR_WGRAD_E_N384;
    if (N == 128 && K == 384) return DG_K_LINEAR_WGRAD_E_K384;
    return DG_K_LINEAR_WGRAD;
}

struct WgradPlan {
    int nt, kt, wn, wk, tr, threads, lds, lds_mask;
};

bool wgrad_plan(int N, int K, WgradPlan* p) {
    if (N % 32 || K % 32 || N < 32 || K < 32) return false;
    p->nt = N / 32;
    p->kt = K / 32;
    struct Row {
        int nt, kt, wn, wk, tr;
    };
    constexpr bool big = false;
    static const Row table[] = {
        {4, 4, 2, 2, big ? 64 : 32},   // 128 x 128   (q,k,v,e,out_e,out_n)
        {12, 4, 4, 2, big ? 32 : 16},  // 384 x 128   (fc1: dW[3C, C])
        {4, 12, 2, 4, big ? 32 : 16},  // 128 x 384   (fc2: dW[C, 3C])
        {4, 2, 2, 2, 32},   // 128 x 64    (embedding layer 2: Linear(64, C))
        {2, 2, 2, 2, 32},   // 64 x 64
        {1, 1, 1, 1, 32},   // 32 x 32     (tiny test models)
        {2, 1, 2, 1, 32},   // 64 x 32
        {1, 2, 1, 2, 32},   // 32 x 64
        {3, 1, 3, 1, 32},   // 96 x 32
        {1, 3, 1, 3, 32},   // 32 x 96
    };
    for (const Row& r : table)
        if (r.nt == p->nt && r.kt == p->kt) {
            p->wn = r.wn;
            p->wk = r.wk;
            p->tr = r.tr;
            p->threads = r.wn * r.wk * 64;
            p->lds = 2 * r.tr * (N + K) * 4;
            p->lds_mask = 2 * r.tr * (2 * N + K) * 4;
            return true;
        }
    return false;
}

// `ring`: the launch keeps a 3-4 deep tile ring in LDS (fp32, 384-wide shapes): one block per CU.  The workspace query
// passes false (more blocks -> the larger workspace serves both).
int wgrad_blocks(int64_t R, const WgradPlan& p, int* tiles_per_block, bool ring = false) {
    const int64_t tiles = (R + p.tr - 1) / p.tr;
    const int per_cu = (ring || p.lds > 64 * 1024) ? 1 : 2;
    int64_t target = 256 * per_cu;
    if (target > tiles) target = tiles < 1 ? 1 : tiles;
    const int64_t tpb = (tiles + target - 1) / target;
    *tiles_per_block = static_cast<int>(tpb < 1 ? 1 : tpb);
    return static_cast<int>((tiles + *tiles_per_block - 1) / *tiles_per_block);
}

}  // namespace
}  // namespace dg

namespace dg {
// out[i] = sum_s part[s][i] over float4 columns, fixed order (shared with ffn_bf16.hip)
void launch_splitk_reduce(const float* part, int S, int64_t n4, float* out, hipStream_t stream) {
    const int blocks = static_cast<int>((n4 + 15) / 16);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, part, S, n4, out, blocks,
                       static_cast<const float*>(nullptr), static_cast<int64_t>(0), static_cast<float*>(nullptr));
}
}  // namespace dg

using namespace dg;

extern "C" size_t dg_linear_wgrad_workspace_bytes(int64_t R, int N, int K) {
    WgradPlan p;
    if (R >= 1 && skinny_ok(N, K)) return static_cast<size_t>(kSkinnyBlocks) * (static_cast<size_t>(N) * K + N) * sizeof(float);
    if (R < 1 || !wgrad_plan(N, K, &p)) return 0;
    int tpb;
    int S = wgrad_blocks(R, p, &tpb);
    if (wgrad_stream_supported(N, K) && wgrad_stream_blocks(R, N, K) > S) S = wgrad_stream_blocks(R, N, K);
    return static_cast<size_t>(S) * (static_cast<size_t>(N) * K + N) * sizeof(float);
}

// ---- deferred reduces: between dg_linear_wgrad_batch_begin() and _end() a (non-skinny) dg_linear_wgrad only runs its
// split-K kernel and records its reduce; _end() runs them all in one launch.  Every call of a batch needs its OWN
// workspace.  Thread-local: one batch per host thread.
static thread_local bool g_batch_on = false;
static thread_local int g_batch_n = 0;
static thread_local RedBatch g_batch;

extern "C" int dg_linear_wgrad_batch_begin(void) {
    g_batch_on = true;
    g_batch_n = 0;
    return 0;
}
extern "C" int dg_linear_wgrad_batch_end(dg_stream_t stream_) {
    const int n = g_batch_n;
    g_batch_on = false;
    g_batch_n = 0;
    // (a weight gradient still waiting for a carrier -- pair.h -- must be on the stream before its partial sums are reduced)
    if (int st = flush_wgrad_stream(static_cast<hipStream_t>(stream_))) return st;
    if (n == 0) return 0;
    int maxb = 0;
    for (int i = 0; i < n; ++i) maxb = g_batch.e[i].blocks > maxb ? g_batch.e[i].blocks : maxb;
    hipLaunchKernelGGL(splitk_reduce_multi_kernel, dim3(maxb, n), dim3(256), 0, static_cast<hipStream_t>(stream_), g_batch);
    return check_launch("dg_linear_wgrad_batch_end");
}

// Other fixed-order partial-sum reductions (LayerNorm's dgamma / dbeta: part[S][n] -> out[n]) can ride in the same launch.
namespace dg {
bool reduce_batch_try_add(const float* part, int S, long long n_floats, float* out) {
    if (!g_batch_on || g_batch_n >= 8 || (n_floats & 3)) return false;
    RedEntry& en = g_batch.e[g_batch_n++];
    en.part = part; en.part_b = nullptr; en.out = out; en.out_b = nullptr;
    en.n4 = n_floats / 4; en.n4_b = 0; en.S = S;
    en.blocks_a = static_cast<int>((en.n4 + 15) / 16);
    en.blocks = en.blocks_a;
    return true;
}
}  // namespace dg

// N <= 16 output rows: dy (float32 or bf16) and x (float32 or bf16) may differ -- the readout's logits are float32 in
// the bf16 configuration too
static int skinny_wgrad(const void* dy_, bool dy_f32, const void* x_, bool x_bf, float* dw, float* db, void* workspace,
                        size_t workspace_bytes, int64_t R, int N, int K, dg_stream_t stream_) {
    if (workspace_bytes < dg_linear_wgrad_workspace_bytes(R, N, K))
        return fail(DG_E_WORKSPACE, "dg_linear_wgrad: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int S = kSkinnyBlocks;
    int64_t rpb = (R + S - 1) / S;
    if (rpb < 64) rpb = 64;
    S = static_cast<int>((R + rpb - 1) / rpb);
    float* part_w = static_cast<float*>(workspace);
    float* part_b = db ? part_w + static_cast<size_t>(S) * N * K : nullptr;
    const int KQ = K / 4, phases = 256 / KQ;
    ProfScope prof(DG_K_LINEAR_WGRAD, stream);
#define SKINNY(TY, T, NM)                                                                                         \
    hipLaunchKernelGGL((skinny_wgrad_kernel<TY, T, NM>), dim3(S), dim3(256), phases * NM * KQ * 16, stream,       \
                       static_cast<const TY*>(dy_), static_cast<const T*>(x_), part_w, part_b, R, N, K, rpb);
#define SKINNY_N(TY, T) { if (N <= 8) { SKINNY(TY, T, 8) } else { SKINNY(TY, T, 16) } }
    if (dy_f32 && x_bf) SKINNY_N(float, bf16_t)
    else if (x_bf) SKINNY_N(bf16_t, bf16_t)
    else SKINNY_N(float, float)
#undef SKINNY_N
#undef SKINNY
    const int64_t nw = static_cast<int64_t>(N) * K;
    // column sums of the [S][N K] / [S][N] partials: 32 row groups per column block, fixed order (one thread per element walking
    // all S partials was 14 us of dependent loads)
    launch_ln_finish(part_w, S, 1, static_cast<int>(nw), dw, nullptr, stream);
    if (db) launch_ln_finish(part_b, S, 1, N, db, nullptr, stream);
    return check_launch("dg_linear_wgrad(skinny)");
}

/* The readout's weight gradient with float32 logit gradients and activations of `dtype` (models.py:67-68 backward). */
extern "C" int dg_skinny_linear_wgrad(const float* dy, const void* x, float* dw, float* db, void* workspace,
                                      size_t workspace_bytes, int64_t R, int N, int K, int dtype, dg_stream_t stream_) {
    if (!dy || !x || !dw || !workspace) return fail(DG_E_ARG, "dg_skinny_linear_wgrad: null pointer");
    if (!dtype_ok(dtype)) return fail(DG_E_ARG, "dg_skinny_linear_wgrad: unknown dtype %d", dtype);
    if (R < 1 || !skinny_ok(N, K)) return fail(DG_E_SHAPE, "dg_skinny_linear_wgrad: unsupported shape R=%lld N=%d K=%d", (long long)R, N, K);
    return skinny_wgrad(dy, true, x, dtype == DG_DTYPE_BF16, dw, db, workspace, workspace_bytes, R, N, K, stream_);
}

extern "C" int dg_linear_wgrad(const void* dy_, const void* dy_mask_, const void* x_, float* dw, float* db,
                               void* workspace, size_t workspace_bytes, int64_t R, int N, int K, int dtype,
                               dg_stream_t stream_) {
    if (!dy_ || !x_ || !dw || !workspace) return fail(DG_E_ARG, "dg_linear_wgrad: null pointer");
    // DG_DTYPE_F32_H16: the 384-wide operand (dy for N = 384, x for K = 384) is an fp16 plane + inverse row scales
    const float* hscale = nullptr;
    int hfmt = 0;
    if (dtype == DG_DTYPE_F32_H16 || dtype == DG_DTYPE_F32_H24 || dtype == DG_DTYPE_F32_H32) {
        const bool narrow = (N == 384 && K == 128) || (N == 128 && K == 384);
        if (narrow) hfmt = hidden_fmt(dtype);
        // H32 (hi | lo | scales): the weight gradient reads the hi plane only -- an fp16 plane whose scales sit behind the lo plane
        const size_t soff = (hfmt == 3 ? 2 : 1) * hidden_scale_offset(R, 384);
        if (hfmt == 3) hfmt = 1;
        if (hfmt == 1) hscale = reinterpret_cast<const float*>(static_cast<const char*>(N == 384 ? dy_ : x_) + soff);
        if (hfmt && dy_mask_) return fail(DG_E_ARG, "dg_linear_wgrad: dy_mask cannot be combined with DG_DTYPE_F32_H16 / _H24");
        dtype = DG_DTYPE_F32;
    }
    if (!dtype_ok(dtype)) return fail(DG_E_ARG, "dg_linear_wgrad: unknown dtype %d", dtype);
    const bool bf = dtype == DG_DTYPE_BF16;
    if (R >= 1 && skinny_ok(N, K)) {
        if (dy_mask_) return fail(DG_E_ARG, "dg_linear_wgrad: dy_mask is not supported for N <= 16");
        return skinny_wgrad(dy_, !bf, x_, bf, dw, db, workspace, workspace_bytes, R, N, K, stream_);
    }
    WgradPlan p;
    if (R < 1 || !wgrad_plan(N, K, &p))
        return fail(DG_E_SHAPE, "dg_linear_wgrad: unsupported shape R=%lld N=%d K=%d", (long long)R, N, K);
    if (workspace_bytes < dg_linear_wgrad_workspace_bytes(R, N, K))
        return fail(DG_E_WORKSPACE, "dg_linear_wgrad: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // fp32 operands: fp16 hi + lo with running column scales (the bf16x6 split and the plain fp32 MFMA of rounds 1 - 2 are no
    // longer selectable); the three encoder shapes run on the producer / consumer kernel (wgrad_stream.hip), every element
    // converted once per workgroup, the other shapes on the symmetric kernel below
    constexpr int split = 2;
    constexpr bool stream_kernel = true;
    const bool use_stream = !bf && split == 2 && stream_kernel && !dy_mask_ && wgrad_stream_supported(N, K);
    if (hfmt && !use_stream)
        return fail(DG_E_ARG, "dg_linear_wgrad: DG_DTYPE_F32_H16 / _H24 needs one of the producer / consumer weight-gradient shapes");
    int tpb;
    const bool may_wait = g_batch_on && g_batch_n < 8;      // (a launch may only wait for a carrier when its reduce is deferred too)
    const int S = use_stream ? wgrad_stream_blocks(R, N, K, may_wait)
                             : wgrad_blocks(R, p, &tpb, !bf && split != 0 && p.nt + p.kt == 16 && p.tr % 16 == 0);
    float* part_w = static_cast<float*>(workspace);
    float* part_b = db ? part_w + static_cast<size_t>(S) * N * K : nullptr;
    const bool big_tiles = (p.nt == 4 && p.kt == 4) ? p.tr == 64 : p.tr == 32;
    ProfScope prof(wgrad_prof_key(R, N, K), stream);
    note_forward(R);
    if (use_stream) {
        if (int st = launch_wgrad_stream(dy_, x_, part_w, part_b, R, N, K, S, stream, nullptr, nullptr, may_wait, hscale, hfmt))
            return st;
    } else {
#define LAUNCH_X(T, NT_, KT_, WN_, WK_, TR_, M_, X_)                                                              \
    {                                                                                                            \
        constexpr int tile_bytes = TR_ * ((M_ ? 2 : 1) * NT_ + KT_) * 32 * static_cast<int>(sizeof(T));           \
        /* fp32 384-wide shapes: one block per CU (registers), so a deeper ring instead of a second block */     \
        constexpr int nbuf = (sizeof(T) == 4 && X_ != 0 && NT_ + KT_ == 16) ? (4 * tile_bytes <= 131072 ? 4 : (3 * tile_bytes <= 131072 ? 3 : 2)) : 2; \
        constexpr int lds_bytes = nbuf * tile_bytes;                                                              \
        DG_OPT_IN_LDS((&wgrad_kernel<T, NT_, KT_, WN_, WK_, TR_, M_, X_, nbuf>), lds_bytes);                      \
        hipLaunchKernelGGL((wgrad_kernel<T, NT_, KT_, WN_, WK_, TR_, M_, X_, nbuf>), dim3(S), dim3(WN_* WK_ * 64), \
                           lds_bytes, stream, static_cast<const T*>(dy_), static_cast<const T*>(dy_mask_),       \
                           static_cast<const T*>(x_), part_w, part_b, R, tpb);                                   \
    }
#define LAUNCH_M(NT_, KT_, WN_, WK_, TR_, M_)                                            \
    {                                                                                   \
        if (bf) LAUNCH_X(bf16_t, NT_, KT_, WN_, WK_, TR_, M_, 0)                         \
        else if (split == 2 && (TR_) % 16 == 0) LAUNCH_X(float, NT_, KT_, WN_, WK_, TR_, M_, 2) \
        else if (split == 1 && (TR_) % 16 == 0) LAUNCH_X(float, NT_, KT_, WN_, WK_, TR_, M_, 1) \
        else LAUNCH_X(float, NT_, KT_, WN_, WK_, TR_, M_, 0)                             \
    }
#define LAUNCH(NT_, KT_, WN_, WK_, TR_)                                   \
    if (p.nt == NT_ && p.kt == KT_) {                                     \
        if (dy_mask_) LAUNCH_M(NT_, KT_, WN_, WK_, TR_, true)             \
        else LAUNCH_M(NT_, KT_, WN_, WK_, TR_, false)                     \
    }
    if (big_tiles) {
        LAUNCH(4, 4, 2, 2, 64)
        LAUNCH(12, 4, 4, 2, 32)
        LAUNCH(4, 12, 2, 4, 32)
    } else {
        LAUNCH(4, 4, 2, 2, 32)
        LAUNCH(12, 4, 4, 2, 16)
        LAUNCH(4, 12, 2, 4, 16)
    }
    LAUNCH(4, 2, 2, 2, 32)
    LAUNCH(2, 2, 2, 2, 32)
    LAUNCH(1, 1, 1, 1, 32)
    LAUNCH(2, 1, 2, 1, 32)
    LAUNCH(1, 2, 1, 2, 32)
    LAUNCH(3, 1, 3, 1, 32)
    LAUNCH(1, 3, 1, 3, 32)
#undef LAUNCH_M
#undef LAUNCH_X
#undef LAUNCH
    }
    const int64_t n4 = static_cast<int64_t>(N) * K / 4;
    const int blocks_w = static_cast<int>((n4 + 15) / 16), blocks_b = db ? (N / 4 + 15) / 16 : 0;
    if (g_batch_on && g_batch_n < 8) {      // the reduce joins the batch (dg_linear_wgrad_batch_end)
        RedEntry& en = g_batch.e[g_batch_n++];
        en.part = part_w; en.part_b = part_b; en.out = dw; en.out_b = db;
        en.n4 = n4; en.n4_b = N / 4; en.S = S; en.blocks_a = blocks_w; en.blocks = blocks_w + blocks_b;
        return check_launch("dg_linear_wgrad");
    }
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks_w + blocks_b), dim3(256), 0, stream, part_w, S, n4, dw, blocks_w,
                       part_b, static_cast<int64_t>(N / 4), db);
    return check_launch("dg_linear_wgrad");
}

/* dW [384,128] = [dy0 | dy1 | dy2]^T x, db [384]: the weight gradients of three Linear(128,128) that share their input
 * (q / k / v of an attention block, reference layers.py:111-113 backward) in ONE launch.  float32; workspace as for
 * dg_linear_wgrad(R, 384, 128); joins a dg_linear_wgrad_batch like any other call.                                  */
extern "C" int dg_linear_wgrad3(const void* dy0, const void* dy1, const void* dy2, const void* x, float* dw, float* db,
                                void* workspace, size_t workspace_bytes, int64_t R, int dtype, dg_stream_t stream_) {
    if (!dy0 || !dy1 || !dy2 || !x || !dw || !workspace) return fail(DG_E_ARG, "dg_linear_wgrad3: null pointer");
    if (dtype != DG_DTYPE_F32) return fail(DG_E_SHAPE, "dg_linear_wgrad3: float32 activations only");
    if (R < 1) return fail(DG_E_SHAPE, "dg_linear_wgrad3: R = %lld", (long long)R);
    constexpr int N = 384, K = 128;
    if (workspace_bytes < dg_linear_wgrad_workspace_bytes(R, N, K)) return fail(DG_E_WORKSPACE, "dg_linear_wgrad3: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool may_wait = g_batch_on && g_batch_n < 8;
    const int S = wgrad_stream_blocks(R, N, K, may_wait);
    float* part_w = static_cast<float*>(workspace);
    float* part_b = db ? part_w + static_cast<size_t>(S) * N * K : nullptr;
    {
        ProfScope prof(wgrad_prof_key(R, N, K), stream);
        note_forward(R);
        if (int st = launch_wgrad_stream(static_cast<const float*>(dy0), static_cast<const float*>(x), part_w, part_b, R, N, K, S,
                                         stream, static_cast<const float*>(dy1), static_cast<const float*>(dy2), may_wait))
            return st;
    }
    const int64_t n4 = static_cast<int64_t>(N) * K / 4;
    const int blocks_w = static_cast<int>((n4 + 15) / 16), blocks_b = db ? (N / 4 + 15) / 16 : 0;
    if (g_batch_on && g_batch_n < 8) {
        RedEntry& en = g_batch.e[g_batch_n++];
        en.part = part_w; en.part_b = part_b; en.out = dw; en.out_b = db;
        en.n4 = n4; en.n4_b = N / 4; en.S = S; en.blocks_a = blocks_w; en.blocks = blocks_w + blocks_b;
        return check_launch("dg_linear_wgrad3");
    }
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks_w + blocks_b), dim3(256), 0, stream, part_w, S, n4, dw, blocks_w,
                       part_b, static_cast<int64_t>(N / 4), db);
    return check_launch("dg_linear_wgrad3");
}
