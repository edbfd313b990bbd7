// Attention half of an Encoder_Block's EDGE branch as ONE kernel per direction -- reference
// src/model/layers.py:116-135 (MHA.forward: e-projection, Hadamard score, softmax over j, AV) and
// :186-190 (edge residual + LayerNorm ln4):
//
//     e   = y We^T + be                                   [N, C] rows j of one (molecule b, query row i)
//     s   = alpha q_i k_j (e^2 + e)          p = softmax_j s          o_i = sum_j p v_j
//     y2  = LN4( y + s Woe^T + boe )
//
// The unfused path runs this as four edge-level launches and eight passes over [B,N,N,C] tensors
// (read y, write e | read e, write s | read s, read y, write y2, write pre); here the tile of one (b, i) -- N rows
// of C channels -- is read once and `e`, `s` never exist in HBM: read y, write y2 + pre (the pre-LayerNorm sum the
// backward needs) = 3 passes.  The softmax runs over j for a fixed (b, i, c), so the working set of a tile is ONE
// [N, C] row block plus the molecule's k, v.
//
// MI355X mapping (bf16 activations; fp32 accumulation / softmax / LayerNorm statistics)
//   * workgroup = 4 waves, persistent over a contiguous range of tiles (molecule b, query row i); wave w owns the 32
//     channels [32 w, 32 w + 32) of both products; weights We / Woe are resident MFMA fragments (P16 order,
//     gemm_bf16.h: 64 VGPRs);
//   * the y tile arrives by LDS-DMA (double buffered, XOR swizzle on the source address);
//   * e = y We^T on v_mfma_f32_16x16x32_bf16, NOT swapped: a lane ends up with channel c = 32 w + 16 nb + (lane & 15)
//     and the rows j = 16 mb + 4 (lane >> 4) + r -- i.e. with 4 MB of the N neighbours of ONE channel, which is the
//     layout the per-channel softmax wants: the max / sum / AV reductions are in-lane loops plus two xor steps over
//     lane bits 4 and 5 (v_permlane16_swap / v_permlane32_swap), no LDS, no barrier; k_j, v_j of those (j, c) pairs
//     stay in registers while the workgroup walks the query rows of its molecule;
//   * s goes to an LDS tile as bf16 (row-major, 2-byte stores) and s Woe^T runs swapped (weights = A operand): a
//     lane gets 4 consecutive channels of one row -> fp32 exchange tile -> row phase: 16 lanes per row, residual from
//     the y tile in LDS, LayerNorm with DPP row sums, whole 256-byte rows out as 16-byte stores.
// Algorithmic bytes per launch: 2 B N^2 C (1 + 2 [edge]) + node-level terms.
#include "gemm_bf16.h"

namespace dg {

int pack_bf16(const float* w, void* packed, int rows, int cols, int mode, int mb_size, hipStream_t stream);
void launch_splitk_reduce(const float* part, int S, int64_t n4, float* out, hipStream_t stream);

namespace {

constexpr int kC = 128;
constexpr int kSec = 8 * 4 * 64;              // bf16x8 entries of one P16-packed 128 x 128 weight (32 KB)
constexpr int kSecE = 0;                      // We          (e = y We^T)
constexpr int kSecOE = kSec;                  // Woe         (s Woe^T)
constexpr int kSecOET = 2 * kSec;             // Woe^T       (ds = dz4 Woe)
constexpr int kSecET = 3 * kSec;              // We^T        (dy = de We)
constexpr float kNegBig = -3.0e38f;

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float vmax(float a, float b) {   // plain v_max_f32 (fmaxf adds a canonicalising v_max per operand)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float xor_vmax16(float x) {     // max over lane bits 4 and 5
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = vmax(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return vmax(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// Workgroup barrier that orders LDS traffic only: the kernels below wait for their LDS-DMA BEFORE issuing a tile's
// global stores (the DMA of the next tile was issued a whole tile earlier, so that wait is free), and must not sit on
// `vmcnt(0)` at the next barrier until those stores have been acknowledged (PMC: waves parked > 50 % of their cycles).
// (The two empty asm statements are COMPILER barriers: s_barrier is IntrNoMem for LLVM, which is otherwise free to move
// LDS loads / stores across it.)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0); vmcnt / expcnt untouched
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
constexpr float kLog2e = 1.4426950408889634f;
// two-wide fp32 arithmetic: hipcc maps <2 x float> mul / add / fma to v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 (one
// issue slot for two lanes' worth of work) -- the fused kernels are VALU-issue bound, not MFMA or HBM bound
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 lo2(f32x4 v) { return f32x2{v[0], v[1]}; }
__device__ __forceinline__ f32x2 hi2(f32x4 v) { return f32x2{v[2], v[3]}; }
__device__ __forceinline__ f32x2 bc2(float x) { return f32x2{x, x}; }
__device__ __forceinline__ f32x2 unpack2_bf16(unsigned w) { return f32x2{lo_bf16(w), hi_bf16(w)}; }
__device__ __forceinline__ float sum16(float x) {      // sum over the 16 lanes of a DPP row, result in all 16
    x = dpp_add<0xB1>(x);
    x = dpp_add<0x4E>(x);
    x = dpp_add<0x141>(x);
    x = dpp_add<0x140>(x);
    return x;
}
__device__ __forceinline__ u32x4_t pack8_bf16(float4 a, float4 b) {
    u32x4_t o;
    o[0] = pack_bf16(a.x, a.y); o[1] = pack_bf16(a.z, a.w);
    o[2] = pack_bf16(b.x, b.y); o[3] = pack_bf16(b.z, b.w);
    return o;
}

// LDS-DMA of the `nrows` (<= 16 MB) rows of one [N, 128] bf16 row block into a swizzled tile; 4 waves take part.
template <int MB>
__device__ __forceinline__ void dma_rows_bf16(const bf16_t* __restrict__ src, int nrows, char* lds_dst, int wave, int lane) {
    const unsigned dst = lds_byte_address(lds_dst);
#pragma unroll
    for (int t = 0; t < MB; ++t) {
        const int ii = wave + 4 * t;
        const int L = ii * 64 + lane;
        const int row = L >> 4, cpos = L & 15;
        if (row < nrows)
            dma16_async(reinterpret_cast<const float*>(src + row * kC + ((cpos ^ (row & 15)) << 3)), dst + ii * 1024);
    }
}

struct HalfFwdArgs {
    const bf16_t* y;       // [B,N,N,C]
    const bf16_t* q;       // [B,N,C]
    const bf16_t* k;
    const bf16_t* v;
    const bf16x8* pk;      // dg_attn_half_pack
    const float* be;
    const float* boe;
    const float* gamma;
    const float* beta;
    bf16_t* o;             // [B,N,C]
    bf16_t* y2;            // [B,N,N,C]
    bf16_t* pre;           // [B,N,N,C]
    float* mean;           // [B N N]
    float* rstd;
    int B, N, abl;
    float alpha, eps;
};

template <int MB, bool EDGE>
__global__ __launch_bounds__(256, (MB <= 3 ? 2 : 1)) void attn_half_fwd_bf16_kernel(const HalfFwdArgs a) {
    constexpr int ROWS = 16 * MB;
    constexpr int YB = ROWS * 256;            // bf16 tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ybuf = smem;                        // [2][ROWS][128] bf16
    char* st = smem + 2 * YB;                 // [ROWS][128] bf16: s
    char* xch = st + YB;                      // [ROWS][128] fp32 exchange tile
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int N = a.N;

    bf16x8 wfe[2][4], wfo[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            wfe[nb][ks] = a.pk[kSecE + ((2 * w + nb) * 4 + ks) * 64 + lane];
            if (EDGE) wfo[nb][ks] = a.pk[kSecOE + ((2 * w + nb) * 4 + ks) * 64 + lane];
        }
    float be2[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) be2[nb] = a.be[32 * w + 16 * nb + r16];
    float* prm = reinterpret_cast<float*>(xch + ROWS * 512);      // [3][128]: boe, gamma, beta (row phase operands)
    char* qbuf = reinterpret_cast<char*>(prm + 3 * 128);          // [2][128] bf16: q_i of the tile in flight / the next one
    if (EDGE && threadIdx.x < 128) {
        prm[threadIdx.x] = a.boe[threadIdx.x];
        prm[128 + threadIdx.x] = a.gamma[threadIdx.x];
        prm[256 + threadIdx.x] = a.beta[threadIdx.x];
    }
    // rows >= N of the y tiles are never written by the DMA: zero them once (their products are finite and unused)
    for (int idx = threadIdx.x; idx < (ROWS - N) * 16 * 2; idx += 256) {
        const int bufi = idx / ((ROWS - N) * 16), rem = idx % ((ROWS - N) * 16);
        *reinterpret_cast<float4*>(ybuf + bufi * YB + N * 256 + rem * 16) = f4(0.f);
    }
    // additive neighbour mask: 0 where row j = 16 mb + 4 kq + r exists, -3e38 where it does not (enters the score
    // through an fma, so masking costs no instruction; exp2 of it is exactly 0)
    f32x4 negm[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) negm[mb][r] = (16 * mb + 4 * kq + r < N) ? 0.f : kNegBig;
    // LDS byte offsets that do not depend on the tile
    unsigned af_off[4];                        // A / B fragment of row block 0, k-step ks (row blocks add 16 * 256)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) af_off[ks] = r16 * 256 + (((4 * ks + kq) ^ r16) << 4);
    // s tile store: element (j = 16 mb + 4 kq + r, c = 32 w + 16 nb + r16) lives at j * 256 + ((chunk ^ (j & 15)) << 4) +
    // (c & 7) * 2 with chunk = 4 w + 2 nb + (r16 >> 3): the XOR only mixes lane bits with the parity of r, so two
    // per-lane bases plus immediates cover all 8 MB stores (hipcc otherwise keeps 8 MB hoisted addresses alive).
    unsigned sw_base[2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
        sw_base[x] = kq * (4 * 256) + ((w ^ kq) << 6) + ((((r16 >> 3) ^ x) & 1) << 4) + (r16 & 7) * 2;
    // exchange-tile store (row 16 mb + r16, channels 32 w + 16 nb + 4 kq ..): slot = 8 w + ((4 nb + kq) ^ (r16 & 7))
    unsigned xw_off[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) xw_off[nb] = r16 * 512 + ((8 * w + ((4 * nb + kq) ^ (r16 & 7))) << 4);
    wait_all_vmem_visible();

    // this workgroup's contiguous range of tiles t = b N + i
    const long long tiles = static_cast<long long>(a.B) * N;
    long long t = tiles * blockIdx.x / gridDim.x;
    const long long t_end = tiles * (blockIdx.x + 1) / gridDim.x;
    if (t >= t_end) return;
    int b = static_cast<int>(t / N);
    // k_j, v_j of this lane's (j, c) pairs
    f32x4 kk[MB][2], vv[MB][2];
    auto load_kv = [&](int mol) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int j = 16 * mb + 4 * kq + r;
                    if (j > N - 1) j = N - 1;
                    const size_t off = (static_cast<size_t>(mol) * N + j) * kC + 32 * w + 16 * nb + r16;
                    kk[mb][nb][r] = static_cast<float>(a.k[off]);
                    vv[mb][nb][r] = static_cast<float>(a.v[off]);
                }
    };
    load_kv(b);
    __syncthreads();              // zero rows written
    const unsigned qdst = lds_byte_address(qbuf);
    auto dma_tile = [&](size_t tile, int bufi) {      // y rows of the tile + its q row (16 lanes of wave 3)
        dma_rows_bf16<MB>(a.y + tile * N * kC, N, ybuf + bufi * YB, w, lane);
        if (w == 3 && lane < 16) dma16_async(reinterpret_cast<const float*>(a.q + tile * kC + lane * 8), qdst + bufi * 256);
    };
    dma_tile(static_cast<size_t>(t), 0);
    wait_all_vmem_visible();
    int buf = 0;
    for (;; buf ^= 1, ++t) {
        const size_t node = static_cast<size_t>(t);
        const bool more = t + 1 < t_end;
        const int nb_ = static_cast<int>((t + 1) / N);      // molecule of the next tile
        lds_barrier();            // tile i landed for every wave (each waited for its share before its last stores);
                                  // s / exchange tiles and the other y buffer are free
        if (more) dma_tile(node + 1, buf ^ 1);
        float aq[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
            aq[nb] = a.alpha * static_cast<float>(*reinterpret_cast<const bf16_t*>(qbuf + buf * 256 + (32 * w + 16 * nb + r16) * 2));
        const char* yt = ybuf + buf * YB;
        // ---- e = y We^T: acc[mb][nb][r] = e[j = 16 mb + 4 kq + r][c = 32 w + 16 nb + r16]
        f32x4 acc[MB][2];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = f32x4{be2[nb], be2[nb], be2[nb], be2[nb]};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(yt + mb * (16 * 256) + af_off[ks]);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = mfma16(af, wfe[nb][ks], acc[mb][nb]);
            }
        // ---- score, softmax over j, AV: per channel, in this lane's registers + two xor steps; pairs of neighbours
        // (r, r + 1) go through the packed fp32 instructions
        float mx[2] = {kNegBig, kNegBig};
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const f32x2 e0 = lo2(acc[mb][nb]), e1 = hi2(acc[mb][nb]);
                const f32x2 s0 = (bc2(aq[nb]) * lo2(kk[mb][nb])) * (e0 * e0 + e0) + lo2(negm[mb]);
                const f32x2 s1 = (bc2(aq[nb]) * hi2(kk[mb][nb])) * (e1 * e1 + e1) + hi2(negm[mb]);
                acc[mb][nb] = f32x4{s0[0], s0[1], s1[0], s1[1]};
                mx[nb] = vmax(vmax(mx[nb], s0[0]), vmax(s0[1], vmax(s1[0], s1[1])));
            }
        f32x2 l2[2] = {{0.f, 0.f}, {0.f, 0.f}}, av2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) mx[nb] = -kLog2e * xor_vmax16(mx[nb]);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const f32x2 x0 = lo2(acc[mb][nb]) * bc2(kLog2e) + bc2(mx[nb]);
                const f32x2 x1 = hi2(acc[mb][nb]) * bc2(kLog2e) + bc2(mx[nb]);
                const f32x2 p0 = {__builtin_amdgcn_exp2f(x0[0]), __builtin_amdgcn_exp2f(x0[1])};
                const f32x2 p1 = {__builtin_amdgcn_exp2f(x1[0]), __builtin_amdgcn_exp2f(x1[1])};
                l2[nb] += p0 + p1;
                av2[nb] += p0 * lo2(vv[mb][nb]) + p1 * hi2(vv[mb][nb]);
            }
        float l[2], av[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            l[nb] = xor_sum<16>(l2[nb][0] + l2[nb][1]);
            av[nb] = xor_sum<16>(av2[nb][0] + av2[nb][1]);
        }
        if (kq < 2) {
            const float on = kq == 0 ? av[0] : av[1], ol = kq == 0 ? l[0] : l[1];
            a.o[node * kC + 32 * w + 16 * kq + r16] = static_cast<__bf16>(on * __builtin_amdgcn_rcpf(ol));
        }
        if (more && nb_ != b) load_kv(nb_);      // block-uniform: the next tile belongs to another molecule
        if (EDGE && !(a.abl & 4)) {
            // ---- s -> LDS tile (bf16, row-major, swizzled like the y tile)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<bf16_t*>(st + sw_base[r & 1] + (16 * mb + r) * 256 + ((nb ^ (r >> 1)) << 5)) =
                            static_cast<__bf16>(acc[mb][nb][r]);
            lds_barrier();
            // ---- s Woe^T + boe (swapped): lane = row 16 mb + r16, channels 32 w + 16 nb + 4 kq + {0..3}
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const bf16x8 sf = *reinterpret_cast<const bf16x8*>(st + mb * (16 * 256) + af_off[ks]);
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = mfma16(wfo[nb][ks], sf, acc[mb][nb]);
                }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    *reinterpret_cast<float4*>(xch + mb * (16 * 512) + xw_off[nb]) =
                        make_float4(acc[mb][nb][0], acc[mb][nb][1], acc[mb][nb][2], acc[mb][nb][3]);
            // the next tile's DMA (issued a whole tile ago) and any k / v reload: waited for HERE, before this tile's
            // stores go out, so that the barrier at the top of the next tile does not have to drain the stores
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) and lgkmcnt(0): DMA landed, exchange tile written
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (a.abl & 2) goto next_tile;
            // ---- row phase: 16 lanes per row (8 channels each), 4 rows per pass.  The lane id is made opaque so that
            // the per-pass LDS offsets are recomputed here instead of living in registers across the whole tile loop.
            int ol = lane;
            asm volatile("" : "+v"(ol));
            const int c16 = ol & 15, rq = ol >> 4;
            const f32x4* prm4 = reinterpret_cast<const f32x4*>(prm);
#pragma unroll
            for (int p = 0; p < MB; ++p) {
                const int rr = 4 * (w * MB + p) + rq;
                const bool ok = rr < N && !(a.abl & 1);
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(xch + xch_off(rr, 8 * c16, kC));
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(xch + xch_off(rr, 8 * c16 + 4, kC));
                const f32x4 o0 = prm4[2 * c16], o1 = prm4[2 * c16 + 1];
                const u32x4_t yr = *reinterpret_cast<const u32x4_t*>(yt + rr * 256 + ((c16 ^ (rr & 15)) << 4));
                f32x2 v[4];
                v[0] = lo2(x0) + lo2(o0) + unpack2_bf16(yr[0]);
                v[1] = hi2(x0) + hi2(o0) + unpack2_bf16(yr[1]);
                v[2] = lo2(x1) + lo2(o1) + unpack2_bf16(yr[2]);
                v[3] = hi2(x1) + hi2(o1) + unpack2_bf16(yr[3]);
                const size_t grow = node * N + rr;
                if (ok) {
                    u32x4_t pk4;
#pragma unroll
                    for (int h = 0; h < 4; ++h) pk4[h] = pack_bf16(v[h][0], v[h][1]);
                    *reinterpret_cast<u32x4_t*>(a.pre + grow * kC + 8 * c16) = pk4;
                }
                const f32x2 t2 = (v[0] + v[1]) + (v[2] + v[3]);
                const float mu = sum16(t2[0] + t2[1]) * (1.0f / 128.0f);
#pragma unroll
                for (int h = 0; h < 4; ++h) v[h] = v[h] - bc2(mu);
                const f32x2 q2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                const float var = sum16(q2[0] + q2[1]) * (1.0f / 128.0f);
                const float rs = __builtin_amdgcn_rsqf(var + a.eps);
                const f32x4 g0 = prm4[32 + 2 * c16], g1 = prm4[32 + 2 * c16 + 1];
                const f32x4 b0 = prm4[64 + 2 * c16], b1 = prm4[64 + 2 * c16 + 1];
                v[0] = v[0] * (bc2(rs) * lo2(g0)) + lo2(b0);
                v[1] = v[1] * (bc2(rs) * hi2(g0)) + hi2(b0);
                v[2] = v[2] * (bc2(rs) * lo2(g1)) + lo2(b1);
                v[3] = v[3] * (bc2(rs) * hi2(g1)) + hi2(b1);
                if (ok) {
                    u32x4_t pk4;
#pragma unroll
                    for (int h = 0; h < 4; ++h) pk4[h] = pack_bf16(v[h][0], v[h][1]);
                    *reinterpret_cast<u32x4_t*>(a.y2 + grow * kC + 8 * c16) = pk4;
                    if (c16 == 0) {
                        a.mean[grow] = mu;
                        a.rstd[grow] = rs;
                    }
                }
            }
        }
    next_tile:
        if (!more) break;
        if (!EDGE || (a.abl & 4)) wait_all_vmem_visible();
        b = nb_;
    }
}

// ------------------------------------------------------------------------------------- backward --
// Gradients of the block above given dz4 = d loss / d (y + s Woe^T + boe)  (the LayerNorm backward of ln4 runs first,
// dg_ln_residual_bwd: it also yields dgamma4 / dbeta4) and do = d loss / d o:
//     e, s, p        recomputed from y (the forward saves neither)
//     ds  = p (do_i v_j - sum_j p do_i v_j) + dz4 Woe          dv_j += p do_i      dk_j += ds g alpha q_i
//     dq_i = alpha sum_j ds g k_j         de = ds alpha q_i k_j (2 e + 1)          g = e^2 + e
//     dy  = dz4 + de We
//     dWe += de^T y    dbe += sum de      dWoe += dz4^T s      dboe += sum dz4
// HBM traffic: read y, dz4, write dy (3 edge passes; unfused: ds GEMM 2 + attention 3 + dy GEMM 3 + two weight
// gradients 4 = 12 after the LayerNorm backward).
//
// MI355X mapping: ONE workgroup of 8 waves per CU, wave w owns the 16 channels [16 w, 16 w + 16):
//   * y / dz4 tiles by LDS-DMA (double buffered) together with the q_i and do_i rows of the tile;
//   * e = y We^T and dz4 Woe run unswapped -> "attention layout" (lane = channel 16 w + (lane & 15), rows 16 mb + 4
//     (lane >> 4) + r); k_j, v_j and the accumulators dk_j, dv_j of those (j, c) pairs live in registers while the
//     workgroup walks the query rows of its work item (molecule, chunk of rows); partial dk / dv per item go to a
//     workspace and are summed in a fixed order afterwards (bit-reproducible, no atomics);
//   * weight gradients: the accumulator layout of a 16x16 block, converted to bf16, IS the A operand of
//     v_mfma_f32_16x16x16_bf16 for the transposed block, so de^T y and s^T dz4 take their A operands straight from the
//     registers of the attention stage; the B operands (4 consecutive rows of one channel) come from the ROW-MAJOR y / dz4
//     tiles through ds_read_b64_tr_b16 (scripts/ubench/tr_wgrad_probe.hip checks both idioms on the hardware).  The
//     whole [128,128] accumulators of both weight gradients stay in registers (64 per lane) for the kernel's lifetime;
//     dboe falls out of one extra MFMA with an all-ones A operand;
//   * de -> LDS (bf16, row-major) -> de We swapped, with dz4 added by one more MFMA against an identity fragment
//     -> bf16 staging tile -> whole 256-byte rows out.
// Chunk swizzle of the backward kernel's [rows][128] bf16 tiles: the 16-byte chunk c of row r sits at position
// c ^ bswz(r).  The tiles are read three ways -- ds_read_b128 MFMA fragments (16 rows x 16 B per hardware lane group
// {0-3,12-15,20-27}, ...), ds_read_b64_tr_b16 weight-gradient operands (per 32 lanes: 8 rows x 32 B) and 2-byte / 8-byte
// stores -- and `r & 15` (what the forward kernel uses) makes rows 2p and 2p + 1 share a 32-byte slot for the
// transposing reads (PMC: 24 % of the LDS cycles of this kernel were bank conflicts).  bswz keeps the fragment reads
// conflict-free (rows {0-3,12-15} map to chunks 0..7, rows {4..11} to 8..15) and gives rows 0..7 / 8..15 eight
// distinct 32-byte slots each.
__device__ __forceinline__ int bswz(int row) {
    const int r = row & 15;
    const int p = (r & 3) | ((((r >> 2) ^ (r >> 3)) & 1) << 2);
    return (p << 1) | (r >> 3);
}

struct HalfBwdArgs {
    const bf16_t* y;       // [B,N,N,C]
    const bf16_t* dz;      // [B,N,N,C]  (EDGE)
    const bf16_t* q;       // [B,N,C]
    const bf16_t* k;
    const bf16_t* v;
    const bf16_t* dO;      // [B,N,C]
    const bf16x8* pk;
    const float* be;
    bf16_t* dy;            // [B,N,N,C]
    bf16_t* dq;            // [B,N,C]
    float* part_kv;        // [B * CH][2][N][128]
    float* part_w;         // [2][grid][128 * 128]   dWe, dWoe
    float* part_b;         // [2][grid][128]         dbe, dboe
    int B, N, CH, RPC;
    float alpha;
};

__device__ __forceinline__ f32x4 mfma16k16(u32x2_t a, u32x2_t b, f32x4 c) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
template <int IMM>
__device__ __forceinline__ u32x2_t tr_read(unsigned addr) {
    u32x2_t r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(IMM) : "memory");
    return r;
}
// The transposing reads are issued from inline asm, so hipcc does not know they are in flight: the wait is explicit,
// with the destination registers threaded through it so that no use can be scheduled above it.  `YOUNGER` = LDS
// operations issued after these four that may stay outstanding (LDS operations complete in order).
template <int YOUNGER>
__device__ __forceinline__ void tr_wait(u32x2_t (&r)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(YOUNGER));
}
__device__ __forceinline__ bf16x8 cat8(u32x2_t lo, u32x2_t hi) {
    const u32x4_t t = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, t);
}
// acc[n] += A^T B over the MB 16-row blocks of one tile, n = 0..7 (16 output columns each): A operands `aop` straight from
// the attention-layout registers, B operands by ds_read_b64_tr_b16 from the row-major tile at `tb` (this lane's source
// address for k-block 0; k-block n flips address bits 5..7).  Row blocks are paired into v_mfma_f32_16x16x32_bf16 (k = 8
// kq + t: t < 4 from the even block, t >= 4 from the odd one -- the same convention for both operands), an odd last block
// uses v_mfma_f32_16x16x16_bf16.  Batches of four reads, the next batch in flight while the current one multiplies.
template <int MB, int BI>
__device__ __forceinline__ void wg_issue(unsigned tb, u32x2_t (&r)[4]) {
    constexpr int NP = MB / 2;
    if constexpr (BI < NP * 4) {
        constexpr int p = BI / 4, n0 = (BI % 4) * 2;
        const unsigned t0 = tb ^ (n0 << 5), t1 = tb ^ ((n0 + 1) << 5);
        r[0] = tr_read<(2 * p) * 4096>(t0);
        r[1] = tr_read<(2 * p + 1) * 4096>(t0);
        r[2] = tr_read<(2 * p) * 4096>(t1);
        r[3] = tr_read<(2 * p + 1) * 4096>(t1);
    } else {
        constexpr int n0 = (BI - NP * 4) * 4;
#pragma unroll
        for (int h = 0; h < 4; ++h) r[h] = tr_read<(MB - 1) * 4096>(tb ^ ((n0 + h) << 5));
    }
}
// (Never issue a 16x16x16 MFMA right behind a 16x16x32 one on the SAME accumulator: ROCm 7.2 hipcc emits no wait states
// between the two opcodes and the sum comes out wrong on gfx950 -- observed when the transposing reads were compiler
// builtins and hipcc scheduled the two back to back; the batches below keep them at least two MFMAs apart.)
template <int MB, int BI>
__device__ __forceinline__ void wg_mfma(const u32x2_t (&aop)[MB], const u32x2_t (&r)[4], f32x4 (&acc)[8]) {
    constexpr int NP = MB / 2;
    if constexpr (BI < NP * 4) {
        constexpr int p = BI / 4, n0 = (BI % 4) * 2;
        const bf16x8 a8 = cat8(aop[2 * p], aop[2 * p + 1]);
        acc[n0] = mfma16(a8, cat8(r[0], r[1]), acc[n0]);
        acc[n0 + 1] = mfma16(a8, cat8(r[2], r[3]), acc[n0 + 1]);
    } else {
        constexpr int n0 = (BI - NP * 4) * 4;
#pragma unroll
        for (int h = 0; h < 4; ++h) acc[n0 + h] = mfma16k16(aop[MB - 1], r[h], acc[n0 + h]);
    }
}
template <int MB, int BI>
__device__ __forceinline__ void wg_step(unsigned tb, const u32x2_t (&aop)[MB], f32x4 (&acc)[8], u32x2_t (&cur)[4], u32x2_t (&nxt)[4]) {
    constexpr int NB = (MB / 2) * 4 + ((MB & 1) ? 2 : 0);
    if constexpr (BI < NB) {
        if constexpr (BI + 1 < NB) {
            wg_issue<MB, BI + 1>(tb, nxt);
            tr_wait<4>(cur);
        } else {
            tr_wait<0>(cur);
        }
        wg_mfma<MB, BI>(aop, cur, acc);
        wg_step<MB, BI + 1>(tb, aop, acc, nxt, cur);
    }
}
template <int MB>
__device__ __forceinline__ void wg_stream(unsigned tb, const u32x2_t (&aop)[MB], f32x4 (&acc)[8]) {
    u32x2_t ra[4], rb[4];
    wg_issue<MB, 0>(tb, ra);
    wg_step<MB, 0>(tb, aop, acc, ra, rb);
}

// FM = first row block that can hold rows >= N for the sizes this instance serves (16 FM < N <= 16 MB)
template <int MB, int FM, bool EDGE, bool WGRAD>
__global__ __launch_bounds__(512, 1) void attn_half_bwd_bf16_kernel(const HalfBwdArgs a) {
    constexpr int ROWS = 16 * MB;
    constexpr bool WLDS = MB <= 3;            // Woe^T / We^T fragments in LDS instead of registers (32 fewer VGPRs)
    constexpr int YB = ROWS * 256;            // one bf16 tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // tile buffers: [buf][y | dz4][ROWS][128] bf16 (dz4 sits YB bytes behind its y tile: one address register serves both)
    char* dt = smem + 4 * YB;                 // [ROWS][128] bf16: de.  The dy staging tile reuses the y buffer of the tile
    char* qbuf = smem + 5 * YB;               // [2][2][128] bf16: q_i, do_i of the tile in flight / the next one
    bf16x8* wl = reinterpret_cast<bf16x8*>(qbuf + 1024);      // [3][8 waves][4][64] fragments: We^T, Woe^T, We (WLDS)
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int N = a.N;

    bf16x8 wfe_r[4], wfoet_r[4], wfet_r[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 t0 = a.pk[kSecE + (w * 4 + ks) * 64 + lane];
        if (WLDS) wl[4096 + (w * 4 + ks) * 64 + lane] = t0; else wfe_r[ks] = t0;
        const bf16x8 t1 = a.pk[kSecET + (w * 4 + ks) * 64 + lane];
        if (WLDS) wl[(w * 4 + ks) * 64 + lane] = t1; else wfet_r[ks] = t1;
        if (EDGE) {
            const bf16x8 t2 = a.pk[kSecOET + (w * 4 + ks) * 64 + lane];
            if (WLDS) wl[2048 + (w * 4 + ks) * 64 + lane] = t2; else wfoet_r[ks] = t2;
        }
    }
    const bf16x8* wlw = wl + w * 256 + lane;   // this lane's fragments: + ks * 64 (We^T), + 2048 + ks * 64 (Woe^T), + 4096 + ks * 64 (We)
    // identity fragment: dy += dz4 as one more k-step of the last product (k-step w >> 1 holds this wave's channels)
    bf16x8 idf;
#pragma unroll
    for (int t = 0; t < 8; ++t) idf[t] = static_cast<__bf16>((8 * kq + t == 16 * (w & 1) + r16) ? 1.0f : 0.0f);
    const float be1 = a.be[16 * w + r16];
    // zero the rows the DMA never writes (y and dz4 tiles, both buffers)
    for (int idx = threadIdx.x; idx < (ROWS - N) * 16 * 4; idx += 512) {
        const int bufi = idx / ((ROWS - N) * 16), rem = idx % ((ROWS - N) * 16);
        *reinterpret_cast<float4*>(smem + bufi * YB + N * 256 + rem * 16) = f4(0.f);
    }
    f32x4 negm[MB - FM];      // additive neighbour mask of the row blocks that can hold rows >= N
#pragma unroll
    for (int mb = FM; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) negm[mb - FM][r] = (16 * mb + 4 * kq + r < N) ? 0.f : kNegBig;
    unsigned af_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) af_off[ks] = r16 * 256 + (((4 * ks + kq) ^ bswz(r16)) << 4);
    // de tile store (attention layout, 2-byte elements): element (j = 16 mb + 4 kq + r, c = 16 w + r16) lives at
    // j * 256 + ((chunk ^ bswz(j)) << 4) + (c & 7) * 2, chunk = 2 w + (r16 >> 3): one base per r, row blocks are immediates
    unsigned dw_base[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        dw_base[r] = (4 * kq + r) * 256 + (((2 * w + (r16 >> 3)) ^ bswz(4 * kq + r)) << 4) + (r16 & 7) * 2;
    // ds_read_b64_tr_b16 source of this lane for k-block 0: row 4 kq + (r16 >> 2) of a 16-row block, channels 4 (r16 & 3)..;
    // k-block nbk flips bits 5..7 (XOR with nbk << 5), row blocks and buffers are immediates
    const int trr = 4 * kq + (r16 >> 2);
    const unsigned tr0 = lds_byte_address(smem) + trr * 256 + (((((r16 >> 1) & 1) ^ bswz(trr)) & 15) << 4) + (r16 & 1) * 8;
    // staging-tile store of the last product: row 16 mb + r16, channels 16 w + 4 kq ..
    const unsigned ow_off = r16 * 256 + (((2 * w + (kq >> 1)) ^ bswz(r16)) << 4) + (kq & 1) * 8;
    const u32x2_t ones = {0x3F803F80u, 0x3F803F80u};
    wait_all_vmem_visible();

    f32x4 accWe[8], accWoe[8], accB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 8; ++n) accWe[n] = accWoe[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbe = 0.f;

    // this workgroup's contiguous range of work items (molecule b, chunk of query rows)
    const int items = a.B * a.CH;
    int item = static_cast<int>(static_cast<long long>(items) * blockIdx.x / gridDim.x);
    const int item_end = static_cast<int>(static_cast<long long>(items) * (blockIdx.x + 1) / gridDim.x);
    const unsigned qdst = lds_byte_address(qbuf);
    auto dma_tile = [&](size_t tile, int bufi) {
        const unsigned ydst = lds_byte_address(smem + bufi * (2 * YB)), zdst = ydst + YB;
        // lane term opaque per call: as loop invariants the per-lane source offsets (64-bit) are hoisted out of the
        // tile loop and spilled, and a scratch reload waits (vmcnt(0)) for the LDS-DMA this very call has just issued
        int ol = lane;
        asm volatile("" : "+v"(ol));
#pragma unroll
        for (int ii0 = 0; ii0 < 4 * MB; ii0 += 8) {
            const int ii = ii0 + w;
            const int L = ii * 64 + ol;
            const int row = L >> 4, cpos = L & 15;
            if (ii < 4 * MB && row < N) {
                const size_t src = tile * N * kC + static_cast<unsigned>(row * kC + ((cpos ^ bswz(row)) << 3));
                dma16_async(reinterpret_cast<const float*>(a.y + src), ydst + ii * 1024);
                if (EDGE) dma16_async(reinterpret_cast<const float*>(a.dz + src), zdst + ii * 1024);
            }
        }
        if (w == 7 && ol < 32) {
            const bf16_t* src = (ol < 16 ? a.q : a.dO) + tile * kC + (ol & 15) * 8;
            dma16_async(reinterpret_cast<const float*>(src), qdst + bufi * 512);
        }
    };
    if (item < item_end) {
        const int b0 = item / a.CH, i00 = (item % a.CH) * a.RPC;
        __syncthreads();
        dma_tile(static_cast<size_t>(b0) * N + i00, 0);
        wait_all_vmem_visible();
    }
    int buf = 0;
    for (; item < item_end; ++item) {
        const int b = item / a.CH, i0 = (item % a.CH) * a.RPC;
        const int i1 = (i0 + a.RPC < N) ? i0 + a.RPC : N;
        f32x4 kk[MB], dkk[MB], dvv[MB];
        u32x2_t vp[MB];                   // v_j as packed bf16 pairs (two uses per tile), k_j in fp32 (three uses)
        int ol = lane;                    // opaque per item: the address arithmetic of the k / v loads and of the partial
        asm volatile("" : "+v"(ol));      // stores below stays inside the item loop instead of occupying registers
        const int o16 = ol & 15, oq = ol >> 4;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int j = 16 * mb + 4 * oq + r;
                if (j > N - 1) j = N - 1;
                const size_t off = (static_cast<size_t>(b) * N + j) * kC + 16 * w + o16;
                kk[mb][r] = static_cast<float>(a.k[off]);
                const unsigned vb = *reinterpret_cast<const unsigned short*>(a.v + off);
                if (r & 1) vp[mb][r >> 1] |= vb << 16;
                else vp[mb][r >> 1] = vb;
            }
            dkk[mb] = dvv[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int i = i0; i < i1; ++i, buf ^= 1) {
            const size_t node = static_cast<size_t>(b) * N + i;
            // next tile: next row of this item, or the first row of the next item
            bool more = true;
            size_t nnode = node + 1;
            if (i + 1 >= i1) {
                more = item + 1 < item_end;
                const int ni = item + 1;
                nnode = static_cast<size_t>(ni / a.CH) * N + (ni % a.CH) * a.RPC;
            }
            lds_barrier();        // tile landed everywhere (each wave waited for its share before its last stores); de /
                                  // staging tiles and the other buffers are free
            if (more) dma_tile(nnode, buf ^ 1);
            const char* yt = smem + buf * (2 * YB);
            const char* zt = yt + YB;
            const float qi = static_cast<float>(*reinterpret_cast<const bf16_t*>(qbuf + buf * 512 + (16 * w + r16) * 2));
            const float wo = static_cast<float>(*reinterpret_cast<const bf16_t*>(qbuf + buf * 512 + 256 + (16 * w + r16) * 2));
            const float aq = a.alpha * qi;
            // ---- e = y We^T + be ; dS = dz4 Woe   (attention layout)
            f32x4 ea[MB], sa[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                ea[mb] = f32x4{be1, be1, be1, be1};
                sa[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 wo_f;
                if (EDGE) wo_f = WLDS ? wlw[2048 + ks * 64] : wfoet_r[ks];
                const bf16x8 we_f = WLDS ? wlw[4096 + ks * 64] : wfe_r[ks];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(yt + mb * (16 * 256) + af_off[ks]);
                    ea[mb] = mfma16(af, we_f, ea[mb]);
                    if (EDGE) {
                        const bf16x8 zf = *reinterpret_cast<const bf16x8*>(zt + mb * (16 * 256) + af_off[ks]);
                        sa[mb] = mfma16(zf, wo_f, sa[mb]);
                    }
                }
            }
            // ---- recompute s, p
            f32x4 pe[MB];
            u32x2_t sA[MB];
            float mx = kNegBig;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const f32x2 e0 = lo2(ea[mb]), e1 = hi2(ea[mb]);
                f32x2 s0 = (bc2(aq) * lo2(kk[mb])) * (e0 * e0 + e0), s1 = (bc2(aq) * hi2(kk[mb])) * (e1 * e1 + e1);
                if (mb >= FM) {
                    s0 += lo2(negm[mb >= FM ? mb - FM : 0]);
                    s1 += hi2(negm[mb >= FM ? mb - FM : 0]);
                }
                pe[mb] = f32x4{s0[0], s0[1], s1[0], s1[1]};
                if (EDGE && WGRAD) {
                    sA[mb][0] = pack_bf16(s0[0], s0[1]);
                    sA[mb][1] = pack_bf16(s1[0], s1[1]);
                }
                mx = vmax(vmax(mx, s0[0]), vmax(s0[1], vmax(s1[0], s1[1])));
            }
            // ---- dWoe^T += s^T dz4, dboe += 1^T dz4 (s is consumed here: its bf16 copy does not live through the softmax)
            if (EDGE && WGRAD) {
                const unsigned tbz = tr0 + buf * (2 * YB) + YB;
                wg_stream<MB>(tbz, sA, accWoe);
                // column sums of dz4 for this wave's 16 channels: all-ones A operand against k-block w
                u32x2_t bz[4];
                const unsigned tw = tbz ^ (w << 5);
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) bz[mb] = mb < MB ? tr_read<0>(tw + (mb < MB ? mb : 0) * 4096) : u32x2_t{0u, 0u};
                tr_wait<0>(bz);
#pragma unroll
                for (int mb = 0; mb < (MB < 4 ? MB : 4); ++mb) accB = mfma16k16(ones, bz[mb], accB);
                if (MB > 4) {
                    u32x2_t bz2[4];
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) bz2[mb] = 4 + mb < MB ? tr_read<0>(tw + (4 + mb < MB ? 4 + mb : 0) * 4096) : u32x2_t{0u, 0u};
                    tr_wait<0>(bz2);
#pragma unroll
                    for (int mb = 4; mb < MB; ++mb) accB = mfma16k16(ones, bz2[mb - 4], accB);
                }
            }
            mx = -kLog2e * xor_vmax16(mx);
            f32x2 l2 = {0.f, 0.f}, A2 = {0.f, 0.f};
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const f32x2 x0 = lo2(pe[mb]) * bc2(kLog2e) + bc2(mx);
                const f32x2 x1 = hi2(pe[mb]) * bc2(kLog2e) + bc2(mx);
                const f32x2 p0 = {__builtin_amdgcn_exp2f(x0[0]), __builtin_amdgcn_exp2f(x0[1])};
                const f32x2 p1 = {__builtin_amdgcn_exp2f(x1[0]), __builtin_amdgcn_exp2f(x1[1])};
                pe[mb] = f32x4{p0[0], p0[1], p1[0], p1[1]};
                l2 += p0 + p1;
                A2 += p0 * unpack2_bf16(vp[mb][0]) + p1 * unpack2_bf16(vp[mb][1]);
            }
            const float inv = __builtin_amdgcn_rcpf(xor_sum<16>(l2[0] + l2[1]));
            const float abar = wo * xor_sum<16>(A2[0] + A2[1]) * inv;      // sum_j p do v_j
            // ---- gradients
            f32x2 dq2 = {0.f, 0.f}, dbe2 = {0.f, 0.f};
            u32x2_t deA[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const f32x2 e[2] = {lo2(ea[mb]), hi2(ea[mb])};
                const f32x2 kx[2] = {lo2(kk[mb]), hi2(kk[mb])};
                const f32x2 vx[2] = {unpack2_bf16(vp[mb][0]), unpack2_bf16(vp[mb][1])};
                const f32x2 px[2] = {lo2(pe[mb]) * bc2(inv), hi2(pe[mb]) * bc2(inv)};
                const f32x2 sx[2] = {lo2(sa[mb]), hi2(sa[mb])};
                f32x2 de[2], dkx[2], dvx[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x2 ds = px[h] * (bc2(wo) * vx[h] - bc2(abar)) + sx[h];
                    dvx[h] = px[h] * bc2(wo);
                    const f32x2 dsg = ds * (e[h] * e[h] + e[h]);
                    dq2 += dsg * kx[h];
                    dkx[h] = dsg * bc2(aq);
                    de[h] = (ds * (bc2(aq) * kx[h])) * (e[h] * bc2(2.f) + bc2(1.f));
                    dbe2 += de[h];
                }
                dkk[mb] += f32x4{dkx[0][0], dkx[0][1], dkx[1][0], dkx[1][1]};
                dvv[mb] += f32x4{dvx[0][0], dvx[0][1], dvx[1][0], dvx[1][1]};
                deA[mb][0] = pack_bf16(de[0][0], de[0][1]);
                deA[mb][1] = pack_bf16(de[1][0], de[1][1]);
                // de -> LDS (row-major bf16): rows 16 mb + 4 kq + r, channel 16 w + r16
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<unsigned short*>(dt + dw_base[r] + mb * (16 * 256)) =
                        static_cast<unsigned short>(deA[mb][r >> 1] >> (16 * (r & 1)));
            }
            dbe += dbe2[0] + dbe2[1];
            {
                const float dqv = a.alpha * xor_sum<16>(dq2[0] + dq2[1]);
                if (kq == 0) a.dq[node * kC + 16 * w + r16] = static_cast<__bf16>(dqv);
            }
            // ---- dWe += de^T y
            if (WGRAD) wg_stream<MB>(tr0 + buf * (2 * YB), deA, accWe);
            lds_barrier();
            // ---- dy = de We (+ dz4), swapped: lane = row 16 mb + r16, channels 16 w + 4 kq + {0..3}
            char* ot = smem + buf * (2 * YB);      // staging tile = this tile's y buffer
            f32x4 oa[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) oa[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const bf16x8 df = *reinterpret_cast<const bf16x8*>(dt + mb * (16 * 256) + af_off[ks]);
                    oa[mb] = mfma16(WLDS ? wlw[ks * 64] : wfet_r[ks], df, oa[mb]);
                }
            if (EDGE) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const bf16x8 zf = *reinterpret_cast<const bf16x8*>(zt + mb * (16 * 256) + r16 * 256 + (((4 * (w >> 1) + kq) ^ bswz(r16)) << 4));
                    oa[mb] = mfma16(idf, zf, oa[mb]);
                }
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                u32x2_t pk2;
                pk2[0] = pack_bf16(oa[mb][0], oa[mb][1]);
                pk2[1] = pack_bf16(oa[mb][2], oa[mb][3]);
                *reinterpret_cast<u32x2_t*>(ot + mb * (16 * 256) + ow_off) = pk2;      // (every wave is past its reads of y)
            }
            wait_all_vmem_visible();      // next tile's DMA (issued a tile ago): before the stores below, not after them
            lds_barrier();
            // ---- whole rows out: 16 lanes per row, 4 rows per instruction
#pragma unroll
            for (int ii0 = 0; ii0 < 4 * MB; ii0 += 8) {
                const int ii = ii0 + w;
                const int row = 4 * ii + kq;
                if (ii < 4 * MB && row < N) {
                    const u32x4_t vrow = *reinterpret_cast<const u32x4_t*>(ot + row * 256 + ((r16 ^ bswz(row)) << 4));
                    *reinterpret_cast<u32x4_t*>(a.dy + (node * N + row) * kC + 8 * r16) = vrow;
                }
            }
        }
        // ---- partial dk, dv of this work item
        asm volatile("" : "+v"(ol));
        float* pkv = a.part_kv + static_cast<size_t>(item) * 2 * N * kC + 16 * w + (ol & 15);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * mb + 4 * (ol >> 4) + r;
                if (j < N) {
                    pkv[j * kC] = dkk[mb][r];
                    pkv[(N + j) * kC] = dvv[mb][r];
                }
            }
    }
    if (!WGRAD) return;
    // ---- weight-gradient partials of this workgroup: part_w = [2][grid][128 * 128], part_b = [2][grid][128]
    float* pw = a.part_w + static_cast<size_t>(blockIdx.x) * kC * kC;
    const size_t wstride = static_cast<size_t>(gridDim.x) * kC * kC;
#pragma unroll
    for (int nbk = 0; nbk < 8; ++nbk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pw[(16 * w + 4 * kq + r) * kC + 16 * nbk + r16] = accWe[nbk][r];                      // dWe[o][k]
            if (EDGE) pw[wstride + (16 * nbk + r16) * kC + 16 * w + 4 * kq + r] = accWoe[nbk][r];  // dWoe[o][c]
        }
    float* pb = a.part_b + static_cast<size_t>(blockIdx.x) * kC;
    const float dbe_all = xor_sum<16>(dbe);
    if (kq == 0) {
        pb[16 * w + r16] = dbe_all;
        if (EDGE) pb[static_cast<size_t>(gridDim.x) * kC + 16 * w + r16] = accB[0];
    }
}

// dk, dv [B,N,C] = sum over the CH chunk partials (fixed order), converted to the activation type
__global__ __launch_bounds__(256) void half_kv_finish_kernel(const float* __restrict__ part, int CH, int N, bf16_t* __restrict__ dk,
                                                           bf16_t* __restrict__ dv, int64_t total) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;      // over B * N * C
    if (idx >= total) return;
    const int64_t per = static_cast<int64_t>(N) * kC;
    const int64_t b = idx / per, rem = idx % per;
    float sk = 0.f, sv = 0.f;
    for (int c = 0; c < CH; ++c) {
        const float* pp = part + (b * CH + c) * 2 * per;
        sk += pp[rem];
        sv += pp[per + rem];
    }
    dk[idx] = static_cast<__bf16>(sk);
    dv[idx] = static_cast<__bf16>(sv);
}

// out[i] = sum_s part[s * n + i] in a fixed order (bias-sized vectors)
__global__ __launch_bounds__(128) void half_bias_reduce_kernel(const float* __restrict__ part, int S, int n, float* __restrict__ out) {
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll 8
    for (int p = 0; p < S; ++p) s += part[static_cast<size_t>(p) * n + i];
    out[i] = s;
}

// work decomposition of the backward: CH chunks of RPC query rows per molecule, one workgroup per CU
struct HalfBwdPlan {
    int CH, RPC, grid;
};
HalfBwdPlan half_bwd_plan(int B, int N) {
    int ch = 1;
    while (ch < N && (static_cast<long long>(B) * ch) % 256 != 0 && static_cast<long long>(B) * ch < 2048) ++ch;
    HalfBwdPlan p;
    p.RPC = (N + ch - 1) / ch;
    p.CH = (N + p.RPC - 1) / p.RPC;
    const long long items = static_cast<long long>(B) * p.CH;
    p.grid = static_cast<int>(items < 256 ? items : 256);
    return p;
}

int half_mb(int N) { return N <= 16 ? 1 : (N <= 48 ? 3 : (N <= 96 ? 6 : 0)); }

}  // namespace
}  // namespace dg

using namespace dg;

extern "C" size_t dg_attn_half_packed_bytes(int dtype) {
    return dtype == DG_DTYPE_BF16 ? static_cast<size_t>(4) * kSec * 16 : 0;
}

extern "C" int dg_attn_half_pack(const float* we, const float* woe, void* packed, int dtype, dg_stream_t stream_) {
    if (!we || !woe || !packed) return fail(DG_E_ARG, "dg_attn_half_pack: null pointer");
    if (dtype != DG_DTYPE_BF16) return fail(DG_E_ARG, "dg_attn_half_pack: dtype %d not supported", dtype);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    bf16x8* p = static_cast<bf16x8*>(packed);
    int st = pack_bf16(we, p + kSecE, kC, kC, 0, 16, stream);
    if (!st) st = pack_bf16(woe, p + kSecOE, kC, kC, 0, 16, stream);
    if (!st) st = pack_bf16(woe, p + kSecOET, kC, kC, 1, 16, stream);
    if (!st) st = pack_bf16(we, p + kSecET, kC, kC, 1, 16, stream);
    return st;
}

extern "C" int dg_attn_half_fwd(const void* y, const void* q, const void* k, const void* v, const void* packed,
                                const float* be, const float* boe, const float* gamma4, const float* beta4, void* o,
                                void* y2, void* pre4, float* mean4, float* rstd4, int B, int N, int C, float alpha,
                                float eps, int dtype, dg_stream_t stream_) {
    if (!y || !q || !k || !v || !packed || !be || !o) return fail(DG_E_ARG, "dg_attn_half_fwd: null pointer");
    const bool edge = y2 != nullptr;
    if (edge && (!boe || !gamma4 || !beta4 || !pre4 || !mean4 || !rstd4))
        return fail(DG_E_ARG, "dg_attn_half_fwd: the edge output needs boe, gamma4, beta4, pre4, mean4 and rstd4");
    if (dtype != DG_DTYPE_BF16) return fail(DG_E_ARG, "dg_attn_half_fwd: dtype %d not supported", dtype);
    const int mb = half_mb(N);
    if (B < 0 || C != kC || N < 1 || !mb)
        return fail(DG_E_SHAPE, "dg_attn_half_fwd: unsupported shape B=%d N=%d C=%d (need C == 128, N <= 96)", B, N, C);
    if (B == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HalfFwdArgs a;
    a.y = static_cast<const bf16_t*>(y); a.q = static_cast<const bf16_t*>(q); a.k = static_cast<const bf16_t*>(k);
    a.v = static_cast<const bf16_t*>(v); a.pk = static_cast<const bf16x8*>(packed); a.be = be; a.boe = boe;
    a.gamma = gamma4; a.beta = beta4; a.o = static_cast<bf16_t*>(o); a.y2 = static_cast<bf16_t*>(y2);
    a.pre = static_cast<bf16_t*>(pre4); a.mean = mean4; a.rstd = rstd4; a.B = B; a.N = N; a.alpha = alpha; a.eps = eps;
    a.abl = 0;
    const long long items = static_cast<long long>(B) * N;      // tiles
    ProfScope prof(DG_K_ATTN_HALF_FWD, stream);
#define LAUNCH(MB_, EDGE_, PER_CU_)                                                                               \
    {                                                                                                             \
        constexpr int lds = 16 * MB_ * (3 * 256 + 512) + 3 * 128 * 4 + 2 * 256;                                                           \
        DG_OPT_IN_LDS((&attn_half_fwd_bf16_kernel<MB_, EDGE_>), lds);                                              \
        const int grid = static_cast<int>(items < 256 * PER_CU_ ? items : 256 * PER_CU_);                         \
        hipLaunchKernelGGL((attn_half_fwd_bf16_kernel<MB_, EDGE_>), dim3(grid), dim3(256), lds, stream, a);        \
    }
    if (mb == 1) { if (edge) LAUNCH(1, true, 2) else LAUNCH(1, false, 2) }
    else if (mb == 3) { if (edge) LAUNCH(3, true, 2) else LAUNCH(3, false, 2) }
    else { if (edge) LAUNCH(6, true, 1) else LAUNCH(6, false, 1) }
#undef LAUNCH
    return check_launch("dg_attn_half_fwd");
}

extern "C" size_t dg_attn_half_bwd_workspace_bytes(int B, int N) {
    if (B < 1 || N < 1) return 0;
    const HalfBwdPlan p = half_bwd_plan(B, N);
    return (static_cast<size_t>(B) * p.CH * 2 * N * kC + static_cast<size_t>(p.grid) * 2 * (kC * kC + kC)) * sizeof(float);
}

extern "C" int dg_attn_half_bwd(const void* y, const void* dz4, const void* q, const void* k, const void* v,
                                const void* d_o, const void* packed, const float* be, void* dy, void* dq, void* dk,
                                void* dv, float* dwe, float* dbe, float* dwoe, float* dboe, void* workspace,
                                size_t workspace_bytes, int B, int N, int C, float alpha, int dtype, dg_stream_t stream_) {
    if (!y || !q || !k || !v || !d_o || !packed || !be || !dy || !dq || !dk || !dv || !workspace)
        return fail(DG_E_ARG, "dg_attn_half_bwd: null pointer");
    const bool edge = dz4 != nullptr, wgrad = dwe != nullptr;
    if (wgrad && (!dbe || (edge && (!dwoe || !dboe))))
        return fail(DG_E_ARG, "dg_attn_half_bwd: weight gradients need dwe, dbe (and dwoe, dboe with the edge output) together");
    if (dtype != DG_DTYPE_BF16) return fail(DG_E_ARG, "dg_attn_half_bwd: dtype %d not supported", dtype);
    const int mb = half_mb(N);
    if (B < 1 || C != kC || N < 1 || !mb)
        return fail(DG_E_SHAPE, "dg_attn_half_bwd: unsupported shape B=%d N=%d C=%d (need C == 128, N <= 96)", B, N, C);
    if (workspace_bytes < dg_attn_half_bwd_workspace_bytes(B, N)) return fail(DG_E_WORKSPACE, "dg_attn_half_bwd: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const HalfBwdPlan p = half_bwd_plan(B, N);
    HalfBwdArgs a;
    a.y = static_cast<const bf16_t*>(y); a.dz = static_cast<const bf16_t*>(dz4); a.q = static_cast<const bf16_t*>(q);
    a.k = static_cast<const bf16_t*>(k); a.v = static_cast<const bf16_t*>(v); a.dO = static_cast<const bf16_t*>(d_o);
    a.pk = static_cast<const bf16x8*>(packed); a.be = be; a.dy = static_cast<bf16_t*>(dy); a.dq = static_cast<bf16_t*>(dq);
    a.part_kv = static_cast<float*>(workspace);
    a.part_w = a.part_kv + static_cast<size_t>(B) * p.CH * 2 * N * kC;
    a.part_b = a.part_w + static_cast<size_t>(p.grid) * 2 * kC * kC;
    a.B = B; a.N = N; a.CH = p.CH; a.RPC = p.RPC; a.alpha = alpha;
    {
        ProfScope prof(DG_K_ATTN_HALF_BWD, stream);
#define LAUNCH(MB_, FM_, EDGE_, WG_)                                                                                       \
    {                                                                                                                 \
        constexpr int lds = 16 * MB_ * 256 * 5 + 1024 + (MB_ <= 3 ? 98304 : 0);                                        \
        DG_OPT_IN_LDS((&attn_half_bwd_bf16_kernel<MB_, FM_, EDGE_, WG_>), lds);                                             \
        hipLaunchKernelGGL((attn_half_bwd_bf16_kernel<MB_, FM_, EDGE_, WG_>), dim3(p.grid), dim3(512), lds, stream, a);     \
    }
#define LAUNCH_MB(MB_, FM_)                                         \
    {                                                               \
        if (edge && wgrad) LAUNCH(MB_, FM_, true, true)             \
        else if (edge) LAUNCH(MB_, FM_, true, false)                \
        else if (wgrad) LAUNCH(MB_, FM_, false, true)               \
        else LAUNCH(MB_, FM_, false, false)                         \
    }
        if (N <= 16) LAUNCH_MB(1, 0) else if (N <= 32) LAUNCH_MB(2, 1) else if (N <= 48) LAUNCH_MB(3, 2) else LAUNCH_MB(6, 3)
#undef LAUNCH_MB
#undef LAUNCH
    }
    const int64_t total = static_cast<int64_t>(B) * N * kC;
    hipLaunchKernelGGL(half_kv_finish_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, stream,
                       a.part_kv, p.CH, N, static_cast<bf16_t*>(dk), static_cast<bf16_t*>(dv), total);
    if (wgrad) {
        launch_splitk_reduce(a.part_w, p.grid, kC * kC / 4, dwe, stream);
        if (edge) launch_splitk_reduce(a.part_w + static_cast<size_t>(p.grid) * kC * kC, p.grid, kC * kC / 4, dwoe, stream);
        hipLaunchKernelGGL(half_bias_reduce_kernel, dim3(1), dim3(128), 0, stream, a.part_b, p.grid, kC, dbe);
        if (edge)
            hipLaunchKernelGGL(half_bias_reduce_kernel, dim3(1), dim3(128), 0, stream, a.part_b + static_cast<size_t>(p.grid) * kC,
                               p.grid, kC, dboe);
    }
    return check_launch("dg_attn_half_bwd");
}
