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
    int buf = 0;
    for (;; buf ^= 1, ++t) {
        const size_t node = static_cast<size_t>(t);
        const bool more = t + 1 < t_end;
        const int nb_ = static_cast<int>((t + 1) / N);      // molecule of the next tile
        wait_all_vmem_visible();  // (through the builtin: hipcc then knows that k / v reloads have landed as well)
        __syncthreads();          // tile i landed for every wave; s / exchange tiles and the other y buffer are free
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
            __syncthreads();
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
            __syncthreads();
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
        b = nb_;
    }
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
    a.abl = getenv("DG_HALF_ABL") ? atoi(getenv("DG_HALF_ABL")) : 0;
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
