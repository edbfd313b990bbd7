// Edge embedding + symmetrisation of Generator / Discriminator (reference
// src/model/models.py:57-61,92-94 and :159-163,197-199):
//
//     f(z)   = act( W2 . act( W1 . z + b1 ) + b2 )          Linear(E,64) - act - Linear(64,C) - act
//     out_ij = ( f(a_ij) + f(a_ji) ) / 2                    (edge + edge.permute(0,2,1,3)) / 2
//
// One fused kernel per direction.  A workgroup owns a tile of 32 atom PAIRS {i <= j} of one
// molecule = 64 edge rows: rows 0..31 hold the (i,j) orientation, rows 32..63 the transposed (j,i)
// one.  After the MFMA stage the two orientations of a pair sit in the SAME accumulator slot of the
// two 32-row blocks, so the symmetrisation (and, in the backward, the symmetrised upstream
// gradient) is register arithmetic; both output rows are written from there.  Layer 1 (E <= 16
// inputs) is VALU work, layer 2 (64 -> C=128) runs on v_mfma_f32_32x32x2_f32 with the packed weight
// fragments resident in VGPRs; the hidden tile lives in LDS (XOR-swizzled, conflict-free b128
// fragment reads).  Nothing but the inputs is saved: the backward recomputes both layers.
#include "bf16.h"
#include "traversal.h"

namespace dg {
void launch_ln_finish(const float* part, int nblocks, int K, int C, float* out0, float* out1, hipStream_t stream);      // layernorm.hip
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kHid = 64;     // hidden width of the embedding MLP (fixed by the reference)
constexpr int kC = 128;      // output width handled by this kernel
constexpr int kPairs = 32;   // atom pairs per tile
constexpr int kMaxE = 16;
constexpr int kD1Pitch = 68;   // floats per row of the dpre1 tile (16-byte aligned rows, conflict-free b128 reads)

enum Act { kRelu = 0, kLeaky = 1, kSigmoid = 2, kTanh = 3 };

// ACT is a template parameter of the kernels: a run-time switch inside the per-element loops costs a branch ladder
// per value and keeps hipcc from scheduling across the elements
template <int act>
__device__ __forceinline__ float act_fwd(float x) {
    switch (act) {
        case kRelu: return fmaxf(x, 0.f);
        case kLeaky: return x > 0.f ? x : 0.01f * x;
        case kSigmoid: return 1.0f / (1.0f + __expf(-x));
        default: return tanhf(x);
    }
}
// derivative expressed through the OUTPUT y = act(x)
template <int act>
__device__ __forceinline__ float act_grad_from_output(float y) {
    switch (act) {
        case kRelu: return y > 0.f ? 1.f : 0.f;
        case kLeaky: return y > 0.f ? 1.f : 0.01f;
        case kSigmoid: return y * (1.f - y);
        default: return 1.f - y * y;
    }
}

struct PairTile {
    int b;        // molecule
    int p0;       // first pair of the tile
};

// pair index p (0 <= p < N(N+1)/2, row-major over i <= j) -> (i, j)
__device__ __forceinline__ void pair_to_ij(int p, int N, int* i_out, int* j_out) {
    int i = 0;
    while (p >= N - i) {
        p -= N - i;
        ++i;
    }
    *i_out = i;
    *j_out = i + p;
}

template <bool BACKWARD>
struct Smem {
    int ij[kPairs][2];
    float a[64][kMaxE];
    float h1[64 * kHid];                          // swizzled [row][64]
};

// Stage the pair table, the input rows and the layer-1 activations of one tile.
// EP = padded number of input features (8 or 16): sizes the per-lane weight / accumulator arrays.
template <int EP, int ACT>
__device__ __forceinline__ void stage_tile(const float* __restrict__ a, const float* __restrict__ w1,
                                           const float* __restrict__ b1, int N, int E, int NP, PairTile t,
                                           int (*ij)[2], float (*at)[kMaxE], float* h1) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // opaque per call: keeps the per-row LDS offsets out of the caller's loop preheader
    if (tid < kPairs) {
        const int p = t.p0 + tid;
        int i = 0, j = 0;
        if (p < NP) pair_to_ij(p, N, &i, &j);
        ij[tid][0] = p < NP ? i : -1;
        ij[tid][1] = j;
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * EP; idx += 256) {
        const int row = idx / EP, e = idx % EP;
        const int pr = row & 31;
        const int i = ij[pr][0], j = ij[pr][1];
        float v = 0.f;
        if (i >= 0 && e < E) {
            const int64_t r = (static_cast<int64_t>(t.b) * N + (row < 32 ? i : j)) * N + (row < 32 ? j : i);
            v = a[r * E + e];
        }
        at[row][e] = v;
    }
    __syncthreads();
    // layer 1: thread = (unit u, row group g of 16 rows)
    const int u = tid & 63, g = tid >> 6;
    float w[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) w[e] = e < E ? w1[u * E + e] : 0.f;
    const float bb = b1[u];
    for (int r = 0; r < 16; ++r) {
        const int row = g * 16 + r;
        float s = bb;
#pragma unroll
        for (int e = 0; e < EP; ++e) s = fmaf(w[e], at[row][e], s);
        h1[row * kHid + (((u >> 2) ^ (row & 15)) << 2) + (u & 3)] = act_fwd<ACT>(s);
    }
    __syncthreads();
}

// ---- bf16 x 3 arithmetic of the MFMA stages (fp32 class) ----------------------------------------------------------
// Every fp32 operand value is split exactly into three bf16 by truncation (h = top 16 bits, m = top 16 bits of x - h,
// l = top 16 bits of x - h - m: together all 24 significand bits) and the six cross products with i + j <= 4 run on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- the arithmetic of the weight-gradient kernel (linear_wgrad.hip).
// 48 MFMAs of 32 cycles per stage instead of 64 fp32 MFMAs of 64 cycles.  LDS tiles stay fp32; fragments are split in
// registers.  Weight operands arrive pre-split ([k-step][plane][lane] x 8 bf16, embed_pack3_kernel).
typedef unsigned u32x4e __attribute__((ext_vector_type(4)));
struct Planes { bf16x8 p[3]; };

__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned a0 = __float_as_uint(x0), a1 = __float_as_uint(x1);
    h = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(a0 & 0xFFFF0000u), r1 = x1 - __uint_as_float(a1 & 0xFFFF0000u);
    const unsigned b0 = __float_as_uint(r0), b1 = __float_as_uint(r1);
    m = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(b0 & 0xFFFF0000u), s1 = r1 - __uint_as_float(b1 & 0xFFFF0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}
__device__ __forceinline__ Planes split8(const float (&v)[8]) {
    u32x4e h, m, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned hh, mm, ll;
        split2(v[2 * i], v[2 * i + 1], hh, mm, ll);
        h[i] = hh;
        m[i] = mm;
        l[i] = ll;
    }
    Planes r;
    r.p[0] = __builtin_bit_cast(bf16x8, h);
    r.p[1] = __builtin_bit_cast(bf16x8, m);
    r.p[2] = __builtin_bit_cast(bf16x8, l);
    return r;
}
__device__ __forceinline__ Planes split8(const float4& a, const float4& b) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return split8(v);
}
__device__ __forceinline__ f32x16 mfma6(const Planes& a, const Planes& b, f32x16 c) {
    constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};   // smallest terms first
#pragma unroll
    for (int t = 0; t < 6; ++t) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[TA[t]], b.p[TB[t]], c, 0, 0, 0);
    return c;
}
// The gradient stages (dW2 += dpre2^T h1, dh = dpre2 W2) are backward-only: their roundings travel through linear maps and stay
// roundings, so they take the three leading cross products (m h', h m', h h': 2^-16 of a product left out) -- half
// the MFMAs of a stage.  The layer-2 RECOMPUTATION keeps all six: its signs are the ReLU mask of the forward, which must not flip
// (all three stages on three products: 1.4e-3 .. 4e-3 on the goldens, the square-root law of a perturbed forward).
__device__ __forceinline__ f32x16 mfma_bwd(const Planes& a, const Planes& b, f32x16 c) {
    constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};   // smallest terms first
#pragma unroll
    for (int t = 0; t < 3; ++t) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[TA[t]], b.p[TB[t]], c, 0, 0, 0);
    return c;
}
// eight consecutive k (chunks c, c + 1 of four floats) of row `row` of an XOR-swizzled fp32 tile with `pitch` floats
__device__ __forceinline__ Planes frag_row(const float* tile, int row, int c, int pitch) {
    const float4 a = ld4(tile + row * pitch + ((c ^ (row & 15)) << 2));
    const float4 b = ld4(tile + row * pitch + (((c + 1) ^ (row & 15)) << 2));
    return split8(a, b);
}
__device__ __forceinline__ Planes load_planes(const bf16x8* p) {   // [plane][lane] of one (tile, k-step)
    Planes r;
#pragma unroll
    for (int q = 0; q < 3; ++q) r.p[q] = p[q * 64];
    return r;
}

// layer 2: this wave's 32-column slab for both orientations, K = 64 = four k-steps.  `w2f` points at the slab's
// pre-split fragments for this lane.
__device__ __forceinline__ void layer2_mfma(const float* h1, const bf16x8* w2f, int lane, f32x16& acc0, f32x16& acc1) {
    const int half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const Planes b = load_planes(w2f + ks * 3 * 64);
        const Planes a0 = frag_row(h1, col, 4 * ks + 2 * half, kHid);
        const Planes a1 = frag_row(h1, 32 + col, 4 * ks + 2 * half, kHid);
        acc0 = mfma6(a0, b, acc0);
        acc1 = mfma6(a1, b, acc1);
    }
}

// Pre-split layer-2 weight operands: P[((t * KS + ks) * 3 + plane) * 64 + lane] = 8 bf16 { B[16 ks + 8 (lane >> 5) + j][32 t + (lane & 31)] }_j.
//   dgrad = 0: B[k][n] = W2[n][k]  (forward: k = hidden unit, n = channel; 4 slabs x 4 k-steps)
//   dgrad = 1: B[k][n] = W2[k][n]  (dh = dz2 W2: k = channel, n = hidden unit; 2 tiles x 8 k-steps)
__global__ void embed_pack3_kernel(const float* __restrict__ w2, bf16x8* __restrict__ p, int dgrad) {
    const int KS = dgrad ? 8 : 4, NT = dgrad ? 2 : 4;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // one (t, ks, lane)
    if (idx >= NT * KS * 64) return;
    const int lane = idx & 63, ks = (idx >> 6) % KS, t = (idx >> 6) / KS;
    const int n = 32 * t + (lane & 31);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * ks + 8 * (lane >> 5) + j;
        v[j] = dgrad ? w2[static_cast<size_t>(k) * kHid + n] : w2[static_cast<size_t>(n) * kHid + k];
    }
    const Planes pl = split8(v);
#pragma unroll
    for (int q = 0; q < 3; ++q) p[(static_cast<size_t>(t * KS + ks) * 3 + q) * 64 + lane] = pl.p[q];
}

// A [32 pairs][128] fp32 LDS tile holds one value per (pair, channel); both orientations (b,i,j) and (b,j,i) of a
// pair receive it as whole rows: one half-wave per row, 16 bytes (fp32) / 8 bytes (bf16) per lane.
template <typename T>
__device__ __forceinline__ void store_pair_rows(const float* xt, const int (*ij)[2], int b, int N, T* __restrict__ out,
                                                int tid) {
    const int hw = tid >> 5, l32 = tid & 31;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int pr = hw + 8 * it;
        const int i = ij[pr][0], j = ij[pr][1];
        if (i < 0) continue;
        const float4 v = ld4(xt + pr * kC + 4 * l32);
        const int64_t base = static_cast<int64_t>(b) * N;
        st4(out + ((base + i) * N + j) * kC + 4 * l32, v);
        if (i != j) st4(out + ((base + j) * N + i) * kC + 4 * l32, v);
    }
}

template <typename T, int EP, int ACT>
__global__ __launch_bounds__(256) void embed_sym_fwd_kernel(const float* __restrict__ a, const float* __restrict__ w1,
                                                          const float* __restrict__ b1,
                                                          const float* __restrict__ w2p,
                                                          const float* __restrict__ b2, T* __restrict__ out,
                                                          int B, int N, int E, int tiles_per_mol) {
    __shared__ int ij[kPairs][2];
    __shared__ float at[64][kMaxE];
    __shared__ __attribute__((aligned(16))) float h1[64 * kHid];
    __shared__ __attribute__((aligned(16))) float xt[kPairs * kC];    // symmetrised outputs of the tile
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, col = lane & 31;
    const int NP = N * (N + 1) / 2;
    const bf16x8* w2f = reinterpret_cast<const bf16x8*>(w2p) + static_cast<size_t>(w) * 4 * 3 * 64 + lane;
    const int n = 32 * w + col;
    const float bias2 = b2[n];
    const int total = B * tiles_per_mol;
    for (int tix = blockIdx.x; tix < total; tix += gridDim.x) {
        const PairTile t{tix / tiles_per_mol, (tix % tiles_per_mol) * kPairs};
        stage_tile<EP, ACT>(a, w1, b1, N, E, NP, t, ij, at, h1);
        f32x16 acc0, acc1;
        layer2_mfma(h1, w2f, lane, acc0, acc1);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int pr = (reg & 3) + 8 * (reg >> 2) + 4 * half;
            xt[pr * kC + n] = 0.5f * (act_fwd<ACT>(acc0[reg] + bias2) + act_fwd<ACT>(acc1[reg] + bias2));
        }
        __syncthreads();
        store_pair_rows(xt, ij, t.b, N, out, static_cast<int>(threadIdx.x));
        __syncthreads();   // LDS tiles are reused by the next iteration
    }
}

// W-gradient stage of the backward kernels: aw2[t] += P^T Q over the 64 tile rows (four k-steps of 16 rows), P = the
// wave's 32 output channels of the [64][128] tile `d2`, Q = unit tiles t = 0, 1 of the [64][64] tile `hq`.  Operands are
// gathered column-wise (eight ds_read_b32 per fragment: a lane's eight rows of one column) and split in registers.
__device__ __forceinline__ void aw2_stage(const float* d2, const float* hq, int n, int col, int half, f32x16 (&aw2)[2]) {
    const int c = n >> 2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        float av[8], b0[8], b1v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 16 * ks + 8 * half + j;
            av[j] = d2[r * kC + (((c & ~15) | ((c & 15) ^ (r & 15))) << 2) + (n & 3)];
            b0[j] = hq[r * kHid + ((((col) >> 2) ^ (r & 15)) << 2) + (col & 3)];
            b1v[j] = hq[r * kHid + ((((32 + col) >> 2) ^ (r & 15)) << 2) + (col & 3)];
        }
        const Planes pa = split8(av);
        aw2[0] = mfma_bwd(pa, split8(b0), aw2[0]);
        aw2[1] = mfma_bwd(pa, split8(b1v), aw2[1]);
    }
}
// dh1 = P W2 for one (32-row block, 32-unit tile): K = 128 channels = eight k-steps, two accumulator chains
__device__ __forceinline__ f32x16 dh_stage(const float* d2, const bf16x8* w2g, int row, int half) {
    f32x16 d0, d1v;
#pragma unroll
    for (int i = 0; i < 16; ++i) d0[i] = d1v[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ks += 2) {
        d0 = mfma_bwd(frag_row(d2, row, 4 * ks + 2 * half, kC), load_planes(w2g + ks * 3 * 64), d0);
        d1v = mfma_bwd(frag_row(d2, row, 4 * (ks + 1) + 2 * half, kC), load_planes(w2g + (ks + 1) * 3 * 64), d1v);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) d0[i] += d1v[i];
    return d0;
}

// ------------------------------------------------------------------------------- backward ----
// g = dL/d out [B,N,N,C].  gs = (g_ij + g_ji)/2 reaches BOTH orientations of a pair.
//   dpre2 = gs * act'(f)                 (f recomputed)          db2 += sum_rows dpre2
//   dW2  += dpre2^T h1                   (MFMA, contraction over the tile rows)
//   dh1   = dpre2 W2                     (MFMA, contraction over C)
//   dpre1 = dh1 * act'(h1)               db1 += sum_rows dpre1,  dW1 += dpre1^T a
//   da    = dpre1 W1                     (optional: only when the input requires a gradient)
// Per-workgroup partial sums go to `part`, reduced afterwards in a fixed order.
struct BwdPart {
    // floats per workgroup: dW2 [128*64] | db2 [128] | dW1 [64*16] | db1 [64]
    static constexpr int kW2 = 0, kB2 = kC * kHid, kW1 = kB2 + kC, kB1 = kW1 + kHid * kMaxE, kTotal = kB1 + kHid;
};

// DA: the input gradient is wanted (a template parameter: with a run-time `if (da)` hipcc allocated the whole tile loop for
// the input-gradient stage -- 255 VGPRs and 84 B of scratch (620 B for EP = 16) against 165-173 VGPRs and none without it)
template <typename T, int EP, int ACT, bool DA>
__global__ __launch_bounds__(256, 2) void embed_sym_bwd_kernel(
    const float* __restrict__ a, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2p, const float* __restrict__ w2d, const float* __restrict__ b2,
    const T* __restrict__ g, float* __restrict__ da, float* __restrict__ part, int B, int N, int E,
    int tiles_per_mol) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* h1 = reinterpret_cast<float*>(smem_raw);                 // [64][64] swizzled
    float* d2 = h1 + 64 * kHid;                                      // dpre2, swizzled [64][128]
    float* d1 = d2 + 64 * kC;                                        // dpre1 [64][kD1Pitch]
    float(*at)[kMaxE] = reinterpret_cast<float(*)[kMaxE]>(d1 + 64 * kD1Pitch);
    int(*ij)[2] = reinterpret_cast<int(*)[2]>(&at[64][0]);
    float4* w1p = reinterpret_cast<float4*>(&ij[kPairs][0]);   // W1 permuted: w1p[u * 4 + eg] = { W1[u][eg + 4 q] }_q
    float* gst = d1;   // symmetrised upstream gradient of the tile [32 pairs][128]: dead before d1 is written
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (DA) {   // visible after the first barrier of the tile loop
        const int u = tid >> 2, eg = tid & 3;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = eg + 4 * q < E ? w1[u * E + eg + 4 * q] : 0.f;
        w1p[tid] = make_float4(v[0], v[1], v[2], v[3]);
    }

    const int half = lane >> 5, col = lane & 31;
    const int NP = N * (N + 1) / 2;
    // weight fragments are re-read per tile (L2 hits) instead of pinning 96 VGPRs for the whole kernel
    const int ut = w & 1, mt = w >> 1;   // dgrad: output tile ut (32 hidden units), row block mt
    const int n = 32 * w + col;
    const float bias2 = b2[n];
    f32x16 aw2[2];                       // dW2 tiles (n tile w) x (unit tile 0,1)
#pragma unroll
    for (int i = 0; i < 16; ++i) aw2[0][i] = aw2[1][i] = 0.f;
    float ab2 = 0.f, ab1 = 0.f, aw1[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) aw1[e] = 0.f;
    const int total = B * tiles_per_mol;
#ifdef DG_EMBED_DBG
    unsigned long long tl = __builtin_amdgcn_s_memtime(), ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t00 = tl;
#define ESTAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ts[i] += t_ - tl; tl = t_; }
#else
#define ESTAMP(i)
#endif
    for (int tix = blockIdx.x; tix < total; tix += gridDim.x) {
        const PairTile t{tix / tiles_per_mol, (tix % tiles_per_mol) * kPairs};
        stage_tile<EP, ACT>(a, w1, b1, N, E, NP, t, ij, at, h1);
        ESTAMP(0)
        // Per-iteration opaque copy of the lane id: every swizzled LDS offset below is derived from it, so hipcc cannot
        // hoist the ~200 loop-invariant offsets out of the tile loop (it did, and spilled 150 of them to scratch).
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const int half = lo >> 5, col = lo & 31, n = 32 * w + col;
        // upstream gradient rows (b,i,j) and (b,j,i) of the tile's 32 pairs: whole rows (one half-wave per row, 8 loads
        // in flight per lane) requested before the layer-2 MFMAs, symmetrised into LDS after them -- instead of 64
        // dependent scalar gathers per lane from the accumulator layout
        typedef typename raw4<T>::type Raw;
        Raw gr[4][2];
        {
            const int hw = lo >> 5 | (w << 1), l32 = lo & 31;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pr = hw + 8 * it;
                const int i = ij[pr][0], j = ij[pr][1];
                const int64_t base = static_cast<int64_t>(t.b) * N;
                const int ii = i >= 0 ? i : 0, jj = i >= 0 ? j : 0;       // empty pairs read a valid row, result unused
                gr[it][0] = ld_raw(g + ((base + ii) * N + jj) * kC + 4 * l32);
                gr[it][1] = ld_raw(g + ((base + jj) * N + ii) * kC + 4 * l32);
            }
        }
        f32x16 acc0, acc1;
        layer2_mfma(h1, reinterpret_cast<const bf16x8*>(w2p) + static_cast<size_t>(w) * 4 * 3 * 64 + lo, lo, acc0, acc1);
        {
            const int hw = lo >> 5 | (w << 1), l32 = lo & 31;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pr = hw + 8 * it;
                const bool diag = ij[pr][0] == ij[pr][1];
                const float4 v = (diag ? 0.25f : 0.5f) * (cvt_raw(gr[it][0]) + cvt_raw(gr[it][1]));   // diagonal: both blocks carry half
                st4(gst + pr * kC + 4 * l32, ij[pr][0] >= 0 ? v : f4(0.f));
            }
        }
        __syncthreads();
        ESTAMP(1)
        // dpre2 in the accumulator layout -> LDS (swizzled like a row-GEMM A tile)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int pr = (reg & 3) + 8 * (reg >> 2) + 4 * half;
            const int i = ij[pr][0], j = ij[pr][1];
            float p0 = 0.f, p1 = 0.f;
            if (i >= 0) {
                const float gs = gst[pr * kC + n];
                p0 = gs * act_grad_from_output<ACT>(act_fwd<ACT>(acc0[reg] + bias2));
                p1 = gs * act_grad_from_output<ACT>(act_fwd<ACT>(acc1[reg] + bias2));
            }
            (void)j;
            ab2 += p0 + p1;
            const int c = n >> 2;
            d2[pr * kC + (((c & ~15) | ((c & 15) ^ (pr & 15))) << 2) + (n & 3)] = p0;
            d2[(32 + pr) * kC + (((c & ~15) | ((c & 15) ^ (pr & 15))) << 2) + (n & 3)] = p1;
            // four rows at a time: without the fence hipcc hoists all 64 gradient loads (and their addresses) above
            // the loop, 428 registers per lane = one wave per SIMD
        }
        __syncthreads();
        ESTAMP(2)
        // dW2 += dpre2^T h1 : contraction over the 64 tile rows, operands by ds_read_b32
        aw2_stage(d2, h1, n, col, half, aw2);
        ESTAMP(3)
        // dh1 = dpre2 W2 for (row block mt, unit tile ut); dpre1 = dh1 * act'(h1)
        const f32x16 dh = dh_stage(d2, reinterpret_cast<const bf16x8*>(w2d) + static_cast<size_t>(ut) * 8 * 3 * 64 + lo, 32 * mt + col, half);
        ESTAMP(4)
        const int u = 32 * ut + col;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = 32 * mt + (reg & 3) + 8 * (reg >> 2) + 4 * half;
            const float hv = h1[row * kHid + (((u >> 2) ^ (row & 15)) << 2) + (u & 3)];
            const float p = dh[reg] * act_grad_from_output<ACT>(hv);
            ab1 += p;
            float av[EP];
#pragma unroll
            for (int e4 = 0; e4 < EP; e4 += 4) {       // the row's inputs as 16-byte LDS reads (broadcast within a half-wave)
                const float4 t4 = ld4(&at[row][e4]);
                av[e4] = t4.x; av[e4 + 1] = t4.y; av[e4 + 2] = t4.z; av[e4 + 3] = t4.w;
            }
#pragma unroll
            for (int e = 0; e < EP; ++e) aw1[e] = fmaf(p, av[e], aw1[e]);
            if (DA) d1[row * kD1Pitch + u] = p;
        }
        ESTAMP(5)
        if (DA) {
            __syncthreads();
            // da[row][e] = sum_u dpre1[row][u] W1[u][e]: thread = (tile row, e mod 4); the two halves of a diagonal pair
            // (rows pr and 32 + pr = lanes pr and 32 + pr of the wave) are summed across the wave
            const int row = lo, eg = w;
            float s[4] = {0.f, 0.f, 0.f, 0.f};
            // dpre1 as 16-byte reads (row pitch 68 floats: the eight lanes of a quarter-wave cover all 32 banks), the four
            // W1 entries of (unit, eg) as one broadcast 16-byte read
#pragma unroll 4
            for (int u4 = 0; u4 < kHid; u4 += 4) {
                const float4 dv = ld4(d1 + row * kD1Pitch + u4);
                const float dvv[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
                for (int uu = 0; uu < 4; ++uu) {
                    const float4 wv = w1p[(u4 + uu) * 4 + eg];
                    s[0] = fmaf(dvv[uu], wv.x, s[0]);
                    s[1] = fmaf(dvv[uu], wv.y, s[1]);
                    s[2] = fmaf(dvv[uu], wv.z, s[2]);
                    s[3] = fmaf(dvv[uu], wv.w, s[3]);
                }
            }
            const int pr = row & 31;
            const int i = ij[pr][0], j = ij[pr][1];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float other = __shfl_xor(s[q], 32, 64);
                const int e = eg + 4 * q;
                if (i < 0 || e >= E || (row >= 32 && i == j)) continue;
                const int64_t r = (static_cast<int64_t>(t.b) * N + (row < 32 ? i : j)) * N + (row < 32 ? j : i);
                da[r * E + e] = i == j ? s[q] + other : s[q];
            }
        }
        __syncthreads();
        ESTAMP(6)
    }
#ifdef DG_EMBED_DBG
    if (lane == 0 && blockIdx.x == 17) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(part + static_cast<size_t>(gridDim.x + 1) * BwdPart::kTotal) + 8 * w;
        for (int i = 0; i < 7; ++i) o[i] = ts[i];
        o[7] = __builtin_amdgcn_s_memtime() - t00;
    }
#endif
    // ---- workgroup partials ----------------------------------------------------------------
    float* pw = part + static_cast<size_t>(blockIdx.x) * BwdPart::kTotal;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int nn = 32 * w + (reg & 3) + 8 * (reg >> 2) + 4 * half;
            pw[BwdPart::kW2 + nn * kHid + 32 * t2 + col] = aw2[t2][reg];
        }
    // db2: lanes of the two halves hold the same column
    ab2 += __shfl_xor(ab2, 32, 64);
    if (half == 0) pw[BwdPart::kB2 + n] = ab2;
    // dW1 / db1: unit u = 32*ut + col is held by 2 half-waves x 2 row blocks (waves ut and ut+2)
    __syncthreads();
    float* red = d2;   // [4 waves][2 halves][32 cols][kMaxE + 1]
    {
        float* slot = red + ((w * 2 + half) * 32 + col) * (kMaxE + 1);
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) slot[e] = e < EP ? aw1[e < EP ? e : 0] : 0.f;
        slot[kMaxE] = ab1;
    }
    __syncthreads();
    for (int idx = tid; idx < kHid * (kMaxE + 1); idx += 256) {
        const int uu = idx / (kMaxE + 1), e = idx % (kMaxE + 1);
        const int utile = uu >> 5, c = uu & 31;
        float s = 0.f;
        for (int mm = 0; mm < 2; ++mm)
            for (int hh = 0; hh < 2; ++hh) s += red[(((utile + 2 * mm) * 2 + hh) * 32 + c) * (kMaxE + 1) + e];
        if (e < kMaxE)
            pw[BwdPart::kW1 + uu * kMaxE + e] = s;
        else
            pw[BwdPart::kB1 + uu] = s;
    }
}

// ---------------------------------------------------------------- second order ----
// Backward of the backward above for the gradient penalty (reference src/model/loss.py:32-39 differentiates
// d out / d a): t = dL/d(da) [B,N,N,E] is the adjoint of the first backward's input gradient.  Piecewise-linear
// activations only (ReLU, LeakyReLU: act'' = 0, so nothing reaches a, b1, b2):
//   da   = W1^T p1,   p1 = (W2^T p2) * act'(h1),   p2 = gs * act'(f)            (first backward, recomputed)
//   q    = (W1 t) * act'(h1)                                                     [rows, 64]
//   x    = (W2 q) * act'(f)            gg_ij = gg_ji = (x_ij + x_ji) / 2         adjoint of the upstream gradient g
//   gW2 += p2^T q                      gW1 += p1^T t
// Same tile program as the first backward with (q, t) in the places of (h1, a) in the two weight-gradient
// stages, plus one more layer-2 MFMA pass for x.
template <typename T, int EP, int ACT>
__global__ __launch_bounds__(256, 2) void embed_sym_bwd2_kernel(
    const float* __restrict__ a, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2p, const float* __restrict__ w2d, const float* __restrict__ b2,
    const T* __restrict__ g, const float* __restrict__ tadj, T* __restrict__ gg, float* __restrict__ part, int B, int N,
    int E, int tiles_per_mol) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* h1 = reinterpret_cast<float*>(smem_raw);                 // [64][64] swizzled
    float* d2 = h1 + 64 * kHid;                                      // p2, swizzled [64][128]
    float* hb = d2 + 64 * kC;                                        // q, swizzled like h1
    float(*at)[kMaxE] = reinterpret_cast<float(*)[kMaxE]>(hb + 64 * kHid);
    float(*tt)[kMaxE] = reinterpret_cast<float(*)[kMaxE]>(&at[64][0]);
    int(*ij)[2] = reinterpret_cast<int(*)[2]>(&tt[64][0]);
    float* gst = reinterpret_cast<float*>(&ij[kPairs][0]);           // symmetrised upstream gradient [32][128], then x
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NP = N * (N + 1) / 2;
    const int ut = w & 1, mt = w >> 1;
    f32x16 aw2[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) aw2[0][i] = aw2[1][i] = 0.f;
    float aw1[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) aw1[e] = 0.f;
    const int total = B * tiles_per_mol;
    for (int tix = blockIdx.x; tix < total; tix += gridDim.x) {
        const PairTile t{tix / tiles_per_mol, (tix % tiles_per_mol) * kPairs};
        stage_tile<EP, ACT>(a, w1, b1, N, E, NP, t, ij, at, h1);
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        // adjoint rows, in the same (i,j) / (j,i) arrangement as the inputs
        for (int idx = tid; idx < 64 * EP; idx += 256) {
            const int row = idx / EP, e = idx % EP;
            const int pr = row & 31;
            const int i = ij[pr][0], j = ij[pr][1];
            float v = 0.f;
            if (i >= 0 && e < E) {
                const int64_t r = (static_cast<int64_t>(t.b) * N + (row < 32 ? i : j)) * N + (row < 32 ? j : i);
                v = tadj[r * E + e];
            }
            tt[row][e] = v;
        }
        __syncthreads();
        {   // q = (W1 t) * act'(h1): thread = (unit u, 16 rows), like layer 1
            const int u = tid & 63, gq = tid >> 6;
            float wv[EP];
#pragma unroll
            for (int e = 0; e < EP; ++e) wv[e] = e < E ? w1[u * E + e] : 0.f;
            for (int r = 0; r < 16; ++r) {
                const int row = gq * 16 + r;
                float sacc = 0.f;
#pragma unroll
                for (int e = 0; e < EP; ++e) sacc = fmaf(wv[e], tt[row][e], sacc);
                const int o = row * kHid + (((u >> 2) ^ (row & 15)) << 2) + (u & 3);
                hb[o] = sacc * act_grad_from_output<ACT>(h1[o]);
            }
        }
        __syncthreads();
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const int half = lo >> 5, col = lo & 31, n = 32 * w + col;
        const float bias2 = b2[n];
        typedef typename raw4<T>::type Raw;
        Raw gr[4][2];
        {   // upstream gradient rows of the 32 pairs, whole rows (see the first backward)
            const int hw = lo >> 5 | (w << 1), l32 = lo & 31;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pr = hw + 8 * it;
                const int i = ij[pr][0], j = ij[pr][1];
                const int64_t base = static_cast<int64_t>(t.b) * N;
                const int ii = i >= 0 ? i : 0, jj = i >= 0 ? j : 0;
                gr[it][0] = ld_raw(g + ((base + ii) * N + jj) * kC + 4 * l32);
                gr[it][1] = ld_raw(g + ((base + jj) * N + ii) * kC + 4 * l32);
            }
        }
        f32x16 acc0, acc1, q0, q1;
        {
            const bf16x8* w2f = reinterpret_cast<const bf16x8*>(w2p) + static_cast<size_t>(w) * 4 * 3 * 64 + lo;
            layer2_mfma(h1, w2f, lo, acc0, acc1);
            layer2_mfma(hb, w2f, lo, q0, q1);
        }
        {
            const int hw = lo >> 5 | (w << 1), l32 = lo & 31;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pr = hw + 8 * it;
                const bool diag = ij[pr][0] == ij[pr][1];
                const float4 v = (diag ? 0.25f : 0.5f) * (cvt_raw(gr[it][0]) + cvt_raw(gr[it][1]));
                st4(gst + pr * kC + 4 * l32, ij[pr][0] >= 0 ? v : f4(0.f));
            }
        }
        __syncthreads();
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int pr = (reg & 3) + 8 * (reg >> 2) + 4 * half;
            const int i = ij[pr][0], j = ij[pr][1];
            float p0 = 0.f, p1 = 0.f;
            if (i >= 0) {
                const float gs = gst[pr * kC + n];   // this (pair, channel) slot belongs to this lane alone: read, then reuse for x
                const float d0 = act_grad_from_output<ACT>(act_fwd<ACT>(acc0[reg] + bias2));
                const float d1v = act_grad_from_output<ACT>(act_fwd<ACT>(acc1[reg] + bias2));
                p0 = gs * d0;
                p1 = gs * d1v;
                gst[pr * kC + n] = 0.5f * (q0[reg] * d0 + q1[reg] * d1v);
            }
            (void)j;
            const int c = n >> 2;
            d2[pr * kC + (((c & ~15) | ((c & 15) ^ (pr & 15))) << 2) + (n & 3)] = p0;
            d2[(32 + pr) * kC + (((c & ~15) | ((c & 15) ^ (pr & 15))) << 2) + (n & 3)] = p1;
        }
        __syncthreads();
        store_pair_rows(gst, ij, t.b, N, gg, tid);
        // gW2 += p2^T q : contraction over the 64 tile rows
        aw2_stage(d2, hb, n, col, half, aw2);
        // dh1 = p2 W2 for (row block mt, unit tile ut); p1 = dh1 * act'(h1); gW1 += p1^T t
        const f32x16 dh = dh_stage(d2, reinterpret_cast<const bf16x8*>(w2d) + static_cast<size_t>(ut) * 8 * 3 * 64 + lo, 32 * mt + col, half);
        const int u = 32 * ut + col;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = 32 * mt + (reg & 3) + 8 * (reg >> 2) + 4 * half;
            const float hv = h1[row * kHid + (((u >> 2) ^ (row & 15)) << 2) + (u & 3)];
            const float p = dh[reg] * act_grad_from_output<ACT>(hv);
            float av[EP];
#pragma unroll
            for (int e4 = 0; e4 < EP; e4 += 4) {
                const float4 t4 = ld4(&tt[row][e4]);
                av[e4] = t4.x; av[e4 + 1] = t4.y; av[e4 + 2] = t4.z; av[e4 + 3] = t4.w;
            }
#pragma unroll
            for (int e = 0; e < EP; ++e) aw1[e] = fmaf(p, av[e], aw1[e]);
        }
        __syncthreads();
    }
    // ---- workgroup partials (same layout as the first backward; the bias slots stay zero) ------------
    const int half = lane >> 5, col = lane & 31;
    float* pw = part + static_cast<size_t>(blockIdx.x) * BwdPart::kTotal;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int nn = 32 * w + (reg & 3) + 8 * (reg >> 2) + 4 * half;
            pw[BwdPart::kW2 + nn * kHid + 32 * t2 + col] = aw2[t2][reg];
        }
    __syncthreads();
    float* red = d2;   // [4 waves][2 halves][32 cols][kMaxE]
    {
        float* slot = red + ((w * 2 + half) * 32 + col) * kMaxE;
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) slot[e] = e < EP ? aw1[e < EP ? e : 0] : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < kHid * kMaxE; idx += 256) {
        const int uu = idx / kMaxE, e = idx % kMaxE;
        const int utile = uu >> 5, c = uu & 31;
        float sum = 0.f;
        for (int mm = 0; mm < 2; ++mm)
            for (int hh = 0; hh < 2; ++hh) sum += red[(((utile + 2 * mm) * 2 + hh) * 32 + c) * kMaxE + e];
        pw[BwdPart::kW1 + uu * kMaxE + e] = sum;
    }
}

// Column sums of the partial vectors [dW2 | db2 | dW1(16-padded) | db1] of the backward kernels, written straight into the
// caller's tensors: one block per 32 elements, 32 row groups stride over the S partials with eight loads in flight, then a
// fixed-order LDS sum (bit-reproducible).  (One thread per element walking all S partials was 33 us of dependent loads; a
// separate scatter launch followed.)
__global__ __launch_bounds__(1024) void embed_finish_kernel(const float* __restrict__ part, int S, float* __restrict__ dw1,
                                                          float* __restrict__ db1, float* __restrict__ dw2,
                                                          float* __restrict__ db2, int E) {
    __shared__ float red[32][33];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    const int g = threadIdx.x >> 5;
    float s = 0.f;
    if (c < BwdPart::kTotal) {
        const float* src = part + c;
        int b = g;
        for (; b + 7 * 32 < S; b += 8 * 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[static_cast<size_t>(b + 32 * u) * BwdPart::kTotal];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; b < S; b += 32) s += src[static_cast<size_t>(b) * BwdPart::kTotal];
    }
    red[g][threadIdx.x & 31] = s;
    __syncthreads();
    if (g == 0 && c < BwdPart::kTotal) {
        float t = 0.f;
        for (int i = 0; i < 32; ++i) t += red[i][threadIdx.x];
        if (c >= BwdPart::kW2 && c < BwdPart::kW2 + kC * kHid) dw2[c - BwdPart::kW2] = t;
        else if (c >= BwdPart::kB2 && c < BwdPart::kB2 + kC) { if (db2) db2[c - BwdPart::kB2] = t; }
        else if (c >= BwdPart::kW1 && c < BwdPart::kW1 + kHid * kMaxE) {
            const int i = c - BwdPart::kW1;
            if (i % kMaxE < E) dw1[(i / kMaxE) * E + i % kMaxE] = t;
        } else if (c >= BwdPart::kB1 && c < BwdPart::kB1 + kHid) { if (db1) db1[c - BwdPart::kB1] = t; }
    }
}

// ---------------------------------------------------------------- one-hot inputs ----
// For a one-hot input graph (the generator's input and the discriminator's real batch: reference
// src/data/utils.py:15-23 builds them with label2onehot) the embedding MLP has only E distinct results,
// T[c] = f(e_c): out[b,i,j,:] = (T[l_ij] + T[l_ji]) / 2 is a table gather (HBM-bound: writes 256-512 B per row,
// reads 8 B of labels), and its backward a segmented sum dT[c] = sum_rows g_ij ([l_ij = c] + [l_ji = c]) / 2.
// The table itself (E x 128) is computed -- and differentiated -- by ordinary torch ops on the host side.
template <typename T>
__global__ __launch_bounds__(256) void onehot_embed_fwd_kernel(const int* __restrict__ labels, const float* __restrict__ table,
                                                             T* __restrict__ out, int64_t rows, int N, int E) {
    __shared__ float4 tab[kMaxE * kC / 4];
    for (int i = threadIdx.x; i < E * kC / 4; i += 256) tab[i] = ld4(table + i * 4);
    __syncthreads();
    const int col = threadIdx.x & 31;
    const int64_t NN = static_cast<int64_t>(N) * N;
    for (int64_t r = static_cast<int64_t>(blockIdx.x) * 8 + (threadIdx.x >> 5); r < rows; r += static_cast<int64_t>(gridDim.x) * 8) {
        const int64_t b = r / NN;
        const int ij = static_cast<int>(r - b * NN), i = ij / N, j = ij - i * N;
        int l1 = labels[r], l2 = labels[b * NN + static_cast<int64_t>(j) * N + i];
        l1 = l1 < 0 ? 0 : (l1 >= E ? E - 1 : l1);      // the caller validated the labels; stay in bounds regardless
        l2 = l2 < 0 ? 0 : (l2 >= E ? E - 1 : l2);
        st4_stream(out + r * kC + 4 * col, 0.5f * (tab[l1 * (kC / 4) + col] + tab[l2 * (kC / 4) + col]));
    }
}

template <typename T, int EP>
__global__ __launch_bounds__(256) void onehot_embed_bwd_kernel(const int* __restrict__ labels, const T* __restrict__ g,
                                                             float* __restrict__ part, int64_t rows, int N, int E) {
    __shared__ float4 red[8][EP][32];
    const int col = threadIdx.x & 31, rp = threadIdx.x >> 5;
    const int64_t NN = static_cast<int64_t>(N) * N;
    float4 acc[EP];
#pragma unroll
    for (int c = 0; c < EP; ++c) acc[c] = f4(0.f);
    for (int64_t r = static_cast<int64_t>(blockIdx.x) * 8 + rp; r < rows; r += static_cast<int64_t>(gridDim.x) * 8) {
        const int64_t b = r / NN;
        const int ij = static_cast<int>(r - b * NN), i = ij / N, j = ij - i * N;
        const int l1 = labels[r], l2 = labels[b * NN + static_cast<int64_t>(j) * N + i];
        const float4 gv = ld4_stream(g + r * kC + 4 * col);
#pragma unroll
        for (int c = 0; c < EP; ++c) {
            const float wgt = 0.5f * ((l1 == c ? 1.f : 0.f) + (l2 == c ? 1.f : 0.f));
            acc[c] = fma4(f4(wgt), gv, acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < EP; ++c) red[rp][c][col] = acc[c];
    __syncthreads();
    for (int idx = threadIdx.x; idx < E * 32; idx += 256) {
        const int c = idx >> 5, cc = idx & 31;
        float4 t = f4(0.f);
        for (int p = 0; p < 8; ++p) t += red[p][c][cc];      // fixed order: bit-reproducible
        st4(part + (static_cast<size_t>(blockIdx.x) * E + c) * kC + 4 * cc, t);
    }
}

constexpr int kOneHotBlocks = 1024;

constexpr int kBwdPerCu = 2;   // backward workgroups per CU (71 KB of LDS each): their serial phases overlap
int embed_grid(int total_tiles, int per_cu) {
    const int cap = 256 * per_cu;
    return total_tiles < cap ? (total_tiles < 1 ? 1 : total_tiles) : cap;
}

bool embed_shape_ok(int N, int E, int H, int C, int act) {
    return N >= 1 && E >= 1 && E <= kMaxE && H == kHid && C == kC && act >= 0 && act <= 3;
}

}  // namespace
}  // namespace dg

using namespace dg;

extern "C" size_t dg_embed_sym_packed_floats(void) { return static_cast<size_t>(4) * 4 * 3 * 64 * 4; }   // 16 B per entry

extern "C" size_t dg_embed_sym_workspace_bytes(int B, int N) {
    const int tiles = B * ((N * (N + 1) / 2 + kPairs - 1) / kPairs);
    return (static_cast<size_t>(embed_grid(tiles, kBwdPerCu)) + 1) * BwdPart::kTotal * sizeof(float) + 256;   // + debug stamps
}

extern "C" int dg_embed_sym_pack(const float* w2, float* packed, dg_stream_t stream_) {
    if (!w2 || !packed) return fail(DG_E_ARG, "dg_embed_sym_pack: null pointer");
    hipLaunchKernelGGL(embed_pack3_kernel, dim3(4), dim3(256), 0, static_cast<hipStream_t>(stream_), w2,
                       reinterpret_cast<bf16x8*>(packed), 0);
    return check_launch("dg_embed_sym_pack");
}

extern "C" size_t dg_embed_sym_dgrad_packed_floats(void) { return static_cast<size_t>(2) * 8 * 3 * 64 * 4; }

extern "C" int dg_embed_sym_pack_dgrad(const float* w2, float* packed, dg_stream_t stream_) {
    if (!w2 || !packed) return fail(DG_E_ARG, "dg_embed_sym_pack_dgrad: null pointer");
    hipLaunchKernelGGL(embed_pack3_kernel, dim3(4), dim3(256), 0, static_cast<hipStream_t>(stream_), w2,
                       reinterpret_cast<bf16x8*>(packed), 1);
    return check_launch("dg_embed_sym_pack_dgrad");
}

extern "C" int dg_embed_sym_fwd(const float* a, const float* w1, const float* b1, const float* w2_packed,
                                const float* b2, void* out, int B, int N, int E, int H, int C, int act, int dtype,
                                dg_stream_t stream_) {
    if (!a || !w1 || !b1 || !w2_packed || !b2 || !out) return fail(DG_E_ARG, "dg_embed_sym_fwd: null pointer");
    if (!dtype_ok(dtype)) return fail(DG_E_ARG, "dg_embed_sym_fwd: unknown dtype %d", dtype);
    if (B < 0 || !embed_shape_ok(N, E, H, C, act))
        return fail(DG_E_SHAPE, "dg_embed_sym_fwd: unsupported N=%d E=%d H=%d C=%d act=%d (need E<=16, H=64, C=128)", N, E,
                    H, C, act);
    if (B == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int tpm = (N * (N + 1) / 2 + kPairs - 1) / kPairs;
    ProfScope prof(DG_K_EMBED_SYM, stream);
    note_forward(static_cast<int64_t>(B) * N * N);
#define FWD_A(T, EP_, ACT_)                                                                                      \
    hipLaunchKernelGGL((embed_sym_fwd_kernel<T, EP_, ACT_>), dim3(embed_grid(B * tpm, 8)), dim3(256), 0, stream, a, w1, \
                       b1, w2_packed, b2, static_cast<T*>(out), B, N, E, tpm);
#define FWD(T, EP_)                                                                                             \
    switch (act) {                                                                                              \
        case kRelu: FWD_A(T, EP_, kRelu) break;                                                                 \
        case kLeaky: FWD_A(T, EP_, kLeaky) break;                                                               \
        case kSigmoid: FWD_A(T, EP_, kSigmoid) break;                                                           \
        default: FWD_A(T, EP_, kTanh) break;                                                                    \
    }
    if (dtype == DG_DTYPE_BF16) {
        if (E <= 8) { FWD(bf16_t, 8) } else { FWD(bf16_t, 16) }
    } else {
        if (E <= 8) { FWD(float, 8) } else { FWD(float, 16) }
    }
#undef FWD_A
#undef FWD
    return check_launch("dg_embed_sym_fwd");
}

extern "C" int dg_embed_sym_bwd(const float* a, const float* w1, const float* b1, const float* w2_packed,
                                const float* w2_dgrad_packed, const float* b2, const void* g, float* da, float* dw1,
                                float* db1, float* dw2, float* db2, void* workspace, size_t workspace_bytes, int B,
                                int N, int E, int H, int C, int act, int dtype, dg_stream_t stream_) {
    if (!a || !w1 || !b1 || !w2_packed || !w2_dgrad_packed || !b2 || !g || !dw1 || !db1 || !dw2 || !db2 || !workspace)
        return fail(DG_E_ARG, "dg_embed_sym_bwd: null pointer");
    if (!dtype_ok(dtype)) return fail(DG_E_ARG, "dg_embed_sym_bwd: unknown dtype %d", dtype);
    if (B < 1 || !embed_shape_ok(N, E, H, C, act))
        return fail(DG_E_SHAPE, "dg_embed_sym_bwd: unsupported B=%d N=%d E=%d H=%d C=%d act=%d", B, N, E, H, C, act);
    if (workspace_bytes < dg_embed_sym_workspace_bytes(B, N))
        return fail(DG_E_WORKSPACE, "dg_embed_sym_bwd: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int tpm = (N * (N + 1) / 2 + kPairs - 1) / kPairs;
    const int grid = embed_grid(B * tpm, kBwdPerCu);
    float* part = static_cast<float*>(workspace);
    constexpr int lds_bytes = (64 * kHid + 64 * kC + 64 * kD1Pitch + 64 * kMaxE + kHid * 16) * 4 + kPairs * 2 * 4;
    ProfScope prof(DG_K_EMBED_SYM, stream);
    note_forward(static_cast<int64_t>(B) * N * N);
#define BWD_D(T, EP_, ACT_, DA_)                                                                                  \
    {                                                                                                             \
        DG_OPT_IN_LDS((&embed_sym_bwd_kernel<T, EP_, ACT_, DA_>), lds_bytes);                                      \
        hipLaunchKernelGGL((embed_sym_bwd_kernel<T, EP_, ACT_, DA_>), dim3(grid), dim3(256), lds_bytes, stream, a, \
                           w1, b1, w2_packed, w2_dgrad_packed, b2, static_cast<const T*>(g), da, part, B, N, E, tpm); \
    }
#define BWD_A(T, EP_, ACT_)                                                                                       \
    {                                                                                                             \
        if (da) BWD_D(T, EP_, ACT_, true) else BWD_D(T, EP_, ACT_, false)                                         \
    }
#define BWD(T, EP_)                                                                                               \
    switch (act) {                                                                                                \
        case kRelu: BWD_A(T, EP_, kRelu) break;                                                                   \
        case kLeaky: BWD_A(T, EP_, kLeaky) break;                                                                 \
        case kSigmoid: BWD_A(T, EP_, kSigmoid) break;                                                             \
        default: BWD_A(T, EP_, kTanh) break;                                                                      \
    }
    if (dtype == DG_DTYPE_BF16) {
        if (E <= 8) BWD(bf16_t, 8) else BWD(bf16_t, 16)
    } else {
        if (E <= 8) BWD(float, 8) else BWD(float, 16)
    }
#undef BWD_A
#undef BWD_D
#undef BWD
    hipLaunchKernelGGL(embed_finish_kernel, dim3((BwdPart::kTotal + 31) / 32), dim3(1024), 0, stream, part, grid, dw1, db1, dw2,
                       db2, E);
    return check_launch("dg_embed_sym_bwd");
}

extern "C" int dg_embed_sym_bwd2(const float* a, const float* w1, const float* b1, const float* w2_packed,
                                 const float* w2_dgrad_packed, const float* b2, const void* g, const float* t, void* gg,
                                 float* gw1, float* gw2, void* workspace, size_t workspace_bytes, int B, int N, int E,
                                 int H, int C, int act, int dtype, dg_stream_t stream_) {
    if (!a || !w1 || !b1 || !w2_packed || !w2_dgrad_packed || !b2 || !g || !t || !gg || !gw1 || !gw2 || !workspace)
        return fail(DG_E_ARG, "dg_embed_sym_bwd2: null pointer");
    if (!dtype_ok(dtype)) return fail(DG_E_ARG, "dg_embed_sym_bwd2: unknown dtype %d", dtype);
    if (act != kRelu && act != kLeaky)
        return fail(DG_E_ARG, "dg_embed_sym_bwd2: only piecewise-linear activations (relu, leaky) have this closed form");
    if (B < 1 || !embed_shape_ok(N, E, H, C, act))
        return fail(DG_E_SHAPE, "dg_embed_sym_bwd2: unsupported B=%d N=%d E=%d H=%d C=%d act=%d", B, N, E, H, C, act);
    if (workspace_bytes < dg_embed_sym_workspace_bytes(B, N))
        return fail(DG_E_WORKSPACE, "dg_embed_sym_bwd2: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int tpm = (N * (N + 1) / 2 + kPairs - 1) / kPairs;
    const int grid = embed_grid(B * tpm, kBwdPerCu);
    float* part = static_cast<float*>(workspace);
    constexpr int lds_bytes = (64 * kHid + 64 * kC + 64 * kHid + 2 * 64 * kMaxE + kPairs * kC) * 4 + kPairs * 2 * 4;
    ProfScope prof(DG_K_EMBED_SYM, stream);
    note_forward(static_cast<int64_t>(B) * N * N);
#define BWD2_A(T, EP_, ACT_)                                                                                       \
    {                                                                                                              \
        DG_OPT_IN_LDS((&embed_sym_bwd2_kernel<T, EP_, ACT_>), lds_bytes);                                           \
        hipLaunchKernelGGL((embed_sym_bwd2_kernel<T, EP_, ACT_>), dim3(grid), dim3(256), lds_bytes, stream, a, w1,  \
                           b1, w2_packed, w2_dgrad_packed, b2, static_cast<const T*>(g), t, static_cast<T*>(gg),   \
                           part, B, N, E, tpm);                                                                    \
    }
#define BWD2(T, EP_)                                                     \
    if (act == kRelu) BWD2_A(T, EP_, kRelu) else BWD2_A(T, EP_, kLeaky)
    if (dtype == DG_DTYPE_BF16) {
        if (E <= 8) BWD2(bf16_t, 8) else BWD2(bf16_t, 16)
    } else {
        if (E <= 8) BWD2(float, 8) else BWD2(float, 16)
    }
#undef BWD2
#undef BWD2_A
    hipLaunchKernelGGL(embed_finish_kernel, dim3((BwdPart::kTotal + 31) / 32), dim3(1024), 0, stream, part, grid, gw1,
                       static_cast<float*>(nullptr), gw2, static_cast<float*>(nullptr), E);
    return check_launch("dg_embed_sym_bwd2");
}

extern "C" size_t dg_onehot_embed_workspace_bytes(int E, int C) {
    return E >= 1 && E <= kMaxE && C == kC ? static_cast<size_t>(kOneHotBlocks) * E * C * sizeof(float) : 0;
}

extern "C" int dg_onehot_embed_fwd(const int* labels, const float* table, void* out, int B, int N, int E, int C, int dtype,
                                   dg_stream_t stream_) {
    if (!labels || !table || !out) return fail(DG_E_ARG, "dg_onehot_embed_fwd: null pointer");
    if (!dtype_ok(dtype)) return fail(DG_E_ARG, "dg_onehot_embed_fwd: unknown dtype %d", dtype);
    if (B < 0 || N < 1 || E < 1 || E > kMaxE || C != kC)
        return fail(DG_E_SHAPE, "dg_onehot_embed_fwd: unsupported N=%d E=%d C=%d (need E<=16, C=128)", N, E, C);
    if (B == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t rows = static_cast<int64_t>(B) * N * N;
    const int grid = static_cast<int>(rows / 8 + 1 < 4096 ? rows / 8 + 1 : 4096);
    ProfScope prof(DG_K_EMBED_SYM, stream);
    note_forward(static_cast<int64_t>(B) * N * N);
    if (dtype == DG_DTYPE_BF16)
        hipLaunchKernelGGL((onehot_embed_fwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, stream, labels, table,
                           static_cast<bf16_t*>(out), rows, N, E);
    else
        hipLaunchKernelGGL((onehot_embed_fwd_kernel<float>), dim3(grid), dim3(256), 0, stream, labels, table,
                           static_cast<float*>(out), rows, N, E);
    return check_launch("dg_onehot_embed_fwd");
}

extern "C" int dg_onehot_embed_bwd(const int* labels, const void* g, float* dtable, void* workspace, size_t workspace_bytes,
                                   int B, int N, int E, int C, int dtype, dg_stream_t stream_) {
    if (!labels || !g || !dtable || !workspace) return fail(DG_E_ARG, "dg_onehot_embed_bwd: null pointer");
    if (!dtype_ok(dtype)) return fail(DG_E_ARG, "dg_onehot_embed_bwd: unknown dtype %d", dtype);
    if (B < 1 || N < 1 || E < 1 || E > kMaxE || C != kC)
        return fail(DG_E_SHAPE, "dg_onehot_embed_bwd: unsupported B=%d N=%d E=%d C=%d", B, N, E, C);
    if (workspace_bytes < dg_onehot_embed_workspace_bytes(E, C)) return fail(DG_E_WORKSPACE, "dg_onehot_embed_bwd: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t rows = static_cast<int64_t>(B) * N * N;
    const int grid = static_cast<int>(rows / 8 + 1 < kOneHotBlocks ? rows / 8 + 1 : kOneHotBlocks);
    float* part = static_cast<float*>(workspace);
    ProfScope prof(DG_K_EMBED_SYM, stream);
    note_forward(static_cast<int64_t>(B) * N * N);
#define BWD(T, EP_)                                                                                         \
    hipLaunchKernelGGL((onehot_embed_bwd_kernel<T, EP_>), dim3(grid), dim3(256), 0, stream, labels,          \
                       static_cast<const T*>(g), part, rows, N, E);
    if (dtype == DG_DTYPE_BF16) {
        if (E <= 8) { BWD(bf16_t, 8) } else { BWD(bf16_t, 16) }
    } else {
        if (E <= 8) { BWD(float, 8) } else { BWD(float, 16) }
    }
#undef BWD
    launch_ln_finish(part, grid, 1, E * C, dtable, nullptr, stream);      // column sums of the [grid][E C] partials, fixed order
    return check_launch("dg_onehot_embed_bwd");
}
