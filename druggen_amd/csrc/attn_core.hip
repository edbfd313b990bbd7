// Graph attention core of DrugGEN's MHA (reference src/model/layers.py:119-134)
// as three streaming kernels for gfx950: forward, backward, backward-of-backward.
//
//   s_ij = alpha q_i k_j (e_ij^2 + e_ij)     p_ij = softmax_j s_ij     o_i = sum_j p_ij v_j
//
// Everything is elementwise in the channel c; there is no contraction over the
// head dimension, hence no MFMA work: the kernels are bound by the HBM traffic
// of the [B,N,N,C] tensors (fwd: read e, write s; bwd: read e, ws, write de;
// bwd2: read e, ws, te, write ge, gws).
//
// Work decomposition (all three kernels)
//   * a wave owns one molecule b and one 32-channel slice (8 float4 "quads");
//     lane = (phase, quad): quad = lane & 7 selects the float4, phase = lane >> 3
//     selects the neighbours j = phase, phase+8, ... (JPL of them per lane).
//     One wave-wide float4 load therefore covers 8 neighbours x 128 contiguous
//     bytes -- full 128 B lines, 16 B per lane.
//   * the wave walks over the query rows i it is responsible for.  The softmax
//     over j is a per-lane loop over its JPL slots plus a 3-step xor butterfly
//     across the 8 phases (lane bits 3..5); no LDS, no barrier.
//   * k_j, v_j (and in the backward the accumulators dk_j, dv_j, which are sums
//     over i) belong to fixed (j, channel) pairs, i.e. to fixed lanes: they stay
//     in registers for the whole walk, so the cross-row reductions need no
//     atomics and are bit-reproducible.  Row groups of one block are combined
//     once at the end through LDS in a fixed order.
//   * narrow models (C < 32) use fewer quads per slice (LQS) and more phases.
#include "bf16.h"
#include "traversal.h"

#include <type_traits>

namespace dg {
namespace {

constexpr float kNegBig = -3.0e38f;

template <int LQS, int JPL>
struct Lane {
    static constexpr int QS = 1 << LQS;   // quads per slice
    static constexpr int P = 64 >> LQS;   // neighbour phases per wave
    int quad, phase;
    bool cok;          // this lane's channels exist
    int c0;            // channel offset (clamped to 0 when !cok)
    unsigned off[JPL]; // element offset of (neighbour slot, channel) inside one [N,C] row block
    bool jok[JPL];     // slot holds a real neighbour
    __device__ __forceinline__ Lane(int lane, int slice, int N, int C) {
        quad = lane & (QS - 1);
        phase = lane >> LQS;
        const int cq = slice * QS + quad;
        cok = cq * 4 < C;
        c0 = cok ? cq * 4 : 0;
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            const int j = phase + t * P;
            jok[t] = j < N;
            off[t] = static_cast<unsigned>((jok[t] ? j : 0) * C + c0);   // clamped: loads stay in bounds, results are masked
        }
    }
};

// ---------------------------------------------------------------- forward ----
template <typename T, int LQS, int JPL>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                      const T* __restrict__ v, const T* __restrict__ e,
                                                      T* __restrict__ s, T* __restrict__ o, int N, int C,
                                                      float alpha, int RG, int B, int reverse) {
    constexpr int QS = 1 << LQS;
    const int lane = threadIdx.x & 63;
    const int slice = blockIdx.y * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (slice * QS * 4 >= C) return;  // wave-uniform; no barriers in this kernel
    // XCD-aware placement (speed only): workgroup id -> XCD is round robin (id % 8), and each XCD has its own L2.
    // All RG row groups of a molecule get ids with the same residue mod 8, consecutive in that XCD's dispatch order,
    // so the molecule's k, v rows are fetched from HBM once instead of once per XCD (PMC: reads 1.30x -> ~1.0x
    // of the algorithmic bytes).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    // (reverse, traversal.h: molecules in descending order -- b -> 8 ceil(B / 8) - 1 - b keeps a molecule's workgroups on one XCD)
    int b = (slot / RG) * 8 + xcd;
    const int rg = slot % RG;
    if (reverse) b = (B + 7) / 8 * 8 - 1 - b;
    if (b >= B) return;
    const Lane<LQS, JPL> L(lane, slice, N, C);
    const size_t NC = static_cast<size_t>(N) * C;

    float4 kk[JPL], vv[JPL];
#pragma unroll
    for (int t = 0; t < JPL; ++t) {
        kk[t] = ld4(k + b * NC + L.off[t]);
        vv[t] = ld4(v + b * NC + L.off[t]);
    }
    constexpr bool PF = false;   // a one-row-ahead prefetch measured slower here (4 waves per SIMD already overlap rows)
    typedef typename raw4<T>::type Raw;
    Raw re[JPL], rq;
    auto request = [&](int i) {
        const size_t row = static_cast<size_t>(b) * N + i;
        rq = ld_raw(q + row * C + L.c0);
#pragma unroll
        for (int t = 0; t < JPL; ++t) re[t] = ld_raw_stream(e + row * NC + L.off[t]);
    };
    if (PF && rg < N) request(rg);
    for (int i = rg; i < N; i += RG) {
        const size_t row = static_cast<size_t>(b) * N + i;
        if (!PF) request(i);
        const float4 aq = alpha * cvt_raw(rq);
        T* sr = s + row * NC;
        float4 sv[JPL];
#pragma unroll
        for (int t = 0; t < JPL; ++t) sv[t] = cvt_raw(re[t]);
        if (PF && i + RG < N) request(i + RG);
        float4 m = f4(kNegBig);
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            const float4 ee = sv[t];
            sv[t] = aq * kk[t] * fma4(ee, ee, ee);
            if (L.jok[t]) {
                m = max4(m, sv[t]);
                if (L.cok && s) st4_stream(sr + L.off[t], sv[t]);
            }
        }
        m = xor_max4<QS>(m);
        float4 l = f4(0.f), acc = f4(0.f);
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            const float4 pe = L.jok[t] ? exp4(sv[t] - m) : f4(0.f);
            l += pe;
            acc = fma4(pe, vv[t], acc);
        }
        l = xor_sum4<QS>(l);
        acc = xor_sum4<QS>(acc);
        if (L.phase == 0 && L.cok) st4(o + row * C + L.c0, acc * rcp4(l));
    }
}

// --------------------------------------------------------------- backward ----
// ADDE: `add_e` [B,N,N,C] is added to de on its way out (the adjoint that the second order of the gradient penalty
// hands to the first-order pass, loss.py:32-47): one more read stream instead of a 3-pass elementwise add afterwards.
template <typename T, int LQS, int JPL, int RW, bool ADDE = false>
__global__ __launch_bounds__(RW * 64, (JPL <= 6 ? 2 : 1)) void attn_bwd_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
    const T* __restrict__ e, const T* __restrict__ ws, const T* __restrict__ wo, const T* __restrict__ add_e,
    T* __restrict__ dq, T* __restrict__ dk, T* __restrict__ dv, T* __restrict__ de, int N, int C,
    float alpha, int SL, int B, int reverse) {
    constexpr int QS = 1 << LQS;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // k_j, v_j in the lane layout, shared by the RW waves: kv[2][JPL][64]; then the reduction area
    float4* kv = reinterpret_cast<float4*>(smem_raw);
    float4* red = kv + 2 * JPL * 64;  // [(RW-1)][2*JPL][64]
    const int lane = threadIdx.x & 63;
    const int rw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // fp32 + ADDE: the row's add_e slots travel HBM -> LDS by LDS-DMA into a wave-private stash (JPL KiB per wave, lane-
    // linear) when the row is requested and are read back one slot at a time right before their store.  Held in
    // registers from the request to the store they were 4 JPL registers too many: 156 B / lane of scratch, each slot
    // spilled after its load and reloaded at its store (279 us against 199 us for the plain kernel at configs[1]).
    constexpr bool STASH = ADDE && std::is_same<T, float>::value;
    float4* stash = red + (RW - 1) * 2 * JPL * 64 + rw * JPL * 64;      // [RW][JPL][64] behind the reduction area
    // XCD-aware placement (speed only; workgroup id -> XCD is id % 8): the SL channel slices of a molecule get
    // consecutive ids on ONE XCD, so they run at the same time behind the same L2.  With bf16 rows a slice covers
    // 64 of the 128 bytes of a cache line: the other half is then an L2 hit instead of a second HBM fetch.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int b = (slot / SL) * 8 + xcd;
    const int slice = slot % SL;
    if (reverse) b = (B + 7) / 8 * 8 - 1 - b;      // molecules in descending order (traversal.h)
    if (b >= B) return;   // block-uniform
    const Lane<LQS, JPL> L(lane, slice, N, C);
    const size_t NC = static_cast<size_t>(N) * C;

    if (rw == 0) {
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            kv[(0 * JPL + t) * 64 + lane] = ld4(k + b * NC + L.off[t]);
            kv[(1 * JPL + t) * 64 + lane] = ld4(v + b * NC + L.off[t]);
        }
    }
    __syncthreads();
    float4 dkk[JPL], dvv[JPL];
#pragma unroll
    for (int t = 0; t < JPL; ++t) dkk[t] = dvv[t] = f4(0.f);
    // bf16: the kernel is bound by per-row latency (2 waves per SIMD at this register count), not by bytes: the
    // operands of row i + RW are requested -- still packed, 2 registers per slot -- before the arithmetic of row i.
    // fp32 rows cost 4 registers per slot: no room for the second set, and that variant sits at the HBM roof already.
    constexpr bool PF = !std::is_same<T, float>::value;
    typedef typename raw4<T>::type Raw;
    Raw re[JPL], rws[JPL], rae[(ADDE && !STASH) ? JPL : 1], rq, rwo;
    auto request = [&](int i) {
        const size_t row = static_cast<size_t>(b) * N + i;
        if (STASH) {      // first: every later wait for a register load of this row then covers them
            const unsigned sb = lds_byte_address(stash);
#pragma unroll
            for (int t = 0; t < JPL; ++t)
                dma16_async(reinterpret_cast<const float*>(add_e + row * NC + L.off[t]), sb + t * 1024);
        }
        rq = ld_raw(q + row * C + L.c0);
        rwo = ld_raw(wo + row * C + L.c0);
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            re[t] = ld_raw_stream(e + row * NC + L.off[t]);
            if (ws) rws[t] = ld_raw_stream(ws + row * NC + L.off[t]);
            if (ADDE && !STASH) rae[(ADDE && !STASH) ? t : 0] = ld_raw_stream(add_e + row * NC + L.off[t]);
        }
    };
    if (PF && rw < N) request(rw);
    for (int i = rw; i < N; i += RW) {
        const size_t row = static_cast<size_t>(b) * N + i;
        int kl = lane;                       // opaque per row: keeps the LDS operand reads inside the
        asm volatile("" : "+v"(kl));         // loop instead of hoisting 2*JPL float4 into registers
        if (!PF) request(i);
        const float4 aq = alpha * cvt_raw(rq);
        const float4 woi = cvt_raw(rwo);
        T* der = de + row * NC;
        float4 ee[JPL], wss[JPL], pe[JPL];
        Raw hae[(ADDE && !STASH) ? JPL : 1];      // still packed: converted at the store
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            ee[t] = cvt_raw(re[t]);
            wss[t] = ws ? cvt_raw(rws[t]) : f4(0.f);
            if (ADDE && !STASH) hae[(ADDE && !STASH) ? t : 0] = rae[(ADDE && !STASH) ? t : 0];
        }
        if (PF && i + RW < N) request(i + RW);
        float4 m = f4(kNegBig);
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            pe[t] = aq * kv[(0 * JPL + t) * 64 + kl] * fma4(ee[t], ee[t], ee[t]);
            if (L.jok[t]) m = max4(m, pe[t]);
        }
        m = xor_max4<QS>(m);
        float4 l = f4(0.f), A = f4(0.f);
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            pe[t] = L.jok[t] ? exp4(pe[t] - m) : f4(0.f);
            l += pe[t];
            A = fma4(pe[t], woi * kv[(1 * JPL + t) * 64 + kl], A);
        }
        l = xor_sum4<QS>(l);
        A = xor_sum4<QS>(A);
        const float4 inv = rcp4(l);
        const float4 abar = A * inv;
        float4 dqa = f4(0.f);
        if (STASH) wait_all_vmem();      // the row's DMA (issued a row of arithmetic ago) has landed; own lanes only: no barrier
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            const float4 kk = kv[(0 * JPL + t) * 64 + kl];
            const float4 p = pe[t] * inv;
            float4 ds = fma4(p, woi * kv[(1 * JPL + t) * 64 + kl] - abar, wss[t]);
            if (!L.jok[t]) ds = f4(0.f);
            dvv[t] = fma4(p, woi, dvv[t]);
            const float4 g = fma4(ee[t], ee[t], ee[t]);
            const float4 dsg = ds * g;
            dqa = fma4(dsg, kk, dqa);
            dkk[t] = fma4(dsg, aq, dkk[t]);
            const float4 g1 = fma4(f4(2.f), ee[t], f4(1.f));
            float4 dev = ds * aq * kk * g1;
            if (STASH) dev += stash[t * 64 + kl];
            else if (ADDE) dev += cvt_raw(hae[(ADDE && !STASH) ? t : 0]);
            if (L.jok[t] && L.cok) st4_stream(der + L.off[t], dev);
            __builtin_amdgcn_sched_barrier(0);   // one slot at a time: bounds the live temporaries
        }
        dqa = xor_sum4<QS>(dqa);
        if (L.phase == 0 && L.cok) st4(dq + row * C + L.c0, alpha * dqa);
    }
    // fixed-order combine of the RW row groups (bit-reproducible)
    if (RW > 1) {
        if (rw > 0) {
#pragma unroll
            for (int t = 0; t < JPL; ++t) {
                red[((rw - 1) * 2 * JPL + 2 * t) * 64 + lane] = dkk[t];
                red[((rw - 1) * 2 * JPL + 2 * t + 1) * 64 + lane] = dvv[t];
            }
        }
        __syncthreads();
    }
    if (rw == 0) {
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            float4 a = dkk[t], c = dvv[t];
            for (int w = 0; w < RW - 1; ++w) {
                a += red[(w * 2 * JPL + 2 * t) * 64 + lane];
                c += red[(w * 2 * JPL + 2 * t + 1) * 64 + lane];
            }
            if (L.jok[t] && L.cok) {
                st4(dk + b * NC + L.off[t], a);
                st4(dv + b * NC + L.off[t], c);
            }
        }
    }
}

// ---------------------------------------------------- backward of backward ----
// Inputs of the first-order backward: (q,k,v,e,ws,wo); (tq,tk,tv,te) are the
// adjoints of its outputs (dq,dk,dv,de).  See tests/kernel_math.py::attn_core_bwd2
// for the closed form (verified against autograd in float64).
template <typename T, int LQS, int JPL, int RW>
__global__ __launch_bounds__(RW * 64) void attn_bwd2_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
    const T* __restrict__ e, const T* __restrict__ ws, const T* __restrict__ wo,
    const T* __restrict__ tq, const T* __restrict__ tk, const T* __restrict__ tv,
    const T* __restrict__ te, T* __restrict__ gq, T* __restrict__ gk, T* __restrict__ gv,
    T* __restrict__ ge, T* __restrict__ gws, T* __restrict__ gwo, int N, int C, float alpha, int SL, int B, int reverse) {
    constexpr int QS = 1 << LQS;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // per-neighbour operands live in LDS (read-only, shared by the RW waves):
    // kv[4][JPL][64] = k, v, tk, tv in the lane layout; then the reduction area.
    float4* kv = reinterpret_cast<float4*>(smem_raw);
    float4* red = kv + 4 * JPL * 64;  // [(RW-1)][2*JPL][64]
    const int lane = threadIdx.x & 63;
    const int rw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware placement (speed only; workgroup id -> XCD is id % 8): the SL channel slices of a molecule get
    // consecutive ids on ONE XCD, so they run at the same time behind the same L2.  With bf16 rows a slice covers
    // 64 of the 128 bytes of a cache line: the other half is then an L2 hit instead of a second HBM fetch.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int b = (slot / SL) * 8 + xcd;
    const int slice = slot % SL;
    if (reverse) b = (B + 7) / 8 * 8 - 1 - b;      // molecules in descending order (traversal.h)
    if (b >= B) return;   // block-uniform
    const Lane<LQS, JPL> L(lane, slice, N, C);
    const size_t NC = static_cast<size_t>(N) * C;

    if (rw == 0) {
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            const size_t off = b * NC + L.off[t];
            kv[(0 * JPL + t) * 64 + lane] = ld4(k + off);
            kv[(1 * JPL + t) * 64 + lane] = ld4(v + off);
            kv[(2 * JPL + t) * 64 + lane] = ld4(tk + off);
            kv[(3 * JPL + t) * 64 + lane] = ld4(tv + off);
        }
    }
    __syncthreads();
    float4 gkk[JPL], gvv[JPL];
#pragma unroll
    for (int t = 0; t < JPL; ++t) gkk[t] = gvv[t] = f4(0.f);

    constexpr bool PF = !std::is_same<T, float>::value;   // bf16: request row i + RW (packed) before the math of row i
    typedef typename raw4<T>::type Raw;
    Raw re[JPL], rws[JPL], rte[JPL], rq, rwo, rtq;
    auto request = [&](int i) {
        const size_t row = static_cast<size_t>(b) * N + i;
        rq = ld_raw(q + row * C + L.c0);
        rwo = ld_raw(wo + row * C + L.c0);
        rtq = ld_raw(tq + row * C + L.c0);
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            const unsigned off = L.off[t];
            re[t] = ld_raw_stream(e + row * NC + off);
            if (ws) rws[t] = ld_raw_stream(ws + row * NC + off);
            rte[t] = ld_raw_stream(te + row * NC + off);
        }
    };
    if (PF && rw < N) request(rw);
    for (int i = rw; i < N; i += RW) {
        const size_t row = static_cast<size_t>(b) * N + i;
        int kl = lane;
        asm volatile("" : "+v"(kl));
        if (!PF) request(i);
        const float4 qi = cvt_raw(rq);
        const float4 aq = alpha * qi;
        const float4 woi = cvt_raw(rwo);
        const float4 tqi = cvt_raw(rtq);
        float4 ee[JPL], wss[JPL], tee[JPL], pe[JPL], sd[JPL];
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            ee[t] = cvt_raw(re[t]);
            wss[t] = ws ? cvt_raw(rws[t]) : f4(0.f);
            tee[t] = cvt_raw(rte[t]);
        }
        if (PF && i + RW < N) request(i + RW);
        float4 m = f4(kNegBig);
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            const float4 kk = kv[(0 * JPL + t) * 64 + kl];
            pe[t] = aq * kk * fma4(ee[t], ee[t], ee[t]);
            if (L.jok[t]) m = max4(m, pe[t]);
        }
        m = xor_max4<QS>(m);
        float4 l = f4(0.f), A = f4(0.f);
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            const float4 vv = kv[(1 * JPL + t) * 64 + kl];
            pe[t] = L.jok[t] ? exp4(pe[t] - m) : f4(0.f);
            l += pe[t];
            A = fma4(pe[t], woi * vv, A);
        }
        l = xor_sum4<QS>(l);
        A = xor_sum4<QS>(A);
        const float4 inv = rcp4(l);
        const float4 abar = A * inv;
        // tangent of s along t, and m = sum_j p sdot
        float4 mm = f4(0.f);
        T* gwr = gws + row * NC;
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            const float4 kk = kv[(0 * JPL + t) * 64 + kl];
            const float4 tkk = kv[(2 * JPL + t) * 64 + kl];
            const float4 g = fma4(ee[t], ee[t], ee[t]);
            const float4 g1 = fma4(f4(2.f), ee[t], f4(1.f));
            pe[t] = pe[t] * inv;  // p
            sd[t] = alpha * (g * fma4(tqi, kk, qi * tkk) + qi * kk * g1 * tee[t]);
            if (L.jok[t] && L.cok && gws) st4_stream(gwr + L.off[t], sd[t]);
            mm = fma4(pe[t], sd[t], mm);
        }
        mm = xor_sum4<QS>(mm);
        float4 od = f4(0.f), PB = f4(0.f);
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            const float4 vv = kv[(1 * JPL + t) * 64 + kl];
            const float4 tvv = kv[(3 * JPL + t) * 64 + kl];
            const float4 a = woi * vv;
            const float4 pdot = pe[t] * (sd[t] - mm);
            od = fma4(pdot, vv, fma4(pe[t], tvv, od));
            const float4 pbar = sd[t] * (a - abar) - mm * a + woi * tvv;
            PB = fma4(pe[t], pbar, PB);
        }
        od = xor_sum4<QS>(od);
        PB = xor_sum4<QS>(PB);
        if (L.phase == 0 && L.cok) st4(gwo + row * C + L.c0, od);
        float4 gqa = f4(0.f);
        T* ger = ge + row * NC;
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            const float4 kk = kv[(0 * JPL + t) * 64 + kl];
            const float4 vv = kv[(1 * JPL + t) * 64 + kl];
            const float4 tkk = kv[(2 * JPL + t) * 64 + kl];
            const float4 tvv = kv[(3 * JPL + t) * 64 + kl];
            const float4 p = pe[t];
            const float4 a = woi * vv;
            const float4 g = fma4(ee[t], ee[t], ee[t]);
            const float4 g1 = fma4(f4(2.f), ee[t], f4(1.f));
            float4 ds = fma4(p, a - abar, wss[t]);
            if (!L.jok[t]) ds = f4(0.f);
            const float4 pdot = p * (sd[t] - mm);
            const float4 pbar = sd[t] * (a - abar) - mm * a + woi * tvv;
            const float4 sbar = p * (pbar - PB);
            const float4 g1te = g1 * tee[t];
            // gq_i += sbar alpha k g + ds alpha (tk g + k g1 te)
            gqa += sbar * kk * g + ds * fma4(tkk, g, kk * g1te);
            // gk_j += sbar alpha q g + ds alpha (tq g + q g1 te)
            gkk[t] += sbar * aq * g + alpha * (ds * fma4(tqi, g, qi * g1te));
            gvv[t] = fma4(pdot, woi, gvv[t]);
            // ge = sbar alpha q k g1 + ds alpha (tq k g1 + q tk g1 + 2 q k te)
            const float4 gev = sbar * aq * kk * g1 +
                               alpha * (ds * (g1 * fma4(tqi, kk, qi * tkk) + 2.f * (qi * kk * tee[t])));
            if (L.jok[t] && L.cok) st4_stream(ger + L.off[t], gev);
            __builtin_amdgcn_sched_barrier(0);
        }
        gqa = xor_sum4<QS>(gqa);
        if (L.phase == 0 && L.cok) st4(gq + row * C + L.c0, alpha * gqa);
    }
    if (RW > 1) {
        if (rw > 0) {
#pragma unroll
            for (int t = 0; t < JPL; ++t) {
                red[((rw - 1) * 2 * JPL + 2 * t) * 64 + lane] = gkk[t];
                red[((rw - 1) * 2 * JPL + 2 * t + 1) * 64 + lane] = gvv[t];
            }
        }
        __syncthreads();
    }
    if (rw == 0) {
#pragma unroll
        for (int t = 0; t < JPL; ++t) {
            float4 a = gkk[t], c = gvv[t];
            for (int w = 0; w < RW - 1; ++w) {
                a += red[(w * 2 * JPL + 2 * t) * 64 + lane];
                c += red[(w * 2 * JPL + 2 * t + 1) * 64 + lane];
            }
            if (L.jok[t] && L.cok) {
                st4(gk + b * NC + L.off[t], a);
                st4(gv + b * NC + L.off[t], c);
            }
        }
    }
}

// ---------------------------------------------------------------- dispatch ----
struct Geometry {
    int lqs, jpl, slices;
};

// narrow = true (second-order kernel): for N > 48 a wave takes 16 channels x 16 neighbour phases instead of 32 x 8, so
// a lane keeps 6 neighbour slots instead of 12 -- the 12-slot second-order kernel needs > 500 registers per lane and
// spills 400 of them (N = 90: 1071 -> 433 us); the slices of a molecule share an XCD, so the half lines meet in L2.
// The first-order backward is faster with 12 slots (216 vs 289 us).
bool pick_geometry(int N, int C, Geometry* g, bool narrow = false) {
    if (C < 8 || (C & 3) || N < 1) return false;
    const int cq = C / 4;
    g->lqs = cq >= 5 ? 3 : (cq >= 3 ? 2 : 1);
    if (narrow && g->lqs == 3 && N > 48 && N <= 96 && (cq & 3) == 0) g->lqs = 2;   // N = 45: 365 vs 428 us, stays wide
    const int qs = 1 << g->lqs, P = 64 >> g->lqs;
    g->slices = (cq + qs - 1) / qs;
    const int need = (N + P - 1) / P;
    const int table[] = {1, 2, 3, 6, 12};
    for (int t : table) {
        if (t >= need) {
            g->jpl = t;
            // only the instantiated combinations
            if (g->lqs == 1 && t > 3) return false;
            if (g->lqs == 2 && t > 6) return false;
            return true;
        }
    }
    return false;
}

constexpr int kRW = 4;  // row groups (waves) per backward block

#define DG_FOR_GEOMETRY(M)                                                          \
    M(1, 1) M(1, 2) M(1, 3) M(2, 1) M(2, 2) M(2, 3) M(2, 6) M(3, 1) M(3, 2) M(3, 3) \
        M(3, 6) M(3, 12)

}  // namespace
}  // namespace dg

using namespace dg;

extern "C" int dg_attn_core_fwd(const void* q_, const void* k_, const void* v_, const void* e_, void* s_, void* o_,
                                int B, int N, int C, float alpha, int dtype, dg_stream_t stream_) {
    if (!q_ || !k_ || !v_ || !e_ || !o_) return fail(DG_E_ARG, "dg_attn_core_fwd: null pointer");  // s may be NULL
    if (!dtype_ok(dtype)) return fail(DG_E_ARG, "dg_attn_core_fwd: unknown dtype %d", dtype);
    Geometry g;
    if (B < 0 || !pick_geometry(N, C, &g))
        return fail(DG_E_SHAPE, "dg_attn_core_fwd: unsupported shape B=%d N=%d C=%d (need C%%4==0, C>=8, N<=96)", B, N, C);
    if (B == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int wpb = g.slices < 4 ? g.slices : 4;
    const int rows_per_wave = 3;   // measured on MI355X: 3 rows/wave (RG = N/3) beats 9 by ~8 %
    int RG = (N + rows_per_wave - 1) / rows_per_wave;
    if (RG > N) RG = N;
    dim3 grid(static_cast<unsigned>((B + 7) / 8 * 8) * RG, (g.slices + wpb - 1) / wpb), block(64 * wpb);
    ProfScope prof(DG_K_ATTN_FWD, stream);
    const int reverse = take_direction(static_cast<int64_t>(B) * N * N);      // per-molecule results: any order
#define LAUNCH_T(T, LQS, JPL)                                                                                     \
    hipLaunchKernelGGL((attn_fwd_kernel<T, LQS, JPL>), grid, block, 0, stream, static_cast<const T*>(q_),         \
                       static_cast<const T*>(k_), static_cast<const T*>(v_), static_cast<const T*>(e_),          \
                       static_cast<T*>(s_), static_cast<T*>(o_), N, C, alpha, RG, B, reverse);
#define LAUNCH(LQS, JPL)                                       \
    if (g.lqs == LQS && g.jpl == JPL) {                        \
        if (dtype == DG_DTYPE_BF16) { LAUNCH_T(bf16_t, LQS, JPL) } \
        else { LAUNCH_T(float, LQS, JPL) }                     \
    }
    DG_FOR_GEOMETRY(LAUNCH)
#undef LAUNCH
#undef LAUNCH_T
    return check_launch("dg_attn_core_fwd");
}

extern "C" int dg_attn_core_bwd(const void* q_, const void* k_, const void* v_, const void* e_, const void* ws_,
                                const void* wo_, void* dq_, void* dk_, void* dv_, void* de_, int B, int N, int C,
                                float alpha, int dtype, dg_stream_t stream_) {
    return dg_attn_core_bwd_add(q_, k_, v_, e_, ws_, wo_, nullptr, dq_, dk_, dv_, de_, B, N, C, alpha, dtype, stream_);
}

extern "C" int dg_attn_core_bwd_add(const void* q_, const void* k_, const void* v_, const void* e_, const void* ws_,
                                    const void* wo_, const void* add_e_, void* dq_, void* dk_, void* dv_, void* de_, int B,
                                    int N, int C, float alpha, int dtype, dg_stream_t stream_) {
    if (!q_ || !k_ || !v_ || !e_ || !wo_ || !dq_ || !dk_ || !dv_ || !de_)
        return fail(DG_E_ARG, "dg_attn_core_bwd: null pointer");  // ws may be NULL (= zeros)
    if (!dtype_ok(dtype)) return fail(DG_E_ARG, "dg_attn_core_bwd: unknown dtype %d", dtype);
    Geometry g;
    if (B < 0 || !pick_geometry(N, C, &g))
        return fail(DG_E_SHAPE, "dg_attn_core_bwd: unsupported shape B=%d N=%d C=%d", B, N, C);
    if (B == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool rw8 = false;      // (eight row-group waves per workgroup measured slower than kRW: not offered)
    dim3 grid(static_cast<unsigned>((B + 7) / 8 * 8) * g.slices), block(64 * (rw8 ? 8 : kRW));
    ProfScope prof(DG_K_ATTN_BWD, stream);
    const int reverse = take_direction(static_cast<int64_t>(B) * N * N);      // per-molecule results: any order
#define LAUNCH_T(T, LQS, JPL, RW_)                                                                              \
    {                                                                                                           \
        constexpr int lds = (2 * JPL + (RW_ - 1) * 2 * JPL + RW_ * JPL) * 64 * 16;   /* + the add_e stash */    \
        if (add_e_) {                                                                                           \
            DG_OPT_IN_LDS((&attn_bwd_kernel<T, LQS, JPL, RW_, true>), lds);                                      \
            hipLaunchKernelGGL((attn_bwd_kernel<T, LQS, JPL, RW_, true>), grid, block, lds, stream,              \
                               static_cast<const T*>(q_), static_cast<const T*>(k_), static_cast<const T*>(v_), \
                               static_cast<const T*>(e_), static_cast<const T*>(ws_), static_cast<const T*>(wo_), \
                               static_cast<const T*>(add_e_), static_cast<T*>(dq_), static_cast<T*>(dk_),       \
                               static_cast<T*>(dv_), static_cast<T*>(de_), N, C, alpha, g.slices, B, reverse);           \
        } else {                                                                                                \
            DG_OPT_IN_LDS((&attn_bwd_kernel<T, LQS, JPL, RW_>), lds);                                            \
            hipLaunchKernelGGL((attn_bwd_kernel<T, LQS, JPL, RW_>), grid, block, lds, stream,                    \
                               static_cast<const T*>(q_), static_cast<const T*>(k_), static_cast<const T*>(v_), \
                               static_cast<const T*>(e_), static_cast<const T*>(ws_), static_cast<const T*>(wo_), \
                               static_cast<const T*>(nullptr), static_cast<T*>(dq_), static_cast<T*>(dk_),      \
                               static_cast<T*>(dv_), static_cast<T*>(de_), N, C, alpha, g.slices, B, reverse);           \
        }                                                                                                       \
    }
#define LAUNCH_RW(LQS, JPL, RW_)                                        \
    {                                                                   \
        if (dtype == DG_DTYPE_BF16) LAUNCH_T(bf16_t, LQS, JPL, RW_)     \
        else LAUNCH_T(float, LQS, JPL, RW_)                             \
    }
#define LAUNCH(LQS, JPL)                                  \
    if (g.lqs == LQS && g.jpl == JPL) {                   \
        if (rw8 && JPL <= 6) LAUNCH_RW(LQS, (JPL <= 6 ? JPL : 1), 8) else LAUNCH_RW(LQS, JPL, kRW) \
    }
    DG_FOR_GEOMETRY(LAUNCH)
#undef LAUNCH
#undef LAUNCH_RW
#undef LAUNCH_T
    return check_launch("dg_attn_core_bwd");
}

extern "C" int dg_attn_core_bwd2(const void* q_, const void* k_, const void* v_, const void* e_, const void* ws_,
                                 const void* wo_, const void* tq_, const void* tk_, const void* tv_, const void* te_,
                                 void* gq_, void* gk_, void* gv_, void* ge_, void* gws_, void* gwo_, int B, int N,
                                 int C, float alpha, int dtype, dg_stream_t stream_) {
    if (!q_ || !k_ || !v_ || !e_ || !wo_ || !tq_ || !tk_ || !tv_ || !te_ || !gq_ || !gk_ || !gv_ || !ge_ || !gwo_)  // ws, gws may be NULL
        return fail(DG_E_ARG, "dg_attn_core_bwd2: null pointer");
    if (!dtype_ok(dtype)) return fail(DG_E_ARG, "dg_attn_core_bwd2: unknown dtype %d", dtype);
    Geometry g;
    if (B < 0 || !pick_geometry(N, C, &g, true))
        return fail(DG_E_SHAPE, "dg_attn_core_bwd2: unsupported shape B=%d N=%d C=%d", B, N, C);
    if (B == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    dim3 grid(static_cast<unsigned>((B + 7) / 8 * 8) * g.slices), block(64 * kRW);
    ProfScope prof(DG_K_ATTN_BWD2, stream);
    const int reverse = take_direction(static_cast<int64_t>(B) * N * N);      // per-molecule results: any order
#define LAUNCH_T(T, LQS, JPL)                                                                                     \
    {                                                                                                             \
        constexpr int lds = (4 * JPL + (kRW - 1) * 2 * JPL) * 64 * 16;                                            \
        DG_OPT_IN_LDS((&attn_bwd2_kernel<T, LQS, JPL, kRW>), lds);                                                 \
        hipLaunchKernelGGL((attn_bwd2_kernel<T, LQS, JPL, kRW>), grid, block, lds, stream,                         \
                           static_cast<const T*>(q_), static_cast<const T*>(k_), static_cast<const T*>(v_),       \
                           static_cast<const T*>(e_), static_cast<const T*>(ws_), static_cast<const T*>(wo_),     \
                           static_cast<const T*>(tq_), static_cast<const T*>(tk_), static_cast<const T*>(tv_),    \
                           static_cast<const T*>(te_), static_cast<T*>(gq_), static_cast<T*>(gk_),                \
                           static_cast<T*>(gv_), static_cast<T*>(ge_), static_cast<T*>(gws_), static_cast<T*>(gwo_), \
                           N, C, alpha, g.slices, B, reverse);                                                             \
    }
#define LAUNCH(LQS, JPL)                                   \
    if (g.lqs == LQS && g.jpl == JPL) {                    \
        if (dtype == DG_DTYPE_BF16) LAUNCH_T(bf16_t, LQS, JPL) \
        else LAUNCH_T(float, LQS, JPL)                     \
    }
    DG_FOR_GEOMETRY(LAUNCH)
#undef LAUNCH
#undef LAUNCH_T
    return check_launch("dg_attn_core_bwd2");
}
