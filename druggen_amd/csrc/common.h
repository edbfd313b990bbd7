// Shared helpers of libdruggen_hip.so (gfx950 only, wave = 64 lanes).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/druggen_hip.h"

namespace dg {

// ---- thread-local error text ------------------------------------------------
char* error_buffer();
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);

// ---- dynamic LDS opt-in -------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: it has to be set
// on every device the kernel is launched on (nn.DataParallel drives several GPUs from one process),
// and a failure must surface.  One bit per device ordinal in `done`; devices >= 64 are always re-set.
int opt_in_dynamic_lds(std::atomic<unsigned long long>* done, const void* kernel, int bytes);
#define DG_OPT_IN_LDS(kernel_ptr, bytes)                                                                         \
    do {                                                                                                         \
        static std::atomic<unsigned long long> dg_lds_done_{0};                                                  \
        if (int dg_lds_st_ = dg::opt_in_dynamic_lds(&dg_lds_done_, reinterpret_cast<const void*>(kernel_ptr), (bytes))) \
            return dg_lds_st_;                                                                                   \
    } while (0)

// ---- profiler (prof.hip) ----------------------------------------------------
struct ProfScope {
    int id;
    hipStream_t stream;
    void* slot;
    ProfScope(int kernel_id, hipStream_t s);
    ~ProfScope();
};

// ---- DG_DTYPE_F32_H24: the top 24 bits of a float32 (1 + 8 + 15: 16 significant bits, round half up), 3 bytes per element ----
// four consecutive elements = 12 bytes = three dwords, little endian
typedef unsigned dg_u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ dg_u32x3 pack_f24x4(float a, float b, float c, float d) {
    const unsigned ua = __float_as_uint(a) + 0x80u, ub = __float_as_uint(b) + 0x80u, uc = __float_as_uint(c) + 0x80u,
                   ud = __float_as_uint(d) + 0x80u;
    // v_perm_b32: selector bytes 0..3 pick from the second operand, 4..7 from the first
    return dg_u32x3{__builtin_amdgcn_perm(ub, ua, 0x05030201u), __builtin_amdgcn_perm(uc, ub, 0x06050302u),
                    __builtin_amdgcn_perm(ud, uc, 0x07060503u)};
}
__device__ __forceinline__ float4 unpack_f24x4(dg_u32x3 w) {
    return make_float4(__uint_as_float(w[0] << 8), __uint_as_float(__builtin_amdgcn_perm(w[1], w[0], 0x0504030Cu)),
                       __uint_as_float(__builtin_amdgcn_perm(w[2], w[1], 0x0403020Cu)), __uint_as_float(w[2] & 0xFFFFFF00u));
}

// ---- edge-level threshold (dg_set_edge_rows; runtime.hip) ---------------------
int64_t edge_rows();

// ---- float4 arithmetic ------------------------------------------------------
__device__ __forceinline__ float4 f4(float x) { return make_float4(x, x, x, x); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 operator*(float a, float4 b) { return make_float4(a * b.x, a * b.y, a * b.z, a * b.w); }
__device__ __forceinline__ float4& operator+=(float4& a, float4 b) { a = a + b; return a; }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 max4(float4 a, float4 b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
__device__ __forceinline__ float4 exp4(float4 a) { return make_float4(__expf(a.x), __expf(a.y), __expf(a.z), __expf(a.w)); }
__device__ __forceinline__ float4 rcp4(float4 a) { return make_float4(1.0f / a.x, 1.0f / a.y, 1.0f / a.z, 1.0f / a.w); }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// streamed once: keep it out of the way of the small reused operands
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_stream(const float* p) {
    const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void st4_stream(float* p, float4 v) {
    v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}

// ---- async global -> LDS copy (LDS-DMA), 16 bytes per lane ----------------------------------
// LDS destination = `lds_base` (wave-uniform byte address) + lane*16; global source is per lane.
// Issued through inline asm on purpose: hipcc orders a builtin LDS-DMA before every later
// ds_read with `s_waitcnt vmcnt(0)`, which would serialise the copy of tile t+1 with the compute
// on tile t.  The caller owns the wait: `s_waitcnt vmcnt(0)` + barrier before reading the tile.
__device__ __forceinline__ unsigned lds_byte_address(const void* p) {
    return static_cast<unsigned>(reinterpret_cast<size_t>((const __attribute__((address_space(3))) void*)p));
}
__device__ __forceinline__ void dma16_async(const float* gsrc, unsigned lds_base) {
    unsigned keep;
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_base);
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(base)
        : "memory");
}
__device__ __forceinline__ void wait_all_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Same wait through the builtin, so that hipcc's own scoreboard also learns that its earlier
// loads have returned (0x0F70 = vmcnt(0), expcnt/lgkmcnt untouched on gfx9 encodings).
__device__ __forceinline__ void wait_all_vmem_visible() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// butterfly over the lanes that differ in bits >= LOW of the lane id.  The steps over lane bits 3, 4 and 5 stay in
// the VALU: xor 8 is a DPP row rotation by 8, xor 16 / xor 32 are v_permlane16_swap / v_permlane32_swap of the
// value with itself (gfx950) -- after the swap one result register holds "my half", the other "the partner half" in
// every lane, so the combine needs no select.  A __shfl_xor is a ds_bpermute round trip through the LDS crossbar
// per step; three dependent ones per reduction dominated the per-row latency of the attention kernels.
template <bool MAX>
__device__ __forceinline__ float xor_step(float x, int m) {
    float y;
    if (m == 8) {
        y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128 /* row_ror:8 */, 0xF, 0xF, true));
    } else if (m == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        x = __uint_as_float(r[0]);
        y = __uint_as_float(r[1]);
    } else if (m == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        x = __uint_as_float(r[0]);
        y = __uint_as_float(r[1]);
    } else {
        y = __shfl_xor(x, m, 64);
    }
    return MAX ? fmaxf(x, y) : x + y;
}
template <int LOW>
__device__ __forceinline__ float xor_sum(float x) {
#pragma unroll
    for (int m = LOW; m < 64; m <<= 1) x = xor_step<false>(x, m);
    return x;
}
template <int LOW>
__device__ __forceinline__ float xor_max(float x) {
#pragma unroll
    for (int m = LOW; m < 64; m <<= 1) x = xor_step<true>(x, m);
    return x;
}
template <int LOW>
__device__ __forceinline__ float4 xor_sum4(float4 a) {
    return make_float4(xor_sum<LOW>(a.x), xor_sum<LOW>(a.y), xor_sum<LOW>(a.z), xor_sum<LOW>(a.w));
}
template <int LOW>
__device__ __forceinline__ float4 xor_max4(float4 a) {
    return make_float4(xor_max<LOW>(a.x), xor_max<LOW>(a.y), xor_max<LOW>(a.z), xor_max<LOW>(a.w));
}

}  // namespace dg
