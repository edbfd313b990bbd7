// bf16 storage helpers shared by the dtype-templated kernels (BASELINE configs[2]: activations in
// bf16 in HBM, fp32 arithmetic in registers, fp32 MFMA accumulation, fp32 softmax / LayerNorm
// statistics).  `act_t<DT>` maps the ABI's dtype code to the element type; ld4 / st4 overloads move
// four consecutive channels (16 B of fp32, 8 B of bf16) between memory and a float4.
#pragma once

#include "common.h"

namespace dg {

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

template <int DT> struct act_type;
template <> struct act_type<DG_DTYPE_F32> { typedef float type; };
template <> struct act_type<DG_DTYPE_BF16> { typedef bf16_t type; };

__device__ __forceinline__ float bf16_bits_to_float(unsigned short b) { return __uint_as_float(static_cast<unsigned>(b) << 16); }
__device__ __forceinline__ float lo_bf16(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf16(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
// two floats -> packed bf16 pair (round to nearest even), low half = a
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = {static_cast<__bf16>(a), static_cast<__bf16>(b)};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ u32x2_t pack4_bf16(float4 v) {
    u32x2_t o;
    o[0] = pack_bf16(v.x, v.y);
    o[1] = pack_bf16(v.z, v.w);
    return o;
}
__device__ __forceinline__ float4 unpack4_bf16(u32x2_t w) {
    return make_float4(lo_bf16(w[0]), hi_bf16(w[0]), lo_bf16(w[1]), hi_bf16(w[1]));
}

// ---- four channels <-> float4 ---------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const bf16_t* p) { return unpack4_bf16(*reinterpret_cast<const u32x2_t*>(p)); }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) { *reinterpret_cast<u32x2_t*>(p) = pack4_bf16(v); }
__device__ __forceinline__ float4 ld4_stream(const bf16_t* p) {
    return unpack4_bf16(__builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p)));
}
__device__ __forceinline__ void st4_stream(bf16_t* p, float4 v) {
    __builtin_nontemporal_store(pack4_bf16(v), reinterpret_cast<u32x2_t*>(p));
}
// raw (unconverted) four-channel loads: a bf16 row slot costs 2 registers until it is unpacked, which makes a
// one-row-ahead register prefetch affordable in the bf16 kernels
template <typename T> struct raw4;
template <> struct raw4<float> { typedef float4 type; };
template <> struct raw4<bf16_t> { typedef u32x2_t type; };
__device__ __forceinline__ float4 ld_raw_stream(const float* p) { return ld4_stream(p); }
__device__ __forceinline__ u32x2_t ld_raw_stream(const bf16_t* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p)); }
__device__ __forceinline__ float4 ld_raw(const float* p) { return ld4(p); }
__device__ __forceinline__ u32x2_t ld_raw(const bf16_t* p) { return *reinterpret_cast<const u32x2_t*>(p); }
__device__ __forceinline__ float4 cvt_raw(float4 v) { return v; }
__device__ __forceinline__ float4 cvt_raw(u32x2_t v) { return unpack4_bf16(v); }

// one channel
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return static_cast<float>(*p); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = static_cast<__bf16>(v); }

inline bool dtype_ok(int dtype) { return dtype == DG_DTYPE_F32 || dtype == DG_DTYPE_BF16; }
// DG_DTYPE_F32_H16 is float32 everywhere except the 384-wide hidden operands
inline int act_dtype(int dtype) { return (dtype >= DG_DTYPE_F32_H16 && dtype <= DG_DTYPE_F32_H32_DH16) ? DG_DTYPE_F32 : dtype; }
// dg_edge_ffn_ln_bwd: the dtype its launches over `h` / over `dh` take (DG_DTYPE_F32_DH16 / _DH24: h float32, dh narrow)
inline int ffn_h_dtype(int dtype) {
    if (dtype == DG_DTYPE_F32_H32_DH16) return DG_DTYPE_F32_H32;
    return (dtype == DG_DTYPE_F32_DH16 || dtype == DG_DTYPE_F32_DH24) ? DG_DTYPE_F32 : dtype;
}
inline int ffn_dh_dtype(int dtype) {
    if (dtype == DG_DTYPE_F32_DH16 || dtype == DG_DTYPE_F32_H32_DH16) return DG_DTYPE_F32_H16;
    return dtype == DG_DTYPE_F32_DH24 ? DG_DTYPE_F32_H24 : dtype;
}
// storage of a 384-wide hidden operand inside the producer / consumer kernels: 0 float32, 1 fp16 plane + row scales, 2 three-byte elements
inline int hidden_fmt(int dtype) { return dtype == DG_DTYPE_F32_H16 ? 1 : (dtype == DG_DTYPE_F32_H24 ? 2 : (dtype == DG_DTYPE_F32_H32 ? 3 : 0)); }
inline size_t hidden_scale_offset(int64_t R, int H) { return (static_cast<size_t>(R) * H * 2 + 255) / 256 * 256; }
inline size_t dtype_size(int dtype) { return dtype == DG_DTYPE_BF16 ? 2 : 4; }

}  // namespace dg
