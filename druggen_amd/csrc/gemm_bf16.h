// Device-side building blocks of the bf16 row kernels (gemm_bf16.hip, ffn_bf16.hip).
//
// Activation tiles: 64 rows x K channels of bf16, row-major with a 2K-byte pitch; inside every
// 256-byte segment of a row the 16-byte chunk index is XOR-ed with (row & 15).  Tiles arrive by
// LDS-DMA (destination lane-linear, the permutation is applied to the per-lane SOURCE address) or
// are written by an MFMA epilogue with the same permutation.  A ds_read_b128 of one MFMA operand
// fragment -- lane = (row, 8-channel group) -- is then bank-conflict free for both bf16 MFMA shapes.
//
// Weights: fp32 nn.Linear parameters are converted once per weight version into bf16 MFMA
// *fragment order*; a wave keeps its fragments in VGPRs for the whole (persistent) kernel.
//   P32[(mb * KS + ks) * 64 + lane] = { W'[32 mb + (lane & 31)][16 ks + 8 (lane >> 5) + j] } j < 8   (32x32x16)
//   P16[(mb * KS + ks) * 64 + lane] = { W'[16 mb + (lane & 15)][32 ks + 8 (lane >> 4) + j] } j < 8   (16x16x32)
// with y = x . W'^T: W' = W (mode 0, forward) or W^T (mode 1, input gradient).  The same fragment
// serves as the A operand ("swapped" product W' . x^T: a lane ends up with 4 consecutive output
// channels of one row) or as the B operand (x . W'^T: 4 consecutive rows of one output channel).
#pragma once

#include "bf16.h"

namespace dg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRowsPerTile = 64;

// byte offset of channel `ch` (multiple of 8 for a 16-byte chunk) of tile row `row`, K channels per row
__device__ __forceinline__ unsigned tile_off(int row, int ch, int K) {
    const int chunk = ch >> 3;
    const int sw = (chunk & ~15) | ((chunk & 15) ^ (row & 15));
    return static_cast<unsigned>(row * (K * 2) + sw * 16 + (ch & 7) * 2);
}

// Start the LDS-DMA of one 64 x K bf16 tile (rows r0.. of a [R, K] matrix) into `lds_dst`; all
// `WAVES` waves of the workgroup take part.  Rows >= R are not fetched (the tile keeps stale bytes
// there; nothing computed from them is ever stored).  The caller waits (vmcnt) and synchronises.
template <int K, int WAVES>
__device__ __forceinline__ void dma_tile_bf16(const bf16_t* __restrict__ a, int64_t r0, int64_t R, char* lds_dst,
                                              int wave, int lane) {
    constexpr int CPR = K / 8;                       // 16-byte chunks per row
    constexpr int INSTR = kRowsPerTile * CPR / 64;   // wave-instructions per tile
    const unsigned dst = lds_byte_address(lds_dst);
    static_assert(INSTR % WAVES == 0, "tile must split evenly over the waves");
#pragma unroll
    for (int t = 0; t < INSTR / WAVES; ++t) {
        const int ii = wave + t * WAVES;
        const int L = ii * 64 + lane;
        const int row = L / CPR, cpos = L % CPR;
        const int src = (cpos & ~15) | ((cpos & 15) ^ (row & 15));
        if (r0 + row < R)
            dma16_async(reinterpret_cast<const float*>(a + (r0 + row) * K + src * 8), dst + ii * 1024);
    }
}

// ---- sum over the 32 lanes of a half-wave (DPP inside 16-lane rows + two scalar reads) -----------
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true);
    return x + __int_as_float(moved);
}
__device__ __forceinline__ float half_wave_sum(float x) {
    x = dpp_add<0xB1>(x);    // quad_perm [1,0,3,2]
    x = dpp_add<0x4E>(x);    // quad_perm [2,3,0,1]
    x = dpp_add<0x141>(x);   // row_half_mirror
    x = dpp_add<0x140>(x);   // row_mirror: every lane holds its 16-lane row total
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
    return (threadIdx.x & 32) ? r2 + r3 : r0 + r1;
}

// ---- fp32 exchange tile [64][N]: accumulators (a lane owns 4 consecutive channels of a row) go in
// as 16-byte slots XOR-ed with (row & 7) (conflict-free ds_write_b128 / ds_read_b128), whole rows
// come out: lane `col` of a half-wave reads channels [4 col, 4 col + 4) of each 128-channel chunk.
__device__ __forceinline__ unsigned xch_off(int row, int ch, int N) {
    const int slot = ch >> 2;
    return static_cast<unsigned>(row * (N * 4) + (((slot & ~7) | ((slot & 7) ^ (row & 7))) << 4));
}

}  // namespace dg
