// Hardware check of the two idioms the fused attention-half backward relies on (gfx950):
//   (1) the accumulator layout of a 16x16 MFMA block (row = 4 (lane >> 4) + r, col = lane & 15), converted to bf16,
//       IS the A operand of v_mfma_f32_16x16x16_bf16 for the TRANSPOSED block (m = col, k = row): X^T Y without an
//       LDS transposition;
//   (2) ds_read_b64_tr_b16 delivers the B operand B[k = 4 (lane >> 4) + t][n = lane & 15] from a ROW-MAJOR [rows][128]
//       bf16 tile (256-byte pitch, 16-byte chunks XOR-swizzled with row & 15) when lane i of a 16-lane group points at
//       (row 4 g + i / 4, channel 16 nb + 4 (i % 4)).
// Build & run:  hipcc --offload-arch=gfx950 -O2 scripts/ubench/tr_wgrad_probe.hip -o scripts/ubench/tr_wgrad_probe && ./...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned tile_off(int row, int ch) { return row * 256 + ((((ch >> 3) ^ (row & 15))) << 4) + (ch & 7) * 2; }

// X [16][16] fp32 (rows j, cols c) in accumulator layout; Y [16][128] bf16 row-major swizzled in LDS.
// out[c][n] = sum_j X[j][c] Y[j][n], n < 128
__global__ void probe(const float* __restrict__ X, const __bf16* __restrict__ Y, float* __restrict__ out, float* __restrict__ trdump) {
    __shared__ __attribute__((aligned(16))) char tile[16 * 256];
    const int lane = threadIdx.x, r16 = lane & 15, g = lane >> 4;
    for (int idx = lane; idx < 16 * 128; idx += 64) {
        const int row = idx / 128, ch = idx % 128;
        *reinterpret_cast<__bf16*>(tile + tile_off(row, ch)) = Y[idx];
    }
    __syncthreads();
    // accumulator layout of X: lane holds X[4 g + r][r16]
    bf16x4 a;
    for (int r = 0; r < 4; ++r) a[r] = static_cast<__bf16>(X[(4 * g + r) * 16 + r16]);
    const unsigned base = static_cast<unsigned>(reinterpret_cast<size_t>((const __attribute__((address_space(3))) void*)tile));
    for (int nb = 0; nb < 8; ++nb) {
        const int row = 4 * g + (r16 >> 2), ch = 16 * nb + 4 * (r16 & 3);
        const unsigned addr = base + tile_off(row, ch);
        u32x2 bv;
        asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(bv) : "v"(addr) : "memory");
        const bf16x4 b = __builtin_bit_cast(bf16x4, bv);
        for (int t = 0; t < 4; ++t) trdump[(nb * 64 + lane) * 4 + t] = static_cast<float>(b[t]);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), acc, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out[(4 * g + r) * 128 + 16 * nb + r16] = acc[r];   // D[m = 4 g + r][n = r16]
    }
}

int main() {
    std::vector<float> X(256), out(16 * 128), tr(8 * 64 * 4);
    std::vector<__bf16> Y(16 * 128);
    srand(1);
    for (auto& x : X) x = static_cast<float>(rand() % 17 - 8);
    for (auto& y : Y) y = static_cast<__bf16>(static_cast<float>(rand() % 15 - 7));
    float *dX, *dO, *dT; __bf16* dY;
    hipMalloc(&dX, 1024); hipMalloc(&dY, 16 * 128 * 2); hipMalloc(&dO, 16 * 128 * 4); hipMalloc(&dT, tr.size() * 4);
    hipMemcpy(dX, X.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(dY, Y.data(), 16 * 128 * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dX, dY, dO, dT);
    hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(tr.data(), dT, tr.size() * 4, hipMemcpyDeviceToHost);
    int bad_tr = 0, bad = 0;
    for (int nb = 0; nb < 8; ++nb)
        for (int lane = 0; lane < 64; ++lane)
            for (int t = 0; t < 4; ++t) {
                const float want = static_cast<float>(Y[(4 * (lane >> 4) + t) * 128 + 16 * nb + (lane & 15)]);
                if (tr[(nb * 64 + lane) * 4 + t] != want) ++bad_tr;
            }
    for (int c = 0; c < 16; ++c)
        for (int n = 0; n < 128; ++n) {
            float want = 0.f;
            for (int j = 0; j < 16; ++j) want += X[j * 16 + c] * static_cast<float>(Y[j * 128 + n]);
            if (out[c * 128 + n] != want) ++bad;
        }
    printf("tr_read mismatches: %d of %d; X^T Y mismatches: %d of %d\n", bad_tr, 8 * 64 * 4, bad, 16 * 128);
    return (bad_tr || bad) ? 1 : 0;
}
