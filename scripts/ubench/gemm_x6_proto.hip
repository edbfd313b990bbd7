// Prototype: 128x128 row GEMM (y = a W^T + bias) with fp32 operands split 3-way into bf16 and
// 6 v_mfma_f32_32x32x16_bf16 products per k-step (fp32-class accuracy, see mfma_bf16_layout.hip).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int TR = 64, K = 128, N = 128, PITCH = 272;   // bytes per LDS row of one bf16 plane
constexpr int PLANE = TR * PITCH;

__device__ __forceinline__ void split4(float4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = static_cast<__bf16>(x[i]);
        const float r1 = x[i] - static_cast<float>(h[i]);
        m[i] = static_cast<__bf16>(r1);
        l[i] = static_cast<__bf16>(r1 - static_cast<float>(m[i]));
    }
}

// packed: [slab 4][plane 3][ks 8][lane 64] x bf16x8
__global__ __launch_bounds__(256, 2) void gemm_x6(const float* __restrict__ a, const bf16x8* __restrict__ packed,
                                                  const float* __restrict__ bias, float* __restrict__ y, long R) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, row = lane & 31, g = lane >> 5;
    bf16x8 bfr[3][8];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bfr[p][ks] = packed[((w * 3 + p) * 8 + ks) * 64 + lane];
    const float bv = bias[32 * w + row];
    const long tiles = R / TR;
    float4 pf[8];
    long tix = blockIdx.x;
    auto fetch = [&](long t) {
        const float* src = a + t * TR * K;
#pragma unroll
        for (int i = 0; i < 8; ++i) pf[i] = *reinterpret_cast<const float4*>(src + (threadIdx.x + 256 * i) * 4);
    };
    if (tix < tiles) fetch(tix);
    for (; tix < tiles; tix += gridDim.x) {
        __syncthreads();   // previous tile's fragment reads are done
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int L = threadIdx.x + 256 * i, r = L >> 5, c4 = L & 31;
            bf16x4 h, m, l;
            split4(pf[i], h, m, l);
            *reinterpret_cast<bf16x4*>(lds + 0 * PLANE + r * PITCH + c4 * 8) = h;
            *reinterpret_cast<bf16x4*>(lds + 1 * PLANE + r * PITCH + c4 * 8) = m;
            *reinterpret_cast<bf16x4*>(lds + 2 * PLANE + r * PITCH + c4 * 8) = l;
        }
        if (tix + gridDim.x < tiles) fetch(tix + gridDim.x);
        __syncthreads();
        f32x16 acc[2] = {};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                bf16x8 af[3];
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    af[p] = *reinterpret_cast<const bf16x8*>(lds + p * PLANE + (32 * m + row) * PITCH + (ks * 16 + 8 * g) * 2);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bfr[0][ks], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[1][ks], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[2][ks], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[0][ks], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[1][ks], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[0][ks], acc[m], 0, 0, 0);
            }
        }
        float* out = y + tix * TR * N + 32 * w + row;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) out[(32 * m + (reg & 3) + 8 * (reg >> 2) + 4 * g) * N] = acc[m][reg] + bv;
    }
}

int main() {
    const long R = 256L * 45 * 45;
    std::vector<float> ha(R * K), hw(N * K), hb(N);
    srand(5);
    for (auto& v : ha) v = rand() / (float)RAND_MAX - 0.5f;
    for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    for (auto& v : hb) v = rand() / (float)RAND_MAX - 0.5f;
    // pack W[n][k] -> B fragments of the 3 bf16 planes
    std::vector<__bf16> hp(4 * 3 * 8 * 64 * 8);
    for (int t = 0; t < 4; ++t)
        for (int ks = 0; ks < 8; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const float v = hw[(32 * t + (l & 31)) * K + ks * 16 + 8 * (l >> 5) + j];
                    const __bf16 h = static_cast<__bf16>(v);
                    const float r1 = v - static_cast<float>(h);
                    const __bf16 m = static_cast<__bf16>(r1);
                    const __bf16 lo = static_cast<__bf16>(r1 - static_cast<float>(m));
                    const __bf16 pl[3] = {h, m, lo};
                    for (int p = 0; p < 3; ++p) hp[((((t * 3 + p) * 8 + ks) * 64 + l) * 8) + j] = pl[p];
                }
    float *a, *y, *b;
    bf16x8* packed;
    hipMalloc(&a, R * K * 4); hipMalloc(&y, R * N * 4); hipMalloc(&b, N * 4); hipMalloc(&packed, hp.size() * 2);
    hipMemcpy(a, ha.data(), R * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(packed, hp.data(), hp.size() * 2, hipMemcpyHostToDevice);
    const int lds_bytes = 3 * PLANE;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(gemm_x6, dim3(512), dim3(256), lds_bytes, 0, a, packed, b, y, R);
    hipDeviceSynchronize();
    std::vector<float> hy(64 * N);
    const long probe_row = 64L * 4321;
    hipMemcpy(hy.data(), y + probe_row * N, hy.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int r = 0; r < 64; ++r)
        for (int n = 0; n < N; ++n) {
            double s = hb[n], sa = fabs(hb[n]);
            for (int kk = 0; kk < K; ++kk) {
                s += (double)ha[(probe_row + r) * K + kk] * hw[n * K + kk];
                sa += fabs((double)ha[(probe_row + r) * K + kk] * hw[n * K + kk]);
            }
            worst = fmax(worst, fabs(hy[r * N + n] - s) / sa);
        }
    printf("max err / sum|ab| on a probe tile: %.3e\n", worst);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(gemm_x6, dim3(512), dim3(256), lds_bytes, 0, a, packed, b, y, R);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("128x128 row GEMM, R=%ld: %.1f us  (%.1f TFLOP/s fp32-equivalent, %.2f TB/s)\n", R, ms * 1e3 / 20,
           2.0 * R * K * N / (ms / 20 * 1e-3) / 1e12, 2.0 * R * K * 4 / (ms / 20 * 1e-3) / 1e12);
    return 0;
}
