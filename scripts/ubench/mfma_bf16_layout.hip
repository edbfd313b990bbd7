// Probe: operand / result layout of v_mfma_f32_32x32x16_bf16 on gfx950, and the accuracy of a
// 3-way bf16 split of fp32 operands (6 cross products, i + j <= 4) against fp64.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float a, __bf16& h, __bf16& m, __bf16& l) {
    h = static_cast<__bf16>(a);
    const float r1 = a - static_cast<float>(h);
    m = static_cast<__bf16>(r1);
    const float r2 = r1 - static_cast<float>(m);
    l = static_cast<__bf16>(r2);
}

// C[32][32] = A[32][K] * B[K][32], K = 128, one wave.  MODE 0: plain bf16 (hi only), 1: 6-term split
template <int MODE>
__global__ void k(const float* A, const float* B, float* C) {
    const int l = threadIdx.x, row = l & 31, g = l >> 5;
    f32x16 acc = {};
    for (int ks = 0; ks < 8; ++ks) {
        bf16x8 a[3], b[3];
        for (int j = 0; j < 8; ++j) {
            const int kk = ks * 16 + 8 * g + j;
            __bf16 h, m, lo;
            split3(A[row * 128 + kk], h, m, lo);
            a[0][j] = h; a[1][j] = m; a[2][j] = lo;
            split3(B[kk * 32 + row], h, m, lo);
            b[0][j] = h; b[1][j] = m; b[2][j] = lo;
        }
        if (MODE == 1) {   // small terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
    for (int reg = 0; reg < 16; ++reg) C[((reg & 3) + 8 * (reg >> 2) + 4 * g) * 32 + row] = acc[reg];
}

__global__ void kf32(const float* A, const float* B, float* C) {   // exact fp32 MFMA for comparison
    const int l = threadIdx.x, row = l & 31, g = l >> 5;
    f32x16 acc = {};
    for (int ks = 0; ks < 64; ++ks)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[row * 128 + 2 * ks + g], B[(2 * ks + g) * 32 + row], acc, 0, 0, 0);
    for (int reg = 0; reg < 16; ++reg) C[((reg & 3) + 8 * (reg >> 2) + 4 * g) * 32 + row] = acc[reg];
}

int main() {
    float hA[32 * 128], hB[128 * 32], hC[1024];
    double ref[1024], scale[1024];
    srand(3);
    for (auto& v : hA) v = rand() / (float)RAND_MAX - 0.5f;
    for (auto& v : hB) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double s = 0, sa = 0;
            for (int kk = 0; kk < 128; ++kk) {
                s += (double)hA[i * 128 + kk] * hB[kk * 32 + j];
                sa += fabs((double)hA[i * 128 + kk] * hB[kk * 32 + j]);
            }
            ref[i * 32 + j] = s;
            scale[i * 32 + j] = sa;
        }
    float *A, *B, *C;
    hipMalloc(&A, sizeof(hA)); hipMalloc(&B, sizeof(hB)); hipMalloc(&C, sizeof(hC));
    hipMemcpy(A, hA, sizeof(hA), hipMemcpyHostToDevice);
    hipMemcpy(B, hB, sizeof(hB), hipMemcpyHostToDevice);
    auto report = [&](const char* what) {
        hipMemcpy(hC, C, sizeof(hC), hipMemcpyDeviceToHost);
        double worst = 0, rms = 0;
        for (int i = 0; i < 1024; ++i) {
            const double e = fabs(hC[i] - ref[i]) / scale[i];   // error relative to sum |a b|
            worst = fmax(worst, e);
            rms += e * e;
        }
        printf("%-40s max err / sum|ab| = %.3e   rms = %.3e\n", what, worst, sqrt(rms / 1024));
    };
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, A, B, C); report("bf16 (hi parts only)");
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, A, B, C); report("bf16 x6 split");
    hipLaunchKernelGGL(kf32, dim3(1), dim3(64), 0, 0, A, B, C); report("fp32 MFMA 32x32x2");
    // fp32 sequential CPU sum for scale
    double worst = 0, rms = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float s = 0.f;
            for (int kk = 0; kk < 128; ++kk) s = fmaf(hA[i * 128 + kk], hB[kk * 32 + j], s);
            const double e = fabs(s - ref[i * 32 + j]) / scale[i * 32 + j];
            worst = fmax(worst, e); rms += e * e;
        }
    printf("%-40s max err / sum|ab| = %.3e   rms = %.3e\n", "CPU fp32 fmaf chain", worst, sqrt(rms / 1024));
    return 0;
}
