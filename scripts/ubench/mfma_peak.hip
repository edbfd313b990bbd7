// Sustained v_mfma_f32_32x32x2_f32 rate on this chip (clock under load included): what the
// "MFMA roofline" of the fp32 row GEMMs can actually reach.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ACC>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters) {
    f32x16 acc[ACC];
    for (int m = 0; m < ACC; ++m)
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = in[(threadIdx.x + 64 * i) & 4095];
        b[i] = in[(threadIdx.x * 3 + 17 * i) & 4095];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int m = 0; m < ACC; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + m) & 7], acc[m], 0, 0, 0);
    }
    float s = 0.f;
    for (int m = 0; m < ACC; ++m)
        for (int i = 0; i < 16; ++i) s += acc[m][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *in, *out;
    hipMalloc(&in, 4096 * 4);
    hipMalloc(&out, 256 * 16 * 256 * 4);
    float h[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * (getenv("SCALE") ? atof(getenv("SCALE")) : 1e-3f);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    for (int blocks_per_cu = 1; blocks_per_cu <= 4; blocks_per_cu *= 2) {
        for (int accn = 1; accn <= 4; accn *= 2) {
            auto launch = [&]() {
                dim3 g(256 * blocks_per_cu), b(256);
                if (accn == 1) hipLaunchKernelGGL(mfma_loop<1>, g, b, 0, 0, in, out, iters);
                if (accn == 2) hipLaunchKernelGGL(mfma_loop<2>, g, b, 0, 0, in, out, iters);
                if (accn == 4) hipLaunchKernelGGL(mfma_loop<4>, g, b, 0, 0, in, out, iters);
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double flops = 5.0 * 256 * blocks_per_cu * 4 * (double)iters * 8 * accn * 4096.0;
            printf("waves/SIMD %d  independent accumulators %d : %.1f TFLOP/s  (%.2f ms per launch)\n", blocks_per_cu, accn,
                   flops / (ms * 1e-3) / 1e12, ms / 5);
        }
    }
    return 0;
}
