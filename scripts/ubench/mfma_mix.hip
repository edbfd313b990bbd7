// What costs MFMA rate in a row-GEMM-like wave program?  Adds one ingredient at a time to a bare
// 2-accumulator v_mfma_f32_32x32x2_f32 stream (2 waves / SIMD, 4 waves / block like the real kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ in, float* __restrict__ out, int tiles) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 128];
    for (int i = threadIdx.x; i < 64 * 128; i += 256) lds[i] = in[i & 4095];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 bf[16];
    for (int q = 0; q < 16; ++q) bf[q] = *reinterpret_cast<const f32x4*>(in + ((q * 64 + lane) * 4 & 4095));
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) c0[i] = c1[i] = 0.f;
    const unsigned base = static_cast<unsigned>(reinterpret_cast<size_t>((const __attribute__((address_space(3))) void*)lds));
    f32x4 fr[2][2];
    fr[0][0] = *reinterpret_cast<f32x4*>(lds + (lane & 31) * 128);
    fr[0][1] = *reinterpret_cast<f32x4*>(lds + (32 + (lane & 31)) * 128);
    fr[1][0] = fr[0][0];
    fr[1][1] = fr[0][1];
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int cb = st & 1;
            const f32x4 b = bf[st];
            if (MODE >= 1) {
                const unsigned a0 = base + 4 * ((lane & 31) * 128 + ((((lane >> 5) * 16 + st) ^ (lane & 15)) << 2));
                const unsigned a1 = a0 + 32 * 512;
                asm volatile("ds_read_b128 %2, %4\n\tds_read_b128 %3, %5\n\t"
                             "v_mfma_f32_32x32x2_f32 %0, %6, %8, %0\n\tv_mfma_f32_32x32x2_f32 %1, %7, %8, %1"
                             : "+a"(c0), "+a"(c1), "=&v"(fr[cb ^ 1][0]), "=&v"(fr[cb ^ 1][1])
                             : "v"(a0), "v"(a1), "v"(fr[cb][0].x), "v"(fr[cb][1].x), "v"(b.x));
            } else {
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %4, %0\n\tv_mfma_f32_32x32x2_f32 %1, %3, %4, %1"
                             : "+a"(c0), "+a"(c1) : "v"(fr[cb][0].x), "v"(fr[cb][1].x), "v"(b.x));
            }
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %6, %0\n\tv_mfma_f32_32x32x2_f32 %1, %5, %6, %1"
                         : "+a"(c0), "+a"(c1), "+v"(fr[cb ^ 1][0]), "+v"(fr[cb ^ 1][1]) : "v"(fr[cb][0].y), "v"(fr[cb][1].y), "v"(b.y));
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %6, %0\n\tv_mfma_f32_32x32x2_f32 %1, %5, %6, %1"
                         : "+a"(c0), "+a"(c1), "+v"(fr[cb ^ 1][0]), "+v"(fr[cb ^ 1][1]) : "v"(fr[cb][0].z), "v"(fr[cb][1].z), "v"(b.z));
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %6, %0\n\tv_mfma_f32_32x32x2_f32 %1, %5, %6, %1\n\ts_waitcnt lgkmcnt(0)"
                         : "+a"(c0), "+a"(c1), "+v"(fr[cb ^ 1][0]), "+v"(fr[cb ^ 1][1]) : "v"(fr[cb][0].w), "v"(fr[cb][1].w), "v"(b.w));
        }
        if (MODE >= 2) __syncthreads();
        if (MODE >= 3) {   // 32 scattered dword stores per tile, accumulator layout
            asm volatile("s_nop 15\n\ts_nop 7");
            float* o = out + (static_cast<size_t>(blockIdx.x) * tiles + t) * 64 * 128 + (threadIdx.x >> 6) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 128] = c0[r];
                o[(32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 128] = c1[r];
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    if (MODE < 3) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const float* in, float* out, const char* what, int tiles = 512) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, in, out, tiles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, in, out, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * 512 * 4 * tiles * 128.0 * 4096.0;
    printf("%-60s tiles/block %4d  %.1f TFLOP/s   %.1f us per launch\n", what, tiles, flops / (ms * 1e-3) / 1e12, ms * 1e3 / 5);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 4096 * 4);
    hipMalloc(&out, 512ull * 512 * 64 * 128 * 4);
    float h[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>(in, out, "bare asm MFMA pairs (2 acc, 2 waves/SIMD)");
    run<1>(in, out, "+ ds_read_b128 fragment prefetch per step");
    run<2>(in, out, "+ __syncthreads per 128-MFMA tile");
    run<3>(in, out, "+ 32 dword stores per tile (unpipelined)");
    for (int t : {8, 16, 32, 64, 128}) run<2>(in, out, "fixed cost per launch: MFMA + LDS + barrier", t);
    return 0;
}
