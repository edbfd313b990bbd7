// Do global stores overlap with MFMA work on gfx950?  256 workgroups x 8 waves; every iteration a wave issues NM MFMAs and NS 16-byte-per-lane
// stores.  Prints microseconds for (MFMAs only), (stores only), (both) per store pattern.
//   hipcc --offload-arch=gfx950 -O3 -o store_overlap store_overlap.hip && ./store_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512) void k(float* out, int iters, int nm, int ns, float* sink) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(i * 0.5f - lane * 0.02f); }
    f32x4 acc[4] = {};
    // each wave owns a region of 16 rows x 512 B per iteration (8 KB), walking forward
    char* base = reinterpret_cast<char*>(out) + (static_cast<size_t>(blockIdx.x) * 8 + w) * static_cast<size_t>(iters) * 8192;
    u32x4 v = {static_cast<unsigned>(lane), 1u, 2u, 3u};
    for (int it = 0; it < iters; ++it) {
        for (int m = 0; m < nm; ++m)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
        char* p = base + static_cast<size_t>(it) * 8192;
        for (int s = 0; s < ns; ++s) {
            size_t off;
            if (PAT == 0) off = static_cast<size_t>(n) * 512 + kq * 32 + (s >> 1) * 128 + (s & 1) * 16;      // y-like: 16 B pieces, every second one
            else if (PAT == 1) off = static_cast<size_t>(s) * 1024 + lane * 16;                              // fully coalesced 1 KB per instruction
            else off = static_cast<size_t>(n) * 512 + kq * 16 + s * 64;                                     // 64 contiguous bytes per row
            *reinterpret_cast<u32x4*>(p + off) = v;
        }
    }
    float r = 0.f;
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (r == 123.456f) sink[0] = r;
}

template <int PAT>
float run(float* out, float* sink, int iters, int nm, int ns) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(512), 0, 0, out, iters, nm, ns, sink);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(512), 0, 0, out, iters, nm, ns, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / 5;
}

int main() {
    const int iters = 320;      // 256 wg x 8 waves x 320 x 8 KB = 5.4 GB region... keep it smaller: see ns
    float *out, *sink;
    const size_t bytes = static_cast<size_t>(256) * 8 * iters * 8192;
    hipMalloc(&out, bytes); hipMalloc(&sink, 4);
    printf("region %.2f GB; per launch with ns stores of 1 KB per wave and iteration: %d iterations\n", bytes / 1e9, iters);
    for (int nm : {0, 32, 64, 128}) {
        for (int ns : {0, 2, 4, 8}) {
            if (nm == 0 && ns == 0) continue;
            const float t0 = run<0>(out, sink, iters, nm, ns), t1 = run<1>(out, sink, iters, nm, ns), t2 = run<2>(out, sink, iters, nm, ns);
            const double gb = 256.0 * 8 * iters * ns * 1024 / 1e9;
            printf("mfma %3d  stores %d (%.2f GB)   strided-16B %8.1f us   coalesced %8.1f us   row-64B %8.1f us\n", nm, ns, gb, t0, t1, t2);
        }
    }
    return 0;
}
