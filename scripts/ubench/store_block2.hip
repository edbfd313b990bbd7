// What makes a store hold up its wave (gfx950)?  8 waves per workgroup, 256 workgroups; per iteration a dependent v_fma chain (the wave's own
// critical path, pipes far from saturated) and two 16-byte-per-lane stores in one of several forms.  Reported: added nanoseconds per store.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// FORM 0: flat, coalesced, loop-invariant data.  1: flat, rows of 512 B (16 B pieces).  2: buffer store (raw, offen), rows of 512 B.
// 3: as 2 with data computed from the chain (fresh registers every iteration).  4: as 1 with fresh data.
template <int FORM, int NS>
__global__ __launch_bounds__(512) void k(float* out, int iters, int nv, float* sink, int rows) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    float a = lane * 0.001f, b = 1.0001f, c = 0.5f;
    char* base = reinterpret_cast<char*>(out) + (static_cast<size_t>(blockIdx.x) * 8 + w) * static_cast<size_t>(iters) * 8192;
    u32x4 v = {static_cast<unsigned>(lane), 1u, 2u, 3u};
    for (int it = 0; it < iters; ++it) {
        for (int m = 0; m < nv; m += 64) asm volatile(".rept 64\n\tv_fma_f32 %0, %0, %1, %2\n\t.endr" : "+v"(a) : "v"(b), "v"(c));
        char* p = base + static_cast<size_t>(it) * 8192;
        if (FORM == 3 || FORM == 4) v = u32x4{__float_as_uint(a), __float_as_uint(a + 1.f), __float_as_uint(a + 2.f), __float_as_uint(a + 3.f)};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (FORM == 0) *reinterpret_cast<u32x4*>(p + s * 1024 + lane * 16) = v;
            else if (FORM == 1 || FORM == 4) *reinterpret_cast<u32x4*>(p + n * 512 + kq * 32 + s * 16) = v;
            else if (FORM == 5) *reinterpret_cast<u32x4*>(p + n * 512 + kq * 16 + s * 64) = v;                       // 64 contiguous bytes per row
            else if (FORM == 6) *reinterpret_cast<u32x4*>(p + (s * 2 + (lane >> 5)) * 512 + (lane & 31) * 16) = v;     // two whole 512-byte rows
            else if (FORM == 7) *reinterpret_cast<u32x4*>(p + (s * 4 + (lane >> 4)) * 512 + (lane & 15) * 16) = v;     // 256 contiguous bytes of four rows
            else if (FORM == 8) *reinterpret_cast<u32x4*>(p + (s * 8 + (lane >> 3)) * 512 + (lane & 7) * 16) = v;      // 128 contiguous bytes of eight rows
            else {
                const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, rows * 512, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(v, r, static_cast<unsigned>(n) * 512u + kq * 32u, s * 16, 0);
            }
        }
    }
    if (a == 123.456f) sink[0] = a;
}

template <int FORM, int NS>
float run(float* out, float* sink, int iters, int nv) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<FORM, NS>), dim3(256), dim3(512), 0, 0, out, iters, nv, sink, 16);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<FORM, NS>), dim3(256), dim3(512), 0, 0, out, iters, nv, sink, 16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / 3;
}

template <int FORM>
void line(float* out, float* sink, int iters, int nv, const char* what) {
    const float t0 = run<FORM, 0>(out, sink, iters, nv), t2 = run<FORM, 2>(out, sink, iters, nv);
    printf("chain %4d  %-52s no store %7.1f us   2 stores %7.1f us   +%6.0f ns per store\n", nv, what, t0, t2, (t2 - t0) * 1e3 / iters / 2);
}

int main() {
    const int iters = 256;
    float *out, *sink;
    hipMalloc(&out, static_cast<size_t>(256) * 8 * iters * 8192); hipMalloc(&sink, 4);
    for (int nv : {512, 768}) {
        line<0>(out, sink, iters, nv, "flat, coalesced, loop-invariant data");
        line<1>(out, sink, iters, nv, "flat, 16 B pieces of 512 B rows, invariant data");
        line<2>(out, sink, iters, nv, "buffer (offen), 16 B pieces of 512 B rows, invariant");
        line<3>(out, sink, iters, nv, "buffer (offen), rows, data fresh from the chain");
        line<4>(out, sink, iters, nv, "flat, rows, data fresh from the chain");
        line<5>(out, sink, iters, nv, "flat, 64 contiguous bytes of each of 16 rows");
        line<8>(out, sink, iters, nv, "flat, 128 contiguous bytes of each of 8 rows");
        line<7>(out, sink, iters, nv, "flat, 256 contiguous bytes of each of 4 rows");
        line<6>(out, sink, iters, nv, "flat, two whole 512-byte rows");
    }
    return 0;
}
