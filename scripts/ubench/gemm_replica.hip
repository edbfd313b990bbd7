// Simplified replica of the 128x128 row-GEMM workgroup program, ingredients switched on one at a time
// (2 blocks/CU x 4 waves, 64-row tiles, 2 accumulators per wave):
//   MODE 0  MFMA + LDS fragment prefetch + barrier per tile (LDS tile never changes)
//   MODE 1  + LDS-DMA of a fresh tile from HBM per iteration (double buffer, vmcnt(0) + barrier)
//   MODE 2  + previous tile's 32 dword stores spread between the MFMAs, vmcnt(32) wait
//   MODE 3  same as 2 but the stores go through an LDS exchange: 8 dwordx4 row stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma(const float* base, unsigned off, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory", "m0");
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ in, const float* __restrict__ wts, float* __restrict__ out,
                                            int tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][64*128]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, col = lane & 31;
    f32x4 bf[16];
    for (int q = 0; q < 16; ++q) bf[q] = *reinterpret_cast<const f32x4*>(wts + ((w * 16 + q) * 64 + lane) * 4);
    const unsigned lbase = static_cast<unsigned>(reinterpret_cast<size_t>((const __attribute__((address_space(3))) void*)lds));
    unsigned doff[8];
    for (int j = 0; j < 8; ++j) {
        const int L = (w + 4 * j) * 64 + lane, row = L / 32, cs = L % 32;
        doff[j] = (row * 128 + ((cs ^ (row & 15)) * 4)) * 4;
    }
    const float* mine = in + static_cast<size_t>(blockIdx.x) * tiles * 64 * 128;
    float* yout = out + static_cast<size_t>(blockIdx.x) * tiles * 64 * 128;
    for (int j = 0; j < 8; ++j) dma(mine, doff[j], lbase + (w + 4 * j) * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 c0, c1, p0, p1;
    for (int i = 0; i < 16; ++i) c0[i] = c1[i] = p0[i] = p1[i] = 0.f;
    int buf = 0;
    for (int t = 0; t < tiles; ++t, buf ^= 1) {
        if (MODE >= 1 && t + 1 < tiles) {
            const float* tb = mine + static_cast<size_t>(t + 1) * 64 * 128;
            for (int j = 0; j < 8; ++j) dma(tb, doff[j], lbase + (buf ^ 1) * 32768 + (w + 4 * j) * 1024);
        }
        const unsigned at = lbase + (MODE >= 1 ? buf * 32768 : 0);
        f32x4 fr[2][2];
        {
            const unsigned a0 = at + 4 * (col * 128 + (((half * 16) ^ (col & 15)) << 2));
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16384\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(fr[0][0]), "=&v"(fr[0][1]) : "v"(a0));
        }
        float* yrow = yout + static_cast<size_t>(t > 0 ? t - 1 : 0) * 64 * 128 + 32 * w + col;
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int cb = st & 1;
            const f32x4 b = bf[st];
            const unsigned a0 = at + 4 * (col * 128 + (((half * 16 + ((st + 1) & 15)) ^ (col & 15)) << 2));
            auto stores = [&](int i) {
                if (MODE == 4 && t > 0 && i == 3) {
#pragma unroll
                    for (int r = st * 2; r < st * 2 + 2; ++r) {
                        const float v = (r < 16 ? p0[r & 15] : p1[r & 15]) + 1.0f;
                        yrow[((r >> 4) * 32 + (r & 3) + 8 * ((r & 15) >> 2) + 4 * half) * 128] = v;
                    }
                }
                if (MODE == 5 && t > 0 && (i & 1)) {
                    const int r = st * 2 + i / 2;
                    const unsigned voff = (((r >> 4) * 32 + (r & 3) + 8 * ((r & 15) >> 2) + 4 * half) * 128 + 32 * w + col) * 4;
                    const float* sb = yout + static_cast<size_t>(t - 1) * 64 * 128;
                    if (r < 16) asm volatile("global_store_dword %0, %1, %2" ::"v"(voff), "a"(p0[r & 15]), "s"(sb) : "memory");
                    else asm volatile("global_store_dword %0, %1, %2" ::"v"(voff), "a"(p1[r & 15]), "s"(sb) : "memory");
                }
                if ((MODE == 2 || MODE == 3) && t > 0) {
                    const int r = st * 2 + i / 2;   // 2 registers per step, after quarters 1 and 3
                    if (i & 1) {
                        const float v = (r < 16 ? p0[r & 15] : p1[r & 15]) + 1.0f;
                        if (MODE == 2 || v == 12345.f) yrow[((r >> 4) * 32 + (r & 3) + 8 * ((r & 15) >> 2) + 4 * half) * 128] = v;
                    }
                }
            };
            asm volatile("ds_read_b128 %2, %4\n\tds_read_b128 %3, %4 offset:16384\n\t"
                         "v_mfma_f32_32x32x2_f32 %0, %5, %7, %0\n\tv_mfma_f32_32x32x2_f32 %1, %6, %7, %1"
                         : "+a"(c0), "+a"(c1), "=&v"(fr[cb ^ 1][0]), "=&v"(fr[cb ^ 1][1])
                         : "v"(a0), "v"(fr[cb][0].x), "v"(fr[cb][1].x), "v"(b.x));
            stores(0);
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %6, %0\n\tv_mfma_f32_32x32x2_f32 %1, %5, %6, %1"
                         : "+a"(c0), "+a"(c1), "+v"(fr[cb ^ 1][0]), "+v"(fr[cb ^ 1][1]) : "v"(fr[cb][0].y), "v"(fr[cb][1].y), "v"(b.y));
            stores(1);
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %6, %0\n\tv_mfma_f32_32x32x2_f32 %1, %5, %6, %1"
                         : "+a"(c0), "+a"(c1), "+v"(fr[cb ^ 1][0]), "+v"(fr[cb ^ 1][1]) : "v"(fr[cb][0].z), "v"(fr[cb][1].z), "v"(b.z));
            stores(2);
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %6, %0\n\tv_mfma_f32_32x32x2_f32 %1, %5, %6, %1\n\ts_waitcnt lgkmcnt(0)"
                         : "+a"(c0), "+a"(c1), "+v"(fr[cb ^ 1][0]), "+v"(fr[cb ^ 1][1]) : "v"(fr[cb][0].w), "v"(fr[cb][1].w), "v"(b.w));
            stores(3);
        }
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        if (MODE == 6) {
            const float* sb = yout + static_cast<size_t>(t) * 64 * 128;
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const unsigned voff = (((r >> 4) * 32 + (r & 3) + 8 * ((r & 15) >> 2) + 4 * half) * 128 + 32 * w + col) * 4;
                if (r < 16) asm volatile("global_store_dword %0, %1, %2" ::"v"(voff), "a"(c0[r & 15]), "s"(sb) : "memory");
                else asm volatile("global_store_dword %0, %1, %2" ::"v"(voff), "a"(c1[r & 15]), "s"(sb) : "memory");
            }
            for (int i = 0; i < 16; ++i) c0[i] = c1[i] = 0.f;
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        } else if (MODE >= 2) {
            p0 = c0;
            p1 = c1;
            for (int i = 0; i < 16; ++i) c0[i] = c1[i] = 0.f;
            if (t > 0 && (MODE == 2 || MODE == 4 || MODE == 5)) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + p0[i] + p1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const float* in, const float* w, float* out, const char* what, int tiles) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 65536, 0, in, w, out, tiles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 65536, 0, in, w, out, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 10.0 * 512 * 4 * tiles * 128.0 * 4096.0;
    printf("%-70s tiles/block %3d  %6.1f TFLOP/s  %7.1f us per launch\n", what, tiles, flops / (ms * 1e-3) / 1e12, ms * 1e3 / 10);
}

int main() {
    const int tiles = 16;
    const size_t n = 512ull * tiles * 64 * 128;
    float *in, *out, *w;
    hipMalloc(&in, n * 4);
    hipMalloc(&out, n * 4);
    hipMalloc(&w, 16384 * 4);
    float* h = (float*)malloc(n * 4);
    srand(1);
    for (size_t i = 0; i < n; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f);
    hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, h, 16384 * 4, hipMemcpyHostToDevice);
    run<0>(in, w, out, "MFMA + LDS prefetch + barrier per tile", tiles);
    run<1>(in, w, out, "+ LDS-DMA of a fresh 32 KB tile per iteration", tiles);
    run<2>(in, w, out, "+ pipelined dword stores of the previous tile, vmcnt(32)", tiles);
    run<3>(in, w, out, "  same VALU / accvgpr reads, stores predicated off", tiles);
    run<4>(in, w, out, "  stores clustered after the 8th MFMA of each step", tiles);
    run<5>(in, w, out, "  stores straight from AGPRs (no VALU), spread", tiles);
    run<6>(in, w, out, "  unpipelined: 32 AGPR stores after the phase", tiles);
    return 0;
}
