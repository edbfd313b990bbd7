// Developer tool: per-phase shader-clock breakdown of the bf16x6 row-GEMM workgroup program.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DDG_TIMING -Iinclude -Idruggen_amd/csrc \
//         scripts/ubench/gemm_timing.cpp druggen_amd/csrc/runtime.hip -o /tmp/gemm_timing && /tmp/gemm_timing
#include "../../druggen_amd/csrc/row_gemm.hip"

#include <vector>

int main() {
    const int64_t R = 256 * 45 * 45;
    struct Shape { int K, N; bool res; } shapes[] = {{128, 128, false}, {128, 128, true}, {128, 384, false}, {384, 128, false}, {384, 128, true}};
    float *a, *y, *w, *packed, *res;
    hipMalloc(&a, R * 384 * 4);
    hipMalloc(&y, R * 384 * 4);
    hipMalloc(&res, R * 128 * 4);
    hipMalloc(&w, 384 * 128 * 4);
    hipMalloc(&packed, 384 * 128 * 4 * 2);
    std::vector<float> h(R * 384);
    srand(2);
    for (auto& v : h) v = rand() / (float)RAND_MAX - 0.5f;
    hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(res, h.data(), R * 128 * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), 384 * 128 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (auto sh : shapes) {
        dg_row_gemm_pack(w, packed, sh.N, sh.K, 0, nullptr);
        auto run = [&]() {
            dg_row_gemm(a, packed, y, R, sh.K, sh.N, nullptr, 0, nullptr, nullptr, sh.res ? res : nullptr, nullptr, nullptr,
                        nullptr, nullptr, nullptr, 1e-5f, nullptr);
        };
        for (int i = 0; i < 3; ++i) run();
        hipDeviceSynchronize();
        unsigned long long zero[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(dg::dg_timing), zero, sizeof(zero));
        const int reps = 10;
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) run();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t[16];
        hipMemcpyFromSymbol(t, HIP_SYMBOL(dg::dg_timing), sizeof(t));
        const double units = (double)t[5];
        printf("K=%d N=%d res=%d: %.1f us | per unit, clocks: consumer total %.0f = mfma-phase %.0f + barrierA %.0f + ex-write %.0f + barrierB %.0f"
               " | mover: split+write %.0f, fetch issue %.0f, barrier wait %.0f, store phase %.0f\n",
               sh.K, sh.N, (int)sh.res, ms * 1e3 / reps, t[4] / units, t[0] / units, t[1] / units, t[2] / units, t[3] / units,
               t[8] / units, t[9] / units, t[10] / units, t[11] / units);
    }
    return 0;
}
