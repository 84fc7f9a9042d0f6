// clock_probe.hip — what shader clock do short dependent launches run at?  Each launch runs a fixed chain of fp32 MFMAs per wave and
// records shader-clock ticks (clock64 = s_memtime) against the constant 100 MHz counter (wall_clock64 = s_memrealtime): ticks / time = MHz.
//   arm A: one long launch (1e5 MFMAs per wave, 256 x 768 threads) — the clock under sustained matrix load
//   arm B: a stream of 2000 short launches (32 MFMAs per wave, 192 x 768 threads: the GPT-2 decode step's products), each timing itself
// hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(768) void chain(int n, float seed, float* sink, unsigned long long* rec) {
    f16v acc;
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    const float a = seed * (threadIdx.x + 1), b = seed * 0.5f;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += acc[j];
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { rec[0] = c1 - c0; rec[1] = w1 - w0; }
}
int main() {
    float* sink; unsigned long long* rec;
    hipMalloc(&sink, 4); hipMalloc(&rec, 2000 * 16);
    unsigned long long h[2];
    for (int rep = 0; rep < 3; ++rep) {
        chain<<<256, 768>>>(100000, 0.001f, sink, rec);
        hipDeviceSynchronize();
        hipMemcpy(h, rec, 16, hipMemcpyDeviceToHost);
        printf("long launch : %llu shader ticks in %.1f us -> %.0f MHz; %.1f ticks per MFMA and wave\n", h[0], h[1] / 100.0, h[0] / (h[1] / 100.0),
               (double)h[0] / 100000);
    }
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int k = 0; k < 2000; ++k) chain<<<192, 768>>>(32, 0.001f, sink, rec + 2 * k);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> v(4000);
        hipMemcpy(v.data(), rec, 32000, hipMemcpyDeviceToHost);
        std::vector<double> mhz, us;
        for (int k = 100; k < 2000; ++k) { mhz.push_back(v[2 * k] / (v[2 * k + 1] / 100.0)); us.push_back(v[2 * k + 1] / 100.0); }
        std::sort(mhz.begin(), mhz.end()); std::sort(us.begin(), us.end());
        printf("short launches: %.2f us per launch start to start; in-kernel chain of 32 MFMAs: median %.2f us at %.0f MHz (p10 %.0f, p90 %.0f)\n", ms * 1e3 / 2000,
               us[us.size() / 2], mhz[mhz.size() / 2], mhz[mhz.size() / 10], mhz[mhz.size() * 9 / 10]);
    }
    return 0;
}
