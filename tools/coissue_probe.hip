// coissue_probe.hip — does a SIMD overlap one wave's MFMAs with ANOTHER wave's VALU work?  512-thread workgroups (two waves per SIMD),
// one per CU.  Even waves run a chain-free MFMA loop (4 accumulators), odd waves a packed-fp16 FMA loop (16 independent chains);
// each role is timed alone (the other role's waves exit at once) and together.  together ~ max(alone): the units overlap;
// together ~ sum: they do not.  A third arm gives BOTH roles to every wave (MFMA and VALU instructions interleaved in one stream).
// Standalone: hipcc --offload-arch=gfx950 -O3 tools/coissue_probe.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

// mode bit 0: MFMA role active, bit 1: VALU role active, bit 2: both roles in EVERY wave
template <int NM, int NV>
__global__ __launch_bounds__(512, 2) void probe(float* out, int iters, int mode) {
    const int t = threadIdx.x, wave = t >> 6;
    const bool both = mode & 4;
    const bool do_m = both || ((mode & 1) && (wave & 1) == 0);
    const bool do_v = both || ((mode & 2) && (wave & 1) == 1);
    f16v acc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * ((t + j) & 31)); b[j] = (_Float16)(0.02f * ((t - j) & 15)); }
    h2 v[16];
    for (int i = 0; i < 16; ++i) { v[i][0] = (_Float16)(0.001f * (t + i)); v[i][1] = (_Float16)(0.002f * i); }
    const h2 m = {(_Float16)0.999f, (_Float16)1.001f}, c = {(_Float16)0.0001f, (_Float16)-0.0001f};
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
#pragma unroll
            for (int k = 0; k < NM; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k & 3], 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k & 15] = v[k & 15] * m + c;          // v_pk_fma_f16
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int i = 0; i < 16; ++i) s += (float)v[i][0] + (float)v[i][1];
    out[blockIdx.x * 512 + t] = s;
}

template <int NM, int NV>
static float run(float* out, int blocks, int iters, int mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 20; ++w) probe<NM, NV><<<blocks, 512>>>(out, iters, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int w = 0; w < 20; ++w) probe<NM, NV><<<blocks, 512>>>(out, iters, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 20;
}

int main() {
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
    float* out;
    hipMalloc(&out, cus * 512 * 4);
    const int iters = 4000;
    printf("# one 512-thread workgroup per CU (two waves per SIMD); per launch: %d iterations\n", iters);
    printf("# per iteration an MFMA wave issues n_mfma v_mfma_f32_32x32x16_f16, a VALU wave n_valu v_pk_fma_f16\n");
    auto arm = [&](auto nmc, auto nvc) {
        constexpr int nm = decltype(nmc)::value, nv = decltype(nvc)::value;
        const float tm = run<nm, nv>(out, cus, iters, 1), tv = run<nm, nv>(out, cus, iters, 2), tb = run<nm, nv>(out, cus, iters, 3),
                    ti = run<nm, nv>(out, cus, iters, 4);
        printf("n_mfma %d n_valu %2d: MFMA waves alone %.3f ms, VALU waves alone %.3f ms, side by side %.3f ms (sum %.3f, max %.3f); "
               "every wave both roles %.3f ms (= 2x the work per SIMD)\n", nm, nv, tm, tv, tb, tm + tv, tm > tv ? tm : tv, ti);
    };
    arm(std::integral_constant<int, 8>{}, std::integral_constant<int, 16>{});
    arm(std::integral_constant<int, 8>{}, std::integral_constant<int, 32>{});
    arm(std::integral_constant<int, 8>{}, std::integral_constant<int, 64>{});
    arm(std::integral_constant<int, 4>{}, std::integral_constant<int, 64>{});
    hipFree(out);
    return 0;
}
