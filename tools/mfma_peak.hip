// mfma_peak.hip — what this box's matrix cores sustain, with and without the LDS fragment traffic of the
// conv kernels.  Each arm is timed over >= 50 ms of back-to-back launches after a 100 ms warm-up (clocks ramped), once with
// ZERO operands (the data the 2495 TFLOP/s figure of MI355X_MICROARCH.md is reached on: the chip clocks to its power
// budget, ~2.3-2.4 GHz on zeros) and once with non-trivial operands (~1.9 GHz under matrix load; guide, DVFS give-back).
// bench.py prices every fraction against the 2.5 PFLOP/s spec peak regardless.  Standalone: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC, int LDSREADS>   // NACC independent accumulators; LDSREADS ds_read_b128 per NACC MFMAs
__global__ __launch_bounds__(256) void probe(float* out, int iters, float scale) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    const int t = threadIdx.x;
    for (int i = t; i < 32768 / 4; i += 256) ((float*)lds)[i] = scale * 0.001f * i;
    __syncthreads();
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(scale * 0.01f * (t + j)); b[j] = (_Float16)(scale * 0.02f * (t - j)); }
    h8 f[LDSREADS > 0 ? LDSREADS : 1];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < LDSREADS; ++r) f[r] = *(const h8*)(lds + ((t * 80 + r * 5120 + it * 16) & 32752));
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            h8 aa = LDSREADS > 0 ? f[i % LDSREADS] : a;
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aa, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + t] = s;
}

// the exact-fp32 matrix instruction of the GPT-2 path (v_mfma_f32_32x32x2_f32: 4096 FLOP per instruction)
template <int NACC>
__global__ __launch_bounds__(256) void probe_f32(float* out, int iters, float scale) {
    const int t = threadIdx.x;
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const float a = scale * 0.01f * (t + 1), b = scale * 0.02f * (t - 7);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + t] = s;
}
template <int NACC>
void run_f32(const char* name, int blocks) {
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    const int iters = 10000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int arm = 0; arm < 2; ++arm) {
        const float scale = arm ? 1.f : 0.f;
        for (int w = 0; w < 30; ++w) probe_f32<NACC><<<blocks, 256>>>(out, iters, scale);
        hipDeviceSynchronize();
        const int launches = 20;
        hipEventRecord(e0);
        for (int w = 0; w < launches; ++w) probe_f32<NACC><<<blocks, 256>>>(out, iters, scale);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)launches * blocks * 4 * iters * NACC * 4096.0;
        printf("%-34s blocks=%5d  %-8s %8.2f ms  %7.1f TFLOP/s\n", name, blocks, arm ? "nonzero" : "zeros", ms, fl / ms * 1e-9);
    }
    hipFree(out);
}

template <int NACC, int LR>
void run(const char* name, int blocks) {
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    const int iters = 20000;            // ~2 ms per launch
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int arm = 0; arm < 2; ++arm) {
        const float scale = arm ? 1.f : 0.f;
        for (int w = 0; w < 50; ++w) probe<NACC, LR><<<blocks, 256>>>(out, iters, scale);   // warm-up: ~100 ms
        hipDeviceSynchronize();
        const int launches = 40;        // >= 50 ms timed region
        hipEventRecord(e0);
        for (int w = 0; w < launches; ++w) probe<NACC, LR><<<blocks, 256>>>(out, iters, scale);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)launches * blocks * 4 * iters * NACC * 32768.0;
        printf("%-34s blocks=%5d  %-8s %8.2f ms  %7.1f TFLOP/s\n", name, blocks, arm ? "nonzero" : "zeros", ms, fl / ms * 1e-9);
    }
    hipFree(out);
}

int main() {
    run<8, 0>("regs only, 1 wave/SIMD", 256);
    run<8, 0>("regs only, 2 waves/SIMD", 512);
    run<8, 0>("regs only, 4 waves/SIMD", 1024);
    run<8, 6>("6 ds_read_b128 / 8 MFMA, 1 w/SIMD", 256);
    run<8, 6>("6 ds_read_b128 / 8 MFMA, 2 w/SIMD", 512);
    run<8, 6>("6 ds_read_b128 / 8 MFMA, 4 w/SIMD", 1024);
    run<4, 4>("4 ds_read_b128 / 4 MFMA, 2 w/SIMD", 512);
    run<4, 4>("4 ds_read_b128 / 4 MFMA, 3 w/SIMD", 768);
    run_f32<4>("fp32 32x32x2, regs, 1 wave/SIMD", 256);
    run_f32<4>("fp32 32x32x2, regs, 2 waves/SIMD", 512);
    run_f32<4>("fp32 32x32x2, regs, 3 waves/SIMD", 768);
    run_f32<4>("fp32 32x32x2, regs, 4 waves/SIMD", 1024);
    run_f32<2>("fp32 32x32x2, 2 accumulators, 1 w", 256);
    run_f32<1>("fp32 32x32x2, 1 accumulator, 1 w", 256);
    return 0;
}
