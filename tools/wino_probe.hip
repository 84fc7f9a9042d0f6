// wino_probe.hip — go / no-go probe for Winograd F(2x2, 3x3) on the LDS-DMA convs: the CEILING of the K loop a
// fused Winograd form of conv_gldsp would run, measured beside the K loop of today's direct form on the same box.
//
// Winograd turns 36 MFMAs (9 taps x 4 output pixels) into 16 (one per transform position), but every output pixel then owns FOUR
// fp32 accumulators instead of one: at the 128 accumulator registers a wave has at two waves per SIMD its tile shrinks to
// 2 positions x (2 blocks of 32 Winograd tiles) x (2 blocks of 32 channels), and the operands of its 8 MFMAs per 16-deep K step are
//   - wino-raw : 16 raw patch fragments (2 rows x 4 columns of the 4x4 input tile, per tile block) -> 48 packed-fp16 adds -> 4 V
//                fragments, + 4 transformed-weight fragments                      (20 ds_read_b128 + 48 v_pk_add per 8 MFMAs)
//   - wino-v   : V written to LDS by a separate transform pass: 4 V + 4 U fragments                 (8 ds_read_b128 per 8 MFMAs)
//   - direct   : conv_gldsp today: 2 pixel + 4 weight fragments                                      (6 ds_read_b128 per 8 MFMAs)
// 512-thread workgroups, one per CU (two waves per SIMD), operands resident in LDS (no DMA, no barrier, no epilogue, no input /
// output transform passes: everything a real kernel adds comes ON TOP).  Reported: matrix-core TFLOP/s actually issued and the
// ALGORITHMIC rate of a 3x3 convolution that loop would deliver (x 2.25 for the Winograd arms).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(float* out, int iters, float scale) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 131072 / 4; i += 512) ((float*)lds)[i] = scale * (0.001f * (i & 1023) - 0.5f);
    __syncthreads();
    f16v acc[8];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    // conflict-free fragment addresses (lane l: 64-byte row l & 31, 16-byte chunk swizzled as the conv kernels do)
    const int frag = ((lane & 31) << 6) + ((((lane >> 5) * 2) ^ ((lane >> 2) & 3)) << 4);
    const int wbase = 65536 + wave * 4096;
    for (int it = 0; it < iters; ++it) {
        const int step = (it & 7) * 2048;
        if (MODE == 0) {            // direct: 2 pixel fragments x 4 weight fragments
            h8 px[2], wf[4];
#pragma unroll
            for (int m = 0; m < 2; ++m) px[m] = *(const h8*)(lds + step + m * 16384 + frag);
#pragma unroll
            for (int n = 0; n < 4; ++n) wf[n] = *(const h8*)(lds + wbase + ((step + n * 2048) & 32767) + frag);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m * 4 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[n], px[m], acc[m * 4 + n], 0, 0, 0);
        } else if (MODE == 1) {     // wino-v: V fragments come ready from LDS
            h8 v[2][2], u[2][2];    // [position][tile block], [position][channel block]
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int m = 0; m < 2; ++m) v[p][m] = *(const h8*)(lds + step + (p * 2 + m) * 8192 + frag);
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int n = 0; n < 2; ++n) u[p][n] = *(const h8*)(lds + wbase + ((step + (p * 2 + n) * 2048) & 32767) + frag);
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[(p * 2 + m) * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(u[p][n], v[p][m], acc[(p * 2 + m) * 2 + n], 0, 0, 0);
        } else {                    // wino-raw: V = B^T d B for positions (i, j0), (i, j1) of two tile blocks, formed in packed fp16
            h8 v[2][2], u[2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                h8 d[2][4];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) d[a][b] = *(const h8*)(lds + ((step + m * 32768 + (a * 4 + b) * 2048) & 65535) + frag);
                h8 r[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) r[b] = d[0][b] - d[1][b];        // row i of B^T: a two-term combination
                v[0][m] = r[0] - r[2];                                       // column j0
                v[1][m] = r[1] + r[2];                                       // column j1
            }
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int n = 0; n < 2; ++n) u[p][n] = *(const h8*)(lds + wbase + ((step + (p * 2 + n) * 2048) & 32767) + frag);
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[(p * 2 + m) * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(u[p][n], v[p][m], acc[(p * 2 + m) * 2 + n], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 512 + t] = s;
}

template <int MODE>
void run(const char* name, double alg_factor) {
    float* out;
    const int blocks = 256;
    (void)hipMalloc(&out, blocks * 512 * 4);
    (void)hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int iters = 8000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int arm = 0; arm < 2; ++arm) {
        const float scale = arm ? 1.f : 0.f;
        for (int w = 0; w < 40; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), 131072, 0, out, iters, scale);
        (void)hipDeviceSynchronize();
        const int launches = 40;
        (void)hipEventRecord(e0);
        for (int w = 0; w < launches; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), 131072, 0, out, iters, scale);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)launches * blocks * 8 * iters * 8 * 32768.0;
        printf("%-10s %-8s %8.2f ms  matrix cores %7.1f TFLOP/s  -> algorithmic 3x3 conv rate %7.1f TFLOP/s\n", name, arm ? "nonzero" : "zeros", ms,
               fl / ms * 1e-9, alg_factor * fl / ms * 1e-9);
    }
    (void)hipFree(out);
}

int main() {
    run<0>("direct", 1.0);
    run<1>("wino-v", 2.25);
    run<2>("wino-raw", 2.25);
    return 0;
}
