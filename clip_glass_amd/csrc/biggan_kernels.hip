// biggan_kernels.hip — the small kernels of the BigGAN-deep generator path (config C3: DeepMindBigGAN256/512).
// The dense work (1x1 / 3x3 convs, attention GEMMs) runs on the shared MFMA kernels (conv_tiled / conv_direct /
// gemm_tiled); what is here is the glue around them, all NHWC fp16 activations with fp32 per-(sample, channel)
// affine tables:
//   cond      latent.py:20-24 (clip z, softmax class bits) + BigGAN.forward (embeddings, cat)
//   bn tables BigGANBatchNorm folded to  y = x * A[b,c] + S[b,c]   (A = gain * rsqrt(var + eps))
//   attention split + 2x2 max-pool, row softmax, tanh + NHWC -> planar RGB.
// (batch norm + ReLU, nearest x2 and the channel-drop skip are fused into the conv kernels: common.h ConvParams
//  pre_shift / in_up / shift / res_cs / res_up.)
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

// ---- cond = [clip(z, -2, 2) | softmax(class bits) @ E^T] ---------------------------------------------
// one block per candidate; x row = [z (zd) | class bits (nc)], et = E^T [nc][zd]
__global__ __launch_bounds__(256) void bg_cond_kernel(const float* __restrict__ x, int L, int zd, int nc,
                                                      const float* __restrict__ et, float* __restrict__ cond) {
    extern __shared__ float sm[];   // [nc] probabilities + [8] reduction scratch
    float* prob = sm;
    float* red = sm + nc;
    const int p = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* row = x + (long long)p * L;
    float mx = -3.4e38f;
    for (int k = t; k < nc; k += 256) mx = fmaxf(mx, row[zd + k]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int k = t; k < nc; k += 256) {
        const float ev = expf(row[zd + k] - mx);
        prob[k] = ev;
        sum += ev;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
    for (int n = t; n < zd; n += 256) {
        cond[(long long)p * 2 * zd + n] = fminf(fmaxf(row[n], -2.f), 2.f);
        float acc = 0.f;
        for (int k = 0; k < nc; ++k) acc += prob[k] * et[(long long)k * zd + n];
        cond[(long long)p * 2 * zd + zd + n] = acc * inv;
    }
}
void launch_bg_cond(const float* x, int P, int L, int zd, int nc, const float* et, float* cond, hipStream_t st) {
    hipLaunchKernelGGL(bg_cond_kernel, dim3(P), dim3(256), (nc + 8) * sizeof(float), st, x, L, zd, nc, et, cond);
}

// ---- batch-norm tables: in place  [gain | off] -> [A | S] ----------------------------------------------
// A = gain * inv_std ; S = off - mean * A + prebias * A   (prebias = bias of the conv feeding this BN)
__global__ void bg_bn_tables_kernel(float* __restrict__ tab, int P, int C, const float* __restrict__ inv_std,
                                    const float* __restrict__ mean, const float* __restrict__ prebias) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)P * C) return;
    const int c = (int)(i % C);
    const long long p = i / C;
    float* g = tab + p * 2 * C + c;
    const float a = g[0] * inv_std[c];
    g[0] = a;
    g[C] = g[C] + (prebias[c] - mean[c]) * a;
}
void launch_bg_bn_tables(float* tab, int P, int C, const float* inv_std, const float* mean, const float* prebias,
                         hipStream_t st) {
    const long long n = (long long)P * C;
    hipLaunchKernelGGL(bg_bn_tables_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tab, P, C, inv_std, mean,
                       prebias);
}

// ---- fp32 -> fp16 -------------------------------------------------------------------------------------
__global__ void bg_to_half_kernel(const float* __restrict__ x, half_t* __restrict__ y, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f4 v = ((const f4*)x)[i];
    h4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = (half_t)v[q];
    ((h4*)y)[i] = o;
}
void launch_bg_to_half(const float* x, half_t* y, long long n, hipStream_t st) {
    const long long n4 = n / 4;
    hipLaunchKernelGGL(bg_to_half_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, y, n4);
}

// ---- SelfAttn split: T [B][H][W][c8 + c8 + c2] (theta | phi | g) ->
//      theta [B][HW][c8] ; phi [B][HW/4][c8] (2x2 max-pool) ; gT [B][c2][HW/4] (2x2 max-pool, transposed)
__global__ void bg_attn_split_kernel(const half_t* __restrict__ T, int H, int W, int c8, int c2,
                                     half_t* __restrict__ theta, half_t* __restrict__ phi, half_t* __restrict__ gT) {
    const int CT = 2 * c8 + c2, hw = H * W, hq = hw / 4, Wq = W / 2;
    const long long b = blockIdx.y;
    const half_t* Tb = T + b * hw * CT;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n_theta = (long long)hw * c8, n_phi = (long long)hq * c8, n_g = (long long)hq * c2;
    if (i < n_theta) {
        const int c = (int)(i % c8);
        const long long m = i / c8;
        theta[b * n_theta + i] = Tb[m * CT + c];
    } else if (i < n_theta + n_phi + n_g) {
        long long j = i - n_theta;
        int c, q;
        const bool is_phi = j < n_phi;
        if (is_phi) {
            c = (int)(j % c8);
            q = (int)(j / c8);
        } else {          // g: consecutive threads walk pooled positions (the contiguous axis of gT)
            j -= n_phi;
            q = (int)(j % hq);
            c = (int)(j / hq);
        }
        const int qy = q / Wq, qx = q - qy * Wq;
        const int ch = is_phi ? c8 + c : 2 * c8 + c;
        const half_t* s = Tb + ((long long)(2 * qy) * W + 2 * qx) * CT + ch;
        const float v = fmaxf(fmaxf((float)s[0], (float)s[CT]), fmaxf((float)s[(long long)W * CT], (float)s[(long long)(W + 1) * CT]));
        if (is_phi) phi[b * n_phi + (long long)q * c8 + c] = (half_t)v;
        else gT[b * n_g + (long long)c * hq + q] = (half_t)v;
    }
}
// Vector form (round 4; the scalar kernel above read the pooled g values 2 bytes at a time, 4 pixels x CT apart: 320 us at 0.87 TB/s for the
// 512 px generator).  blockIdx.x walks three regions: theta rows (16-byte copies), phi (2x2 max of 16-byte vectors), and 32 pooled
// positions x 64 channels tiles of g, pooled from 16-byte vectors and transposed through LDS so that gT's rows leave as 16-byte runs.
__global__ __launch_bounds__(256) void bg_attn_split_vec_kernel(const half_t* __restrict__ T, int H, int W, int c8, int c2, half_t* __restrict__ theta,
                                                                half_t* __restrict__ phi, half_t* __restrict__ gT, int nb_theta, int nb_phi) {
    __shared__ half_t Ls[64][40];                    // [channel][32 pooled positions + pad]
    const int CT = 2 * c8 + c2, hw = H * W, hq = hw / 4, Wq = W / 2, t = threadIdx.x;
    const long long b = blockIdx.y;
    const half_t* Tb = T + b * hw * CT;
    int blk = blockIdx.x;
    auto pooled = [&](int q, int ch) {               // 2x2 max of the 8 channels starting at ch
        const int qy = q / Wq, qx = q - qy * Wq;
        const half_t* s = Tb + ((long long)(2 * qy) * W + 2 * qx) * CT + ch;
        const h8 a = *(const h8*)s, c = *(const h8*)(s + CT), d = *(const h8*)(s + (long long)W * CT), e = *(const h8*)(s + (long long)(W + 1) * CT);
        return __builtin_elementwise_max(__builtin_elementwise_max(a, c), __builtin_elementwise_max(d, e));
    };
    if (blk < nb_theta) {
        const int v8 = c8 >> 3;
        const long long i = (long long)blk * 256 + t;
        if (i < (long long)hw * v8) {
            const long long m = i / v8;
            const int v = (int)(i - m * v8);
            *(h8*)(theta + (b * hw + m) * c8 + v * 8) = *(const h8*)(Tb + m * CT + v * 8);
        }
        return;
    }
    blk -= nb_theta;
    if (blk < nb_phi) {
        const int v8 = c8 >> 3;
        const long long i = (long long)blk * 256 + t;
        if (i < (long long)hq * v8) {
            const int q = (int)(i / v8), v = (int)(i - (long long)q * v8);
            *(h8*)(phi + (b * hq + q) * c8 + v * 8) = pooled(q, c8 + v * 8);
        }
        return;
    }
    blk -= nb_phi;
    const int tiles_c = c2 >> 6;
    const int q0 = (blk / tiles_c) * 32, c0 = (blk % tiles_c) * 64;
    {
        const int ql = t >> 3, v = t & 7;
        const h8 m = pooled(q0 + ql, 2 * c8 + c0 + v * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) Ls[v * 8 + j][ql] = m[j];
    }
    __syncthreads();
    {
        const int c = t >> 2, part = t & 3;
        *(h8*)(gT + (b * c2 + c0 + c) * hq + q0 + part * 8) = *(const h8*)(&Ls[c][part * 8]);
    }
}
void launch_bg_attn_split(const half_t* T, int B, int H, int W, int c8, int c2, half_t* theta, half_t* phi, half_t* gT,
                          hipStream_t st) {
    const int hw = H * W, hq = hw / 4;
    static const bool scalar_only = glass_knob("GLASS_BG_SPLIT_SCALAR") != nullptr;      // A/B knob
    if (!scalar_only && c8 % 8 == 0 && c2 % 64 == 0 && hq % 32 == 0 && W % 2 == 0) {
        const int nb_theta = (int)(((long long)hw * (c8 >> 3) + 255) / 256), nb_phi = (int)(((long long)hq * (c8 >> 3) + 255) / 256);
        const int nb_g = (hq / 32) * (c2 >> 6);
        hipLaunchKernelGGL(bg_attn_split_vec_kernel, dim3((unsigned)(nb_theta + nb_phi + nb_g), B), dim3(256), 0, st, T, H, W, c8, c2, theta, phi, gT,
                           nb_theta, nb_phi);
        return;
    }
    const long long n = (long long)H * W * c8 + (long long)(H * W / 4) * (c8 + c2);
    hipLaunchKernelGGL(bg_attn_split_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, T, H, W, c8, c2, theta,
                       phi, gT);
}

// ---- row softmax: S fp32 [rows][n] -> P fp16 [rows][n]; one wave per row, the row held in registers (one pass over
// memory, all loads issued up front) for n <= 64 * 32; longer rows take the three-pass loop ---------------------------
template <int NV>   // NV float4 per lane: n == NV * 256
__global__ __launch_bounds__(256) void bg_softmax_reg_kernel(const float* __restrict__ S, long long rows, half_t* __restrict__ Pm) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int n = NV * 256;
    const float* s = S + row * n;
    f4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = *(const f4*)(s + (k * 64 + lane) * 4);
    float mx = -3.4e38f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) mx = fmaxf(mx, v[k][q]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[k][q] = expf(v[k][q] - mx);
            sum += v[k][q];
        }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.f / sum;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        h4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (half_t)(v[k][q] * inv);
        *(h4*)(Pm + row * n + (k * 64 + lane) * 4) = o;
    }
}
__global__ __launch_bounds__(256) void bg_softmax_kernel(const float* __restrict__ S, long long rows, int n,
                                                         half_t* __restrict__ Pm) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* s = S + row * n;
    float mx = -3.4e38f;
    for (int k = lane; k < n; k += 64) mx = fmaxf(mx, s[k]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int k = lane; k < n; k += 64) sum += expf(s[k] - mx);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.f / sum;
    for (int k = lane; k < n; k += 64) Pm[row * n + k] = (half_t)(expf(s[k] - mx) * inv);
}
void launch_bg_softmax(const float* S, long long rows, int n, half_t* Pm, hipStream_t st) {
    const dim3 g((unsigned)((rows + 3) / 4));
    if (n == 1024) hipLaunchKernelGGL(bg_softmax_reg_kernel<4>, g, dim3(256), 0, st, S, rows, Pm);
    else if (n == 256) hipLaunchKernelGGL(bg_softmax_reg_kernel<1>, g, dim3(256), 0, st, S, rows, Pm);
    else hipLaunchKernelGGL(bg_softmax_kernel, g, dim3(256), 0, st, S, rows, n, Pm);
}

// ---- y[b][c][p] = tanh(x[b][p][c]), c < 3 : conv_to_rgb output (NHWC, C channels) -> planar fp32 -------
__global__ void bg_rgb_tanh_kernel(const half_t* __restrict__ x, long long hw, int C, float* __restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long b = i / hw, p = i - b * hw;
    const half_t* s = x + i * C;
#pragma unroll
    for (int c = 0; c < 3; ++c) y[(b * 3 + c) * hw + p] = tanhf((float)s[c]);
}
void launch_bg_rgb_tanh(const half_t* x, int B, long long hw, int C, float* y, hipStream_t st) {
    const long long n = (long long)B * hw;
    hipLaunchKernelGGL(bg_rgb_tanh_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, hw, C, y, n);
}
