// conv_direct.hip — generic MFMA implicit-GEMM convolution / GEMM, operands straight
// from global memory (no LDS).  This is the shape-agnostic path: it serves the
// low-resolution layers (W < 32), odd shapes and every test as the in-library
// cross-check of the LDS-tiled kernel (conv_tiled.hip).  Same epilogue contract.
//
// Replaces, per layer, the reference's per-sample weight materialisation + grouped
// conv (stylegan2/modules.py:920-967) by activation-side modulation:
//   conv(x, W*s*d) == d[b,o] * conv(x * s[b,i], W)      (SURVEY 8a note 1)
// and the transposed conv + FIR (modules.py:1089-1139) by a 3x3 conv I -> 4*O with
// the FIR folded into four phase kernels + depth-to-space (engine.cpp fold_upconv()).
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

// Block = one 32 x (32*NW) output tile; the K loop (taps x 16-channel steps) is split
// round-robin over the 4 waves (split-K, summed through LDS) and software-pipelined one
// step ahead, so the small-M layers are no longer a single wave chasing load latency.
// SPLITK = true : block = one 32-row tile, K split over the 4 waves (small M: more parallelism)
// SPLITK = false: block = 128 rows, wave w owns rows 32w..32w+31 over the full K (large M: the
//                 4 waves read the same weight fragments -> L1 hits instead of 4x the L2 traffic)
template <int NW, bool SPLITK>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvParams p) {
    __shared__ float red[SPLITK ? 3 : 1][SPLITK ? NW : 1][16][SPLITK ? 64 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, kh = lane >> 5;
    const long long M = (long long)p.B * p.Hc * p.Wc;
    const long long mw = SPLITK ? (long long)blockIdx.x * 32 : (long long)blockIdx.x * 128 + wave * 32;  // first GEMM row of this wave
    const int n0 = blockIdx.y * (32 * NW);
    const long long m = mw + r;
    const bool mvalid = m < M;
    int b = 0, oy = 0, ox = 0;
    if (mvalid) {
        const int hw = p.Hc * p.Wc;
        b = (int)(m / hw);
        const int rem = (int)(m - (long long)b * hw);
        oy = rem / p.Wc;
        ox = rem - oy * p.Wc;
    }
    f16x acc[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

    const half_t* xb = p.x + (long long)b * p.x_bstride + kh * 8;
    const float* snb = p.sn ? p.sn + (long long)b * p.sn_stride + kh * 8 : nullptr;
    const float* psb = p.pre_shift ? p.pre_shift + (long long)b * p.sn_stride + kh * 8 : nullptr;
    const int cps = p.Cin >> 4;                 // 16-channel steps per tap
    const int total = p.KS * p.KS * cps;
    bool nvalid[NW];
#pragma unroll
    for (int nw = 0; nw < NW; ++nw) nvalid[nw] = n0 + nw * 32 + r < p.Neff;

    // Every load of a K step is UNCONDITIONAL (out-of-image pixels / out-of-range weight rows read a valid clamped
    // address): a load under a branch makes the compiler wait for it at the join, which drained the software pipeline
    // every step.  Padding is applied when the fragment is consumed (`ok`); out-of-range columns are never stored.
    struct Frag { h8 a; h8 bf[NW]; f4 s0, s1, t0, t1; bool ok; };
    int nrow[NW];
#pragma unroll
    for (int nw = 0; nw < NW; ++nw) nrow[nw] = min(n0 + nw * 32 + r, p.Neff - 1);
    auto load = [&](int s, Frag& f) {
        const int tap = s / cps, i0 = (s - tap * cps) << 4;
        const int ty = tap / p.KS, tx = tap - ty * p.KS;
        const int iy = oy * p.stride + ty - p.pad, ix = ox * p.stride + tx - p.pad;
        f.ok = mvalid && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const long long poff = f.ok ? ((long long)(iy >> p.in_up) * (p.W >> p.in_up) + (ix >> p.in_up)) * p.Cin : 0;
        f.a = *(const h8*)(xb + poff + i0);
        if (snb) {
            f.s0 = *(const f4*)(snb + i0);
            f.s1 = *(const f4*)(snb + i0 + 4);
        }
        if (psb) {
            f.t0 = *(const f4*)(psb + i0);
            f.t1 = *(const f4*)(psb + i0 + 4);
        }
        const half_t* wp = p.w + (long long)tap * p.Neff * p.Cin + kh * 8 + i0;
#pragma unroll
        for (int nw = 0; nw < NW; ++nw) f.bf[nw] = *(const h8*)(wp + (long long)nrow[nw] * p.Cin);
    };
    auto compute = [&](Frag& f) {
        h8 a = f.a;
        if (psb) {   // pre-activation: relu(x * s + shift) for in-bounds pixels
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = (half_t)fmaxf((float)a[j] * f.s0[j] + f.t0[j], 0.f);
                a[j + 4] = (half_t)fmaxf((float)a[j + 4] * f.s1[j] + f.t1[j], 0.f);
            }
        } else if (snb) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = (half_t)((float)a[j] * f.s0[j]);
                a[j + 4] = (half_t)((float)a[j + 4] * f.s1[j]);
            }
        }
        if (!f.ok) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = (half_t)0.f;
        }
#pragma unroll
        for (int nw = 0; nw < NW; ++nw) acc[nw] = mfma32(a, f.bf[nw], acc[nw]);
    };
    Frag f0, f1;
    constexpr int KS_ = SPLITK ? 4 : 1;   // K-step stride between this wave's steps
    int s = SPLITK ? wave : 0;
    if (s < total) load(s, f0);
    for (; s < total; s += 2 * KS_) {     // two steps per iteration: static register double buffer
        if (s + KS_ < total) load(s + KS_, f1);
        compute(f0);
        if (s + KS_ < total) {
            if (s + 2 * KS_ < total) load(s + 2 * KS_, f0);
            compute(f1);
        }
    }
    // ---- split-K reduction: waves 1..3 -> LDS -> wave 0 --------------------------------------
    if (SPLITK) {
    if (wave > 0) {
#pragma unroll
        for (int nw = 0; nw < NW; ++nw)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) red[wave - 1][nw][reg][lane] = acc[nw][reg];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int nw = 0; nw < NW; ++nw)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            acc[nw][reg] += red[0][nw][reg][lane] + red[1][nw][reg][lane] + red[2][nw][reg][lane];
    }

    // ---- epilogue: demod, noise, bias, activation, residual, store -------------
    const int col = lane & 31;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = mfma32_row(reg, lane);
        const int rb = __shfl(b, row), ry = __shfl(oy, row), rx = __shfl(ox, row);
        const bool rvalid = (mw + row) < M;
#pragma unroll
        for (int nw = 0; nw < NW; ++nw) {
            const int n = n0 + nw * 32 + col;
            if (!rvalid || n >= p.Neff) continue;
            int o = n, py = ry, px = rx;
            if (p.up) {
                const int ph = n / p.Cout;
                o = n - ph * p.Cout;
                py = 2 * ry + (ph >> 1);
                px = 2 * rx + (ph & 1);
            }
            float val = acc[nw][reg];
            if (p.dscale) val *= p.dscale[(long long)rb * p.ds_stride + o];
            if (p.noise) val += p.noise_strength * p.noise[((long long)(rb / p.batch_size) * p.Ho + py) * p.Wo + px];
            if (p.bias) val += p.bias[o];
            if (p.shift) val += p.shift[(long long)rb * p.ds_stride + o];
            if (p.act == 1) val = lrelu_sqrt2(val);
            else if (p.act == 2) val = fmaxf(val, 0.f);
            const long long oidx = (((long long)rb * p.Ho + py) * p.Wo + px) * p.Cout + o;
            if (p.res) {
                const int rcs = p.res_cs ? p.res_cs : p.Cout;
                const long long ridx = p.res_up ? (((long long)rb * (p.Ho >> 1) + (py >> 1)) * (p.Wo >> 1) + (px >> 1)) * rcs + o
                                                : (((long long)rb * p.Ho + py) * p.Wo + px) * rcs + o;
                val += (float)p.res[ridx];
            }
            val *= p.out_scale;
            if (p.y32) p.y32[oidx] = val;
            else p.y[oidx] = (half_t)val;
        }
    }
}

const char* launch_conv_direct(const ConvParams& p, hipStream_t st) {
    if (p.x_planar8 || p.y_planar8 || p.x_planar32) return nullptr;   // chunk-planar maps (common.h): not implemented here
    if (p.w_bstride != 0) return nullptr;  // per-sample weights are a tiled / up-conv-only path: refuse, the caller reports it
    const long long M = (long long)p.B * p.Hc * p.Wc;
    // instance choice at the NOMINAL population (common.h): the row-parallel and the split-K forms sum in different orders
    const long long Mn = (long long)GLASS_NOMINAL_POP * p.Hc * p.Wc;
    if (p.Neff > 64 && ((Mn + 127) / 128) * ((p.Neff + 127) / 128) >= 256) {   // enough 128 x 128 blocks to fill the chip: row-parallel
        hipLaunchKernelGGL((conv_direct_kernel<4, false>), dim3((unsigned)((M + 127) / 128), (p.Neff + 127) / 128), dim3(256), 0,
                           st, p);
        return "conv_direct_kernel<4,rows>";
    }
    const unsigned gx = (unsigned)((M + 31) / 32);
    // wide n tiles re-use the activation fragment; narrow ones give small problems more blocks
    if (p.Neff > 64 && ((Mn + 31) / 32) * ((p.Neff + 127) / 128) >= 512) {
        hipLaunchKernelGGL((conv_direct_kernel<4, true>), dim3(gx, (p.Neff + 127) / 128), dim3(256), 0, st, p);
        return "conv_direct_kernel<4>";
    }
    if (p.Neff > 32) {
        hipLaunchKernelGGL((conv_direct_kernel<2, true>), dim3(gx, (p.Neff + 63) / 64), dim3(256), 0, st, p);
        return "conv_direct_kernel<2>";
    }
    hipLaunchKernelGGL((conv_direct_kernel<1, true>), dim3(gx, 1), dim3(256), 0, st, p);
    return "conv_direct_kernel<1>";
}

// ---------------------------------------------------------------------------------
// GEMM  out[M][N] = A[M][K] * W[N][K]^T  (+bias, epilogue modes) — CLIP linears,
// patch embedding, D dense0.  Replaces nn.Linear / conv1 of clip/model.py:166-235.
// ---------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(256) void gemm_direct_kernel(GemmParams p) {
    if (p.batch > 1) {   // batched problems (BigGAN self-attention): one z-slice per problem
        p.a += (long long)blockIdx.z * p.a_bs;
        p.w += (long long)blockIdx.z * p.w_bs;
        if (p.out16) p.out16 += (long long)blockIdx.z * p.o_bs;
        if (p.out32) p.out32 += (long long)blockIdx.z * p.o_bs;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, kh = lane >> 5;
    const int mw = blockIdx.x * 128 + wave * 32;
    const int n0 = blockIdx.y * (32 * NW);
    const int m = mw + r;
    const bool mvalid = m < p.M;
    f16x acc[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const half_t* ap = p.a + (long long)m * p.K + kh * 8;
    const int rs = p.kpt ? p.kpt : p.K;
    for (int k0 = 0; k0 < p.K; k0 += 16) {
        const half_t* wp = (p.kpt ? p.w + (long long)(k0 / p.kpt) * p.w_tap_stride + (k0 % p.kpt) - k0 : p.w) + (long long)(n0 + r) * rs + kh * 8;
        h8 a;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = (half_t)0.f;
        if (mvalid) a = *(const h8*)(ap + k0);
#pragma unroll
        for (int nw = 0; nw < NW; ++nw) {
            h8 bf;
#pragma unroll
            for (int j = 0; j < 8; ++j) bf[j] = (half_t)0.f;
            if (n0 + nw * 32 + r < p.N) bf = *(const h8*)(wp + (long long)nw * 32 * rs + k0);
            acc[nw] = mfma32(a, bf, acc[nw]);
        }
    }
    const int col = lane & 31;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int mr = mw + mfma32_row(reg, lane);
        if (mr >= p.M) continue;
#pragma unroll
        for (int nw = 0; nw < NW; ++nw) {
            const int n = n0 + nw * 32 + col;
            if (n >= p.N) continue;
            float v = acc[nw][reg];
            if (p.bias) v += p.bias[n];
            const long long oi = (long long)mr * p.ldo + n;
            switch (p.mode) {
                case 0: p.out16[oi] = (half_t)v; break;
                case 1: p.out16[oi] = (half_t)(v / (1.f + __expf(-1.702f * v))); break;  // QuickGELU
                case 2: p.out32[oi] += v; break;
                case 3: p.out32[oi] = v; break;
                default: p.out32[oi] = lrelu_sqrt2(v); break;
            }
        }
    }
}

const char* launch_gemm_direct(const GemmParams& p, hipStream_t st) {
    const unsigned gx = (unsigned)((p.M + 127) / 128), gz = p.batch > 1 ? p.batch : 1;
    if (p.N > 64) {
        hipLaunchKernelGGL(gemm_direct_kernel<4>, dim3(gx, (p.N + 127) / 128, gz), dim3(256), 0, st, p);
        return "gemm_direct_kernel<4>";
    } else if (p.N > 32) {
        hipLaunchKernelGGL(gemm_direct_kernel<2>, dim3(gx, 1, gz), dim3(256), 0, st, p);
        return "gemm_direct_kernel<2>";
    }
    hipLaunchKernelGGL(gemm_direct_kernel<1>, dim3(gx, 1, gz), dim3(256), 0, st, p);
    return "gemm_direct_kernel<1>";
}
