// conv_gemm.hip — the low-resolution layers (conv grid <= 16 x 16: 4x4 .. 16x16 maps of 512 channels) as im2col + the LDS-tiled
// GEMM + a finishing pass.
//
// These layers are 3 % of the FLOPs but cost 5 % of the pass on conv_direct.hip, whose lanes fetch every operand fragment from
// global memory themselves (1.25 loads per MFMA, no LDS re-use: 24 - 200 TFLOP/s).  With 64 candidates the whole population's
// conv grid is only M = 1024 .. 4096 rows, so the patch matrix A[M][9 Cin] is a 9 - 38 MB scratch buffer (written and read once,
// ~20 us of HBM time) and the product runs on gemm_tiled.hip's 128-row tiles straight from the conv weights in their packed
// [tap][n][Cin] layout (GemmParams::kpt).  The epilogue (demodulation, noise, bias, activation, residual, depth-to-space of the
// folded up-conv) is the same arithmetic as conv_direct's, applied to the fp32 product.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>
#include <string.h>

// A[m][tap][ci] = x[b][oy * stride + ty - pad][ox * stride + tx - pad][ci] * style[b][ci]   (zero outside the image)
__global__ __launch_bounds__(256) void conv_im2col_kernel(ConvParams p, half_t* A, long long n_vec) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;      // one 8-channel vector of A
    if (idx >= n_vec) return;
    const int c8n = p.Cin >> 3, taps = p.KS * p.KS;
    const int c8 = (int)(idx % c8n);
    const long long mt = idx / c8n;
    const int tap = (int)(mt % taps);
    const long long m = mt / taps;
    const int hw = p.Hc * p.Wc;
    const int b = (int)(m / hw), rem = (int)(m - (long long)b * hw);
    const int oy = rem / p.Wc, ox = rem - oy * p.Wc;
    const int ty = tap / p.KS, tx = tap - ty * p.KS;
    const int iy = oy * p.stride + ty - p.pad, ix = ox * p.stride + tx - p.pad;
    h8 a = {0, 0, 0, 0, 0, 0, 0, 0};
    if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
        // (in_up: the input is read through a nearest x2 upsample — H, W are the upsampled dims)
        a = *(const h8*)(p.x + (long long)b * p.x_bstride + ((long long)(iy >> p.in_up) * (p.W >> p.in_up) + (ix >> p.in_up)) * p.Cin + c8 * 8);
        if (p.pre_shift) {   // relu(x * s + shift) for in-bounds pixels (BigGAN batch norm + ReLU ahead of the conv): conv_direct's rounding
            const float* s = p.sn + (long long)b * p.sn_stride + c8 * 8;
            const float* t = p.pre_shift + (long long)b * p.sn_stride + c8 * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = (half_t)fmaxf((float)a[j] * s[j] + t[j], 0.f);
        } else if (p.sn) {      // (half)(x * s): conv_direct's rounding
            const float* s = p.sn + (long long)b * p.sn_stride + c8 * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = (half_t)((float)a[j] * s[j]);
        }
    }
    *(h8*)(A + idx * 8) = a;
}

// y = epilogue(C[m][n]) for 4 consecutive n per thread — conv_direct.hip's epilogue contract
__global__ __launch_bounds__(256) void conv_finish_kernel(ConvParams p, const float* C, long long n_quad, int S, long long slab) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_quad) return;
    const int nq = p.Neff >> 2;
    const int n = (int)(idx % nq) * 4;
    const long long m = idx / nq;
    const int hw = p.Hc * p.Wc;
    const int b = (int)(m / hw), rem = (int)(m - (long long)b * hw);
    int py = rem / p.Wc, px = rem - py * p.Wc, o = n;
    if (p.up) {
        const int ph = n / p.Cout;
        o = n - ph * p.Cout;
        py = 2 * py + (ph >> 1);
        px = 2 * px + (ph & 1);
    }
    f4 c = *(const f4*)(C + m * p.Neff + n);
    if (S > 1) {                               // split-K slices (S <= 4): loaded together, added in slice order
        f4 cz[3];
#pragma unroll
        for (int z = 1; z < 4; ++z) cz[z - 1] = z < S ? *(const f4*)(C + z * slab + m * p.Neff + n) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int z = 1; z < 4; ++z) c += cz[z - 1];
    }
    const float nz = p.noise ? p.noise_strength * p.noise[((long long)(b / p.batch_size) * p.Ho + py) * p.Wo + px] : 0.f;
    const long long oidx = (((long long)b * p.Ho + py) * p.Wo + px) * p.Cout + o;
    h4 r = {0, 0, 0, 0};
    if (p.res) {
        const int rcs = p.res_cs ? p.res_cs : p.Cout;
        r = *(const h4*)(p.res + (p.res_up ? (((long long)b * (p.Ho >> 1) + (py >> 1)) * (p.Wo >> 1) + (px >> 1)) * rcs + o
                                           : (((long long)b * p.Ho + py) * p.Wo + px) * rcs + o));
    }
    h4 out;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v = c[q];
        if (p.dscale) v *= p.dscale[(long long)b * p.ds_stride + o + q];
        v += nz;
        if (p.bias) v += p.bias[o + q];
        if (p.shift) v += p.shift[(long long)b * p.ds_stride + o + q];
        if (p.act == 1) v = lrelu_sqrt2(v);
        else if (p.act == 2) v = fmaxf(v, 0.f);
        if (p.res) v += (float)r[q];
        out[q] = (half_t)(v * p.out_scale);
    }
    *(h4*)(p.y + oidx) = out;
}

// cap_a / cap_c: scratch capacity PER CANDIDATE (halfs of A, floats of C); the buffers hold p.B candidates
const char* launch_conv_gemm(const ConvParams& p, half_t* ws_a, long long cap_a, float* ws_c, long long cap_c, hipStream_t st) {
    static const bool off = glass_knob("GLASS_NO_CONV_GEMM") != nullptr;   // A/B knob: these layers stay on conv_direct
    if (p.x_planar8 || p.y_planar8 || p.x_planar32) return nullptr;   // chunk-planar maps (common.h): not implemented here
    if (off || !ws_a || !ws_c || p.y32 || !p.y || p.w_bstride != 0 || p.rgb_y || p.trgb_yout || p.skip_x) return nullptr;
    if (p.pre_shift && !p.sn) return nullptr;
    if (p.xs_out || p.post_scale16) return nullptr;   // by-products / output transforms this path does not implement: refuse, never ignore
    if ((p.KS != 1 && p.KS != 3) || p.Cin % 64 != 0 || p.Neff % 64 != 0 || (p.Cout & 3) || (p.res_cs & 3)) return nullptr;
    const long long M = (long long)p.B * p.Hc * p.Wc, K = (long long)p.KS * p.KS * p.Cin;
    if ((long long)p.Hc * p.Wc * K > cap_a || (long long)p.Hc * p.Wc * p.Neff > cap_c || M * K >= (1LL << 31)) return nullptr;
    const long long n_vec = M * K / 8, n_quad = M * p.Neff / 4;
    // a 1x1 convolution of an un-modulated, contiguous map IS its patch matrix (the D blocks' skip branches: the im2col pass was a copy);
    // with nothing to do after the product either (no demodulation / noise / bias / activation / residual / gain) the GEMM's own fp16
    // store (mode 0: the same (half_t) rounding of the same fp32 sum) is the finishing pass: one launch instead of three (round 6)
    const bool direct_a = p.KS == 1 && p.stride == 1 && p.pad == 0 && !p.sn && !p.pre_shift && !p.in_up && p.H == p.Hc && p.W == p.Wc &&
                          p.x_bstride == (long long)p.Hc * p.Wc * p.Cin;
    const bool direct_y = direct_a && !p.up && !p.dscale && !p.noise && !p.bias && !p.shift && p.act == 0 && !p.res && p.out_scale == 1.f &&
                          p.Ho == p.Hc && p.Wo == p.Wc && p.Neff == p.Cout;
    if (p.dry_run) return direct_y ? "gemm_tiled_kernel" : "conv_gemm(im2col+gemm_tiled+finish)";
    // no activation-side transform of the input (the D blocks' convolutions): the GEMM's loader walks the map itself (GemmParams::g_*,
    // gemm_tiled_kernel<.., gather>) — the patch matrix, a 9x copy of the map written to HBM and read back, is never materialised (round 6)
    static const bool no_gather = glass_knob("GLASS_CONV_GEMM_NO_GATHER") != nullptr;   // A/B knob
    bool gather = !no_gather && !direct_a && !p.sn && !p.pre_shift && !p.in_up && p.Cin % 64 == 0;
    GemmParams g;
    memset(&g, 0, sizeof g);
    auto set_gather = [&](bool on) {
        g.g_on = on ? 1 : 0;
        g.a = on || direct_a ? p.x : ws_a;
        if (on) { g.g_h = p.H; g.g_w = p.W; g.g_hc = p.Hc; g.g_wc = p.Wc; g.g_stride = p.stride; g.g_pad = p.pad; g.g_ks = p.KS; g.g_cin = p.Cin; g.g_xbs = p.x_bstride; }
    };
    set_gather(gather);
    g.w = p.w; g.M = (int)M; g.N = p.Neff; g.K = (int)K;
    if (direct_y) {
        g.kpt = p.Cin; g.w_tap_stride = (long long)p.Neff * p.Cin;
        g.mode = 0; g.out16 = p.y; g.ldo = p.Cout;
        g.cand_rows = p.Hc * p.Wc;
        if (const char* k = launch_gemm_tiled(g, st)) return k;
    }
    g.kpt = p.Cin; g.w_tap_stride = (long long)p.Neff * p.Cin;        // weights stay [tap][n][Cin]; (kpt: any M is accepted —
                                                                      // the kernel choice must not depend on the candidate count)
    g.mode = 3; g.out32 = ws_c; g.ldo = p.Neff;
    g.cand_rows = p.Hc * p.Wc;
    // split-K (round 3): a 4 x 4 / 8 x 8 grid per candidate is 16 / 64 rows — at 64 candidates the product has 64 / 256 tiles walking
    // 72 K steps each.  S slices (a function of the per-candidate geometry only, like every choice here) of raw sums, added in a fixed
    // order by the finishing pass; the scratch holds them while S x grid x Neff fits its per-candidate capacity.
    static const bool no_split = glass_knob("GLASS_CONV_GEMM_NO_SPLIT") != nullptr;   // A/B knob
    int S = 1;
    if (!no_split && p.KS == 3) {
        const int px = p.Hc * p.Wc;
        S = px <= 16 ? 4 : (px <= 64 ? 2 : 1);
        while (S > 1 && (K % (64LL * S) != 0 || (long long)S * px * p.Neff > cap_c)) S >>= 1;
    }
    const long long slab = M * p.Neff;
    if (S > 1) { g.ld = (int)K; g.K = (int)(K / S); g.batch = S; g.a_bs = gather ? 0 : g.K; g.w_bs = 0; g.o_bs = slab; }
    const char* gk = nullptr;
    if (gather) {
        gk = launch_gemm_tiled(g, st);
        if (!gk) {                             // the gather instance refused: materialise the patch matrix after all
            gather = false;
            set_gather(false);
            if (S > 1) g.a_bs = g.K;
        }
    }
    if (!gk) {
        if (!direct_a) hipLaunchKernelGGL(conv_im2col_kernel, dim3((unsigned)((n_vec + 255) / 256)), dim3(256), 0, st, p, ws_a, n_vec);
        gk = launch_gemm_tiled(g, st);
    }
    if (!gk) {
        if (S > 1) { S = 1; g.ld = 0; g.K = (int)K; g.batch = 0; g.a_bs = g.o_bs = 0; }
        if (!launch_gemm_tiled(g, st)) launch_gemm_direct(g, st);
    }
    hipLaunchKernelGGL(conv_finish_kernel, dim3((unsigned)((n_quad + 255) / 256)), dim3(256), 0, st, p, ws_c, n_quad, S, slab);
    return gather ? "conv_gemm(gather+gemm_tiled+finish)" : "conv_gemm(im2col+gemm_tiled+finish)";
}
