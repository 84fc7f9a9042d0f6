// upfir.hip — fused minimum-FLOP up-convolution for gfx950:
//   conv_transpose2d(stride 2, 3x3)  ->  4x4 FIR (gain 4, pad 1)  ->  demod/noise/bias/lrelu
// (stylegan2/modules.py:1089-1139, 414-453, 276-297) in ONE kernel.
//
// The folded formulation (conv_tiled.hip with Neff = 4*Cout) spends 36*I*O MACs per input
// pixel; the transposed convolution itself needs only 9*I*O.  Here the transposed conv is
// computed exactly once on the matrix cores, split by output parity
//     t[2m+r] += x[m - (k>>1)] * w[k],   r = k & 1          (per axis; k = tap index)
// i.e. 4 parity classes with 4 / 2 / 2 / 1 taps, each tap feeding ONE accumulator class.
// The 16 x 64 tile of t (32 channels) then goes to LDS (fp16, demod already applied — it
// commutes with the FIR) and the FIR + epilogue runs from LDS with a sliding-window
// separable filter, producing a 12 x 60 output tile.  Tiles advance by 6 x 30 input
// positions (8 x 32 are computed: the 1-pixel t halo the FIR needs is recomputed, 70%
// MFMA efficiency -> ~12.8 I*O MACs per input pixel instead of 36).
//
// Block = 4 waves; wave w owns m-rows {2w, 2w+1} of the tile; acc[row][parity class].
// Staging / pipeline identical to conv_tiled.hip (patch once per 32-channel chunk, weight
// slice per tap-row stage, register prefetch).
#include "common.h"
#include "kernels.h"

#define ROWB 80

__global__ __launch_bounds__(256, 2) void upfir_kernel(ConvParams p, int NTn, int tiles_x, int tiles_y, int PT) {
    constexpr int PH = 9, PW = 33;              // x patch: rows my0-1 .. my0+7, cols mx0-1 .. mx0+31
    constexpr int NVA = PH * PW * 4, NA = (NVA + 255) / 256;   // 1188 -> 5
    constexpr int NVB = 9 * 32 * 4, NB = (NVB + 255) / 256;    // 1152 -> 5 (all 9 taps of a chunk)
    constexpr int A_BYTES = ((PH * PW * ROWB + 15) / 16) * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + A_BYTES;
    half_t* T = (half_t*)smem;                  // [16][64][32] fp16, overlays the staging area afterwards

    const int id = blockIdx.x;
    const int lo = id & 7, rest = id >> 3;
    const int nt = rest % NTn, pt = (rest / NTn) * 8 + lo;
    if (pt >= PT) return;
    const int tpi = tiles_x * tiles_y;
    const int b = pt / tpi;
    const int trem = pt - b * tpi;
    const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
    const int my0 = tyi * 6 - 1, mx0 = txi * 30 - 1;
    const int n0 = nt * 32;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, kh = lane >> 5;
    const int part = t & 3;

    int a_goff[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int v = t + 256 * k;
        const int pix = v >> 2;
        const int pr = pix / PW, pc = pix - pr * PW;
        const int iy = my0 - 1 + pr, ix = mx0 - 1 + pc;
        const bool ok = (v < NVA) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        a_goff[k] = ok ? (iy * p.W + ix) * p.Cin + part * 8 : -1;
    }
    const half_t* xb = p.x + (long long)b * p.x_bstride;
    const half_t* wb = p.w_up + (long long)b * p.w_bstride;
    const half_t* snb = p.sn16 ? p.sn16 + (long long)b * p.sn_stride + part * 8 : nullptr;

    h8 ra[NA], rb[NB];
    h8 sh;   // style of this thread's 8 channels of the current chunk (fp16: packed multiply at staging)
#pragma unroll
    for (int j = 0; j < 8; ++j) sh[j] = (half_t)1.f;
    auto load_a = [&](int c0) {
#pragma unroll
        for (int k = 0; k < NA; ++k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ra[k][j] = (half_t)0.f;
            if (a_goff[k] >= 0) ra[k] = *(const h8*)(xb + a_goff[k] + c0);
        }
        if (snb) sh = *(const h8*)(snb + c0);
    };
    auto load_b = [&](int c0) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int u = t + 256 * k;
            if (u < NVB) {
                const int tap = u >> 7;           // / (32 * 4)
                const int n = (u >> 2) & 31;
                rb[k] = *(const h8*)(wb + ((long long)tap * p.Cout + n0 + n) * p.Cin + c0 + part * 8);
            }
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int v = t + 256 * k;
            if (v < NVA) {
                h8 a = ra[k];
                if (snb) a = a * sh;   // 4 x v_pk_mul_f16
                *(h8*)(As + (v >> 2) * ROWB + part * 16) = a;
            }
        }
    };
    auto store_b = [&]() {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int u = t + 256 * k;
            if (u < NVB) *(h8*)(Bs + (u >> 2) * ROWB + part * 16) = rb[k];
        }
    };

    f16x acc[2][4];   // [m-row of this wave][parity class ry*2+rx]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    const int n_stages = p.Cin >> 5;
    load_a(0);
    load_b(0);
    for (int s = 0; s < n_stages; ++s) {
        if (s > 0) __syncthreads();
        store_a();
        store_b();
        __syncthreads();
        if (s + 1 < n_stages) {
            load_a((s + 1) * 32);
            load_b((s + 1) * 32);
        }
        // tap (ky,kx) feeds parity class (ky&1, kx&1) and reads x[m - (ky>>1), n - (kx>>1)]: the 9 taps
        // use only 4 distinct input shifts, so each x fragment is loaded once per shift and re-used
        // by every tap of that shift (17 LDS fragment reads per 18 MFMAs instead of 27).
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int ay = 0; ay < 2; ++ay) {
#pragma unroll
                for (int ax = 0; ax < 2; ++ax) {
                    h8 wf[2][2];   // taps of this shift: ky in {2ay, 2ay+1 (if ay == 0)}, kx likewise
#pragma unroll
                    for (int ky = ay * 2; ky < (ay ? 3 : 2); ++ky)
#pragma unroll
                        for (int kx = ax * 2; kx < (ax ? 3 : 2); ++kx)
                            wf[ky & 1][kx & 1] = *(const h8*)(Bs + ((ky * 3 + kx) * 32 + lr) * ROWB + kk * 32 + kh * 16);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int prow = wave * 2 + i + 1 - ay;
                        const h8 xf = *(const h8*)(As + (prow * PW + lr + 1 - ax) * ROWB + kk * 32 + kh * 16);
#pragma unroll
                        for (int ky = ay * 2; ky < (ay ? 3 : 2); ++ky)
#pragma unroll
                            for (int kx = ax * 2; kx < (ax ? 3 : 2); ++kx)
                                acc[i][(ky & 1) * 2 + (kx & 1)] = mfma32(wf[ky & 1][kx & 1], xf, acc[i][(ky & 1) * 2 + (kx & 1)]);
                    }
                }
            }
        }
    }
    __syncthreads();   // everyone is done with the staging area: overlay T

    // Everything the epilogue needs from global memory is fetched HERE, unconditionally and in one batch (a load
    // under a branch, consumed at once, costs a full round trip each: 32 serial demod loads + 12 serial noise
    // loads per block before): 4 demod quads, 2 bias quads, this thread's 12 noise values.
    const int cg = t & 3, oxl = t >> 2;              // FIR phase: 8-channel group, local output column 0..59 (t < 240)
    const int px = min(txi * 60 + oxl, p.Wo - 1);
    f4 dq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        dq[g] = f4{1.f, 1.f, 1.f, 1.f};
        if (p.dscale) dq[g] = *(const f4*)(p.dscale + (long long)b * p.ds_stride + n0 + 8 * g + 4 * kh);
    }
    f4 bq0 = {0.f, 0.f, 0.f, 0.f}, bq1 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        bq0 = *(const f4*)(p.bias + n0 + cg * 8);
        bq1 = *(const f4*)(p.bias + n0 + cg * 8 + 4);
    }
    float nzv[12];
    if (p.noise) {
        const float* nzp = p.noise + (long long)(b / p.batch_size) * p.Ho * p.Wo + px;
#pragma unroll
        for (int r = 0; r < 12; ++r) nzv[r] = nzp[(long long)min(tyi * 12 + r, p.Ho - 1) * p.Wo];
    }

    // ---- t tile -> LDS (demod applied; it commutes with the FIR) ---------------------------------
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int lty = 2 * (wave * 2 + i) + (ph >> 1), ltx = 2 * lr + (ph & 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f4 d = dq[g];
                h4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (half_t)(acc[i][ph][g * 4 + q] * d[q]);
                // quad position XOR-swizzled by the pixel (2-way instead of 16-way write conflicts)
                const int qp = (2 * g + kh) ^ (lr & 7);
                *(h4*)(T + ((lty * 64 + ltx) * 32 + qp * 4)) = o;
            }
        }
    }
    __syncthreads();

    // ---- FIR (separable [1,3,3,1]/4 per axis, sliding window) + noise + bias + lrelu -----------
    // Packed-fp16 arithmetic (v_pk_fma_f16): the t tile is fp16 already; 4+4 taps with weights
    // {1/4,3/4} add ~2 fp16 roundings per output — same class as the fp16 activation store.
    if (t >= 240 || txi * 60 + oxl >= p.Wo) return;
    const half_t fq = (half_t)0.25f, ft = (half_t)0.75f;
    h8 bias8;
#pragma unroll
    for (int j = 0; j < 4; ++j) { bias8[j] = (half_t)bq0[j]; bias8[j + 4] = (half_t)bq1[j]; }
    const half_t k1 = (half_t)(GLASS_SQRT2 * p.out_scale), k2 = (half_t)(0.2f * GLASS_SQRT2 * p.out_scale);
    h8 hs[4];
#pragma unroll
    for (int r = 1; r < 16; ++r) {
        h8 hv[4];
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
            const int ltx = oxl + 1 + jx;
            const int sw = (ltx >> 1) & 7;           // writer's lr & 7 for this t column
            const h8 v = *(const h8*)(T + ((r * 64 + ltx) * 32 + ((cg ^ (sw >> 1)) * 8)));
            if (sw & 1) {                            // the two quads of the pair sit swapped
#pragma unroll
                for (int j = 0; j < 4; ++j) { hv[jx][j] = v[j + 4]; hv[jx][j + 4] = v[j]; }
            } else {
                hv[jx] = v;
            }
        }
        hs[r & 3] = (hv[0] + hv[3]) * fq + (hv[1] + hv[2]) * ft;
        if (r >= 4) {
            const int py = tyi * 12 + (r - 4);
            if (py < p.Ho) {
                h8 v = (hs[(r - 3) & 3] + hs[r & 3]) * fq + (hs[(r - 2) & 3] + hs[(r - 1) & 3]) * ft;
                half_t nv = (half_t)0.f;
                if (p.noise) nv = (half_t)(p.noise_strength * nzv[r - 4]);
                v = v + bias8 + nv;
                if (p.act) {
                    const h8 a = v * k1, c2 = v * k2;     // lrelu(v)*sqrt2*scale = max(v*k1, v*k2) for k1 > k2 > 0
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = a[j] > c2[j] ? a[j] : c2[j];
                } else {
                    v = v * (half_t)p.out_scale;
                }
                *(h8*)(p.y + (((long long)b * p.Ho + py) * p.Wo + px) * p.Cout + n0 + cg * 8) = v;
            }
        }
    }
}

const char* launch_upconv_fused(const ConvParams& p, hipStream_t st) {
    if (!p.up || !p.w_up || p.y32 || !p.y || p.res || (p.sn && !p.sn16)) return nullptr;
    if (p.Cin % 32 != 0 || p.Cout % 32 != 0 || p.W < 16 || p.KS != 3) return nullptr;
    if (p.x_bstride == 0 && p.B > 1) return nullptr;
    if ((long long)p.H * p.W * p.Cin >= (1LL << 31)) return nullptr;
    constexpr int LDS = 64 * 1024;  // T tile (16*64*32*2 B); staging (31.5 KB) lives inside it
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)upfir_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr = true;
    }
    const int tiles_y = (p.Ho + 11) / 12, tiles_x = (p.Wo + 59) / 60;
    const int PT = p.B * tiles_x * tiles_y;
    const int NTn = p.Cout / 32;
    const int PT8 = (PT + 7) / 8 * 8;
    hipLaunchKernelGGL(upfir_kernel, dim3(PT8 * NTn), dim3(256), LDS, st, p, NTn, tiles_x, tiles_y, PT);
    return "upfir_kernel";
}
