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

    // Loads are UNCONDITIONAL (out-of-image / out-of-range vectors read a valid address instead) and masked when they are
    // written to LDS, on border tiles only: a conditional load costs a saveexec + branch + zero-fill each, every stage.
    int a_goff[NA];
    int okm = 0;                                 // bit k: vector k is inside the image
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int v = t + 256 * k;
        const int pix = v >> 2;
        const int pr = pix / PW, pc = pix - pr * PW;
        const int iy = my0 - 1 + pr, ix = mx0 - 1 + pc;
        const bool ok = (v < NVA) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        a_goff[k] = ok ? (iy * p.W + ix) * p.Cin + part * 8 : part * 8;
        okm |= (ok ? 1 : 0) << k;
    }
    const bool border = my0 < 1 || mx0 < 1 || my0 + 7 >= p.H || mx0 + 31 >= p.W;     // uniform
    const half_t* xb = p.x + (long long)b * p.x_bstride;
    const half_t* wb = p.w_up + (long long)b * p.w_bstride;
    const half_t* snb = p.sn16 ? p.sn16 + (long long)b * p.sn_stride + part * 8 : nullptr;
    long long b_goff[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int u = min(t + 256 * k, NVB - 1);
        b_goff[k] = ((long long)(u >> 7) * p.Cout + n0 + ((u >> 2) & 31)) * p.Cin + part * 8;
    }

    h8 ra[NA], rb[NB];
    h8 sh;   // style of this thread's 8 channels of the current chunk (fp16: packed multiply at staging)
#pragma unroll
    for (int j = 0; j < 8; ++j) sh[j] = (half_t)1.f;
    auto load_a = [&](int c0) {
#pragma unroll
        for (int k = 0; k < NA; ++k) ra[k] = *(const h8*)(xb + a_goff[k] + c0);
        if (p.sn16) sh = *(const h8*)(snb + c0);
    };
    auto load_b = [&](int c0) {
#pragma unroll
        for (int k = 0; k < NB; ++k) rb[k] = *(const h8*)(wb + b_goff[k] + c0);
    };
    auto store_a = [&]() {
        if (!border && !p.sn16) {                // interior tile of a layer whose weights carry the style: registers -> LDS
#pragma unroll
            for (int k = 0; k < NA; ++k) {
                const int v = t + 256 * k;
                if (k < NA - 1 || v < NVA) *(h8*)(As + (v >> 2) * ROWB + part * 16) = ra[k];
            }
        } else {
            const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < NA; ++k) {
                const int v = t + 256 * k;
                if (k < NA - 1 || v < NVA) {
                    h8 a = ((okm >> k) & 1) ? ra[k] : zero;
                    *(h8*)(As + (v >> 2) * ROWB + part * 16) = a * sh;   // 4 x v_pk_mul_f16 (1.0 without a style)
                }
            }
        }
    };
    auto store_b = [&]() {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int u = t + 256 * k;
            if (k < NB - 1 || u < NVB) *(h8*)(Bs + (u >> 2) * ROWB + part * 16) = rb[k];
        }
    };

    f16x acc[2][4];   // [m-row of this wave][parity class ry*2+rx]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // tap (ky,kx) feeds parity class (ky&1, kx&1) and reads x[m - (ky>>1), n - (kx>>1)]: the 9 taps
    // use only 4 distinct input shifts, so each x fragment is loaded once per shift and re-used
    // by every tap of that shift (17 LDS fragment reads per 18 MFMAs instead of 27).
    auto mfma_block = [&]() {
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
    };
    const int n_stages = p.Cin >> 5;
    load_a(0);
    load_b(0);
    for (int s = 0; s + 1 < n_stages; ++s) {
        if (s > 0) __syncthreads();
        store_a();
        store_b();
        __syncthreads();
        load_a((s + 1) * 32);
        load_b((s + 1) * 32);
        mfma_block();
    }
    // last stage, peeled: in place of a next stage's operands, everything the epilogue needs from global memory is fetched
    // HERE (4 demod quads, 2 bias quads, this thread's 12 noise values), unconditionally and in one batch, so that it lands
    // under the last MFMA block instead of costing the epilogue a round trip of its own (~1.5 us of a ~7 us tile on the
    // two-stage 1024^2 layer).  Peeled because as loop-carried values these 36 registers would be live through every stage.
    if (n_stages > 1) __syncthreads();
    store_a();
    store_b();
    __syncthreads();
    const int cg = t & 3, oxl = t >> 2;              // FIR phase: 8-channel group, local output column 0..59 (t < 240)
    const int px = min(txi * 60 + oxl, p.Wo - 1);
    f4 dq[4];
    if (p.dscale) {
#pragma unroll
        for (int g = 0; g < 4; ++g) dq[g] = *(const f4*)(p.dscale + (long long)b * p.ds_stride + n0 + 8 * g + 4 * kh);
    }
    f4 bq0 = {0.f, 0.f, 0.f, 0.f}, bq1 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        bq0 = *(const f4*)(p.bias + n0 + cg * 8);
        bq1 = *(const f4*)(p.bias + n0 + cg * 8 + 4);
    }
    h8 ps8;                                           // the consumer's style for this thread's 8 channels (or 1)
#pragma unroll
    for (int j = 0; j < 8; ++j) ps8[j] = (half_t)1.f;
    if (p.post_scale16) ps8 = *(const h8*)(p.post_scale16 + (long long)b * p.post_stride + n0 + cg * 8);
    float nzv[12];
#pragma unroll
    for (int r = 0; r < 12; ++r) nzv[r] = 0.f;
    if (p.noise) {
        const float* nzp = p.noise + (long long)(b / p.batch_size) * p.Ho * p.Wo + px;
#pragma unroll
        for (int r = 0; r < 12; ++r) nzv[r] = nzp[(long long)min(tyi * 12 + r, p.Ho - 1) * p.Wo];
    }
    mfma_block();
    __syncthreads();   // everyone is done with the staging area: overlay T

    // ---- t tile -> LDS (demod applied; it commutes with the FIR) ---------------------------------
    // Pixel (lty, ltx) is a 64-byte row of four 16-byte channel pairs (8g .. 8g+7: the quads of lane halves kh = 0 | 1);
    // the pair slot is XOR-swizzled by (ltx >> 1) & 3 (4-way instead of 16-way conflicts for the 8-byte writes, and the
    // FIR's 16-byte reads need no fix-up).  Round 2 instruction diet: the epilogue was ~2500 instructions per thread per
    // tile (a phase trace: 17000 cycles, more than the K loop of every layer below 256 input channels) — 16 v_cndmask per
    // FIR row to un-swap an 8-byte swizzle, an element-wise activation, 64-bit store addressing per row.
    {
        char* tw = (char*)T + ((2 * wave * 2) * 64 + 2 * lr) * 64 + kh * 8;
        int so[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) so[g] = (g ^ (lr & 3)) * 16;
        auto put = [&](bool scaled) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        h4 o;
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = (half_t)(scaled ? acc[i][ph][g * 4 + q] * dq[g][q] : acc[i][ph][g * 4 + q]);
                        *(h4*)(tw + so[g] + ((2 * i + (ph >> 1)) * 64 + (ph & 1)) * 64) = o;
                    }
        };
        if (p.dscale) put(true);
        else put(false);          // weights carry the demodulation already
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
    // activation as max(v * k1, v * k2): lrelu * sqrt2 * scale -> (sqrt2 s, 0.2 sqrt2 s); none -> (s, s)
    const half_t k1 = (half_t)((p.act ? GLASS_SQRT2 : 1.f) * p.out_scale), k2 = (half_t)((p.act ? 0.2f * GLASS_SQRT2 : 1.f) * p.out_scale);
    const char* tr[4];
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) {
        const int ltx = oxl + 1 + jx;
        tr[jx] = (const char*)T + ltx * 64 + ((cg ^ ((ltx >> 1) & 3)) * 16);
    }
    const long long rowpitch = (long long)p.Wo * p.Cout;
    half_t* yp = p.y + (((long long)b * p.Ho + tyi * 12) * p.Wo + px) * p.Cout + n0 + cg * 8;
    h8 hs[4];
#pragma unroll
    for (int r = 1; r < 16; ++r) {
        const h8 v0 = *(const h8*)(tr[0] + r * 4096), v1 = *(const h8*)(tr[1] + r * 4096), v2 = *(const h8*)(tr[2] + r * 4096),
                 v3 = *(const h8*)(tr[3] + r * 4096);
        hs[r & 3] = (v0 + v3) * fq + (v1 + v2) * ft;
        if (r >= 4) {
            if (tyi * 12 + (r - 4) < p.Ho) {
                const h8 bn = bias8 + (half_t)(p.noise_strength * nzv[r - 4]);
                h8 v = (hs[(r - 3) & 3] + hs[r & 3]) * fq + ((hs[(r - 2) & 3] + hs[(r - 1) & 3]) * ft + bn);
                *(h8*)yp = __builtin_elementwise_max(v * k1, v * k2) * ps8;
            }
            yp += rowpitch;
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
    if (p.dry_run) return "upfir_kernel";
    hipLaunchKernelGGL(upfir_kernel, dim3(PT8 * NTn), dim3(256), LDS, st, p, NTn, tiles_x, tiles_y, PT);
    return "upfir_kernel";
}
