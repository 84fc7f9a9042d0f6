// gemm_tiled.hip — LDS-tiled MFMA GEMM  out[M][N] = A[M][K] * W[N][K]^T (+bias, epilogue)
// for the CLIP ViT linears / patch embedding (clip/model.py:166-235) and the D dense head.
// 128 x BN block tile, 4 waves as 2(M) x 2(N), 64-deep K stages staged through LDS with a
// register-prefetch pipeline; rows are 144 B (128 B data + 16 B pad) -> conflict-free
// ds_read_b128 fragment reads.  Operands swapped (A-role = W rows) so each lane owns 4
// consecutive output columns of one output row per accumulator quad (16/8-byte stores).
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

#define GROWB 144

// GATHER (round 6): the A operand is the IMPLICIT patch matrix of a convolution (GemmParams::g_*): the loader walks x itself, the im2col
// pass of conv_gemm.hip (a 9x copy of the map to HBM and back) disappears for the layers without an activation-side style
template <int BN, bool GATHER = false>
__global__ __launch_bounds__(256, 2) void gemm_tiled_kernel(GemmParams p) {
    if (p.batch > 1) {   // batched problems (BigGAN self-attention): one z-slice per problem
        p.a += (long long)blockIdx.z * p.a_bs;
        p.w += (long long)blockIdx.z * p.w_bs;
        if (p.out16) p.out16 += (long long)blockIdx.z * p.o_bs;
        if (p.out32) p.out32 += (long long)blockIdx.z * p.o_bs;
    }
    constexpr int NJ = BN / 64;                  // 32-wide n tiles per wave
    constexpr int NVA = 128 * 8, NVB = BN * 8;   // 16-byte vectors per stage
    constexpr int NA = NVA / 256, NB = NVB / 256;
    __shared__ __attribute__((aligned(16))) char smem[(128 + BN) * GROWB];
    char* As = smem;
    char* Bs = smem + 128 * GROWB;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, kh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile walk (round 6).  Workgroup i of a launch runs on XCD i % 8, each XCD has its own 4 MB L2: with the (m, n) tiles dealt
    // x-fastest every XCD saw every panel of both operands (CLIP's linears, M = 3200: 25 x 18 tiles of 2 x 196 KB panels each — PMC traffic
    // 6.1 x the algorithmic bytes, the kernel bound by the fabric).  Now XCD x owns a CONTIGUOUS slice of the m-major tile list: ~3 of the 25
    // activation panels (kept) and one pass over the weights.  Same tiles, same arithmetic: bit-identical outputs.
    const int gy = p.N / BN, n_tiles = ((p.M + 127) >> 7) * gy;
    const int per_xcd = (n_tiles + 7) >> 3;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= min(((int)(blockIdx.x & 7) + 1) * per_xcd, n_tiles)) return;      // (padding of the last slices; uniform)
    const int m0 = (tile / gy) * 128, n0 = (tile % gy) * BN;
    const int part = t & 7;                      // which 16-byte piece of the 128-byte row chunk

    h8 ra[NA], rb[NB];
    long long gbase[GATHER ? NA : 1];            // GATHER: this thread's rows of the conv grid — element offset of tap (0, 0), its input row / column
    int giy[GATHER ? NA : 1], gix[GATHER ? NA : 1], okm = 0;   // okm bit k: vector k of the stage in registers lies inside the image
    if (GATHER) {
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int row = min(m0 + (t >> 3) + 32 * k, p.M - 1), hw = p.g_hc * p.g_wc;
            const int b = row / hw, rem = row - b * hw, oy = rem / p.g_wc, ox = rem - oy * p.g_wc;
            giy[k] = oy * p.g_stride - p.g_pad;
            gix[k] = ox * p.g_stride - p.g_pad;
            gbase[k] = (long long)b * p.g_xbs + ((long long)giy[k] * p.g_w + gix[k]) * p.g_cin + part * 8;
        }
    }
    auto load = [&](int k0) {
        const int kw = k0 + ((p.kpt && p.ld) ? (int)blockIdx.z * p.K : 0);   // split-K over a packed conv weight: this slice's first k
        if (GATHER) {
            const int tap = kw / p.g_cin, ci = kw - tap * p.g_cin, ty = tap / p.g_ks, tx = tap - ty * p.g_ks;      // uniform
            const long long toff = ((long long)ty * p.g_w + tx) * p.g_cin + ci;
            okm = 0;
#pragma unroll
            for (int k = 0; k < NA; ++k) {     // unconditional loads at a valid address; the zero padding is applied when the vector goes to LDS
                const bool ok = (unsigned)(giy[k] + ty) < (unsigned)p.g_h && (unsigned)(gix[k] + tx) < (unsigned)p.g_w;
                ra[k] = *(const h8*)(p.a + (ok ? gbase[k] + toff : (long long)part * 8));
                okm |= (ok ? 1 : 0) << k;
            }
        } else {
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            // unconditional (rows past M read row M - 1; their outputs are never stored): a conditional load costs a
            // saveexec + branch + zero-fill per vector per stage
            const int row = min(m0 + (t >> 3) + 32 * k, p.M - 1);
            ra[k] = *(const h8*)(p.a + (long long)row * (p.ld ? p.ld : p.K) + k0 + part * 8);
        }
        }
        const int rs = p.kpt ? p.kpt : (p.ld ? p.ld : p.K);      // weight row stride; a conv weight is walked tap by tap
        const half_t* wk = p.kpt ? p.w + (long long)(kw / p.kpt) * p.w_tap_stride + (kw % p.kpt) : p.w + k0;   // uniform
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int row = (t >> 3) + 32 * k;
            rb[k] = *(const h8*)(wk + (long long)(n0 + row) * rs + part * 8);
        }
    };
    auto store = [&]() {
        const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < NA; ++k) *(h8*)(As + ((t >> 3) + 32 * k) * GROWB + part * 16) = (!GATHER || ((okm >> k) & 1)) ? ra[k] : zero;
#pragma unroll
        for (int k = 0; k < NB; ++k) *(h8*)(Bs + ((t >> 3) + 32 * k) * GROWB + part * 16) = rb[k];
    };

    f16x acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    load(0);
    for (int k0 = 0; k0 < p.K; k0 += 64) {
        if (k0 > 0) __syncthreads();
        store();
        __syncthreads();
        if (k0 + 64 < p.K) load(k0 + 64);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            h8 wf[NJ], xf[2];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                wf[j] = *(const h8*)(Bs + (wn * (BN / 2) + j * 32 + lr) * GROWB + kk * 32 + kh * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                xf[i] = *(const h8*)(As + (wm * 64 + i * 32 + lr) * GROWB + kk * 32 + kh * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(wf[j], xf[i], acc[i][j]);
        }
    }
    // epilogue: lane = output row m, quads of 4 consecutive n.  Bias quads and (mode 2) the residual quads of a row are
    // fetched as one batch of unconditional loads: a load consumed right after it is issued costs a full round trip, and
    // there were two of them per accumulator quad here.
    f4 bq[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bq[j][g] = f4{0.f, 0.f, 0.f, 0.f};
            if (p.bias) bq[j][g] = *(const f4*)(p.bias + n0 + wn * (BN / 2) + j * 32 + 8 * g + 4 * kh);
        }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + lr;
        const int mc = min(m, p.M - 1);                       // clamped row for the unconditional residual loads
        f4 rq[NJ][4];
        if (p.mode == 2) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    rq[j][g] = *(const f4*)(p.out32 + (long long)mc * p.ldo + n0 + wn * (BN / 2) + j * 32 + 8 * g + 4 * kh);
        }
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * (BN / 2) + j * 32 + 8 * g + 4 * kh;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = acc[i][j][g * 4 + q] + bq[j][g][q];
                const long long oi = (long long)m * p.ldo + n;
                if (p.mode <= 1) {
                    h4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        o[q] = (half_t)(p.mode == 1 ? v[q] * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v[q])) : v[q]);   // QuickGELU; v_rcp_f32 (1 ulp) instead of the ~15-instruction IEEE division: the result is rounded to fp16
                    *(h4*)(p.out16 + oi) = o;
                } else {
                    f4 o;
                    if (p.mode == 2) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = rq[j][g][q] + v[q];
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = p.mode == 4 ? lrelu_sqrt2(v[q]) : v[q];
                    }
                    *(f4*)(p.out32 + oi) = o;
                }
            }
        }
    }
}

const char* launch_gemm_tiled(const GemmParams& p, hipStream_t st) {
    if (p.K % 64 != 0 || p.ldo % 4 != 0 || (p.M < 64 && !p.kpt) || (p.kpt && p.kpt % 64 != 0)) return nullptr;
    const unsigned gx = (unsigned)((p.M + 127) / 128), gz = p.batch > 1 ? p.batch : 1;
    // 128-wide n tiles under-fill the chip on the N = 768 CLIP linears (25 x 6 = 150 workgroups for 256 CUs): take 64-wide
    // tiles whenever the 128-wide grid has fewer workgroups than CUs
    static const bool wide_only = glass_knob("GLASS_GEMM_BN128") != nullptr;   // A/B knob
    // (evaluated at the nominal population where the caller says how M / the batch scale with it — the instances are bit-identical
    // per output element, the rule just never looks at the launch size)
    const long long gx_n = p.cand_rows ? ((long long)p.cand_rows * GLASS_NOMINAL_POP + 127) / 128 : gx;
    const long long gz_n = p.cand_batch ? GLASS_NOMINAL_POP : gz;
    const bool fills = gx_n * (p.N / 128) * gz_n >= 256;
    if (p.g_on && (!p.kpt || p.kpt != p.g_cin || p.g_cin % 64 != 0)) return nullptr;      // (a K step must sit inside one tap)
    if (p.N % 128 == 0 && (fills || wide_only)) {
        if (p.g_on) {
            hipLaunchKernelGGL((gemm_tiled_kernel<128, true>), dim3(8 * ((gx * (p.N / 128) + 7) / 8), 1, gz), dim3(256), 0, st, p);
            return "gemm_tiled_kernel<128,true>";
        }
        hipLaunchKernelGGL(gemm_tiled_kernel<128>, dim3(8 * ((gx * (p.N / 128) + 7) / 8), 1, gz), dim3(256), 0, st, p);
        return "gemm_tiled_kernel<128>";
    }
    if (p.N % 64 == 0) {
        if (p.g_on) {
            hipLaunchKernelGGL((gemm_tiled_kernel<64, true>), dim3(8 * ((gx * (p.N / 64) + 7) / 8), 1, gz), dim3(256), 0, st, p);
            return "gemm_tiled_kernel<64,true>";
        }
        hipLaunchKernelGGL(gemm_tiled_kernel<64>, dim3(8 * ((gx * (p.N / 64) + 7) / 8), 1, gz), dim3(256), 0, st, p);
        return "gemm_tiled_kernel<64>";
    }
    return nullptr;
}
