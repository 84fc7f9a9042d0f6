// conv_stream.hip — 3x3 stride-1 convolution, Cin = Cout = 32, for the HBM-bound 1024^2 layers
// (G conv_blocks.8.conv_block.1 and D conv_blocks.0.conv_block.0: 64 MB in + 64 MB out per candidate for 9.7 GFLOP).
//
// conv_tiled.hip serves these layers at ~2.6-3.1 TB/s although the same tile access pattern with no compute
// streams at 5.7 TB/s (tools/tile_copy.hip): one tile per workgroup leaves ~70 KB of loads in flight per CU
// (4 resident blocks x 24 KB patch, only while a block is in its load phase), and at the loaded HBM latency
// (~10 us) that is what caps the rate.  This kernel keeps the bytes in flight instead:
//   * persistent workgroups (2 per CU), each walking a contiguous range of TH x 32 tiles;
//   * the input patches of the NEXT THREE tiles are always in flight in registers (3 x 24 KB per block,
//     ~145 KB per CU — the flat-copy level), refilled right after a register set is written to LDS;
//   * the whole 3x3x32x32 weight tensor lives in LDS for the block's lifetime (re-staged only when the sample,
//     i.e. the pre-modulated weight set, changes): no weight traffic and no barriers inside a tile's 36 MFMAs;
//   * epilogue as in conv_tiled (demod, noise, bias, lrelu; per-sample constants from LDS, the tile's noise values
//     prefetched with its patch — no late global loads, see below) with the LDS-transposed
//     16-byte row-order stores, in a per-wave LDS image so it overlaps the other waves' MFMA blocks.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

#define ROWB 80

namespace {
constexpr int TH = 8, PH = TH + 2, PW = 34;
constexpr int NVA = PH * PW * 4, NA = (NVA + 255) / 256;   // 1360 16-byte vectors -> 6 per thread
constexpr int W_BYTES = 9 * 32 * ROWB;                     // 23040
constexpr int A_BYTES = PH * PW * ROWB;                    // 27200
constexpr int O_BYTES = 4 * 32 * ROWB;                     // 10240
constexpr int C_BYTES = 32 * 4 + 32 * 4 + 32 * 2;          // per-sample demod scale, bias (fp32) and style (fp16)
constexpr int LDS_BYTES = W_BYTES + A_BYTES + O_BYTES + C_BYTES;   // 60800 -> 2 workgroups per CU
}  // namespace

// FRGB: the input map is produced on the fly from the skip image y (D's fromRGB, stylegan2/models.py:1125-1143: biggan
// denorm(norm(y)) -> 1x1 conv 3 -> 32 + bias + lrelu*sqrt2), 12 bytes per pixel read instead of 64; the tile's interior
// of that map is also written out (p.rgb_x_out) for the D block's skip path, so the separate fromRGB pass disappears.
template <bool FRGB>
__global__ __launch_bounds__(256, 2) void conv_stream_kernel(ConvParams p, int tiles_x, int tiles_y, int PT, int per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ws = smem;
    char* As = smem + W_BYTES;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    char* Os = smem + W_BYTES + A_BYTES + wave * (32 * ROWB);
    // per-sample constants live in LDS: with three tiles of loads in flight ANY late global load that is consumed at once
    // would drain the whole queue (vmcnt retires in order), so the steady state issues patch / noise loads and stores only
    float* Cd = (float*)(smem + W_BYTES + A_BYTES + O_BYTES);   // demod scale [32]
    float* Cb = Cd + 32;                                        // bias [32]
    half_t* Cs = (half_t*)(Cb + 32);                            // style [32]
    const int lr = lane & 31, kh = lane >> 5;
    const int part = t & 3;
    const int tpi = tiles_x * tiles_y;

    // per-thread patch geometry (same for every tile): vector k covers patch pixel (pr, pc), channels part*8..+7
    int prc[NA];      // pr << 8 | pc, or -1 past the end of the patch
    int rel[NA];      // element offset of that pixel relative to the patch origin
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int v = t + 256 * k;
        const int pix = v >> 2;
        const int pr = pix / PW, pc = pix - pr * PW;
        prc[k] = v < NVA ? (pr << 8 | pc) : -1;
        // in_up (nearest x2 input): tile origins are even, so (ty0 - 1 + pr) >> 1 = ty0 / 2 + ((pr - 1) >> 1)
        rel[k] = FRGB ? pr * p.W + pc
                      : (p.in_up ? (((pr - 1) >> 1) * (p.W >> 1) + ((pc - 1) >> 1)) * 32 : (pr * p.W + pc) * 32) + part * 8;
    }
    // FRGB: this thread's 8 output channels of the 1x1 fromRGB conv (part is fixed per thread)
    // packed fp16, sqrt(2) folded in: lrelu(z) * sqrt2 = max(z', 0.2 z') with z' = z * sqrt2
    h8 fw0, fw1, fw2, fbv;
    if (FRGB) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            fw0[j] = (half_t)(p.rgb_w[(part * 8 + j) * 3] * GLASS_SQRT2);
            fw1[j] = (half_t)(p.rgb_w[(part * 8 + j) * 3 + 1] * GLASS_SQRT2);
            fw2[j] = (half_t)(p.rgb_w[(part * 8 + j) * 3 + 2] * GLASS_SQRT2);
            fbv[j] = (half_t)(p.rgb_b[part * 8 + j] * GLASS_SQRT2);
        }
    }

    const int first = blockIdx.x * per_block;
    const int last = min(first + per_block, PT);
    if (first >= last) return;

    // Patch loads are UNCONDITIONAL (out-of-image / out-of-range vectors read the patch origin of a valid tile instead) and
    // are masked to zero when they are written to LDS: a conditional load makes the compiler wait for it at the join.
    auto tile_ok = [&](int id, int k, int ty0, int tx0) {
        const int iy = ty0 - 1 + (prc[k] >> 8), ix = tx0 - 1 + (prc[k] & 255);
        return id < last && prc[k] >= 0 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    };
    struct RSet { h8 a[FRGB ? 1 : NA]; float y3[FRGB ? NA : 1][3]; float nz[2]; };   // one tile's loads in flight
    auto load = [&](int id, RSet& R) {   // issue the patch (+ noise) loads of tile `id`
        const int idc = id < last ? id : first;
        const int b = idc / tpi, trem = idc - b * tpi;
        const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * 32;
        if (FRGB) {
            const long long hw = (long long)p.H * p.W;
            const float* yb = p.rgb_y + (long long)b * 3 * hw;
            const long long org = (long long)(ty0 - 1) * p.W + (tx0 - 1);
#pragma unroll
            for (int k = 0; k < NA; ++k) {
                const long long off = tile_ok(id, k, ty0, tx0) ? org + rel[k] : 0;
#pragma unroll
                for (int c = 0; c < 3; ++c) R.y3[k][c] = yb[c * hw + off];
            }
        } else {
            const half_t* img = p.x + (long long)b * p.x_bstride;
            const long long org = p.in_up ? ((long long)(ty0 >> 1) * (p.W >> 1) + (tx0 >> 1)) * 32
                                          : ((long long)(ty0 - 1) * p.W + (tx0 - 1)) * 32;
#pragma unroll
            for (int k = 0; k < NA; ++k) {
                const long long off = tile_ok(id, k, ty0, tx0) ? org + rel[k] : 0;
                R.a[k] = *(const h8*)(img + off);
            }
        }
        if (p.noise) {
            const float* nzp = p.noise + ((long long)(b / p.batch_size) * p.Ho + ty0 + wave * 2) * p.Wo + tx0 + lr;
            R.nz[0] = nzp[0];
            R.nz[1] = nzp[p.Wo];
        }
    };

    int wb = -1;   // sample whose weights are resident in Ws
    auto step = [&](int id, RSet& R) {   // returns after tile `id` is computed and stored; refills ra with tile id + 3
        const int b = id / tpi, trem = id - b * tpi;
        const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * 32;
        __syncthreads();                           // every wave is done reading As / Ws of the previous tile
        if (wb != b && (wb < 0 || p.w_bstride != 0 || p.dscale || p.shift || p.sn16)) {
            const half_t* wsrc = p.w + (long long)b * p.w_bstride;   // [9][32][32]
            for (int u = t; u < 9 * 32 * 4; u += 256)
                *(h8*)(Ws + (u >> 2) * ROWB + (u & 3) * 16) = *(const h8*)(wsrc + (long long)(u >> 2) * 32 + (u & 3) * 8);
            if (t < 32) {
                Cd[t] = p.dscale ? p.dscale[(long long)b * p.ds_stride + t] : 1.f;
                Cb[t] = (p.bias ? p.bias[t] : 0.f) + (p.shift ? p.shift[(long long)b * p.ds_stride + t] : 0.f);
                Cs[t] = p.sn16 ? p.sn16[(long long)b * p.sn_stride + t] : (half_t)1.f;
            }
            wb = b;
            __syncthreads();
        }
        {
            h8 sh;
            if (p.sn16) sh = *(const h8*)(Cs + part * 8);
            const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < NA; ++k) {
                const int v = t + 256 * k;
                if (NVA % 256 == 0 || v < NVA) {
                    const bool ok = tile_ok(id, k, ty0, tx0);
                    h8 a;
                    if (FRGB) {
                        float c3[3];
#pragma unroll
                        for (int c = 0; c < 3; ++c) c3[c] = fminf(fmaxf((R.y3[k][c] + 1.f) * 0.5f, 0.f), 1.f) * 2.f - 1.f;
                        const half_t h0 = (half_t)c3[0], h1 = (half_t)c3[1], h2 = (half_t)c3[2];
                        const h8 z = fw0 * h0 + fw1 * h1 + fw2 * h2 + fbv;          // v_pk_fma_f16
                        a = __builtin_elementwise_max(z, z * (half_t)0.2f);
                        const int pr = prc[k] >> 8, pc = prc[k] & 255;
                        if (p.rgb_x_out && ok && pr >= 1 && pr <= TH && pc >= 1 && pc <= 32)      // tile interior: the map itself, for the skip path
                            *(h8*)(p.rgb_x_out + (((long long)b * p.H + ty0 - 1 + pr) * p.W + tx0 - 1 + pc) * 32 + part * 8) = a;
                        if (!ok) a = zero;
                    } else {
                        a = ok ? R.a[k] : zero;
                        if (p.sn16) a = a * sh;
                    }
                    *(h8*)(As + (v >> 2) * ROWB + part * 16) = a;
                }
            }
        }
        __syncthreads();
        const float nz0 = p.noise ? p.noise_strength * R.nz[0] : 0.f, nz1 = p.noise ? p.noise_strength * R.nz[1] : 0.f;
        load(id + 3, R);                           // three tiles stay in flight

        f16x acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const h8 wf = *(const h8*)(Ws + ((ty * 3 + tx) * 32 + lr) * ROWB + kk * 32 + kh * 16);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const h8 xf = *(const h8*)(As + ((wave * 2 + i + ty) * PW + lr + tx) * ROWB + kk * 32 + kh * 16);
                        acc[i] = mfma32(wf, xf, acc[i]);
                    }
                }

        if (FRGB && p.rgb_xs_out) {
            // the D block's skip branch wants FIR 4x4 (pad 1) + ::2 of the fromRGB map (modules.py:1238-1254 via 1587-1601): the
            // 8 x 32 tile (+ halo, zeros outside the image) sits in LDS, so its 4 x 16 down-sampled pixels are 16 reads + 5
            // packed-fp16 FIRs per thread — and the 64-byte-per-pixel map itself never has to travel to HBM for the skip path
            const int pix = t >> 2, ly = pix >> 4, lx = pix & 15;
            h8 hr[4];
#pragma unroll
            for (int jy = 0; jy < 4; ++jy) {
                const char* rowp = As + ((2 * ly + jy) * PW + 2 * lx) * ROWB + part * 16;
                const h8 a0 = *(const h8*)rowp, a1 = *(const h8*)(rowp + ROWB), a2 = *(const h8*)(rowp + 2 * ROWB), a3 = *(const h8*)(rowp + 3 * ROWB);
                hr[jy] = (a0 + a3) * (half_t)0.125f + (a1 + a2) * (half_t)0.375f;
            }
            const h8 o = (hr[0] + hr[3]) * (half_t)0.125f + (hr[1] + hr[2]) * (half_t)0.375f;
            *(h8*)(p.rgb_xs_out + (((long long)b * (p.H >> 1) + (ty0 >> 1) + ly) * (p.W >> 1) + (tx0 >> 1) + lx) * 32 + part * 8) = o;
        }
        // ---- epilogue: lane = pixel lr of tile row (wave*2 + i); quads of 4 consecutive channels -------------
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = ty0 + wave * 2 + i;
            const float nz = i ? nz1 : nz0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int o = 8 * g + 4 * kh;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = acc[i][g * 4 + q];
                const f4 d = *(const f4*)(Cd + o), bb = *(const f4*)(Cb + o);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = v[q] * d[q] + nz + bb[q];
                if (p.act == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = lrelu_sqrt2(v[q]);
                } else if (p.act == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                h4 out;
#pragma unroll
                for (int q = 0; q < 4; ++q) out[q] = (half_t)(v[q] * p.out_scale);
                *(h4*)(Os + lr * ROWB + o * 2) = out;
            }
            __builtin_amdgcn_wave_barrier();
            half_t* yrow = p.y + (((long long)b * p.Ho + oy) * p.Wo + tx0) * 32;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int v = lane + 64 * k;   // 128 16-byte vectors = 32 px x 4
                *(h8*)(yrow + (long long)(v >> 2) * 32 + (v & 3) * 8) = *(const h8*)(Os + (v >> 2) * ROWB + (v & 3) * 16);
            }
            __builtin_amdgcn_wave_barrier();
        }
    };

    RSet r0, r1, r2;
    load(first, r0);
    load(first + 1, r1);
    load(first + 2, r2);
    for (int id = first; id < last; id += 3) {
        step(id, r0);
        if (id + 1 >= last) break;
        step(id + 1, r1);
        if (id + 2 >= last) break;
        step(id + 2, r2);
    }
}

const char* launch_conv_stream(const ConvParams& p, hipStream_t st) {
    static const bool off = getenv("GLASS_NO_STREAM") != nullptr;   // experiment knob
    if (off || p.up || p.y32 || !p.y || p.KS != 3 || p.stride != 1 || p.pad != 1 || (p.sn && !p.sn16)) return nullptr;
    const bool frgb = p.rgb_y != nullptr;
    if (frgb && (!p.rgb_w || !p.rgb_b || (!p.rgb_x_out && !p.rgb_xs_out) || p.sn)) return nullptr;
    if (p.Cin != 32 || p.Neff != 32 || p.Cout != 32 || p.res || p.pre_shift || p.res_cs || p.res_up) return nullptr;
    if (frgb && (p.in_up || p.shift)) return nullptr;
    if (p.Wc % 32 != 0 || p.Hc % TH != 0 || p.W >= 256 * 32 || (!frgb && p.x_bstride == 0 && p.B > 1)) return nullptr;
    if ((long long)p.H * p.W * p.Cin >= (1LL << 31)) return nullptr;
    const int tiles_x = p.Wc / 32, tiles_y = p.Hc / TH;
    const int PT = p.B * tiles_x * tiles_y;
    static int slots = 0;
    if (!slots) {
        (void)hipFuncSetAttribute((const void*)conv_stream_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv_stream_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        hipDeviceProp_t prop;
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipGetDeviceProperties(&prop, dev);
        slots = prop.multiProcessorCount * 2;
    }
    if (PT < slots * 8) return nullptr;        // streaming only pays with many tiles per workgroup
    const int per_block = (PT + slots - 1) / slots;
    const int grid = (PT + per_block - 1) / per_block;
    if (frgb) {
        hipLaunchKernelGGL(conv_stream_kernel<true>, dim3(grid), dim3(256), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_block);
        return "conv_stream_kernel<fromrgb>";
    }
    hipLaunchKernelGGL(conv_stream_kernel<false>, dim3(grid), dim3(256), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_block);
    return "conv_stream_kernel";
}
