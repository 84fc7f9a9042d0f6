// conv_stream.hip — 3x3 stride-1 convolution, Cin = Cout = 32, for the HBM-bound 1024^2 layers
// (G conv_blocks.8.conv_block.1 and D conv_blocks.0.conv_block.0: 64 MB in + 64 MB out per candidate for 9.7 GFLOP).
//
// conv_tiled.hip serves these layers at ~2.6-3.1 TB/s although the same tile access pattern with no compute
// streams at 5.7 TB/s (tools/tile_copy.hip): one tile per workgroup leaves ~70 KB of loads in flight per CU
// (4 resident blocks x 24 KB patch, only while a block is in its load phase), and at the loaded HBM latency
// (~10 us) that is what caps the rate.  This kernel keeps the bytes in flight instead:
//   * persistent workgroups (3 per CU), each walking a contiguous range of TH x 32 tiles;
//   * the input patches of the next TWO tiles are in flight in registers, refilled one vector per MFMA tap (round 2: a plain
//     read stream reaches 5.8 TB/s with 16 KB in flight per CU — tools/inflight_probe — so depth was never the limit; what a
//     phase trace shows instead is ~2000 cycles per tile with all four waves stalled ISSUING a burst of loads);
//   * the whole 3x3x32x32 weight tensor lives in LDS for the block's lifetime (re-staged only when the sample,
//     i.e. the pre-modulated weight set, changes): no weight traffic and no barriers inside a tile's 36 MFMAs;
//   * epilogue as in conv_tiled (demod, noise, bias, lrelu; per-sample constants from LDS, the tile's noise values
//     prefetched with its patch — no late global loads, see below) with the LDS-transposed
//     16-byte row-order stores, in a per-wave LDS image so it overlaps the other waves' MFMA blocks.
#include "common.h"
#include "kernels.h"
#include <stdio.h>
#include <stdlib.h>

namespace {
constexpr int TH = 8, PH = TH + 2, PW = 34;
constexpr int NVA = PH * PW * 4, NA = (NVA + 255) / 256;   // 1360 16-byte vectors -> 6 per thread
constexpr int W_BYTES = 9 * 32 * 64;                       // 18432  weight image, rows tap * 32 + n
constexpr int PP = 36;                                     // LDS pitch of a patch row in pixels (multiple of 4: the swizzle key ignores the row)
constexpr int A_BYTES = PH * PP * 64;                      // 23040  patch image, rows pr * 36 + pc
constexpr int O_BYTES = 4 * 32 * 64;                       // 8192   per-wave output transposition
constexpr int C_BYTES = 32 * 4 + 32 * 4 + 32 * 2 + 4 * 32 * 2;   // per-sample demod scale, bias (fp32), style (fp16); fromRGB rows (fp16)
constexpr int LDS_BYTES = W_BYTES + A_BYTES + O_BYTES + C_BYTES;   // 51520 -> three workgroups per CU
// dense 64-byte rows, 16-byte chunk XOR-swizzled by the row (conv_glds.hip's layout: conflict-free 32-lane fragment walks)
__device__ __forceinline__ int swz(int row, int chunk) { return (row << 6) + ((chunk ^ ((row >> 2) & 3)) << 4); }
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// patch image: pixel (pr, pc) at row pr * PP + pc, chunk swizzled by the COLUMN only (a row step is an immediate offset)
__device__ __forceinline__ int swa(int pr, int pc, int chunk) { return ((pr * PP + pc) << 6) + ((chunk ^ ((pc >> 2) & 3)) << 4); }
}  // namespace

// dev tool (GLASS_STREAM_TRACE=path): phase timestamps (shader clocks) of workgroup 0, first 64 tiles; the production
// instance (TR = false) carries no trace code
__device__ unsigned long long* g_stream_trace = nullptr;
#define STRACE(ph) \
    if (TR && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && id - first < 64) \
        g_stream_trace[((id - first) * 8 + (ph)) * 4 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime()

// FRGB: the input map is produced on the fly from the skip image y (D's fromRGB, stylegan2/models.py:1125-1143: biggan
// denorm(norm(y)) -> 1x1 conv 3 -> 32 + bias + lrelu*sqrt2), 12 bytes per pixel read instead of 64; the tile's interior
// of that map can be written out (p.rgb_x_out) and / or its FIR (pad 1) + ::2 (p.rgb_xs_out) for the D block's skip path.
template <bool FRGB, bool TR = false>
__global__ __launch_bounds__(256, 3) void conv_stream_kernel(ConvParams p, int tiles_x, int tiles_y, int PT, int per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ws = smem;
    char* As = smem + W_BYTES;
    const int t = threadIdx.x;
    // per-sample constants live in LDS: ANY late global load that is consumed at once would retire the whole in-order vmcnt
    // queue, so the steady state issues patch / noise loads and stores only
    float* Cd = (float*)(smem + W_BYTES + A_BYTES + O_BYTES);   // demod scale [32]
    float* Cb = Cd + 32;                                        // bias [32]
    half_t* Cs = (half_t*)(Cb + 32);                            // style [32]
    const int tpi = tiles_x * tiles_y;

    // patch vector k of thread t covers patch pixel (pr, pc) = divmod((t + 256 k) >> 2, PW), channels part*8..+7.  The geometry is
    // re-derived where it is used from an opaque copy of the thread id: as loop invariants the 12 values per thread (and every
    // address term the compiler derives from them) get hoisted and, at three workgroups per CU (168 VGPRs), spilled.
    auto vec_pix = [&](int tt, int k) { return (tt + 256 * k) >> 2; };
    half_t* Cf = Cs + 32;                                        // [4][32]: fromRGB weight rows r, g, b and bias
    if (FRGB) {
        if (t < 32) {
            Cf[t] = (half_t)(p.rgb_w[t * 3] * GLASS_SQRT2);
            Cf[32 + t] = (half_t)(p.rgb_w[t * 3 + 1] * GLASS_SQRT2);
            Cf[64 + t] = (half_t)(p.rgb_w[t * 3 + 2] * GLASS_SQRT2);
            Cf[96 + t] = (half_t)(p.rgb_b[t] * GLASS_SQRT2);
        }
        __syncthreads();
    }

    const int first = blockIdx.x * per_block;
    const int last = min(first + per_block, PT);
    if (first >= last) return;

    // Patch loads are UNCONDITIONAL (out-of-image / out-of-range vectors read the patch origin of a valid tile instead) and
    // are masked to zero when they are written to LDS: a conditional load makes the compiler wait for it at the join.
    auto tile_ok = [&](int id, int pr, int pc, bool in_patch, int ty0, int tx0) {
        const int iy = ty0 - 1 + pr, ix = tx0 - 1 + pc;
        return id < last && in_patch && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    };
    struct RSet { h8 a[FRGB ? 1 : NA]; float y3[FRGB ? NA : 1][3]; float nz[2]; };   // one tile's loads in flight
    // The loads of a refill are issued ONE PATCH VECTOR AT A TIME between the taps of the MFMA loop (a burst of 8 - 20 load
    // instructions keeps the CU's address unit busy for ~2000 cycles with all four waves stalled at issue — phase trace).
    auto load_part = [&](int id, RSet& R, int k) {
        const int idc = id < last ? id : first;
        const int b = idc / tpi, trem = idc - b * tpi;
        const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * 32;
        const int tt = opaque(threadIdx.x);
        if (k < NA) {
            const int pix = vec_pix(tt, k), pr = pix / PW, pc = pix - pr * PW;
            const bool ok = tile_ok(id, pr, pc, tt + 256 * k < NVA, ty0, tx0);
            if (FRGB) {
                const int hw = p.H * p.W;
                const float* yb = p.rgb_y + (long long)b * 3 * hw + ((ty0 - 1) * p.W + (tx0 - 1));     // uniform
                const int off = ok ? pr * p.W + pc : (1 - ty0) * p.W + (1 - tx0);                        // masked: the image origin
#pragma unroll
                for (int c = 0; c < 3; ++c) R.y3[k][c] = yb[c * hw + off];
            } else {
                const half_t* img = p.x + (long long)b * p.x_bstride;                                     // uniform
                // in_up (nearest x2 input): tile origins are even, so (ty0 - 1 + pr) >> 1 = ty0 / 2 + ((pr - 1) >> 1)
                const int org = p.in_up ? ((ty0 >> 1) * (p.W >> 1) + (tx0 >> 1)) * 32 : ((ty0 - 1) * p.W + (tx0 - 1)) * 32;
                const int rel = p.in_up ? (((pr - 1) >> 1) * (p.W >> 1) + ((pc - 1) >> 1)) * 32 : (pr * p.W + pc) * 32;
                const int off = ok ? org + rel + (tt & 3) * 8 : 0;                                        // H * W * 32 < 2^31
                R.a[k] = *(const h8*)(img + off);
            }
        } else if (p.noise) {
            const float* nzp = p.noise + ((long long)(b / p.batch_size) * p.Ho + ty0 + ((tt >> 6) & 3) * 2) * p.Wo + tx0 + (tt & 31);
            R.nz[0] = nzp[0];
            R.nz[1] = nzp[p.Wo];
        }
    };
    auto load = [&](int id, RSet& R) {
#pragma unroll
        for (int k = 0; k <= NA; ++k) load_part(id, R, k);
    };

    int wb = -1;   // sample whose weights are resident in Ws
    auto step = [&](int id, RSet& R) {   // tile `id` (ids past `last` pad the loop: computed on a valid tile's data, never stored)
        const bool valid = id < last;
        const int idc = valid ? id : first;
        const int b = idc / tpi, trem = idc - b * tpi;
        const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * 32;
        STRACE(0);
        __syncthreads();                           // every wave is done reading As / Ws of the previous tile
        STRACE(1);
        if (wb != b && (wb < 0 || p.w_bstride != 0 || p.dscale || p.shift || p.sn16)) {
            const half_t* wsrc = p.w + (long long)b * p.w_bstride;   // [9][32][32]
            for (int u = t; u < 9 * 32 * 4; u += 256)
                *(h8*)(Ws + swz(u >> 2, u & 3)) = *(const h8*)(wsrc + (long long)(u >> 2) * 32 + (u & 3) * 8);
            if (t < 32) {
                Cd[t] = p.dscale ? p.dscale[(long long)b * p.ds_stride + t] : 1.f;
                Cb[t] = (p.bias ? p.bias[t] : 0.f) + (p.shift ? p.shift[(long long)b * p.ds_stride + t] : 0.f);
                Cs[t] = p.sn16 ? p.sn16[(long long)b * p.sn_stride + t] : (half_t)1.f;
            }
            wb = b;
            __syncthreads();
        }
        {
            const int tt = opaque(threadIdx.x), part = tt & 3;
            h8 sh;
            if (p.sn16) sh = *(const h8*)(Cs + part * 8);
            h8 fw0, fw1, fw2, fbv;
            if (FRGB) {
                fw0 = *(const h8*)(Cf + part * 8); fw1 = *(const h8*)(Cf + 32 + part * 8);
                fw2 = *(const h8*)(Cf + 64 + part * 8); fbv = *(const h8*)(Cf + 96 + part * 8);
            }
            const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            // interior tiles need no per-vector bounds test (uniform): only the tail vectors past the patch end are skipped
            const bool border = !valid || ty0 == 0 || tx0 == 0 || ty0 + TH >= p.H || tx0 + 32 >= p.W;
#pragma unroll
            for (int k = 0; k < NA; ++k) {
                const int v = tt + 256 * k;
                if (NVA % 256 == 0 || v < NVA) {
                    const int pix = vec_pix(tt, k), pr = pix / PW, pc = pix - pr * PW;
                    const bool ok = border ? tile_ok(id, pr, pc, true, ty0, tx0) : true;
                    h8 a;
                    if (FRGB) {
                        float c3[3];
#pragma unroll
                        for (int c = 0; c < 3; ++c) c3[c] = fminf(fmaxf((R.y3[k][c] + 1.f) * 0.5f, 0.f), 1.f) * 2.f - 1.f;
                        const half_t h0 = (half_t)c3[0], h1 = (half_t)c3[1], h2 = (half_t)c3[2];
                        const h8 z = fw0 * h0 + fw1 * h1 + fw2 * h2 + fbv;          // v_pk_fma_f16
                        a = __builtin_elementwise_max(z, z * (half_t)0.2f);
                        if (p.rgb_x_out && ok && pr >= 1 && pr <= TH && pc >= 1 && pc <= 32)      // tile interior: the map itself, for the skip path
                            *(h8*)(p.rgb_x_out + (((long long)b * p.H + ty0 - 1 + pr) * p.W + tx0 - 1 + pc) * 32 + part * 8) = a;
                        if (!ok) a = zero;
                    } else {
                        a = ok ? R.a[k] : zero;
                        if (p.sn16) a = a * sh;
                    }
                    *(h8*)(As + swa(pr, pc, part)) = a;
                }
            }
        }
        STRACE(2);
        __syncthreads();
        STRACE(3);
        const float nz0 = p.noise ? p.noise_strength * R.nz[0] : 0.f, nz1 = p.noise ? p.noise_strength * R.nz[1] : 0.f;
        STRACE(4);

        f16x acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
        const int tm = opaque(threadIdx.x), lr = tm & 31, kh = (tm >> 5) & 1, wave = tm >> 6, lane = tm & 63;
        char* Os = smem + W_BYTES + A_BYTES + wave * (32 * 64);
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const int tap = ty * 3 + tx;
                if (tap <= NA) load_part(id + 2, R, tap);       // refill (two tiles stay in flight), one vector per tap
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const h8 wf = *(const h8*)(Ws + swz(tap * 32 + lr, kk * 2 + kh));
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const h8 xf = *(const h8*)(As + swa(wave * 2 + i + ty, lr + tx, kk * 2 + kh));
                        acc[i] = mfma32(wf, xf, acc[i]);
                    }
                }
            }

        STRACE(5);
        if (FRGB && p.rgb_xs_out && valid) {
            // the D block's skip branch wants FIR 4x4 (pad 1) + ::2 of the fromRGB map (modules.py:1238-1254 via 1587-1601): the
            // 8 x 32 tile (+ halo, zeros outside the image) sits in LDS, so its 4 x 16 down-sampled pixels are 16 reads + 5
            // packed-fp16 FIRs per thread — and the 64-byte-per-pixel map itself never has to travel to HBM for the skip path
            const int tx_ = opaque(threadIdx.x), part = tx_ & 3;
            const int pix = tx_ >> 2, ly = pix >> 4, lx = pix & 15;
            h8 hr[4];
#pragma unroll
            for (int jy = 0; jy < 4; ++jy) {
                const int pr = 2 * ly + jy, pc = 2 * lx;
                const h8 a0 = *(const h8*)(As + swa(pr, pc, part)), a1 = *(const h8*)(As + swa(pr, pc + 1, part)),
                         a2 = *(const h8*)(As + swa(pr, pc + 2, part)), a3 = *(const h8*)(As + swa(pr, pc + 3, part));
                hr[jy] = (a0 + a3) * (half_t)0.125f + (a1 + a2) * (half_t)0.375f;
            }
            const h8 o = (hr[0] + hr[3]) * (half_t)0.125f + (hr[1] + hr[2]) * (half_t)0.375f;
            *(h8*)(p.rgb_xs_out + (((long long)b * (p.H >> 1) + (ty0 >> 1) + ly) * (p.W >> 1) + (tx0 >> 1) + lx) * 32 + part * 8) = o;
        }
        STRACE(6);
        // ---- epilogue: lane = pixel lr of tile row (wave*2 + i); quads of 4 consecutive channels.  fp32 up to the activation
        // input (acc * demod + noise + bias), then packed fp16: act 0 / 1 / 2 = max(v * k1, v * k2) with (k1, k2) =
        // (s, s) / (sqrt2 s, 0.2 sqrt2 s) / (s, 0) -------------------------------------------------------------------------------
        const half_t k1 = (half_t)((p.act == 1 ? GLASS_SQRT2 : 1.f) * p.out_scale);
        const half_t k2 = (half_t)((p.act == 1 ? 0.2f * GLASS_SQRT2 : p.act == 2 ? 0.f : 1.f) * p.out_scale);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = ty0 + wave * 2 + i;
            const float nz = i ? nz1 : nz0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int o = 8 * g + 4 * kh;
                const f4 d = *(const f4*)(Cd + o), bb = *(const f4*)(Cb + o);
                h4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (half_t)(acc[i][g * 4 + q] * d[q] + (nz + bb[q]));
                *(h4*)(Os + swz(lr, o >> 3) + (o & 4) * 2) = __builtin_elementwise_max(v * k1, v * k2);
            }
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                half_t* yrow = p.y + (((long long)b * p.Ho + oy) * p.Wo + tx0) * 32;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int v = lane + 64 * k;   // 128 16-byte vectors = 32 px x 4
                    *(h8*)(yrow + (long long)(v >> 2) * 32 + (v & 3) * 8) = *(const h8*)(Os + swz(v >> 2, v & 3));
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    };

    // two register sets; no exit between the two steps (hipcc's wait-count merge at the loop header otherwise stops counting
    // the other set's refill as younger and every patch wait drains the queue): an odd tile count is padded
    RSet r0, r1;
    load(first, r0);
    load(first + 1, r1);
    for (int id = first; id < last; id += 2) {
        step(id, r0);
        step(id + 1, r1);
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
        (void)hipFuncSetAttribute((const void*)conv_stream_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv_stream_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        hipDeviceProp_t prop;
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipGetDeviceProperties(&prop, dev);
        slots = prop.multiProcessorCount * 3;      // 168 VGPRs, 51.5 KB of LDS: three workgroups per CU
    }
    if (PT < slots * 6) return nullptr;        // streaming only pays with many tiles per workgroup
    const int per_block = (PT + slots - 1) / slots;
    const int grid = (PT + per_block - 1) / per_block;
    if (const char* trace_path = getenv("GLASS_STREAM_TRACE")) {       // dev tool: traced instance, one launch, timestamps to a file
        unsigned long long* dtr = nullptr;
        (void)hipMalloc(&dtr, 64 * 8 * 4 * sizeof(unsigned long long));
        (void)hipMemset(dtr, 0, 64 * 8 * 4 * sizeof(unsigned long long));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stream_trace), &dtr, sizeof dtr);
        (void)hipFuncSetAttribute((const void*)conv_stream_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv_stream_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (frgb) hipLaunchKernelGGL((conv_stream_kernel<true, true>), dim3(grid), dim3(256), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_block);
        else hipLaunchKernelGGL((conv_stream_kernel<false, true>), dim3(grid), dim3(256), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_block);
        static unsigned long long hb[64 * 8 * 4];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hb, dtr, sizeof hb, hipMemcpyDeviceToHost);
        (void)hipFree(dtr);
        if (FILE* f = fopen(trace_path, "a")) {
            fprintf(f, "# conv_stream%s: tile phase t[wave0..3]; phases 0 enter, 1 after sync, 2 patch staged, 3 after sync, 4 refill issued, "
                       "5 MFMAs done, 6 skip by-product done, (next 0) epilogue done; per_block=%d\n", frgb ? "<fromrgb>" : "", per_block);
            for (int i = 0; i < 64 && i < per_block; ++i)
                for (int ph = 0; ph < 7; ++ph) {
                    fprintf(f, "%d %d", i, ph);
                    for (int w = 0; w < 4; ++w) fprintf(f, " %llu", hb[(i * 8 + ph) * 4 + w] - hb[0]);
                    fprintf(f, "\n");
                }
            fclose(f);
        }
        return frgb ? "conv_stream_kernel<fromrgb>" : "conv_stream_kernel";
    }
    if (frgb) {
        hipLaunchKernelGGL((conv_stream_kernel<true, false>), dim3(grid), dim3(256), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_block);
        return "conv_stream_kernel<fromrgb>";
    }
    hipLaunchKernelGGL((conv_stream_kernel<false, false>), dim3(grid), dim3(256), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_block);
    return "conv_stream_kernel";
}
