// common.h — shared types for the gfx950 kernels (wave64, MFMA f16 -> f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>

// A/B knobs (environment variables that switch a dispatcher to an older kernel or change a tile rule) exist in the developer
// build only: `make AB=1` (-DGLASS_AB_KNOBS -> tools/lib/libglass_ab.so, driven by tools/ab_bench.sh through GLASS_LIB).  In the
// release library every knob reads as "not set" and the branch it guards folds away — the product has ONE code path per layer.
#ifdef GLASS_AB_KNOBS
inline const char* glass_knob(const char* name) { return getenv(name); }
#else
inline const char* glass_knob(const char*) { return nullptr; }
#endif

typedef _Float16 half_t;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));

#define GLASS_SQRT2 1.41421356237309504880f

// v_mfma_f32_32x32x16_f16: D[32x32] += A[32x16] * B[16x32].
// lane l holds A[i = l&31][k = (l>>5)*8 + j] and B[k = (l>>5)*8 + j][n = l&31], j = 0..7;
// D register r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ f16x mfma32(h8 a, h8 b, f16x c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ---- toRGB applied to a wave's output tile while it is still in registers (conv_stream / conv_tiled / conv_glds) -------------
// The activated fp16 quads a lane holds after the epilogue math (lane (px, kh): channels j*32 + 8g + 4kh + q of pixel px) are a
// valid MFMA B operand as they stand — the K order of an MFMA is free as long as A uses the same one.  A = 16-row weight table
// per image row i of the wave: rows 0-2 (i = 0) / 4-6 (i = 1) the fp16 hi part of w[c][.] * style, rows 8-10 / 12-14 the lo part
// * 2^11; after the MFMAs lane half kh holds image row kh: registers 0-2 hi, 4-6 lo.
// Table element (tab, n, hidx), hidx = ((j*2 + gp)*2 + kh)*8 + e  <->  channel j*32 + 8*(2gp + (e>>2)) + 4kh + (e&3).
__device__ __forceinline__ int trgb_channel(int hidx) {
    const int chunk = hidx >> 3, e = hidx & 7;
    return (chunk >> 2) * 32 + 8 * (2 * ((chunk >> 1) & 1) + (e >> 2)) + 4 * (chunk & 1) + (e & 3);
}
__device__ __forceinline__ half_t trgb_table_value(int tab, int n, float w) {
    if ((n & 3) == 3 || ((n >> 2) & 1) != tab) return (half_t)0.f;
    const half_t hv = (half_t)w;
    return (n & 8) ? (half_t)((w - (float)hv) * 2048.f) : hv;
}
// FIR-upsampled skip image (modules.py:580-602; zero-insert, pad [3,1], 4x4 FIR * 4): out[2m] = .75 x[m-1] + .25 x[m],
// out[2m+1] = .25 x[m-1] + .75 x[m]; taps t[dy*2+dx] = yprev[my-1+dy][mx-1+dx] (clamped loads, zero weight outside)
__device__ __forceinline__ float trgb_skip(const float t[4], int oy, int ox) {
    const int my = oy >> 1, mx = ox >> 1;
    const float wy0 = (oy & 1) ? 0.25f : 0.75f, wx0 = (ox & 1) ? 0.25f : 0.75f;
    float s = 0.f;
    s += ((my >= 1 && mx >= 1) ? wy0 * wx0 : 0.f) * t[0];
    s += (my >= 1 ? wy0 * (1.f - wx0) : 0.f) * t[1];
    s += (mx >= 1 ? (1.f - wy0) * wx0 : 0.f) * t[2];
    s += (1.f - wy0) * (1.f - wx0) * t[3];
    return s;
}

// ---- per-device launch set-up (function attributes, CU counts): one engine per (process, GPU), but ONE process may drive
// several GPUs — a function-local `static bool` would configure the first device only (VERDICT r2 / ADVICE r2) ------------------
// Engines on one GPU may be driven from different host threads: first() is true until SOME caller has come back from the set-up
// block it guards (the set-up calls are idempotent, so a race repeats them; it never lets a thread launch before the attribute is
// set — ADVICE r4: `exchange(true)` marked the device done before the winner's hipFuncSetAttribute had run).  Usage:
//     static DevOnce once;  once.run([&] { hipFuncSetAttribute(...); });
struct DevOnce {
    std::atomic<bool> done[32] = {};
    static int dev() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 32) return -1;
        return d;
    }
    bool first() const {        // true while the current device's set-up has not been completed by anyone
        const int d = dev();
        return d < 0 || !done[d].load(std::memory_order_acquire);
    }
    void set() {
        const int d = dev();
        if (d >= 0) done[d].store(true, std::memory_order_release);
    }
    template <class F> void run(F&& setup) {     // the usual form: once.run([&] { hipFuncSetAttribute(...); });
        if (first()) {
            setup();
            set();
        }
    }
};
struct GlassDevProps { int cus, lds_optin; };
inline GlassDevProps glass_dev_props() {   // CU count and the largest dynamic-LDS block a kernel may opt in to, current device
    static std::atomic<int> cus[32] = {}, lds[32] = {};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 32) d = 0;
    if (!cus[d].load()) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess) {
            lds[d].store((int)prop.sharedMemPerBlockOptin > 0 ? (int)prop.sharedMemPerBlockOptin : (int)prop.sharedMemPerBlock);
            cus[d].store(prop.multiProcessorCount);
        }
    }
    return GlassDevProps{cus[d].load() > 0 ? cus[d].load() : 256, lds[d].load() > 0 ? lds[d].load() : 64 * 1024};
}
inline int glass_cu_count() { return glass_dev_props().cus; }
// Launchers of kernels that opt in to more than the default 64 KB of dynamic LDS ask this first and REFUSE the layer (nullptr /
// false: the dispatcher falls through to the next kernel family) when the device does not offer the block — instead of launching
// into an asynchronous failure that only surfaces as a generic HIP error at the pass's final synchronisation.
// A refusal is reported once per (device, size) on stderr: a runtime that advertises less LDS than gfx950 has (160 KiB) would
// otherwise move every large-LDS family onto its slow fallback in silence.
inline bool glass_lds_fits(int bytes) {
    const GlassDevProps dp = glass_dev_props();
    if (bytes <= dp.lds_optin) return true;
    static std::mutex mu;
    static int seen_dev[16], seen_bytes[16], n_seen = 0;
    int d = 0;
    (void)hipGetDevice(&d);
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < n_seen; ++i)
        if (seen_dev[i] == d && seen_bytes[i] == bytes) return false;
    if (n_seen == 16) return false;            // table full: the last line printed said so
    seen_dev[n_seen] = d;
    seen_bytes[n_seen++] = bytes;
    if (n_seen == 16) fprintf(stderr, "libglass: further LDS refusals are not reported\n");
    fprintf(stderr, "libglass: device %d offers %d B of opt-in LDS per workgroup; a kernel family that needs %d B is refused and its "
                    "layers fall back to a slower kernel (expected on gfx950: 163840 B)\n", d, dp.lds_optin, bytes);
    return false;
}

// Launch-size thresholds ("does this grid fill the chip?") are evaluated at this NOMINAL population, never at the launch's own
// candidate count: the kernel instance a layer runs on is then a function of the layer geometry alone, and a population scored
// in one call, in chunks or as shards on several GPUs goes through the same arithmetic — bitwise equal rows (tests:
// test_pop512_as_eight_shards_of_64, test_full_size_ffhq_full_population, test_full_size_offset_shards).  64 = the headline
// population per GPU (BASELINE.json configs[1], configs[3]).
#define GLASS_NOMINAL_POP 64

__device__ __forceinline__ float lrelu_sqrt2(float v) { return (v > 0.f ? v : 0.2f * v) * GLASS_SQRT2; }
// Branch-free activation + output scale for the conv epilogues (s > 0): max(v k1, v k2) with
//   act 1 (lrelu 0.2 * sqrt2): (sqrt2 s, 0.2 sqrt2 s);  act 2 (relu): (s, 0);  none: (s, s).
// As f4 arithmetic it compiles to v_pk_mul_f32 pairs + v_max_f32 — a third of the per-value compare / select / multiply chains
// the epilogues spent before (they are VALU-issue bound: every wave of the workgroup is in its epilogue at the same time).
struct ActK { float k1, k2; };
__device__ __forceinline__ ActK act_consts(int act, float s) {
    return ActK{(act == 1 ? GLASS_SQRT2 : 1.f) * s, (act == 1 ? 0.2f * GLASS_SQRT2 : act == 2 ? 0.f : 1.f) * s};
}
__device__ __forceinline__ f4 act_apply(f4 v, ActK k) { return __builtin_elementwise_max(v * k.k1, v * k.k2); }

// Parameters of one convolution launch (implicit GEMM, NHWC fp16 activations).
// GEMM view: M = B*Hc*Wc (conv grid), N = Neff, K = KS*KS*Cin.
struct ConvParams {
    const half_t* x;        // input  [B][H][W][Cin]
    long long x_bstride;    // elements between images (0 = broadcast, e.g. the learned const)
    int B, H, W, Cin;
    int Hc, Wc;             // conv grid (== output grid unless up)
    int KS, stride, pad;
    const half_t* w;        // [KS*KS][Neff][Cin], Cin contiguous
    const half_t* w_up;     // up only: un-folded weights [9][Cout][Cin] for the fused path (nullable)
    long long w_bstride;    // elements between per-sample weight sets (0 = shared weights); pre-modulated path
    int Neff, Cout;         // Neff = Cout * (up ? 4 : 1); n = phase*Cout + o
    int up;                 // 1: depth-to-space 2x2 (folded transposed conv + FIR)
    int Ho, Wo;             // output grid
    const float* sn;        // [B][sn_stride] normalised style (nullable)
    int sn_stride;
    const half_t* sn16;     // fp16 copy of sn, same indexing: what the LDS-tiled kernels read (one 16-byte load per chunk, no
                            // convert on the load path — a convert there would make the kernel wait for its own prefetch)
    const half_t* pre_shift16;  // fp16 copy of pre_shift
    const float* pre_shift; // [B][sn_stride] (nullable; needs sn): x <- relu(x * sn + pre_shift) on the way in, in-bounds
                            // pixels only (zero padding stays zero) — BigGAN batch norm + ReLU ahead of the conv
    int in_up;              // 1: the input is read through a nearest x2 upsample (H, W = upsampled dims; x holds H/2 x W/2)
    const float* dscale;    // [B][ds_stride] demod * smax (nullable)
    int ds_stride;
    const float* noise;     // [n_minibatch][Ho][Wo] (nullable)
    float noise_strength;
    int batch_size;         // candidates per noise plane
    const float* bias;      // [Cout] (nullable)
    const float* shift;     // [B][ds_stride] per-sample per-channel shift added after bias (nullable; BigGAN conditional BN)
    int act;                // 1: leaky-relu(0.2) * sqrt(2); 2: relu
    const half_t* res;      // residual [B][Ho][Wo][res_cs] added after activation (nullable)
    int res_cs;             // channel stride of the residual rows (0 = Cout): first Cout of res_cs channels are used
    int res_up;             // 1: the residual is read through a nearest x2 upsample (it holds Ho/2 x Wo/2)
    float out_scale;
    // conv_stream only: build the 32-channel input map on the fly from the skip image (D fromRGB fused into the first D conv)
    const float* rgb_y;     // [B][3][H][W] fp32 (nullable: normal input x)
    const float* rgb_w;     // [32][3]
    const float* rgb_b;     // [32]
    half_t* rgb_x_out;      // [B][H][W][32]: the fromRGB map, written as a side output (skip path input; nullable)
    half_t* rgb_xs_out;     // [B][H/2][W/2][32]: FIR (pad 1) + ::2 of the fromRGB map — the D block's skip-branch input, taken
                            // from the tile already staged in LDS (nullable)
    // conv_stream only: the generator's LAST conv feeds nothing but toRGB (stylegan2/models.py:852-870, 1004-1013), so the
    // kernel applies it to the activated tile in its accumulators and writes the skip image only — the feature map never
    // reaches HBM.  trgb_yout != nullptr selects this mode (p.y is not written).
    const float* trgb_w;    // [3][Cout] toRGB weights (runtime coefficient applied)
    const float* trgb_b;    // [3]
    const float* trgb_sn;   // [B][trgb_sn_stride] normalised style of the toRGB layer
    int trgb_sn_stride;
    const float* trgb_smax; // [B * trgb_smax_stride]: the style's normaliser (toRGB has no demod to cancel it)
    int trgb_smax_stride;
    const float* trgb_yprev;// [B][3][Ho/2][Wo/2] skip image of the previous block (nullable)
    float* trgb_yout;       // [B][3][Ho][Wo]
    float* rgb_tanh_out;    // conv_tiled (3x3, one 32-wide n tile, fast path): [B][3][Ho][Wo] = tanh of output channels 0..2 taken from the fp32
                            // accumulators (BigGAN's conv_to_rgb[:, :3] + tanh, oracle/biggan_ref.py generator()); p.y is NOT written
    float* trgb_part;       // conv_glds persistent form, layers with SEVERAL 128-wide n tiles: [NTn][B][3][Ho][Wo] fp32 — every n tile writes the toRGB
                            // partial sum of its 128 channels (no bias, no skip image); launch_trgb_finish adds them in n-tile order
    const half_t* trgb_tab; // conv_tiled / conv_glds (their output map IS stored too): [B][2][16][Neff] fp16 weight tables from
                            // launch_trgb_tables (the MFMA A operand of the 1x1 conv in accumulator-lane channel order)
    // conv_tiled<3,1,8,N> only: FIR 4x4 (pad 1) + ::2 of the INPUT map [B][H/2][W/2][Cin] as a by-product of the staged patch — the D
    // block's skip-branch input (modules.py:1238-1254 via 1587-1601), which is a function of the same tensor the block's first conv reads
    half_t* xs_out;
    // upfir only: per-(sample, channel) factor applied to the finished output — the NEXT layer's style, so that the consumer runs
    // without its activation-side modulation (x * s is the same product wherever it is formed)
    const half_t* post_scale16;   // [B][post_stride] (nullable)
    int post_stride;
    // conv_tiled stride 2 only: the D block's skip branch (1x1 conv of the down-sampled block input, modules.py:1238-1254) as extra
    // K stages of the same kernel — after the 3x3 stages the activation is applied IN the accumulators and the skip MFMAs land on
    // top of it (no separate 1x1 pass, no fp16 round trip of its result)
    const half_t* skip_x;   // [B][Ho][Wo][Cin] (nullable)
    const half_t* skip_w;   // [Neff][Cin]
    int dry_run;            // conv_tiled / conv_glds launchers: report the kernel that would run, launch nothing
    // chunk-planar activation maps [B][C / 8][H][W][8] (round 6): conv_wreg stages the 16-channel chunks of its input at different times,
    // and in the pixel-major layout the pieces of every 128-byte line crossed the fabric once per chunk (the first LDS-resident form of this layer, two
    // 32-channel chunks: 4.6 GB fetched per launch for 2.15 GB of input, PMC).  With 8-channel planes a line belongs to ONE chunk and a
    // 64-lane LDS-DMA piece (16 B per lane) is 1 KB of contiguous memory.  Only the producer / consumer pairs that implement the layout
    // accept the flags (upfir2<false> / dblock0 write it, conv_wreg reads it); every other launcher refuses them.
    int x_planar32;        // conv_s2 only (round 6): its blurred input map is [B][Cin / 32][H][W][32] — written so by blur_kernel for it.  The kernel
                            // stages ONE 32-channel chunk of a patch per K step; pixel-major, the 64-byte pieces of a 128-byte line belong to two K
                            // steps and every line crossed the fabric twice (PMC traffic 1.59 x the algorithmic bytes: the r256 layer ran at 5.6 TB/s)
    int x_planar8;         // the input map is chunk-planar
    int y_planar8;         // the output map is written chunk-planar
    int no_tstore;          // experiment knob: 1 = scattered 8-byte stores (no LDS-transposed epilogue)
    int row_walk;           // conv_stream A/B knob: row-major walk of the persistent workgroups (round 2) instead of down the columns
    half_t* y;              // output [B][Ho][Wo][Cout] fp16 (or)
    float* y32;             // output fp32, same layout
};

struct GemmParams {
    const half_t* a;  // [M][K]
    const half_t* w;  // [N][K]
    int M, N, K;
    const float* bias;  // [N] nullable
    int mode;           // 0: out16 = v ; 1: out16 = quickgelu(v) ; 2: x32 += v (in place) ; 3: out32 = v ; 4: out32 = lrelu(v)*sqrt2
    half_t* out16;
    float* out32;
    int ldo;
    int kpt;                // 0: w rows are K long.  > 0: w is a packed conv weight [K / kpt][N][kpt] (K steps never straddle a tap)
    long long w_tap_stride; // elements between the taps of such a weight
    int batch;              // 0/1: single problem; >1: blockIdx.z walks problems a_bs / w_bs / o_bs elements apart
    long long a_bs, w_bs, o_bs;
    int cand_rows;          // rows of M one candidate contributes (0: M does not scale with the population) and
    int cand_batch;         // 1: blockIdx.z walks candidates — the tile-width choice is made at the nominal population (GLASS_NOMINAL_POP)
    // implicit patch matrix (conv_gemm.hip, round 6; gemm_tiled only): row m = (b, oy, ox) of the conv grid, column k = (tap, ci) —
    // A[m][k] = x[b][oy * g_stride + ty - g_pad][ox * g_stride + tx - g_pad][ci] (zero outside the image), a = x, K steps never straddle a tap
    int g_on, g_h, g_w, g_hc, g_wc, g_stride, g_pad, g_ks, g_cin;
    long long g_xbs;        // elements between the images of x
    int ld;                 // 0: rows of a and w are K long.  > 0: their row stride (a K slice of longer rows: split-K as `batch` slices,
                            // a_bs = w_bs = K, raw partial sums to out32 + z * o_bs; gemm_tiled only.  With kpt: w_bs = 0, slice z starts at k = z * K of the tap walk)
};
