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
constexpr int TH = 8, PH = TH + 2;                      // tile rows, patch rows (the patch is 34 px wide: 32 + two halo columns)
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
//
// TRGB: the layer is the generator's last conv and its output feeds only toRGB (1x1 modulated conv 32 -> 3, no demod, + bias +
// the FIR-upsampled skip image of the previous block; stylegan2/models.py:852-870, 1004-1013, modules.py:580-602).  The
// activated fp16 tile is the B operand of that 1x1 conv exactly as it sits in the accumulator lanes — lane (px, kh) holds
// channels 8g + 4kh + q, and the MFMA's K order is free as long as the A operand (the 3 weight rows) uses the same order — so
// toRGB is 4 MFMAs per wave straight from registers: no LDS round trip, no feature-map store (64 bytes per pixel) and no
// separate toRGB pass re-reading it.  The weight rows are split hi + lo * 2^-11 in fp16 (rows 0-2 / 8-10 for the wave's first
// image row, 4-6 / 12-14 for the second: lane half kh then owns image row kh), so the product matches the fp32 pass's to ~1e-7.
template <bool FRGB, bool TRGB, bool TR = false>
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
    half_t* Cf = Cs + 32;                                        // [4][32]: fromRGB weight rows r, g, b and bias
    // TRGB re-uses the output-transposition area: two 16-row x 64-byte toRGB weight tables, the skip-image window
    // [3][5][17] fp32 (rows ty0/2 - 1 .., cols tx0/2 - 1 ..; zeros outside the image) and the 3 biases
    char* Tw = smem + W_BYTES + A_BYTES;
    float* Ys = (float*)(Tw + 2048);
    float* Tb = Ys + 256;
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

    // ---- the instruction diet (round 2): the kernel is bound by instruction ISSUE (three waves per SIMD, one instruction per
    // ~3.7 cycles per SIMD — the wave64 VALU rate), so everything per-tile that is uniform is kept uniform (scalar unit), and the
    // thread -> patch mapping makes everything per-thread tile-independent:
    //   * tile walk by a uniform cursor (b, ty, tx), advanced with compares — no integer divisions;
    //   * body of the patch (10 rows x 32 px, image columns tx0 .. tx0+31): vector k of thread t is row 2k + (t >> 7) (wave-
    //     uniform), pixel (t >> 2) & 31, 8-channel part t & 3 — its global address is a UNIFORM row base + (t & 127) * 16 bytes
    //     and its LDS address a per-thread constant + k * 4608: no per-vector address arithmetic, row masks are scalar;
    //   * the two halo columns (10 rows x 2 px x 4 parts = 80 vectors): one extra vector for threads 0..79;
    //   * FRGB: one thread per PIXEL (3 loads and one clamp per pixel instead of per 8-channel part): 256 + 84 pixels.
    struct Cur { int b, ty, tx; };
    auto cur_at = [&](int id) {
        Cur c;
        const int tpi = tiles_x * tiles_y;
        // the walk goes DOWN a tile column (tile row fastest): consecutive patches share 2 of their 10 rows, which the second one
        // finds in L2 (row-major, the vertical halo of every patch came from HBM / MALL: 6.0 GB fetched for 4.3 GB of input)
        c.b = id / tpi;
        const int trem = id - c.b * tpi;
        if (p.row_walk) { c.ty = trem / tiles_x; c.tx = trem - c.ty * tiles_x; }
        else { c.tx = trem / tiles_y; c.ty = trem - c.tx * tiles_y; }
        return c;
    };
    auto advance = [&](Cur& c) {
        if (p.row_walk) {
            if (++c.tx == tiles_x) { c.tx = 0; if (++c.ty == tiles_y) { c.ty = 0; ++c.b; } }
        } else if (++c.ty == tiles_y) {
            c.ty = 0;
            if (++c.tx == tiles_x) { c.tx = 0; ++c.b; }
        }
    };
    const Cur c_first = cur_at(first);
    auto pick = [&](const Cur& c, bool live) {   // a cursor past the range reads (and computes on) the first tile instead
        Cur r;
        r.b = live ? c.b : c_first.b; r.ty = live ? c.ty : c_first.ty; r.tx = live ? c.tx : c_first.tx;
        return r;
    };
    // Per-thread geometry is re-derived in each phase from an opaque copy of the thread id: as loop invariants these values
    // (and every address term the compiler derives from them) get hoisted and, at three workgroups per CU (168 VGPRs), spilled.
    //   hi: body row parity of this wave; body vector: byte offset tbyte within its image row segment, patch column bpc, LDS
    //   address lbody + k * (2 * PP * 64); halo vector (t < 80): patch row t >> 3, column 0 / 33;
    //   FRGB pixels: #0 = (t >> 5, (t & 31) + 1); #1 (t < 84) = rows 8, 9 of the body (t < 64) or halo pixel t - 64
#define GEO(tt)                                                                                                       \
    const int hi = __builtin_amdgcn_readfirstlane((tt) >> 7);                                                          \
    const unsigned tbyte = (unsigned)((tt) & 127) * 16u;                                                                \
    const int bpc = (((tt) >> 2) & 31) + 1;                                                                             \
    const int lbody = swa(hi, bpc, (tt) & 3);                                                                           \
    const int hpr = (tt) >> 3, hpc = (((tt) >> 2) & 1) * 33;                                                            \
    const int lhalo = swa(hpr, hpc, (tt) & 3);                                                                          \
    const int f1r = (tt) < 64 ? 8 + ((tt) >> 5) : ((tt) - 64) >> 1, f1c = (tt) < 64 ? ((tt) & 31) + 1 : (((tt) - 64) & 1) * 33; \
    (void)hi; (void)tbyte; (void)bpc; (void)lbody; (void)hpr; (void)hpc; (void)lhalo; (void)f1r; (void)f1c

    struct RSet { h8 a[FRGB ? 1 : 6]; float y3[FRGB ? 2 : 1][3]; float nz[2]; float ys[TRGB ? 3 : 1]; };   // one tile's loads in flight
    // The loads of a refill are issued ONE AT A TIME between the taps of the MFMA loop (a burst of 8 - 20 load instructions keeps
    // the CU's address unit busy for ~2000 cycles with all four waves stalled at issue — phase trace).
    // Loads are UNCONDITIONAL (clamped addresses) and masked to zero when they are written to LDS: a conditional load makes the
    // compiler wait for it at the join.
    auto load_part = [&](const Cur& c, RSet& R, int k) {
        const int ty0 = c.ty * TH, tx0 = c.tx * 32;
        const int t = opaque(threadIdx.x);
        GEO(t);
        if (FRGB) {
            const int hw = p.H * p.W;
            const float* yb = p.rgb_y + (long long)c.b * 3 * hw;                                     // uniform
            if (k == 0 || (k == 1 && t < 84)) {
                const int pr = k ? f1r : (t >> 5), pc = k ? f1c : (t & 31) + 1;
                const int iy = min(max(ty0 - 1 + pr, 0), p.H - 1), ix = min(max(tx0 - 1 + pc, 0), p.W - 1);
                const int off = iy * p.W + ix;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) R.y3[k][ch] = yb[ch * hw + off];
            }
        } else {
            const half_t* img = p.x + (long long)c.b * p.x_bstride;                                  // uniform
            if (k < 5) {
                const int row = min(max(ty0 - 1 + 2 * k + hi, 0), p.H - 1);                           // uniform
                const char* rowp = (const char*)(img + ((long long)row * p.W + tx0) * 32);           // uniform
                R.a[k] = *(const h8*)(rowp + tbyte);
            } else if (k == 5) {
                if (t < 80) {
                    const int iy = min(max(ty0 - 1 + hpr, 0), p.H - 1), ix = min(max(tx0 - 1 + hpc, 0), p.W - 1);
                    R.a[5] = *(const h8*)(img + ((long long)iy * p.W + ix) * 32 + (t & 3) * 8);
                }
            }
        }
        if (k == 6) {
            if (p.noise) {
                const float* nzp = p.noise + ((long long)(c.b / p.batch_size) * p.Ho + ty0) * p.Wo + tx0;   // uniform
                const int noff = ((t >> 6) & 3) * 2 * p.Wo + (t & 31);
                R.nz[0] = nzp[noff];
                R.nz[1] = nzp[noff + p.Wo];
            }
            if (TRGB && p.trgb_yprev) {   // skip-image window [3][5][17]: lane (r, col) = (t >> 5, t & 31) fetches its three colours
                const int r = t >> 5, col = t & 31;                                  // (masked when they are written to LDS)
                if (r < 5 && col < 17) {
                    const int h2 = p.Ho >> 1, w2 = p.Wo >> 1;
                    const int sy = max((ty0 >> 1) - 1 + r, 0), sx = max((tx0 >> 1) - 1 + col, 0);
                    const float* yp = p.trgb_yprev + (long long)c.b * 3 * h2 * w2 + sy * w2 + sx;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) R.ys[ch] = yp[ch * h2 * w2];
                }
            }
        }
    };
    auto load = [&](const Cur& c, RSet& R) {
#pragma unroll
        for (int k = 0; k <= 6; ++k) load_part(c, R, k);
    };

    int wb = -1;   // sample whose weights are resident in Ws
    // tile `id` at cursor cc (ids past `last` pad the loop: computed on the first tile's data, never stored); the refill of this
    // register set (tile id + 2, cursor lc) is threaded through the MFMA taps
    auto step = [&](int id, const Cur& cc, const Cur& lc, RSet& R) {
        const bool valid = id < last;
        const int b = cc.b, ty0 = cc.ty * TH, tx0 = cc.tx * 32;
        STRACE(0);
        __syncthreads();                           // every wave is done reading As / Ws of the previous tile
        STRACE(1);
        if (wb != b && (TRGB || wb < 0 || p.w_bstride != 0 || p.dscale || p.shift || p.sn16)) {
            const half_t* wsrc = p.w + (long long)b * p.w_bstride;   // [9][32][32]
            for (int u = t; u < 9 * 32 * 4; u += 256)
                *(h8*)(Ws + swz(u >> 2, u & 3)) = *(const h8*)(wsrc + (long long)(u >> 2) * 32 + (u & 3) * 8);
            if (t < 32) {
                Cd[t] = p.dscale ? p.dscale[(long long)b * p.ds_stride + t] : 1.f;
                Cb[t] = (p.bias ? p.bias[t] : 0.f) + (p.shift ? p.shift[(long long)b * p.ds_stride + t] : 0.f);
                Cs[t] = p.sn16 ? p.sn16[(long long)b * p.sn_stride + t] : (half_t)1.f;
            }
            if (TRGB) {
                // table entry (tab, n, hidx): row n = colour (n & 3; 3 = unused) | image row (n & 4) | lo part (n & 8); half hidx
                // = MFMA kk (hidx >> 4), lane half kh ((hidx >> 3) & 1), element j: the channel lane half kh holds in
                // accumulator quad g = 2 kk + (j >> 2), i.e. 8 g + 4 kh + (j & 3)
                const float sm = p.trgb_smax[(long long)b * p.trgb_smax_stride];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = t + 256 * u, tab = e >> 9, n = (e >> 5) & 15, hidx = e & 31;
                    const int col = n & 3, j = hidx & 7;
                    const int ch = 8 * (2 * (hidx >> 4) + (j >> 2)) + 4 * ((hidx >> 3) & 1) + (j & 3);
                    half_t val = (half_t)0.f;
                    if (col < 3 && ((n >> 2) & 1) == tab) {
                        const float w = p.trgb_w[col * 32 + ch] * p.trgb_sn[(long long)b * p.trgb_sn_stride + ch] * sm;
                        const half_t hv = (half_t)w;
                        val = (n & 8) ? (half_t)((w - (float)hv) * 2048.f) : hv;
                    }
                    ((half_t*)Tw)[e] = val;
                }
                if (t < 3) Tb[t] = p.trgb_b[t];
            }
            wb = b;
            __syncthreads();
        }
        const int t = opaque(threadIdx.x);
        GEO(t);
        // interior tiles need no bounds test at all (uniform)
        const bool border = ty0 == 0 || tx0 == 0 || ty0 + TH >= p.H || tx0 + 32 >= p.W;
        const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        if (FRGB) {
            // fromRGB of this thread's pixel(s): clamp once, then the 32 channels as four packed-fp16 vectors
            h8 fw[4][4];
#pragma unroll
            for (int part = 0; part < 4; ++part)
#pragma unroll
                for (int r = 0; r < 4; ++r) fw[part][r] = *(const h8*)(Cf + r * 32 + part * 8);
            auto pixel = [&](int k, bool masked) {
                const int pr = k ? f1r : (t >> 5), pc = k ? f1c : (t & 31) + 1;
                const int iy = ty0 - 1 + pr, ix = tx0 - 1 + pc;
                float c3[3];
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) c3[ch] = fminf(fmaxf((R.y3[k][ch] + 1.f) * 0.5f, 0.f), 1.f) * 2.f - 1.f;
                const half_t h0 = (half_t)c3[0], h1 = (half_t)c3[1], h2 = (half_t)c3[2];
                const int lrow = (pr * PP + pc) << 6, key = (pc >> 2) & 3;
                h8 a[4];
#pragma unroll
                for (int part = 0; part < 4; ++part) {
                    const h8 z = fw[part][0] * h0 + fw[part][1] * h1 + fw[part][2] * h2 + fw[part][3];          // v_pk_fma_f16
                    a[part] = __builtin_elementwise_max(z, z * (half_t)0.2f);
                }
                if (p.rgb_x_out && valid && pr >= 1 && pr <= TH && pc >= 1 && pc <= 32) {   // the map itself, for an un-fused skip path
                    half_t* xo = p.rgb_x_out + (((long long)b * p.H + iy) * p.W + ix) * 32;
#pragma unroll
                    for (int part = 0; part < 4; ++part) *(h8*)(xo + part * 8) = a[part];
                }
                const bool ok = !masked || (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W);
#pragma unroll
                for (int part = 0; part < 4; ++part) *(h8*)(As + lrow + ((part ^ key) << 4)) = ok ? a[part] : zero;
            };
            if (border) {
                pixel(0, true);
                if (t < 84) pixel(1, true);
            } else {
                pixel(0, false);
                if (t < 84) pixel(1, false);
            }
        } else {
            if (!border && !p.sn16) {
                // the common case (interior tile of a layer whose weights carry the style): registers -> LDS, nothing else
#pragma unroll
                for (int k = 0; k < 5; ++k) *(h8*)(As + lbody + k * (2 * PP * 64)) = R.a[k];
                if (t < 80) *(h8*)(As + lhalo) = R.a[5];
            } else {
                const h8 sh = *(const h8*)(Cs + (t & 3) * 8);                                        // 1.0 without a style
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    h8 a = R.a[k];
                    if (border && (unsigned)(ty0 - 1 + 2 * k + hi) >= (unsigned)p.H) a = zero;      // uniform condition
                    *(h8*)(As + lbody + k * (2 * PP * 64)) = a * sh;
                }
                if (t < 80) {
                    h8 a = R.a[5];
                    if (border && ((unsigned)(ty0 - 1 + hpr) >= (unsigned)p.H || (unsigned)(tx0 - 1 + hpc) >= (unsigned)p.W)) a = zero;
                    *(h8*)(As + lhalo) = a * sh;
                }
            }
        }
        if (TRGB && p.trgb_yprev) {
            const int r = t >> 5, col = t & 31;
            if (r < 5 && col < 17) {
                const bool ok = (ty0 >> 1) - 1 + r >= 0 && (tx0 >> 1) - 1 + col >= 0;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) Ys[ch * 85 + r * 17 + col] = ok ? R.ys[ch] : 0.f;
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
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const int tap = ty * 3 + tx;
                if (tap <= 6) load_part(lc, R, tap);            // refill (two tiles stay in flight), one load per tap
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
        __builtin_amdgcn_s_setprio(0);

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
        if (TRGB) {
            f16x rgb;
#pragma unroll
            for (int q = 0; q < 16; ++q) rgb[q] = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float nz = i ? nz1 : nz0;
                h4 va[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int o = 8 * g + 4 * kh;
                    const f4 d = *(const f4*)(Cd + o), bb = *(const f4*)(Cb + o);
                    h4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = (half_t)(acc[i][g * 4 + q] * d[q] + (nz + bb[q]));
                    va[g] = __builtin_elementwise_max(v * k1, v * k2);
                }
                const char* Tt = Tw + i * 1024 + (lr & 15) * 64 + kh * 16;
                rgb = mfma32(*(const h8*)Tt, __builtin_shufflevector(va[0], va[1], 0, 1, 2, 3, 4, 5, 6, 7), rgb);
                rgb = mfma32(*(const h8*)(Tt + 32), __builtin_shufflevector(va[2], va[3], 0, 1, 2, 3, 4, 5, 6, 7), rgb);
            }
            // lane (lr, kh) now owns pixel (row wave * 2 + kh, column lr) of the tile: registers 0-2 hi part, 4-6 lo part
            float r3[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) r3[ch] = Tb[ch] + (rgb[ch] + rgb[4 + ch] * (1.f / 2048.f));
            if (p.trgb_yprev) {
                // Upsample (zero-insert, pad [3,1], 4x4 FIR * 4): out[2m] = .75 x[m-1] + .25 x[m], out[2m+1] = .25 x[m-1] + .75 x[m]
                const float wy0 = kh ? 0.25f : 0.75f, wx0 = (lr & 1) ? 0.25f : 0.75f;
                const float* yw = Ys + wave * 17 + (lr >> 1);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    float sacc = 0.f;
                    sacc += wy0 * wx0 * yw[ch * 85];
                    sacc += wy0 * (1.f - wx0) * yw[ch * 85 + 1];
                    sacc += (1.f - wy0) * wx0 * yw[ch * 85 + 17];
                    sacc += (1.f - wy0) * (1.f - wx0) * yw[ch * 85 + 18];
                    r3[ch] += sacc;
                }
            }
            if (valid) {
                const long long hw = (long long)p.Ho * p.Wo;
                float* yo = p.trgb_yout + (long long)b * 3 * hw + (long long)(ty0 + wave * 2 + kh) * p.Wo + tx0 + lr;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) yo[ch * hw] = r3[ch];
            }
        } else {
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
        }
    };

    // two register sets; no exit between the two steps (hipcc's wait-count merge at the loop header otherwise stops counting
    // the other set's refill as younger and every patch wait drains the queue): an odd tile count is padded
    RSet r0, r1;
    Cur cc = c_first, lc = c_first;     // compute cursor (tile id) and load cursor (tile id + 2)
    load(pick(lc, true), r0);
    advance(lc);
    load(pick(lc, first + 1 < last), r1);
    advance(lc);
    for (int id = first; id < last; id += 2) {
        step(id, pick(cc, true), pick(lc, id + 2 < last), r0);
        advance(cc); advance(lc);
        step(id + 1, pick(cc, id + 1 < last), pick(lc, id + 3 < last), r1);
        advance(cc); advance(lc);
    }
}

static int stream_slots() {
    static DevOnce once;
    once.run([&] {
        (void)hipFuncSetAttribute((const void*)conv_stream_kernel<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv_stream_kernel<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv_stream_kernel<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    });
    return glass_cu_count() * 3;      // 168 VGPRs, 51.5 KB of LDS: three workgroups per CU
}

bool conv_stream_applies(const ConvParams& p) {
    static const bool off = glass_knob("GLASS_NO_STREAM") != nullptr;   // experiment knob
    const bool trgb = p.trgb_yout != nullptr;
    if (!glass_lds_fits(LDS_BYTES)) return false;
    if (p.x_planar8 || p.y_planar8 || p.x_planar32) return false;   // chunk-planar maps (common.h): not implemented here
    if (off || p.up || p.xs_out || p.y32 || (!p.y && !trgb) || p.KS != 3 || p.stride != 1 || p.pad != 1 || (p.sn && !p.sn16)) return false;
    const bool frgb = p.rgb_y != nullptr;
    if (frgb && (!p.rgb_w || !p.rgb_b || (!p.rgb_x_out && !p.rgb_xs_out) || p.sn || trgb)) return false;
    if (trgb && (!p.trgb_w || !p.trgb_b || !p.trgb_sn || !p.trgb_smax || p.Ho != p.Hc || p.Wo != p.Wc)) return false;
    if (p.Cin != 32 || p.Neff != 32 || p.Cout != 32 || p.res || p.pre_shift || p.res_cs || p.res_up) return false;
    if (p.in_up || (frgb && p.shift)) return false;
    if (p.Wc % 32 != 0 || p.Hc % TH != 0 || p.W >= 256 * 32 || (!frgb && p.x_bstride == 0 && p.B > 1)) return false;
    if ((long long)p.H * p.W * p.Cin >= (1LL << 31)) return false;
    // streaming only pays with many tiles per workgroup — judged at the nominal population (common.h), not at this launch's B:
    // the choice between this kernel and conv_tiled (different epilogue rounding) must not depend on chunking or sharding
    const long long PT_nominal = (long long)GLASS_NOMINAL_POP * (p.Wc / 32) * (p.Hc / TH);
    return PT_nominal >= (long long)stream_slots() * 6;
}

const char* launch_conv_stream(const ConvParams& p0, hipStream_t st) {
    if (!conv_stream_applies(p0)) return nullptr;
    ConvParams p = p0;
    static const bool row_walk = glass_knob("GLASS_ROW_WALK") != nullptr;      // A/B knob: round 2's row-major tile walk
    // measured (same box, column vs row walk): <torgb> 1821 vs 1844 us, conv_down 1762 vs 1823 us, <fromrgb> 2593 vs 2530 us — the
    // planar fp32 image the fromRGB form reads is friendlier to the row-major walk
    p.row_walk = (row_walk || p.rgb_y) ? 1 : 0;
    const bool trgb = p.trgb_yout != nullptr, frgb = p.rgb_y != nullptr;
    const int tiles_x = p.Wc / 32, tiles_y = p.Hc / TH;
    const int PT = p.B * tiles_x * tiles_y;
    const int slots = stream_slots();
    const int per_block = (PT + slots - 1) / slots;
    const int grid = (PT + per_block - 1) / per_block;
    const char* name = frgb ? "conv_stream_kernel<fromrgb>" : trgb ? "conv_stream_kernel<torgb>" : "conv_stream_kernel";
#ifdef GLASS_DEV_TRACE      // dev build (make TRACE=1): traced instance, one launch, timestamps to a file; synchronises, single engine only
    if (const char* trace_path = getenv("GLASS_STREAM_TRACE")) {       // dev tool: traced instance, one launch, timestamps to a file
        unsigned long long* dtr = nullptr;
        (void)hipMalloc(&dtr, 64 * 8 * 4 * sizeof(unsigned long long));
        (void)hipMemset(dtr, 0, 64 * 8 * 4 * sizeof(unsigned long long));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stream_trace), &dtr, sizeof dtr);
        (void)hipFuncSetAttribute((const void*)conv_stream_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv_stream_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv_stream_kernel<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (frgb) hipLaunchKernelGGL((conv_stream_kernel<true, false, true>), dim3(grid), dim3(256), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_block);
        else if (trgb) hipLaunchKernelGGL((conv_stream_kernel<false, true, true>), dim3(grid), dim3(256), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_block);
        else hipLaunchKernelGGL((conv_stream_kernel<false, false, true>), dim3(grid), dim3(256), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_block);
        static unsigned long long hb[64 * 8 * 4];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hb, dtr, sizeof hb, hipMemcpyDeviceToHost);
        (void)hipFree(dtr);
        if (FILE* f = fopen(trace_path, "a")) {
            fprintf(f, "# %s: tile phase t[wave0..3]; phases 0 enter, 1 after sync, 2 patch staged, 3 after sync, 4 refill issued, "
                       "5 MFMAs done, 6 skip by-product done, (next 0) epilogue done; per_block=%d\n", name, per_block);
            for (int i = 0; i < 64 && i < per_block; ++i)
                for (int ph = 0; ph < 7; ++ph) {
                    fprintf(f, "%d %d", i, ph);
                    for (int w = 0; w < 4; ++w) fprintf(f, " %llu", hb[(i * 8 + ph) * 4 + w] - hb[0]);
                    fprintf(f, "\n");
                }
            fclose(f);
        }
        return name;
    }
#endif
    if (frgb) hipLaunchKernelGGL((conv_stream_kernel<true, false, false>), dim3(grid), dim3(256), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_block);
    else if (trgb) hipLaunchKernelGGL((conv_stream_kernel<false, true, false>), dim3(grid), dim3(256), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_block);
    else hipLaunchKernelGGL((conv_stream_kernel<false, false, false>), dim3(grid), dim3(256), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_block);
    return name;
}
