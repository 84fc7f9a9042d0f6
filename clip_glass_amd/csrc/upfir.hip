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
#include <stdlib.h>
#include <algorithm>
#include <mutex>
#include <vector>

#define ROWB 80
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

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


// =====================================================================================================================
// upfir2 (round 3): the same arithmetic on a different tile geometry.  Per-launch MFMA work of upfir_kernel ran at 0.77-0.92
// PFLOP/s, but only 33 % (r64) / 48 % (r128) / 58 % (r256) of it was useful: 8 x 32 computed per 6 x 30 kept (t halo), 60-wide
// output tiles over 64 / 128 / 256 columns, 12-row tiles over 2^k rows.  Two changes, both pure index math:
//   * ROLLING STRIPS: a workgroup walks S consecutive 8-row steps down the image and keeps the FIR's sliding window (three
//     horizontally filtered rows per thread) in registers, so only a segment's first step recomputes the vertical t halo:
//     16 output rows per 8 computed rows instead of 12.
//   * VIRTUAL IMAGE GRID (shared-weight layers): the NXI x NYI candidates of a launch are laid side by side with ONE zero row /
//     column between them (pitch W + 1).  That zero column is exactly what the transposed conv needs at an image edge
//     (t[2W] = x[W-1] w[2] + 0 * w[0], t[-1] = 0), so tiles and strips run straight across image boundaries and the
//     quantisation loss of a 60-wide tile over a 64-wide image (53 % useful) becomes 32 / 33.  Every per-sample operand
//     (style, demodulation, consumer style, noise plane) is looked up per lane / per row from small LDS tables filled at
//     the top of each step; a tile touches at most 4 x 2 images (W >= 16, H >= 8).
//   * work order: XCD (block id % 8) owns a fixed group of n tiles (weights of a group stay in its 4 MiB L2) and a slice of
//     the pixel work (the 4.7 MB of 512 x 512 weights were re-streamed per pixel tile, 15-25x read over-fetch).
// =====================================================================================================================
struct UpGeo {
    int NTn, ngroups;          // 32-wide n tiles; XCD groups the n tiles are dealt to (divides 8 and NTn)
    int tiles_x, n_seg, S;     // 60-column tiles per virtual row, segments per grid, 8-row steps per segment
    int NXI, NYI, n_grids;     // candidates per virtual grid (x, y), grids per launch
    int WT;                    // pixel work items = n_grids * n_seg * tiles_x
    unsigned invPX, invPY;     // ceil(2^32 / (W + 1)), ceil(2^32 / (H + 1)): exact n / pitch for the ranges used here
    unsigned inv2PX, inv2PY;   // same for the output pitches 2 (W + 1), 2 (H + 1)
    int prefetch;              // GRID = false: fetch the next step's stage-0 operands under this step's FIR (A/B knob)
#ifdef GLASS_AB_KNOBS
    int ablate;                // developer build only (make AB=1, GLASS_UPFIR_ABLATE): timing experiments that switch phases of the single-image
                               // instance off — 1 MFMAs, 2 T write + FIR + stores, 4 global stores, 8 FIR arithmetic, 16 operand loads, 32 weight loads after a
                               // segment's first step, 64 patch loads, 128 weight LDS writes after the first step (WRONG RESULTS); 256: no request of the next step's stage 0
                               // behind the FIR (correct results: the r04 order, addresses derived and loads issued after the step barrier).
#endif
};
#if defined(U_ABLATE_CT)        // compile-time bits (tools/build_ablations.sh): a run-time bit costs branches and lets nothing be deleted (DESIGN "Round 6")
#define U_ABL(bit) (!GRID && (((U_ABLATE_CT) & (bit)) != 0))
#elif defined(GLASS_AB_KNOBS)
#define U_ABL(bit) (!GRID && (g.ablate & (bit)))
#else
#define U_ABL(bit) false
#endif

namespace {
constexpr int U_PH = 9, U_PW = 33;
constexpr int U_NVB = 9 * 32 * 4, U_NB = (U_NVB + 255) / 256;          // 1152 -> 5
constexpr int U_A_BYTES = ((U_PH * U_PW * ROWB + 15) / 16) * 16;       // 23760
// Step geometry by RW = m rows per wave (r05).  RW = 2 is the round-3 / round-4 kernel: 8 m rows per step, T tile [16][64], 81.6 KB of LDS and
// ~250 VGPRs -> TWO workgroups per CU.  RW = 1 halves the step (4 m rows, T [8][64], 64 accumulator registers): 51 KB and <= 168 VGPRs ->
// THREE workgroups per CU, for the single-image instance whose step is made of exposed memory phases ("Round 5": a
// third workgroup is one more chance that somebody computes while the others wait); it pays with the weights staged per 4 rows instead of 8.
template <int RW>
struct UStep {
    static constexpr int MR = 4 * RW, TR = 2 * MR, PH = MR + 1;
    static constexpr int NVA = PH * U_PW * 4, NA = (NVA + 255) / 256;                   // RW 2: 1188 -> 5;  RW 1: 660 -> 3
    static constexpr int A_BYTES = ((PH * U_PW * ROWB + 15) / 16) * 16;                 // 23760 / 13200
    static constexpr int T_BYTES = TR * 64 * 64;                                        // 65536 / 32768
    static constexpr int STAGE = A_BYTES + 9 * 32 * ROWB;                               // 46800 / 36240
    static constexpr int R0 = ((T_BYTES > STAGE ? T_BYTES : STAGE) + 1023) / 1024 * 1024;   // 65536 / 36864: T overlays the staging images
    static constexpr int OFF_LNZ = R0;                                                  // GRID = false: noise table [TR][60] fp32 behind the T tile
    static constexpr int OFF_HS = OFF_LNZ + TR * 60 * 4;                                // [256 threads][3 rows] h8: the FIR window between steps
    static constexpr int LDS = OFF_HS + 256 * 48;                                       // 81664 / 51072
};
// per-step constant tables live in the tail of the T tile that the staging images do not reach (46800 ..): they are read
// into registers before the T tile is written
constexpr int U_OFF_STY = 46848;                   // [8 images][Cin] fp16 style rows (Cin <= 512)
constexpr int U_OFF_DS = U_OFF_STY + 8 * 1024;     // [8][32] fp32 demodulation
constexpr int U_OFF_PS = U_OFF_DS + 8 * 128;       // [8][32] fp16 consumer style
constexpr int U_OFF_BI = U_OFF_PS + 8 * 64;        // [32] fp32 bias
constexpr int U_OFF_NZ = U_OFF_BI + 128;           // [16 rows][60 cols] fp32 noise * strength
static_assert(UStep<2>::LDS == 81664 && 2 * UStep<2>::LDS <= 163840 && 3 * UStep<1>::LDS <= 163840, "two / three workgroups per CU");
static_assert(U_OFF_NZ + 16 * 60 * 4 <= 65536 && U_OFF_STY >= U_A_BYTES + 9 * 32 * ROWB && UStep<2>::A_BYTES == U_A_BYTES, "constant tables fit behind the staging images");
__device__ __forceinline__ int u_opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// T tile of upfir2 (r04): a row's 64 columns sit de-interleaved — column x at position (x & 1) * 32 + (x >> 1), 64 B each, the 16-byte
// chunk index XORed with (x >> 2) & 3.  The MFMA lanes write columns 2 lr + c: with the columns in order, the 16 lanes of one ds_write_b64
// group were 128 B apart and hit the same eight banks four deep (16 LDS cycles per instruction where 4 is ideal); 64 B apart and
// chunk-rotated every second lane they are two deep (8).  The FIR's ds_read_b128 groups (lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31})
// stay conflict-free when each group holds four columns of distinct (x >> 1) & 3: the FIR thread of column index ci takes column
// u_fir_col(ci) (2 <-> 3 and 4 <-> 5 swapped within every 8).  tests/emu_ops.py carries the same index maps.
// staging order of the padded 80-byte rows (as conv_tiled.hip's tl_row): the 8 lanes of a ds_write_b128 bank group hold rows r and r + 4
// (disjoint banks) instead of r and r + 1 (the second row wraps onto the first one's banks: 16 LDS cycles where 8 is the floor)
__device__ __forceinline__ int u_stage_row(int r) { return (r & ~7) | ((r & 7) >> 1) | ((r & 1) << 2); }
__device__ __forceinline__ int u_tpos(int x) { return (x & 1) * 32 + (x >> 1); }
__device__ __forceinline__ int u_fir_col(int ci) { return ci ^ (((ci >> 2) ^ (ci >> 1)) & 1); }
}  // namespace

// GRID = true: shared weights, candidates on a virtual grid, per-image operands through the LDS tables.
// GRID = false: per-sample weights (which carry style and demodulation): one candidate per grid, no tables, no index
// divisions — the 512^2 / 1024^2 layers are instruction-issue bound, every instruction per step counts.
template <bool GRID, int RW = 2, bool BS = false>
__global__ __launch_bounds__(256, 4 - RW) void upfir2_kernel(ConvParams p, UpGeo g) {
    static_assert(RW == 2 || !GRID, "the half-height step exists for the single-image instance only");
    static_assert(!BS || !GRID, "the buffer-store form exists for the single-image instance only");
    using US = UStep<RW>;
    constexpr int MR = US::MR, TR = US::TR;
    constexpr int PW = U_PW, NA = US::NA, NVB = U_NVB, NB = U_NB, A_BYTES = US::A_BYTES;
    constexpr int U_OFF_LNZ = US::OFF_LNZ, U_OFF_HS = US::OFF_HS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + A_BYTES;
    half_t* T = (half_t*)smem;                  // [TR][64][32] fp16, overlays the staging area afterwards
    // workgroup barrier of the K loop / T overlay.  BS (r05 experiment, single-image instance): LDS-only — a __syncthreads() also waits
    // vmcnt(0), i.e. for the previous step's 16 row stores, at the FIRST barrier of the next step; the counted waits hipcc places in front
    // of each operand's first use are all the K loop needs from global memory
    auto kbar = [&]() {
        if constexpr (BS) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            __syncthreads();
        }
    };

    // ---- work item: XCD xcd owns n-tile group xcd % G and pixel slice xcd / G ---------------------------------------
    const int id = blockIdx.x;
    const int xcd = id & 7, rest = id >> 3;
    const int G = g.ngroups, NTg = g.NTn / G, Pp = 8 / G;
    const int nt = (xcd % G) * NTg + rest % NTg;
    const int wt = (rest / NTg) * Pp + xcd / G;
    if (wt >= g.WT) return;
    const int txi = wt % g.tiles_x, seg = (wt / g.tiles_x) % g.n_seg, gi = wt / (g.tiles_x * g.n_seg);
    const int n0 = nt * 32;
    const int img0 = gi * g.NXI * g.NYI;
    const int PX = p.W + 1, PY = p.H + 1;
    const int mx0 = txi * 30 - 1;                           // virtual m column of lane 0
    const int Y0 = seg * ((TR - 4) + TR * (g.S - 1));       // first virtual output row of the segment
    const int out_rows = 2 * PY * g.NYI - 2;                // virtual output rows that exist
    const int ixi0 = GRID ? (int)__umulhi((unsigned)max(mx0 - 1, 0), g.invPX) : 0;     // first image column the tile's patch touches

    const half_t* wb = p.w_up + (p.w_bstride ? (long long)img0 * p.w_bstride : 0LL);   // per-sample weights: one image per grid
    // weight vector u = t + 256 k sits at tap (u >> 7), row u_stage_row(u >> 2) & 31: k only moves the tap, by a uniform 2 k * Cout * Cin
    // (the last, half-empty round re-reads round 3's vector in threads >= 128)
    const int b_goff0 = ((threadIdx.x >> 7) * p.Cout + n0 + (u_stage_row(threadIdx.x >> 2) & 31)) * p.Cin + (threadIdx.x & 3) * 8;
    const int b_step = 2 * p.Cout * p.Cin;
    // FIR threads keep a sliding window of horizontally filtered t rows that runs on from step to step: its three live rows wait
    // in a thread-private LDS slot during the K loop (16 registers the K loop needs, next to the prefetched operands)
    char* hs_slot = smem + U_OFF_HS + threadIdx.x * 48;

    int a_goff[NA];
    int okm = 0, selm = 0;
    h8 ra[NA], rb[NB];
    bool have = false;          // (uniform) stage 0 of this step was fetched during the previous step's FIR phase (GRID = false)
    for (int step = 0; step < g.S; ++step) {
        const int o_first = Y0 + (step ? TR * step - 4 : 0);   // first output row this step emits
        if (o_first >= out_rows) break;                         // (uniform) the segment runs off the grid
        const int my0 = (Y0 >> 1) - 1 + MR * step;              // virtual m row of the step's first computed row
        const int iyi0 = GRID ? (int)__umulhi((unsigned)max(my0 - 1, 0), g.invPY) : 0;
        // table entry sel = dy * 4 + dx  <->  image (iyi0 + dy, ixi0 + dx), clamped to an image that exists (its values then
        // only ever meet masked operands, but they must be finite)
        auto sel_img = [&](int sel) {
            const int iyi = min(iyi0 + (sel >> 2), g.NYI - 1), ixi = min(ixi0 + (sel & 3), g.NXI - 1);
            return min(img0 + iyi * g.NXI + ixi, p.B - 1);
        };

        // ---- staging geometry of a step (re-derived from an opaque thread id: nothing here may be hoisted and kept) ----
        auto aim = [&](int my0, int iyi0) {
            okm = 0;
            selm = 0;
            const int t = u_opaque(threadIdx.x), part = t & 3;
#pragma unroll
            for (int k = 0; k < NA; ++k) {
                const int pix = u_stage_row(t >> 2) + 64 * k;
                const int pr = pix / PW, pc = pix - pr * PW;
                const int vy = my0 - 1 + pr, vx = mx0 - 1 + pc;
                if (GRID) {
                    const int iyi = (int)__umulhi((unsigned)max(vy, 0), g.invPY), ixi = (int)__umulhi((unsigned)max(vx, 0), g.invPX);
                    const int iy = vy - iyi * PY, ix = vx - ixi * PX;
                    const int img = img0 + iyi * g.NXI + ixi;
                    const bool ok = pix < US::PH * U_PW && vy >= 0 && vx >= 0 && iy < p.H && ix < p.W && iyi < g.NYI && ixi < g.NXI && img < p.B;
                    a_goff[k] = ok ? img * (int)p.x_bstride + (iy * p.W + ix) * p.Cin + part * 8 : part * 8;
                    okm |= (ok ? 1 : 0) << k;
                    selm |= ((((iyi - iyi0) << 2) + (ixi - ixi0)) & 7) << (3 * k);
                } else {
                    const bool ok = pix < US::PH * U_PW && vy >= 0 && vx >= 0 && vy < p.H && vx < p.W;
                    a_goff[k] = ok ? img0 * (int)p.x_bstride + (vy * p.W + vx) * p.Cin + part * 8 : part * 8;
                    okm |= (ok ? 1 : 0) << k;
                }
            }
        };
        // interior tile of a single-image grid whose weights carry the style: registers -> LDS as they are
        const bool plain = !GRID && my0 >= 1 && mx0 >= 1 && my0 + MR - 1 < p.H && mx0 + 31 < p.W;

        auto load_a = [&](int c0) {
            if (U_ABL(16) || U_ABL(64)) return;
#pragma unroll
            for (int k = 0; k < NA; ++k) ra[k] = *(const h8*)(p.x + a_goff[k] + c0);
        };
        auto load_b = [&](int c0) {
            if (U_ABL(16) || (U_ABL(32) && step > 0)) return;
            const half_t* wp = wb + u_opaque(b_goff0) + c0;       // (opaque: five hoisted 64-bit pointers would be spilled)
#pragma unroll
            for (int k = 0; k < NB; ++k) rb[k] = *(const h8*)(wp + ((k == NB - 1 && threadIdx.x >= 128) ? k - 1 : k) * b_step);
        };
        auto store_a = [&](int c0) {
            const int t = threadIdx.x, part = t & 3;
            const int trow = u_stage_row(t >> 2);
            char* ab = As + trow * ROWB + part * 16;              // vector k sits 64 rows further down
            if (plain) {
#pragma unroll
                for (int k = 0; k < NA; ++k)
                    if (k < NA - 1 || trow + 64 * k < US::PH * U_PW) *(h8*)(ab + k * 64 * ROWB) = ra[k];
            } else {
                const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
                const char* sty = smem + U_OFF_STY + (c0 + part * 8) * 2;
#pragma unroll
                for (int k = 0; k < NA; ++k) {
                    if (k < NA - 1 || trow + 64 * k < US::PH * U_PW) {
                        h8 a = ((okm >> k) & 1) ? ra[k] : zero;
                        if (GRID && p.sn16) a = a * *(const h8*)(sty + ((selm >> (3 * k)) & 7) * 1024);   // the vector's own image's style
                        *(h8*)(ab + k * 64 * ROWB) = a;
                    }
                }
            }
        };
        auto store_b = [&]() {
            if (U_ABL(128) && step > 0) return;
            const int t = threadIdx.x, part = t & 3;
            char* bb = Bs + u_stage_row(t >> 2) * ROWB + part * 16;
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (k < NB - 1 || t + 256 * k < NVB) *(h8*)(bb + k * 64 * ROWB) = rb[k];
        };

        if (GRID) {
            __syncthreads();   // the previous step's FIR is done with the T tile (and with the tables behind it)
        } else {
            // LDS-only barrier: a __syncthreads() here would also wait (vmcnt(0)) for the previous FIR's 16 row stores and for the
            // stage-0 operands that were fetched under it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (!have) aim(my0, iyi0);
        if (GRID) {
            // ---- constant tables of this step: ONE batch of global loads, issued ahead of stage 0's operands --------------
            const int t = u_opaque(threadIdx.x);
            const int cin8 = p.Cin >> 3;
            h8 sv[2];
            float dsv = 1.f, biv = 0.f, nzv4[4];
            half_t psv = (half_t)1.f;
            if (p.sn16) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int e = min(t + 256 * u, 8 * cin8 - 1);
                    const int sel = e / cin8, piece = e - sel * cin8;
                    sv[u] = *(const h8*)(p.sn16 + (long long)sel_img(sel) * p.sn_stride + piece * 8);
                }
            }
            const int sel = t >> 5, ch = t & 31;
            if (p.dscale) dsv = p.dscale[(long long)sel_img(sel) * p.ds_stride + n0 + ch];
            if (p.post_scale16) psv = p.post_scale16[(long long)sel_img(sel) * p.post_stride + n0 + ch];
            if (p.bias && t < 32) biv = p.bias[n0 + t];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                nzv4[u] = 0.f;
                if (p.noise) {
                    const int e = min(t + 256 * u, 959);
                    const int r = e / 60, c = e - r * 60;
                    const int ovy = max(Y0 + TR * step + r - 4, 0), ovx = txi * 60 + c;
                    const int iyi = (int)__umulhi((unsigned)ovy, g.inv2PY), ixi = (int)__umulhi((unsigned)ovx, g.inv2PX);
                    const int oy = min(ovy - iyi * 2 * PY, p.Ho - 1), ox = min(ovx - ixi * 2 * PX, p.Wo - 1);
                    const int img = min(img0 + min(iyi, g.NYI - 1) * g.NXI + min(ixi, g.NXI - 1), p.B - 1);
                    nzv4[u] = p.noise[((long long)(img / p.batch_size) * p.Ho + oy) * p.Wo + ox];
                }
            }
            load_a(0);
            load_b(0);
            if (p.sn16) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int e = t + 256 * u;
                    if (e < 8 * cin8) {
                        const int sel = e / cin8, piece = e - sel * cin8;
                        *(h8*)(smem + U_OFF_STY + sel * 1024 + piece * 16) = sv[u];
                    }
                }
            }
            *(float*)(smem + U_OFF_DS + t * 4) = dsv;
            *(half_t*)(smem + U_OFF_PS + t * 2) = psv;
            if (t < 32) *(float*)(smem + U_OFF_BI + t * 4) = biv;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (t + 256 * u < 960) *(float*)(smem + U_OFF_NZ + (t + 256 * u) * 4) = p.noise_strength * nzv4[u];
            __syncthreads();   // style rows visible to stage 0's store_a
        } else if (!have) {
            load_a(0);
            load_b(0);
        }
        have = false;

        f16x acc[RW][4];   // [m-row of this wave][parity class ry*2+rx]
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        // tap (ky,kx) feeds parity class (ky&1, kx&1) and reads x[m - (ky>>1), n - (kx>>1)] (see upfir_kernel)
        auto mfma_block = [&]() {
            if (U_ABL(1)) return;
            const int tm = u_opaque(threadIdx.x), lr = tm & 31, kh = (tm >> 5) & 1, wave = tm >> 6;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int ay = 0; ay < 2; ++ay) {
#pragma unroll
                    for (int ax = 0; ax < 2; ++ax) {
                        h8 wf[2][2];
#pragma unroll
                        for (int ky = ay * 2; ky < (ay ? 3 : 2); ++ky)
#pragma unroll
                            for (int kx = ax * 2; kx < (ax ? 3 : 2); ++kx)
                                wf[ky & 1][kx & 1] = *(const h8*)(Bs + ((ky * 3 + kx) * 32 + lr) * ROWB + kk * 32 + kh * 16);
#pragma unroll
                        for (int i = 0; i < RW; ++i) {
                            const int prow = wave * RW + i + 1 - ay;
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
            __builtin_amdgcn_s_setprio(0);
        };
        const int n_stages = p.Cin >> 5;
        for (int s = 0; s + 1 < n_stages; ++s) {
            if (s > 0) kbar();
            store_a(s * 32);
            store_b();
            kbar();
            load_a((s + 1) * 32);
            load_b((s + 1) * 32);
            mfma_block();
        }
        if (n_stages > 1) kbar();
        store_a((n_stages - 1) * 32);
        store_b();
        kbar();

        const int ovy0 = Y0 + TR * step - 4;                       // virtual output row of T row 0 (negative / not emitted in step 0)
        if (GRID) {
            mfma_block();
            // ---- this step's constants: LDS tables -> registers (the T tile is about to cover them) -------------------------
            const int t = u_opaque(threadIdx.x), lane = t & 63, wave = t >> 6, lr = lane & 31, kh = lane >> 5;
            const int cg = t & 3, oxl = u_fir_col(t >> 2);   // FIR phase: 8-channel group, local output column 0..59 (oxl < 60)
            f4 dq[RW][4];
            {
                const int ixl = (int)__umulhi((unsigned)max(mx0 + lr, 0), g.invPX) - ixi0;
#pragma unroll
                for (int i = 0; i < RW; ++i) {
                    const int iyl = (int)__umulhi((unsigned)max(my0 + RW * wave + i, 0), g.invPY) - iyi0;
                    const char* dp = smem + U_OFF_DS + (((iyl << 2) + ixl) & 7) * 128 + kh * 16;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) dq[i][gq] = *(const f4*)(dp + gq * 32);
                }
            }
            const int ovx = txi * 60 + min(oxl, 59);
            const int ixo = (int)__umulhi((unsigned)ovx, g.inv2PX);
            const int ox = ovx - ixo * 2 * PX;
            const bool col_on = t < 240 && ox < p.Wo && ixo < g.NXI;
            h8 bias8, psa, psb;
            {
                const f4 b0 = *(const f4*)(smem + U_OFF_BI + cg * 32), b1 = *(const f4*)(smem + U_OFF_BI + cg * 32 + 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) { bias8[j] = (half_t)b0[j]; bias8[j + 4] = (half_t)b1[j]; }
                const int sx = (ixo - ixi0) & 3;
                psa = *(const h8*)(smem + U_OFF_PS + sx * 64 + cg * 16);
                psb = *(const h8*)(smem + U_OFF_PS + (4 + sx) * 64 + cg * 16);
            }
            float nzr[TR];
#pragma unroll
            for (int r = 0; r < TR; ++r) nzr[r] = *(const float*)(smem + U_OFF_NZ + (r * 60 + min(oxl, 59)) * 4);
            __syncthreads();   // everyone is done with the staging area and the tables: overlay T

            // ---- t tile -> LDS (demod applied; it commutes with the FIR); layout as in upfir_kernel --------------------------
            {
                char* tw = (char*)T + ((2 * wave * RW) * 64 + lr) * 64 + kh * 8;          // column 2 lr + (ph & 1) -> position (ph & 1) * 32 + lr
                int so[4];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) so[gq] = (gq ^ ((lr >> 1) & 3)) * 16;
#pragma unroll
                for (int i = 0; i < RW; ++i)
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            h4 o;
#pragma unroll
                            for (int q = 0; q < 4; ++q) o[q] = (half_t)(acc[i][ph][gq * 4 + q] * dq[i][gq][q]);
                            *(h4*)(tw + so[gq] + ((2 * i + (ph >> 1)) * 64 + (ph & 1) * 32) * 64) = o;
                        }
            }
            __syncthreads();

            // ---- FIR (separable [1,3,3,1]/4 per axis) + noise + bias + lrelu; T row r <-> virtual output row ovy0 + r ----
            if (col_on) {
                h8 hs[4];
                hs[1] = *(const h8*)(hs_slot); hs[2] = *(const h8*)(hs_slot + 16); hs[3] = *(const h8*)(hs_slot + 32);   // rows 13, 14, 15 of the previous step
                const half_t f3 = (half_t)3.f, fq4 = (half_t)0.0625f, ft4 = (half_t)0.1875f;   // [1,3,3,1]/4 per axis: horizontal pass unscaled (x 4), vertical / 16
                // (activation as max(v, slope v) * (gain * consumer style): see the single-image instance below)
                const half_t slope = (half_t)(p.act ? 0.2f : 1.f);
                const half_t gain = (half_t)((p.act ? GLASS_SQRT2 : 1.f) * p.out_scale);
                const h8 kpsa = psa * gain, kpsb = psb * gain;
                const char* tr[4];
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    const int ltx = oxl + 1 + jx;
                    tr[jx] = (const char*)T + u_tpos(ltx) * 64 + ((cg ^ ((ltx >> 2) & 3)) * 16);
                }
                const int yb = 2 * PY * (iyi0 + 1);                        // first virtual output row of the step's second image row
                // Software-pipelined by hand (r04): row r + 1's four T vectors are requested before row r is filtered, and only the STORE is
                // predicated.  As a plain loop every row was: 4 ds_read_b128, s_waitcnt, 8 packed ops, branch, ~30 packed ops, store — an
                // exposed LDS round trip per row, 16 per step, in a phase where nothing else is in flight.
                h8 cv[4];
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) cv[jx] = *(const h8*)(tr[jx]);
#pragma unroll
                for (int r = 0; r < TR; ++r) {
                    h8 nv[4];
                    if (r + 1 < TR) {
#pragma unroll
                        for (int jx = 0; jx < 4; ++jx) nv[jx] = *(const h8*)(tr[jx] + (r + 1) * 4096);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    hs[r & 3] = (cv[1] + cv[2]) * f3 + (cv[0] + cv[3]);   // 4 x the filtered row: the 1/4 rides in the vertical pass's weights
                    const int ovy = ovy0 + r;
                    const bool second = ovy >= yb;
                    const int iyo = iyi0 + (second ? 1 : 0);
                    const int oy = ovy - 2 * PY * iyo;
                    const int img = img0 + iyo * g.NXI + ixo;
                    const bool emit = (step > 0 || r >= 4) && oy >= 0 && oy < p.Ho && iyo < g.NYI && img < p.B;
                    const h8 bn = bias8 + (half_t)nzr[r];
                    const h8 v = (hs[(r - 3) & 3] + hs[r & 3]) * fq4 + ((hs[(r - 2) & 3] + hs[(r - 1) & 3]) * ft4 + bn);
                    half_t* yp = p.y + (((long long)img * p.Ho + oy) * p.Wo + ox) * p.Cout + n0 + cg * 8;
                    if (emit) *(h8*)yp = __builtin_elementwise_max(v, v * slope) * (second ? kpsb : kpsa);
                    __builtin_amdgcn_sched_barrier(0);
                    if (r + 1 < TR) {
#pragma unroll
                        for (int jx = 0; jx < 4; ++jx) cv[jx] = nv[jx];
                    }
                }
                *(h8*)(hs_slot) = hs[1]; *(h8*)(hs_slot + 16) = hs[2]; *(h8*)(hs_slot + 32) = hs[3];
            }
        } else {
            // ---- single image: everything the FIR needs from global memory is fetched HERE, unconditionally and in one batch, so that
            // it lands under the last MFMA block (upfir_kernel's scheme) ----
            const int t = u_opaque(threadIdx.x), lane = t & 63, wave = t >> 6, lr = lane & 31, kh = lane >> 5;
            const int cg = t & 3, oxl = u_fir_col(t >> 2);
            const int ox = txi * 60 + oxl;
            const int pxc = min(ox, p.Wo - 1);
            f4 bq0 = {0.f, 0.f, 0.f, 0.f}, bq1 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
                bq0 = *(const f4*)(p.bias + n0 + cg * 8);
                bq1 = *(const f4*)(p.bias + n0 + cg * 8 + 4);
            }
            h8 ps8;
#pragma unroll
            for (int j = 0; j < 8; ++j) ps8[j] = (half_t)1.f;
            if (p.post_scale16) ps8 = *(const h8*)(p.post_scale16 + (long long)img0 * p.post_stride + n0 + cg * 8);
            float nzr[TR];
#pragma unroll
            for (int r = 0; r < TR; ++r) nzr[r] = 0.f;
            if (p.noise) {
                const float* nzp = p.noise + (long long)(img0 / p.batch_size) * p.Ho * p.Wo + pxc;
#pragma unroll
                for (int r = 0; r < TR; ++r) nzr[r] = nzp[(long long)min(max(ovy0 + r, 0), p.Ho - 1) * p.Wo];
            }
            mfma_block();
            // the step's noise values park in LDS behind the T tile (one column per cg == 0 thread): as registers they would be live
            // through the whole FIR next to the prefetched operands of the next step, and a spilled value's scratch reload is a VMEM
            // op that waits for every older prefetch load (in-order vmcnt)
            if (cg == 0 && t < 240) {
#pragma unroll
                for (int r = 0; r < TR; ++r) *(float*)(smem + U_OFF_LNZ + (r * 60 + oxl) * 4) = p.noise_strength * nzr[r];
            }
            kbar();            // everyone is done with the staging area: overlay T
            if (U_ABL(2)) { if (acc[0][0][0] == 12345.678f) p.y[0] = (half_t)1.f; continue; }
            {
                char* tw = (char*)T + ((2 * wave * RW) * 64 + lr) * 64 + kh * 8;          // column 2 lr + (ph & 1) -> position (ph & 1) * 32 + lr
                int so[4];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) so[gq] = (gq ^ ((lr >> 1) & 3)) * 16;
#pragma unroll
                for (int i = 0; i < RW; ++i)
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            h4 o;
#pragma unroll
                            for (int q = 0; q < 4; ++q) o[q] = (half_t)acc[i][ph][gq * 4 + q];
                            *(h4*)(tw + so[gq] + ((2 * i + (ph >> 1)) * 64 + (ph & 1) * 32) * 64) = o;
                        }
            }
            kbar();
            // stage 0 of the NEXT step: its operands travel while this step's FIR runs (the step is latency-bound: two exposed
            // global round trips + the store drain at the step barrier were ~2/3 of its 14 us on the two-stage 1024^2 layer)
            if (g.prefetch && step + 1 < g.S && Y0 + TR * (step + 1) - 4 < out_rows) {
                aim(my0 + MR, 0);
                load_a(0);
                load_b(0);
                have = true;
            }
            if (t < 240 && ox < p.Wo) {
                h8 hs[4];
                hs[1] = *(const h8*)(hs_slot); hs[2] = *(const h8*)(hs_slot + 16); hs[3] = *(const h8*)(hs_slot + 32);   // rows 13, 14, 15 of the previous step
                const half_t f3 = (half_t)3.f, fq4 = (half_t)0.0625f, ft4 = (half_t)0.1875f;   // [1,3,3,1]/4 per axis: horizontal pass unscaled (x 4), vertical / 16
                h8 bias8;
#pragma unroll
                for (int j = 0; j < 4; ++j) { bias8[j] = (half_t)bq0[j]; bias8[j + 4] = (half_t)bq1[j]; }
                // activation + output gain + consumer style as  max(v, slope v) * (gain * style)  (r05): three packed ops per register where
                // max(v k1, v k2) * style  took four plus a select (the optional style multiply was compiled as multiply + v_cndmask: 8 of the
                // row's ~60 VALU instructions in a loop that is bound by VALU issue).  ps8 is all ones without a consumer style.
                const half_t slope = (half_t)(p.act ? 0.2f : 1.f);
                const h8 kps = ps8 * (half_t)((p.act ? GLASS_SQRT2 : 1.f) * p.out_scale);
                const char* tr[4];
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    const int ltx = oxl + 1 + jx;
                    tr[jx] = (const char*)T + u_tpos(ltx) * 64 + ((cg ^ ((ltx >> 2) & 3)) * 16);
                }
                // pixel-major [B][Ho][Wo][Cout], or chunk-planar [B][Cout / 8][Ho][Wo][8] for conv_wreg (common.h): the lane's eight channels are a plane
                const long long rowpitch = (long long)p.Wo * (p.y_planar8 ? 8 : p.Cout);
                half_t* yp = p.y_planar8 ? p.y + ((((long long)img0 * 4 * g.NTn + (n0 >> 3) + cg) * p.Ho + ovy0) * p.Wo + ox) * 8
                                          : p.y + (((long long)img0 * p.Ho + ovy0) * p.Wo + ox) * p.Cout + n0 + cg * 8;   // (row ovy0 + r is only touched when it exists)
#ifdef GLASS_AB_KNOBS
                // (r05 experiment, developer build: the BS instance) buffer form: descriptor over this image's output map (uniform), 32-bit byte offset per lane
                const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (long long)img0 * p.Ho * p.Wo * p.Cout), 0,
                                                                                      (int)((long long)p.Ho * p.Wo * p.Cout * 2), 0x00020000);
                unsigned yoff = (unsigned)((((long long)ovy0 * p.Wo + ox) * p.Cout + n0 + cg * 8) * 2);   // (wraps for the rows of step 0 that are not written)
#endif
                // (software-pipelined like the grid instance's loop above: row r + 1's T vectors and noise value are requested before row r
                // is filtered, only the store is predicated — the plain loop had TWO exposed LDS round trips per row)
                h8 cv[4];
                float cnz;
                auto rdrow = [&](int r, h8 (&v_)[4], float& nz_) {
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) v_[jx] = *(const h8*)(tr[jx] + r * 4096);
                    nz_ = *(const float*)(smem + U_OFF_LNZ + (r * 60 + oxl) * 4);
                };
                rdrow(0, cv, cnz);
#pragma unroll
                for (int r = 0; r < TR; ++r) {
                    h8 nv[4];
                    float nnz;
                    if (r + 1 < TR) rdrow(r + 1, nv, nnz);
                    __builtin_amdgcn_sched_barrier(0);
                    hs[r & 3] = (cv[1] + cv[2]) * f3 + (cv[0] + cv[3]);   // 4 x the filtered row: the 1/4 rides in the vertical pass's weights
                    const h8 bn = bias8 + (half_t)cnz;
                    h8 v = (hs[(r - 3) & 3] + hs[r & 3]) * fq4 + ((hs[(r - 2) & 3] + hs[(r - 1) & 3]) * ft4 + bn);
                    v = __builtin_elementwise_max(v, v * slope) * kps;
                    if (U_ABL(8)) v = cv[0];
#ifdef GLASS_AB_KNOBS
                    if constexpr (BS) {
                        // UNCONDITIONAL buffer store, rows that must not be written get an out-of-range offset (the hardware drops them): the
                        // number of memory operations in flight is then static, so the wait in front of the next step's prefetched operands is a
                        // counted vmcnt(16) that lets the row stores keep draining (a store under a branch forces vmcnt(0))
                        const unsigned off = ((step > 0 || r >= 4) && ovy0 + r < p.Ho && !U_ABL(4)) ? yoff : 0xFFFFFFF0u;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), yrsrc, off, 0, 0);
                        yoff += (unsigned)rowpitch * 2u;
                    } else
#endif
                    if ((step > 0 || r >= 4) && ovy0 + r < p.Ho && (!U_ABL(4) || v[0] == (half_t)777.f)) *(h8*)yp = v;
                    yp += rowpitch;
                    __builtin_amdgcn_sched_barrier(0);
                    if (r + 1 < TR) {
#pragma unroll
                        for (int jx = 0; jx < 4; ++jx) cv[jx] = nv[jx];
                        cnz = nnz;
                    }
                }
                *(h8*)(hs_slot) = hs[1]; *(h8*)(hs_slot + 16) = hs[2]; *(h8*)(hs_slot + 32) = hs[3];
            }
            // (r05) stage 0 of the NEXT step is requested HERE, behind the FIR and ahead of the step barrier: the loads go to registers
            // that are free from here on, travel across the barrier, and their addresses cost five adds — an interior step follows an
            // interior step MR rows further down, so the full address derivation (aim(): ~300 VALU instructions with divisions, which
            // sat between the barrier and the first load of every step) only runs at the image borders.  Weights first: their addresses
            // do not depend on the step.
            if (!have && step + 1 < g.S && Y0 + TR * (step + 1) - 4 < out_rows && !U_ABL(256)) {
                const int my0n = my0 + MR;
                const bool plain_n = mx0 >= 1 && my0n >= 1 && my0n + MR - 1 < p.H && mx0 + 31 < p.W;
                load_b(0);
                if (plain && plain_n) {
                    const int dstep = MR * p.W * p.Cin;
#pragma unroll
                    for (int k = 0; k < NA; ++k) a_goff[k] += dstep;
                } else {
                    aim(my0n, 0);
                }
                load_a(0);
                have = true;
            }
        }
    }
}

static unsigned u_inv(int d) { return (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); }

// steps per segment of the image-grid instance by a list-scheduling model (see launch_upfir2_t); per_seg = workgroups per segment
// index, fallback = the count-rule's answer (kept when no length fills the slots).  Cached per geometry: the launcher runs every pass.
static int upfir2_model_steps(int per_seg, int out_rows, int TR, int slots, int fallback) {
    struct Key { int per_seg, out_rows, slots, S; };
    static std::mutex mu;
    static std::vector<Key> cache;
    std::lock_guard<std::mutex> lk(mu);
    for (const Key& k : cache)
        if (k.per_seg == per_seg && k.out_rows == out_rows && k.slots == slots) return k.S;
    int best = fallback;
    double best_t = 1e30;
    std::vector<double> heap;
    for (int S = 1; S <= 16; ++S) {
        const int R = (TR - 4) + TR * (S - 1);
        if (R > out_rows + TR - 1 && S > 1) break;
        const int n_seg = (out_rows + R - 1) / R;
        if ((long long)per_seg * n_seg < slots) break;             // longer segments would leave slots empty
        heap.assign(slots, 0.0);                                    // min-heap of slot finish times
        auto cmp = [](double a, double b) { return a > b; };
        for (int seg = 0; seg < n_seg; ++seg) {
            int steps = 0;
            for (int st = 0; st < S; ++st) {
                if (seg * R + (st ? TR * st - 4 : 0) >= out_rows) break;
                ++steps;
            }
            for (int w = 0; w < per_seg; ++w) {
                std::pop_heap(heap.begin(), heap.end(), cmp);
                heap.back() += steps + 0.5;
                std::push_heap(heap.begin(), heap.end(), cmp);
            }
        }
        double t = 0;
        for (double v : heap) t = v > t ? v : t;
        if (t < best_t - 1e-9) { best_t = t; best = S; }
    }
    cache.push_back({per_seg, out_rows, slots, best});
    return best;
}

// RW = m rows per wave: 2 = the two-workgroup step (8 m rows), 1 = the half-height step of the single-image instance (three workgroups per CU)
template <int RW>
static const char* launch_upfir2_t(const ConvParams& p, hipStream_t st, bool lean) {
    using US = UStep<RW>;
    constexpr int LDS = US::LDS, TR = US::TR;
    if (!glass_lds_fits(LDS)) return nullptr;                 // (the caller falls through to upfir_kernel / the folded form)
    static DevOnce once;
    once.run([&] {
        if (RW == 2) (void)hipFuncSetAttribute((const void*)upfir2_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)upfir2_kernel<false, RW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
    static const int env_ng = glass_knob("GLASS_UPFIR_NG") ? atoi(glass_knob("GLASS_UPFIR_NG")) : 0;      // A/B knobs
    static const int env_s = glass_knob("GLASS_UPFIR_S") ? atoi(glass_knob("GLASS_UPFIR_S")) : 0;
    static const bool no_grid = glass_knob("GLASS_UPFIR_NO_GRID") != nullptr;
    UpGeo g;
    g.NTn = p.Cout / 32;
    // candidates per virtual grid: per-sample weights -> one (its tiles share a weight set); shared weights -> up to 8 x 8
    if (p.w_bstride || no_grid) { g.NXI = 1; g.NYI = 1; }
    else {
        // a tile's 33-pixel patch must not touch more than the FOUR images a table row holds: at pitches below 11 (the 8 x 8 input of the
        // r16 layer, round 6) a virtual row is three images wide — one 30-column tile covers it
        const int nxi_max = p.W + 1 >= 11 ? 8 : 3;
        g.NXI = p.B < nxi_max ? p.B : nxi_max;
        g.NYI = (p.B + g.NXI - 1) / g.NXI;
        if (g.NYI > 8) g.NYI = 8;
    }
    g.n_grids = (p.B + g.NXI * g.NYI - 1) / (g.NXI * g.NYI);
    const int PX = p.W + 1, PY = p.H + 1;
    g.tiles_x = (2 * PX * g.NXI - 2 + 59) / 60;
    const int out_rows = 2 * PY * g.NYI - 2;
    // steps per segment: as long as possible (only a segment's first step recomputes the t halo) while the launch still has
    // >= 8 workgroups per CU-slot pair to balance (2048) — >= 20 (5120) for segments longer than 8 steps, whose workgroups run twice as
    // long (round 6, same box: the r1024 layer 1955 -> 1880 us at S = 16, 5760 workgroups; the r512 layer 1377 -> 1420 us at its 3456:
    // it stays at S = 8); never longer than the grid
    int S = 32 / RW;
    if (RW == 2 && S > 16) S = 16;
    for (; S > 1; --S) {
        const int R = (TR - 4) + TR * (S - 1);
        const long long wgs = (long long)g.n_grids * g.tiles_x * ((out_rows + R - 1) / R) * g.NTn;
        if (wgs >= (S > 8 ? 5120 : 2048) && R <= out_rows + TR - 1) break;
    }
    // image-grid instance (round 6): the rule above left the low-resolution layers at S = 1 / 2 / 4 (25 / 12 / 6 % of their rows recomputed as
    // halo) for the sake of a workgroup count they do not need.  A sweep of S (tools: GLASS_UPFIR_S on the developer build, medians of 5) follows
    // a plain list-scheduling model — workgroups of (steps + 1/2) step times dealt in launch order to 2 x CUs slots — closely (r64: minima at
    // S = 5 and 10, maximum at 8, as measured: 457 / 455 / 588 us against 514 at the old S = 2; r32: 183 -> 143 us at S = 3): S = the model's
    // minimum among the lengths that still fill every slot once — where the count rule ends at S <= 2 (at S = 4 / 8, r128 / r256, the model's pick measured
    // 1-3 % slower than the rule's: those launches have enough rounds to average out).  (S only decides which t rows are recomputed: results do not depend on it.)
    if (!lean && RW == 2 && S <= 2) S = upfir2_model_steps(g.n_grids * g.tiles_x * g.NTn, out_rows, TR, 2 * glass_cu_count(), S);
    if (env_s > 0) S = env_s;
    g.S = S;
    const int R = (TR - 4) + TR * (S - 1);
    g.n_seg = (out_rows + R - 1) / R;
    g.WT = g.n_grids * g.n_seg * g.tiles_x;
    // n tiles per XCD group: the group's weights (9 * 32 * Cin * 2 B per n tile) should stay resident in a 4 MiB L2 next to the
    // input patches; per-sample weights are image-local already
    int ng = 1;
    if (!p.w_bstride) {
        const long long per_tile = 9LL * 32 * p.Cin * 2;
        while (ng < 8 && g.NTn % (ng * 2) == 0 && (g.NTn / ng) * per_tile > (1200LL << 10)) ng *= 2;
    }
    if (env_ng > 0 && 8 % env_ng == 0 && g.NTn % env_ng == 0) ng = env_ng;
    g.ngroups = ng;
    // measured (round 3, same box): prefetch on 2298 / 1476 / 1217 us vs off 2183 / 1496 / 1235 us on the r1024 / r512 / r256
    // layers — a wash, as round 2's persistent-prefetch experiment was (round 5: 2127 vs 2016 us).  Off.
    static const bool prefetch = glass_knob("GLASS_UPFIR_PREFETCH") != nullptr;
    g.prefetch = prefetch ? 1 : 0;
#ifdef GLASS_AB_KNOBS
    static const int ablate = glass_knob("GLASS_UPFIR_ABLATE") ? atoi(glass_knob("GLASS_UPFIR_ABLATE")) : 0;
    g.ablate = ablate;
    // (r05 experiment, developer build) bit 0: unconditional buffer stores + LDS-only barriers (template instance BS), bit 1: operand prefetch
    static const int bstore = glass_knob("GLASS_UPFIR_BSTORE") ? atoi(glass_knob("GLASS_UPFIR_BSTORE")) : 0;
    if (lean && (bstore & 2)) g.prefetch = 1;
    if (lean && RW == 2 && (bstore & 1)) {
        static DevOnce once_bs;
        once_bs.run([&] { (void)hipFuncSetAttribute((const void*)upfir2_kernel<false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); });
        if (p.dry_run) return "upfir2_kernel<false,2,true>";
        const int Pp_ = 8 / ng;
        hipLaunchKernelGGL((upfir2_kernel<false, 2, true>), dim3(8 * ((g.WT + Pp_ - 1) / Pp_) * (g.NTn / ng)), dim3(256), LDS, st, p, g);
        return "upfir2_kernel<false,2,true>";
    }
#endif
    g.invPX = u_inv(PX); g.invPY = u_inv(PY); g.inv2PX = u_inv(2 * PX); g.inv2PY = u_inv(2 * PY);
    const char* name = !lean ? "upfir2_kernel<true>" : RW == 2 ? "upfir2_kernel<false>" : "upfir2_kernel<false,1>";
    if (p.dry_run) return name;
    const int Pp = 8 / ng;
    const int grid = 8 * ((g.WT + Pp - 1) / Pp) * (g.NTn / ng);
    int lds_req = LDS;
#ifdef GLASS_AB_KNOBS
    // developer build: one workgroup per CU (LDS request raised): is a workgroup's step shorter when it has the CU to itself?
    static const bool one_wg = glass_knob("GLASS_UPFIR_ONE_WG") != nullptr;
    if (one_wg && lean) {
        lds_req = 120 * 1024;
        static DevOnce once1;
        once1.run([&] { (void)hipFuncSetAttribute((const void*)upfir2_kernel<false, RW>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024); });
    }
#endif
    if (lean) hipLaunchKernelGGL((upfir2_kernel<false, RW>), dim3(grid), dim3(256), lds_req, st, p, g);
    else if (RW == 2) hipLaunchKernelGGL((upfir2_kernel<true, 2>), dim3(grid), dim3(256), LDS, st, p, g);
    else return nullptr;
    return name;
}

static const char* launch_upfir2(const ConvParams& p, hipStream_t st) {
    if (p.Cin > 512 || p.H < 8 || p.W < 8) return nullptr;
    if ((long long)p.B * p.H * p.W * p.Cin >= (1LL << 31) || 9LL * p.Cout * p.Cin >= (1LL << 31)) return nullptr;
    if (p.x_bstride != (long long)p.H * p.W * p.Cin) return nullptr;
    static const bool no_grid = glass_knob("GLASS_UPFIR_NO_GRID") != nullptr;
    // per-sample weights carry style and demodulation: the lean single-image instance; anything else goes through the tables
    const bool lean = p.w_bstride && !p.sn16 && !p.dscale;
    (void)no_grid;
    if (p.y_planar8 && !lean) return nullptr;    // the chunk-planar output (common.h) is the single-image instance's
#ifdef GLASS_AB_KNOBS
    // half-height steps (4 m rows, 51 KB of LDS, 156 VGPRs: THREE workgroups per CU) for the single-image instance — developer build only.
    // Measured (r05, same box, parity-green on the op tests and the goldens): r1024 1997 -> 2193 us, r512 1393 -> 1575 us: a third workgroup
    // does not pay for weights staged per 4 rows instead of 8 and a 5-rows-for-4 patch.  Not in the release library.
    static const int rw1 = glass_knob("GLASS_UPFIR_RW1") ? atoi(glass_knob("GLASS_UPFIR_RW1")) : 0;
    if (lean && rw1) return launch_upfir2_t<1>(p, st, true);
#endif
    return launch_upfir2_t<2>(p, st, lean);
}

const char* launch_upconv_fused(const ConvParams& p, hipStream_t st) {
    if (!p.up || !p.w_up || p.y32 || !p.y || p.res || (p.sn && !p.sn16)) return nullptr;
    if (p.Cin % 32 != 0 || p.Cout % 32 != 0 || p.KS != 3) return nullptr;
    if (p.x_bstride == 0 && p.B > 1) return nullptr;
    if (p.x_planar8 || p.x_planar32) return nullptr;   // chunk-planar input (common.h): not implemented here
    static const bool v1 = glass_knob("GLASS_UPFIR_V1") != nullptr;      // A/B knob: round 2's one-tile-per-workgroup kernel
    if (!v1) {
        const char* k = launch_upfir2(p, st);
        if (k) return k;
    }
    if (p.y_planar8) return nullptr;
    if (p.W < 16) return nullptr;

    if ((long long)p.H * p.W * p.Cin >= (1LL << 31)) return nullptr;
    constexpr int LDS = 64 * 1024;  // T tile (16*64*32*2 B); staging (31.5 KB) lives inside it
    if (!glass_lds_fits(LDS)) return nullptr;
    static DevOnce once;
    once.run([&] { (void)hipFuncSetAttribute((const void*)upfir_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); });
    const int tiles_y = (p.Ho + 11) / 12, tiles_x = (p.Wo + 59) / 60;
    const int PT = p.B * tiles_x * tiles_y;
    const int NTn = p.Cout / 32;
    const int PT8 = (PT + 7) / 8 * 8;
    if (p.dry_run) return "upfir_kernel";
    hipLaunchKernelGGL(upfir_kernel, dim3(PT8 * NTn), dim3(256), LDS, st, p, NTn, tiles_x, tiles_y, PT);
    return "upfir_kernel";
}
