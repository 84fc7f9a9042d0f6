// conv_wres.hip — 3x3 stride-1 convolution 64 -> 64 channels at 512^2 (G's last-but-one conv with its toRGB, D's second block's first conv
// with the blur-down by-product; stylegan2/modules.py:920-967, models.py:852-870, modules.py:1238-1254) with the WHOLE weight tensor
// resident in LDS and the input patches on an LDS-DMA double buffer — the ping-pong schedule of conv_glds.hip's persistent kernel.
//
// Why its own kernel (round 6).  Rounds 1-5 ran these two layers on conv_tiled<3,1,8,64>: register-staged, three 256-thread workgroups per
// CU, 1.84 + 1.56 ms where the HBM roof (2.1 GB in + 2.1 GB out per layer at 6.3 TB/s) is 0.7 ms.  A tile of that kernel moves 43 KB of
// patch AND the layer's whole 74 KB of weights through registers into LDS; loads, MFMAs (40 % busy) and stores run back to back per
// tile.  Here the 9 x 64 x 64 weights (73 728 B) are loaded ONCE per workgroup (per candidate for pre-modulated per-sample weights: a
// workgroup walks a contiguous range of tiles), a tile is 16 rows x 32 px x 64 channels on eight waves (wave = 2 rows x 2 n blocks), its two
// 32-channel patch chunks ARE the double buffer (chunk 0 of tile t + 1 lands while chunk 1 of tile t is read), and the K loop is six
// phases of 24 MFMAs on 24 fragments (one tap row of one chunk) played as a ping-pong of the two wave groups:
//     load interval: 24 ds_read_b128, [vmcnt wait], lgkmcnt(0) | barrier | MFMA interval: 24 MFMAs at s_setprio 1, <= 3 ring pieces | barrier
// with waves 4-7 one barrier behind waves 0-3 (one wave of each group per SIMD).  Per tile and wave: 10 patch pieces of 1 KB
//     phase 0: chunk 1 of THIS tile (its buffer held the output slices of the previous tile's epilogue until then) — waited for in phase 2
//     phase 3: chunk 0 of the NEXT tile (buffer 0 is free after phase 2)                                             — waited for in phase 5
// (the epilogue operands — per-channel constants, noise values, toRGB table — are LDS-DMA pieces of phase 1)
// (vmcnt(0) both times: everything else in the queue — the epilogue's stores, the by-product's — is at least a phase older).
// Epilogue = conv_glds.hip's: operands by LDS-DMA, constants through LDS, output slices transposed through patch buffer 1, 16-byte
// row-order stores; toRGB (TRGB) and the blur-down by-product (XS) as there.  Same MFMA order per accumulator as conv_tiled.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

namespace {
constexpr int NT = 64, NTHR = 512, TW = 32, TH = 16, RW = 2;
constexpr int PH = TH + 2, PW = TW + 2;
constexpr int NVA = PH * PW * 4;                         // 16-byte vectors of a patch chunk (2448)
constexpr int NA = (NVA + NTHR - 1) / NTHR;              // 5 pieces per thread per chunk
constexpr int A_BYTES = NA * NTHR * 16;                  // 40960
constexpr int NVW = 2 * 9 * NT * 4;                      // 4608 vectors of the weight image [chunk][tap][n][32 ch]
constexpr int NWP = NVW / NTHR;                          // 9 pieces per thread
constexpr int OFF_W = 2 * A_BYTES;                       // 81920
constexpr int OFF_C = OFF_W + NVW * 16;                  // 155648: [dscale | bias | shift][NT] fp32
constexpr int OFF_T = OFF_C + 3 * NT * 4;                // TRGB: toRGB weight rows [hi r,g,b | lo r,g,b][NT] fp16
constexpr int OFF_N = OFF_T + 6 * NT * 2;                // the tile's noise values [8 waves][2 rows][32 px] fp32
constexpr int LDS_BYTES = OFF_N + 8 * 64 * 4;            // 159232
static_assert(LDS_BYTES <= 163840 && NVW % NTHR == 0 && 8 * RW * 32 * 80 <= A_BYTES, "one workgroup per CU; the output slices fit a patch buffer");

__device__ __attribute__((aligned(64))) half_t g_wres_zero_page[32];   // zero-initialised: source of the zero padding

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
__device__ __forceinline__ void dma16(const half_t* src, char* lds_wave_base) {    // LDS destination = wave-uniform base + lane * 16
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const float* src, char* lds_wave_base) {      // LDS destination = wave-uniform base + lane * 4
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_wave_base, 4, 0, 0);
}
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
#define WR_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
}  // namespace

template <bool TRGB, bool XS>
__global__ __launch_bounds__(512, 1) void conv_wres_kernel(ConvParams p, int tiles_x, int tiles_y, int PT, int per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2;                 // 0: waves 0-3 lead; 1: waves 4-7 run one barrier behind
    const int tpi = tiles_x * tiles_y;
    const int first = blockIdx.x * per_wg, last = min(first + per_wg, PT);     // a contiguous range of tiles: consecutive tiles are x neighbours
    if (first >= last) return;
    struct Item { int b, ty0, tx0, pad; };
    auto decode = [&](int pt) {
        Item w;
        w.b = pt / tpi;
        const int trem = pt - w.b * tpi;
        w.ty0 = (trem / tiles_x) * TH;
        w.tx0 = (trem % tiles_x) * TW;
        w.pad = 0;
        return w;
    };
    int id = first;
    Item cur = decode(id);

    // ---- DMA sources --------------------------------------------------------------------------------------------------------------
    // vector v = k * 512 + t of an LDS image sits at byte v * 16: row = v >> 2, physical chunk v & 3 holding the row's LOGICAL 8-channel
    // chunk (v & 3) ^ ((row >> 2) & 3) (swizzle on the source side: LDS-DMA writes lane-linear)
    const half_t* xb = p.x;
    int a_src[NA];            // element offset into the image (< 2^31: launcher), or -1 = zero page
    auto aim_a = [&](const Item& w) {
        const int t = opaque(threadIdx.x);
        xb = p.x + (long long)w.b * p.x_bstride;
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int v = k * NTHR + t, pix = v >> 2;
            const int pr = pix / PW, pc = pix - pr * PW;
            const int iy = w.ty0 - 1 + pr, ix = w.tx0 - 1 + pc;
            const int lc = (v & 3) ^ ((pix >> 2) & 3);
            const bool ok = v < NVA && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            a_src[k] = ok ? (iy * p.W + ix) * (p.x_planar32 ? 32 : p.Cin) + lc * 8 : -1;
        }
    };
    // the second chunk of a pixel: 32 channels on in the pixel-major layout, a plane on in the chunk-planar one (common.h: there a 128-byte
    // line holds ONE chunk of two pixels — pixel-major, its two halves were fetched a half tile apart and crossed the fabric twice)
    const int c_step = p.x_planar32 ? p.H * p.W * 32 : 32;
    auto issue_a1 = [&](int k, int c) {        // piece k of chunk c -> patch buffer c (k is a constant wherever this is called)
        char* dst = smem + c * A_BYTES + wave * 1024 + k * (NTHR * 16);
        dma16(a_src[k] >= 0 ? xb + a_src[k] + c * c_step : g_wres_zero_page + (threadIdx.x & 3) * 8, dst);
    };
    auto issue_w = [&](int b) {                // the weight image of candidate b (shared weights: w_bstride = 0)
        const half_t* wb = p.w + (long long)b * p.w_bstride;
        const int t = opaque(threadIdx.x);
#pragma unroll
        for (int k = 0; k < NWP; ++k) {
            const int v = k * NTHR + t, row = v >> 2;              // row = (c * 9 + tap) * 64 + n
            const int n = row & 63, ct = row >> 6, c = ct / 9, tap = ct - 9 * c;
            const int lc = (v & 3) ^ ((row >> 2) & 3);
            dma16(wb + ((long long)tap * p.Neff + n) * p.Cin + c * 32 + lc * 8, smem + OFF_W + wave * 1024 + k * (NTHR * 16));
        }
    };

    {   // operand arrays the layer does not have keep their neutral values for the kernel's lifetime (the epilogue reads all of them)
        const int t = threadIdx.x;
        float* Cc = (float*)(smem + OFF_C);
        if (t < NT) {
            if (!p.dscale) Cc[t] = 1.f;
            if (!p.bias) Cc[NT + t] = 0.f;
            if (!p.shift) Cc[2 * NT + t] = 0.f;
        }
        if (!p.noise) *(float*)(smem + OFF_N + t * 4) = 0.f;
    }
    __syncthreads();
    aim_a(cur);
    issue_w(cur.b);
#pragma unroll
    for (int k = 0; k < NA; ++k) issue_a1(k, 0);
    WR_WAIT_VM0();
    __builtin_amdgcn_s_barrier();
    h8 wf[2][3][2][2];                         // weight fragments of the current / the next phase: [phase parity][tx][kk][j]
    int wbase[2][2];                           // lane-constant LDS offsets of the weight image's fragment rows [chunk][kk], computed once and kept
                                               // opaque: rebuilt per read they were ~100 VALU instructions inside every MFMA interval
    {
        const int t = threadIdx.x, lr = t & 31, kh = (t >> 5) & 1;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) wbase[cc][kk] = opaque(OFF_W + cc * 9 * NT * 64 + lr * 64 + (((kk * 2 + kh) ^ ((lr >> 2) & 3)) << 4));
    }
    bool have_w = false;                       // (uniform) phase 0's fragments were requested in the previous tile's last phase
    for (;;) {
        const bool has_next = id + 1 < last;
        const Item nxt = has_next ? decode(id + 1) : cur;     // the ring never branches: the last tile re-requests itself (nobody reads that buffer)
        const int b = cur.b, ty0 = cur.ty0, tx0 = cur.tx0;

        f16x acc[RW][2];
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

        // epilogue operands (per-channel constants, the tile's noise values, the toRGB table) by LDS-DMA, ahead of phase 3's patch pieces
        auto prefetch_epilogue = [&]() {
            const int t = opaque(threadIdx.x), lr = t & 31, kh = (t >> 5) & 1;
            if (wave == 0) {                   // channels t < NT = 64: [dscale | bias | shift][NT] fp32
                char* cc = smem + OFF_C;
                if (p.dscale) dma4(p.dscale + (long long)b * p.ds_stride + t, cc);
                if (p.bias) dma4(p.bias + t, cc + NT * 4);
                if (p.shift) dma4(p.shift + (long long)b * p.ds_stride + t, cc + 2 * NT * 4);
            }
            if (p.noise) dma4(p.noise + ((long long)(b / p.batch_size) * p.Ho + ty0 + wave * RW + kh) * p.Wo + tx0 + lr, smem + OFF_N + wave * 256);
            if (TRGB && t < 6 * (NT / 8)) {
                const int row6 = t / (NT / 8), piece = t % (NT / 8);
                const int n = row6 < 3 ? row6 : 8 + (row6 - 3);
                dma16(p.trgb_tab + ((long long)b * 32 + n) * NT + piece * 8, smem + OFF_T + wave * 1024);
            }
        };
        if (grp) __builtin_amdgcn_s_barrier();     // group 1 falls one barrier behind: its load intervals face group 0's MFMA intervals
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int c = q / 3, ty = q - 3 * c;
            const char* As = smem + c * A_BYTES;
            // weight fragments of a phase: [tx][kk][j]; the weights are resident, so the NEXT phase's twelve are requested in this phase's
            // MFMA interval (two per group of four MFMAs) into the other register set — 24 reads in the load interval took longer than the
            // partner's 24 MFMAs (~45 clocks per read and wave with four waves reading), the matrix pipe idled a third of every interval
            auto rd_w = [&](int qq, int tx, int kk, int j) {      // one instruction: lane-constant base + immediate offset
                const int cc = qq / 3, tyy = qq - 3 * cc;
                return *(const h8*)(smem + wbase[cc][kk] + ((tyy * 3 + tx) * NT + j * 32) * 64);
            };
            // ---- load interval ---------------------------------------------------------------------------------------------------
            h8 xf[3][2][RW];                   // [tx][kk][i]
            {
                const int tm = opaque(threadIdx.x), lr = tm & 31, kh = (tm >> 5) & 1;
                if (q == 0 && !have_w) {       // first tile of the range / behind a weight reload: nobody prefetched phase 0's fragments
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                            for (int j = 0; j < 2; ++j) wf[0][tx][kk][j] = rd_w(0, tx, kk, j);
                }
#pragma unroll
                for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const int lc = kk * 2 + kh;
#pragma unroll
                        for (int i = 0; i < RW; ++i) {
                            const int pix = (wave * RW + i + ty) * PW + lr + tx;
                            xf[tx][kk][i] = *(const h8*)(As + pix * 64 + ((lc ^ ((pix >> 2) & 3)) << 4));
                        }
                    }
            }
            if (q == 3) aim_a(nxt);            // (~150 address instructions with divisions: in the short load interval, not between MFMAs; this tile's
                                               // pieces were all issued in phase 0)
            if (q == 2 || q == 5) WR_WAIT_VM0();   // this tile's chunk 1 / the next tile's chunk 0 (and the epilogue operands) have landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- MFMA interval ---------------------------------------------------------------------------------------------------
            auto ring_pieces = [&]() {
                if (q == 0) {
#pragma unroll
                    for (int k = 0; k < NA; ++k) issue_a1(k, 1);
                } else if (q == 1) {
                    prefetch_epilogue();
                } else if (q == 3) {
#pragma unroll
                    for (int k = 0; k < NA; ++k) issue_a1(k, 0);
                }
            };
            // XS by-product (a chunk's first phase): FIR 4x4 (pad 1) + ::2 of this chunk of the INPUT map, 8 x 16 pixels x 4 parts = one vector
            // per thread, from the patch: one patch row of four vectors per MFMA group, folded a group later — in the MFMA stream's shadow
            // (explicit FMA forms: conv_glds.hip)
            const bool xs_on = XS && ty == 0;
            h8 xa[4], s03, s12, k125, k375;
            int xs_part = 0, xs_ly = 0, xs_lx = 0;
            if (xs_on) {
                const int tq = opaque(threadIdx.x);
                xs_part = tq & 3; xs_ly = tq >> 6; xs_lx = (tq >> 2) & 15;
#pragma unroll
                for (int e = 0; e < 8; ++e) { k125[e] = (half_t)0.125f; k375[e] = (half_t)0.375f; }
            }
            auto xs_read = [&](int jy) {
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    const int P = (2 * xs_ly + jy) * PW + 2 * xs_lx + jx;
                    xa[jx] = *(const h8*)(As + P * 64 + ((xs_part ^ ((P >> 2) & 3)) << 4));
                }
            };
            auto xs_fold = [&](int jy) {
                const h8 hr = __builtin_elementwise_fma(xa[1] + xa[2], k375, (xa[0] + xa[3]) * k125);     // (conv_tiled<xs>'s contraction, made explicit there too)
                if (jy == 0) s03 = hr;
                else if (jy == 1) s12 = hr;
                else if (jy == 2) s12 = s12 + hr;
                else s03 = s03 + hr;
            };
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < RW; ++i) acc[i][j] = mfma32(wf[q & 1][tx][kk][j], xf[tx][kk][i], acc[i][j]);
                    const int m = tx * 2 + kk;
                    if (m == 0) { __builtin_amdgcn_sched_barrier(0); ring_pieces(); __builtin_amdgcn_sched_barrier(0); }
                    {   // the next phase's weight fragments (phase 0 of the next tile behind phase 5)
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < 2; ++j) wf[(q + 1) & 1][tx][kk][j] = rd_w((q + 1) % 6, tx, kk, j);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (xs_on && m >= 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (m >= 2) xs_fold(m - 2);
                        if (m <= 4) xs_read(m - 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            if (xs_on) {
                const h8 o = __builtin_elementwise_fma(s12, k375, s03 * k125);
                *(h8*)(p.xs_out + (((long long)b * (p.H >> 1) + (ty0 >> 1) + xs_ly) * (p.W >> 1) + (tx0 >> 1) + xs_lx) * p.Cin + c * 32 + xs_part * 8) = o;
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }

        // ---- epilogue (conv_glds.hip's): group 0 re-aligns first; every fragment read of the tile was complete before the barrier both groups
        // have passed by then, and the phase-5 wait + barrier made the operands visible.  A candidate switch of per-sample weights reloads
        // the weight image here (every wave is out of the K loop) and is waited for behind the epilogue. ---------------------------------
        if (!grp) __builtin_amdgcn_s_barrier();
        const bool reload = has_next && p.w_bstride != 0 && nxt.b != b;      // uniform
        if (reload) issue_w(nxt.b);
        const int t = opaque(threadIdx.x), lane = t & 63, lr = lane & 31, kh = lane >> 5;
        const float* Cc = (const float*)(smem + OFF_C);
        constexpr int OP = 80;                                      // bytes per staged pixel slice (64 + 16: bank spread)
        char* Os = smem + A_BYTES + wave * (RW * 32 * OP);
        const int oyb = ty0 + wave * RW, ox = tx0 + lr;              // lane's pixel of tile row i: (oyb + i, ox)
        const int rcs = p.res_cs ? p.res_cs : p.Cout;
        const ActK ak = act_consts(p.act, p.out_scale);
        float ytap[3][4];                      // (a load into registers: used at the END of the epilogue)
        if (TRGB && p.trgb_yprev) {
            const int my = (oyb + kh) >> 1, mx = ox >> 1, h2 = p.Ho >> 1, w2 = p.Wo >> 1;
            const float* yp = p.trgb_yprev + (long long)b * 3 * h2 * w2;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    ytap[cc][q] = yp[(cc * h2 + max(my - 1 + (q >> 1), 0)) * w2 + max(mx - 1 + (q & 1), 0)];
        }
        float nzr[RW];
#pragma unroll
        for (int i = 0; i < RW; ++i) nzr[i] = p.noise_strength * *(const float*)(smem + OFF_N + (wave * 64 + i * 32 + lr) * 4);
        f16x rgb;
#pragma unroll
        for (int q = 0; q < 16; ++q) rgb[q] = 0.f;
        const int tn = lr & 15;
        const char* Trow = smem + OFF_T + (((tn >> 3) & 1) * 3 + min(tn & 3, 2)) * (NT * 2) + kh * 16;
        const bool trow_ok = (tn & 3) < 3;
        const h8 hzero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            h4 va[RW][4];
            h4 rq[4][RW];
            if (p.res) {
#pragma unroll
                for (int i = 0; i < RW; ++i) {
                    const int oy = oyb + i;
                    const half_t* rp = p.res + (p.res_up ? (((long long)b * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * rcs
                                                         : (((long long)b * p.Ho + oy) * p.Wo + ox) * rcs) + j * 32 + 4 * kh;
#pragma unroll
                    for (int g = 0; g < 4; ++g) rq[g][i] = *(const h4*)(rp + 8 * g);
                }
            }
            f4 dq[4], bq[4];                   // this slice's eight constant quads as ONE batch of LDS reads
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = j * 32 + 8 * g + 4 * kh;
                dq[g] = *(const f4*)(Cc + nl);
                bq[g] = *(const f4*)(Cc + NT + nl) + *(const f4*)(Cc + 2 * NT + nl);      // bias + shift
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int i = 0; i < RW; ++i) {
                    const f4 a = {acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]};
                    f4 v = act_apply(a * dq[g] + bq[g] + nzr[i], ak);
                    if (p.res) v += f4{(float)rq[g][i][0], (float)rq[g][i][1], (float)rq[g][i][2], (float)rq[g][i][3]} * p.out_scale;
                    h4 out;
#pragma unroll
                    for (int q = 0; q < 4; ++q) out[q] = (half_t)v[q];
                    *(h4*)(Os + (i * 32 + lr) * OP + (8 * g + 4 * kh) * 2) = out;
                    if (TRGB) va[i][g] = out;
                }
            }
            if (TRGB) {
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const h8 wt = *(const h8*)(Trow + ((j * 2 + gp) * 2) * 16);
#pragma unroll
                    for (int i = 0; i < RW; ++i) {
                        const h8 wi = (trow_ok && ((tn >> 2) & 1) == i) ? wt : hzero;
                        rgb = mfma32(wi, __builtin_shufflevector(va[i][2 * gp], va[i][2 * gp + 1], 0, 1, 2, 3, 4, 5, 6, 7), rgb);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();        // LDS is in-order per wave: only pin the compiler's order
#pragma unroll
            for (int k = 0; k < 4; ++k) {           // 2 rows x 32 px x four 16-byte pieces of this 32-channel slice
                const int v = lane + 64 * k, i = v >> 7, pix = (v >> 2) & 31, piece = v & 3;
                half_t* dst = p.y + (((long long)b * p.Ho + oyb + i) * p.Wo + tx0 + pix) * p.Cout + j * 32 + piece * 8;
                *(h8*)dst = *(const h8*)(Os + (i * 32 + pix) * OP + piece * 16);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (TRGB) {
            const long long hw = (long long)p.Ho * p.Wo;
            float* yo = p.trgb_yout + (long long)b * 3 * hw + (long long)(oyb + kh) * p.Wo + ox;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                float r = p.trgb_b[cc] + (rgb[cc] + rgb[4 + cc] * (1.f / 2048.f));
                if (p.trgb_yprev) r += trgb_skip(ytap[cc], oyb + kh, ox);
                yo[cc * hw] = r;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the staged slices are read: patch buffer 1 may be refilled once every wave is past its epilogue
        if (reload) {                          // the next candidate's weights: every wave's pieces landed before anybody reads them
            WR_WAIT_VM0();
            __builtin_amdgcn_s_barrier();
        }
        have_w = !reload;                      // (the fragments requested in phase 5 were the OLD candidate's)
        if (!has_next) break;
        ++id;
        cur = nxt;
    }
    WR_WAIT_VM0();                             // the last tile's self-prefetch
}

// would a plain 3x3 layer of this geometry run here?  (the producers of its input ask before they write the chunk-planar layout)
bool conv_wres_supported(int Cin, int Cout, int H, int W) {
    static const bool off = glass_knob("GLASS_NO_WRES") != nullptr;       // A/B knob (developer build): conv_tiled<3,1,8,64> instead
    if (off || Cin != 64 || Cout != NT || H % TH != 0 || W % TW != 0 || (long long)H * W * Cin >= (1LL << 31)) return false;
    if (!glass_lds_fits(LDS_BYTES)) return false;
    // worth a persistent workgroup per CU only with several tiles each — judged at the nominal population (common.h), so that a layer runs on
    // the same kernel whatever the size of this launch
    return (long long)GLASS_NOMINAL_POP * (W / TW) * (H / TH) >= 8LL * glass_cu_count();
}

// nullptr: the layer does not qualify (the caller goes on to conv_tiled)
const char* launch_conv_wres(const ConvParams& p, hipStream_t st) {
    if (!conv_wres_supported(p.Cin, p.Cout, p.Hc, p.Wc)) return nullptr;
    if (p.Cin != 64 || p.Neff != NT || p.Cout != NT || p.up || p.y32 || !p.y || p.KS != 3 || p.stride != 1 || p.pad != 1) return nullptr;
    if (p.sn || p.sn16 || p.pre_shift || p.in_up || p.rgb_y || p.rgb_tanh_out || p.skip_x || p.post_scale16 || p.trgb_part) return nullptr;
    if (p.Hc % TH != 0 || p.Wc % TW != 0 || p.Hc != p.H || p.Wc != p.W || (p.x_bstride == 0 && p.B > 1)) return nullptr;
    if (p.xs_out && p.trgb_yout) return nullptr;
    if (p.y_planar32) return nullptr;              // (reads the chunk-planar layout, writes pixel-major)
    if (p.trgb_yout && (!p.trgb_tab || !p.trgb_b)) return nullptr;
    const int tiles_x = p.Wc / TW, tiles_y = p.Hc / TH;
    const int n_cu = glass_cu_count();
    const int PT = p.B * tiles_x * tiles_y;
    const char* name = p.trgb_yout ? "conv_wres_kernel<true,false>" : p.xs_out ? "conv_wres_kernel<false,true>" : "conv_wres_kernel<false,false>";
    if (p.dry_run) return name;
    static DevOnce once;
    once.run([&] {
        (void)hipFuncSetAttribute((const void*)conv_wres_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv_wres_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv_wres_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    });
    // contiguous tile ranges: the tiles of a candidate split over whole workgroups where they can (per-sample weights load once per range)
    const int per_wg = (PT + n_cu - 1) / n_cu;
    const int grid = (PT + per_wg - 1) / per_wg;
    if (p.trgb_yout) hipLaunchKernelGGL((conv_wres_kernel<true, false>), dim3(grid), dim3(NTHR), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_wg);
    else if (p.xs_out) hipLaunchKernelGGL((conv_wres_kernel<false, true>), dim3(grid), dim3(NTHR), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_wg);
    else hipLaunchKernelGGL((conv_wres_kernel<false, false>), dim3(grid), dim3(NTHR), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_wg);
    return name;
}
