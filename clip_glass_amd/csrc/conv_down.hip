// conv_down.hip — the second half of the discriminator's full-resolution block in ONE kernel (gfx950):
//     h  -> FIR 4x4 (pad 2) -> conv3x3 stride 2 (32 -> 64) + bias + lrelu*sqrt2 \
//     xs -> conv1x1 (32 -> 64, no bias / activation)                             +-> (a + b) / sqrt2
// (stylegan2/modules.py:1204-1254 ConvDownLayer, 1587-1601 DiscriminatorConvBlock.forward).  `h` is the block's first conv
// output; `xs` is the block input after the skip branch's FIR (pad 1) + ::2, which the kernel that PRODUCES the block
// input emits as a by-product of the tile it already holds in LDS (conv_stream<fromrgb>), or launch_blur_down.
// As separate passes (blur, blur-down, 1x1 skip conv, stride-2 conv) the 1024^2 block of a 64-candidate population
// moved 30 GB through HBM for maps that are 5.4 GB in and 2.1 GB out; here the FIR is applied while the conv's input
// window is staged and the skip branch is four extra MFMAs per wave.
//
// The layer is HBM-/latency-bound (0.6 TFLOP over 7.5 GB): what matters is bytes in flight ("keep loads in flight").
//   * persistent workgroups (2 per CU, 256 threads), each walking a contiguous range of 4 x 32 output-pixel tiles;
//   * a tile's raw window (12 x 68 px of h, 32 channels = 52 KB) and its skip fragments (8 KB) are fetched into
//     REGISTERS two tiles ahead (two named register sets, 16 + 2 16-byte loads per thread each, unconditional + clamped;
//     the zero padding is a mask applied at use, on border tiles only) — ~120 KB of loads in flight per workgroup;
//   * thread (column, 8-channel group) owns a window column: the VERTICAL FIR runs on its registers (packed fp16) and the
//     result goes to LDS; waves then own whole rows for the HORIZONTAL FIR, which rewrites each row IN PLACE as the MFMA
//     operand image (even | odd columns de-interleaved for the stride-2 fragment walk) — LDS ops are in order per wave,
//     so no workgroup barrier sits between a row's reads and its rewrites;
//   * the 4 window columns a 64-column thread grid does not cover travel as four extra loads per thread (144 threads fetch
//     rows r .. r + 3 of one edge column each), get their vertical FIR in registers too, and sit in a small side image
//     that the four lanes per row that need them read in the horizontal pass;
//   * LDS images are dense 64-byte rows; the vertical-pass image rotates each column inside its aligned group of four
//     (stride-4 sliding-window reads and stride-1 writes both conflict-free), the operand image XOR-swizzles the
//     16-byte chunk by the column slot; both swizzles depend on the column only, so a row step is an immediate offset;
//   * all weights (9 x 64 x 32 main, 64 x 32 skip) and the bias stay in LDS for the workgroup's lifetime: steady state
//     issues window loads and output stores only (a late small load would retire the whole in-order vmcnt queue);
//   * the skip branch's B operand needs no staging: lane (pixel, k-half) loads its own MFMA fragment from xs;
//   * epilogue: bias + lrelu in the accumulators (the sqrt2 gain cancels against the merge's 1/sqrt2, which the skip
//     weights carry instead), + skip MFMAs; each wave transposes its tile row through
//     the one operand-image row only it reads (no workgroup barrier) and stores 16-byte vectors in row order.
// Register discipline: the compiler hoists every per-thread address term out of the persistent loop and, with the
// window sets owning the register file, spills them — and a scratch reload is a VMEM op whose wait retires every older
// window load.  Each phase therefore re-derives its lane geometry from an opaque copy of the thread id.  The loop has no
// exit between its two tile steps (hipcc's wait-count merge at the loop header otherwise stops counting the other
// set's refill as younger and every window wait drains the queue): an odd tile count is padded with a tile whose
// stores are masked.
#include "common.h"
#include "kernels.h"
#include <stdio.h>
#include <stdlib.h>

namespace {
constexpr int TH = 4, NT = 64, CIN = 32, VP = 65;   // VP: column slots per image row (64 window columns / 65 blurred columns)
constexpr int V_BYTES = 9 * VP * 64;          // vertical-pass image / operand image (in place): 37440
constexpr int W_BYTES = 9 * NT * 64;          // 36864
constexpr int EM_BYTES = 9 * 4 * 64;          // edge columns after the vertical FIR: 2304
constexpr int OFF_W = V_BYTES, OFF_EM = OFF_W + W_BYTES, OFF_C = OFF_EM + EM_BYTES, OFF_WS = OFF_C + NT * 4;
constexpr int LDS_BYTES = OFF_WS + NT * 64;   // 81728 -> two workgroups per CU

// vertical-pass image: column rotated inside its aligned group of 4 by the group index
__device__ __forceinline__ int vaddr(int row, int col, int cg) {
    return row * (VP * 64) + (((col & ~3) | ((col + (col >> 2)) & 3)) << 6) + (cg << 4);
}
// operand image: 16-byte chunk XOR-swizzled by the column slot
__device__ __forceinline__ int aaddr(int row, int slot, int lc) { return row * (VP * 64) + (slot << 6) + ((lc ^ ((slot >> 2) & 3)) << 4); }
// weight image rows (tap * 64 + n): chunk XOR-swizzled by the row (conv_glds.hip's layout)
__device__ __forceinline__ int waddr(int row, int lc) { return (row << 6) + ((lc ^ ((row >> 2) & 3)) << 4); }
__device__ __forceinline__ h8 fir4(h8 a, h8 b, h8 c, h8 d) {   // [1,3,3,1]/8, packed fp16
    return (a + d) * (half_t)0.125f + (b + c) * (half_t)0.375f;
}
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
struct RSet { h8 a[12]; h8 e[4]; h8 s[2]; int b, ty0, tx0, valid; };   // window column, edge-column rows, skip fragments, tile (SGPRs)
}  // namespace

struct DownParams {
    const half_t* h;    // [B][R][R][32]       first conv's output
    const half_t* xs;   // [B][R/2][R/2][32]   block input after FIR (pad 1) + ::2
    const half_t* w1;   // [9][64][32]
    const half_t* ws;   // [64][32]
    const float* b1;    // [64]
    half_t* y;          // [B][R/2][R/2][64]
    int B, R;
    int row_walk;       // A/B knob (GLASS_ROW_WALK): round 2's row-major tile walk
    unsigned long long* trace;   // phase timestamps of workgroup 0 (GLASS_DOWN_TRACE; nullable)
};
#define TRACE(ph)                                                                                                  \
    if (TR && p.trace && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && it - first < 64)                             \
        p.trace[((it - first) * 8 + (ph)) * 4 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime()

template <bool TR>      // TR: phase-timestamp build (dev tool); the production instance carries no trace code at all
__global__ __launch_bounds__(256, 2) void conv_down_kernel(DownParams p, int tiles_x, int tiles_y, int n_tiles, int per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Vs = smem;
    char* Ws = smem + OFF_W;
    float* Cb = (float*)(smem + OFF_C);
    const int tpi = tiles_x * tiles_y;
    const int R = p.R, Ro = R >> 1;
    const int first = blockIdx.x * per_block;
    const int last = min(first + per_block, n_tiles);
    if (first >= last) return;
    const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};

    // window + skip loads of the NEXT tile of this workgroup's walk into Rg: 12 rows of this thread's column, four rows of
    // the edge columns (thread (row r, edge column, channel group), r < 9, fetches raw rows r .. r + 3), two skip fragments.
    // Always 18 loads, all unconditional (clamped coordinates; tiles past the end re-read the last tile).
    int nx_it = first, nx_b, nx_ty, nx_tx;             // the walk: tile index -> (sample, tile row, tile column), no divisions
    {
        // The walk goes DOWN a tile column (tile row fastest): consecutive windows of a workgroup then share 4 of their 12 rows,
        // which the second one finds in L2 — with the row-major walk of round 2 every window's vertical halo came from HBM / MALL
        // (PMC: 7.66 GB fetched for 4.8 GB of input).
        nx_b = uni(first / tpi);
        const int trem = first - nx_b * tpi;
        if (p.row_walk) { nx_ty = uni(trem / tiles_x); nx_tx = uni(trem - nx_ty * tiles_x); }
        else { nx_tx = uni(trem / tiles_y); nx_ty = uni(trem - nx_tx * tiles_y); }
    }
    // The 18 loads of a refill are NOT issued as one burst: a burst keeps the CU's address unit busy for ~2000 cycles with every
    // wave of the workgroup stalled at issue.  They go out in three groups of six, threaded between the rows of the
    // horizontal pass, so the address unit works in the shadow of LDS / VALU work.
    const half_t* is_img = nullptr;      // state of the refill in progress (uniform / per-thread offsets)
    int is_oy = 0, is_rs = 0;
    auto issue_begin = [&](RSet& Rg) {
        Rg.valid = nx_it < last;
        Rg.b = nx_b; Rg.ty0 = nx_ty * TH; Rg.tx0 = nx_tx * 32;
        is_img = p.h + (long long)Rg.b * R * R * CIN;                            // uniform (SGPRs)
        is_oy = 2 * Rg.ty0 - 2;
        is_rs = R * CIN;
        if (nx_it + 1 < last) {        // advance the walk (uniform); past the end it stays on the last tile
            ++nx_it;
            if (p.row_walk) { if (++nx_tx == tiles_x) { nx_tx = 0; if (++nx_ty == tiles_y) { nx_ty = 0; ++nx_b; } } }
            else if (++nx_ty == tiles_y) { nx_ty = 0; if (++nx_tx == tiles_x) { nx_tx = 0; ++nx_b; } }
        } else {
            nx_it = last;
        }
    };
    auto issue_rows = [&](RSet& Rg, int k0) {     // window rows k0 .. k0 + 5 of this thread's column
        const int t = opaque(threadIdx.x), cg = t & 3, cs = t >> 2;
        const int xo = min(max(2 * Rg.tx0 - 2 + cs, 0), R - 1) * CIN + cg * 8;   // R * R * 32 < 2^31: 32-bit element offsets
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int iy = min(max(is_oy + k0 + k, 0), R - 1);                    // uniform
            Rg.a[k0 + k] = *(const h8*)(is_img + iy * is_rs + xo);
        }
    };
    auto issue_rest = [&](RSet& Rg) {             // edge-column rows and skip fragments
        const int t = opaque(threadIdx.x), cg = t & 3;
        const int er = min(t >> 4, 8);
        const int exo = min(max(2 * Rg.tx0 - 2 + 64 + ((t >> 2) & 3), 0), R - 1) * CIN + cg * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) Rg.e[k] = *(const h8*)(is_img + min(max(is_oy + er + k, 0), R - 1) * is_rs + exo);
        // skip fragments: lane (pixel lr of output row ty0 + wave, k-half kh) -> channels kk * 16 + kh * 8 .. + 7
        const int lr = t & 31, kh = (t >> 5) & 1, wave = t >> 6;
        const half_t* xp = p.xs + (((long long)Rg.b * Ro + Rg.ty0 + wave) * Ro + Rg.tx0 + lr) * CIN + kh * 8;
        Rg.s[0] = *(const h8*)xp;
        Rg.s[1] = *(const h8*)(xp + 16);
    };
    auto issue = [&](RSet& Rg) {
        issue_begin(Rg);
        issue_rows(Rg, 0);
        issue_rows(Rg, 6);
        issue_rest(Rg);
    };

    // ---- resident weights: main taps and skip rows in LDS (swizzled source chunk, linear destination), bias in LDS ------
    {
        const int t = threadIdx.x;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int v = k * 256 + t, row = v >> 2;               // row = tap * 64 + n
            const int lc = (v & 3) ^ ((row >> 2) & 3);
            *(h8*)(Ws + v * 16) = *(const h8*)(p.w1 + (long long)row * CIN + lc * 8);
        }
        // (lrelu(a + b1) * sqrt2 + skip) / sqrt2 = lrelu(a + b1) + skip / sqrt2: the skip rows carry the 1/sqrt2
        const int row = t >> 2, lc = (t & 3) ^ ((row >> 2) & 3);
        const h8 wsv = *(const h8*)(p.ws + (long long)row * CIN + lc * 8);
        h8 wss;
#pragma unroll
        for (int q = 0; q < 8; ++q) wss[q] = (half_t)((float)wsv[q] * 0.70710678118654752440f);
        *(h8*)(smem + OFF_WS + t * 16) = wss;
        if (t < NT) Cb[t] = p.b1[t];
    }

    auto step = [&](int it, RSet& Rg) {
        const int b = Rg.b, ty0 = Rg.ty0, tx0 = Rg.tx0, valid = Rg.valid;
        const int oy = 2 * ty0 - 2, ox = 2 * tx0 - 2;
        TRACE(0);
        __syncthreads();       // B0: every wave is done with the operand image (MFMA reads, own output transposition) of the previous tile
        TRACE(1);
        // ---- vertical FIR on this thread's window column -> LDS; edge vector parked raw -----------------------------
        {
            const int t = opaque(threadIdx.x), cg = t & 3, cs = t >> 2;
            const bool border = oy < 0 || ox < 0 || oy + 12 > R || ox + 68 > R;     // uniform: interior tiles need no padding mask
            if (border) {
                const bool colok = (unsigned)(ox + cs) < (unsigned)R;
#pragma unroll
                for (int k = 0; k < 12; ++k) Rg.a[k] = (colok && (unsigned)(oy + k) < (unsigned)R) ? Rg.a[k] : zero;
                const int er = t >> 4;
                const bool ecok = (unsigned)(ox + 64 + ((t >> 2) & 3)) < (unsigned)R;
#pragma unroll
                for (int k = 0; k < 4; ++k) Rg.e[k] = (ecok && (unsigned)(oy + er + k) < (unsigned)R) ? Rg.e[k] : zero;
            }
#pragma unroll
            for (int r = 0; r < 9; ++r) *(h8*)(Vs + vaddr(r, cs, cg)) = fir4(Rg.a[r], Rg.a[r + 1], Rg.a[r + 2], Rg.a[r + 3]);
            if (t < 144) *(h8*)(smem + OFF_EM + t * 16) = fir4(Rg.e[0], Rg.e[1], Rg.e[2], Rg.e[3]);   // [row][edge column][cg]
        }
        const h8 xs0 = Rg.s[0], xs1 = Rg.s[1];     // this tile's skip fragments (the set is refilled next)
        TRACE(2);
        issue_begin(Rg);       // refill, first third (two tiles of window loads stay in flight)
        issue_rows(Rg, 0);
        TRACE(3);
        __syncthreads();       // B1: vertical-pass image complete
        TRACE(4);
        // ---- horizontal FIR: each wave owns whole rows and rewrites them in place as the operand image ------------------
        {
            const int t = opaque(threadIdx.x), lane = t & 63, wave = uni(t >> 6);
            const int j = lane >> 2, cgl = lane & 3;
            auto hrow = [&](int rr) {
                h8 v[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = *(const h8*)(Vs + vaddr(rr, 4 * j + k, cgl));
                v[7] = zero;
                if (j < 15) {
#pragma unroll
                    for (int k = 4; k < 7; ++k) v[k] = *(const h8*)(Vs + vaddr(rr, 4 * j + k, cgl));
                } else {      // window columns 64..67: the edge image
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[4 + k] = *(const h8*)(smem + OFF_EM + ((rr * 4 + k) * 4 + cgl) * 16);
                }
                h8 o[5];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = fir4(v[i], v[i + 1], v[i + 2], v[i + 3]);
                o[4] = fir4(v[4], v[5], v[6], v[7]);
                __builtin_amdgcn_wave_barrier();     // reads of the row are issued before its rewrites (LDS is in order per wave)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = 4 * j + i;
                    const int slot = (c & 1) ? 33 + (c >> 1) : (c >> 1);
                    *(h8*)(Vs + aaddr(rr, slot, cgl)) = o[i];
                }
                if (j == 15) *(h8*)(Vs + aaddr(rr, 32, cgl)) = o[4];     // blurred column 64
                __builtin_amdgcn_wave_barrier();
            };
            hrow(wave);
            __builtin_amdgcn_sched_barrier(0);       // rows one at a time: three rows in flight would cost 150 VGPRs
            issue_rows(Rg, 6);                       // refill, second third
            hrow(wave + 4);
            __builtin_amdgcn_sched_barrier(0);
            issue_rest(Rg);                          // refill, last third
            if (wave == 0) hrow(8);
        }
        TRACE(5);
        __syncthreads();       // B2: operand image complete
        TRACE(6);
        // ---- MFMA: 9 taps x 2 k16 steps x 2 n blocks; wave = output row ty0 + wave -------------------------------------------
        const int tm = opaque(threadIdx.x), lr = tm & 31, kh = (tm >> 5) & 1, wave = uni(tm >> 6);
        f16x acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int lc = kk * 2 + kh;
                    const h8 xf = *(const h8*)(Vs + aaddr(2 * wave + ky, (kx == 1 ? 33 : (kx >> 1)) + lr, lc));
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const h8 wf = *(const h8*)(Ws + waddr((ky * 3 + kx) * NT + j * 32 + lr, lc));
                        acc[j] = mfma32(wf, xf, acc[j]);
                    }
                    if (kk && kx == 2) __builtin_amdgcn_sched_barrier(0);   // one tap ROW's fragments live at a time (the window sets own the registers)
                }
        TRACE(7);
        // ---- activation in the accumulators, then the skip branch on top -------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f4 bb = *(const f4*)(Cb + j * 32 + 8 * g + 4 * kh);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = acc[j][g * 4 + q] + bb[q];
                    acc[j][g * 4 + q] = fmaxf(v, 0.2f * v);        // lrelu; its sqrt2 gain cancels against the merge's 1/sqrt2
                }
            }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const h8 wf = *(const h8*)(smem + OFF_WS + waddr(j * 32 + lr, kk * 2 + kh));
                acc[j] = mfma32(wf, kk ? xs1 : xs0, acc[j]);
            }
        // ---- epilogue: lane = pixel lr of output row ty0 + wave; transposition through operand-image row 2 * wave + 1, which
        // only this wave reads (rows 2w and 2w + 2 are shared with the neighbouring waves) --------------------------------------
        const int lane = tm & 63;
        char* Os = Vs + (2 * wave + 1) * (VP * 64);      // 32 px x 128 B, 16-byte chunk XOR-swizzled by the pixel
        __builtin_amdgcn_wave_barrier();                  // this wave's fragment reads are issued before the rewrites
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h4 out;
#pragma unroll
                for (int q = 0; q < 4; ++q) out[q] = (half_t)acc[j][g * 4 + q];
                *(h4*)(Os + lr * 128 + (((j * 4 + g) ^ (lr & 7)) << 4) + kh * 8) = out;
            }
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            half_t* yrow = p.y + (((long long)b * Ro + ty0 + wave) * Ro + tx0) * NT;
#pragma unroll
            for (int k = 0; k < NT / 16; ++k) {
                const int v = lane + 64 * k;
                const int pix = v >> 3, chv = v & 7;
                *(h8*)(yrow + (long long)pix * NT + chv * 8) = *(const h8*)(Os + pix * 128 + ((chv ^ (pix & 7)) << 4));
            }
        }
    };

    RSet r0, r1;
    issue(r0);
    issue(r1);
    for (int it = first; it < last; it += 2) {     // no exit between the two steps (see the header)
        step(it, r0);
        step(it + 1, r1);
    }
}

bool conv_down_supported(int R, int Cin, int Cout) {
    static const bool off = glass_knob("GLASS_NO_DOWN") != nullptr;   // A/B knob
    return !off && glass_lds_fits(LDS_BYTES) && R % 64 == 0 && R >= 64 && Cin == CIN && Cout == NT && (long long)R * R * Cin < (1LL << 31);
}

// Returns the kernel symbol, or nullptr when the block does not qualify (caller runs the separate passes).
const char* launch_conv_down(const half_t* h, const half_t* xs, const half_t* w1, const half_t* ws, const float* b1, half_t* y,
                             int B, int R, int Cin, int Cout, hipStream_t st) {
    if (!conv_down_supported(R, Cin, Cout)) return nullptr;
    DownParams p;
    p.h = h; p.xs = xs; p.w1 = w1; p.ws = ws; p.b1 = b1; p.y = y; p.B = B; p.R = R;
    p.trace = nullptr;
    static const bool row_walk = glass_knob("GLASS_ROW_WALK") != nullptr;
    p.row_walk = row_walk ? 1 : 0;
    const char* trace_path = nullptr;
#ifdef GLASS_DEV_TRACE      // dev build (make TRACE=1): per-phase shader-clock timestamps of workgroup 0; synchronises, single engine only
    trace_path = getenv("GLASS_DOWN_TRACE");
#endif
    if (trace_path) (void)hipMalloc(&p.trace, 64 * 8 * 4 * sizeof(unsigned long long));
    if (p.trace) (void)hipMemsetAsync(p.trace, 0, 64 * 8 * 4 * sizeof(unsigned long long), st);
    const int Ro = R / 2, tiles_x = Ro / 32, tiles_y = Ro / TH;
    const long long tiles = (long long)B * tiles_x * tiles_y;
    if (tiles >= (1LL << 30)) return nullptr;
    static DevOnce once;
    once.run([&] {
        (void)hipFuncSetAttribute((const void*)conv_down_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv_down_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    });
    const int slots = glass_cu_count() * 2;
    const int per_block = (int)((tiles + slots - 1) / slots);
    const int grid = (int)((tiles + per_block - 1) / per_block);
    if (p.trace) hipLaunchKernelGGL(conv_down_kernel<true>, dim3(grid), dim3(256), LDS_BYTES, st, p, tiles_x, tiles_y, (int)tiles, per_block);
    else hipLaunchKernelGGL(conv_down_kernel<false>, dim3(grid), dim3(256), LDS_BYTES, st, p, tiles_x, tiles_y, (int)tiles, per_block);
    if (p.trace) {
        static unsigned long long hbuf[64 * 8 * 4];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hbuf, p.trace, sizeof hbuf, hipMemcpyDeviceToHost);
        (void)hipFree(p.trace);
        if (FILE* f = fopen(trace_path, "w")) {
            fprintf(f, "# tile phase: t[wave0..3] (shader clocks, relative to the first stamp); phases: 0 enter, 1 after B0, 2 after vertical pass, "
                       "3 after refill issue, 4 after B1, 5 after horizontal pass, 6 after B2, 7 after main MFMAs; per_block=%d\n", per_block);
            const unsigned long long t0 = hbuf[0];
            for (int i = 0; i < 64 && i < per_block; ++i)
                for (int ph = 0; ph < 8; ++ph) {
                    fprintf(f, "%d %d", i, ph);
                    for (int w = 0; w < 4; ++w) fprintf(f, " %llu", hbuf[(i * 8 + ph) * 4 + w] ? hbuf[(i * 8 + ph) * 4 + w] - t0 : 0ULL);
                    fprintf(f, "\n");
                }
            fclose(f);
        }
    }
    return "conv_down_kernel";
}
