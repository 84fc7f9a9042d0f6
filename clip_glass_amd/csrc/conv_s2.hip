// conv_s2.hip — the discriminator's stride-2 3x3 convolution + 1x1 skip branch (stylegan2/modules.py:1238-1254, 1587-1601), staged by
// LDS-DMA (`global_load_lds_dwordx4`) on a persistent ring, for the blocks below full resolution.
//
// Why (round 3, profiles/r03_phase_trace_stride2.txt): in the register-staged conv_tiled<3,2,4,128,skip> a K stage of 24 MFMAs per
// wave (768 MFMA cycles) takes 5700-6000 cycles — 1050-1120 of them are the four waves writing the stage's 37 KB to LDS with
// ds_write_b128 (~79 B/clk per CU), 560-800 waiting for the operands, 1050 the two barriers, 700-1200 issuing the next stage's loads;
// deeper register prefetch does not help (measured).  Here nothing goes through registers:
//   * ONE 512-thread workgroup per CU (8 waves = 4 row pairs x 2 n halves), tile = 8 output rows x 32 px x 128 channels — twice the
//     pixels per weight byte of the 4 x 32 tile;
//   * a stride-2 tile reads 17 x 65 input pixels per 32-channel chunk (71 KB): too much to double-buffer.  It is held as TWO row-parity
//     halves instead — the tap rows of a chunk run in the order ky = 1 (odd input rows), 0, 2 (even rows), so the odd half of the NEXT
//     chunk is refilled while the even half is read and vice versa; columns are de-interleaved (even | odd) so that a stride-2
//     fragment walk reads consecutive 64-byte LDS rows (XOR-swizzled chunks: conflict-free);
//   * weights: ring of three 24 KB slots, one (chunk, tap row) stage each, requested two stages ahead;
//   * the skip branch's 1x1 conv is a FOURTH stage of every chunk (its own 16 KB operand buffer, its 8 KB weight slice in the ring) into
//     separate accumulators, so the activation of the main branch needs no extra pass: out = (lrelu(acc + b) * sqrt2 + acc_skip) * s;
//   * persistent: a workgroup walks work items id, id + grid, ... and the ring runs on into the next item;
//   * raw s_barrier + counted s_waitcnt vmcnt (per wave: the tail DMA round of a half is issued by the waves that own it only).
// DMA schedule (after the barrier at the top of stage f of chunk c; W(g) = weight slice of global stage g):
//     f = 0 (ky 1, odd half):   skip operand(c), W(g + 2)
//     f = 1 (ky 0, even half):  odd half(c + 1), W(g + 2) [the skip weights]
//     f = 2 (ky 2, even half):  W(g + 2)
//     f = 3 (skip):             even half(c + 1), W(g + 2)
// so every operand has at least two stages of MFMA time to land ("c + 1" runs on into the next work item).
// The input of this layer is never padded (pad 0: a (2 Ho + 1)^2 blurred map), so there is no zero page.
#include "common.h"
#include "kernels.h"
#include <stdio.h>
#include <stdlib.h>

// dev tool (GLASS_S2_TRACE=path): shader-clock stamps of the first stages of ONE workgroup in the middle of the grid (TR instance only)
__device__ unsigned long long* g_s2_trace = nullptr;
#define S2TRACE(ph) \
    if (TR && blockIdx.x == gridDim.x / 2 + 3 && (threadIdx.x & 63) == 0 && gst < 96) \
        g_s2_trace[(gst * 8 + (ph)) * 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime()

namespace {
constexpr int NT = 128, NTHR = 512, TH = 8;
constexpr int PXR = 65;                                  // input pixels per patch row
constexpr int ODD_V = 8 * PXR * 4, EVEN_V = 9 * PXR * 4; // 16-byte vectors of the two halves (2080, 2340)
constexpr int ODD_WR = (ODD_V + 63) / 64, EVEN_WR = (EVEN_V + 63) / 64;   // wave-rounds (64 vectors each): 33, 37
constexpr int OFF_ODD = 0;
// every piece is a FULL 64-lane instruction (an LDS-DMA under a lane mask in divergent control flow is what hipcc mis-merged in conv_wreg.hip,
// DESIGN "Round 6"): the idle lanes of a half's last, partial round fetch a zero page into the slack behind the half
constexpr int OFF_EVEN = ODD_WR * 1024;                  // 33792 (33280 used)
constexpr int OFF_XS = OFF_EVEN + EVEN_WR * 1024;        // 71680 (37440 used): skip operand [8][32] px x 64 B
constexpr int XS_BYTES = TH * 32 * 64;                   // 16384
constexpr int OFF_W = OFF_XS + XS_BYTES;                 // 88064
constexpr int W_SLOT = 3 * NT * 64;                      // 24576
constexpr int OFF_C = OFF_W + 3 * W_SLOT;                // 161792: bias [Neff <= 512] fp32
constexpr int MAX_N = 512;
constexpr int LDS_BYTES = OFF_C + MAX_N * 4;             // 163840 of 163840
static_assert(LDS_BYTES <= 163840, "one workgroup per CU");
__device__ __attribute__((aligned(64))) half_t g_s2_zero_page[32];   // zero-initialised: what the idle lanes of a partial round fetch

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
__device__ __forceinline__ void dma16(const half_t* src, char* lds_wave_base) {   // LDS destination = wave-uniform base + lane * 16
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int opq(int v) { asm volatile("" : "+v"(v)); return v; }
#define S2_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
}  // namespace

template <bool TR, bool IL>
__global__ __launch_bounds__(512, 1) void conv_s2_kernel(ConvParams p, int NTn, int tiles_x, int tiles_y, int PT) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tpi = tiles_x * tiles_y;
    const int n_work = ((PT + 7) & ~7) * NTn;
    struct Item { int b, ty0, tx0, n0; bool valid; };
    auto decode = [&](int id) {   // work item -> (pixel tile, n tile): the n tiles of one pixel tile sit on one XCD (id % 8)
        Item w;
        const int lo = id & 7, rest = id >> 3;
        const int nt = rest % NTn, pt = (rest / NTn) * 8 + lo;
        w.valid = id < n_work && pt < PT;
        const int ptc = w.valid ? pt : 0;
        w.b = ptc / tpi;
        const int trem = ptc - w.b * tpi;
        w.ty0 = (trem / tiles_x) * TH;
        w.tx0 = (trem % tiles_x) * 32;
        w.n0 = nt * NT;
        return w;
    };
    int id = blockIdx.x;
    Item cur = decode(id);
    while (id < n_work && !cur.valid) { id += gridDim.x; cur = decode(id); }
    if (id >= n_work) return;

    // ---- per-thread DMA source offsets (elements), tile- and chunk-independent: every tile is interior -----------------------------
    // a half's vector v sits at LDS byte v * 16: pixel P = v >> 2 (row P / 65, de-interleaved column q = P % 65), physical chunk v & 3
    // holding the pixel's LOGICAL 8-channel chunk (v & 3) ^ ((P >> 2) & 3).  Wave-round r = k * 8 + wave covers vectors r * 64 + lane.
    // input layout: pixel-major [H][W][Cin], or 32-channel planes [Cin / 32][H][W][32] (common.h x_planar32: a line then belongs to ONE K step)
    const int pixs = p.x_planar32 ? 32 : p.Cin;
    const long long c_step = p.x_planar32 ? (long long)p.H * p.W * 32 : 32;
    int o_src[5], e_src[5];        // -1: this lane does not take part in the (partial) round
    int na_o = 0, na_e = 0;        // wave-uniform: DMA instructions this wave issues per half
    {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int r = k * 8 + wave;
            {
                const int v = r * 64 + lane, P = v >> 2, row = P / PXR, q = P - row * PXR;
                const int col = q < 33 ? 2 * q : 2 * (q - 33) + 1;
                const int lc = (v & 3) ^ ((P >> 2) & 3);
                o_src[k] = (r < ODD_WR && v < ODD_V) ? ((2 * row + 1) * p.W + col) * pixs + lc * 8 : -1;
                if (r < ODD_WR) na_o = k + 1;
            }
            {
                const int v = r * 64 + lane, P = v >> 2, row = P / PXR, q = P - row * PXR;
                const int col = q < 33 ? 2 * q : 2 * (q - 33) + 1;
                const int lc = (v & 3) ^ ((P >> 2) & 3);
                e_src[k] = (r < EVEN_WR && v < EVEN_V) ? ((2 * row) * p.W + col) * pixs + lc * 8 : -1;
                if (r < EVEN_WR) na_e = k + 1;
            }
        }
    }
    int w_src[3], x_src[2], ws_src;
    {
        const int t = threadIdx.x;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int v = k * NTHR + t, row = v >> 2;          // row = tx * 128 + n
            const int tx = row >> 7, n = row & 127;
            w_src[k] = (tx * p.Neff + n) * p.Cin + (((v & 3) ^ ((row >> 2) & 3)) << 3);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int v = k * NTHR + t, px = v >> 2;           // px = row * 32 + col of the output tile
            x_src[k] = ((px >> 5) * p.Wo + (px & 31)) * p.Cin + (((v & 3) ^ ((px >> 2) & 3)) << 3);
        }
        ws_src = (t >> 2) * p.Cin + (((t & 3) ^ (((t >> 2) >> 2) & 3)) << 3);
    }
    na_o = __builtin_amdgcn_readfirstlane(na_o);
    na_e = __builtin_amdgcn_readfirstlane(na_e);

    const int n_chunks = p.Cin >> 5;
    // the item whose operands are being LOADED (the current one, or the next one near the end of an item)
    struct Src { const half_t* hb; const half_t* xs; const half_t* w; const half_t* ws; };
    auto src_of = [&](const Item& it) {
        Src s;
        s.hb = p.x + (long long)it.b * p.x_bstride + ((long long)(2 * it.ty0) * p.W + 2 * it.tx0) * pixs;
        s.xs = p.skip_x + (((long long)it.b * p.Ho + it.ty0) * p.Wo + it.tx0) * p.Cin;
        s.w = p.w + (long long)it.n0 * p.Cin;
        s.ws = p.skip_w + (long long)it.n0 * p.Cin;
        return s;
    };
    // one DMA instruction (1 KB per wave) of each operand: k-th wave-round of a half / the skip operand / a weight slice
    auto odd_piece = [&](const Src& s, int c, int k) {
        if (k < na_o) {
            dma16(o_src[k] >= 0 ? s.hb + o_src[k] + c * c_step : g_s2_zero_page, smem + OFF_ODD + wave * 1024 + k * 8192);
        }
    };
    auto even_piece = [&](const Src& s, int c, int k) {
        if (k < na_e) {
            dma16(e_src[k] >= 0 ? s.hb + e_src[k] + c * c_step : g_s2_zero_page, smem + OFF_EVEN + wave * 1024 + k * 8192);
        }
    };
    auto xs_piece = [&](const Src& s, int c, int k) { dma16(s.xs + x_src[k] + c * 32, smem + OFF_XS + wave * 1024 + k * 8192); };
    // weight slice of stage (chunk c, phase f): f = 0, 1, 2 -> tap row ky = 1, 0, 2 (3 pieces); f = 3 -> skip weights (1 piece)
    auto w_piece = [&](const Src& s, int c, int f, int slot, int k) {
        char* dst = smem + OFF_W + slot * W_SLOT + wave * 1024;
        if (f < 3) {
            const int ky = f == 0 ? 1 : (f == 1 ? 0 : 2);
            dma16(s.w + (long long)ky * 3 * p.Neff * p.Cin + c * 32 + w_src[k], dst + k * 8192);
        } else {
            dma16(s.ws + ws_src + c * 32, dst);
        }
    };
    auto issue_odd = [&](const Src& s, int c) {
#pragma unroll
        for (int k = 0; k < 5; ++k) odd_piece(s, c, k);
    };
    auto issue_even = [&](const Src& s, int c) {
#pragma unroll
        for (int k = 0; k < 5; ++k) even_piece(s, c, k);
    };
    auto issue_w = [&](const Src& s, int c, int f, int slot) {
#pragma unroll
        for (int k = 0; k < (f < 3 ? 3 : 1); ++k) w_piece(s, c, f, slot, k);
    };

    // bias of every output channel -> LDS once (the n tile changes from item to item; a global load in the loop would make the
    // epilogue wait for the DMA queue behind it)
    {
        float* Cc = (float*)(smem + OFF_C);
        for (int n = threadIdx.x; n < p.Neff; n += NTHR) Cc[n] = p.bias ? p.bias[n] : 0.f;
        __syncthreads();
    }
    Src cs = src_of(cur);
    // prologue: W(0), odd(0), even(0), W(1) — the order the steady state leaves behind at the top of a chunk
    issue_w(cs, 0, 0, 0);
    issue_odd(cs, 0);
    issue_even(cs, 0);
    issue_w(cs, 0, 1, 1);
    int gst = 0;                                // (trace only) stages run so far
    int slot = 0;                               // weight slot of the stage about to run (stage g lives in slot g % 3)
    const int wr = wave & 3, wn = wave >> 2;
    for (;;) {
        int nid = id + gridDim.x;
        Item nxt = decode(nid);
        while (nid < n_work && !nxt.valid) { nid += gridDim.x; nxt = decode(nid); }
        const bool has_next = nid < n_work;
        const Src ns = src_of(has_next ? nxt : cur);
        const int b = cur.b, ty0 = cur.ty0, tx0 = cur.tx0, n0 = cur.n0;

        f16x acc[2][2], acs[2][2];              // main / skip branch accumulators: [tile row of the pair][32-wide n block of the half]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) { acc[i][j][q] = 0.f; acs[i][j][q] = 0.f; }
        for (int c = 0; c < n_chunks; ++c) {
            const bool last_c = c + 1 == n_chunks;
            const bool more = !last_c || has_next;                  // a chunk follows this one (possibly the next item's first)
            const Src& nsrc = last_c ? ns : cs;
            const int nc = last_c ? 0 : c + 1;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                S2TRACE(0);
                // s_waitcnt vmcnt(N), N = DMA instructions this wave issued AFTER the last operand of this stage (the counter retires in order):
                if (f == 0) {                      // even(c) [na_e], W(g + 1) [3] behind odd(c) / W(g)
                    if (na_e == 5) S2_WAIT(8); else S2_WAIT(7);
                } else if (f == 1) {               // skip operand(c) [2], W(g + 1) [3] behind W(g)
                    S2_WAIT(5);
                } else if (f == 2) {               // odd(c + 1) [na_o], skip weights [1] behind W(g)
                    if (!more) S2_WAIT(1); else if (na_o == 5) S2_WAIT(6); else S2_WAIT(5);
                } else {                           // W(g + 1) [3] behind the skip weights
                    if (more) S2_WAIT(3); else S2_WAIT(0);
                }
                S2TRACE(1);
                __builtin_amdgcn_s_barrier();      // this stage's operands are visible to every wave; whatever stage g - 1 read is free
                S2TRACE(2);
                const int slot2 = slot == 0 ? 2 : slot - 1;          // (g + 2) % 3
                // the i-th DMA instruction of this stage (the order fixes the vmcnt counts above):
                //   f = 0: skip operand(c) [2], W(g + 2) [3]        f = 1: odd(c + 1) [<= 5], skip weights [1]
                //   f = 2: W(g + 2) [3]                              f = 3: even(c + 1) [<= 5], W(g + 2) [3]
                auto piece = [&](int i) {
                    if (f == 0) {
                        if (i < 2) xs_piece(cs, c, i);
                        else if (i < 5) w_piece(cs, c, 2, slot2, i - 2);
                    } else if (f == 1) {
                        if (i < 5) { if (more) odd_piece(nsrc, nc, i); }
                        else if (i == 5) w_piece(cs, c, 3, slot2, 0);
                    } else if (f == 2) {
                        if (i < 3 && more) w_piece(nsrc, nc, 0, slot2, i);
                    } else {
                        if (i < 5) { if (more) even_piece(nsrc, nc, i); }
                        else if (i < 8 && more) w_piece(nsrc, nc, 1, slot2, i - 5);
                    }
                };
                if (!IL) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) piece(i);
                }
                S2TRACE(3);
                // ---- MFMAs ---------------------------------------------------------------------------------------------------------
                const int tm = opq(threadIdx.x), lr = tm & 31, kh = (tm >> 5) & 1;
                const char* Ws = smem + OFF_W + slot * W_SLOT;
                __builtin_amdgcn_s_setprio(1);
                if (f < 3) {
                    const int ky = f == 0 ? 1 : (f == 1 ? 0 : 2);
                    const char* As = smem + (f == 0 ? OFF_ODD : OFF_EVEN);
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) {
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            const int lc = kk * 2 + kh;
                            h8 wf[2];
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const int row = tx * NT + (wn * 2 + j) * 32 + lr;
                                wf[j] = *(const h8*)(Ws + row * 64 + ((lc ^ ((row >> 2) & 3)) << 4));
                            }
                            h8 xf[2];
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const int r = wr * 2 + i;                                    // output row of the tile
                                const int prow = ky == 1 ? r : r + (ky >> 1);                // row of the half: input row 2 r + ky
                                const int q = (tx & 1) ? 33 + lr : lr + (tx >> 1);           // de-interleaved column of input column 2 lr + tx
                                const int P = prow * PXR + q;
                                xf[i] = *(const h8*)(As + P * 64 + ((lc ^ ((P >> 2) & 3)) << 4));
                            }
                            if (IL) {               // one DMA instruction per block of 4 MFMAs: its issue cost hides under the MFMAs in flight
                                piece(tx * 2 + kk);
                                if (tx == 2 && kk == 1) { piece(6); piece(7); }
                            }
#pragma unroll
                            for (int i = 0; i < 2; ++i)
#pragma unroll
                                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(wf[j], xf[i], acc[i][j]);
                        }
                    }
                } else {
                    const char* Xs = smem + OFF_XS;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const int lc = kk * 2 + kh;
                        h8 wf[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int row = (wn * 2 + j) * 32 + lr;
                            wf[j] = *(const h8*)(Ws + row * 64 + ((lc ^ ((row >> 2) & 3)) << 4));
                        }
                        h8 xf[2];
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int P = (wr * 2 + i) * 32 + lr;
                            xf[i] = *(const h8*)(Xs + P * 64 + ((lc ^ ((P >> 2) & 3)) << 4));
                        }
                        if (IL) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) piece(kk * 4 + i);
                        }
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) acs[i][j] = mfma32(wf[j], xf[i], acs[i][j]);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                S2TRACE(4);
                if (TR) ++gst;
                slot = slot == 2 ? 0 : slot + 1;
            }
        }

        // ---- epilogue: out = (act(acc + bias) + acc_skip) * out_scale.  In flight meanwhile: the next item's odd AND even halves and its
        // first two weight slices, so the only free LDS is the skip operand buffer: 2 KB per wave = one tile row x 32 channels, staged
        // (16-byte pieces XOR-swizzled by the pixel: conflict-free without padding) and stored as 16-byte vectors in row order. -------
        {
            const int t = opq(threadIdx.x), lane = t & 63, lr = lane & 31, kh = lane >> 5;
            const float* Cc = (const float*)(smem + OFF_C) + n0;
            __builtin_amdgcn_s_barrier();          // every wave is done with the skip operand buffer (its LDS reads have returned)
            if (TR) { --gst; S2TRACE(5); }
            char* Os = smem + OFF_XS + wave * 2048;
            const ActK ak = act_consts(p.act, p.out_scale);
            const int oyb = ty0 + wr * 2;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f4 bq[4];                                   // the slice's four bias quads as ONE batch of LDS reads (see conv_glds.hip)
#pragma unroll
                for (int g = 0; g < 4; ++g) bq[g] = *(const f4*)(Cc + (wn * 2 + j) * 32 + 8 * g + 4 * kh);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f4 bb = bq[g];
                        const f4 a = {acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]};
                        const f4 sk = {acs[i][j][g * 4], acs[i][j][g * 4 + 1], acs[i][j][g * 4 + 2], acs[i][j][g * 4 + 3]};
                        const f4 v = act_apply(a + bb, ak) + sk * p.out_scale;
                        h4 out;
#pragma unroll
                        for (int q = 0; q < 4; ++q) out[q] = (half_t)v[q];
                        *(h4*)(Os + lr * 64 + ((g ^ ((lr >> 2) & 3)) << 4) + kh * 8) = out;
                    }
                    __builtin_amdgcn_wave_barrier();        // LDS is in-order per wave: only pin the compiler's order
#pragma unroll
                    for (int k = 0; k < 2; ++k) {           // 32 px x four 16-byte pieces
                        const int v = lane + 64 * k, pix = v >> 2, piece = v & 3;
                        half_t* dst = p.y + (((long long)b * p.Ho + oyb + i) * p.Wo + tx0 + pix) * p.Cout + n0 + (wn * 2 + j) * 32 + piece * 8;
                        *(h8*)dst = *(const h8*)(Os + pix * 64 + ((piece ^ ((pix >> 2) & 3)) << 4));
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (TR) { S2TRACE(6); ++gst; }
        if (!has_next) break;
        id = nid;
        cur = nxt;
        cs = ns;
    }
}

// Tried and dropped (round 3, same-box A/B on the four D layers; tools/trace_s2.py stamps): a two-stage version — skip + ky 1 /
// ky 0 + ky 2, weights double-buffered as 32 + 48 KB, the skip operand loaded straight into registers, fragment reads software-pipelined
// one MFMA block ahead — was 3-6 % SLOWER (3.09 vs 2.92 ms): half the barriers, but one stage of DMA lead instead of two, and the waits
// at the stage tops grew by what the barriers saved.  Bunching a stage's DMA instructions at its front is worse again (+4 %): eight waves'
// pieces queue behind each other.  What bounds both: per 32-channel chunk a workgroup moves 166 KB L2 -> LDS (21 DMA instructions per
// wave, 100-185 cycles of issue each beside MFMAs) and reads 640 KB of fragments back (one ds_read_b128 per MFMA at 2 x 2 register
// blocking: 2560 LDS cycles) for 5120 MFMA cycles per SIMD — the three pipes are within 2x of each other, so ~45 % of the MFMA rate is
// what this tile shape gives; a bigger tile does not fit 160 KB of LDS / 256 registers.

const char* launch_conv_s2(const ConvParams& p, hipStream_t st, bool force) {
    static const bool off = glass_knob("GLASS_NO_S2DMA") != nullptr;      // A/B knob: the register-staged conv_tiled<3,2,4,128,skip> instead
    if (p.x_planar8 || p.y_planar8) return nullptr;   // chunk-planar maps (common.h): not implemented here
    if ((off && !force) || !p.skip_x || !p.skip_w || p.KS != 3 || p.stride != 2 || p.pad != 0 || p.up || p.y32 || !p.y) return nullptr;
    if (p.res || p.dscale || p.noise || p.shift || p.sn || p.pre_shift || p.in_up || p.xs_out || p.trgb_yout || p.post_scale16 || p.rgb_y) return nullptr;
    if (p.Neff != p.Cout || p.Neff % NT != 0 || p.Neff > MAX_N || p.Cin % 32 != 0 || p.Hc % TH != 0 || p.Wc % 32 != 0) return nullptr;
    if (p.H != 2 * p.Hc + 1 || p.W != 2 * p.Wc + 1 || p.Ho != p.Hc || p.Wo != p.Wc || p.w_bstride != 0) return nullptr;
    if (p.x_bstride != (long long)p.H * p.W * p.Cin) return nullptr;
    if ((long long)p.H * p.W * p.Cin >= (1LL << 31) || 9LL * p.Neff * p.Cin >= (1LL << 31) || (long long)p.Ho * p.Wo * p.Cin >= (1LL << 31)) return nullptr;
    if (!glass_lds_fits(LDS_BYTES)) return nullptr;          // 162 880 B: conv_tiled<3,2,..,skip> where the device offers less
    const int tiles_x = p.Wc / 32, tiles_y = p.Hc / TH;
    const int PT = p.B * tiles_x * tiles_y;
    const int NTn = p.Neff / NT;
    const int n_work = ((PT + 7) & ~7) * NTn;
    const int n_cu = glass_cu_count() - glass_cu_count() % 8;          // a workgroup keeps its XCD (id % 8) across items
    // too small to fill the chip with one workgroup per CU — judged at the nominal population (common.h), so that a layer runs on the
    // same kernel whatever the size of this launch
    if ((long long)GLASS_NOMINAL_POP * tiles_x * tiles_y * NTn < n_cu && !force) return nullptr;
    const int grid = n_work < n_cu ? n_work : n_cu;                     // (n_work is a multiple of 8)
    if (p.dry_run) return "conv_s2_kernel";
    static DevOnce once;
    once.run([&] {
        (void)hipFuncSetAttribute((const void*)conv_s2_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
#ifdef GLASS_DEV_TRACE
        (void)hipFuncSetAttribute((const void*)conv_s2_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
#endif
        (void)hipFuncSetAttribute((const void*)conv_s2_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    });
#ifdef GLASS_DEV_TRACE      // dev build (make TRACE=1): traced instance, stamps to a file; synchronises, single engine only
    if (const char* tp = getenv("GLASS_S2_TRACE")) {          // dev tool: traced instance, stamps to a file
        unsigned long long* dtr = nullptr;
        constexpr int NTR = 96 * 8 * 8;
        (void)hipMalloc(&dtr, NTR * sizeof(unsigned long long));
        (void)hipMemset(dtr, 0, NTR * sizeof(unsigned long long));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_s2_trace), &dtr, sizeof dtr);
        hipLaunchKernelGGL((conv_s2_kernel<true, true>), dim3(grid), dim3(NTHR), LDS_BYTES, st, p, NTn, tiles_x, tiles_y, PT);
        static unsigned long long hb[NTR];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hb, dtr, sizeof hb, hipMemcpyDeviceToHost);
        (void)hipFree(dtr);
        if (FILE* f = fopen(tp, "a")) {
            fprintf(f, "# conv_s2_kernel<trace> Cin=%d Cout=%d Hc=%d B=%d: stage phase t[wave0..7]; phases 0 top, 1 operands landed, 2 after barrier, 3 DMAs issued, 4 MFMAs done, 5 epilogue barrier, 6 epilogue done\n", p.Cin, p.Cout, p.Hc, p.B);
            for (int i = 0; i < 96; ++i)
                for (int ph = 0; ph < 7; ++ph) {
                    fprintf(f, "%d %d", i, ph);
                    for (int w = 0; w < 8; ++w) fprintf(f, " %llu", hb[(i * 8 + ph) * 8 + w] ? hb[(i * 8 + ph) * 8 + w] - hb[0] : 0ULL);
                    fprintf(f, "\n");
                }
            fclose(f);
        }
        return "conv_s2_kernel<trace>";
    }
#endif
    static const bool no_il = glass_knob("GLASS_S2_NO_IL") != nullptr;     // A/B knob: every DMA of a stage issued before its MFMAs
    if (no_il) {
        hipLaunchKernelGGL((conv_s2_kernel<false, false>), dim3(grid), dim3(NTHR), LDS_BYTES, st, p, NTn, tiles_x, tiles_y, PT);
        return "conv_s2_kernel<noil>";
    }
    hipLaunchKernelGGL((conv_s2_kernel<false, true>), dim3(grid), dim3(NTHR), LDS_BYTES, st, p, NTn, tiles_x, tiles_y, PT);
    return "conv_s2_kernel";
}
