// bg_tail.hip — BigGAN-deep's last stage in one kernel (round 4): the final up-block's conv_3 (1x1, 32 -> 128) + skip, the generator's
// unconditional bn + relu, conv_to_rgb (3x3, 128 -> 3 used channels) and tanh (oracle/biggan_ref.py gen_block() / generator(); the
// published BigGAN-deep generator, pytorch-pretrained-biggan 0.1.1 — see the oracle's header for the pin status).
//
// Why: the 128-channel 512 x 512 map between the two convolutions is 4.3 GB per population of 64 — written once (bg.b13.conv3, 1.42 ms at
// the HBM write roof) and read once (bg.final.conv_to_rgb, 1.6 ms, whose MFMAs compute 32 output columns for 3 used ones).  Here it
// never exists:
//   * conv_3's MFMA output D[channel][pixel] is, lane for lane, the B operand of a second MFMA chain over the 128 channels (K order =
//     accumulator-lane order, the weight fragments are gathered in that order once per workgroup — the toRGB fusion's trick);
//   * that second chain does NOT compute the 3x3 convolution tap by tap: its 32 output columns are the 27 (tap, colour) PARTIAL products
//     P[pixel][tap * 3 + c] = sum_ch z[pixel][ch] * W[tap][c][ch], so every MFMA column but five is used and the 9 taps cost 8 MFMAs per
//     32 pixels instead of 72; out[y][x][c] = sum_taps P[(y, x) + tap][tap * 3 + c] is then 27 LDS reads and adds per output pixel;
//   * a tile = 8 x 32 output pixels needs P on its 10 x 34 halo region (conv_3 recomputed on the halo: x 1.33 of a K = 32 product).
// Per 32 pixels: 8 + 8 MFMAs, ~300 VALU instructions; HBM: the 32-channel input, the 128-channel quarter-resolution skip source, 12 B
// per pixel out — ~3.4 GB per population instead of 13.
#include "common.h"
#include "kernels.h"

namespace {
constexpr int BT_TH = 8, BT_TW = 32, BT_PH = BT_TH + 2, BT_PW = BT_TW + 2, BT_NPX = BT_PH * BT_PW, BT_NBLK = (BT_NPX + 31) / 32;   // 340 px, 11 blocks
}

__global__ __launch_bounds__(256, 2) void bg_tail_kernel(BgTailParams p, int tiles_x, int tiles_y, int n_tiles) {
    __shared__ __attribute__((aligned(16))) float Ps[BT_NBLK * 32][32];   // P[pixel][27 used of 32]; 16-byte chunk index ^ (pixel & 7)
    __shared__ __attribute__((aligned(16))) half_t Ca[128], Cc[128];      // final bn scale; conv_3 bias * scale + shift (fp16: the epilogue is packed)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 31, kh = lane >> 5;
    const int R = p.R, R2 = R >> 1;
    // weight fragments (MFMA A operands), resident for the whole launch
    // (conv_3's rows carry the final bn's scale: bn(conv + bias + skip) = (scale * W) x + skip * scale + (bias * scale + shift))
    h8 w3f[4][2], wtf[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const half_t sc = (half_t)p.tab[p.bnf_off + j * 32 + lr];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) w3f[j][kk] = *(const h8*)(p.w3 + (j * 32 + lr) * 32 + kk * 16 + kh * 8) * sc;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {          // row n = tap * 3 + colour; K slots in accumulator-lane channel order
        const int j = s >> 1, gp = s & 1;
        h4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
        if (lr < 27) {
            const int tap = lr / 3, col = lr - tap * 3;
            const half_t* wp = p.rgb_w + ((long long)tap * p.cpad + col) * 128 + j * 32 + 16 * gp + 4 * kh;
            lo = *(const h4*)wp;
            hi = *(const h4*)(wp + 8);
        }
        wtf[s] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    if (t < 128) {                         // (the generator's last bn is unconditional: every sample's table row holds the same values)
        const float a = p.tab[p.bnf_off + t], s = p.tab[p.ctot + p.bnf_off + t];
        Ca[t] = (half_t)a;
        Cc[t] = (half_t)(p.b3[t] * a + s);
    }
    const float rb0 = p.rgb_b[0], rb1 = p.rgb_b[1], rb2 = p.rgb_b[2];
    __syncthreads();
    const int tpi = tiles_x * tiles_y;
    h8 nx0, nx1;
    h4 nsk[4];
    struct { const half_t* sp; bool inb; } nb = {p.x0, false};
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / tpi, trem = tile - b * tpi;
        const int ty0 = (trem / tiles_x) * BT_TH, tx0 = (trem % tiles_x) * BT_TW;
        // operands of a 32-pixel block are requested one block ahead (the wave's first block of the NEXT tile included): as plain
        // load -> use the kernel was a chain of exposed HBM round trips, three per wave and tile
        auto aim = [&](int tl, int blk, h8& xa, h8& xb, h4 (&sq)[4]) {
            const int bb = tl / tpi, tr = tl - bb * tpi;
            const int y0 = (tr / tiles_x) * BT_TH, x0 = (tr % tiles_x) * BT_TW;
            const int pl = blk * 32 + lr;
            const int pr = pl / BT_PW, pc = pl - pr * BT_PW;
            const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
            const int cy = min(max(gy, 0), R - 1), cx = min(max(gx, 0), R - 1);
            const half_t* hp = p.h + (((long long)bb * R + cy) * R + cx) * 32 + kh * 8;
            xa = *(const h8*)hp;
            xb = *(const h8*)(hp + 16);
            decltype(nb) r;
            r.sp = p.x0 + (((long long)bb * R2 + (cy >> 1)) * R2 + (cx >> 1)) * 128 + 4 * kh;
            r.inb = pl < BT_NPX && gy >= 0 && gy < R && gx >= 0 && gx < R;
#pragma unroll
            for (int g = 0; g < 4; ++g) sq[g] = *(const h4*)(r.sp + 8 * g);
            return r;
        };
        if (tile == (int)blockIdx.x) nb = aim(tile, wave, nx0, nx1, nsk);       // (first tile of this workgroup: nothing was in flight)
#pragma unroll 1
        for (int blk = wave; blk < BT_NBLK; blk += 4) {
            const int pl = blk * 32 + lr;
            const h8 x0f = nx0, x1f = nx1;
            const half_t* sp = nb.sp;
            const bool inb = nb.inb;
            h4 skc[4], skn[4];              // the skip's quads of this pixel, one 32-channel slice ahead of their use
#pragma unroll
            for (int g = 0; g < 4; ++g) skc[g] = nsk[g];
            if (blk + 4 < BT_NBLK) nb = aim(tile, blk + 4, nx0, nx1, nsk);
            else if (tile + (int)gridDim.x < n_tiles) nb = aim(tile + gridDim.x, wave, nx0, nx1, nsk);
            f16x pacc;
#pragma unroll
            for (int q = 0; q < 16; ++q) pacc[q] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {          // one 32-channel slice at a time: conv_3 (2 MFMAs) -> bn / relu -> its two K steps of P
                if (j < 3) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) skn[g] = *(const h4*)(sp + (j + 1) * 32 + 8 * g);
                }
                f16x acc;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = 0.f;
                acc = mfma32(w3f[j][0], x0f, acc);
                acc = mfma32(w3f[j][1], x1f, acc);
                h4 zq[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {              // packed fp16 from here on (the unfused path rounded this map to fp16 and applied bn there too)
                    const int c0 = j * 32 + 8 * g + 4 * kh;
                    const h4 a = {(half_t)acc[g * 4], (half_t)acc[g * 4 + 1], (half_t)acc[g * 4 + 2], (half_t)acc[g * 4 + 3]};
                    const h4 v = a + (skc[g] * *(const h4*)(Ca + c0) + *(const h4*)(Cc + c0));
                    zq[g] = inb ? __builtin_elementwise_max(v, h4{0, 0, 0, 0}) : h4{0, 0, 0, 0};   // relu; the conv's zero padding outside the image
                }
#pragma unroll
                for (int gp = 0; gp < 2; ++gp)
                    pacc = mfma32(wtf[j * 2 + gp], __builtin_shufflevector(zq[2 * gp], zq[2 * gp + 1], 0, 1, 2, 3, 4, 5, 6, 7), pacc);
                if (j < 3) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) skc[g] = skn[g];
                }
                if (j & 1) __builtin_amdgcn_sched_barrier(0);      // two slices' chains may interleave (four spilled)
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f4 o = {pacc[g * 4], pacc[g * 4 + 1], pacc[g * 4 + 2], pacc[g * 4 + 3]};
                *(f4*)(&Ps[pl][(((2 * g + kh) ^ (pl & 7)) << 2)]) = o;
            }
        }
        __syncthreads();
        {
            const int oy = t >> 5, ox = t & 31;
            float s0 = rb0, s1 = rb1, s2 = rb2;
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) {
                    const int pp = (oy + ty) * BT_PW + ox + tx, n = (ty * 3 + tx) * 3;
                    const float* row = &Ps[pp][0];
                    const int key = pp & 7;
                    s0 += row[((((n) >> 2) ^ key) << 2) + ((n) & 3)];
                    s1 += row[((((n + 1) >> 2) ^ key) << 2) + ((n + 1) & 3)];
                    s2 += row[((((n + 2) >> 2) ^ key) << 2) + ((n + 2) & 3)];
                }
            const long long hw = (long long)R * R;
            float* yo = p.y + (long long)b * 3 * hw + (long long)(ty0 + oy) * R + tx0 + ox;
            yo[0] = tanhf(s0);
            yo[hw] = tanhf(s1);
            yo[2 * hw] = tanhf(s2);
        }
        __syncthreads();
    }
}

bool bg_tail_supported(int R, int mid, int cout, int cin, int up, int cpad) {
    return R % 32 == 0 && R >= 32 && mid == 32 && cout == 128 && cin == 128 && up == 1 && cpad >= 3;
}
bool launch_bg_tail(const BgTailParams& p, hipStream_t st) {
    const int tiles_x = p.R / BT_TW, tiles_y = p.R / BT_TH;
    const int n_tiles = p.B * tiles_x * tiles_y;
    const int grid = std::min(n_tiles, 2 * glass_cu_count());
    hipLaunchKernelGGL(bg_tail_kernel, dim3(grid), dim3(256), 0, st, p, tiles_x, tiles_y, n_tiles);
    return true;
}
