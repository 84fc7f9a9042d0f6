// conv_down.hip — the whole second half of a discriminator block in ONE kernel (gfx950):
//     h -> FIR 4x4 (pad 2) -> conv3x3 stride 2 + bias + lrelu*sqrt2 \
//     x -> FIR 4x4 (pad 1) -> ::2 -> conv1x1 (no bias / activation)  +-> (a + b) / sqrt2
// (stylegan2/modules.py:1204-1254 ConvDownLayer, 1587-1601 DiscriminatorConvBlock.forward; `h` is the block's first conv
// output, `x` the block input).  As separate passes this is five launches — blur, blur-down, 1x1 skip conv, stride-2
// conv, each re-reading or re-writing a full-resolution map: 30 GB of HBM traffic for the 1024^2 block of a 64-candidate
// population where the maps themselves are 8.6 GB in and 2.1 GB out.  Here both FIRs are applied while the conv's input
// patch is staged, and the skip branch rides in the same tile as a second accumulator pass.
//
// Structure (HBM-bound layers: what matters is bytes in flight, see DESIGN.md "keep loads in flight"):
//   * persistent workgroups, ONE per CU (256 threads, up to 512 VGPRs each: the register file is where the in-flight bytes
//     live), each walking a contiguous range of (4 x 32 output pixel tile, 64-channel n tile) items; an item is a stream
//     of STAGES: skip(c) for every 32-channel chunk c, then main(c);
//   * a stage's raw input window (main: 12 x 68 px of h, skip: 10 x 66 px of x, 32 channels) is fetched into REGISTERS
//     FOUR stages ahead (four named register sets, 13 16-byte loads per thread each, unconditional + clamped; the zero
//     padding is a mask applied when the data is consumed) — ~200 KB of loads in flight per CU;
//   * thread (column, 8-channel group) owns a window column: the VERTICAL FIR runs on its registers (packed fp16) and the
//     result goes to LDS; waves then own whole rows for the HORIZONTAL FIR, which rewrites each row IN PLACE as the MFMA
//     operand image (even | odd columns de-interleaved for the stride-2 fragment walk) — LDS ops are in order per wave,
//     so no workgroup barrier sits between a row's reads and its writes;
//   * the 4 window columns a 64-column thread grid does not cover travel as a 13th load per thread, are parked raw in
//     LDS, and are blurred on the fly by the four lanes per row that need them;
//   * LDS images are dense 64-byte rows: the vertical-pass image rotates each pixel inside its aligned group of four
//     (stride-4 sliding-window reads and stride-1 writes both conflict-free), the operand image XOR-swizzles the
//     16-byte chunk (conv_glds.hip's layout);
//   * WRES (Cout = 64, Cin <= 64 — the 1024^2 block): all weights (9 x 64 x Cin main + 64 x Cin skip) stay in LDS for the
//     workgroup's lifetime: steady state issues window loads and output stores only.  Otherwise the stage's weight slice
//     is re-staged through registers each stage (its loads are issued before the window refill, but consuming them
//     still retires every older window load — vmcnt is in order).
//   * epilogue: bias, lrelu*sqrt2, + skip (rounded to fp16 as the separate pass stored it), / sqrt2, LDS-transposed
//     16-byte row-order stores.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

namespace {
constexpr int TH = 4, NT = 64, VP = 65;      // VP: column slots per image row (64 window columns / 65 blurred columns)
constexpr int V_BYTES = 9 * VP * 64;          // vertical-pass image / operand image (in place): 39168
constexpr int W_BYTES = 9 * NT * 64;          // 36864
constexpr int EM_BYTES = 12 * 4 * 64;         // main edge columns, raw: 3072
constexpr int OFF_EX = 5 * VP * 64;           // skip edge columns, raw (1280 B): rows 5.. of the image are free in a skip stage
constexpr int OFF_W = V_BYTES, OFF_EM = OFF_W + W_BYTES, OFF_C = OFF_EM + EM_BYTES;
constexpr int OFF_WS = OFF_C + NT * 4;                                    // skip weights of chunk 0 (WRES)
constexpr int LDS_BYTES_STREAM = OFF_WS, LDS_BYTES_RES = OFF_WS + NT * 64;   // 77632 / 81728: two workgroups per CU
constexpr int OROW = NT * 2 + 16;

// Both patch images put image row r at byte r * VP * 64 and swizzle by the COLUMN only, so a row step is an immediate offset.
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
// The compiler hoists every per-thread address term out of the persistent loop and, with the window sets owning the
// register file, spills them — and a scratch reload is a VMEM op whose wait retires every older window load.  Each phase
// therefore re-derives its lane geometry from an opaque copy of the thread id: nothing loop-invariant to hoist.
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
struct RSet { h8 a[13]; };
struct Stage { int kind, c, b, ty0, tx0, n0, first, last_skip, last_main; };
}  // namespace

struct DownParams {
    const half_t* h;    // [B][R][R][Cin]  first conv's output
    const half_t* x;    // [B][R][R][Cin]  block input
    const half_t* w1;   // [9][Cout][Cin]
    const half_t* ws;   // [Cout][Cin]
    const float* b1;    // [Cout]
    half_t* y;          // [B][R/2][R/2][Cout]
    int B, R, Cin, Cout;
};

template <bool WRES>
__global__ __launch_bounds__(256, 2) void conv_down_kernel(DownParams p, int tiles_x, int tiles_y, int NTn, int n_items, int per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Vs = smem;
    char* Ws = smem + OFF_W;
    float* Cb = (float*)(smem + OFF_C);
    const int t = threadIdx.x;
    const int nc = p.Cin >> 5, spi = 2 * nc;      // 32-channel chunks; stages per item
    const int tpi = tiles_x * tiles_y;
    const int R = p.R, Ro = R >> 1;
    const int first = blockIdx.x * per_block;
    const int last = min(first + per_block, n_items);
    if (first >= last) return;
    const int n_st = (last - first) * spi;
    const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};

    auto decode = [&](int si) {
        Stage s;
        const int it = first + si / spi, r = si % spi;
        s.kind = r < nc ? 0 : 1;
        s.c = s.kind ? r - nc : r;
        s.first = r == 0;
        s.last_skip = r == nc - 1;
        s.last_main = r == spi - 1;
        const int pt = it / NTn;
        s.n0 = (it - pt * NTn) * NT;
        s.b = pt / tpi;
        const int trem = pt - s.b * tpi;
        s.ty0 = (trem / tiles_x) * TH;
        s.tx0 = (trem % tiles_x) * 32;
        // everything here is workgroup-uniform: pin it to SGPRs (the integer divisions run on the vector ALU)
        s.kind = __builtin_amdgcn_readfirstlane(s.kind); s.c = __builtin_amdgcn_readfirstlane(s.c);
        s.first = __builtin_amdgcn_readfirstlane(s.first); s.last_skip = __builtin_amdgcn_readfirstlane(s.last_skip);
        s.last_main = __builtin_amdgcn_readfirstlane(s.last_main); s.n0 = __builtin_amdgcn_readfirstlane(s.n0);
        s.b = __builtin_amdgcn_readfirstlane(s.b); s.ty0 = __builtin_amdgcn_readfirstlane(s.ty0);
        s.tx0 = __builtin_amdgcn_readfirstlane(s.tx0);
        return s;
    };
    // window loads of stage si into R: 12 rows of this thread's column + one vector of the edge columns.  Always 13 loads,
    // all unconditional (clamped coordinates; stages past the end re-read stage 0): the padding mask is applied at use.
    auto issue = [&](int si, RSet& Rg) {
        const int t = opaque(threadIdx.x), cg = t & 3, cs = t >> 2;
        const Stage s = decode(si < n_st ? si : 0);
        const half_t* img = (s.kind ? p.h : p.x) + (long long)s.b * R * R * p.Cin + s.c * 32;    // uniform (SGPRs)
        const int oy = 2 * s.ty0 - (s.kind ? 2 : 1), ox = 2 * s.tx0 - (s.kind ? 2 : 1);
        const int nrow = s.kind ? 12 : 10;
        const int xo = min(max(ox + cs, 0), R - 1) * p.Cin + cg * 8;       // R * R * Cin < 2^31: 32-bit element offsets
        const int rs = R * p.Cin;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int iy = min(max(oy + min(k, nrow - 1), 0), R - 1);      // uniform
            Rg.a[k] = *(const h8*)(img + iy * rs + xo);
        }
        const int er = s.kind ? min(t >> 4, 11) : min(t >> 3, 9);
        const int ec = 64 + (s.kind ? ((t >> 2) & 3) : ((t >> 2) & 1));
        const int ey = min(max(oy + er, 0), R - 1), ex = min(max(ox + ec, 0), R - 1);
        Rg.a[12] = *(const h8*)(img + ey * rs + ex * p.Cin + cg * 8);
    };

    // ---- resident weights (WRES, one chunk): main taps and skip rows in LDS, bias in LDS ---------------------------------
    if (WRES) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int v = k * 256 + t, row = v >> 2;               // row = tap * 64 + n
            const int lc = (v & 3) ^ ((row >> 2) & 3);
            *(h8*)(Ws + v * 16) = *(const h8*)(p.w1 + ((long long)(row >> 6) * p.Cout + (row & 63)) * p.Cin + lc * 8);
        }
        const int row = t >> 2, lc = (t & 3) ^ ((row >> 2) & 3);
        *(h8*)(smem + OFF_WS + t * 16) = *(const h8*)(p.ws + (long long)row * p.Cin + lc * 8);
        if (t < NT) Cb[t] = p.b1[t];
    }

    f16x acc[2];       // ONE accumulator set: the skip stages of an item run first and retire into `sk` (fp16)
    h4 sk[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

    auto step = [&](int si, RSet& Rg) {
        const Stage s = decode(si);
        const int oy = 2 * s.ty0 - (s.kind ? 2 : 1), ox = 2 * s.tx0 - (s.kind ? 2 : 1);
        __syncthreads();       // B0: every wave is done with the operand image / weights / epilogue image of the previous stage
        // ---- vertical FIR on this thread's window column -> LDS; edge vector parked raw -----------------------------
        {
        const int t = opaque(threadIdx.x), cg = t & 3, cs = t >> 2;
        const bool colok = (unsigned)(ox + cs) < (unsigned)R;
        if (s.kind) {
#pragma unroll
            for (int k = 0; k < 12; ++k) Rg.a[k] = (colok && (unsigned)(oy + k) < (unsigned)R) ? Rg.a[k] : zero;
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                *(h8*)(Vs + vaddr(r, cs, cg)) = fir4(Rg.a[r], Rg.a[r + 1], Rg.a[r + 2], Rg.a[r + 3]);
                if (r % 3 == 2) __builtin_amdgcn_sched_barrier(0);
            }
            if (t < 192) {
                const int er = t >> 4, ecl = (t >> 2) & 3;
                const bool ok = (unsigned)(oy + er) < (unsigned)R && (unsigned)(ox + 64 + ecl) < (unsigned)R;
                *(h8*)(smem + OFF_EM + ((er * 4 + ecl) * 4 + cg) * 16) = ok ? Rg.a[12] : zero;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 10; ++k) Rg.a[k] = (colok && (unsigned)(oy + k) < (unsigned)R) ? Rg.a[k] : zero;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(h8*)(Vs + vaddr(q, cs, cg)) = fir4(Rg.a[2 * q], Rg.a[2 * q + 1], Rg.a[2 * q + 2], Rg.a[2 * q + 3]);
            if (t < 80) {
                const int er = t >> 3, ecl = (t >> 2) & 1;
                const bool ok = (unsigned)(oy + er) < (unsigned)R && (unsigned)(ox + 64 + ecl) < (unsigned)R;
                *(h8*)(smem + OFF_EX + ((er * 2 + ecl) * 4 + cg) * 16) = ok ? Rg.a[12] : zero;
            }
        }
        }
        // ---- this stage's weight slice (not resident): loads issued BEFORE the window refill so that consuming them does
        // not wait for the refill (vmcnt retires in order) ------------------------------------------------------------------
        h8 wr[9];
        if (!WRES) {
            const int t = opaque(threadIdx.x);
            if (s.kind) {
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int v = k * 256 + t, row = v >> 2;
                    const int lc = (v & 3) ^ ((row >> 2) & 3);
                    wr[k] = *(const h8*)(p.w1 + ((long long)(row >> 6) * p.Cout + s.n0 + (row & 63)) * p.Cin + s.c * 32 + lc * 8);
                }
            } else {
                const int row = t >> 2, lc = (t & 3) ^ ((row >> 2) & 3);
                wr[0] = *(const h8*)(p.ws + (long long)(s.n0 + row) * p.Cin + s.c * 32 + lc * 8);
                if (s.first && t < NT) Cb[t] = p.b1[s.n0 + t];
            }
        }
        issue(si + 2, Rg);     // refill: two stages of window loads stay in flight
        __syncthreads();       // B1: vertical-pass image complete
        // ---- horizontal FIR: each wave owns whole rows and rewrites them in place as the operand image ------------------
        {
            const int t = opaque(threadIdx.x), lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
            const int j = lane >> 2, cgl = lane & 3;
            if (s.kind) {
#pragma nounroll
                for (int ri = 0; ri < 3; ++ri) {         // rows one at a time: three rows in flight would cost 150 VGPRs
                    const int rr = wave + 4 * ri;
                    if (rr < 9) {
                        h8 v[8];
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = *(const h8*)(Vs + vaddr(rr, 4 * j + k, cgl));
                        v[7] = zero;
                        if (j < 15) {
#pragma unroll
                            for (int k = 4; k < 7; ++k) v[k] = *(const h8*)(Vs + vaddr(rr, 4 * j + k, cgl));
                        } else {      // window columns 64..67: vertical FIR of the raw edge vectors, on the fly
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const char* e = smem + OFF_EM + ((rr * 4 + k) * 4 + cgl) * 16;
                                v[4 + k] = fir4(*(const h8*)e, *(const h8*)(e + 256), *(const h8*)(e + 512), *(const h8*)(e + 768));
                            }
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
                    }
                }
            } else {
                const int q = wave;
                h8 v[6];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = *(const h8*)(Vs + vaddr(q, 4 * j + k, cgl));
                if (j < 15) {
#pragma unroll
                    for (int k = 4; k < 6; ++k) v[k] = *(const h8*)(Vs + vaddr(q, 4 * j + k, cgl));
                } else {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const char* e = smem + OFF_EX + ((2 * q * 2 + k) * 4 + cgl) * 16;
                        v[4 + k] = fir4(*(const h8*)e, *(const h8*)(e + 128), *(const h8*)(e + 256), *(const h8*)(e + 384));
                    }
                }
                const h8 o0 = fir4(v[0], v[1], v[2], v[3]), o1 = fir4(v[2], v[3], v[4], v[5]);
                __builtin_amdgcn_wave_barrier();
                *(h8*)(Vs + aaddr(q, 2 * j, cgl)) = o0;
                *(h8*)(Vs + aaddr(q, 2 * j + 1, cgl)) = o1;
            }
        }
        if (!WRES) {
            const int t = opaque(threadIdx.x);
            if (s.kind) {
#pragma unroll
                for (int k = 0; k < 9; ++k) *(h8*)(Ws + (k * 256 + t) * 16) = wr[k];
            } else {
                *(h8*)(Ws + t * 16) = wr[0];
            }
        }
        __syncthreads();       // B2: operand image (+ weights) complete
        // ---- MFMA ---------------------------------------------------------------------------------------------------------
        const char* Wm = Ws;                                      // this stage's main / skip weight image
        const char* Wk = WRES ? smem + OFF_WS : Ws;
        const int tm = opaque(threadIdx.x), lr = tm & 31, kh = (tm >> 5) & 1, wave = __builtin_amdgcn_readfirstlane(tm >> 6);
        if (s.kind) {
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
                            const h8 wf = *(const h8*)(Wm + waddr((ky * 3 + kx) * NT + j * 32 + lr, lc));
                            acc[j] = mfma32(wf, xf, acc[j]);
                        }
                        if (kk) __builtin_amdgcn_sched_barrier(0);   // one tap's fragments live at a time (the window sets own the registers)
                    }
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int lc = kk * 2 + kh;
                const h8 xf = *(const h8*)(Vs + aaddr(wave, lr, lc));
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const h8 wf = *(const h8*)(Wk + waddr(j * 32 + lr, lc));
                    acc[j] = mfma32(wf, xf, acc[j]);
                }
            }
            if (s.last_skip) {      // the skip branch's value as the separate pass stored it: fp16
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            sk[j][g][q] = (half_t)acc[j][g * 4 + q];
                            acc[j][g * 4 + q] = 0.f;
                        }
            }
        }
        // ---- item epilogue: lane = pixel lr of output row ty0 + wave -------------------------------------------------------
        if (s.kind && s.last_main) {
            __syncthreads();   // every wave is done reading the operand image: the per-wave output image overlays it
            const int lane = tm & 63;
            char* Os = smem + wave * (32 * OROW);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = j * 32 + 8 * g + 4 * kh;
                    const f4 bb = *(const f4*)(Cb + nl);
                    h4 out;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v = lrelu_sqrt2(acc[j][g * 4 + q] + bb[q]) + (float)sk[j][g][q];
                        out[q] = (half_t)(v * 0.70710678118654752440f);
                        acc[j][g * 4 + q] = 0.f;
                    }
                    *(h4*)(Os + lr * OROW + nl * 2) = out;
                }
            __builtin_amdgcn_wave_barrier();
            half_t* yrow = p.y + (((long long)s.b * Ro + s.ty0 + wave) * Ro + s.tx0) * p.Cout + s.n0;
#pragma unroll
            for (int k = 0; k < NT / 16; ++k) {
                const int v = lane + 64 * k;
                const int pix = v >> 3, chv = v & 7;
                *(h8*)(yrow + (long long)pix * p.Cout + chv * 8) = *(const h8*)(Os + pix * OROW + chv * 16);
            }
        }
    };

    RSet r0, r1;
    issue(0, r0);
    issue(1, r1);
    // n_st is even (an item is 2 * nc stages): no exit between the two steps — with a mid-loop exit hipcc's wait-count
    // merge at the loop header stops counting the other set's refill as younger and every window wait drains the queue
    for (int si = 0; si < n_st; si += 2) {
        step(si, r0);
        step(si + 1, r1);
    }
}

// Returns the kernel symbol, or nullptr when the block does not qualify (caller runs the separate passes).
const char* launch_conv_down(const half_t* h, const half_t* x, const half_t* w1, const half_t* ws, const float* b1, half_t* y,
                             int B, int R, int Cin, int Cout, hipStream_t st) {
    static const bool off = getenv("GLASS_NO_DOWN") != nullptr;   // A/B knob
    if (off || R % 64 != 0 || Cin % 32 != 0 || Cout % NT != 0 || Cout > 2 * NT || R < 64) return nullptr;
    if ((long long)R * R * Cin >= (1LL << 31)) return nullptr;
    DownParams p;
    p.h = h; p.x = x; p.w1 = w1; p.ws = ws; p.b1 = b1; p.y = y; p.B = B; p.R = R; p.Cin = Cin; p.Cout = Cout;
    const int Ro = R / 2, tiles_x = Ro / 32, tiles_y = Ro / TH;
    const int NTn = Cout / NT;
    const long long items = (long long)B * tiles_x * tiles_y * NTn;
    if (items >= (1LL << 30)) return nullptr;
    static int slots = 0;
    if (!slots) {
        (void)hipFuncSetAttribute((const void*)conv_down_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_RES);
        (void)hipFuncSetAttribute((const void*)conv_down_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_STREAM);
        hipDeviceProp_t prop;
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipGetDeviceProperties(&prop, dev);
        slots = prop.multiProcessorCount * 2;
    }
    int per_block = (int)((items + slots - 1) / slots);
    per_block = (per_block + NTn - 1) / NTn * NTn;       // the n tiles of one pixel tile stay in one workgroup (second one hits L2)
    const int grid = (int)((items + per_block - 1) / per_block);
    const bool wres = Cin == 32 && NTn == 1;
    if (wres) {
        hipLaunchKernelGGL(conv_down_kernel<true>, dim3(grid), dim3(256), LDS_BYTES_RES, st, p, tiles_x, tiles_y, NTn, (int)items, per_block);
        return "conv_down_kernel<wres>";
    }
    hipLaunchKernelGGL(conv_down_kernel<false>, dim3(grid), dim3(256), LDS_BYTES_STREAM, st, p, tiles_x, tiles_y, NTn, (int)items, per_block);
    return "conv_down_kernel";
}
