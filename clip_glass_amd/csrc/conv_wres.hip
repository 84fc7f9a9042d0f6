// conv_wres.hip — 3x3 stride-1 convolution 64 -> 64 channels at 512^2 (G's last-but-one conv with its toRGB, D's second block's first conv
// with the blur-down by-product; stylegan2/modules.py:920-967, models.py:852-870, modules.py:1238-1254) with the WHOLE weight tensor
// resident in LDS and the input patches on a four-deep LDS-DMA ring — the ping-pong schedule of conv_glds.hip's persistent kernel.
//
// Why its own kernel (round 6).  Rounds 1-5 ran these two layers on conv_tiled<3,1,8,64>: register-staged, three 256-thread workgroups per
// CU, 1.84 + 1.56 ms where the HBM roof (2.1 GB in + 2.1 GB out per layer at 6.3 TB/s) is 0.7 ms.  A tile of that kernel moves 43 KB of
// patch AND the layer's whole 74 KB of weights through registers into LDS; loads, MFMAs (40 % busy) and stores run back to back per
// tile.  Here the 9 x 64 x 64 weights (73 728 B) are loaded ONCE per workgroup (per candidate for pre-modulated per-sample weights: a
// workgroup walks a contiguous range of tiles), a tile is 16 rows x 32 px x 64 channels on eight waves (wave = 2 rows x 2 n blocks).
//
// Second form (this one).  The first form kept the patch as two 32-channel chunks = two 40 KB buffers: one chunk in flight while the other
// was read, requested two phases (~4000 clocks) ahead of its wait.  Ablations (developer build, GLASS_WRES_ABLATE; DESIGN "Round 6"): without
// MFMAs, fragment reads and epilogue the kernel still took 1.1 of its 1.7 ms — one 40 KB request per CU and a wait per half tile is a
// latency chain (10.5 MB per CU at 4.8 B/clock); the epilogue was another 22 %.  Now
//   * the patch is FOUR 16-channel chunks (18 x 34 px x 32 B = 19 584 B each) = four buffers: chunk c of tile t + 1 is requested in phase
//     c + 1 of tile t, right behind the last read of chunk c of tile t — three chunks (59 KB) are in flight all the time and every chunk has
//     three phases + the epilogue (> 7000 clocks) to land;
//   * a phase is one chunk: 36 MFMAs per wave on 12 patch fragments (4 rows x 3 columns, each feeding up to three tap rows) + 18 weight
//     fragments — 0.83 fragment reads per MFMA where the tap-row phases of the first form read 1.0 (the kernel sits at the LDS read rate);
//   * the epilogue needs no LDS: a lane pair (px, kh = 0 / 1) exchanges quads with v_permlane32_swap and each lane stores 16 contiguous
//     bytes; the tile's addresses are scalar bases + lane constants (the first form spent ~250 of its ~1000 epilogue instructions on
//     64-bit address arithmetic) — so no patch buffer is borrowed for the output and the ring never stops;
//   * counted waits: every wave issues exactly three patch pieces per phase (2 x 512 vectors + a 25-lane tail per wave), so "chunk c has
//     landed" is vmcnt(<= the number of pieces requested since) — stores and operand pieces in between only make the wait earlier.
// Ping-pong: waves 4-7 run one barrier behind waves 0-3 (one wave of each group per SIMD):
//     load interval: 12 ds_read_b128, counted vmcnt, lgkmcnt(0) | barrier | MFMA interval: 36 MFMAs at s_setprio 1, the next tap row's six
//     weight fragments behind each tap row, the ring's three pieces behind the first four MFMAs | barrier
// Input layout: pixel-major [B][H][W][64], or chunk-planar [B][4][H][W][16] (common.h x_planar8) written by upfir2<false> / dblock0 for
// this kernel — pixel-major, the four 32-byte quarters of every 128-byte line cross the fabric at four different times.
// toRGB (TRGB) and the blur-down by-product (XS) as in conv_glds.hip.  K order per accumulator: chunk, tap row, tap column.
#include "common.h"
#include "kernels.h"
#include <stdio.h>
#include <stdlib.h>

namespace {
constexpr int NT = 64, NTHR = 512, TW = 32, TH = 16, RW = 2;
constexpr int PH = TH + 2, PW = TW + 2;
constexpr int NCH = 4;                                   // 16-channel chunks = ring buffers
constexpr int NVA = PH * PW * 2;                         // 16-byte vectors of a patch chunk (1224): 2 x 512 + 8 waves x 25
constexpr int NTAIL = (NVA - 2 * NTHR) / 8;              // 25 lanes of every wave carry the third piece
constexpr int A_BYTES = (NVA * 16 + 511) / 512 * 512;    // 19968
constexpr int NVW = NCH * 9 * NT * 2;                    // 4608 vectors of the weight image [chunk][tap][n][16 ch]
constexpr int NWP = NVW / NTHR;                          // 9 pieces per thread
constexpr int OFF_W = NCH * A_BYTES;                     // 79872
constexpr int OFF_C = OFF_W + NVW * 16;                  // 153600: [dscale | bias | shift][NT] fp32
constexpr int OFF_T = OFF_C + 3 * NT * 4;                // TRGB: toRGB weight rows [hi r,g,b | lo r,g,b][NT] fp16
constexpr int OFF_N = OFF_T + 6 * NT * 2;                // the tile's noise values [8 waves][2 rows][32 px] fp32
constexpr int OFF_Y = OFF_N + 8 * 64 * 4;                // TRGB: the previous skip image's taps of the tile [8 waves][3 ch][2 rows][17 cols] fp32 (128 slots per wave)
constexpr int LDS_BYTES = OFF_Y + 8 * 128 * 4;           // 161280
static_assert(LDS_BYTES <= 163840 && NVW % NTHR == 0 && 2 * NTHR + 8 * NTAIL == NVA && NTAIL <= 64, "one workgroup per CU; three pieces per wave and chunk");

__device__ __attribute__((aligned(256))) float g_wres_zero_page[64];   // zero-initialised: source of the zero padding and of absent operands
__device__ __attribute__((aligned(256))) float g_wres_ones_page[64] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
__device__ __forceinline__ void dma16(const void* src, char* lds_wave_base) {      // LDS destination = wave-uniform base + lane * 16
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const float* src, char* lds_wave_base) {      // LDS destination = wave-uniform base + lane * 4
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_wave_base, 4, 0, 0);
}
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// v_permlane32_swap, register by register: the upper 32 lanes of `a` trade places with the lower 32 lanes of `b`
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4w __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void swap32(h4& a, h4& b) {
    u2v x = __builtin_bit_cast(u2v, a), y = __builtin_bit_cast(u2v, b);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const auto r = __builtin_amdgcn_permlane32_swap(x[e], y[e], false, false);
        x[e] = r[0];
        y[e] = r[1];
    }
    a = __builtin_bit_cast(h4, x);
    b = __builtin_bit_cast(h4, y);
}
__device__ __forceinline__ void swap32(h8& a, h8& b) {
    u4w x = __builtin_bit_cast(u4w, a), y = __builtin_bit_cast(u4w, b);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const auto r = __builtin_amdgcn_permlane32_swap(x[e], y[e], false, false);
        x[e] = r[0];
        y[e] = r[1];
    }
    a = __builtin_bit_cast(h8, x);
    b = __builtin_bit_cast(h8, y);
}
#define WR_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// Fragment reads of the K loop as inline assembly: hipcc puts lgkmcnt(0) in front of the first MFMA that uses ANY outstanding ds_read, i.e. it
// also waits for the reads requested a moment ago for the NEXT tap row (stamps: 2400-3800 clocks per interval of 36 MFMAs).  A read it cannot see
// needs no wait in its eyes; the waits are the explicit ones below, tied to the registers they release by "+v" operands.
#define WR_LDS_RD(dst, vaddr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(vaddr), "n"(imm))
#define WR_WAIT_LGKM6(a, b, c, d, e, f) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f)::"memory")
#define WR_WAIT_LGKM4(a, b, c, d) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory")
// developer build only (make AB=1, GLASS_WRES_ABLATE): timing experiments that switch parts of a tile off — 1 the whole epilogue, 2 its global
// stores, 4 the K loop's MFMAs, 8 the patch fragment reads of the load interval, 16 the weight fragment reads, 32 the ring's requests (WRONG RESULTS)
#ifdef GLASS_AB_KNOBS
#define WR_ABL_ARG , int abl
#define WR_ABL(bit) (abl & (bit))
// bit 64: shader-clock stamps of the sixth tile of one workgroup in the middle of the grid, printed by the launcher (which synchronises)
__device__ unsigned long long* g_wres_trace = nullptr;
#define WRT(ph) \
    if (WR_ABL(64) && blockIdx.x == 77 && (threadIdx.x & 63) == 0 && id - first == 5) g_wres_trace[(ph) * 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime()
#else
#define WR_ABL_ARG
#define WR_ABL(bit) false
#define WRT(ph)
#endif
}  // namespace

template <bool TRGB, bool XS>
__global__ __launch_bounds__(512, 1) void conv_wres_kernel(ConvParams p, int tiles_x, int tiles_y, int PT, int per_wg WR_ABL_ARG) {
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
    // vector v of a patch chunk sits at byte v * 16: pixel v >> 1, physical half v & 1 holding the pixel's LOGICAL 8-channel half
    // (v & 1) ^ ((pixel >> 3) & 1) (swizzle on the source side — LDS-DMA writes lane-linear: any 16 consecutive pixels of a fragment read then
    // cover the 64 banks once).  Thread t carries vectors t, 512 + t and (lanes < 25) 1024 + 25 wave + lane.
    const int pixs = p.Cin;                    // elements between pixels / between the 16-channel chunks of a pixel (pixel-major input only)
    const int c_step = 16;
    int a_geo[3];             // lane constants: patch row | patch column << 8 | logical half << 16 | vector exists << 17
    {
        const int t = opaque(threadIdx.x), lane = t & 63;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int v = k < 2 ? k * NTHR + t : 2 * NTHR + wave * NTAIL + lane;
            const int pix = v >> 1, pr = pix / PW, pc = pix - pr * PW;
            a_geo[k] = opaque(pr | (pc << 8) | ((((v & 1) ^ ((pix >> 3) & 1))) << 16) | ((k < 2 || lane < NTAIL) ? 1 << 17 : 0));
        }
    }
    const half_t* xb = p.x;
    int a_src[3];             // element offset into the image (< 2^31: launcher), or -1 = zero page
    auto aim_a = [&](const Item& w) {
        xb = p.x + (long long)w.b * p.x_bstride;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int iy = w.ty0 - 1 + (a_geo[k] & 255), ix = w.tx0 - 1 + ((a_geo[k] >> 8) & 255);
            const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            a_src[k] = ok ? (iy * p.W + ix) * pixs + ((a_geo[k] >> 16) & 1) * 8 : -1;
        }
    };
    // (the page's address is taken ONCE: inside the tile loop it was a GOT load + lgkmcnt(0) in the middle of every MFMA interval)
    const half_t* zp;
    const float* ones_page;
    {
        unsigned long long za = (unsigned long long)g_wres_zero_page, oa = (unsigned long long)g_wres_ones_page;
        asm volatile("" : "+s"(za), "+s"(oa));
        zp = (const half_t*)za;
        ones_page = (const float*)oa;
    }
    auto issue_chunk = [&](int c) {            // the three pieces of chunk c -> ring buffer c (c is a constant wherever this is called)
        char* buf = smem + c * A_BYTES;
#pragma unroll
        for (int k = 0; k < 2; ++k) dma16(a_src[k] >= 0 ? xb + a_src[k] + c * c_step : zp, buf + (k * NTHR + wave * 64) * 16);
        if ((a_geo[2] >> 17) & 1) dma16(a_src[2] >= 0 ? xb + a_src[2] + c * c_step : zp, buf + (2 * NTHR + wave * NTAIL) * 16);
    };
    auto issue_w = [&](int b) {                // the weight image of candidate b (shared weights: w_bstride = 0)
        const half_t* wb = p.w + (long long)b * p.w_bstride;
        const int t = opaque(threadIdx.x);
#pragma unroll
        for (int k = 0; k < NWP; ++k) {
            const int v = k * NTHR + t, row = v >> 1;              // row = (c * 9 + tap) * 64 + n: 32 bytes, halves swizzled by (n >> 3) & 1
            const int n = row & 63, ct = row >> 6, c = ct / 9, tap = ct - 9 * c;
            const int lh = (v & 1) ^ ((n >> 3) & 1);
            dma16(wb + ((long long)tap * p.Neff + n) * p.Cin + c * 16 + lh * 8, smem + OFF_W + wave * 1024 + k * (NTHR * 16));
        }
    };

    if (!p.noise) *(float*)(smem + OFF_N + threadIdx.x * 4) = 0.f;     // (absent per-channel operands are DMA'd from the ones / zero pages)
    float tb[3] = {0.f, 0.f, 0.f};             // toRGB bias: read ONCE (a register load inside the tile loop waits for the whole ring)
    if (TRGB) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)         // (readfirstlane: the value is CONSUMED here, so hipcc's wait for it sits here and not in the loop)
            tb[cc] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.trgb_b[cc])));
    }
    __syncthreads();
    aim_a(cur);
    issue_w(cur.b);
    issue_chunk(0);
    issue_chunk(1);
    issue_chunk(2);
    WR_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();

    // lane-constant LDS offsets of the fragments, computed once and kept opaque (rebuilt per read they were ~100 VALU instructions inside
    // every MFMA interval)
    int xoff[4][3];                            // patch fragment of row wave * 2 + rr, column lr + tx: pixel * 32 + swizzled half
    int wbase[2];                              // weight fragment rows of chunks 0-1 / 2-3 (the immediate offset of a ds_read reaches 64 KB)
    int yoff;                                  // output: byte offset of the lane's 16 bytes inside its wave's row pair
    int lds_base;
    {
        const int t = threadIdx.x, lr = t & 31, kh = (t >> 5) & 1;
        const int lds0 = (int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;      // LDS byte address of smem (the asm reads address LDS directly)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const int pix = (wave * RW + rr) * PW + lr + tx;
                xoff[rr][tx] = opaque(lds0 + pix * 32 + ((kh ^ ((pix >> 3) & 1)) << 4));
            }
        const int wl = lds0 + OFF_W + lr * 32 + ((kh ^ ((lr >> 3) & 1)) << 4);
        wbase[0] = opaque(wl);
        wbase[1] = opaque(wl + 2 * 9 * NT * 32);
        yoff = opaque((lr * p.Cout + 8 * kh) * 2);
        lds_base = lds0;
    }
    h8 wf[2][3][2];                            // weight fragments of the current / the next tap row: [parity][tx][j]
    bool have_w = false;                       // (uniform) chunk 0's first tap row was requested in the previous tile's last phase
    for (;;) {
        const bool has_next = id + 1 < last;
        const Item nxt = has_next ? decode(id + 1) : cur;     // the ring never branches: the last tile re-requests itself (nobody reads those buffers)
        const int b = cur.b, ty0 = cur.ty0, tx0 = cur.tx0;

        f16x acc[RW][2];
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

        // epilogue operands (per-channel constants, the tile's noise values, the toRGB table) by LDS-DMA: the same number of pieces from every
        // wave whatever the layer has (absent arrays come from the ones / zero pages; waves 3-7 repeat an array: same bytes, same place)
        auto prefetch_epilogue = [&]() {
            const int t = opaque(threadIdx.x), lane = t & 63, lr = t & 31, kh = (t >> 5) & 1;
            const int arr = wave % 3;          // 0 dscale, 1 bias, 2 shift
            const float* src = arr == 0 ? (p.dscale ? p.dscale + (long long)b * p.ds_stride : ones_page)
                             : arr == 1 ? (p.bias ? p.bias : (const float*)zp)
                                        : (p.shift ? p.shift + (long long)b * p.ds_stride : (const float*)zp);
            dma4(src + lane, smem + OFF_C + arr * NT * 4);
            if (p.noise) dma4(p.noise + ((long long)(b / p.batch_size) * p.Ho + ty0 + wave * RW + kh) * p.Wo + tx0 + lr, smem + OFF_N + wave * 256);
            if (TRGB && lane < 6 * (NT / 8)) {
                const int row6 = lane / (NT / 8), piece = lane % (NT / 8);
                const int n = row6 < 3 ? row6 : 8 + (row6 - 3);
                dma16(p.trgb_tab + ((long long)b * 32 + n) * NT + piece * 8, smem + OFF_T);
            }
            if (TRGB && p.trgb_yprev) {
                // the wave's two output rows share the previous image's rows my - 1, my and columns mx0 - 1 .. mx0 + 15 (clamped loads, zero weight
                // outside: trgb_skip): 3 x 2 x 17 values as two pieces (a load into REGISTERS here made hipcc wait vmcnt(0) — for the whole ring)
                const int my = (ty0 + wave * RW) >> 1, mx0 = tx0 >> 1, h2 = p.Ho >> 1, w2 = p.Wo >> 1;
                const float* yp = p.trgb_yprev + (long long)b * 3 * h2 * w2;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int e = min(lane + 64 * u, 101), cc = e / 34, rem = e - cc * 34, dy = rem / 17, dx = rem - dy * 17;
                    dma4(yp + (cc * h2 + max(my - 1 + dy, 0)) * w2 + max(mx0 - 1 + dx, 0), smem + OFF_Y + wave * 512 + u * 256);
                }
            }
        };
        WRT(0);
        if (grp) __builtin_amdgcn_s_barrier();     // group 1 falls one barrier behind: its load intervals face group 0's MFMA intervals
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const char* As = smem + c * A_BYTES;
            // weight fragment (tap row s % 3 of chunk (s / 3) % 4, column tx, n block j) -> dst: one instruction, lane-constant base + immediate offset
#define WR_RD_W(dst, s_, tx_, j_)                                                                                                              \
    do {                                                                                                                                       \
        const int cc_ = ((s_) / 3) % NCH, tyy_ = (s_) % 3;                                                                                     \
        if (WR_ABL(16)) dst = h8{0, 0, 0, 0, 0, 0, 0, 0};                                                                                      \
        else WR_LDS_RD(dst, wbase[cc_ >> 1], (((cc_ & 1) * 9 + tyy_ * 3 + (tx_)) * NT + (j_) * 32) * 32);                                       \
    } while (0)
            // ---- load interval ---------------------------------------------------------------------------------------------------
            h8 xf[4][3];                       // [patch row rr = i + ty][tx]
            // chunk c of this tile has landed: it was requested three phases ago, and at least 3 x (phases since) pieces behind it (8 output
            // stores of the previous tile on top of that for chunks 0-2; chunk 3's epilogue operands were requested AHEAD of phase 1's pieces)
            if (c == 3) WR_WAIT_VM(6); else WR_WAIT_VM(14);
            WRT(1 + 5 * c);
            if (c == 0 && !have_w) {           // first tile of the range / behind a weight reload: nobody prefetched the first tap row
#pragma unroll
                for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                    for (int j = 0; j < 2; ++j) WR_RD_W(wf[0][tx][j], 0, tx, j);
                WR_WAIT_LGKM6(wf[0][0][0], wf[0][0][1], wf[0][1][0], wf[0][1][1], wf[0][2][0], wf[0][2][1]);
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) {
                    if (!WR_ABL(8)) WR_LDS_RD(xf[rr][tx], xoff[rr][tx], c * A_BYTES);
                    else xf[rr][tx] = h8{0, 0, 0, 0, 0, 0, 0, 0};
                }
            if (c == 1) aim_a(nxt);            // (the pieces of THIS tile's chunk 3 went out in phase 0; from here on the ring requests the next tile)
            WR_WAIT_LGKM6(xf[0][0], xf[0][1], xf[0][2], xf[1][0], xf[1][1], xf[1][2]);
            WR_WAIT_LGKM6(xf[2][0], xf[2][1], xf[2][2], xf[3][0], xf[3][1], xf[3][2]);
            WRT(2 + 5 * c);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            WRT(3 + 5 * c);
            // ---- MFMA interval ---------------------------------------------------------------------------------------------------
            auto ring_pieces = [&]() {
                if (WR_ABL(32)) return;
                if (c == 0) {
                    issue_chunk(3);            // of THIS tile: buffer 3 was read last in the previous tile's phase 3
                } else {
                    if (c == 1) prefetch_epilogue();
                    issue_chunk(c - 1);        // of the NEXT tile: both groups are past their last read of that buffer
                }
            };
            // XS by-product: FIR 4x4 (pad 1) + ::2 of this chunk of the INPUT map, 8 x 16 pixels x 2 halves = 256 vectors: a lane pair (l, l + 32)
            // shares one — lanes < 32 filter tap rows 0-1, lanes >= 32 rows 2-3, v_permlane32_swap brings the four rows together in the lower
            // lane, which folds them in conv_tiled<xs>'s order (explicit FMA forms: conv_glds.hip) and stores 16 bytes
            h8 xa[4], hr[2], k125, k375;
            int xs_half = 0, xs_ly = 0, xs_lx = 0, xs_jy = 0;
            if (XS) {
                const int tq = opaque(threadIdx.x);
                xs_half = tq & 1; xs_lx = (tq >> 1) & 15; xs_jy = (tq >> 5) & 1; xs_ly = wave;
#pragma unroll
                for (int e = 0; e < 8; ++e) { k125[e] = (half_t)0.125f; k375[e] = (half_t)0.375f; }
            }
            auto xs_read = [&](int r2) {       // tap row 2 * xs_jy + r2
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    const int P = (2 * xs_ly + 2 * xs_jy + r2) * PW + 2 * xs_lx + jx;
                    const int va = lds_base + P * 32 + ((xs_half ^ ((P >> 3) & 1)) << 4);
                    WR_LDS_RD(xa[jx], va, c * A_BYTES);
                }
            };
            auto xs_fold = [&](int r2) {
                hr[r2] = __builtin_elementwise_fma(xa[1] + xa[2], k375, (xa[0] + xa[3]) * k125);     // (conv_tiled<xs>'s contraction, made explicit there too)
            };
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ty = 0; ty < 3; ++ty) {
                const int s = c * 3 + ty;
                // the NEXT tap row's six weight fragments (the next tile's first row behind the last one: the weights do not change) are requested at
                // the TOP of this row: hipcc waits lgkmcnt(0) in front of a row's first MFMA whatever the order — requested between the MFMA groups,
                // the last pair was 4 MFMAs old at that wait and every tap row began with an exposed LDS round trip (stamps: 2400-3800 clocks per
                // interval of 36 MFMAs); at the top they have the row's 12 MFMAs to land
                __builtin_amdgcn_sched_barrier(0);
                // this row's fragments (requested at the top of the previous row — the previous phase's last row for ty = 0) have landed: nothing
                // newer is outstanding at this point, so the wait is for reads that had 12 MFMAs to come back
                WR_WAIT_LGKM6(wf[s & 1][0][0], wf[s & 1][0][1], wf[s & 1][1][0], wf[s & 1][1][1], wf[s & 1][2][0], wf[s & 1][2][1]);
                if (XS && ty > 0) WR_WAIT_LGKM4(xa[0], xa[1], xa[2], xa[3]);
#pragma unroll
                for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                    for (int j = 0; j < 2; ++j) WR_RD_W(wf[(s + 1) & 1][tx][j], s + 1, tx, j);
                if (XS) {                      // two tap rows per lane: requested beside the weight fragments, folded a tap row later
                    if (ty == 1) xs_fold(0);
                    if (ty == 2) xs_fold(1);
                    if (ty < 2) xs_read(ty);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < RW; ++i)
                            if (!WR_ABL(4)) acc[i][j] = mfma32(wf[s & 1][tx][j], xf[i + ty][tx], acc[i][j]);
                    if (ty == 0 && tx == 0) { __builtin_amdgcn_sched_barrier(0); ring_pieces(); __builtin_amdgcn_sched_barrier(0); }
                }
            }
            if (XS) {
                // lanes < 32 hold rows 0, 1; lanes >= 32 rows 2, 3: the swap hands the upper lanes' registers to the lower lanes
                h8 r2v = hr[0], r3v = hr[1], d0 = hr[0], d1 = hr[1];     // (upper lanes: hr[0] = row 2, hr[1] = row 3)
                swap32(r2v, d0);               // lower lanes of d0 / d1 = the upper lanes' rows 2 / 3
                swap32(r3v, d1);
                const h8 s03 = hr[0] + d1, s12 = hr[1] + d0;
                const h8 o = __builtin_elementwise_fma(s12, k375, s03 * k125);
                if (xs_jy == 0)
                    *(h8*)(p.xs_out + (((long long)b * (p.H >> 1) + (ty0 >> 1) + xs_ly) * (p.W >> 1) + (tx0 >> 1) + xs_lx) * p.Cin + c * 16 + xs_half * 8) = o;
            }
            __builtin_amdgcn_s_setprio(0);
            WRT(4 + 5 * c);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            WRT(5 + 5 * c);
        }

        // ---- epilogue: group 0 re-aligns first (a weight reload must not overtake group 1's last reads; the operands of phase 1 were waited
        // for in phase 3 and two barriers lie in between).  No LDS but the operand tables: the ring keeps running underneath. ---------------
        if (!grp) __builtin_amdgcn_s_barrier();
        WRT(21);
        const bool reload = has_next && p.w_bstride != 0 && nxt.b != b;      // uniform
        if (reload) issue_w(nxt.b);
        const int t = opaque(threadIdx.x), lane = t & 63, lr = lane & 31, kh = lane >> 5;
        const float* Cc = (const float*)(smem + OFF_C);
        const int oyb = ty0 + wave * RW, ox = tx0 + lr;              // lane's pixel of tile row i: (oyb + i, ox)
        const int rcs = p.res_cs ? p.res_cs : p.Cout;
        const ActK ak = act_consts(p.act, p.out_scale);
        // the wave's two output rows: scalar base (tile, wave) + lane constant
        char* yrow = (char*)(p.y + (((long long)b * p.Ho + oyb) * p.Wo + tx0) * p.Cout);
        const long long yrow_pitch = (long long)p.Wo * p.Cout * 2;
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
            if (WR_ABL(1)) { if (acc[0][j][0] == 12345.678f) p.y[0] = (half_t)1.f; continue; }
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
                    va[i][g] = out;
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
            // lane (px, kh) holds channels 8 g + 4 kh .. + 3 of every g: the pair (px, 0) / (px, 1) trades quads so that the lower lane owns the
            // eight channels of g = 2 gp and the upper lane those of g = 2 gp + 1 — 16 contiguous bytes each, 32 per pixel and instruction
#pragma unroll
            for (int i = 0; i < RW; ++i)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    h4 lo = va[i][2 * gp], hi = va[i][2 * gp + 1];
                    swap32(lo, hi);
                    // after the swap: lower lanes lo = own quad of g0, hi = the partner's quad of g0; upper lanes lo = the partner's quad of g1, hi = own quad of g1
                    const h8 ov = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    if (!WR_ABL(2) || ov[0] == (half_t)123.f) *(h8*)(yrow + i * yrow_pitch + (j * 32 + 16 * gp) * 2 + yoff) = ov;
                }
        }
        if (TRGB) {
            const long long hw = (long long)p.Ho * p.Wo;
            float* yo = p.trgb_yout + (long long)b * 3 * hw + (long long)(oyb + kh) * p.Wo + ox;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                float r = tb[cc] + (rgb[cc] + rgb[4 + cc] * (1.f / 2048.f));
                if (p.trgb_yprev) {
                    float ytap[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) ytap[q] = *(const float*)(smem + OFF_Y + wave * 512 + (cc * 34 + (q >> 1) * 17 + (lr >> 1) + (q & 1)) * 4);
                    r += trgb_skip(ytap, oyb + kh, ox);
                }
                yo[cc * hw] = r;
            }
        }
        WRT(22);
        if (reload) {                          // the next candidate's weights: every wave's pieces landed before anybody reads them
            WR_WAIT_VM(0);
            __builtin_amdgcn_s_barrier();
        }
        have_w = !reload;                      // (the fragments requested in the last phase were the OLD candidate's)
        if (!has_next) break;
        ++id;
        cur = nxt;
    }
    WR_WAIT_VM(0);                             // the last tile's self-prefetch
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
    if (p.x_planar8 || p.y_planar8) return nullptr;   // (pixel-major maps only: the chunk-planar layout is conv_wreg's)
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
#ifdef GLASS_AB_KNOBS
    static const int abl = glass_knob("GLASS_WRES_ABLATE") ? atoi(glass_knob("GLASS_WRES_ABLATE")) : 0;
    struct TraceDump {         // bit 64: print the stamps of this launch when the scope ends (synchronises)
        unsigned long long* d = nullptr;
        hipStream_t st;
        const char* name;
        ~TraceDump() {
            if (!d) return;
            unsigned long long hb[23 * 8];
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(hb, d, sizeof hb, hipMemcpyDeviceToHost);
            (void)hipFree(d);
            for (int w = 0; w < 8; w += 4) {
                fprintf(stderr, "[wres %s w%d] per chunk: vm-wait reads barrierA mfma barrierB |", name, w);
                for (int c = 0; c < 4; ++c) {
                    fprintf(stderr, " %llu", hb[(1 + 5 * c) * 8 + w] - hb[(c ? 5 * c : 0) * 8 + w]);
                    for (int ph = 2; ph <= 5; ++ph) fprintf(stderr, " %llu", hb[(ph + 5 * c) * 8 + w] - hb[(ph - 1 + 5 * c) * 8 + w]);
                    fprintf(stderr, " |");
                }
                fprintf(stderr, " realign %llu epilogue %llu total %llu\n", hb[21 * 8 + w] - hb[20 * 8 + w], hb[22 * 8 + w] - hb[21 * 8 + w], hb[22 * 8 + w] - hb[w]);
            }
        }
    } dump;
    if (abl & 64) {
        (void)hipMalloc(&dump.d, 23 * 8 * sizeof(unsigned long long));
        (void)hipMemset(dump.d, 0, 23 * 8 * sizeof(unsigned long long));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wres_trace), &dump.d, sizeof dump.d);
        dump.st = st;
        dump.name = name;
    }
#define WR_ABL_PASS , abl
#else
#define WR_ABL_PASS
#endif
    if (p.trgb_yout) hipLaunchKernelGGL((conv_wres_kernel<true, false>), dim3(grid), dim3(NTHR), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_wg WR_ABL_PASS);
    else if (p.xs_out) hipLaunchKernelGGL((conv_wres_kernel<false, true>), dim3(grid), dim3(NTHR), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_wg WR_ABL_PASS);
    else hipLaunchKernelGGL((conv_wres_kernel<false, false>), dim3(grid), dim3(NTHR), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_wg WR_ABL_PASS);
    return name;
}
