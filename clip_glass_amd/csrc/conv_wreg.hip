// conv_wreg.hip — 3x3 stride-1 convolution 64 -> 64 channels at 512^2 (G's last-but-one conv with its toRGB, D's second block's first conv
// with the blur-down by-product; stylegan2/modules.py:920-967, models.py:852-870, modules.py:1238-1254) with the WHOLE weight tensor in
// REGISTERS: one wave per SIMD (256 threads, 512 registers per lane), 288 of them the layer's 72 weight fragments.
//
// Why (round 6).  conv_wres.hip keeps the weights in LDS and runs two waves per SIMD in a ping-pong.  Its stamps and ablations (DESIGN
// "Round 6") say where its 1.6-1.7 ms go: an interval of 36 MFMAs takes ~2150 clocks instead of 1152, because every MFMA needs a weight
// fragment AND a patch fragment from LDS (0.83-1.0 ds_read_b128 per MFMA and wave: the CU's LDS read rate, ~91 B/clock measured, is the
// bound, and a wave's own reads sit in its MFMA stream), and the epilogue's VALU runs beside nothing.  At 64 -> 64 channels the weights are
// 73 728 B = 72 fragments of 16 B per lane: they FIT the register file of a wave that has a SIMD to itself.  Then
//   * an MFMA reads one operand from registers that never change; the patch fragments are 12 ds_read_b128 per 36 MFMAs (0.33 per MFMA,
//     42 B/clock per CU) — the LDS is no longer near its rate and nothing is read inside an MFMA run but the next chunk's 12 fragments;
//   * no weight image in LDS: the 160 KB hold a ring of THREE tiles of patch (12 chunk buffers), requested two tiles ahead — ~87 KB in
//     flight per CU all the time, one barrier per tile;
//   * a wave is its own pipeline: the next chunk's fragments are requested (inline asm: hipcc would wait lgkmcnt(0) for them at once)
//     in front of this chunk's 36 MFMAs, the ring's pieces and the by-product's arithmetic sit between the MFMAs.
// Tile = 8 rows x 32 px x 64 channels on four waves (wave = 2 rows x 2 n blocks, as conv_wres), K loop = four 16-channel chunks.
// Patch chunk image in LDS: [10 rows][2 halves of 8 channels][34 px] x 16 B — a fragment read is 32 lanes x 16 contiguous bytes (no swizzle
// needed), a tap column is +16 B, a tap row +1088 B: one address register, immediate offsets.
// Input layout: pixel-major [B][H][W][64], or chunk-planar [B][8][H][W][8] (common.h x_planar8) written by upfir2<false> / dblock0 for this
// kernel: a piece of the ring is then 64 lanes x 16 B = 1 KB of contiguous memory, and no 128-byte line is fetched twice.
// Epilogue: conv_wres.hip's second form (v_permlane32_swap, 16-byte stores, scalar bases); operands (per-channel constants, noise, toRGB
// table, the previous skip image's taps) by LDS-DMA one tile ahead.  toRGB (TRGB) and the blur-down by-product (XS) as in conv_glds.hip.
// K order per accumulator: chunk, tap row, tap column.
#include "common.h"
#include "kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace {
constexpr int NT = 64, NTHR = 256, TW = 32, TH = 8, RW = 2;
constexpr int PH = TH + 2, PW = TW + 2;
constexpr int NCH = 4;                                   // 16-channel chunks
constexpr int HALFB = PW * 16;                           // 544: one half (8 channels) of a patch row
constexpr int ROWB = 2 * HALFB;                          // 1088
constexpr int NVA = PH * 2 * PW;                         // 680 vectors of a chunk
constexpr int NVP = 704;                                 // ... padded to 11 wave-pieces of 64: every piece is a FULL wave instruction
constexpr int A_BYTES = NVP * 16;                        // 11264
constexpr int NSLOT = 3;                                 // tiles in the ring
constexpr int SLOT_BYTES = NCH * A_BYTES;                // 45056
constexpr int OFF_C = NSLOT * SLOT_BYTES;                // 135168: 3 x [dscale | bias | shift][NT] fp32
constexpr int C_BYTES = 3 * NT * 4;
constexpr int OFF_T = OFF_C + NSLOT * C_BYTES;           // 3 x toRGB weight rows [hi r,g,b | lo r,g,b][NT] fp16
constexpr int T_BYTES = 1024;                            // (768 B of table + the 16 idle lanes of its one full-wave piece)
constexpr int OFF_N = OFF_T + NSLOT * T_BYTES;           // 3 x the tile's noise values [4 waves][2 rows][32 px] fp32
constexpr int N_BYTES = 4 * 64 * 4;
constexpr int OFF_Y = OFF_N + NSLOT * N_BYTES;           // 3 x the previous skip image's taps [4 waves][3 ch][2 rows][17 cols] fp32 (128 slots per wave)
constexpr int Y_BYTES = 4 * 128 * 4;
constexpr int LDS_BYTES = OFF_Y + NSLOT * Y_BYTES;       // 149760
static_assert(LDS_BYTES <= 163840 && NVP == 2 * NTHR + 3 * 64 && NVP >= NVA, "one workgroup per CU; three full pieces per wave and chunk");

__device__ __attribute__((aligned(256))) float g_wreg_zero_page[64];   // zero-initialised: source of the zero padding and of absent operands
__device__ __attribute__((aligned(256))) float g_wreg_ones_page[64] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
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
#define WG_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// Fragment reads as inline assembly: hipcc puts lgkmcnt(0) in front of the first MFMA that uses ANY outstanding ds_read, i.e. it would wait
// for the reads just requested for the NEXT chunk.  A read it cannot see needs no wait in its eyes; the waits are the explicit ones below,
// tied to the registers they release by "+v" operands.
#define WG_LDS_RD(dst, vaddr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(vaddr), "n"(imm))
#define WG_WAIT_LGKM6(a, b, c, d, e, f) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f)::"memory")
#define WG_WAIT_LGKM4(a, b, c, d) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory")
// developer builds only: timing experiments that switch parts of a tile off — 1 the whole epilogue, 2 its global
// stores, 4 the K loop's MFMAs, 8 the patch fragment reads, 32 the ring's requests, 64 the epilogue's lane swaps, 128 its arithmetic (WRONG RESULTS)
// (a RUN-TIME bit costs a branch per MFMA group and distorts what it measures: the bits are a COMPILE-time constant, -DWG_ABLATE_CT=<bits>,
// one library per experiment — tools/build_ablations.sh)
#define WG_ABL_ARG
#ifdef WG_ABLATE_CT
#define WG_ABL(bit) (((WG_ABLATE_CT) & (bit)) != 0)
#else
#define WG_ABL(bit) false
#endif
}  // namespace

template <bool TRGB, bool XS>
__global__ __launch_bounds__(256, 1) void conv_wreg_kernel(ConvParams p, int tiles_x, int tiles_y, int PT, int per_wg WG_ABL_ARG) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    const int lds0 = (int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;      // LDS byte address of smem (the asm reads address LDS directly)

    // ---- DMA sources --------------------------------------------------------------------------------------------------------------
    // vector v of a patch chunk sits at byte v * 16: patch row v / 68, half (v % 68) / 34, pixel (v % 68) % 34.  Thread t carries vectors
    // t, 256 + t and 512 + 64 min(wave, 2) + lane: EVERY piece is a full 64-lane instruction — vectors 680 .. 703 are padding fed from the
    // zero page, and wave 3 repeats wave 2's third piece (same bytes, same place).  A piece under a lane mask (the first form: 42 lanes per wave)
    // put the LDS-DMA inside divergent control flow, and hipcc merged the masked region with the unmasked pieces behind it: both paths got
    // their own M0, the join took lane 0's (readfirstlane) — the other lanes' piece landed a chunk further on (wrong results, toRGB instance).
    const int pixs = p.x_planar8 ? 8 : p.Cin;                     // elements between pixels
    const int halfs = p.x_planar8 ? p.H * p.W * 8 : 8;            // between the 8-channel halves of a chunk
    const int c_step = p.x_planar8 ? p.H * p.W * 16 : 16;         // between the 16-channel chunks
    const half_t* zp;
    const float* ones_page;
    {   // (the pages' addresses are taken ONCE: inside the tile loop they are GOT loads + lgkmcnt(0))
        unsigned long long za = (unsigned long long)g_wreg_zero_page, oa = (unsigned long long)g_wreg_ones_page;
        asm volatile("" : "+s"(za), "+s"(oa));
        zp = (const half_t*)za;
        ones_page = (const float*)oa;
    }
    const half_t* xb = p.x;
    int a_src[3];             // element offset into the image (< 2^31: launcher), or -1 = zero page
    auto aim_a = [&](const Item& w) {
        xb = p.x + (long long)w.b * p.x_bstride;
        const int org = ((w.ty0 - 1) * p.W + w.tx0 - 1) * pixs;      // the patch's first pixel (uniform; negative at the top / left border)
        // (the vectors' patch coordinates are re-derived per tile from an opaque thread id — ~50 instructions: as lane constants they were six
        // registers this kernel does not have: hipcc spilled them and waited vmcnt(0) for the reload, i.e. for the whole ring, every tile)
        const int t = opaque(threadIdx.x), lane = t & 63;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int v = k < 2 ? k * NTHR + t : 2 * NTHR + min(wave, 2) * 64 + lane;
            const int pr = v / (2 * PW), wv = v - pr * (2 * PW), hf = wv / PW, pc = wv - hf * PW;
            const int iy = w.ty0 - 1 + pr, ix = w.tx0 - 1 + pc;
            const bool ok = v < NVA && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            a_src[k] = ok ? org + (pr * p.W + pc) * pixs + hf * halfs : -1;
        }
    };
    auto issue_chunk = [&](int slot_off, int c) {      // the three pieces of chunk c of the aimed tile -> ring slot at byte slot_off
        char* buf = smem + slot_off + c * A_BYTES;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            dma16(a_src[k] >= 0 ? xb + a_src[k] + c * c_step : zp, buf + (k < 2 ? k * NTHR + wave * 64 : 2 * NTHR + min(wave, 2) * 64) * 16);
    };
    // epilogue operands of a tile (per-channel constants, noise values, the toRGB table, the previous skip image's taps) by LDS-DMA into
    // operand slot es: the same number of pieces from every wave whatever the layer has (absent arrays come from the ones / zero pages;
    // wave 3 repeats array 0: same bytes, same place)
    auto issue_operands = [&](const Item& w, int es) {
        const int t = opaque(threadIdx.x), lane = t & 63, lr = t & 31, kh = (t >> 5) & 1;
        const int arr = wave % 3;              // 0 dscale, 1 bias, 2 shift
        const float* src = arr == 0 ? (p.dscale ? p.dscale + (long long)w.b * p.ds_stride : ones_page)
                         : arr == 1 ? (p.bias ? p.bias : (const float*)zp)
                                    : (p.shift ? p.shift + (long long)w.b * p.ds_stride : (const float*)zp);
        dma4(src + lane, smem + OFF_C + es * C_BYTES + arr * NT * 4);
        if (p.noise) dma4(p.noise + ((long long)(w.b / p.batch_size) * p.Ho + w.ty0 + wave * RW + kh) * p.Wo + w.tx0 + lr, smem + OFF_N + es * N_BYTES + wave * 256);
        if (TRGB) {                            // (a full-wave piece: lanes >= 48 fetch the zero page into the slot's slack — no LDS-DMA under a lane mask, see above)
            const int row6 = lane / (NT / 8), piece = lane % (NT / 8);
            const int n = row6 < 3 ? row6 : 8 + (row6 - 3);
            dma16(lane < 6 * (NT / 8) ? p.trgb_tab + ((long long)w.b * 32 + n) * NT + piece * 8 : zp, smem + OFF_T + es * T_BYTES);
        }
        if (TRGB && p.trgb_yprev) {
            // the wave's two output rows share the previous image's rows my - 1, my and columns mx0 - 1 .. mx0 + 15 (clamped loads, zero weight
            // outside: trgb_skip): 3 x 2 x 17 values as two pieces
            const int my = (w.ty0 + wave * RW) >> 1, mx0 = w.tx0 >> 1, h2 = p.Ho >> 1, w2 = p.Wo >> 1;
            const float* yp = p.trgb_yprev + (long long)w.b * 3 * h2 * w2;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = min(lane + 64 * u, 101), cc = e / 34, rem = e - cc * 34, dy = rem / 17, dx = rem - dy * 17;
                dma4(yp + (cc * h2 + max(my - 1 + dy, 0)) * w2 + max(mx0 - 1 + dx, 0), smem + OFF_Y + es * Y_BYTES + wave * 512 + u * 256);
            }
        }
    };

    // ---- the weights: 72 fragments per lane, loaded once per candidate ------------------------------------------------------------------
    // lane (lr, kh) of n block j holds w[tap][n = j * 32 + lr][channels c * 16 + kh * 8 .. + 7] — the MFMA's A operand as it stands
    h8 wreg[NCH][9][2];
    auto load_w = [&](int b) {
        const half_t* wb = p.w + (long long)b * p.w_bstride;
        const int t = opaque(threadIdx.x), lr = t & 31, kh = (t >> 5) & 1;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int j = 0; j < 2; ++j) wreg[c][tap][j] = *(const h8*)(wb + ((long long)tap * p.Neff + j * 32 + lr) * p.Cin + c * 16 + kh * 8);
        __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0), visible to hipcc: the fragments are complete HERE, no wait is owed at their uses in the loop
    };

    if (!p.noise) {
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) *(float*)(smem + OFF_N + s * N_BYTES + threadIdx.x * 4) = 0.f;
    }
    float tb[3] = {0.f, 0.f, 0.f};             // toRGB bias: read ONCE (a register load inside the tile loop waits for the whole ring)
    if (TRGB) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
            tb[cc] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.trgb_b[cc])));
    }
    __syncthreads();
    int id = first;
    Item cur = decode(id);
    Item nxt = decode(min(id + 1, last - 1));
    load_w(cur.b);
    // prologue: tiles `first` and `first + 1` into slots 0 and 1, the first tile's operands into operand slot 0
    issue_operands(cur, 0);
    aim_a(cur);
#pragma unroll
    for (int c = 0; c < NCH; ++c) issue_chunk(0, c);
    aim_a(nxt);
#pragma unroll
    for (int c = 0; c < NCH; ++c) issue_chunk(SLOT_BYTES, c);
    WG_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();

    // lane constants of the tile loop
    int xbase0, yoff;
    {
        const int t = threadIdx.x, lr = t & 31, kh = (t >> 5) & 1;
        xbase0 = opaque(lds0 + (wave * RW) * ROWB + kh * HALFB + lr * 16);       // patch fragment (row rr, column tx) of chunk c: + c * A_BYTES + rr * ROWB + tx * 16
        yoff = opaque((lr * p.Cout + 8 * kh) * 2);                              // output: byte offset of the lane's 16 bytes inside its wave's row pair
    }
    int slot_off = 0;                          // ring slot of the current tile (bytes); the next tile's is one on, the requested tile's two on
    int es = 0;                                // operand slot of the current tile
    h8 xf[2][4];                               // patch fragments of two tap columns: [column parity][patch row rr = i + ty]
    // Rolling by tap column (column index q = chunk * 3 + tx, 12 per tile): a column's four fragments are dead behind its 12 MFMAs, and column
    // q + 2 is requested into the same registers right there — 12 MFMAs (~400 clocks) ahead of its use.  (A whole second chunk of fragments
    // does not fit beside 288 weight registers and 64 accumulators.)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) WG_LDS_RD(xf[q][rr], xbase0, rr * ROWB + q * 16);
    bool first_tile = true;
    for (;;) {
        // the tile's first two columns of fragments, ahead of the address work below.  (NOT ahead of the previous tile's epilogue, where they
        // would have the most cover: hipcc believes an asm read's registers are written AT the asm statement and, under the epilogue's register
        // pressure, parked them in AGPRs before the data had landed — wrong results in the toRGB instance.)
        if (!first_tile) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) WG_LDS_RD(xf[q][rr], xbase0 + slot_off, rr * ROWB + q * 16);
        }
        first_tile = false;
        const bool has_next = id + 1 < last;
        const Item nx2 = decode(min(id + 2, last - 1));       // the ring never branches: past the end it re-requests the last tile (nobody reads those slots)
        const int b = cur.b, ty0 = cur.ty0, tx0 = cur.tx0;
        const int slot_nxt = slot_off + SLOT_BYTES >= NSLOT * SLOT_BYTES ? 0 : slot_off + SLOT_BYTES;
        const int slot_req = slot_nxt + SLOT_BYTES >= NSLOT * SLOT_BYTES ? 0 : slot_nxt + SLOT_BYTES;
        const int es_nxt = es == NSLOT - 1 ? 0 : es + 1;
        const int xbase = xbase0 + slot_off;
        aim_a(nx2);

        f16x acc[RW][2];
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            // XS by-product: FIR 4x4 (pad 1) + ::2 of this chunk of the INPUT map, 4 x 16 pixels x 2 halves = 128 vectors: a lane pair (l, l + 32)
            // shares one — lanes < 32 filter tap rows 0-1, lanes >= 32 rows 2-3, v_permlane32_swap brings the four rows together in the lower
            // lane, which folds them in conv_tiled<xs>'s order (explicit FMA forms: conv_glds.hip) and stores 16 bytes
            h8 xa[4], hr[2], k125, k375;
            int xs_half = 0, xs_lx = 0, xs_jy = 0, xs_va = 0;
            if (XS) {
                const int tq = opaque(threadIdx.x);
                xs_half = tq & 1; xs_lx = (tq >> 1) & 15; xs_jy = (tq >> 5) & 1;
                xs_va = lds0 + slot_off + (2 * wave + 2 * xs_jy) * ROWB + xs_half * HALFB + 2 * xs_lx * 16;     // patch row 2 ly + 2 jy, column 2 lx
#pragma unroll
                for (int e = 0; e < 8; ++e) { k125[e] = (half_t)0.125f; k375[e] = (half_t)0.375f; }
            }
            auto xs_read = [&](int r2) {       // tap row 2 * xs_jy + r2: four pixels = 64 contiguous bytes
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) WG_LDS_RD(xa[jx], xs_va, c * A_BYTES + r2 * ROWB + jx * 16);
            };
            auto xs_fold = [&](int r2) {
                hr[r2] = __builtin_elementwise_fma(xa[1] + xa[2], k375, (xa[0] + xa[3]) * k125);     // (conv_tiled<xs>'s contraction, made explicit there too)
            };
            __builtin_amdgcn_sched_barrier(0);
            if (XS) xs_read(0);
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                // this column's four fragments were requested 12 MFMAs ago (columns 0, 1 of a tile: behind the previous tile's barrier, in front
                // of its epilogue); LDS returns in order, so "at most the reads requested since" is the wait: one column = 4 (the tile's last
                // column: 0).  The by-product's reads in between only make it earlier.
                const int q = c * 3 + tx, qs = q & 1;
                __builtin_amdgcn_sched_barrier(0);
                if (q == 3 * NCH - 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[qs][0]), "+v"(xf[qs][1]), "+v"(xf[qs][2]), "+v"(xf[qs][3])::"memory");
                else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xf[qs][0]), "+v"(xf[qs][1]), "+v"(xf[qs][2]), "+v"(xf[qs][3])::"memory");
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < RW; ++i)
                            if (!WG_ABL(4)) acc[i][j] = mfma32(wreg[c][ty * 3 + tx][j], xf[qs][i + ty], acc[i][j]);
                    if (ty == 0 && tx == 0 && !WG_ABL(32)) {      // the ring: tile id + 2, one chunk per chunk; tile id + 1's operands ahead of the first
                        __builtin_amdgcn_sched_barrier(0);
                        if (c == 0) issue_operands(nxt, es_nxt);
                        issue_chunk(slot_req, c);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (XS) {                      // two tap rows per lane, a column of MFMAs apart; waited for BEFORE anything newer is requested
                    if (tx == 0) { WG_WAIT_LGKM4(xa[0], xa[1], xa[2], xa[3]); xs_fold(0); xs_read(1); }
                    if (tx == 1) { WG_WAIT_LGKM4(xa[0], xa[1], xa[2], xa[3]); xs_fold(1); }
                }
                if (q + 2 < 3 * NCH) {         // column q + 2 of this tile, into the registers this column just released
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        if (!WG_ABL(8)) WG_LDS_RD(xf[qs][rr], xbase, ((q + 2) / 3) * A_BYTES + rr * ROWB + ((q + 2) % 3) * 16);
                        else xf[qs][rr] = h8{0, 0, 0, 0, 0, 0, 0, 0};
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (XS) {
                // lanes < 32 hold rows 0, 1; lanes >= 32 rows 2, 3: the swap hands the upper lanes' registers to the lower lanes
                h8 r2v = hr[0], r3v = hr[1], d0 = hr[0], d1 = hr[1];     // (upper lanes: hr[0] = row 2, hr[1] = row 3)
                swap32(r2v, d0);               // lower lanes of d0 / d1 = the upper lanes' rows 2 / 3
                swap32(r3v, d1);
                const h8 s03 = hr[0] + d1, s12 = hr[1] + d0;
                const h8 o = __builtin_elementwise_fma(s12, k375, s03 * k125);
                if (xs_jy == 0)
                    *(h8*)(p.xs_out + (((long long)b * (p.H >> 1) + (ty0 >> 1) + wave) * (p.W >> 1) + (tx0 >> 1) + xs_lx) * p.Cin + c * 16 + xs_half * 8) = o;
            }
        }

        // ---- the next tile has landed (requested during the previous tile's K loop: at least this tile's twelve pieces are younger), every
        // wave is done reading this tile's slot; the next tile's first chunk is requested ahead of the epilogue -------------------------
        __builtin_amdgcn_sched_barrier(0);
        if (XS) WG_WAIT_VM(16); else WG_WAIT_VM(12);      // (XS: + the four by-product stores of this K loop)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);

        // ---- epilogue (conv_wres.hip's second form): no LDS but the operand tables ---------------------------------------------------------
        const bool reload = has_next && p.w_bstride != 0 && nxt.b != b;      // uniform
        const int t = opaque(threadIdx.x), lane = t & 63, lr = lane & 31, kh = lane >> 5;
        const float* Cc = (const float*)(smem + OFF_C + es * C_BYTES);
        const int oyb = ty0 + wave * RW, ox = tx0 + lr;              // lane's pixel of tile row i: (oyb + i, ox)
        const ActK ak = act_consts(p.act, p.out_scale);
        char* yrow = (char*)(p.y + (((long long)b * p.Ho + oyb) * p.Wo + tx0) * p.Cout);      // the wave's two output rows: scalar base + lane constant
        const long long yrow_pitch = (long long)p.Wo * p.Cout * 2;
        float nzr[RW];
#pragma unroll
        for (int i = 0; i < RW; ++i) nzr[i] = p.noise_strength * *(const float*)(smem + OFF_N + es * N_BYTES + (wave * 64 + i * 32 + lr) * 4);
        f16x rgb;
#pragma unroll
        for (int q = 0; q < 16; ++q) rgb[q] = 0.f;
        const int tn = lr & 15;
        const char* Trow = smem + OFF_T + es * T_BYTES + (((tn >> 3) & 1) * 3 + min(tn & 3, 2)) * (NT * 2) + kh * 16;
        const bool trow_ok = (tn & 3) < 3;
        const h8 hzero = {0, 0, 0, 0, 0, 0, 0, 0};
        // PLAIN (both engine layers: pre-modulated weights / a plain D conv): no per-channel scale, no shift — one constant quad per channel
        // group instead of three, a slice's four requested up front (a wave alone on its SIMD pays every LDS round trip it waits for in line)
        auto epilogue = [&](auto plain_tag) {
            constexpr bool PLAIN = decltype(plain_tag)::value;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (WG_ABL(1)) { if (acc[0][j][0] == 12345.678f) p.y[0] = (half_t)1.f; continue; }
                f4 bias4[4];
                if (PLAIN) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) bias4[g] = *(const f4*)(Cc + NT + j * 32 + 8 * g + 4 * kh);
                }
                h4 va[RW][4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = j * 32 + 8 * g + 4 * kh;
                    f4 dq = {1.f, 1.f, 1.f, 1.f}, bq;
                    if (PLAIN) {
                        bq = bias4[g];
                    } else {
                        dq = *(const f4*)(Cc + nl);
                        bq = *(const f4*)(Cc + NT + nl) + *(const f4*)(Cc + 2 * NT + nl);      // bias + shift
                    }
#pragma unroll
                    for (int i = 0; i < RW; ++i) {
                        const f4 a = {acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]};
                        f4 v = WG_ABL(128) ? a : act_apply(PLAIN ? a + bq + nzr[i] : a * dq + bq + nzr[i], ak);      // (a * 1 + b == a + b: the same bits)
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
                        if (!WG_ABL(64)) swap32(lo, hi);
                        const h8 ov = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        if (!WG_ABL(2) || ov[0] == (half_t)123.f) *(h8*)(yrow + i * yrow_pitch + (j * 32 + 16 * gp) * 2 + yoff) = ov;
                    }
            }
        };
        if (!p.dscale && !p.shift) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        if (TRGB) {
            const long long hw = (long long)p.Ho * p.Wo;
            float* yo = p.trgb_yout + (long long)b * 3 * hw + (long long)(oyb + kh) * p.Wo + ox;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                float r = tb[cc] + (rgb[cc] + rgb[4 + cc] * (1.f / 2048.f));
                if (p.trgb_yprev) {
                    float ytap[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        ytap[q] = *(const float*)(smem + OFF_Y + es * Y_BYTES + wave * 512 + (cc * 34 + (q >> 1) * 17 + (lr >> 1) + (q & 1)) * 4);
                    r += trgb_skip(ytap, oyb + kh, ox);
                }
                yo[cc * hw] = r;
            }
        }
        if (reload) load_w(nxt.b);             // the next candidate's weights (per-sample weights: once per ~four workgroups)
        if (!has_next) break;
        ++id;
        cur = nxt;
        nxt = nx2;
        slot_off = slot_nxt;
        es = es_nxt;
    }
    WG_WAIT_VM(0);                             // the last tiles' self-prefetches
}

// would a plain 3x3 layer of this geometry run here?  (the producers of its input ask before they write the chunk-planar layout)
bool conv_wreg_supported(int Cin, int Cout, int H, int W) {
    static const bool off = glass_knob("GLASS_NO_WREG") != nullptr;       // A/B knob (developer build): conv_wres / conv_tiled<3,1,8,64> instead
    if (off || Cin != 64 || Cout != NT || H % TH != 0 || W % TW != 0 || (long long)H * W * Cin >= (1LL << 31)) return false;
    if (!glass_lds_fits(LDS_BYTES)) return false;
    // worth a persistent workgroup per CU only with several tiles each — judged at the nominal population (common.h), so that a layer runs on
    // the same kernel whatever the size of this launch
    return (long long)GLASS_NOMINAL_POP * (W / TW) * (H / TH) >= 16LL * glass_cu_count();
}

// nullptr: the layer does not qualify (the caller goes on to conv_wres / conv_tiled)
const char* launch_conv_wreg(const ConvParams& p, hipStream_t st) {
    if (!conv_wreg_supported(p.Cin, p.Cout, p.Hc, p.Wc)) return nullptr;
    if (p.Cin != 64 || p.Neff != NT || p.Cout != NT || p.up || p.y32 || !p.y || p.KS != 3 || p.stride != 1 || p.pad != 1) return nullptr;
    if (p.sn || p.sn16 || p.pre_shift || p.in_up || p.rgb_y || p.rgb_tanh_out || p.skip_x || p.post_scale16 || p.trgb_part) return nullptr;
    if (p.Hc % TH != 0 || p.Wc % TW != 0 || p.Hc != p.H || p.Wc != p.W || (p.x_bstride == 0 && p.B > 1)) return nullptr;
    if (p.xs_out && p.trgb_yout) return nullptr;
    if (p.res) return nullptr;                     // (no residual input: neither layer has one; conv_wres takes such a call)
    if (p.y_planar8 || p.x_planar32) return nullptr;   // (reads the 8-channel-plane layout, writes pixel-major)
    if (p.trgb_yout && (!p.trgb_tab || !p.trgb_b)) return nullptr;
    const int tiles_x = p.Wc / TW, tiles_y = p.Hc / TH;
    const int n_cu = glass_cu_count();
    const int PT = p.B * tiles_x * tiles_y;
    const char* name = p.trgb_yout ? "conv_wreg_kernel<true,false>" : p.xs_out ? "conv_wreg_kernel<false,true>" : "conv_wreg_kernel<false,false>";
    if (p.dry_run) return name;
    static DevOnce once;
    once.run([&] {
        (void)hipFuncSetAttribute((const void*)conv_wreg_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv_wreg_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv_wreg_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    });
    // contiguous tile ranges: the tiles of a candidate split over whole workgroups where they can (per-sample weights load once per range)
    const int per_wg = (PT + n_cu - 1) / n_cu;
    const int grid = (PT + per_wg - 1) / per_wg;
#define WG_ABL_PASS
    if (p.trgb_yout) hipLaunchKernelGGL((conv_wreg_kernel<true, false>), dim3(grid), dim3(NTHR), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_wg WG_ABL_PASS);
    else if (p.xs_out) hipLaunchKernelGGL((conv_wreg_kernel<false, true>), dim3(grid), dim3(NTHR), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_wg WG_ABL_PASS);
    else hipLaunchKernelGGL((conv_wreg_kernel<false, false>), dim3(grid), dim3(NTHR), LDS_BYTES, st, p, tiles_x, tiles_y, PT, per_wg WG_ABL_PASS);
    return name;
}
