// conv_tiled.hip — LDS-tiled MFMA implicit-GEMM convolution for gfx950 (fast path).
//
// One workgroup (4 waves) computes a TH x 32 tile of the conv grid for NT output
// channels.  K = KS*KS*Cin is walked in stages (32-channel chunk c, tap-row ty):
//   * the input patch of the chunk ((TH-1)*S+KS) x (31*S+KS) pixels x 32 ch is staged
//     ONCE per chunk in LDS (halo included -> every input element is fetched from
//     HBM/L2 once per chunk, then re-used by all KS*KS taps out of LDS); the style
//     modulation x*s[b,i] (activation-side, SURVEY 8a note 1) is applied in registers
//     on the way in, so the main loop is a plain shared-weight GEMM;
//   * the weight slice of the stage (KS taps x NT x 32 ch) is staged next to it;
//   * global loads of stage s+1 are issued before the MFMA block of stage s and
//     written to LDS after it (register-staged software pipeline, guide T14);
//   * rows of both LDS images are 80 B (64 B of data + 16 B pad) so the ds_read_b128
//     fragment reads (lane -> row, 16 B) are bank-conflict free.
// MFMA operands are swapped (A = weights, B = pixels): D[n][pixel], so every lane ends
// up with 4 consecutive output channels of ONE pixel per accumulator quad -> 8-byte
// NHWC stores and per-lane (not per-register) pixel decoding in the epilogue.
// Epilogue contract identical to conv_direct.hip (demod, noise, bias, lrelu, residual).
#include "common.h"
#include "kernels.h"
#include <stdio.h>
#include <stdlib.h>

#define ROWB 80  // bytes per LDS row (32 halfs + 8 pad)
// Staging order (r04): the thread that would stage row q of a group of eight takes row (q >> 1) + 4 * (q & 1) instead, so the 8 lanes of one
// ds_write_b128 bank group hold rows r and r + 4 (320 B apart = 16 banks: disjoint) rather than r and r + 1 (80 B apart: the second
// row's last 16 B wrap onto the first row's banks — 16 LDS cycles per instruction where 8 is the floor).  Global loads still cover the
// same sixteen 64-byte rows per wave instruction.  tests/test_host.py::test_conv_tiled_lds_rows models both orders.
__device__ __forceinline__ int tl_row(int r) { return (r & ~7) | ((r & 7) >> 1) | ((r & 1) << 2); }
// blur-down by-product: thread column index ci filters output column tl_col(ci) (2 <-> 3 and 4 <-> 5 swapped in every 8) so that each
// ds_read_b128 bank group (lanes {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}) holds columns of one parity: 160-byte strides then tile
// the 64 banks exactly (4 LDS cycles instead of 8)
__device__ __forceinline__ int tl_col(int ci) { return ci ^ (((ci >> 2) ^ (ci >> 1)) & 1); }

// dev tool (GLASS_TILED_TRACE=path): shader-clock stamps of the K stages of ONE workgroup in the middle of the grid (TR instance only)
__device__ unsigned long long* g_tiled_trace = nullptr;
#define TTRACE(ph) \
    if (TR && blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0 && s < 64) \
        g_tiled_trace[(s * 8 + (ph)) * 4 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime()

// PERSIST: the block walks several work items and prefetches the first stage of the next tile during
// the last MFMA block + store epilogue of the current one (memory-bound small-K layers); otherwise
// one work item per block (MFMA-bound layers: fewer live registers).
// TRGB (fast path, TH = 8, one n tile = all output channels): toRGB + skip-image sum of the block (stylegan2/models.py:852-870,
// 1004-1013) applied to the finished tile while it is in registers (common.h: 2 MFMAs per 32 channels per tile row) — the
// separate toRGB pass would re-read the whole map.
// SKIP (stride 2, fast path): ConvParams::skip_x / skip_w — the residual branch's 1x1 conv as extra K stages after an in-register
// activation.
// XS (3x3 stride 1, TH = 8, one n tile): ConvParams::xs_out — the blur-down of the input map from the patch already in LDS.
// SPL: the four waves form a 2 x 2 grid (row pair x n half) instead of 4 x 1: each weight fragment a wave reads feeds two tile rows
// (stride-2 convs, TH = 4: 8 fragment reads per 8 MFMAs instead of 10).
// B2: the weight slices alternate between TWO register sets and are requested two stages ahead (costs NB * 4 VGPRs).
// DEEP: the next chunk's patch is requested as soon as this chunk's patch is in LDS (three stages ahead instead of one).
template <int KS, int S, int TH, int NT, bool PERSIST = false, bool TRGB = false, bool SKIP = false, bool XS = false, bool SPL = false, bool B2 = false,
          bool DEEP = false, bool TR = false>
__global__ __launch_bounds__(256, (DEEP && NT == 64 && S == 1) ? 3 : 2) void conv_tiled_kernel(ConvParams p, int NTn, int tiles_x, int tiles_y, int PT) {
    constexpr int RW = SPL ? TH / 2 : TH / 4;  // tile rows per wave
    constexpr int NJ = SPL ? NT / 64 : NT / 32; // 32-wide n tiles per wave
    constexpr int PH = (TH - 1) * S + KS;      // patch rows
    constexpr int PW = 31 * S + KS;            // patch cols
    constexpr int NVA = PH * PW * 4;           // 16-byte vectors in the A patch
    constexpr int NA = (NVA + 255) / 256;
    constexpr int NVB = KS * NT * 4;           // 16-byte vectors in one weight stage
    constexpr int NB = (NVB + 255) / 256;
    constexpr int A_BYTES = ((PH * PW * ROWB + 15) / 16) * 16;
    constexpr int LDS_K_ = A_BYTES + KS * NT * ROWB, LDS_O_ = 4 * RW * 32 * (NJ * 64 + 16);
    constexpr int CC_OFF = LDS_K_ > LDS_O_ ? LDS_K_ : LDS_O_;   // epilogue constants sit behind both images

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + A_BYTES;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = SPL ? (wave & 1) : wave, wn = SPL ? (wave >> 1) : 0;   // this wave's row group / n group
    const int lr = lane & 31, kh = lane >> 5;
    const int part = t & 3;
    const int trow = tl_row(t >> 2);           // staging: this thread's row within each block of 64 rows (vector k: row trow + 64 k)
    const int tpi = tiles_x * tiles_y;
    const int PT8 = (PT + 7) & ~7;
    const int n_work = PT8 * NTn;              // work items = (pixel tile, n tile)

    // ---- persistent block: work items id, id + gridDim.x, ...  An item decodes to (pixel tile, n tile)
    // such that the n-tiles of one pixel tile sit on one XCD (id % 8) and share its L2. -------------
    struct Tile { int b, ty0, tx0, n0; bool valid; };
    auto decode = [&](int id) {
        Tile w;
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

    // staging state of the tile being LOADED (may be one tile ahead of the tile being computed).
    // Loads are UNCONDITIONAL (vectors outside the image / past the patch read a valid address) and masked when they are
    // written to LDS, on border tiles only; the per-vector LDS offsets are computed once per tile; every option of the input
    // transform is tested on the (uniform) launch parameters, not on per-thread pointers — round 2's instruction diet: the
    // compiler had turned this staging into ~25 exec-mask / zero-fill instructions per vector per chunk.
    int a_goff[NA];   // element offset into the image (without chunk offset)
    int a_loff[(NA + 1) / 2];   // LDS byte offsets, two 16-bit values per register
    int okm = 0;      // bit k: vector k is a pixel of the image
    bool border = false;        // uniform: the patch reaches outside the image
    const half_t* xb = p.x;
    const half_t* wb = p.w;
    const half_t* snb = nullptr;
    const half_t* psb = nullptr;
    int ld_n0 = 0;
    auto aim = [&](const Tile& w) {
        okm = 0;
#pragma unroll
        for (int k = 0; k < (NA + 1) / 2; ++k) a_loff[k] = 0;
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int pix = trow + 64 * k;
            const int pr = pix / PW, pc = pix - pr * PW;
            const int iy = w.ty0 * S - p.pad + pr, ix = w.tx0 * S - p.pad + pc;
            const bool ok = (k < NA - 1 || pix < PH * PW) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            a_goff[k] = ok ? ((iy >> p.in_up) * (p.W >> p.in_up) + (ix >> p.in_up)) * p.Cin + part * 8 : part * 8;
            okm |= (ok ? 1 : 0) << k;
            int lrow = pix;
            if (S == 2)     // de-interleave the patch columns (even | odd): a stride-2 fragment read then walks CONSECUTIVE LDS rows
                lrow = pr * PW + ((pc & 1) ? (PW + 1) / 2 + (pc >> 1) : (pc >> 1));   // (pixel order is 2-way bank-conflicted for any 16-byte-aligned pitch)
            a_loff[k >> 1] |= (lrow * ROWB + part * 16) << ((k & 1) * 16);
        }
        const int y_lo = w.ty0 * S - p.pad, x_lo = w.tx0 * S - p.pad;
        border = y_lo < 0 || x_lo < 0 || y_lo + PH > p.H || x_lo + PW > p.W;
        xb = p.x + (long long)w.b * p.x_bstride;
        wb = p.w + (long long)w.b * p.w_bstride;
        snb = p.sn16 ? p.sn16 + (long long)w.b * p.sn_stride + part * 8 : nullptr;
        psb = p.pre_shift16 ? p.pre_shift16 + (long long)w.b * p.sn_stride + part * 8 : nullptr;
        ld_n0 = w.n0;
    };
    static_assert(PH * PW * ROWB + 64 < 65536, "LDS offsets are packed in 16 bits");

    h8 ra[NA], rb[NB], rb2[B2 ? NB : 1];   // B2: two weight-stage register sets, stage s is stored from set s & 1
    h8 sh;   // style of this thread's 8 channels of the current chunk (fp16: packed multiply at staging)
    int ld_c0 = 0;   // chunk offset of the patch held in ra (pre-activation shift is fetched at store time)
#pragma unroll
    for (int j = 0; j < 8; ++j) sh[j] = (half_t)1.f;

    auto load_a = [&](int c0) {
#pragma unroll
        for (int k = 0; k < NA; ++k) ra[k] = *(const h8*)(xb + a_goff[k] + c0);
        if (p.sn16) sh = *(const h8*)(snb + c0);
        ld_c0 = c0;
    };
    auto load_b = [&](h8 (&RB)[NB], int c0, int ty) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int row = min(trow + 64 * k, KS * NT - 1);      // (KS * NT rows: whole groups of eight)
            const int tx = row / NT;
            const int n = row % NT;
            RB[k] = *(const h8*)(wb + ((long long)(ty * KS + tx) * p.Neff + ld_n0 + n) * p.Cin + c0 + part * 8);
        }
    };
    auto a_lds = [&](int k) { return As + ((a_loff[k >> 1] >> ((k & 1) * 16)) & 0xffff); };
    auto store_a = [&]() {
        const bool last_in = trow + 64 * (NA - 1) < PH * PW;   // this thread's last vector is a pixel of the patch
        if (!border && !p.sn16 && !p.pre_shift16) {      // interior tile, no input transform: registers -> LDS
#pragma unroll
            for (int k = 0; k < NA; ++k)
                if (k < NA - 1 || last_in) *(h8*)a_lds(k) = ra[k];
            return;
        }
        const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        if (p.pre_shift16) {   // BigGAN: relu(x * sh + shf), 4 x v_pk_fma_f16 + 4 x v_pk_max_f16; padding pixels stay zero
            const h8 shf = *(const h8*)(psb + ld_c0);    // fetched here to keep it out of the K-loop registers
#pragma unroll
            for (int k = 0; k < NA; ++k)
                if (k < NA - 1 || last_in)
                    *(h8*)a_lds(k) = ((okm >> k) & 1) ? __builtin_elementwise_max(ra[k] * sh + shf, zero) : zero;
        } else {
#pragma unroll
            for (int k = 0; k < NA; ++k)
                if (k < NA - 1 || last_in)
                    *(h8*)a_lds(k) = (((okm >> k) & 1) ? ra[k] : zero) * sh;    // 4 x v_pk_mul_f16 (1.0 without a style)
        }
    };
    auto store_b = [&](const h8 (&RB)[NB]) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int u = t + 256 * k;
            if (k < NB - 1 || u < NVB) *(h8*)(Bs + (trow + 64 * k) * ROWB + part * 16) = RB[k];
        }
    };

    const int n_chunks = p.Cin >> 5;
    const int n_stages = n_chunks * KS;

    int id = blockIdx.x;
    Tile cur = decode(id);
    while (id < n_work && !cur.valid) { id += gridDim.x; cur = decode(id); }   // skip padding items
    if (id >= n_work) return;
    // per-channel epilogue constants of this block's n tile (demod scale, bias, shift) are parked in LDS by the epilogue
    // prologue: one batched round trip instead of one per accumulator quad
    const bool fast = TRGB || SKIP || (!PERSIST && !p.up && (p.Cout & 7) == 0 && !p.no_tstore);
    float* Cc = (float*)(smem + CC_OFF);   // [3][NT]
    char* Tt = smem + CC_OFF + 3 * NT * 4;  // TRGB: this sample's weight tables [2][16][NT] fp16
    // DEEP prefetch (round 3): every register-staged kernel ran at ~3.3 us per K stage whatever its MFMA count (24 - 36 per
    // wave = 0.3 - 0.5 us) — one stage of prefetch distance means every stage waits out a loaded L2 / HBM round trip.  The patch
    // of the NEXT chunk is now requested right after this chunk's patch has gone to LDS (three stages ahead instead of one: its
    // registers are free from then on), and the weight slices alternate between two register sets, requested two stages ahead.
    constexpr bool deep = DEEP && !PERSIST;
    aim(cur);
    load_a(0);
    load_b(rb, 0, 0);
    if constexpr (B2) {
        if (deep && n_stages > 1) load_b(rb2, KS > 1 ? 0 : 32, KS > 1 ? 1 : 0);
    }

    for (;;) {
        // next valid work item of this block (uniform across the block)
        int nid = id + gridDim.x;
        Tile nxt = decode(nid);
        while (nid < n_work && !nxt.valid) { nid += gridDim.x; nxt = decode(nid); }
        const bool has_next = PERSIST && nid < n_work;

        f16x acc[RW][NJ];
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

        int c = 0, ty = 0;
        auto stage = [&](int s, h8 (&RBc)[NB], h8 (&RBo)[NB]) {   // RBc: the set this stage stores from; RBo: the other one
            TTRACE(0);
            __syncthreads();  // previous stage's (or previous tile's) fragment reads are done
            TTRACE(1);
            if (TR) {                         // traced instance only: split "operands landed" from "operands written to LDS"
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                TTRACE(6);
            }
            if (ty == 0) store_a();
            store_b(RBc);
            if (TR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TTRACE(2);
            __syncthreads();
            TTRACE(3);
            if (XS && ty == 0) {
                // the 8 x 32 tile of this 32-channel chunk (+ halo, zeros outside the image) sits in LDS: its 4 x 16 down-sampled
                // pixels are 16 reads + 5 packed-fp16 FIRs per thread, and the separate blur-down pass (a full read of the map) goes
                static_assert(!XS || (KS == 3 && S == 1 && TH == 8), "patch = tile + 1-pixel halo");
                int tq = threadIdx.x;
                asm volatile("" : "+v"(tq));                      // (geometry derived here, not hoisted above the MFMA blocks as loop invariants)
                const int part = tq & 3, pix = tq >> 2, ly = pix >> 4, lx = tl_col(pix & 15);
                h8 s03, s12;                                     // rows 0 + 3, rows 1 + 2 of the horizontal pass (one row live at a time)
                h8 k125, k375;
#pragma unroll
                for (int e = 0; e < 8; ++e) { k125[e] = (half_t)0.125f; k375[e] = (half_t)0.375f; }
#pragma unroll
                for (int jy = 0; jy < 4; ++jy) {
                    const char* rp = As + ((2 * ly + jy) * PW + 2 * lx) * ROWB + part * 16;
                    const h8 a0 = *(const h8*)rp, a1 = *(const h8*)(rp + ROWB), a2 = *(const h8*)(rp + 2 * ROWB), a3 = *(const h8*)(rp + 3 * ROWB);
                    // (explicit FMA forms — the ones -ffp-contract chose in rounds 2-5: left to the compiler the fused product changes from
                    // build to build, one fp16 ulp of the by-product, enough to move the D-logit regression guards)
                    const h8 hr = __builtin_elementwise_fma(a1 + a2, k375, (a0 + a3) * k125);
                    if (jy == 0) s03 = hr;
                    else if (jy == 1) s12 = hr;
                    else if (jy == 2) s12 = s12 + hr;
                    else s03 = s03 + hr;
                    __builtin_amdgcn_sched_barrier(0);
                }
                const h8 o = __builtin_elementwise_fma(s12, k375, s03 * k125);
                *(h8*)(p.xs_out + (((long long)cur.b * (p.H >> 1) + (cur.ty0 >> 1) + ly) * (p.W >> 1) + (cur.tx0 >> 1) + lx) * p.Cin + c * 32 + part * 8) = o;
            }
            int nc = c, nty = ty + 1;
            if (nty == KS) { nty = 0; nc = c + 1; }
            if (deep) {
                if (ty == 0 && c + 1 < n_chunks) load_a((c + 1) * 32);          // this chunk's patch is in LDS: its registers carry the next one
                if (B2) {
                    if (s + 2 < n_stages) {                                      // the set just stored takes stage s + 2
                        const int t2 = ty + 2;
                        load_b(RBc, (c + t2 / KS) * 32, t2 % KS);
                    }
                } else if (s + 1 < n_stages) {
                    load_b(RBo, nc * 32, nty);
                }
            } else if (s + 1 < n_stages) {  // prefetch the next stage while this one computes
                if (nty == 0) load_a(nc * 32);
                load_b(RBo, nc * 32, nty);
            } else if (has_next) {   // last stage: prefetch stage 0 of the NEXT tile; it stays in flight
                aim(nxt);            // through this tile's MFMA block and store epilogue
                load_a(0);
                load_b(RBo, 0, 0);
            }
            TTRACE(4);
            // ---- MFMA block: KS taps x 2 k16 steps x RW x NJ --------------------------------
#pragma unroll
            for (int tx = 0; tx < KS; ++tx) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    h8 wf[NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        wf[j] = *(const h8*)(Bs + (tx * NT + (wn * NJ + j) * 32 + lr) * ROWB + kk * 32 + kh * 16);
#pragma unroll
                    for (int i = 0; i < RW; ++i) {
                        const int prow = (wr * RW + i) * S + ty;
                        const int pcol = S == 2 ? ((tx & 1) ? (PW + 1) / 2 + lr + (tx >> 1) : lr + (tx >> 1)) : lr + tx;
                        const h8 xf = *(const h8*)(As + (prow * PW + pcol) * ROWB + kk * 32 + kh * 16);
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(wf[j], xf, acc[i][j]);
                    }
                }
            }
            TTRACE(5);
            c = nc;
            ty = nty;
        };
        if constexpr (B2) {
            for (int s = 0; s < n_stages; s += 2) {
                stage(s, rb, rb2);
                if (s + 1 < n_stages) stage(s + 1, rb2, rb);
            }
        } else {
            for (int s = 0; s < n_stages; ++s) stage(s, rb, rb);          // (single weight set)
        }

        const int b = cur.b;
        if (SKIP) {
            static_assert(!SKIP || (TH * 32 * 4 <= 2 * 256 && NT * 4 <= NB * 256), "skip stage operands fit the staging registers");
            // operands of skip stage 0 first (their latency hides under the activation math)
            const int sv0 = t >> 2;                         // vector k: pixel (sv0 >> 5) + 2k of the tile... TH*32 px, 4 parts
            auto load_s = [&](int c0) {
#pragma unroll
                for (int k = 0; k < TH / 2; ++k) {
                    const int px = sv0 + 64 * k, row = px >> 5, col = px & 31;
                    ra[k] = *(const h8*)(p.skip_x + (((long long)b * p.Ho + cur.ty0 + row) * p.Wo + cur.tx0 + col) * p.Cin + c0 + part * 8);
                }
#pragma unroll
                for (int k = 0; k < NT / 64; ++k)
                    rb[k] = *(const h8*)(p.skip_w + (long long)(cur.n0 + sv0 + 64 * k) * p.Cin + c0 + part * 8);
            };
            load_s(0);
            // activation in the accumulators: lrelu(acc + bias) * sqrt2 (the D path has no demodulation, noise or shift)
            f4 bq[NJ][4];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bq[j][g] = f4{0.f, 0.f, 0.f, 0.f};
                    if (p.bias) bq[j][g] = *(const f4*)(p.bias + cur.n0 + (wn * NJ + j) * 32 + 8 * g + 4 * kh);
                }
            const ActK aks = act_consts(p.act == 1 ? 1 : 0, 1.f);
#pragma unroll
            for (int i = 0; i < RW; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f4 a = {acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]};
                        const f4 v = act_apply(a + bq[j][g], aks);
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[i][j][g * 4 + q] = v[q];
                    }
            for (int cs = 0; cs < n_chunks; ++cs) {
                __syncthreads();                             // previous stage's fragment reads are done
#pragma unroll
                for (int k = 0; k < TH / 2; ++k) *(h8*)(As + (sv0 + 64 * k) * ROWB + part * 16) = ra[k];
#pragma unroll
                for (int k = 0; k < NT / 64; ++k) *(h8*)(Bs + (sv0 + 64 * k) * ROWB + part * 16) = rb[k];
                __syncthreads();
                if (cs + 1 < n_chunks) load_s((cs + 1) * 32);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    h8 wf[NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) wf[j] = *(const h8*)(Bs + ((wn * NJ + j) * 32 + lr) * ROWB + kk * 32 + kh * 16);
#pragma unroll
                    for (int i = 0; i < RW; ++i) {
                        const h8 xf = *(const h8*)(As + ((wr * RW + i) * 32 + lr) * ROWB + kk * 32 + kh * 16);
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(wf[j], xf, acc[i][j]);
                    }
                }
            }
        }

        // ---- epilogue: lane = one pixel (col lr of tile row), 4 consecutive channels per quad ------------------
        if (fast) {
            // Fast path.  Nothing here waits on a global load it has just issued: the per-channel constants were staged
            // in LDS at kernel start, the noise values and residual quads of a tile row are fetched as one batch, and
            // the finished fp16 quads go through a per-wave LDS image and leave as 16-byte vectors in row order (one
            // store instruction = 1 KB of whole 64-byte lines instead of 64 scattered 8-byte pieces).
            constexpr int OROW = NJ * 64 + 16;            // bytes per staged pixel: this wave's channels (+16: bank spread)
            char* Os = smem + wave * (RW * 32 * OROW);    // this wave's RW tile rows
            const int oy0 = cur.ty0 + wr * RW, ox = cur.tx0 + lr;
            // ONE batch of global loads: this thread's per-channel constants (threads < NT) and its pixels' noise values
            float c_d = 1.f, c_b = 0.f, c_s = 0.f;
            if (t < NT && !SKIP) {                         // (SKIP: bias and activation were applied before the skip stages)
                const int o = cur.n0 + t;
                if (p.dscale) c_d = p.dscale[(long long)b * p.ds_stride + o];
                if (p.bias) c_b = p.bias[o];
                if (p.shift) c_s = p.shift[(long long)b * p.ds_stride + o];
            }
            float nzr[RW];
#pragma unroll
            for (int i = 0; i < RW; ++i) {
                nzr[i] = 0.f;
                if (p.noise) nzr[i] = p.noise_strength * p.noise[((long long)(b / p.batch_size) * p.Ho + oy0 + i) * p.Wo + ox];
            }
            float ytap[3][4];                             // TRGB: skip-image taps of this lane's pixel (tile row kh, column lr)
            if (TRGB) {
                static_assert(!TRGB || RW == 2, "lane half kh owns tile row kh of the wave");
#pragma unroll
                for (int u = 0; u < (4 * NT + 255) / 256; ++u)
                    if (t + 256 * u < 4 * NT)
                        *(h8*)(Tt + (t + 256 * u) * 16) = *(const h8*)(p.trgb_tab + (long long)b * 32 * NT + (t + 256 * u) * 8);
                if (p.trgb_yprev) {
                    const int my = (oy0 + kh) >> 1, mx = ox >> 1, h2 = p.Ho >> 1, w2 = p.Wo >> 1;
                    const float* yp = p.trgb_yprev + (long long)b * 3 * h2 * w2;
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            ytap[c][q] = yp[(c * h2 + max(my - 1 + (q >> 1), 0)) * w2 + max(mx - 1 + (q & 1), 0)];
                }
            }
            if (t < NT) { Cc[t] = c_d; Cc[NT + t] = c_b + c_s; }   // Cc sits behind As / Bs / Os
            const ActK ak = act_consts(SKIP ? 0 : p.act, p.out_scale);
            __syncthreads();                              // every wave is done reading As / Bs; constants visible
            f16x rgb;
#pragma unroll
            for (int q = 0; q < 16; ++q) rgb[q] = 0.f;
            const int rcs = p.res_cs ? p.res_cs : p.Cout;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                h4 va[RW][4];                             // TRGB: the finished quads of this slice, the 1x1 conv's B operand
                h4 rq[4][RW];                             // residual quads of this 32-channel slice: one batch of loads
                if (p.res) {
#pragma unroll
                    for (int i = 0; i < RW; ++i) {
                        const int oy = oy0 + i;
                        const half_t* rp = p.res + (p.res_up ? (((long long)b * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * rcs
                                                             : (((long long)b * p.Ho + oy) * p.Wo + ox) * rcs) + cur.n0 + (wn * NJ + j) * 32 + 4 * kh;
#pragma unroll
                        for (int g = 0; g < 4; ++g) rq[g][i] = *(const h4*)(rp + 8 * g);
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nw = j * 32 + 8 * g + 4 * kh;   // first of 4 consecutive channels, local to the wave's channels
                    const int nl = wn * NJ * 32 + nw;          // ... local to the n tile
                    const f4 d = *(const f4*)(Cc + nl), bb = *(const f4*)(Cc + NT + nl);      // bb = bias + shift
#pragma unroll
                    for (int i = 0; i < RW; ++i) {
                        const f4 a = {acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]};
                        f4 v = act_apply(a * d + bb + nzr[i], ak);           // (SKIP: activation was applied before the skip stages: ak = scale only)
                        if (p.res) v += f4{(float)rq[g][i][0], (float)rq[g][i][1], (float)rq[g][i][2], (float)rq[g][i][3]} * p.out_scale;
                        if (NT == 32 && !TRGB && !SKIP && !XS && p.rgb_tanh_out) {   // channels 0..2 of this pixel: the lanes of half 0, quad 0
                            if (nl == 0) {
                                const long long hw = (long long)p.Ho * p.Wo;
                                float* yo = p.rgb_tanh_out + (long long)b * 3 * hw + (long long)(oy0 + i) * p.Wo + ox;
#pragma unroll
                                for (int cc = 0; cc < 3; ++cc) yo[cc * hw] = tanhf(v[cc]);
                            }
                            continue;
                        }
                        h4 out;
#pragma unroll
                        for (int q = 0; q < 4; ++q) out[q] = (half_t)v[q];
                        *(h4*)(Os + (i * 32 + lr) * OROW + nw * 2) = out;
                        if (TRGB) va[i][g] = out;
                    }
                    __builtin_amdgcn_sched_barrier(0);    // keep one quad's constants live at a time (no hoisting of all
                }                                         // 16 quads' LDS reads to the top: that spills)
                if (TRGB) {
#pragma unroll
                    for (int i = 0; i < RW; ++i)
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) {
                            const h8 wt = *(const h8*)(Tt + (i * 16 + (lr & 15)) * (NT * 2) + ((j * 2 + gp) * 2 + kh) * 16);
                            rgb = mfma32(wt, __builtin_shufflevector(va[i][2 * gp], va[i][2 * gp + 1], 0, 1, 2, 3, 4, 5, 6, 7), rgb);
                        }
                }
            }
            if (TRGB) {
                const long long hw = (long long)p.Ho * p.Wo;
                float* yo = p.trgb_yout + (long long)b * 3 * hw + (long long)(oy0 + kh) * p.Wo + ox;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float r = p.trgb_b[c] + (rgb[c] + rgb[4 + c] * (1.f / 2048.f));
                    if (p.trgb_yprev) r += trgb_skip(ytap[c], oy0 + kh, ox);
                    yo[c * hw] = r;
                }
            }
            __builtin_amdgcn_wave_barrier();              // LDS is in-order per wave: only pin the compiler's order
            if (!(NT == 32 && !TRGB && !SKIP && !XS && p.rgb_tanh_out)) {
#pragma unroll
            for (int i = 0; i < RW; ++i) {
                half_t* yrow = p.y + (((long long)b * p.Ho + oy0 + i) * p.Wo + cur.tx0) * p.Cout + cur.n0 + wn * NJ * 32;
#pragma unroll
                for (int k = 0; k < NJ * 2; ++k) {
                    const int v = lane + 64 * k;
                    const int pix = v / (NJ * 4), chv = v % (NJ * 4);
                    *(h8*)(yrow + (long long)pix * p.Cout + chv * 8) = *(const h8*)(Os + (i * 32 + pix) * OROW + chv * 16);
                }
            }
            }
        } else {
            // generic path (folded up-conv with depth-to-space, odd channel counts, persistent variant)
            static_assert(!SPL || SKIP, "the 2 x 2 wave grid exists for the fast-path stride-2 instances only");
#pragma unroll
            for (int i = 0; i < RW; ++i) {
                const int oy = cur.ty0 + wave * RW + i, ox = cur.tx0 + lr;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nb = cur.n0 + j * 32 + 8 * g + 4 * kh;  // first of 4 consecutive n
                        int o = nb, py = oy, px = ox;
                        if (p.up) {
                            const int ph = nb / p.Cout;
                            o = nb - ph * p.Cout;
                            py = 2 * oy + (ph >> 1);
                            px = 2 * ox + (ph & 1);
                        }
                        float v[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = acc[i][j][g * 4 + q];
                        if (p.dscale) {
                            const f4 d = *(const f4*)(p.dscale + (long long)b * p.ds_stride + o);
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] *= d[q];
                        }
                        if (p.noise) {
                            const float nz = p.noise_strength * p.noise[((long long)(b / p.batch_size) * p.Ho + py) * p.Wo + px];
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] += nz;
                        }
                        if (p.bias) {
                            const f4 bb = *(const f4*)(p.bias + o);
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] += bb[q];
                        }
                        if (p.shift) {
                            const f4 sh4 = *(const f4*)(p.shift + (long long)b * p.ds_stride + o);
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] += sh4[q];
                        }
                        if (p.act == 1) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] = lrelu_sqrt2(v[q]);
                        } else if (p.act == 2) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                        }
                        const long long oidx = (((long long)b * p.Ho + py) * p.Wo + px) * p.Cout + o;
                        if (p.res) {
                            const int rcs = p.res_cs ? p.res_cs : p.Cout;
                            const long long ridx = p.res_up ? (((long long)b * (p.Ho >> 1) + (py >> 1)) * (p.Wo >> 1) + (px >> 1)) * rcs + o
                                                            : (((long long)b * p.Ho + py) * p.Wo + px) * rcs + o;
                            const h4 r = *(const h4*)(p.res + ridx);
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] += (float)r[q];
                        }
                        h4 out;
#pragma unroll
                        for (int q = 0; q < 4; ++q) out[q] = (half_t)(v[q] * p.out_scale);
                        *(h4*)(p.y + oidx) = out;
                    }
                }
            }
        }
        if (!has_next) break;
        id = nid;
        cur = nxt;
    }
}

template <int KS, int S, int TH, int NT, bool PERSIST = false, bool TRGB = false, bool SKIP = false, bool XS = false, bool SPL = false, bool B2 = false, bool DEEP = false, bool TR = false>
static const char* launch_inst(const ConvParams& p, hipStream_t st, const char* name) {
    constexpr int PH = (TH - 1) * S + KS, PW = 31 * S + KS;
    constexpr int A_BYTES = ((PH * PW * ROWB + 15) / 16) * 16;
    constexpr int LDS_K = A_BYTES + KS * NT * ROWB, LDS_O = 4 * (SPL ? TH / 2 : TH / 4) * 32 * ((SPL ? NT / 64 : NT / 32) * 64 + 16);   // K-loop images | epilogue image
    constexpr int LDS = (LDS_K > LDS_O ? LDS_K : LDS_O) + 3 * NT * 4 + (TRGB ? 64 * NT : 0);
    if (!glass_lds_fits(LDS)) return nullptr;
    static DevOnce once;                       // (one per template instance)
    if (LDS > 64 * 1024)
        once.run([&] {
            (void)hipFuncSetAttribute((const void*)conv_tiled_kernel<KS, S, TH, NT, PERSIST, TRGB, SKIP, XS, SPL, B2, DEEP, TR>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        });
    const int tiles_x = p.Wc / 32, tiles_y = p.Hc / TH;
    const int PT = p.B * tiles_x * tiles_y;
    const int NTn = p.Neff / NT;
    const int PT8 = (PT + 7) / 8 * 8;
    // persistent grid: as many workgroups as are resident at once (256 CUs x blocks/CU), a multiple of 8
    // so a block keeps its XCD; each block walks its work items with cross-tile prefetch
    int resident = 1 << 30;
    if (PERSIST) {
        static int per_cu_cache = 0;           // occupancy is a property of the kernel + architecture; the CU count is per device
        if (!per_cu_cache) {
            int per_cu = 1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv_tiled_kernel<KS, S, TH, NT, PERSIST, TRGB, SKIP, XS, SPL, B2, DEEP, TR>, 256, LDS);
            per_cu_cache = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
        }
        resident = glass_cu_count() * per_cu_cache;
        resident -= resident % 8;
    }
    const int n_work = PT8 * NTn;
    const int grid = n_work < resident ? n_work : resident;
    if (p.dry_run) return name;
    hipLaunchKernelGGL((conv_tiled_kernel<KS, S, TH, NT, PERSIST, TRGB, SKIP, XS, SPL, B2, DEEP, TR>), dim3(grid), dim3(256), LDS, st, p, NTn, tiles_x, tiles_y, PT);
    return name;
}

const char* launch_conv_tiled(const ConvParams& p0, hipStream_t st) {
    ConvParams p = p0;
    if (p.x_planar8 || p.y_planar8 || p.x_planar32) return nullptr;   // chunk-planar maps (common.h): not implemented here
    static const bool no_ts = glass_knob("GLASS_NO_TSTORE") != nullptr;   // experiment knob
    if (no_ts) p.no_tstore = 1;
    // A/B knob GLASS_DEEP: 0 = round 2's one-stage prefetch distance, 1 = patch three stages ahead, 2 = + two weight register sets (stride 2)
    static const int deep_on = glass_knob("GLASS_DEEP") ? atoi(glass_knob("GLASS_DEEP")) : 1;
    if (p.rgb_tanh_out) {   // planar tanh(channels 0..2) from the accumulators: the one-n-tile 3x3 instance's fast path only
        if (p.y32 || p.trgb_yout || p.xs_out || p.up || p.KS != 3 || p.stride != 1 || p.pad != 1 || p.no_tstore || p.Neff != 32 || p.Cout != 32 ||
            p.Hc % 8 != 0 || p.Wc % 32 != 0 || p.Cin % 32 != 0 || (p.sn && !p.sn16) || (p.pre_shift && !p.pre_shift16) || p.res || p.noise ||
            (p.x_bstride == 0 && p.B > 1) || (long long)p.H * p.W * p.Cin >= (1LL << 31))
            return nullptr;
        return launch_inst<3, 1, 8, 32>(p, st, "conv_tiled_kernel<3,1,8,32>");
    }
    if (p.y32 || !p.y) return nullptr;
    if (p.trgb_yout) {   // fused toRGB: only where one workgroup holds every output channel of its pixels
        if (!p.trgb_tab || !p.trgb_b || p.up || p.KS != 3 || p.stride != 1 || p.pad != 1 || p.no_tstore || p.Neff != 64 || p.Cout != 64 ||
            p.Hc % 8 != 0 || p.Wc % 32 != 0 || p.Cin % 32 != 0 || (p.sn && !p.sn16) || (p.pre_shift && !p.pre_shift16) ||
            (p.x_bstride == 0 && p.B > 1) || (long long)p.H * p.W * p.Cin >= (1LL << 31))
            return nullptr;
        if (deep_on) return launch_inst<3, 1, 8, 64, false, true, false, false, false, false, true>(p, st, "conv_tiled_kernel<3,1,8,64,torgb,deep>");
        return launch_inst<3, 1, 8, 64, false, true>(p, st, "conv_tiled_kernel<3,1,8,64,torgb>");
    }
    if (p.xs_out) {   // blur-down of the input as a by-product: un-transformed input, every chunk staged exactly once per pixel tile
        if (p.KS != 3 || p.stride != 1 || p.pad != 1 || p.sn || p.pre_shift || p.in_up || p.up || p.Neff != 64 || p.Cout != 64 || p.no_tstore ||
            p.Hc % 8 != 0 || p.Wc % 32 != 0 || p.Cin % 32 != 0 || p.trgb_yout || (p.x_bstride == 0 && p.B > 1) ||
            (long long)p.H * p.W * p.Cin >= (1LL << 31))
            return nullptr;
        if (deep_on) return launch_inst<3, 1, 8, 64, false, false, false, true, false, false, true>(p, st, "conv_tiled_kernel<3,1,8,64,xs,deep>");
        return launch_inst<3, 1, 8, 64, false, false, false, true>(p, st, "conv_tiled_kernel<3,1,8,64,xs>");
    }
    if ((p.sn && !p.sn16) || (p.pre_shift && !p.pre_shift16)) return nullptr;   // fp16 tables not provided: direct path
    if (p.x_bstride == 0 && p.B > 1) return nullptr;  // broadcast input (4x4 const): direct path
    if (p.Cin % 32 != 0 || p.Wc % 32 != 0 || p.Cout % 4 != 0) return nullptr;
    if ((long long)p.H * p.W * p.Cin >= (1LL << 31)) return nullptr;
    const int KS = p.KS, S = p.stride;
    if (KS == 3 && S == 1 && p.pad == 1) {
        static const int th4 = glass_knob("GLASS_TH4") ? atoi(glass_knob("GLASS_TH4")) : 0;   // experiment knob
        static const bool nt64 = glass_knob("GLASS_NT64") != nullptr;   // experiment knob
        if (!nt64 && p.Neff % 128 == 0 && p.Hc % 8 == 0) return launch_inst<3, 1, 8, 128>(p, st, "conv_tiled_kernel<3,1,8,128>");
        if ((th4 & 2) && p.Neff % 64 == 0 && p.Hc % 4 == 0) return launch_inst<3, 1, 4, 64>(p, st, "conv_tiled_kernel<3,1,4,64>");
        static const bool persist = glass_knob("GLASS_PERSIST") != nullptr;   // measured slower (more live registers -> lower occupancy): off
        if (persist && p.Neff % 64 == 0 && p.Hc % 8 == 0) return launch_inst<3, 1, 8, 64, true>(p, st, "conv_tiled_kernel<3,1,8,64,persist>");
        if (p.Neff % 64 == 0 && p.Hc % 8 == 0) return launch_inst<3, 1, 8, 64>(p, st, "conv_tiled_kernel<3,1,8,64>");
        // memory-bound, tiny K: small tiles = more workgroups per CU = more bytes in flight
        if ((th4 & 1) && p.Neff % 32 == 0 && p.Hc % 4 == 0) return launch_inst<3, 1, 4, 32>(p, st, "conv_tiled_kernel<3,1,4,32>");
        if (persist && p.Neff % 32 == 0 && p.Hc % 8 == 0) return launch_inst<3, 1, 8, 32, true>(p, st, "conv_tiled_kernel<3,1,8,32,persist>");
        if (p.Neff % 32 == 0 && p.Hc % 8 == 0) return launch_inst<3, 1, 8, 32>(p, st, "conv_tiled_kernel<3,1,8,32>");
        return nullptr;
    }
    if (p.skip_x) {   // skip branch as extra K stages: stride-2 fast path without demodulation / noise / shift / residual
        if (KS != 3 || S != 2 || p.pad != 0 || !p.skip_w || p.res || p.dscale || p.noise || p.shift || p.sn || p.pre_shift || p.up ||
            (p.Cout & 7) || p.no_tstore || p.Hc % 4 != 0)
            return nullptr;
#ifdef GLASS_DEV_TRACE
        static const bool tiled_trace = getenv("GLASS_TILED_TRACE") != nullptr;
#else
        constexpr bool tiled_trace = false;
#endif
        if (!tiled_trace)
            if (const char* k = launch_conv_s2(p, st)) return k;        // LDS-DMA ring kernel where its geometry applies
        static const bool spl = glass_knob("GLASS_NO_S2_SPLIT") == nullptr;  // 2 x 2 wave grid (A/B knob: GLASS_NO_S2_SPLIT=1 -> 4 x 1; measured -4.4 % on the four stride-2 layers)
        if (spl && deep_on == 2 && p.Neff % 128 == 0) return launch_inst<3, 2, 4, 128, false, false, true, false, true, true, true>(p, st, "conv_tiled_kernel<3,2,4,128,skip,spl,b2,deep>");
#ifdef GLASS_DEV_TRACE      // dev build (make TRACE=1): traced instance, stamps of one mid-grid workgroup to a file; synchronises, single engine only
        if (const char* tp = getenv("GLASS_TILED_TRACE")) {      // dev tool: traced instance, stamps of one mid-grid workgroup to a file
            if (spl && p.Neff % 128 == 0 && !p.dry_run) {
                unsigned long long* dtr = nullptr;
                (void)hipMalloc(&dtr, 64 * 8 * 4 * sizeof(unsigned long long));
                (void)hipMemset(dtr, 0, 64 * 8 * 4 * sizeof(unsigned long long));
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tiled_trace), &dtr, sizeof dtr);
                const char* nm = launch_inst<3, 2, 4, 128, false, false, true, false, true, false, true, true>(p, st, "conv_tiled_kernel<3,2,4,128,skip,spl,deep,trace>");
                static unsigned long long hb[64 * 8 * 4];
                (void)hipStreamSynchronize(st);
                (void)hipMemcpy(hb, dtr, sizeof hb, hipMemcpyDeviceToHost);
                (void)hipFree(dtr);
                if (FILE* f = fopen(tp, "a")) {
                    fprintf(f, "# %s Cin=%d Cout=%d Hc=%d: stage phase t[wave0..3]; phases 0 top, 1 after barrier, 2 operands stored, 3 after barrier, 4 loads issued, 5 MFMAs done, 6 operands landed (between 1 and 2)\n", nm, p.Cin, p.Cout, p.Hc);
                    for (int i = 0; i < 64; ++i)
                        for (int ph = 0; ph < 7; ++ph) {
                            fprintf(f, "%d %d", i, ph);
                            for (int w = 0; w < 4; ++w) fprintf(f, " %llu", hb[(i * 8 + ph) * 4 + w] ? hb[(i * 8 + ph) * 4 + w] - hb[0] : 0ULL);
                            fprintf(f, "\n");
                        }
                    fclose(f);
                }
                return nm;
            }
        }
#endif
        if (spl && deep_on && p.Neff % 128 == 0) return launch_inst<3, 2, 4, 128, false, false, true, false, true, false, true>(p, st, "conv_tiled_kernel<3,2,4,128,skip,spl,deep>");
        if (spl && p.Neff % 128 == 0) return launch_inst<3, 2, 4, 128, false, false, true, false, true>(p, st, "conv_tiled_kernel<3,2,4,128,skip,spl>");
        if (p.Neff % 128 == 0) return launch_inst<3, 2, 4, 128, false, false, true>(p, st, "conv_tiled_kernel<3,2,4,128,skip>");
        if (p.Neff % 64 == 0) return launch_inst<3, 2, 4, 64, false, false, true>(p, st, "conv_tiled_kernel<3,2,4,64,skip>");
        return nullptr;
    }
    if (KS == 3 && S == 2 && p.pad == 0) {
        if (p.Neff % 128 == 0 && p.Hc % 4 == 0) return launch_inst<3, 2, 4, 128>(p, st, "conv_tiled_kernel<3,2,4,128>");
        if (p.Neff % 64 == 0 && p.Hc % 4 == 0) return launch_inst<3, 2, 4, 64>(p, st, "conv_tiled_kernel<3,2,4,64>");
        if (p.Neff % 32 == 0 && p.Hc % 4 == 0) return launch_inst<3, 2, 4, 32>(p, st, "conv_tiled_kernel<3,2,4,32>");
        return nullptr;
    }
    if (KS == 1 && S == 1 && p.pad == 0) {
        if (p.Neff % 128 == 0 && p.Hc % 8 == 0) return launch_inst<1, 1, 8, 128>(p, st, "conv_tiled_kernel<1,1,8,128>");
        if (p.Neff % 64 == 0 && p.Hc % 8 == 0) return launch_inst<1, 1, 8, 64>(p, st, "conv_tiled_kernel<1,1,8,64>");
        if (p.Neff % 32 == 0 && p.Hc % 8 == 0) return launch_inst<1, 1, 8, 32>(p, st, "conv_tiled_kernel<1,1,8,32>");
        return nullptr;
    }
    return nullptr;
}
