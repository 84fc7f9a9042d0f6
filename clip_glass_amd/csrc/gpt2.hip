// gpt2.hip — GPT-2 greedy decode kernels for the img2txt path (config C5; reference
// gpt2/model.py:126-211, gpt2/sample.py:21-36, models.py:45-62).
//
// Everything here is fp32: greedy decoding is an arg-max over 50 257 logits, i.e. an INDEX
// result of floating-point work — a half-precision trunk flips tokens whenever the top-2
// margin is below ~1e-3 of the logit scale, and one flipped token changes the whole
// continuation.  The GEMMs therefore run on the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak = the fp32 vector rate, bit-equivalent to an fmaf
// chain), LDS-tiled 128 x 128 x 32, so token parity with the fp32 reference only depends on
// summation order.
#include "common.h"
#include "kernels.h"

// ---- token + position embedding (model.py:163-171): x[r] = wte[tok[r]] + wpe[pos0 + r % L] -------------
__global__ void gpt2_embed_kernel(const int* tok, const float* wte, const float* wpe, int L, int pos0, int D, float* x) {
    const int row = blockIdx.x;
    const float* te = wte + (long long)tok[row] * D;
    const float* pe = wpe + (long long)(pos0 + row % L) * D;
    for (int i = threadIdx.x; i < D; i += blockDim.x) x[(long long)row * D + i] = te[i] + pe[i];
}
// single-token decode step inside a hipGraph: the step's state lives in device memory (state[0] = past length, state[1] = step
// index), so ONE captured graph serves every step.  Tokens of the previous step are read from gen[(step - 1) * P + row].
// stats != nullptr (fused step, D <= 1024): also the row's LayerNorm statistics {mean, rstd} for the first layer's fused LayerNorm — the
// arithmetic of gpt2_finalize_kernel with 256 threads (two-pass mean / variance, lanes xor tree then waves 0..3).
__global__ __launch_bounds__(256) void gpt2_embed_step_kernel(const int* gen, const int* state, int P, const float* wte, const float* wpe, int D,
                                                              float* x, float* stats, int partial_fmt) {
    __shared__ float red[4];
    const int row = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int past = state[0], step = state[1];
    const float* te = wte + (long long)gen[(long long)(step - 1) * P + row] * D;
    const float* pe = wpe + (long long)past * D;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (!stats) {
        for (int i = t; i < D; i += 256) x[(long long)row * D + i] = te[i] + pe[i];
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = t + 256 * k;
        if (i < D) { v[k] = te[i] + pe[i]; x[(long long)row * D + i] = v[k]; }
    }
    auto block_sum = [&](float q) {
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        __syncthreads();
        if (lane == 0) red[wave] = q;
        __syncthreads();
        return ((red[0] + red[1]) + red[2]) + red[3];
    };
    const float mean = block_sum((v[0] + v[1]) + (v[2] + v[3])) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float d = t + 256 * k < D ? v[k] - mean : 0.f; q += d * d; }
    const float m2 = block_sum(q), var = m2 / (float)D;
    if (t == 0) { stats[2 * row] = mean; stats[2 * row + 1] = partial_fmt ? m2 : rsqrtf(var + 1e-5f); }      // partial_fmt: ONE (mean, M2) partial over the row
}
void launch_gpt2_embed_step(const int* gen, const int* state, int P, const float* wte, const float* wpe, int D, float* x, hipStream_t st,
                            float* stats, bool partial_fmt) {
    hipLaunchKernelGGL(gpt2_embed_step_kernel, dim3(P), dim3(256), 0, st, gen, state, P, wte, wpe, D, x, D <= 1024 ? stats : nullptr, partial_fmt ? 1 : 0);
}
__global__ void gpt2_advance_kernel(int* state) {
    state[0] += 1;
    state[1] += 1;
}
void launch_gpt2_advance(int* state, hipStream_t st) { hipLaunchKernelGGL(gpt2_advance_kernel, dim3(1), dim3(1), 0, st, state); }

void launch_gpt2_embed(const int* tok, const float* wte, const float* wpe, int rows, int L, int pos0, int D, float* x,
                       hipStream_t st) {
    hipLaunchKernelGGL(gpt2_embed_kernel, dim3(rows), dim3(128), 0, st, tok, wte, wpe, L, pos0, D, x);
}

// ---- fp32 GEMM: out[M][N] = A[M][K] @ W[N][K]^T (+bias) with epilogue ------------------------------------
// mode 0: out = v ; 1: out = gelu_tanh(v) (model.py:12-13) ; 2: out += v (residual, in place)
#define F32_BK 32
#define F32_LD 33   // +1 float pad: conflict-free ds_read_b32 fragment reads
// BM x BN block tile, 4 waves as 2 x 2, each wave (BM/2) x (BN/2) = TI x TJ tiles of 32 x 32.
// <128,128> for the prefill (M = P * 23), <64,64> for the single-token decode steps (M = P): more, smaller
// blocks — those GEMMs stream the weights once and are latency/launch-bound, not flop-bound.
// blockIdx.z = K slice (split-K): with more than one slice the block writes its raw partial sums to part[z][M][N]
// (mode / bias are applied by splitk_reduce_kernel, which adds the slices in a fixed order: deterministic).
template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* A, const float* W, const float* bias, float* out,
                                                       int M, int N, int K, int lda, int ldo, int mode, float* part) {
    constexpr int TI = BM / 64, TJ = BN / 64;
    __shared__ float As[BM][F32_LD];
    __shared__ float Ws[BN][F32_LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, kh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    f16x acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const int kc = gridDim.z > 1 ? ((K / F32_BK + gridDim.z - 1) / gridDim.z) * F32_BK : K;    // K slice of this block
    const int k_lo = blockIdx.z * kc, k_hi = min(K, k_lo + kc);
    // the next K step's operand vectors are requested before this step's MFMAs (register prefetch): a step used to start with a full
    // L2 / HBM round trip in front of 2-4 us of fp32 MFMAs
    constexpr int NVA = BM * 8 / 256, NVW = BN * 8 / 256;
    f4 ra[NVA], rw[NVW];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int e = i * 256 + t, row = e >> 3, c4 = (e & 7) * 4;
            ra[i] = f4{0.f, 0.f, 0.f, 0.f};
            if (m0 + row < M && k0 + c4 < k_hi) ra[i] = *(const f4*)(A + (long long)(m0 + row) * lda + k0 + c4);
        }
#pragma unroll
        for (int i = 0; i < NVW; ++i) {
            const int e = i * 256 + t, row = e >> 3, c4 = (e & 7) * 4;
            rw[i] = f4{0.f, 0.f, 0.f, 0.f};
            if (n0 + row < N && k0 + c4 < k_hi) rw[i] = *(const f4*)(W + (long long)(n0 + row) * K + k0 + c4);
        }
    };
    load(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += F32_BK) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int e = i * 256 + t, row = e >> 3, c4 = (e & 7) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) As[row][c4 + j] = ra[i][j];
        }
#pragma unroll
        for (int i = 0; i < NVW; ++i) {
            const int e = i * 256 + t, row = e >> 3, c4 = (e & 7) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) Ws[row][c4 + j] = rw[i][j];
        }
        __syncthreads();
        if (k0 + F32_BK < k_hi) load(k0 + F32_BK);
#pragma unroll
        for (int kk = 0; kk < F32_BK; kk += 2) {
            float wf[TJ], xf[TI];
#pragma unroll
            for (int j = 0; j < TJ; ++j) wf[j] = Ws[wn * (BN / 2) + j * 32 + lr][kk + kh];
#pragma unroll
            for (int i = 0; i < TI; ++i) xf[i] = As[wm * (BM / 2) + i * 32 + lr][kk + kh];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j], xf[i], acc[i][j], 0, 0, 0);  // D[n][m]
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int m = m0 + wm * (BM / 2) + i * 32 + lr;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int n = n0 + wn * (BN / 2) + j * 32 + mfma32_row(reg, lane);
                if (n >= N) continue;
                if (gridDim.z > 1) {
                    part[((long long)blockIdx.z * M + m) * N + n] = acc[i][j][reg];
                    continue;
                }
                float v = acc[i][j][reg] + (bias ? bias[n] : 0.f);
                const long long oi = (long long)m * ldo + n;
                if (mode == 1) {
                    const float c = 0.7978845608028654f;  // sqrt(2/pi)
                    v = 0.5f * v * (1.f + tanhf(c * (v + 0.044715f * v * v * v)));
                } else if (mode == 2) {
                    v += out[oi];
                }
                out[oi] = v;
            }
        }
    }
}
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* part, int S, const float* bias, float* out, int M, int N,
                                                            int ldo, int mode) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)M * N) return;
    const int m = (int)(e / N), n = (int)(e - (long long)m * N);
    float v = 0.f;
    for (int z0 = 0; z0 < S; z0 += 4) {        // four slices' loads in flight together; added in slice order
        float pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pv[u] = z0 + u < S ? part[((long long)(z0 + u) * M + m) * N + n] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) v += pv[u];
    }
    v += bias ? bias[n] : 0.f;
    const long long oi = (long long)m * ldo + n;
    if (mode == 1) {
        const float c = 0.7978845608028654f;  // sqrt(2/pi)
        v = 0.5f * v * (1.f + tanhf(c * (v + 0.044715f * v * v * v)));
    } else if (mode == 2) {
        v += out[oi];
    }
    out[oi] = v;
}
// ---- single-token steps (M <= 64 rows): weight-streaming form ------------------------------------------------------------------------
// Round 3 (rocprofv3 of `bench.py --config gpt2`): the decode steps' 4266 launches of gemm_f32_kernel<64,64> averaged 18.6 us — 44 % of
// the whole C5 step — for 0.4-9 MB of weights each: scalar LDS staging of BOTH operands, two barriers per 32-deep K step, no load in
// flight across them, and 64 fp32 MFMAs (64 cycles each) in a row per wave and K step.  Here neither operand touches LDS: a wave owns a
// 32 (rows m) x 32 (columns n) output block over a K part and reads BOTH fragments straight from global memory in MFMA layout — lane
// (lr, kh) loads 16 bytes = four consecutive k of ITS row (activation row m0 + lr / weight row n0 + lr): k = 8 b + 4 kh + s feeds MFMA
// step (b, s); the K order of an MFMA is free as long as both operands use the same one.  A 64-deep chunk (8 + 8 loads per lane) is
// requested one chunk ahead of its 32 MFMAs; no barrier in the K loop.  A workgroup = one 32-column block x (2 row blocks x NK K parts)
// waves, so the fp32 MFMA work (the floor of these steps: 15.9 GFLOP at 157 TFLOP/s) spreads over every SIMD while the global split-K
// factor (part[z][M][N], fixed-order sum in splitk_reduce_kernel) stays small; the NK partial blocks meet through LDS in a fixed order.
// Operands are ordered (a = activations, b = weights): a lane holds ONE column n for rows m = mfma32_row, stores are coalesced along n.
#define GS_KC 64
// Weight fragments of the single-token steps.  Non-temporal loads (MI355X_MICROARCH.md price list, row nt-weights: for weights that ONE CU
// reads once) were measured here and LOSE (r05, same box: decode 27.7 -> 32.1 ms per population): the two row-block waves of a workgroup
// read the same fragments, and the second one finds them in the L1 only with the default policy.  -DGLASS_GPT2_NT rebuilds the experiment.
#ifdef GLASS_GPT2_NT
#define GS_WLOAD(p) __builtin_nontemporal_load((const f4*)(p))
#else
#define GS_WLOAD(p) (*(const f4*)(p))
#endif
// LNX: the activation operand is LayerNorm(A) (model.py:15-28), applied to the fragments in registers: (a - mean[m]) * rstd[m] * g[k] + b[k] with
// the row statistics from gpt2_finalize_kernel (same arithmetic as layernorm_kernel) and g / b of the workgroup's K slice parked in LDS —
// the separate LayerNorm launch and its [M][D] round trip are gone.
// ONE: a wave's K part is a single chunk (every step product but the vocabulary projection): one register set instead of two — ~100
// VGPRs, so two 512-thread workgroups share a CU and the 288-workgroup grids of the MLP products run in one round.
template <int NK, bool LNX, bool ONE>
__global__ __launch_bounds__(128 * NK) void gemm_f32_stream_kernel(const float* A, const float* W, const float* bias, float* out, int M, int N,
                                                                   int K, int lda, int ldo, int mode, float* part, int kslice,
                                                                   const float* stats, const float* lng, const float* lnb) {
    __shared__ float Rs[NK > 1 ? NK - 1 : 1][2][16][64];     // partial blocks of K parts 1 .. NK-1
    __shared__ __attribute__((aligned(16))) float Gs[LNX ? 1024 : 4], Bs[LNX ? 1024 : 4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 31, kh = lane >> 5;
    const int mi = wave & 1, kp = wave >> 1;
    const int n0 = blockIdx.x * 32;
    const int kpart = kslice / NK, k_lo = blockIdx.y * kslice + kp * kpart, n_ch = kpart / GS_KC;
    const float* wrow = W + (long long)min(n0 + lr, N - 1) * K + k_lo + 4 * kh;          // rows past N / M re-read the last row (never stored)
    const float* xrow = A + (long long)min(mi * 32 + lr, M - 1) * lda + k_lo + 4 * kh;
    f4 wr[ONE ? 1 : 2][8], xr[ONE ? 1 : 2][8];
    auto load = [&](int c, int set) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            wr[set][b] = GS_WLOAD(wrow + c * GS_KC + 8 * b);
            xr[set][b] = *(const f4*)(xrow + c * GS_KC + 8 * b);
        }
    };
    f16x acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    load(0, 0);
    float mu = 0.f, rs = 1.f;
    if (LNX) {
        const int m = min(mi * 32 + lr, M - 1);
        mu = stats[2 * m]; rs = stats[2 * m + 1];
        for (int i = t; i < kslice; i += 128 * NK) { Gs[i] = lng[blockIdx.y * kslice + i]; Bs[i] = lnb[blockIdx.y * kslice + i]; }
        __syncthreads();
    }
    auto chunk = [&](int c, int set) {            // set = c & 1, a compile-time constant at both call sites
        if (!ONE && c + 1 < n_ch) load(c + 1, ONE ? 0 : set ^ 1);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            f4 xv = xr[set][b];
            if (LNX) {
                const int kl = kp * kpart + c * GS_KC + 8 * b + 4 * kh;
                xv = (xv - mu) * rs * *(const f4*)(Gs + kl) + *(const f4*)(Bs + kl);
            }
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[s2], wr[set][b][s2], acc, 0, 0, 0);   // D[m][n]
        }
    };
    if (ONE) {
        chunk(0, 0);                               // (n_ch == 1: nothing to prefetch)
    } else {
        for (int c = 0; c < n_ch; c += 2) {
            chunk(c, 0);
            if (c + 1 < n_ch) chunk(c + 1, 1);
        }
    }
    if (NK > 1) {                                  // K parts 1 .. NK-1 hand their blocks to part 0, which adds them in order
        if (kp > 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) Rs[kp - 1][mi][q][lane] = acc[q];
        }
        __syncthreads();
        if (kp > 0) return;
#pragma unroll
        for (int z = 0; z < NK - 1; ++z)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] += Rs[z][mi][q][lane];
    }
    const int n = n0 + lr;
    if (n >= N) return;
    const float bv = (bias && gridDim.y == 1) ? bias[n] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int m = mi * 32 + mfma32_row(reg, lane);
        if (m >= M) continue;
        if (gridDim.y > 1) {
            part[((long long)blockIdx.y * M + m) * N + n] = acc[reg];
            continue;
        }
        float v = acc[reg] + bv;
        const long long oi = (long long)m * ldo + n;
        if (mode == 1) {
            const float c = 0.7978845608028654f;  // sqrt(2/pi)
            v = 0.5f * v * (1.f + tanhf(c * (v + 0.044715f * v * v * v)));
        } else if (mode == 2) {
            v += out[oi];
        }
        out[oi] = v;
    }
}
template <int NK>
static void launch_stream_inst(const float* A, const float* W, const float* bias, float* out, int M, int N, int K, int lda, int ldo, int mode,
                               hipStream_t st, float* part, int S, const float* stats = nullptr, const float* lng = nullptr,
                               const float* lnb = nullptr) {
    const bool one = K / S / NK == GS_KC;
    const dim3 g((N + 31) / 32, S), b(128 * NK);
#define GS_LAUNCH(LN, ON) hipLaunchKernelGGL((gemm_f32_stream_kernel<NK, LN, ON>), g, b, 0, st, A, W, bias, out, M, N, K, lda, ldo, mode, part, K / S, stats, lng, lnb)
    if (stats) { if (one) GS_LAUNCH(true, true); else GS_LAUNCH(true, false); }
    else { if (one) GS_LAUNCH(false, true); else GS_LAUNCH(false, false); }
#undef GS_LAUNCH
}

// ---- single-token steps, complete-output form (round 4) ---------------------------------------------------------------------------------
// rocprofv3's timeline of a step (profiles/r04_gpt2_step_timeline.txt): every launch of the captured graph costs ~4.2 us before its first
// instruction, whatever it does — gpt2_advance_kernel, ONE thread, is 4.2 us start to start; HIP_FORCE_DEV_KERNARG, packet capture, eager
// launches and a second stream with half the rows change nothing (the dispatch cost is serial device-wide) — and a step was ~100 launches:
// 420 of its 940 us.  What a step can still save is LAUNCHES.  The split-K products needed two helpers per layer that only existed to
// add slices (splitk_reduce for the GELU input, gpt2_finalize for the residual stream + LayerNorm statistics) and an attention kernel
// that summed slices on load.  Here a workgroup owns 32 rows x 32 columns over the WHOLE K: NK = 12 waves take one 64-deep chunk each
// (K = 768; four chunks each through two register sets at K = 3072) — every load of the workgroup is in flight at once, as in
// gemm_f32_stream_kernel, but two row blocks per column block double the workgroup count instead of a global split (48 / 192 workgroups of
// 768 threads for the two products that use it) — and the twelve partial blocks meet through LDS in a fixed order.  Bias, GELU and the residual add happen
// in the epilogue; the epilogue of a residual product also leaves each row's (mean, M2) over its 32 columns, and the next LayerNorm-fused
// product combines a row's 24 partials (equal ranges: mean of means, M2s + range size x squared mean offsets; fixed order) while its operands
// travel.  Measured per product (us, start to start, old split form incl. its helper launch -> complete form): attention output 7.1 + 5.0 ->
// 11.4, MLP first 13.4 + 4.9 -> 14.4-16; the qkv product 9.5 -> 12-14 (its slices are summed by the attention kernel for free) and the
// MLP's second one 11.9 + 5.0 -> 34 (K = 3072: four chunks per wave on 48 workgroups) stay split: 6 launches per layer instead of 8.
// pst layout: [row][np][2] = (mean, sum of squared deviations) of the row's np equal column ranges.
template <int NK, bool LNX, bool ONE>
__global__ __launch_bounds__(64 * NK) void gemm_f32_rowblk_kernel(const float* A, const float* W, const float* bias, float* out, int M, int N,
                                                                  int K, int lda, int ldo, int mode, const float* pst_in, int np_in,
                                                                  const float* lng, const float* lnb, float* pst_out) {
    static_assert(NK >= 8, "eight waves finish the block");
    __shared__ float Rs[NK][16][64];                          // the K parts' partial blocks
    __shared__ __attribute__((aligned(16))) float Gs[LNX ? 1024 : 4], Bs[LNX ? 1024 : 4];
    __shared__ float Ms[32], Is[32];                          // LNX: mean / rstd of the workgroup's 32 rows
    const int t = threadIdx.x, lane = t & 63, kp = t >> 6, lr = lane & 31, kh = lane >> 5;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int kpart = K / NK, k_lo = kp * kpart, n_ch = kpart / GS_KC;
    const float* wrow = W + (long long)min(n0 + lr, N - 1) * K + k_lo + 4 * kh;          // rows past N / M re-read the last row (never stored)
    const float* xrow = A + (long long)min(m0 + lr, M - 1) * lda + k_lo + 4 * kh;
    f4 wr[ONE ? 1 : 2][8], xr[ONE ? 1 : 2][8];
    auto load = [&](int c, int set) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            wr[set][b] = GS_WLOAD(wrow + c * GS_KC + 8 * b);
            xr[set][b] = *(const f4*)(xrow + c * GS_KC + 8 * b);
        }
    };
    f16x acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    load(0, 0);
    float mu = 0.f, rs = 1.f;
    if (LNX) {
        for (int i = t; i < K; i += 64 * NK) { Gs[i] = lng[i]; Bs[i] = lnb[i]; }
        if (t < 32) {
            const float* pp = pst_in + (long long)min(m0 + t, M - 1) * np_in * 2;
            constexpr int NPMAX = 24;
            float pm[NPMAX], pq[NPMAX];
#pragma unroll
            for (int j = 0; j < NPMAX; ++j) {                 // one batch of loads (clamped index), then the fixed-order combine
                const int jj = min(j, np_in - 1);
                pm[j] = pp[2 * jj]; pq[j] = pp[2 * jj + 1];
            }
            // equal ranges: mean = mean of the means, M2 = sum of the M2s + count * sum of the squared mean offsets (fixed order j = 0 ..)
            float msum = 0.f, m2 = 0.f;
#pragma unroll
            for (int j = 0; j < NPMAX; ++j) {
                msum += j < np_in ? pm[j] : 0.f;
                m2 += j < np_in ? pq[j] : 0.f;
            }
            const float mean = msum / (float)np_in;
            float off = 0.f;
#pragma unroll
            for (int j = 0; j < NPMAX; ++j) {
                const float d = pm[j] - mean;
                off += j < np_in ? d * d : 0.f;
            }
            m2 += (float)(K / np_in) * off;
            Ms[t] = mean;
            Is[t] = rsqrtf(m2 / (float)K + 1e-5f);
        }
        __syncthreads();
        mu = Ms[lr]; rs = Is[lr];
    }
    auto chunk = [&](int c, int set) {            // set = c & 1, a compile-time constant at both call sites
        if (!ONE && c + 1 < n_ch) load(c + 1, ONE ? 0 : set ^ 1);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            f4 xv = xr[set][b];
            if (LNX) {
                const int kl = k_lo + c * GS_KC + 8 * b + 4 * kh;
                xv = (xv - mu) * rs * *(const f4*)(Gs + kl) + *(const f4*)(Bs + kl);
            }
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[s2], wr[set][b][s2], acc, 0, 0, 0);   // D[m][n]
        }
    };
    if (ONE) {
        chunk(0, 0);
    } else {
        for (int c = 0; c < n_ch; c += 2) {
            chunk(c, 0);
            if (c + 1 < n_ch) chunk(c + 1, 1);
        }
    }
    // the NK partial blocks meet in LDS; waves 0 .. 7 then finish two accumulator registers (= 2 x 2 rows of the block) each: K parts added
    // in order 0 .. NK-1, bias / GELU / residual, store, row partials
#pragma unroll
    for (int q = 0; q < 16; ++q) Rs[kp][q][lane] = acc[q];
    __syncthreads();
    if (kp >= 8) return;
    const int n = n0 + lr;
    const bool n_ok = n < N;
    const float bv = (bias && n_ok) ? bias[n] : 0.f;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
        const int reg = 2 * kp + qq;
        const int m = m0 + mfma32_row(reg, lane);
        const bool ok = n_ok && m < M;
        const float rv = (mode == 2 && ok) ? out[(long long)m * ldo + n] : 0.f;       // residual value: in flight under the LDS reads
        float pz[NK];
#pragma unroll
        for (int z = 0; z < NK; ++z) pz[z] = Rs[z][reg][lane];
        float v = pz[0];
#pragma unroll
        for (int z = 1; z < NK; ++z) v += pz[z];
        v += bv;
        if (mode == 1) {
            const float c = 0.7978845608028654f;  // sqrt(2/pi)
            v = 0.5f * v * (1.f + tanhf(c * (v + 0.044715f * v * v * v)));
        } else if (mode == 2) {
            v += rv;
        }
        if (ok) out[(long long)m * ldo + n] = v;
        if (pst_out) {                             // (N % 32 == 0: every lane holds a column) this row's 32 columns sit in the lanes of one kh half
            float sum = v;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            const float mean = sum * (1.f / 32.f), d = v - mean;
            float q = d * d;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o);
            if (lr == 0 && m < M) {
                float* po = pst_out + ((long long)m * gridDim.x + blockIdx.x) * 2;
                po[0] = mean; po[1] = q;
            }
        }
    }
}
bool gemm_f32_rowblk_supported(int M, int N, int K, int lda, bool ln_fused, bool stats_out) {
    return M > 0 && M <= 64 && N % 32 == 0 && K % (12 * GS_KC) == 0 && lda % 4 == 0 && !(ln_fused && K > 1024) && !(stats_out && N / 32 > 24);
}
// false: shape not covered, nothing launched.  pst_in / np_in: the LayerNorm-fused operand's row partials; pst_out: [M][N / 32][2]
bool launch_gemm_f32_rowblk(const float* A, const float* W, const float* bias, float* out, int M, int N, int K, int lda, int ldo, int mode,
                            hipStream_t st, const float* pst_in, int np_in, const float* lng, const float* lnb, float* pst_out) {
    if (!gemm_f32_rowblk_supported(M, N, K, lda, pst_in != nullptr, pst_out != nullptr) || (pst_in && (np_in < 1 || np_in > 24 || K % np_in != 0)))
        return false;
    const dim3 g(N / 32, (M + 31) / 32), b(64 * 12);
    const bool one = K == 12 * GS_KC;
#define RB_LAUNCH(LN, ON) hipLaunchKernelGGL((gemm_f32_rowblk_kernel<12, LN, ON>), g, b, 0, st, A, W, bias, out, M, N, K, lda, ldo, mode, pst_in, np_in, lng, lnb, pst_out)
    if (pst_in) { if (one) RB_LAUNCH(true, true); else RB_LAUNCH(true, false); }
    else { if (one) RB_LAUNCH(false, true); else RB_LAUNCH(false, false); }
#undef RB_LAUNCH
    return true;
}

// x[m][:] += bias + sum_s part[s][m][:] (split-K slices in a fixed order; part == nullptr: x as it is), then the row's LayerNorm statistics
// {mean, rstd} for the NEXT matrix product's fused LayerNorm (two-pass mean / variance as ln_row, kernels_clip.hip) — one workgroup per
// row.  D <= 1024.
__global__ __launch_bounds__(256) void gpt2_finalize_kernel(const float* part, int S, const float* bias, float* x, int M, int D, float* stats) {
    __shared__ float red[4];
    const int row = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float* xr = x + (long long)row * D;
    float v[4], a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = t + 256 * k < D ? xr[t + 256 * k] : 0.f; a[k] = 0.f; }
    if (part) {
        int z = 0;
        for (; z + 4 <= S; z += 4) {                  // four slices' loads in flight together; per element the slices add in order
            float pv[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) pv[u][k] = t + 256 * k < D ? part[((long long)(z + u) * M + row) * D + t + 256 * k] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] += pv[u][k];
        }
        for (; z < S; ++z)
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] += t + 256 * k < D ? part[((long long)z * M + row) * D + t + 256 * k] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = t + 256 * k;
            if (i < D) {
                v[k] = v[k] + (a[k] + (bias ? bias[i] : 0.f));          // (splitk_reduce_kernel's order: slices, + bias, + residual)
                xr[i] = v[k];
            }
        }
    }
    auto block_sum = [&](float q) {                   // fixed order: lanes (xor tree), then waves 0..3
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        __syncthreads();
        if (lane == 0) red[wave] = q;
        __syncthreads();
        return ((red[0] + red[1]) + red[2]) + red[3];
    };
    const float mean = block_sum((v[0] + v[1]) + (v[2] + v[3])) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float d = t + 256 * k < D ? v[k] - mean : 0.f; q += d * d; }
    const float var = block_sum(q) / (float)D;
    if (t == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rsqrtf(var + 1e-5f); }
}
void launch_gpt2_finalize(const float* part, int S, const float* bias, float* x, int M, int D, float* stats, hipStream_t st) {
    hipLaunchKernelGGL(gpt2_finalize_kernel, dim3(M), dim3(256), 0, st, part, S, bias, x, M, D, stats);
}

// ---- vocabulary projection of a single-token step: LayerNorm and the greedy pick's first stage fused -----------------------------------
// logits[m][n] = ln_f(x)[m] . wte[n] for M <= 64 rows and V = 50257 columns is the one product of a step with real arithmetic (4.9 GFLOP:
// 31 us of fp32 MFMA at the peak) — gemm_f32_stream_kernel ran it at 123 us: each of its 1571 workgroups re-read the activation
// fragments (300 MB out of L2).  Here a workgroup's waves own one 32-column block each and SHARE the activation chunk through LDS (64 k x
// 64 rows, transposed, LayerNorm applied on the way in, double-buffered, one barrier per chunk); weight fragments stream straight from
// global memory as in gemm_f32_stream_kernel, one 64-deep chunk ahead; both row blocks per wave (64 MFMAs per chunk).  Each wave then
// reduces every row's 32 columns to (max, lowest index) — stage 1 of the arg-max (sample.py:28-34, greedy) — and stores that pair per
// (row, column block) instead of the logits (logits != nullptr: they are written too).
// (A persistent form with the activation fragments resident in registers — 96 + 64 + 16 VGPRs live — spilled 130-180 registers.)
#define HD_NW 4
#define HD_LD 65
__global__ __launch_bounds__(64 * HD_NW) void gpt2_head_kernel(const float* A, const float* W, int M, int N, int K, int lda, const float* stats,
                                                              const float* lng, const float* lnb, float* logits, float* pv, int* pi, int NB) {
    __shared__ float Xs[2][GS_KC][HD_LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 31, kh = lane >> 5;
    const int nb = blockIdx.x * HD_NW + wave, n_ch = K / GS_KC;
    const float* wrow = W + (long long)min(nb * 32 + lr, N - 1) * K + 4 * kh;      // rows past N re-read row N - 1 (masked below)
    constexpr int XV = 1024 / (64 * HD_NW);                  // float4 loads per thread per chunk: (row, 16-byte piece of the row's 256 bytes)
    f4 xr[XV];
    float mu[XV], rs[XV];
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        const int m = min((i * 64 * HD_NW + t) >> 4, M - 1);
        mu[i] = stats[2 * m]; rs[i] = stats[2 * m + 1];
    }
    auto load_x = [&](int c) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = i * 64 * HD_NW + t, m = min(v >> 4, M - 1), k = c * GS_KC + (v & 15) * 4;
            xr[i] = (*(const f4*)(A + (long long)m * lda + k) - mu[i]) * rs[i] * *(const f4*)(lng + k) + *(const f4*)(lnb + k);
        }
    };
    auto store_x = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = i * 64 * HD_NW + t, m = v >> 4, kq = v & 15;
#pragma unroll
            for (int j = 0; j < 4; ++j) Xs[buf][kq * 4 + j][m] = xr[i][j];
        }
    };
    f4 wr[2][8];
    auto load_w = [&](int c, int set) {
#pragma unroll
        for (int b = 0; b < 8; ++b) wr[set][b] = GS_WLOAD(wrow + c * GS_KC + 8 * b);
    };
    f16x acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    load_x(0);
    load_w(0, 0);
    store_x(0);
    auto chunk = [&](int c, int set) {            // set = c & 1, a compile-time constant at both call sites
        __syncthreads();                          // chunk c's activations are in Xs[set]; everyone is done with Xs[set ^ 1]
        if (c + 1 < n_ch) { load_x(c + 1); load_w(c + 1, set ^ 1); }
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const int k = 8 * b + 4 * kh + s2;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[set][k][i * 32 + lr], wr[set][b][s2], acc[i], 0, 0, 0);   // D[m][n]
            }
        if (c + 1 < n_ch) store_x(set ^ 1);
    };
    for (int c = 0; c < n_ch; c += 2) {
        chunk(c, 0);
        if (c + 1 < n_ch) chunk(c + 1, 1);
    }
    if (nb >= NB) return;
    const int n = nb * 32 + lr;
    // Row maxima across the 32 columns of the block (the lanes of one kh half): the xor butterfly runs STEP-major over all 32 rows a
    // lane holds (r04: row-major, every step waited for its own ds_bpermute round trip — 160 dependent ones, 16 us of the kernel's 87),
    // values only; the winning column is the lowest lane of the half whose value equals the maximum (one ballot per row).
    float best[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = i * 32 + mfma32_row(reg, lane);
            if (logits && n < N && m < M) logits[(long long)m * N + n] = acc[i][reg];
            best[i][reg] = n < N ? acc[i][reg] : -INFINITY;
        }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        float other[2][16];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) other[i][reg] = __shfl_xor(best[i][reg], o);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) best[i][reg] = fmaxf(best[i][reg], other[i][reg]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = i * 32 + mfma32_row(reg, lane);
            const float mine = n < N ? acc[i][reg] : -INFINITY;
            const unsigned long long hit = __ballot(mine == best[i][reg]);             // lanes holding their row's maximum
            const unsigned half = (unsigned)(hit >> (32 * kh));                          // this half's rows
            if (lr == 0 && m < M) {
                pv[(long long)m * NB + nb] = best[i][reg];
                pi[(long long)m * NB + nb] = nb * 32 + (half ? __builtin_ctz(half) : 0);   // lowest column on exact ties (torch.topk's first maximum)
            }
        }
}
// stage 2: one workgroup per row over its NB (value, index) pairs -> out[step * rows + row]
__global__ __launch_bounds__(256) void argmax_pairs_kernel(const float* pv, const int* pi, int NB, int* out, const int* step_dev) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < NB; i += 256) {
        const float v = pv[(long long)blockIdx.x * NB + i];
        const int j = pi[(long long)blockIdx.x * NB + i];
        if (v > best || (v == best && j < idx)) { best = v; idx = j; }
    }
    bv[threadIdx.x] = best;
    bi[threadIdx.x] = idx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float v2 = bv[threadIdx.x + o];
            const int i2 = bi[threadIdx.x + o];
            if (v2 > bv[threadIdx.x] || (v2 == bv[threadIdx.x] && i2 < bi[threadIdx.x])) { bv[threadIdx.x] = v2; bi[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(step_dev ? (long long)step_dev[1] * gridDim.x : 0) + blockIdx.x] = bi[0];
}
// stage 2 + the NEXT step's first kernel + the state update in one launch (round 4: a launch is ~4 us of dispatch whatever it does): the
// row's token goes to gen[step][row], its embedding at position past + 1 and the first layer's LayerNorm statistics are left exactly as
// gpt2_embed_step_kernel would leave them at the top of the next step (same arithmetic), and the LAST workgroup to get here advances
// {past, step} — every workgroup has read the state before it takes its ticket.  state[2] is the ticket counter (zero between launches).
__global__ __launch_bounds__(256) void gpt2_pick_embed_kernel(const float* pv, const int* pi, int NB, int* gen, int* state, int P, const float* wte,
                                                              const float* wpe, int D, float* x, float* stats) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    __shared__ float red[4];
    const int row = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int past = state[0], step = state[1];
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = t; i < NB; i += 256) {
        const float v = pv[(long long)row * NB + i];
        const int j = pi[(long long)row * NB + i];
        if (v > best || (v == best && j < idx)) { best = v; idx = j; }
    }
    bv[t] = best;
    bi[t] = idx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            const float v2 = bv[t + o];
            const int i2 = bi[t + o];
            if (v2 > bv[t] || (v2 == bv[t] && i2 < bi[t])) { bv[t] = v2; bi[t] = i2; }
        }
        __syncthreads();
    }
    const int tok = bi[0];
    if (t == 0) gen[(long long)step * P + row] = tok;
    const float* te = wte + (long long)tok * D;
    const float* pe = wpe + (long long)(past + 1) * D;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = t + 256 * k;
        if (i < D) { v[k] = te[i] + pe[i]; x[(long long)row * D + i] = v[k]; }
    }
    auto block_sum = [&](float q) {
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        __syncthreads();
        if (lane == 0) red[wave] = q;
        __syncthreads();
        return ((red[0] + red[1]) + red[2]) + red[3];
    };
    const float mean = block_sum((v[0] + v[1]) + (v[2] + v[3])) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float d = t + 256 * k < D ? v[k] - mean : 0.f; q += d * d; }
    const float var = block_sum(q) / (float)D;
    if (t == 0) {
        stats[2 * row] = mean; stats[2 * row + 1] = rsqrtf(var + 1e-5f);
        __threadfence();
        if (atomicAdd(&state[2], 1) == (int)gridDim.x - 1) { state[0] = past + 1; state[1] = step + 1; state[2] = 0; }
    }
}
bool gpt2_head_supported(int M, int N, int K, int lda) { return M <= 64 && K % GS_KC == 0 && lda % 4 == 0 && N >= 4096; }
// the vocabulary projection + the fused pick / embed / advance tail (D <= 1024); false: shape not covered, nothing launched
bool launch_gpt2_head_tail(const float* A, const float* W, int M, int N, int K, int lda, const float* stats_in, const float* lng, const float* lnb,
                           float* pairs, int* gen, int* state, const float* wte, const float* wpe, float* x, float* stats_out, hipStream_t st) {
    if (!gpt2_head_supported(M, N, K, lda) || !pairs || K > 1024) return false;
    const int NB = (N + 31) / 32;
    int* pi = (int*)(pairs + (size_t)M * NB);
    hipLaunchKernelGGL(gpt2_head_kernel, dim3((NB + HD_NW - 1) / HD_NW), dim3(64 * HD_NW), 0, st, A, W, M, N, K, lda, stats_in, lng, lnb, (float*)nullptr, pairs, pi, NB);
    hipLaunchKernelGGL(gpt2_pick_embed_kernel, dim3(M), dim3(256), 0, st, pairs, pi, NB, gen, state, M, wte, wpe, K, x, stats_out);
    return true;
}
// false: shape not covered (caller: generic product + launch_argmax).  pairs: scratch of 2 * M * ceil(N / 32) words.
bool launch_gpt2_head(const float* A, const float* W, int M, int N, int K, int lda, const float* stats, const float* lng, const float* lnb,
                      float* logits, float* pairs, int* out, const int* step_dev, hipStream_t st) {
    if (!gpt2_head_supported(M, N, K, lda) || !pairs) return false;
    const int NB = (N + 31) / 32;
    int* pi = (int*)(pairs + (size_t)M * NB);
    hipLaunchKernelGGL(gpt2_head_kernel, dim3((NB + HD_NW - 1) / HD_NW), dim3(64 * HD_NW), 0, st, A, W, M, N, K, lda, stats, lng, lnb, logits, pairs, pi, NB);
    hipLaunchKernelGGL(argmax_pairs_kernel, dim3(M), dim3(256), 0, st, pairs, pi, NB, out, step_dev);
    return true;
}

// A single-token-step product of the fused step (engine.cpp): launch_gemm_f32's split policy, but the caller finishes the slices — returns
// the global split S: S == 1: bias / mode were applied by the kernel; S > 1: raw sums are in `part` (splitk_reduce_kernel via
// launch_gpt2_reduce, or gpt2_finalize_kernel for the residual products).  stats != nullptr: LayerNorm fused on the activation operand
// (K <= 1024).  Returns 0 when the shape does not fit (caller: unfused path).
bool gemm_f32_step_supported(int M, int K, int lda, bool ln_fused) {
    return M <= 64 && K % GS_KC == 0 && lda % 4 == 0 && !(ln_fused && K > 1024);
}
int launch_gemm_f32_step(const float* A, const float* W, const float* bias, float* out, int M, int N, int K, int lda, int ldo, int mode,
                         hipStream_t st, float* part, size_t part_elems, const float* stats, const float* lng, const float* lnb) {
    if (!gemm_f32_step_supported(M, K, lda, stats != nullptr)) return 0;
    const int nb = (N + 31) / 32;
    // K = C chunks of 64 = S global slices x NK parts inside a workgroup (one chunk per wave).  Of the factorizations with NK in
    // {1, 2, 4, 6} take the one with the most workgroups that still fit the chip in ONE round (nb * S <= CUs: 288 workgroups on 256 CUs
    // ran the MLP products at 14 us against 9.7 us for the 216 of the qkv product), fewest slices among equals; a product too small to
    // fill the CUs either way takes the most slices its scratch allows.
    static const bool old_rule = glass_knob("GLASS_GPT2_OLD_SPLIT") != nullptr;     // A/B knob
    const int C = K / GS_KC, n_cu = glass_cu_count();
    int S = 0, NK = 1;
    if (!old_rule && nb < 1024) {
        static const int nks[] = {6, 4, 2, 1};          // (8 parts = 1024 threads = a 128-VGPR budget: spills)
        int best_wg = -1;
        for (int nk : nks) {
            if (C % nk != 0) continue;
            const int sl = C / nk;
            if (sl > 1 && (!part || (size_t)sl * M * N > part_elems || sl > 16)) continue;
            const int wg = nb * sl;
            const bool fits = wg <= n_cu;
            const int score = fits ? wg : -wg;              // prefer fitting grids, then the larger one; non-fitting: the smaller
            if (S == 0 || score > best_wg) { best_wg = score; S = sl; NK = nk; }
        }
    }
    if (S == 0) {      // the vocabulary projection (never split globally), shapes the rule above does not cover, and the A/B knob
        S = 1; NK = 1;
        static const int cand[] = {1, 2, 3, 4, 6, 8, 12, 16};
        for (int c : cand) {
            if (K % (c * GS_KC) != 0 || (c > 1 && (!part || (size_t)c * M * N > part_elems || nb >= 1024))) continue;
            S = c;
            const int ks = K / c;
            NK = ks % (4 * GS_KC) == 0 ? 4 : (ks % (2 * GS_KC) == 0 ? 2 : 1);
            if (nb * c >= 200 && ks / NK <= 2 * GS_KC) break;
        }
    }
    switch (NK) {
        case 6: launch_stream_inst<6>(A, W, bias, out, M, N, K, lda, ldo, mode, st, part, S, stats, lng, lnb); break;
        case 4: launch_stream_inst<4>(A, W, bias, out, M, N, K, lda, ldo, mode, st, part, S, stats, lng, lnb); break;
        case 2: launch_stream_inst<2>(A, W, bias, out, M, N, K, lda, ldo, mode, st, part, S, stats, lng, lnb); break;
        default: launch_stream_inst<1>(A, W, bias, out, M, N, K, lda, ldo, mode, st, part, S, stats, lng, lnb); break;
    }
    return S;
}
void launch_gpt2_reduce(const float* part, int S, const float* bias, float* out, int M, int N, int ldo, int mode, hipStream_t st) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((long long)M * N + 255) / 256)), dim3(256), 0, st, part, S, bias, out, M, N, ldo, mode);
}
// `part`: scratch for split-K partial sums (nullable = never split); sized by the caller for GPT2_SPLITK_MAX slices of M x N.
void launch_gemm_f32(const float* A, const float* W, const float* bias, float* out, int M, int N, int K, int lda, int ldo,
                     int mode, hipStream_t st, float* part, size_t part_elems, bool prefill) {
    static const bool no_stream = glass_knob("GLASS_GPT2_NO_STREAM") != nullptr;      // A/B knob: round 2's gemm_f32_kernel<64,64> for the steps
    if (!prefill && M <= 64 && K % GS_KC == 0 && lda % 4 == 0 && !no_stream) {
        // a workgroup per 32 weight rows; global K split S (small: the partial sums are traffic) x NK K parts inside the workgroup so
        // that a wave's share is one or two 64-deep chunks; the vocabulary projection (1571 workgroups) is not split globally
        const int nb = (N + 31) / 32;
        int S = 1, NK = 1;
        static const int cand[] = {1, 2, 3, 4, 6, 8, 12, 16};
        for (int c : cand) {
            if (K % (c * GS_KC) != 0 || (c > 1 && (!part || (size_t)c * M * N > part_elems || nb >= 1024))) continue;
            S = c;
            const int ks = K / c;
            NK = ks % (4 * GS_KC) == 0 ? 4 : (ks % (2 * GS_KC) == 0 ? 2 : 1);
            if (nb * c >= 200 && ks / NK <= 2 * GS_KC) break;
        }
        if (NK == 4) launch_stream_inst<4>(A, W, bias, out, M, N, K, lda, ldo, mode, st, part, S);
        else if (NK == 2) launch_stream_inst<2>(A, W, bias, out, M, N, K, lda, ldo, mode, st, part, S);
        else launch_stream_inst<1>(A, W, bias, out, M, N, K, lda, ldo, mode, st, part, S);
        if (S > 1)
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((long long)M * N + 255) / 256)), dim3(256), 0, st, part, S, bias, out,
                               M, N, ldo, mode);
        return;
    }
    if (!prefill && M <= 64) {
        // single-token steps stream each weight once: what matters is how many workgroups pull on HBM.  N / 64 column tiles
        // alone are 12 workgroups at N = 768: split K until ~256 workgroups are live.
        const int tiles = (N + 63) / 64;
        int S = 1;
        if (part) {
            while (tiles * S < 192 && S < 16 && K / (2 * S) >= 2 * F32_BK && (size_t)(2 * S) * M * N <= part_elems) S *= 2;
        }
        dim3 g((M + 63) / 64, tiles, S);
        hipLaunchKernelGGL((gemm_f32_kernel<64, 64>), g, dim3(256), 0, st, A, W, bias, out, M, N, K, lda, ldo, mode, part);
        if (S > 1)
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((long long)M * N + 255) / 256)), dim3(256), 0, st, part, S, bias, out,
                               M, N, ldo, mode);
    } else {
        // prefill: 64 x 64 tiles for every product.  Measured per shape over the whole decode (M = 64 x 23 rows, GLASS_GPT2_TILE): 128 x 128
        // everywhere 30.4 ms, 128 x 64 28.3, the shape whose grid quantises best on the CUs 28.2, 64 x 64 27.8 — a 64 x 64 workgroup is 17 KB
        // of LDS and 32 VGPRs, so a CU holds many of them and their barriers / LDS round trips overlap; the larger tiles run one
        // workgroup (one wave per SIMD) per CU.  Every shape adds an element's k terms in the same order: the choice never changes a value.
        static const int shapes[3][2] = {{128, 128}, {128, 64}, {64, 64}};
        static const int force = glass_knob("GLASS_GPT2_TILE") ? atoi(glass_knob("GLASS_GPT2_TILE")) : -1;   // A/B knob
        const int best = (force >= 0 && force < 3) ? force : 2;
        const dim3 g((M + shapes[best][0] - 1) / shapes[best][0], (N + shapes[best][1] - 1) / shapes[best][1], 1);
        if (best == 0) hipLaunchKernelGGL((gemm_f32_kernel<128, 128>), g, dim3(256), 0, st, A, W, bias, out, M, N, K, lda, ldo, mode, part);
        else if (best == 1) hipLaunchKernelGGL((gemm_f32_kernel<128, 64>), g, dim3(256), 0, st, A, W, bias, out, M, N, K, lda, ldo, mode, part);
        else hipLaunchKernelGGL((gemm_f32_kernel<64, 64>), g, dim3(256), 0, st, A, W, bias, out, M, N, K, lda, ldo, mode, part);
    }
}

// ---- attention with KV cache (model.py:59-95): one workgroup per (sequence, head) -----------------------
// qkv: [P*nd][3*D] for the nd new positions past..past+nd-1; kc/vc: caches [P][Tmax][D] (this call appends).
// w = q.k / sqrt(64); masked (key j > past + i) -> -1e10; softmax; a = w @ v.
// past_dev != nullptr: the past length is read from device memory (graph replay), `past` is ignored.
__global__ __launch_bounds__(256) void gpt2_attention_kernel(const float* qkv, float* kc, float* vc, int nd, int past,
                                                             int Tmax, int heads, float* out, const int* past_dev) {
    extern __shared__ float sm[];
    if (past_dev) past = *past_dev;
    const int hd = 64, D = heads * hd;
    const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
    const int ns = past + nd;
    float* q = sm;                    // [nd][65]
    float* k = q + nd * 65;           // [ns][65]
    float* v = k + ns * 65;           // [ns][65]
    float* s = v + ns * 65;           // [nd][ns + 1]
    // append the new keys / values to the cache, then load the whole history
    for (int e = threadIdx.x; e < nd * hd; e += 256) {
        const int i = e / hd, d = e - i * hd;
        const float* r = qkv + ((long long)seq * nd + i) * 3 * D + h * hd + d;
        q[i * 65 + d] = r[0];
        kc[((long long)seq * Tmax + past + i) * D + h * hd + d] = r[D];
        vc[((long long)seq * Tmax + past + i) * D + h * hd + d] = r[2 * D];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < ns * hd; e += 256) {
        const int j = e / hd, d = e - j * hd;
        k[j * 65 + d] = kc[((long long)seq * Tmax + j) * D + h * hd + d];
        v[j * 65 + d] = vc[((long long)seq * Tmax + j) * D + h * hd + d];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nd * ns; e += 256) {
        const int i = e / ns, j = e - i * ns;
        float a = 0.f;
        for (int d = 0; d < hd; ++d) a += q[i * 65 + d] * k[j * 65 + d];
        a *= 0.125f;
        if (j > past + i) a = -1e10f;   // w * b - 1e10 * (1 - b)
        s[i * (ns + 1) + j] = a;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < nd; i += 4) {
        float m = -INFINITY;
        for (int j = lane; j < ns; j += 64) m = fmaxf(m, s[i * (ns + 1) + j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float z = 0.f;
        for (int j = lane; j < ns; j += 64) {
            const float e2 = expf(s[i * (ns + 1) + j] - m);
            s[i * (ns + 1) + j] = e2;
            z += e2;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o);
        for (int j = lane; j < ns; j += 64) s[i * (ns + 1) + j] /= z;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nd * hd; e += 256) {
        const int i = e / hd, d = e - i * hd;
        float a = 0.f;
        for (int j = 0; j < ns; ++j) a += s[i * (ns + 1) + j] * v[j * 65 + d];
        out[((long long)seq * nd + i) * D + h * hd + d] = a;
    }
}
void launch_gpt2_attention(const float* qkv, float* kc, float* vc, int P, int nd, int past, int Tmax, int heads,
                           float* out, hipStream_t st, const int* past_dev) {
    const int ns = past_dev ? Tmax : past + nd;      // graph replay: LDS sized for the longest history
    const size_t lds = (size_t)(nd * 65 + 2 * ns * 65 + nd * (ns + 1)) * sizeof(float);
    static DevOnce once;
    once.run([&] { (void)hipFuncSetAttribute((const void*)gpt2_attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    hipLaunchKernelGGL(gpt2_attention_kernel, dim3(P * heads), dim3(256), lds, st, qkv, kc, vc, nd, past, Tmax, heads, out, past_dev);
}

// Single-token step (nd = 1, history <= 64 positions): ONE WAVE per (sequence, head), and the split-K slices of the qkv product are
// summed here (part / S / bias as in splitk_reduce_kernel; part == nullptr: qkv holds finished values) — the general kernel above spent
// 10 us on five barriers and scalar cache loads, plus 4.6 us for the reduce launch in front of it.  Lane = feature d for q / k / v and
// the output, lane = key position j for the scores (each lane reads one 256-byte key row); q and the probabilities cross lanes through
// LDS; the value rows are read coalesced.  Sum orders as in gpt2_attention_kernel.
__global__ __launch_bounds__(256) void gpt2_attention_step_kernel(const float* qkv, const float* part, int S, const float* bias, float* kc, float* vc,
                                                                  int P, int Tmax, int heads, float* out, const int* past_dev) {
    __shared__ __attribute__((aligned(16))) float qs[4][64], ks[4][64], ps[4][64], hs[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wave;
    if (item >= P * heads) return;                    // (no workgroup barrier below: waves are independent)
    const int past = *past_dev, ns = past + 1;
    const int D = heads * 64, seq = item / heads, h = item - seq * heads;
    // One round trip (round 4): this lane's key row (16 x 16 B), the value rows of the whole history (up to 64 coalesced rows, in
    // wave-uniform groups of 16) — the cache reads depend on `past` only — and the first four qkv slices are all requested before anything
    // is summed.  As written in round 3 the kernel was a chain of dependent round trips (slices, key rows, one per 16 value rows): 11.5 us.
    float r[3], bq[3] = {0.f, 0.f, 0.f}, pv[4][3];
    f4 kk[16];
    {
        const float* kr = kc + ((long long)seq * Tmax + min(lane, Tmax - 1)) * D + h * 64;      // (rows past the history: loaded, never used)
#pragma unroll
        for (int d4 = 0; d4 < 16; ++d4) kk[d4] = *(const f4*)(kr + 4 * d4);
    }
    const float* vr = vc + (long long)seq * Tmax * D + h * 64 + lane;
    float vv[4][16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (16 * g < past) {                          // (wave-uniform; clamped row, zero weight past the history)
#pragma unroll
            for (int u = 0; u < 16; ++u) vv[g][u] = vr[(long long)min(16 * g + u, past - 1) * D];
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) vv[g][u] = 0.f;
        }
    }
    if (part) {
        if (bias) {
#pragma unroll
            for (int w3 = 0; w3 < 3; ++w3) bq[w3] = bias[w3 * D + h * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int w3 = 0; w3 < 3; ++w3)       // (unconditional, clamped slice: no branch between the loads)
                pv[u][w3] = part[((long long)min(u, S - 1) * P + seq) * 3 * D + w3 * D + h * 64 + lane];
    } else {
#pragma unroll
        for (int w3 = 0; w3 < 3; ++w3) r[w3] = qkv[(long long)seq * 3 * D + w3 * D + h * 64 + lane];
    }
    if (part) {      // slice order per element; + 0 for the slices past S; further groups of four (S > 4) as they come
        float a[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int w3 = 0; w3 < 3; ++w3) a[w3] += u < S ? pv[u][w3] : 0.f;
        for (int z0 = 4; z0 < S; z0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int w3 = 0; w3 < 3; ++w3)
                    pv[u][w3] = part[((long long)min(z0 + u, S - 1) * P + seq) * 3 * D + w3 * D + h * 64 + lane];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int w3 = 0; w3 < 3; ++w3) a[w3] += z0 + u < S ? pv[u][w3] : 0.f;
        }
#pragma unroll
        for (int w3 = 0; w3 < 3; ++w3) r[w3] = a[w3] + bq[w3];
    }
    kc[((long long)seq * Tmax + past) * D + h * 64 + lane] = r[1];
    vc[((long long)seq * Tmax + past) * D + h * 64 + lane] = r[2];
    qs[wave][lane] = r[0];
    ks[wave][lane] = r[1];
    __builtin_amdgcn_wave_barrier();                  // LDS is in order per wave
    float sc = -INFINITY;
    if (lane < ns) {
        float a = 0.f;
        if (lane == past) {
#pragma unroll 16
            for (int d = 0; d < 64; ++d) a += qs[wave][d] * ks[wave][d];
        } else {
#pragma unroll
            for (int d4 = 0; d4 < 16; ++d4) {
                const f4 qq = *(const f4*)(&qs[wave][4 * d4]);
#pragma unroll
                for (int j = 0; j < 4; ++j) a += qq[j] * kk[d4][j];
            }
        }
        sc = a * 0.125f;
    }
    float m = sc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float e2 = lane < ns ? expf(sc - m) : 0.f;
    float z = e2;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o);
    const float prob = e2 / z;
    ps[wave][lane] = prob;
    hs[wave][lane] = lane < past ? prob : 0.f;        // the history's weights alone (zero from the current position on)
    __builtin_amdgcn_wave_barrier();
    float a = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (16 * g < past) {                          // sixteen value rows per group, key order
#pragma unroll
            for (int u4 = 0; u4 < 4; ++u4) {
                const f4 pw = *(const f4*)(&hs[wave][16 * g + 4 * u4]);
#pragma unroll
                for (int j = 0; j < 4; ++j) a += pw[j] * vv[g][4 * u4 + j];
            }
        }
    }
    a += ps[wave][past] * r[2];
    out[(long long)seq * D + h * 64 + lane] = a;
}
void launch_gpt2_attention_step(const float* qkv, const float* part, int S, const float* bias, float* kc, float* vc, int P, int Tmax, int heads,
                                float* out, hipStream_t st, const int* past_dev) {
    hipLaunchKernelGGL(gpt2_attention_step_kernel, dim3((P * heads + 3) / 4), dim3(256), 0, st, qkv, part, S, bias, kc, vc, P, Tmax, heads, out, past_dev);
}

// ---- greedy pick (sample.py:28-34 with sample=False): arg-max of softmax(top_k(logits / T)) == arg-max of the
// logits, lowest index on exact ties (torch.topk returns the first maximum) ------------------------------------
// step_dev != nullptr: row r's pick goes to out[step * gridDim.x + r] with step = step_dev[1] (graph replay)
// Two stages (round 3: one workgroup per row scanned 200 KB on its own, 56 us): stage 1 = ARGMAX_SEG segments per row -> (value, index)
// pairs, stage 2 = one wave per row over the pairs.
#define ARGMAX_SEG 32
__device__ __forceinline__ void argmax_block(float best, int idx, float* bv, int* bi) {      // 256 threads -> bv[0], bi[0]
    bv[threadIdx.x] = best;
    bi[threadIdx.x] = idx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float v2 = bv[threadIdx.x + o];
            const int i2 = bi[threadIdx.x + o];
            if (v2 > bv[threadIdx.x] || (v2 == bv[threadIdx.x] && i2 < bi[threadIdx.x])) { bv[threadIdx.x] = v2; bi[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void argmax_seg_kernel(const float* logits, int N, float* pv, int* pi) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    const int row = blockIdx.x / ARGMAX_SEG, seg = blockIdx.x % ARGMAX_SEG;
    const int per = (N + ARGMAX_SEG - 1) / ARGMAX_SEG, lo = seg * per, hi = min(N, lo + per);
    const float* r = logits + (long long)row * N;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        const float v = r[i];
        if (v > best) { best = v; idx = i; }
    }
    argmax_block(best, idx, bv, bi);
    if (threadIdx.x == 0) { pv[blockIdx.x] = bv[0]; pi[blockIdx.x] = bi[0]; }
}
__global__ __launch_bounds__(64) void argmax_final_kernel(const float* pv, const int* pi, int* out, const int* step_dev) {
    const int row = blockIdx.x, lane = threadIdx.x;
    float best = lane < ARGMAX_SEG ? pv[row * ARGMAX_SEG + lane] : -INFINITY;
    int idx = lane < ARGMAX_SEG ? pi[row * ARGMAX_SEG + lane] : 0x7fffffff;
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(best, o);
        const int i2 = __shfl_xor(idx, o);
        if (v2 > best || (v2 == best && i2 < idx)) { best = v2; idx = i2; }
    }
    if (lane == 0) out[(step_dev ? (long long)step_dev[1] * gridDim.x : 0) + row] = idx;
}
__global__ __launch_bounds__(256) void argmax_kernel(const float* logits, int N, int* out, const int* step_dev) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    const float* r = logits + (long long)blockIdx.x * N;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float v = r[i];
        if (v > best) { best = v; idx = i; }
    }
    argmax_block(best, idx, bv, bi);
    if (threadIdx.x == 0) out[(step_dev ? (long long)step_dev[1] * gridDim.x : 0) + blockIdx.x] = bi[0];
}
// scratch: rows * ARGMAX_SEG floats + as many ints (nullptr: the one-stage kernel)
void launch_argmax(const float* logits, int rows, int N, int* out, hipStream_t st, const int* step_dev, float* scratch) {
    if (scratch && N >= 8192) {
        int* pi = (int*)(scratch + (size_t)rows * ARGMAX_SEG);
        hipLaunchKernelGGL(argmax_seg_kernel, dim3(rows * ARGMAX_SEG), dim3(256), 0, st, logits, N, scratch, pi);
        hipLaunchKernelGGL(argmax_final_kernel, dim3(rows), dim3(64), 0, st, scratch, pi, out, step_dev);
        return;
    }
    hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(256), 0, st, logits, N, out, step_dev);
}
