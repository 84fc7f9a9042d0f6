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
template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* A, const float* W, const float* bias, float* out,
                                                       int M, int N, int K, int lda, int ldo, int mode) {
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
    for (int k0 = 0; k0 < K; k0 += F32_BK) {
        for (int e = t; e < BM * 8; e += 256) {   // stage BM x 32 of A (coalesced along k), zero-filled outside
            const int row = e >> 3, c4 = (e & 7) * 4;
            f4 va = {0.f, 0.f, 0.f, 0.f};
            if (m0 + row < M && k0 + c4 < K) va = *(const f4*)(A + (long long)(m0 + row) * lda + k0 + c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) As[row][c4 + j] = va[j];
        }
        for (int e = t; e < BN * 8; e += 256) {
            const int row = e >> 3, c4 = (e & 7) * 4;
            f4 vw = {0.f, 0.f, 0.f, 0.f};
            if (n0 + row < N && k0 + c4 < K) vw = *(const f4*)(W + (long long)(n0 + row) * K + k0 + c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) Ws[row][c4 + j] = vw[j];
        }
        __syncthreads();
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
void launch_gemm_f32(const float* A, const float* W, const float* bias, float* out, int M, int N, int K, int lda, int ldo,
                     int mode, hipStream_t st) {
    if (M <= 64) {
        dim3 g((M + 63) / 64, (N + 63) / 64);
        hipLaunchKernelGGL((gemm_f32_kernel<64, 64>), g, dim3(256), 0, st, A, W, bias, out, M, N, K, lda, ldo, mode);
    } else {
        dim3 g((M + 127) / 128, (N + 127) / 128);
        hipLaunchKernelGGL((gemm_f32_kernel<128, 128>), g, dim3(256), 0, st, A, W, bias, out, M, N, K, lda, ldo, mode);
    }
}

// ---- attention with KV cache (model.py:59-95): one workgroup per (sequence, head) -----------------------
// qkv: [P*nd][3*D] for the nd new positions past..past+nd-1; kc/vc: caches [P][Tmax][D] (this call appends).
// w = q.k / sqrt(64); masked (key j > past + i) -> -1e10; softmax; a = w @ v.
__global__ __launch_bounds__(256) void gpt2_attention_kernel(const float* qkv, float* kc, float* vc, int nd, int past,
                                                             int Tmax, int heads, float* out) {
    extern __shared__ float sm[];
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
                           float* out, hipStream_t st) {
    const int ns = past + nd;
    const size_t lds = (size_t)(nd * 65 + 2 * ns * 65 + nd * (ns + 1)) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gpt2_attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(gpt2_attention_kernel, dim3(P * heads), dim3(256), lds, st, qkv, kc, vc, nd, past, Tmax, heads, out);
}

// ---- greedy pick (sample.py:28-34 with sample=False): arg-max of softmax(top_k(logits / T)) == arg-max of the
// logits, lowest index on exact ties (torch.topk returns the first maximum) ------------------------------------
__global__ __launch_bounds__(256) void argmax_kernel(const float* logits, int N, int* out) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    const float* r = logits + (long long)blockIdx.x * N;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float v = r[i];
        if (v > best) { best = v; idx = i; }
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
    if (threadIdx.x == 0) out[blockIdx.x] = bi[0];
}
void launch_argmax(const float* logits, int rows, int N, int* out, hipStream_t st) {
    hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(256), 0, st, logits, N, out);
}
