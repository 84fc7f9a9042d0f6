// kernels_clip.hip — CLIP ViT glue kernels: token assembly + ln_pre, LayerNorm,
// multi-head attention for short sequences (L <= 128: 50 visual tokens, 77 text tokens),
// cosine similarity and objective assembly.  GEMMs live in conv_direct.hip / gemm_tiled.hip.
// Reference: clip/model.py:152-187 (LayerNorm, QuickGELU, ResidualAttentionBlock),
// :218-235 (VisualTransformer.forward), generator.py:51 (cosine), problem.py:21-27.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One wave per row; two-pass mean/variance in fp32 (clip/model.py:152-158, eps 1e-5).  Rows up to 1024 wide are read from
// global memory ONCE and held in registers (16 values per lane) for the three passes.
template <typename LoadF>
__device__ __forceinline__ void ln_row(LoadF load, int D, const float* g, const float* b, half_t* o16, float* o32) {
    const int lane = threadIdx.x & 63;
    if (D <= 1024) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = lane + 64 * k < D ? load(lane + 64 * k) : 0.f;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += v[k];
        const float mean = wsum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const float d = lane + 64 * k < D ? v[k] - mean : 0.f; q += d * d; }
        const float rstd = rsqrtf(wsum(q) / (float)D + 1e-5f);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = lane + 64 * k;
            if (i < D) {
                const float o = (v[k] - mean) * rstd * g[i] + b[i];
                if (o16) o16[i] = (half_t)o;
                if (o32) o32[i] = o;
            }
        }
        return;
    }
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s += load(i);
    const float mean = wsum(s) / (float)D;
    float q = 0.f;
    for (int i = lane; i < D; i += 64) { const float d = load(i) - mean; q += d * d; }
    const float rstd = rsqrtf(wsum(q) / (float)D + 1e-5f);
    for (int i = lane; i < D; i += 64) {
        const float v = (load(i) - mean) * rstd * g[i] + b[i];
        if (o16) o16[i] = (half_t)v;
        if (o32) o32[i] = v;
    }
}

// x[p][0] = class_embedding + pos[0]; x[p][1+t] = patch_emb[p*T0+t] + pos[1+t]; then ln_pre.
__global__ __launch_bounds__(256) void embed_lnpre_kernel(const float* pe, const float* cls, const float* pos,
                                                          const float* g, const float* b, int P, int T, int D,
                                                          float* x) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= P * T) return;
    const int p = row / T, t = row - p * T;
    const float* src = t == 0 ? cls : pe + ((long long)p * (T - 1) + (t - 1)) * D;
    const float* ps = pos + (long long)t * D;
    ln_row([&](int i) { return src[i] + ps[i]; }, D, g, b, nullptr, x + (long long)row * D);
}
void launch_embed_lnpre(const float* patch_emb, const float* cls, const float* pos, const float* g,
                        const float* b, int P, int T, int D, float* x, hipStream_t st) {
    hipLaunchKernelGGL(embed_lnpre_kernel, dim3((P * T + 3) / 4), dim3(256), 0, st, patch_emb, cls, pos, g, b, P,
                       T, D, x);
}

// token + positional embedding of the text tower (clip/model.py:308-310)
__global__ void embed_text_kernel(const int* tokens, const float* tok_emb, const float* pos, int ctx, int D, float* x) {
    const int row = blockIdx.x;
    const float* te = tok_emb + (long long)tokens[row] * D;
    const float* pe = pos + (long long)(row % ctx) * D;
    for (int i = threadIdx.x; i < D; i += blockDim.x) x[(long long)row * D + i] = te[i] + pe[i];
}
void launch_embed_text(const int* tokens, const float* tok_emb, const float* pos, int n_rows, int ctx, int D, float* x,
                       hipStream_t st) {
    hipLaunchKernelGGL(embed_text_kernel, dim3(n_rows), dim3(128), 0, st, tokens, tok_emb, pos, ctx, D, x);
}

__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, long long row_stride, int M, int D,
                                                        const float* g, const float* b, half_t* o16, float* o32) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (long long)row * row_stride;
    ln_row([&](int i) { return xr[i]; }, D, g, b, o16 ? o16 + (long long)row * D : nullptr,
           o32 ? o32 + (long long)row * D : nullptr);
}
void launch_layernorm(const float* x, long long row_stride, int M, int D, const float* g, const float* b,
                      half_t* out16, float* out32, hipStream_t st) {
    hipLaunchKernelGGL(layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, st, x, row_stride, M, D, g, b, out16,
                       out32);
}

// LayerNorm of gathered rows: out[m] = LN(x[rows[m]]) (the text tower's ln_final on each text's EOT row, clip/model.py:316-318)
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* x, const int* rows, int M, int D, const float* g, const float* b, float* o32) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* xr = x + (long long)rows[m] * D;
    ln_row([&](int i) { return xr[i]; }, D, g, b, nullptr, o32 + (long long)m * D);
}
void launch_layernorm_rows(const float* x, const int* rows, int M, int D, const float* g, const float* b, float* out32, hipStream_t st) {
    hipLaunchKernelGGL(layernorm_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, st, x, rows, M, D, g, b, out32);
}

// nn.MultiheadAttention forward for one (image, head) per workgroup: softmax(q k^T / sqrt(hd) [+causal]) v.
// qkv: [n_img*L][3*heads*hd] fp16 (q | k | v), out: [n_img*L][heads*hd] fp16.  hd == 64.
__global__ __launch_bounds__(256) void attention_kernel(const half_t* qkv, int L, int heads, int causal,
                                                        half_t* out) {
    extern __shared__ float sm[];
    const int hd = 64;
    float* q = sm;                 // [L][hd+1]
    float* k = q + L * (hd + 1);   // [L][hd+1]
    float* v = k + L * (hd + 1);   // [L][hd+1]
    float* s = v + L * (hd + 1);   // [L][L+1]
    const int img = blockIdx.x / heads, h = blockIdx.x % heads;
    const int D = heads * hd;
    const half_t* base = qkv + (long long)img * L * 3 * D + h * hd;
    // 16-byte loads, several in flight per thread (2-byte loads with one in flight each made this phase the kernel):
    // piece e -> (token t, which of q|k|v, 8-wide slice d8)
    for (int e0 = threadIdx.x; e0 < L * 24; e0 += 256 * 4) {
        h8 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = min(e0 + 256 * u, L * 24 - 1);
            const int t = e / 24, rem = e - t * 24, which = rem >> 3, d8 = rem & 7;
            r[u] = *(const h8*)(base + (long long)t * 3 * D + which * D + d8 * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 256 * u;
            if (e < L * 24) {
                const int t = e / 24, rem = e - t * 24, which = rem >> 3, d8 = rem & 7;
                float* dst = (which == 0 ? q : which == 1 ? k : v) + t * (hd + 1) + d8 * 8;
                const float sc = which == 0 ? 0.125f : 1.f;   // q * hd^-0.5 (hd = 64)
#pragma unroll
                for (int j = 0; j < 8; ++j) dst[j] = (float)r[u][j] * sc;
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < L * L; e += 256) {
        const int i = e / L, j = e - i * L;
        float a = 0.f;
#pragma unroll 16
        for (int d = 0; d < hd; ++d) a += q[i * (hd + 1) + d] * k[j * (hd + 1) + d];
        if (causal && j > i) a = -INFINITY;
        s[i * (L + 1) + j] = a;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < L; i += 4) {
        float m = -INFINITY;
        for (int j = lane; j < L; j += 64) m = fmaxf(m, s[i * (L + 1) + j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float z = 0.f;
        for (int j = lane; j < L; j += 64) {
            const float e2 = __expf(s[i * (L + 1) + j] - m);
            s[i * (L + 1) + j] = e2;
            z += e2;
        }
        z = wsum(z);
        const float inv = 1.f / z;
        for (int j = lane; j < L; j += 64) s[i * (L + 1) + j] *= inv;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < L * hd; e += 256) {
        const int i = e / hd, d = e - i * hd;
        float a = 0.f;
        for (int j = 0; j < L; ++j) a += s[i * (L + 1) + j] * v[j * (hd + 1) + d];
        out[((long long)img * L + i) * D + h * hd + d] = (half_t)a;
    }
}
// MFMA form for L <= 64 (the visual tower's 50 tokens; the scalar kernel above spends 49 us per layer on LDS-fed fp32 FMAs).
// TWO waves per (image, head): wave qb owns 32 queries and all 64 key slots.
//   S^T[key][query] = K Q^T: both operands are 16-byte global loads of a token's 8 consecutive head dims — no staging;
//   the softmax runs in the accumulator registers (a query's keys live in one lane pair: 32 registers + one xor-32 shuffle);
//   O^T[d][query] = V^T P: the MFMA's K order is free, so P stays where the softmax left it (lane half kh, registers 8g+4kh+q)
//   and V^T goes to LDS once per pair with its keys permuted to match (144-byte rows: conflict-free 16-byte reads);
//   P is split hi + lo * 2^-11 in fp16 so the product keeps the fp32 softmax (fp16 x fp16 products are exact in the fp32
//   accumulator: the scores themselves are the scalar kernel's up to summation order).
// NKB = key (and query) blocks of 32: 2 for L <= 64 — two (image, head) pairs per workgroup, two waves each; 3 for L <= 96 (round 4: the text
// tower's 77 tokens ran on the scalar kernel above, 111 us per layer) — one pair per workgroup, waves 0..2 own 32 queries each, wave 3
// only helps staging V^T.
template <int NKB>
__global__ __launch_bounds__(256) void attention_mfma_kernel(const half_t* qkv, int L, int heads, int n_pairs, int causal,
                                                             half_t* out) {
    constexpr int PP = NKB == 2 ? 2 : 1;                // pairs per workgroup
    constexpr int WP = 4 / PP;                          // waves per pair
    constexpr int VROW = NKB * 64 + 16;                 // bytes per V^T row: 32 NKB keys + pad (144 / 208: conflict-free 16-byte reads)
    __shared__ __attribute__((aligned(16))) char vts[PP][64 * VROW];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 31, kh = lane >> 5;
    const int pair = blockIdx.x * PP + wave / WP, qb = wave % WP;
    const int pc = min(pair, n_pairs - 1);
    const int img = pc / heads, h = pc - img * heads, D = heads * 64;
    const half_t* base = qkv + (long long)img * L * 3 * D + h * 64;
    const long long ts = 3LL * D;                       // token stride
    const int query = qb * 32 + lr;
    const bool qwave = qb < NKB;                        // (NKB = 3: the fourth wave has no queries)
    h8 qf[4], kf[NKB][4];
    if (qwave) {
        const half_t* qp = base + min(query, L - 1) * ts + kh * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const h8*)(qp + kk * 16);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const half_t* kp = base + min(kb * 32 + lr, L - 1) * ts + D + kh * 8;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) kf[kb][kk] = *(const h8*)(kp + kk * 16);
        }
    }
    constexpr int TP = 64 * WP;                         // threads of a pair
    constexpr int NV = NKB * 32 * 8 / TP;               // V vectors per thread (4 / 3)
    const int pt = t % TP;                              // thread within the pair
    h8 vv[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int e = pt + TP * u, tok = e >> 3, d8 = e & 7;
        vv[u] = *(const h8*)(base + min(tok, L - 1) * ts + 2 * D + d8 * 8);
    }
    char* vt = vts[wave / WP];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int e = pt + TP * u, tok = e >> 3, d8 = e & 7;
        // key tok = kb*32 + 8g + 4kh' + q sits at K position (kb*2 + (g>>1))*16 + kh'*8 + (g&1)*4 + q of its row
        const int r = tok & 31, g = r >> 3;
        const int pos = ((tok >> 5) * 2 + (g >> 1)) * 16 + ((r >> 2) & 1) * 8 + (g & 1) * 4 + (r & 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) *(half_t*)(vt + (d8 * 8 + j) * VROW + pos * 2) = tok < L ? vv[u][j] : (half_t)0.f;
    }
    f16x s[NKB];
    float inv = 0.f;
    if (qwave) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) s[kb] = mfma32(kf[kb][kk], qf[kk], s[kb]);
        }
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                float v = s[kb][r] * 0.125f;                // hd^-0.5, hd = 64
                if (key >= L || (causal && key > query)) v = -INFINITY;
                s[kb][r] = v;
                m = fmaxf(m, v);
            }
        m = fmaxf(m, __shfl_xor(m, 32));
        float z = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e2 = __expf(s[kb][r] - m);
                s[kb][r] = e2;
                z += e2;
            }
        z += __shfl_xor(z, 32);
        inv = 1.f / z;
    }
    __syncthreads();                                    // V^T of the workgroup's pairs is in place
    if (!qwave) return;
    f16x o[2], ol[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[db][r] = 0.f; ol[db][r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 2 * NKB; ++ks) {
        h8 ph, pl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float pv = s[ks >> 1][(2 * (ks & 1) + (e >> 2)) * 4 + (e & 3)] * inv;
            ph[e] = (half_t)pv;
            pl[e] = (half_t)((pv - (float)ph[e]) * 2048.f);
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const h8 vf = *(const h8*)(vt + (db * 32 + lr) * VROW + (ks * 16 + kh * 8) * 2);
            o[db] = mfma32(vf, ph, o[db]);
            ol[db] = mfma32(vf, pl, ol[db]);
        }
    }
    if (pair < n_pairs && query < L) {
        half_t* op = out + ((long long)img * L + query) * D + h * 64 + 4 * kh;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h4 w;
#pragma unroll
                for (int q = 0; q < 4; ++q) w[q] = (half_t)(o[db][g * 4 + q] + ol[db][g * 4 + q] * (1.f / 2048.f));
                *(h4*)(op + db * 32 + 8 * g) = w;
            }
    }
}

void launch_attention(const half_t* qkv, int n_img, int L, int heads, int hd, int causal, half_t* out,
                      hipStream_t st) {
    (void)hd;  // 64 (asserted by the engine)
    static const bool no_mfma = glass_knob("GLASS_NO_ATTN_MFMA") != nullptr;   // A/B knob
    if (L <= 64 && !no_mfma) {
        const int n_pairs = n_img * heads;
        hipLaunchKernelGGL(attention_mfma_kernel<2>, dim3((n_pairs + 1) / 2), dim3(256), 0, st, qkv, L, heads, n_pairs, causal, out);
        return;
    }
    if (L <= 96 && !no_mfma) {          // the text tower's context (77)
        const int n_pairs = n_img * heads;
        hipLaunchKernelGGL(attention_mfma_kernel<3>, dim3(n_pairs), dim3(256), 0, st, qkv, L, heads, n_pairs, causal, out);
        return;
    }
    const size_t lds = (size_t)(3 * L * 65 + L * (L + 1)) * sizeof(float);
    static DevOnce once;
    // text tower (L = 77) needs > 64 KiB of the 160 KiB LDS
    once.run([&] { (void)hipFuncSetAttribute((const void*)attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    hipLaunchKernelGGL(attention_kernel, dim3(n_img * heads), dim3(256), lds, st, qkv, L, heads, causal, out);
}

// torch.cosine_similarity(feat[P,D], target[1,D]) (generator.py:51): x.y / max(|x||y|, 1e-8)
__global__ void cosine_kernel(const float* feat, const float* target, int D, float* sim) {
    const int p = blockIdx.x, lane = threadIdx.x;
    float xy = 0.f, xx = 0.f, yy = 0.f;
    for (int i = lane; i < D; i += 64) {
        const float a = feat[(long long)p * D + i], b = target[i];
        xy += a * b; xx += a * a; yy += b * b;
    }
    xy = wsum(xy); xx = wsum(xx); yy = wsum(yy);
    if (lane == 0) sim[p] = xy / fmaxf(sqrtf(xx) * sqrtf(yy), 1e-8f);
}
void launch_cosine(const float* feat, const float* target, int P, int D, float* sim, hipStream_t st) {
    hipLaunchKernelGGL(cosine_kernel, dim3(P), dim3(64), 0, st, feat, target, D, sim);
}

// problem.py:23-27: F = column_stack(-sim, relu(1 - dis)) or F = -sim
__global__ void assemble_F_kernel(const float* sim, const float* dis, int P, int n_obj, float* F) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    F[(long long)p * n_obj] = -sim[p];
    if (n_obj == 2) F[(long long)p * n_obj + 1] = fmaxf(1.f - dis[p], 0.f);
}
void launch_assemble_F(const float* sim, const float* dis, int P, int n_obj, float* F, hipStream_t st) {
    hipLaunchKernelGGL(assemble_F_kernel, dim3((P + 63) / 64), dim3(64), 0, st, sim, dis, P, n_obj, F);
}

// images already resized / normalised by the caller (clip.py:68-74 preprocess) -> patch-embedding operand
__global__ void image_patches_kernel(const float* img, int n, int S, int ps, half_t* patches) {
    const int G = S / ps;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * 3 * S * S) return;
    const int X = (int)(idx % S), Y = (int)((idx / S) % S);
    const int c = (int)((idx / ((long long)S * S)) % 3), b = (int)(idx / ((long long)S * S * 3));
    const int gy = Y / ps, iy = Y - gy * ps, gx = X / ps, ix = X - gx * ps;
    const long long row = ((long long)b * G + gy) * G + gx;
    patches[row * (3LL * ps * ps) + ((long long)c * ps + iy) * ps + ix] = (half_t)img[idx];
}
void launch_image_patches(const float* img, int n, int S, int ps, half_t* patches, hipStream_t st) {
    const long long total = (long long)n * 3 * S * S;
    hipLaunchKernelGGL(image_patches_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, img, n, S, ps, patches);
}
