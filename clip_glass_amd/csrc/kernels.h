// kernels.h — host-callable launch wrappers (one per kernel family).
#pragma once
#include "common.h"

// Each launcher returns the kernel symbol it launched (for the per-kernel profile).
const char* launch_conv_direct(const ConvParams& p, hipStream_t st);
const char* launch_gemm_direct(const GemmParams& p, hipStream_t st);
// LDS-tiled fast paths; return nullptr when the shape is not supported (caller falls back to direct).
const char* launch_conv_tiled(const ConvParams& p, hipStream_t st);
const char* launch_gemm_tiled(const GemmParams& p, hipStream_t st);
// persistent streaming 3x3 conv for the 32 -> 32 channel 1024^2 layers (conv_stream.hip); nullptr when unsupported
const char* launch_conv_stream(const ConvParams& p, hipStream_t st);
// the low-resolution layers as im2col + gemm_tiled + finishing pass (conv_gemm.hip); ws_a / ws_c: scratch of cap_a halfs / cap_c floats PER CANDIDATE
const char* launch_conv_gemm(const ConvParams& p, half_t* ws_a, long long cap_a, float* ws_c, long long cap_c, hipStream_t st);
bool conv_stream_applies(const ConvParams& p);   // trgb_yout set: the fused conv + toRGB form
// LDS-DMA staged 3x3 conv for the MFMA-bound mid-resolution layers (conv_glds.hip); nullptr when unsupported / disabled
const char* launch_conv_glds(const ConvParams& p, hipStream_t st, bool force = false);
// conv_wreg.hip: 3x3 stride-1 conv 64 -> 64 channels with the WHOLE weight tensor in registers, one wave per SIMD, the patches on a three-tile
// LDS-DMA ring (tried first by launch_conv_glds); reads pixel-major or chunk-planar input; nullptr: the layer does not qualify
const char* launch_conv_wreg(const ConvParams& p, hipStream_t st);
bool conv_wreg_supported(int Cin, int Cout, int H, int W);
// conv_s2.hip: the D blocks' stride-2 3x3 conv + fused 1x1 skip branch on an LDS-DMA ring (nullptr: not applicable -> conv_tiled)
const char* launch_conv_s2(const ConvParams& p, hipStream_t st, bool force = false);
// second half of the full-resolution discriminator block in one kernel (conv_down.hip):
//   y = (lrelu(conv3x3 stride 2 (fir_pad2(h)) + b1) * sqrt2 + conv1x1(xs)) / sqrt2, xs = fir_pad1(x)[::2] (32 -> 64 channels);
// nullptr when the shape does not qualify (caller runs the separate passes)
const char* launch_conv_down(const half_t* h, const half_t* xs, const half_t* w1, const half_t* ws, const float* b1, half_t* y,
                             int B, int R, int Cin, int Cout, hipStream_t st);
bool conv_down_supported(int R, int Cin, int Cout);
// the WHOLE full-resolution discriminator block in one kernel (conv_d0.hip): skip image -> fromRGB -> conv3x3 32 -> 32 -> FIR (pad 2) ->
// conv3x3 stride 2 32 -> 64, + the 1x1 skip branch of FIR (pad 1)[::2] of the fromRGB map, merged; x and h never leave the CU.
// nullptr when the block does not qualify (caller: conv_stream<fromrgb> + conv_down)
const char* launch_dblock0(const float* rgb_y, const float* rgb_w, const float* rgb_b, const half_t* w0, const float* b0, const half_t* w1,
                           const half_t* ws, const float* b1, half_t* y, int B, int R, int Cin, int Cout, hipStream_t st, int y_planar8 = 0);
bool dblock0_supported(int R, int Cin, int Cout);
// fused transposed-conv + FIR + epilogue (upfir.hip); nullptr when unsupported
const char* launch_upconv_fused(const ConvParams& p, hipStream_t st);

// --- small fp32 ops (mapping network, style affines, demodulation, heads) ---------
void launch_pixelnorm(const float* z, float* out, int P, int L, float eps, hipStream_t st);
// out[p][n] = epi( sum_k f(x[p][k]) * wt[k][n] + bias[n] ); in_sq: f = square;
// mode 0 none, 1 lrelu*sqrt2, 2 rsqrt(v + eps_row[p*eps_stride])
void launch_dense01_finish(const float* part, int S, long long slab, const float* bias0, const float* w1, const float* b1, float* out, int P, int N,
                           hipStream_t st);
void launch_dense_splitk(const float* x, int ldx, int P, int K, const float* wt, int N, const float* bias, float* out, int ldo,
                         int mode, hipStream_t st);   // K % 64 == 0, K <= 768: the mapping-network layers
void launch_dense(const float* x, int ldx, int P, int K, const float* wt, int N, const float* bias,
                  float* out, int ldo, int in_sq, int mode, const float* eps_row, int eps_stride,
                  hipStream_t st);
// pixel norm + every mapping layer in one launch (L = 256 / 512, <= 8 layers); false: not applicable, run the per-layer path
bool launch_mapping_fused(const float* z, float* out, int P, int L, float eps, const float* const* wt, const float* const* b, int n_layers,
                          hipStream_t st);
struct DenseDesc {
    const float* x; int ldx; int K; const float* wt; int N; const float* bias; float* out; int ldo;
    const float* eps_row; int eps_stride;
};
void launch_dense_multi(const DenseDesc* d_desc, int n_desc, int max_N, int P, int in_sq, int mode, hipStream_t st);
// per (p, layer): smax = max|s|, s /= smax, eps_row = eps / smax^2
void launch_style_norm(float* s, int ld, int P, int n_layers, const int* d_off, const int* d_len,
                       float* smax, float* eps_row, float eps, hipStream_t st);
// per-sample modulated+demodulated weights: wm[b][e] = w[e] * sn[b][e % Cin] * dscale[b][(e / Cin) % Cout]
void launch_modulate_weights(const half_t* w, long long elems, int Cin, int Cout, const float* sn, int sn_stride,
                              const float* dscale, int ds_stride, int P, half_t* wm, hipStream_t st);
void launch_noise(float* out, int n_mb, int hw, uint32_t layer, uint32_t mb0, uint32_t generation,
                  uint64_t seed, hipStream_t st);

// --- image-space ops ---------------------------------------------------------------
// y[b][c][p] = bias[c] + smax[b]*sum_i wrgb[c][i]*sn[b][i]*x[b][p][i] + upfir(yprev)
void launch_trgb_tables(const float* wrgb, const float* sn, int sn_stride, const float* smax, int smax_stride, int B, int NT,
                        half_t* tab, hipStream_t st);   // [B][2][16][NT] fp16: A operand of the toRGB fused into a conv epilogue
// skip image from the per-n-tile toRGB partial sums of a conv epilogue (ConvParams::trgb_part [ntn][B][3][R][R])
void launch_trgb_finish(const float* part, int ntn, int B, int R, const float* bias, const float* yprev, float* yout, hipStream_t st);
// false: channel width not instantiated (16 ... 512 in powers of two) — nothing launched
bool launch_torgb(const half_t* x, int B, int H, int W, int C, const float* wrgb, const float* bias,
                  const float* sn, int sn_stride, const float* smax, int smax_stride,
                  const float* yprev, float* yout, hipStream_t st);
// img = clip((y+1)/2, 0, 1)
void launch_finalize_image(const float* y, float* img, long long n, hipStream_t st);
// bilinear (align_corners=False) resize of clip((y+1)/2,0,1) into the patch matrix [B*G*G][3*ps*ps] fp16
void launch_resize_patches(const float* y, int B, int R, int clip_res, int ps, half_t* patches,
                           hipStream_t st);
void launch_fromrgb(const float* y, int B, int R, int Cout, const float* w, const float* bias,
                    half_t* out, hipStream_t st);
// 4x4 FIR [1,3,3,1]^2/64, zero pad 2, stride 1: [B,H,W,C] -> [B,H+1,W+1,C]
void launch_blur_pad2(const half_t* x, int B, int H, int W, int C, half_t* out, hipStream_t st, int planar32 = 0);   // planar32: [C/32][H+1][W+1][32] for conv_s2
bool blur_pad2_planar32_ok(int C);      // the channel counts that form exists for
// 4x4 FIR, zero pad 1, then ::2 subsample: [B,H,W,C] -> [B,H/2,W/2,C]
void launch_blur_down(const half_t* x, int B, int H, int W, int C, half_t* out, hipStream_t st);
// minibatch-std (reference quirk: features are group-mean subtracted): [B,hw,C] -> [B,hw,Cpad]
void launch_mbstd(const half_t* x, int B, int hw, int C, int Cpad, int batch_size, int group, float eps,
                  half_t* out, hipStream_t st);

// --- CLIP --------------------------------------------------------------------------
void launch_embed_lnpre(const float* patch_emb, const float* cls, const float* pos, const float* g,
                        const float* b, int P, int T, int D, float* x, hipStream_t st);
// x[n*ctx + t][:] = tok_emb[tokens[n*ctx+t]][:] + pos[t][:]
void launch_embed_text(const int* tokens, const float* tok_emb, const float* pos, int n_rows, int ctx, int D, float* x,
                       hipStream_t st);
void launch_layernorm(const float* x, long long row_stride, int M, int D, const float* g, const float* b,
                      half_t* out16, float* out32, hipStream_t st);
void launch_layernorm_rows(const float* x, const int* rows, int M, int D, const float* g, const float* b, float* out32, hipStream_t st);
void launch_attention(const half_t* qkv, int n_img, int L, int heads, int hd, int causal, half_t* out,
                      hipStream_t st);
void launch_cosine(const float* feat, const float* target, int P, int D, float* sim, hipStream_t st);
void launch_assemble_F(const float* sim, const float* dis, int P, int n_obj, float* F, hipStream_t st);

// --- BigGAN-deep glue (biggan_kernels.hip) ---------------------------------------------------------
void launch_bg_cond(const float* x, int P, int L, int zd, int nc, const float* et, float* cond, hipStream_t st);
void launch_bg_bn_tables(float* tab, int P, int C, const float* inv_std, const float* mean, const float* prebias,
                         hipStream_t st);
void launch_bg_to_half(const float* x, half_t* y, long long n, hipStream_t st);
void launch_bg_attn_split(const half_t* T, int B, int H, int W, int c8, int c2, half_t* theta, half_t* phi, half_t* gT,
                          hipStream_t st);
void launch_bg_softmax(const float* S, long long rows, int n, half_t* Pm, hipStream_t st);
void launch_bg_rgb_tanh(const half_t* x, int B, long long hw, int C, float* y, hipStream_t st);

// --- BigGAN-deep's last stage in one kernel (bg_tail.hip): conv_3 + skip -> bn -> relu -> conv_to_rgb[:3] -> tanh ---------------
struct BgTailParams {
    const half_t* h;        // [B][R][R][32]: relu(bn_3(conv_2)) of the last block (conv_2's epilogue applied them)
    const half_t* x0;       // [B][R/2][R/2][128]: the block's input (skip source, nearest x2)
    const half_t* w3;       // [128][32] conv_3 weights (spectral norm folded)
    const float* b3;        // [128]
    const float* tab;       // this chunk's first row of the batch-norm affine table: scale at [bnf_off + c], shift at [ctot + bnf_off + c]
    int bnf_off, ctot;
    const half_t* rgb_w;    // [9][cpad][128] conv_to_rgb weights (rows 0..2 of each tap used)
    int cpad;
    const float* rgb_b;     // [>= 3]
    float* y;               // [B][3][R][R] fp32
    int B, R;
};
bool bg_tail_supported(int R, int mid, int cout, int cin, int up, int cpad);
bool launch_bg_tail(const BgTailParams& p, hipStream_t st);
// --- GPT-2 (fp32, gpt2.hip) ----------------------------------------------------------
void launch_gpt2_embed(const int* tok, const float* wte, const float* wpe, int rows, int L, int pos0, int D, float* x,
                       hipStream_t st);
// part / part_elems: split-K scratch for the single-token (M <= 64) steps (nullptr = never split)
// prefill: the rows are P sequences x nd > 1 positions — always the tiled kernels, so that the kernel (and with it the summation order)
// a sequence's prefill runs on does not depend on how many sequences the launch holds
void launch_gemm_f32(const float* A, const float* W, const float* bias, float* out, int M, int N, int K, int lda, int ldo,
                     int mode, hipStream_t st, float* part = nullptr, size_t part_elems = 0, bool prefill = false);
bool gemm_f32_step_supported(int M, int K, int lda, bool ln_fused);   // launch_gemm_f32_step's shape conditions
// fused single-token step (round 3): products with LayerNorm on the activation operand / complete outputs, and the residual + statistics pass
int launch_gemm_f32_step(const float* A, const float* W, const float* bias, float* out, int M, int N, int K, int lda, int ldo, int mode,
                         hipStream_t st, float* part, size_t part_elems, const float* stats, const float* lng, const float* lnb);
bool gpt2_head_supported(int M, int N, int K, int lda);
// vocabulary projection + pick + the next step's embedding / first LayerNorm statistics + state advance (state[2] = ticket counter, zero)
bool launch_gpt2_head_tail(const float* A, const float* W, int M, int N, int K, int lda, const float* stats_in, const float* lng, const float* lnb,
                           float* pairs, int* gen, int* state, const float* wte, const float* wpe, float* x, float* stats_out, hipStream_t st);
bool launch_gpt2_head(const float* A, const float* W, int M, int N, int K, int lda, const float* stats, const float* lng, const float* lnb,
                      float* logits, float* pairs, int* out, const int* step_dev, hipStream_t st);
void launch_gpt2_reduce(const float* part, int S, const float* bias, float* out, int M, int N, int ldo, int mode, hipStream_t st);
void launch_gpt2_finalize(const float* part, int S, const float* bias, float* x, int M, int D, float* stats, hipStream_t st);
// past_dev / step_dev: device-resident step state {past length, step index} for the captured single-token step
void launch_gpt2_attention(const float* qkv, float* kc, float* vc, int P, int nd, int past, int Tmax, int heads,
                           float* out, hipStream_t st, const int* past_dev = nullptr);
void launch_gpt2_attention_step(const float* qkv, const float* part, int S, const float* bias, float* kc, float* vc, int P, int Tmax, int heads,
                                float* out, hipStream_t st, const int* past_dev);
void launch_argmax(const float* logits, int rows, int N, int* out, hipStream_t st, const int* step_dev = nullptr, float* scratch = nullptr);
void launch_gpt2_embed_step(const int* gen, const int* state, int P, const float* wte, const float* wpe, int D, float* x, hipStream_t st,
                            float* stats = nullptr, bool partial_fmt = false);
// complete-output step product: a workgroup = 32 rows x 32 columns over the whole K (gpt2.hip); false = shape not covered, nothing launched
bool gemm_f32_rowblk_supported(int M, int N, int K, int lda, bool ln_fused, bool stats_out);
bool launch_gemm_f32_rowblk(const float* A, const float* W, const float* bias, float* out, int M, int N, int K, int lda, int ldo, int mode,
                            hipStream_t st, const float* pst_in, int np_in, const float* lng, const float* lnb, float* pst_out);
void launch_gpt2_advance(int* state, hipStream_t st);
// NCHW fp32 image [n][3][S][S] -> CLIP patch matrix [n*G*G][3*ps*ps] fp16
void launch_image_patches(const float* img, int n, int S, int ps, half_t* patches, hipStream_t st);
