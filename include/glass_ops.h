/* glass_ops.h — diagnostic per-kernel entry points of libglass.so.
 *
 * NOT part of the drop-in boundary (that is include/glass.h).  Each function runs ONE
 * device kernel family of the fitness path on host float32 buffers (converted to the
 * kernel's fp16/fp32 layouts internally), so tests/ can compare every kernel against
 * the oracle's corresponding torch op in isolation.  All return GLASS_OK or a negative
 * status (glass_last_error()).  Layouts are NHWC for activations.
 */
#ifndef GLASS_OPS_H
#define GLASS_OPS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct glass_conv_desc {
    int32_t B, H, W, Cin, Cout;
    int32_t KS, stride, pad;
    int32_t up;            /* 1: conv_transpose2d(stride 2) + 4x4 FIR (modules.py:1089-1139), weights folded */
    int32_t Ho, Wo;        /* output size */
    int32_t broadcast_x;   /* x is [1,H,W,Cin], shared by all B (the learned const) */
    int32_t act;           /* leaky-relu 0.2 * sqrt(2) */
    int32_t batch_size;    /* candidates per noise plane */
    int32_t impl;          /* 0 auto, 1 direct, 2 tiled, 3 fused up-conv, 4 streaming 32->32, 5 LDS-DMA, 6 im2col + GEMM (error if unsupported) */
    float noise_strength, out_scale;
    const float* x;        /* [B,H,W,Cin] */
    const float* w;        /* reference layout [Cout,Cin,KS,KS], un-scaled (coef applied inside) */
    const float* sn;       /* [B,Cin] or NULL */
    const float* dscale;   /* [B,Cout] or NULL */
    const float* noise;    /* [B/batch_size,Ho,Wo] or NULL */
    const float* bias;     /* [Cout] or NULL */
    const float* res;      /* [B,Ho,Wo,Cout] or NULL */
    float* y;              /* [B,Ho,Wo,Cout] */
    /* impl 2 / 4 / 5, all or none: toRGB fused into the conv epilogue — trgb_yout is returned; y too by the forms that store the map (2 / 5) */
    const float* trgb_w;     /* [3,Cout] scaled */
    const float* trgb_b;     /* [3] */
    const float* trgb_sn;    /* [B,Cout] normalised toRGB style */
    const float* trgb_smax;  /* [B] */
    const float* trgb_yprev; /* [B,3,Ho/2,Wo/2] or NULL */
    float* trgb_yout;        /* [B,3,Ho,Wo] */
    /* impl 2, stride 2, both or none: the D block's skip branch as extra K stages — y = (act(conv + bias) + conv1x1(skip_x)) * out_scale */
    const float* skip_x;     /* [B,Ho,Wo,Cin] */
    const float* skip_w;     /* [Cout,Cin,1,1] reference layout, un-scaled */
    /* impl 2, 3x3 stride 1, 64 -> 64: FIR 4x4 (pad 1) + ::2 of the INPUT map as a by-product of the staged patch */
    float* xs_out;           /* [B,H/2,W/2,Cin] or NULL */
    /* impl 5, 64 -> 64: x is handed over chunk-planar, [B,Cin/8,H,W,8] (the layout conv_wreg's producers write for it) */
    int32_t x_planar8;
    /* impl 5 with the fused skip branch (conv_s2): x is handed over in 32-channel planes, [B,Cin/32,H,W,32] (the layout the pad-2 blur writes for it) */
    int32_t x_planar32;
} glass_conv_desc;

int glass_op_conv(int32_t device, const glass_conv_desc* d);
/* out[M,N] = epi(a[M,K] @ w[N,K]^T + bias); mode as GemmParams (0 plain,1 quickgelu,2 +=,3 f32,4 lrelu) */
int glass_op_gemm(int32_t device, int32_t M, int32_t N, int32_t K, const float* a, const float* w,
                  const float* bias, int32_t mode, int32_t impl, float* out);
int glass_op_dense(int32_t device, int32_t P, int32_t K, int32_t N, const float* x, const float* wt /*[K,N]*/,
                   const float* bias, int32_t in_sq, int32_t mode, const float* eps_row, float* out);
int glass_op_torgb(int32_t device, int32_t B, int32_t H, int32_t C, const float* x, const float* wrgb /*[3,C] scaled*/,
                   const float* bias, const float* sn, const float* smax, const float* yprev, float* yout);
int glass_op_blur(int32_t device, int32_t mode /*0: pad2 stride1, 1: pad1 + ::2, 2: pad2 stride1 written as [B,C/32,H+1,H+1,32]*/, int32_t B, int32_t H, int32_t C,
                  const float* x, float* out);
/* second half of a discriminator block in one kernel (conv_down.hip; stylegan2/modules.py:1204-1254, 1587-1601):
 * y = (lrelu(conv3x3 stride 2 (fir pad 2 (h)) + b1) * sqrt2 + conv1x1(fir pad 1 (x)[::2])) / sqrt2.
 * h, x [B,R,R,Cin]; w1 [Cout,Cin,3,3], wskip [Cout,Cin,1,1] (reference layouts, un-scaled); y [B,R/2,R/2,Cout] */
int glass_op_dblock_down(int32_t device, int32_t B, int32_t R, int32_t Cin, int32_t Cout, const float* h, const float* x,
                         const float* w1, const float* wskip, const float* b1, float* y);
/* the discriminator's whole full-resolution block (conv_d0.hip; stylegan2/models.py:1125-1143, modules.py:1204-1254, 1587-1601):
 * y [B,3,R,R] skip image -> denorm(norm(y)) -> fromRGB (3 -> 32) -> conv3x3 (32 -> 32) -> FIR pad 2 -> conv3x3 stride 2 (32 -> 64),
 * + conv1x1 of FIR pad 1 [::2] of the fromRGB map, merged / sqrt2.  frgb_w [32,3] scaled; w0 [32,32,3,3], w1 [64,32,3,3],
 * wskip [64,32,1,1] reference layouts, un-scaled; out [B,R/2,R/2,64].  impl 0: the fused kernel; 1: conv_stream<fromrgb> + conv_down;
 * 2: the fused kernel writing out chunk-planar, [B,8,R/2,R/2,8] */
int glass_op_dblock0(int32_t device, int32_t B, int32_t R, int32_t impl, const float* y, const float* frgb_w, const float* frgb_b,
                     const float* w0, const float* b0, const float* w1, const float* wskip, const float* b1, float* out);
int glass_op_fromrgb(int32_t device, int32_t B, int32_t R, int32_t Cout, const float* y /*[B,3,R,R]*/,
                     const float* w /*[Cout,3] scaled*/, const float* bias, float* out /*[B,R,R,Cout]*/);
int glass_op_mbstd(int32_t device, int32_t B, int32_t hw, int32_t C, int32_t Cpad, int32_t batch_size, int32_t group,
                   const float* x, float* out);
int glass_op_resize(int32_t device, int32_t B, int32_t R, int32_t S, int32_t ps, const float* y /*[B,3,R,R]*/,
                    float* patches /*[B*G*G, 3*ps*ps]*/);
int glass_op_layernorm(int32_t device, int32_t M, int32_t D, const float* x, const float* g, const float* b, float* out);
int glass_op_attention(int32_t device, int32_t n_img, int32_t L, int32_t heads, int32_t causal, const float* qkv,
                       float* out);
int glass_op_noise(int32_t device, int32_t n_mb, int32_t hw, uint32_t layer, uint32_t mb0, uint32_t generation,
                   uint64_t seed, float* out);
/* raw MFMA layout probe: D = A[32x16] * B[16x32] through the fragment mapping of common.h */
int glass_op_mfma_probe(int32_t device, const float* a /*[32,16]*/, const float* b /*[16,32]*/, float* d /*[32,32]*/);

/* Host-only (no GPU): the weight repacking finalize() applies, for CPU tests.
 * out: [KS*KS][Neff][Cin] float32 (values already rounded to fp16), Neff = up ? 4*Cout : Cout. */
int glass_host_pack_conv(const float* w, int32_t Cout, int32_t Cin, int32_t KS, int32_t up, float* out);

#ifdef __cplusplus
}
#endif
#endif
