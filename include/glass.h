/* glass.h — C ABI of the MI355X CLIP-GLaSS fitness-evaluation engine (libglass.so).
 *
 * Drop-in boundary: the reference evaluates a population through
 *   GenerationProblem._evaluate(x, out)            /root/reference/problem.py:14-29
 *     -> Generator.generate / clip_similarity / discriminate   generator.py:29-60
 *     -> models.StyleGAN2.generate / discriminate              models.py:108-129
 * in one Python process on one device.  This library replaces everything below
 * `_evaluate` with one engine object per (process, GPU); the Python shim
 * clip_glass_amd/problem.py keeps the pymoo-facing signature and calls these
 * entry points through ctypes (see INTEGRATION.md for the stub a reference
 * maintainer would add).
 *
 * Conventions: plain pointers and sizes, host buffers owned by the caller, no
 * aliasing retained after return.  Every function returns GLASS_OK (0) or a
 * negative status; glass_last_error() gives the thread-local message.  The
 * engine is not thread-safe; evaluate() is blocking.
 */
#ifndef GLASS_H
#define GLASS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLASS_OK 0
#define GLASS_ERR_ARG -1      /* bad argument / shape (reference: assert, models.py:112,124) */
#define GLASS_ERR_STATE -2    /* call order (weights missing: reference sys.exit(1), models.py:93-101) */
#define GLASS_ERR_HIP -3      /* HIP runtime failure */
#define GLASS_ERR_NOMEM -4

#define GLASS_MAX_BLOCKS 12
#define GLASS_MAX_BG_LAYERS 16
#define GLASS_GEN_STYLEGAN2 0
#define GLASS_GEN_BIGGAN_DEEP 1

typedef struct glass_engine glass_engine;

/* Architecture + semantics.  Mirrors the fields the reference reads from its merged
 * argparse/config Namespace (config.py:80-95; stylegan2/models.py kwargs). */
typedef struct glass_config {
    int32_t device;               /* HIP device ordinal (reference: --device, run.py:17) */
    int32_t n_blocks;             /* number of resolutions, 4*2^(n_blocks-1) px output (9 -> 1024);
                                     0 = no GAN (text-only engine for the GPT2 / img2txt config) */
    int32_t channels[GLASS_MAX_BLOCKS]; /* channels per resolution, LOW -> HIGH res
                                     (reference G order reversed: stylegan2/models.py:748-750) */
    int32_t latent_size;          /* 512 */
    int32_t mapping_layers;       /* 8   (stylegan2/models.py:547) */
    int32_t batch_size;           /* config.batch_size: semantic minibatch — one noise plane per G call
                                     (modules.py:428-452) and mbstd groups per D call (modules.py:726) */
    int32_t mbstd_group;          /* 4   (stylegan2/models.py:1047) */
    int32_t use_discriminator;    /* config.use_discriminator */
    int32_t n_obj;                /* problem_args["n_obj"] (problem.py:21) */
    int32_t max_pop;              /* capacity in candidates (rows of x) */
    int32_t chunk;                /* candidates resident per pass at high resolution (multiple of batch_size; 0 = auto) */
    int32_t clip_width, clip_layers, clip_heads, clip_patch, clip_res, clip_embed; /* 768,12,12,32,224,512 */
    int32_t noise_mode;           /* 0 none, 1 device Philox N(0,1) planes, 2 caller-provided planes */
    uint64_t noise_seed;
    /* --- BigGAN-deep generator (configs DeepMindBigGAN256/512, config.py:31-74; models.py:64-86 calls
     * pytorch-pretrained-biggan's BigGAN.forward(z, class_label, truncation)).  With generator =
     * GLASS_GEN_BIGGAN_DEEP: n_blocks = 0, use_discriminator = 0, latent_size = bg_z_dim + bg_num_classes
     * (one population row = [z | class bits], latent.py:16-18), output = 4 * 2^(#up layers) px. */
    int32_t generator;            /* GLASS_GEN_STYLEGAN2 (0, default) | GLASS_GEN_BIGGAN_DEEP */
    int32_t bg_ch;                /* channel_width (128) */
    int32_t bg_z_dim;             /* config.dim_z (128) */
    int32_t bg_num_classes;       /* config.num_classes (1000) */
    int32_t bg_n_layers;          /* GenBlocks (14 for biggan-deep-512) */
    int32_t bg_layers[GLASS_MAX_BG_LAYERS][3]; /* (up_sample, in_mult, out_mult) per GenBlock */
    int32_t bg_attention_pos;     /* SelfAttn inserted before this GenBlock index (8); -1 = none */
    int32_t bg_n_stats;           /* stored truncation steps of the batch-norm statistics (51) */
    float bg_eps;                 /* batch-norm eps (1e-4) */
    float bg_truncation;          /* config.truncation (1.0): selects / blends the statistics row */
} glass_config;

/* Caller-provided noise (noise_mode 2): planes[m * n_layers + l] points at a host
 * float32 [res_l, res_l] plane for global minibatch m and noise layer l in execution
 * order (== G.static_noise(noise_tensors=...) order, stylegan2/models.py:945-959). */
typedef struct glass_noise {
    int32_t n_minibatches;
    int32_t n_layers;
    const float* const* planes;
} glass_noise;

const char* glass_last_error(void);
const char* glass_version(void);

int glass_engine_create(const glass_config* cfg, glass_engine** out);
void glass_engine_destroy(glass_engine* e);

/* Hand one reference tensor to the engine: `name` is the reference state-dict key
 * prefixed with its sub-model ("G_mapping.", "G_synthesis.", "D.", "clip."), data is
 * host float32, row-major, dims[rank].  Replaces stylegan2.models.load (models.py:183-196)
 * + clip.load/build_model (clip/model.py:363-399).  Tensors are repacked into kernel
 * layouts (pre-scaled, fp16, folded FIR) by glass_engine_finalize. */
int glass_engine_load_tensor(glass_engine* e, const char* name, const float* data,
                             int32_t rank, const int64_t* dims);
int glass_engine_finalize(glass_engine* e);

/* Target text feature, host float32 [clip_embed] (generator.py:23-24: encode_text once). */
int glass_engine_set_target(glass_engine* e, const float* feat, int32_t n);

/* CLIP text tower (clip/model.py:307-320), run once at init by the reference (generator.py:23-24).
 * Available when the text-tower tensors ("clip.token_embedding.weight", "clip.positional_embedding",
 * "clip.transformer.resblocks.*", "clip.ln_final.*", "clip.text_projection") were loaded before
 * finalize().  tokens: host int32 [n_texts, ctx] as produced by clip.tokenize (clip/clip.py:125-138);
 * out_feat: host float32 [n_texts, clip_embed]. */
int glass_engine_encode_text(glass_engine* e, const int32_t* tokens, int32_t n_texts, int32_t ctx, float* out_feat);

/* CLIP image tower on caller-supplied images (generator.py:26-27: the img2txt target), host float32
 * [n,3,clip_res,clip_res] already preprocessed as clip.py:68-74 does; out_feat float32 [n, clip_embed]. */
int glass_engine_encode_image(glass_engine* e, const float* images, int32_t n, float* out_feat);

/* GPT-2 greedy decode (config GPT2 / img2txt: models.py:45-62, gpt2/sample.py:21-36 with sample=False).
 * Needs the "gpt2.transformer.*" tensors (reference GPT2LMHeadModel keys after gpt2/utils.py load_weight).
 * context: host int32 [P, n_ctx_tok] (latent tokens ++ init_text tokens); out: host int32
 * [P, n_ctx_tok + length] = context ++ `length` greedily decoded tokens.  All arithmetic is fp32. */
int glass_engine_gpt2_decode(glass_engine* e, const int32_t* context, int32_t P, int32_t n_ctx_tok, int32_t length,
                             int32_t* out_tokens);

/* THE HOT PATH — replaces GenerationProblem._evaluate (problem.py:14-29).
 * latents: host float32 [P, latent_size] row-major (latent.py:37-38);
 * generation: index folded into the device noise stream (noise_mode 1);
 * first_minibatch: global index of this call's first minibatch (population shards);
 * noise: nullable, used when noise_mode == 2;
 * out_F: host float32 [P, n_obj]: F[:,0] = -cosine, F[:,1] = relu(1 - D) (problem.py:23-27).
 * P must be a multiple of batch_size (reference asserts: models.py:112). */
int glass_engine_evaluate(glass_engine* e, const float* latents, int32_t P, int32_t generation,
                          int32_t first_minibatch, const glass_noise* noise, float* out_F);

/* Extra outputs of the same pass (nullable each): CLIP image features [P, clip_embed],
 * raw discriminator logits [P], cosine similarities [P]. Valid after evaluate(). */
int glass_engine_last_details(glass_engine* e, int32_t P, float* features, float* dis, float* sim);

/* Generator.generate (generator.py:29-34): images host float32 [P,3,R,R] NCHW after
 * biggan_norm (utils.py:14-17).  Used by run.py's callbacks (run.py:45,118). */
int glass_engine_generate(glass_engine* e, const float* latents, int32_t P, int32_t generation,
                          int32_t first_minibatch, const glass_noise* noise, float* images);

/* GPU time of the last evaluate() in ms, from hipEvents on the engine's stream. */
int glass_engine_last_gpu_ms(glass_engine* e, float* ms);
/* Device address of the fitness rows [P][n_obj] float32 of the last evaluate() (the buffer evaluate() copied `out_F` from; valid until the
 * engine's next call).  The multi-GPU host code hands it to the one RCCL all-gather of a generation (clip_glass_amd/parallel.py) without
 * bouncing the rows through host memory. */
int glass_engine_last_F_device(glass_engine* e, int32_t P, void** dev_ptr);

/* Per-kernel profile (hipEvent pairs around every launch while enabled).
 * After evaluate(): n rows of {name, launches, total_ms, flops, bytes} — algorithmic
 * flops/bytes per DESIGN.md.  Used by bench.py for the `roofline` object. */
typedef struct glass_prof_row {
    char name[96];             /* "<layer tag>@<kernel symbol>" */
    int64_t launches;
    double total_ms;
    double flops;
    double bytes;
} glass_prof_row;
int glass_engine_set_profiling(glass_engine* e, int32_t on);
/* Restrict the per-launch events to launches of kernels whose symbol contains `kernel_substr`
 * (as resolved in the previous profiled pass); NULL/"" = every launch.  Keeps the event overhead
 * out of a timed region that only needs the dominant kernel. */
int glass_engine_set_profile_filter(glass_engine* e, const char* kernel_substr);
int glass_engine_get_profile(glass_engine* e, glass_prof_row* rows, int32_t max_rows, int32_t* n_rows);

/* Stream mode of evaluate() — this call is the ONLY control (the library reads no environment variable).  2 (default): generator and
 * discriminator on one stream, CLIP's image tower (short latency-bound launches) on a second, highest-priority stream next to the
 * discriminator.  1: synthesis of chunk k+1 || resize + D + CLIP of chunk k.  0: one stream — the mode for clean per-kernel profiles
 * (co-running kernels stretch each other).  Results are identical in every mode. */
int glass_engine_set_overlap(glass_engine* e, int32_t on);
/* BigGAN-deep diagnostic: record the activation after GenBlock `block` (-1: after the self-attention block, -2: off) of the first
 * chunk of the next evaluate / generate; get_biggan_tap returns it as NHWC float32 [dims[0]][dims[1]][dims[2]][dims[3]] (out may
 * be NULL to query dims).  The package behind models.py:64-86 is absent from the reference tree, so this path is checked
 * block by block against the oracle's restatement: a real checkpoint that mismatches fails at the first wrong block. */
int glass_engine_set_biggan_tap(glass_engine* e, int32_t block);
int glass_engine_get_biggan_tap(glass_engine* e, float* out, int64_t capacity, int32_t dims[4]);

/* Device info for bench.py (CU count, name, HBM bytes). */
int glass_device_info(int32_t device, char* name, int32_t name_len, int32_t* cus, int64_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* GLASS_H */
