// engine.h — host-side engine object behind the C ABI (include/glass.h).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/glass.h"
#include "common.h"

void glass_set_error(const std::string& s);
#define GLASS_HIP(call)                                                                        \
    do {                                                                                       \
        hipError_t _e = (call);                                                                \
        if (_e != hipSuccess) {                                                                \
            glass_set_error(std::string(#call) + " failed: " + hipGetErrorString(_e) + " at " + \
                            __FILE__ + ":" + std::to_string(__LINE__));                        \
            return GLASS_ERR_HIP;                                                              \
        }                                                                                      \
    } while (0)

struct HostTensor {
    std::vector<int64_t> dims;
    std::vector<float> data;
};

struct GConv {  // one modulated 3x3 conv of the synthesis network
    int cin, cout, res_in, res_out, up;
    int style_idx, style_off, ds_off, noise_idx;
    half_t* w = nullptr;   // [9][Neff][cin]
    half_t* w_up = nullptr;  // up layers: un-folded [9][cout][cin]
    half_t* wm = nullptr;    // pre-modulated per-sample weights [P][welems] (small high-res layers only)
    long long welems = 0;
    bool premod = false;
    float* wsq = nullptr;  // [cin][cout]
    float* bias = nullptr;
    float noise_strength = 0.f;
};
struct GRgb {
    int cin, res, style_idx, style_off;
    float* w = nullptr;  // [3][cin]
    float* bias = nullptr;
};
struct DBlock {
    int cin, cout, res;
    half_t *w0 = nullptr, *w1 = nullptr, *wskip = nullptr;
    float *b0 = nullptr, *b1 = nullptr;
};
struct ClipBlock {
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    half_t *w_qkv, *w_out, *w_fc, *w_proj;
    float *b_qkv, *b_out, *b_fc, *b_proj;
};

// BigGAN-deep generator (biggan.cpp)
struct BgBlock {   // GenBlock: bn0-relu-conv1x1 / bn1-relu-[up]-conv3x3 / bn2-relu-conv3x3 / bn3-relu-conv1x1 + skip
    int cin, cout, mid, up, res_in;
    int bn_off[4];           // column offsets of the four batch norms in the affine tables
    half_t* w[4] = {nullptr, nullptr, nullptr, nullptr};   // [ks*ks][cout][cin] fp16, spectral norm folded
    float* b3 = nullptr;     // conv_3 bias (the other biases are folded into the next BN's shift)
};
struct BgState {
    int R = 0, Ctot = 0, zd = 0, nc = 0, c0 = 0;
    float *et = nullptr, *genz_wt = nullptr, *genz_b = nullptr;
    float *bn_wt = nullptr, *bn_bias = nullptr, *bn_inv_std = nullptr, *bn_mean = nullptr, *bn_prebias = nullptr;
    std::vector<BgBlock> blocks;
    int attn_before = -1, attn_C = 0, attn_res = 0;
    half_t *attn_w_tpg = nullptr, *attn_w_o = nullptr;
    int final_bn_off = 0, rgb_cpad = 32;
    half_t* rgb_w = nullptr;
    float* rgb_b = nullptr;
    // activations
    half_t* tab16 = nullptr;   // fp16 copy of tab
    float *cond = nullptr, *tab = nullptr, *h32 = nullptr, *a_S = nullptr;
    half_t *x[2] = {nullptr, nullptr}, *t1 = nullptr, *t2 = nullptr, *t3 = nullptr;
    half_t *a_T = nullptr, *a_theta = nullptr, *a_phi = nullptr, *a_gT = nullptr, *a_P = nullptr, *a_O = nullptr;
};
int glass_biggan_finalize(glass_engine* e);       // weights -> device layouts + activation buffers
int glass_biggan_prepare(glass_engine* e, int P); // cond vectors + batch-norm tables for the population in d_z
// synthesize candidates [c0, c0 + B) -> planar fp32 RGB in (-1, 1) at `y` [B][3][R][R]
int glass_biggan_chunk(glass_engine* e, int c0, int B, float* y);

struct ProfEvent {
    std::string name;
    double flops, bytes;
    hipEvent_t e0, e1;
};

struct glass_engine {
    glass_config cfg;
    int R = 0;        // output resolution
    int n_style = 0;  // number of style (affine) layers
    int S_total = 0, D_total = 0;
    int chunk = 0;
    bool finalized = false, has_target = false;
    hipStream_t stream = nullptr, stream_d = nullptr, cur = nullptr;  // main, second (D/resize), current target
    bool overlap = false, clip_overlap = false;
    std::vector<hipEvent_t> ev_g, ev_d;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_noise = nullptr, ev_rgb = nullptr;
    float last_ms = 0.f;
    int last_P = 0;
    std::string launch_error;   // a launcher refused a layer during the pass (reported by run_pass; nothing aborts)

    std::map<std::string, HostTensor> host;
    std::vector<void*> allocs;

    // ---- G ----
    std::vector<float*> map_wt, map_b;  // [L][L] transposed, pre-scaled
    float *style_wt = nullptr, *style_b = nullptr;  // [L][S_total], [S_total]
    int *d_style_off = nullptr, *d_style_len = nullptr;
    std::vector<int> style_off, style_len;
    half_t* g_const = nullptr;  // [4][4][C0]
    void* d_demod_desc = nullptr;  // DenseDesc[gconv.size()] for the single-launch demodulation
    int demod_max_n = 0;
    std::vector<GConv> gconv;
    std::vector<GRgb> grgb;
    BgState bg;
    // ---- D ----
    float *d_frgb_w = nullptr, *d_frgb_b = nullptr;
    std::vector<DBlock> dblk;
    half_t* d_final_w = nullptr;
    float* d_final_b = nullptr;
    int d_final_cpad = 0;
    half_t* d_dense0_w = nullptr;
    float *d_dense0_b = nullptr, *d_dense1_wt = nullptr, *d_dense1_b = nullptr;
    // ---- CLIP ----
    half_t* c_patch_w = nullptr;
    float *c_cls = nullptr, *c_pos = nullptr, *c_lnpre_g = nullptr, *c_lnpre_b = nullptr;
    float *c_lnpost_g = nullptr, *c_lnpost_b = nullptr, *c_proj = nullptr;
    std::vector<ClipBlock> cblk;
    // GPT-2 (optional, fp32; config C5)
    struct Gpt2Block { float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *w_qkv, *b_qkv, *w_o, *b_o, *w_fc, *b_fc, *w_pr, *b_pr; };
    std::vector<Gpt2Block> gblk;
    float *g_wte = nullptr, *g_wpe = nullptr, *g_lnf_g = nullptr, *g_lnf_b = nullptr;
    int g_vocab = 0, g_dim = 0, g_npos = 0;
    // decode workspace, kept between calls (one geometry at a time): buffers + the captured single-token step
    struct Gpt2Work {
        int P = 0, nctx = 0, length = 0;
        int *d_tok = nullptr, *d_gen = nullptr, *d_state = nullptr;
        float *x = nullptr, *ln = nullptr, *qkv = nullptr, *att = nullptr, *hid = nullptr, *last = nullptr, *logits = nullptr,
              *kc = nullptr, *vc = nullptr, *part = nullptr, *stats = nullptr, *pairs = nullptr, *pst = nullptr;   // stats: [P][2] LayerNorm {mean, rstd} of the fused step
        size_t part_elems = 0;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        float last_ms = 0.f;      // device time of the last decode (hipEvents around the passes)
    } gwork, gwork_alt;       // current workspace + the previous geometry's (glass_engine_gpt2_decode's row groups: 64 rows and a remainder)
    // text tower (optional)
    std::vector<ClipBlock> tblk;
    float *t_tok = nullptr, *t_pos = nullptr, *t_lnf_g = nullptr, *t_lnf_b = nullptr, *t_proj = nullptr;
    struct TextWork {      // encode_text's activations, kept between calls of the same size (eight hipMalloc / hipFree pairs per call otherwise)
        int n_texts = 0;
        int *d_tok = nullptr, *d_rows = nullptr;
        float *x = nullptr, *cls = nullptr, *feat = nullptr;
        half_t *ln16 = nullptr, *qkv = nullptr, *att = nullptr, *hid = nullptr;
    } twork;
    int t_width = 0, t_ctx = 0, t_vocab = 0;
    float* d_target = nullptr;

    // ---- activations / scratch ----
    half_t* d_s16 = nullptr;   // fp16 copy of d_s (normalised styles)
    half_t* ws_a = nullptr;   // conv_gemm.hip scratch: patch matrix of a low-resolution layer (cap_a halfs) and its fp32 product (cap_c)
    float* ws_c = nullptr;
    half_t* ws_a2 = nullptr;  // second scratch set: launches on the second stream never share a buffer with the main stream's
    float* ws_c2 = nullptr;
    long long cap_a = 0, cap_c = 0;
    half_t* d_trgb_tab = nullptr;   // [P][2][16][<= 512] fp16: toRGB weight tables of the fused conv epilogues
    float* d_trgb_part = nullptr;   // [C / 128][chunk][3][res][res]: toRGB partial sums of the blocks wider than 128 channels
    float *d_z = nullptr, *d_w0 = nullptr, *d_w1 = nullptr, *d_s = nullptr, *d_smax = nullptr, *d_epsrow = nullptr,
          *d_dscale = nullptr;
    std::vector<float*> d_noise;  // per noise layer: [n_mb_max][res*res]
    half_t* act[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // per-chunk, high resolution
    half_t* low[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // whole population, res <= low_res
    float* ylow[2] = {nullptr, nullptr};
    int low_res = 32, n_low = 0;
    size_t act_elems = 0;
    float* ybuf[4] = {nullptr, nullptr, nullptr, nullptr};
    float* d_img = nullptr;
    half_t *d_patches = nullptr, *d_ln16 = nullptr, *d_qkv = nullptr, *d_attn = nullptr, *d_hid = nullptr;
    float *d_pe = nullptr, *d_x = nullptr, *d_cls = nullptr, *d_feat = nullptr, *d_sim = nullptr, *d_dis = nullptr,
          *d_F = nullptr, *d_dh = nullptr, *d_dh_part = nullptr;
    half_t* d_dfin = nullptr;
    float* h_pinned = nullptr;
    size_t h_pinned_bytes = 0;

    // ---- BigGAN per-block tap (diagnostic: localises a mismatch to the first wrong GenBlock) ----
    int bg_tap = -2;                       // -2 off; -1 self-attention output; i >= 0: output of GenBlock i
    std::vector<float> bg_tap_data;        // [B][R][R][C] NHWC of the first chunk
    int bg_tap_dims[4] = {0, 0, 0, 0};
    // ---- profiling ----
    bool profiling = false;
    std::string prof_filter;                       // non-empty: only launches of kernels containing this substring
    std::map<std::string, std::string> tag_kernel;  // layer tag -> kernel symbol used in the previous pass
    std::vector<ProfEvent> prof_events;
    std::vector<glass_prof_row> prof_rows;
    std::vector<hipEvent_t> event_pool;
    size_t event_next = 0;
};

// ------------------------------------------------------------------------------------
// helpers shared by engine.cpp and biggan.cpp
// ------------------------------------------------------------------------------------
#define REQUIRE(cond, code, msg)          \
    do {                                  \
        if (!(cond)) {                    \
            glass_set_error(msg);         \
            return code;                  \
        }                                 \
    } while (0)

// ------------------------------------------------------------------------------------
// allocation / upload helpers
// ------------------------------------------------------------------------------------
template <typename T>
inline int dev_alloc(glass_engine* e, T** p, size_t n) {
    void* q = nullptr;
    hipError_t err = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
    if (err != hipSuccess) {
        glass_set_error(std::string("hipMalloc failed: ") + hipGetErrorString(err));
        return GLASS_ERR_NOMEM;
    }
    e->allocs.push_back(q);
    *p = (T*)q;
    return GLASS_OK;
}
template <typename T>
inline int upload(glass_engine* e, T** p, const std::vector<T>& v) {
    int rc = dev_alloc(e, p, v.size());
    if (rc) return rc;
    GLASS_HIP(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return GLASS_OK;
}

inline const HostTensor* find(glass_engine* e, const std::string& name) {
    auto it = e->host.find(name);
    return it == e->host.end() ? nullptr : &it->second;
}
#define GET(var, name)                                                            \
    const HostTensor* var = find(e, name);                                        \
    REQUIRE(var != nullptr, GLASS_ERR_STATE, std::string("missing tensor: ") + (name))

inline size_t numel(const HostTensor* t) {
    size_t n = 1;
    for (auto d : t->dims) n *= (size_t)d;
    return n;
}

// profiling scope (hipEvent pair per launch when enabled)
struct Prof {
    glass_engine* e;
    bool on;
    ProfEvent pe;
    Prof(glass_engine* e_, const char* name, double flops, double bytes) : e(e_), on(e_->profiling) {
        if (on && !e->prof_filter.empty()) {   // only launches whose kernel symbol (as of the previous pass) matches
            auto it = e->tag_kernel.find(name);
            on = it != e->tag_kernel.end() && it->second.find(e->prof_filter) != std::string::npos;
        }
        if (!on) return;
        auto get = [&]() {
            if (e->event_next == e->event_pool.size()) {
                hipEvent_t ev;
                hipEventCreate(&ev);
                e->event_pool.push_back(ev);
            }
            return e->event_pool[e->event_next++];
        };
        pe.name = name;
        pe.flops = flops;
        pe.bytes = bytes;
        pe.e0 = get();
        pe.e1 = get();
        hipEventRecord(pe.e0, e->cur);
    }
    ~Prof() {
        if (!on) return;
        hipEventRecord(pe.e1, e->cur);
        e->prof_events.push_back(pe);
    }
};

void collect_profile(glass_engine* e);
void run_conv(glass_engine* e, const ConvParams& p, const char* tag, double flops, double bytes);
void run_gemm(glass_engine* e, const GemmParams& p, const char* tag);
ConvParams conv_defaults();
void run_clip(glass_engine* e, int P, int l0, int l1);
// Stream mode 2: the patch embedding and this many layers of CLIP's image tower run on the MAIN stream, alone on the chip, before the
// discriminator starts; the rest of the tower runs on the second stream beside it.  Forked right behind the resize (rounds 2-5), the tower's
// first launch raced the discriminator's first kernel — a persistent kernel that fills every CU for 3.9 ms — and lost about every other
// process: 31.0-31.5 ms per pass against 29.6 with the embedding and ONE layer in front (sweep 0 / 1 / 2 / 3 / 4 / 6 / 8 / 10 / 12 layers:
// 29.66 / 29.59 / 29.60 / 29.74 / 29.79 / 29.91 / 29.96 / 30.13 / 30.53 ms, three runs each within 0.1 ms; DESIGN section 5 "Round 6").
#define GLASS_CLIP_SERIAL_LAYERS 1
std::vector<_Float16> to_half(const float* p, size_t n, float scale = 1.f);
std::vector<float> scaled(const float* p, size_t n, float scale);
std::vector<float> transposed(const float* W, int N, int K, float coef);

// engine_ops.cpp helpers shared with the diagnostic op ABI
int glass_fold_upconv(const float* W, int cout, int cin, std::vector<_Float16>& out);  // [9][4*cout][cin]
void glass_pack_conv(const float* W, int cout, int cin, int ks, int cin_pad, std::vector<_Float16>& out);  // [ks*ks][cout][cin_pad]
