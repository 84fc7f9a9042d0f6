// engine.h — host-side engine object behind the C ABI (include/glass.h).
#pragma once
#include <hip/hip_runtime.h>

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
    bool overlap = false;
    std::vector<hipEvent_t> ev_g, ev_d;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.f;
    int last_P = 0;

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
    // text tower (optional)
    std::vector<ClipBlock> tblk;
    float *t_tok = nullptr, *t_pos = nullptr, *t_lnf_g = nullptr, *t_lnf_b = nullptr, *t_proj = nullptr;
    int t_width = 0, t_ctx = 0, t_vocab = 0;
    float* d_target = nullptr;

    // ---- activations / scratch ----
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
          *d_F = nullptr, *d_dh = nullptr;
    half_t* d_dfin = nullptr;
    float* h_pinned = nullptr;
    size_t h_pinned_bytes = 0;

    // ---- profiling ----
    bool profiling = false;
    std::string prof_filter;                       // non-empty: only launches of kernels containing this substring
    std::map<std::string, std::string> tag_kernel;  // layer tag -> kernel symbol used in the previous pass
    std::vector<ProfEvent> prof_events;
    std::vector<glass_prof_row> prof_rows;
    std::vector<hipEvent_t> event_pool;
    size_t event_next = 0;
};

// engine_ops.cpp helpers shared with the diagnostic op ABI
int glass_fold_upconv(const float* W, int cout, int cin, std::vector<_Float16>& out);  // [9][4*cout][cin]
void glass_pack_conv(const float* W, int cout, int cin, int ks, int cin_pad, std::vector<_Float16>& out);  // [ks*ks][cout][cin_pad]
