// engine.cpp — C ABI (include/glass.h) + orchestration of the fitness pass.
//
// Host-side mirror of problem.py:14-29 / generator.py:29-60 / models.py:108-129:
//   latents -> mapping -> styles/demod -> [per chunk: synthesis -> toRGB/skip -> resize -> D] -> CLIP -> F
// One engine per (process, GPU); one private stream; weights repacked once in finalize().
#include "engine.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <stdlib.h>

#include "kernels.h"

static thread_local std::string g_err;
void glass_set_error(const std::string& s) { g_err = s; }
extern "C" const char* glass_last_error(void) { return g_err.c_str(); }
extern "C" const char* glass_version(void) { return "clip-glass-amd 0.1 (gfx950)"; }

// ------------------------------------------------------------------------------------
// weight repacking (reference layouts -> kernel layouts)
// ------------------------------------------------------------------------------------
// plain conv: W[o][i][ky][kx] * coef -> [tap][o][i_pad] fp16
void glass_pack_conv(const float* W, int cout, int cin, int ks, int cin_pad, std::vector<_Float16>& out) {
    const float coef = 1.0f / sqrtf((float)cin * ks * ks);  // modules.py:103-108 (gain 1, lr_mul 1)
    out.assign((size_t)ks * ks * cout * cin_pad, (_Float16)0.f);
    for (int o = 0; o < cout; ++o)
        for (int i = 0; i < cin; ++i)
            for (int t = 0; t < ks * ks; ++t)
                out[((size_t)t * cout + o) * cin_pad + i] = (_Float16)(W[((size_t)o * cin + i) * ks * ks + t] * coef);
}

// up conv: conv_transpose2d(stride 2, 3x3) followed by the 4x4 FIR (gain 4, pad 1)
// (modules.py:1089-1139) == 3x3 conv (pad 1) with 4 phase kernels + depth-to-space.
//   1-D: t[q] = sum_i x[i] w[q-2i];  out[p] = sum_j f[j] t[p+j-1]  =>  out[p] = sum_i x[i] g[p-2i],
//   g[r] = sum_j f[j] w[r+j-1], r in [-2,3];  out[2m+ph] = sum_{d=0..2} x[m-1+d] g[(2+ph)-2d].
// Output layout [tap = dy*3+dx][n = (py*2+px)*cout + o][i].
int glass_fold_upconv(const float* W, int cout, int cin, std::vector<_Float16>& out) {
    const float coef = 1.0f / sqrtf((float)cin * 9.f);
    const float f1[4] = {1.f, 3.f, 3.f, 1.f};
    float F[4][4];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) F[a][b] = f1[a] * f1[b] / 64.f * 4.f;  // modules.py:201-202, up_factor^2
    out.assign((size_t)9 * 4 * cout * cin, (_Float16)0.f);
    std::vector<float> g(36);
    for (int o = 0; o < cout; ++o)
        for (int i = 0; i < cin; ++i) {
            const float* w = W + ((size_t)o * cin + i) * 9;
            for (int ry = -2; ry <= 3; ++ry)
                for (int rx = -2; rx <= 3; ++rx) {
                    float s = 0.f;
                    for (int jy = 0; jy < 4; ++jy) {
                        const int ky = ry + jy - 1;
                        if (ky < 0 || ky > 2) continue;
                        for (int jx = 0; jx < 4; ++jx) {
                            const int kx = rx + jx - 1;
                            if (kx < 0 || kx > 2) continue;
                            s += F[jy][jx] * w[ky * 3 + kx];
                        }
                    }
                    g[(ry + 2) * 6 + (rx + 2)] = s * coef;
                }
            for (int py = 0; py < 2; ++py)
                for (int px = 0; px < 2; ++px)
                    for (int dy = 0; dy < 3; ++dy)
                        for (int dx = 0; dx < 3; ++dx) {
                            const int ry = (2 + py) - 2 * dy, rx = (2 + px) - 2 * dx;
                            out[((size_t)(dy * 3 + dx) * 4 * cout + (size_t)(py * 2 + px) * cout + o) * cin + i] =
                                (_Float16)g[(ry + 2) * 6 + (rx + 2)];
                        }
        }
    return 0;
}

std::vector<_Float16> to_half(const float* p, size_t n, float scale) {
    std::vector<_Float16> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = (_Float16)(p[i] * scale);
    return v;
}
std::vector<float> scaled(const float* p, size_t n, float scale) {
    std::vector<float> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = p[i] * scale;
    return v;
}
// W[N][K] * coef -> Wt[K][N]
std::vector<float> transposed(const float* W, int N, int K, float coef) {
    std::vector<float> v((size_t)N * K);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) v[(size_t)k * N + n] = W[(size_t)n * K + k] * coef;
    return v;
}

// ------------------------------------------------------------------------------------
// create / destroy / load
// ------------------------------------------------------------------------------------
extern "C" int glass_engine_create(const glass_config* cfg, glass_engine** out) {
    REQUIRE(cfg && out, GLASS_ERR_ARG, "null argument");
    REQUIRE(cfg->n_blocks >= 0 && cfg->n_blocks <= GLASS_MAX_BLOCKS, GLASS_ERR_ARG, "n_blocks out of range");
    for (int i = 0; i < cfg->n_blocks; ++i) {
        const int ch = cfg->channels[i];    // the toRGB / fromRGB kernels are instantiated for these widths (kernels_misc.hip)
        REQUIRE(ch == 16 || ch == 32 || ch == 64 || ch == 128 || ch == 256 || ch == 512, GLASS_ERR_ARG,
                "channels must be one of 16, 32, 64, 128, 256, 512");
    }
    REQUIRE(cfg->latent_size > 0 && cfg->latent_size % 4 == 0, GLASS_ERR_ARG, "latent_size must be a multiple of 4");
    REQUIRE(cfg->batch_size > 0 && cfg->max_pop > 0, GLASS_ERR_ARG, "batch_size/max_pop must be positive");
    REQUIRE(cfg->max_pop % cfg->batch_size == 0, GLASS_ERR_ARG, "max_pop must be a multiple of batch_size");
    REQUIRE(cfg->n_obj == 1 || cfg->n_obj == 2, GLASS_ERR_ARG, "n_obj must be 1 or 2");
    if (cfg->use_discriminator) {
        REQUIRE(cfg->mbstd_group >= 2 && cfg->mbstd_group <= 8 && cfg->batch_size % cfg->mbstd_group == 0, GLASS_ERR_ARG,
                "batch_size must be a multiple of mbstd_group (modules.py:716)");
    }
    REQUIRE(cfg->clip_width > 0 && cfg->clip_width % 16 == 0 && cfg->clip_heads > 0 &&
                cfg->clip_width / cfg->clip_heads == 64 && cfg->clip_patch > 0 &&
                cfg->clip_res % cfg->clip_patch == 0 && (3 * cfg->clip_patch * cfg->clip_patch) % 16 == 0,
            GLASS_ERR_ARG, "unsupported CLIP geometry (head dim must be 64)");
    REQUIRE(cfg->noise_mode >= 0 && cfg->noise_mode <= 2, GLASS_ERR_ARG, "noise_mode must be 0,1,2");
    int bg_res = 0;
    if (cfg->generator == GLASS_GEN_BIGGAN_DEEP) {
        REQUIRE(cfg->n_blocks == 0 && !cfg->use_discriminator && cfg->n_obj == 1, GLASS_ERR_ARG,
                "BigGAN-deep: n_blocks must be 0, no discriminator, n_obj 1 (config.py:31-74)");
        REQUIRE(cfg->bg_n_layers > 0 && cfg->bg_n_layers <= GLASS_MAX_BG_LAYERS, GLASS_ERR_ARG, "bg_n_layers out of range");
        REQUIRE(cfg->bg_ch > 0 && cfg->bg_ch % 32 == 0 && cfg->bg_z_dim > 0 && cfg->bg_z_dim % 4 == 0 && cfg->bg_num_classes > 0,
                GLASS_ERR_ARG, "BigGAN-deep: bad channel_width / z_dim / num_classes");
        REQUIRE(cfg->latent_size == cfg->bg_z_dim + cfg->bg_num_classes, GLASS_ERR_ARG,
                "BigGAN-deep: latent_size must be z_dim + num_classes (latent.py:16-18)");
        REQUIRE(cfg->bg_n_stats >= 2 && cfg->bg_eps > 0.f && cfg->bg_truncation > 0.f && cfg->bg_truncation <= 1.f,
                GLASS_ERR_ARG, "BigGAN-deep: bad n_stats / eps / truncation");
        bg_res = 4;
        for (int i = 0; i < cfg->bg_n_layers; ++i) {
            const int cin = cfg->bg_ch * cfg->bg_layers[i][1], cout = cfg->bg_ch * cfg->bg_layers[i][2];
            REQUIRE(cin > 0 && cout > 0 && (cin == cout || cin == 2 * cout), GLASS_ERR_ARG,
                    "BigGAN-deep GenBlock: out channels must equal in or in/2 (channel-drop skip)");
            REQUIRE(cin % 128 == 0 && cout % 32 == 0, GLASS_ERR_ARG,
                    "BigGAN-deep GenBlock: in/4 and out channels must be multiples of 32");
            bg_res <<= (cfg->bg_layers[i][0] ? 1 : 0);
        }
        REQUIRE(cfg->bg_layers[0][1] == 16 && cfg->bg_layers[cfg->bg_n_layers - 1][2] == 1, GLASS_ERR_ARG,
                "BigGAN-deep: first GenBlock takes 16*ch channels, last produces ch");
    } else {
        REQUIRE(cfg->generator == GLASS_GEN_STYLEGAN2, GLASS_ERR_ARG, "unknown generator kind");
    }
    int ndev = 0;
    GLASS_HIP(hipGetDeviceCount(&ndev));
    REQUIRE(cfg->device >= 0 && cfg->device < ndev, GLASS_ERR_ARG, "no such HIP device");
    GLASS_HIP(hipSetDevice(cfg->device));
    glass_engine* e = new glass_engine();
    e->cfg = *cfg;
    e->R = cfg->n_blocks > 0 ? 4 << (cfg->n_blocks - 1) : bg_res;
    int chunk = cfg->chunk > 0 ? cfg->chunk : std::max(cfg->batch_size, (64 / cfg->batch_size) * cfg->batch_size);   // 288 GB of HBM: one chunk of 64 candidates (36 GB of activations at 1024 px) keeps every launch large
    chunk = std::min(chunk, cfg->max_pop);
    if (chunk % cfg->batch_size != 0) {
        delete e;
        glass_set_error("chunk must be a multiple of batch_size");
        return GLASS_ERR_ARG;
    }
    e->chunk = chunk;
    // Stream mode: glass_engine_set_overlap() is the only control in the release library (glass_knob() reads nothing there; the two
    // variables below exist in the developer build, and bench.py translates them into set_overlap() calls for the measure_* scripts).
    // Chunk pipelining (mode 1) stretches every co-running kernel ~2x, which makes per-kernel profiles meaningless: opt-in.
    e->overlap = glass_knob("GLASS_OVERLAP") != nullptr;
    // default (mode 2): CLIP's image tower (short, latency-bound launches) on the second stream next to the discriminator, which only
    // shares the finished image with it
    e->clip_overlap = glass_knob("GLASS_NO_CLIP_OVERLAP") == nullptr;
    hipError_t err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (err == hipSuccess) {
        // the second stream carries CLIP's ~75 short dependent launches beside the discriminator's chip-filling kernels: at the highest
        // priority its workgroups are placed first whenever a CU has room (the main stream's persistent kernels never yield one)
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        err = hipStreamCreateWithPriority(&e->stream_d, hipStreamNonBlocking, prio_hi);
    }
    e->cur = e->stream;
    if (err == hipSuccess) err = hipEventCreate(&e->ev0);
    if (err == hipSuccess) err = hipEventCreate(&e->ev1);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_noise, hipEventDisableTiming);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_rgb, hipEventDisableTiming);
    if (err != hipSuccess) {
        delete e;
        glass_set_error(std::string("stream/event creation failed: ") + hipGetErrorString(err));
        return GLASS_ERR_HIP;
    }
    *out = e;
    return GLASS_OK;
}

static void gpt2_work_free(glass_engine* e);
static void text_work_free(glass_engine* e) {
    auto& w = e->twork;
    hipFree(w.d_tok); hipFree(w.d_rows); hipFree(w.x); hipFree(w.cls); hipFree(w.feat); hipFree(w.ln16); hipFree(w.qkv); hipFree(w.att); hipFree(w.hid);
    w = glass_engine::TextWork();
}
extern "C" void glass_engine_destroy(glass_engine* e) {
    if (!e) return;
    hipSetDevice(e->cfg.device);
    if (e->stream) hipStreamSynchronize(e->stream);
    if (e->stream_d) hipStreamSynchronize(e->stream_d);
    gpt2_work_free(e);
    std::swap(e->gwork, e->gwork_alt);
    gpt2_work_free(e);
    text_work_free(e);
    for (void* p : e->allocs) hipFree(p);
    if (e->h_pinned) hipHostFree(e->h_pinned);
    for (auto ev : e->event_pool) hipEventDestroy(ev);
    for (auto ev : e->ev_g) hipEventDestroy(ev);
    for (auto ev : e->ev_d) hipEventDestroy(ev);
    if (e->ev0) hipEventDestroy(e->ev0);
    if (e->ev1) hipEventDestroy(e->ev1);
    if (e->ev_noise) hipEventDestroy(e->ev_noise);
    if (e->ev_rgb) hipEventDestroy(e->ev_rgb);
    if (e->stream) hipStreamDestroy(e->stream);
    if (e->stream_d) hipStreamDestroy(e->stream_d);
    delete e;
}

extern "C" int glass_engine_load_tensor(glass_engine* e, const char* name, const float* data, int32_t rank,
                                        const int64_t* dims) {
    REQUIRE(e && name && data && rank >= 0 && (rank == 0 || dims), GLASS_ERR_ARG, "null argument");
    REQUIRE(!e->finalized, GLASS_ERR_STATE, "engine already finalized");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < rank; ++i) {
        REQUIRE(dims[i] > 0, GLASS_ERR_ARG, "non-positive dimension");
        t.dims.push_back(dims[i]);
        n *= (size_t)dims[i];
    }
    t.data.assign(data, data + n);
    e->host[name] = std::move(t);
    return GLASS_OK;
}

// ------------------------------------------------------------------------------------
// finalize: build every device-side tensor
// ------------------------------------------------------------------------------------
static int finalize_generator(glass_engine* e) {
    const glass_config& c = e->cfg;
    const int L = c.latent_size;
    char nm[256];
    // mapping network (stylegan2/models.py:566-588): weight coef = lr_mul/sqrt(fan_in), bias coef = lr_mul
    const float lr = 0.01f;
    for (int i = 0; i < c.mapping_layers; ++i) {
        snprintf(nm, sizeof nm, "G_mapping.main.%d.layer.weight", i);
        GET(w, nm);
        REQUIRE(numel(w) == (size_t)L * L, GLASS_ERR_ARG, std::string("bad shape: ") + nm);
        snprintf(nm, sizeof nm, "G_mapping.main.%d.bias", i);
        GET(b, nm);
        REQUIRE(numel(b) == (size_t)L, GLASS_ERR_ARG, std::string("bad shape: ") + nm);
        float *dw, *db;
        int rc = upload(e, &dw, transposed(w->data.data(), L, L, lr / sqrtf((float)L)));
        if (rc) return rc;
        rc = upload(e, &db, scaled(b->data.data(), L, lr));
        if (rc) return rc;
        e->map_wt.push_back(dw);
        e->map_b.push_back(db);
    }
    // layer list (stylegan2/models.py:812-896, 969-1014)
    int style_idx = 0, soff = 0, dsoff = 0, noise_idx = 0;
    for (int b = 0; b < c.n_blocks; ++b) {
        const int res = 4 << b;
        const int nl = b == 0 ? 1 : 2;
        for (int l = 0; l < nl; ++l) {
            GConv g;
            g.up = (b > 0 && l == 0);
            g.cin = (b == 0) ? c.channels[0] : (l == 0 ? c.channels[b - 1] : c.channels[b]);
            g.cout = c.channels[b];
            g.res_out = res;
            g.res_in = g.up ? res / 2 : res;
            g.style_idx = style_idx++;
            g.style_off = soff;
            soff += g.cin;
            g.ds_off = dsoff;
            dsoff += g.cout;
            g.noise_idx = noise_idx++;
            e->gconv.push_back(g);
        }
        GRgb r;
        r.cin = c.channels[b];
        r.res = res;
        r.style_idx = style_idx++;
        r.style_off = soff;
        soff += r.cin;
        e->grgb.push_back(r);
    }
    e->n_style = style_idx;
    e->S_total = soff;
    e->D_total = dsoff;
    // style affines, concatenated: s = w @ (A/sqrt(L))^T + b   (modules.py:879-894, 936)
    std::vector<float> swt((size_t)L * e->S_total), sb((size_t)e->S_total);
    e->style_off.assign(e->n_style, 0);
    e->style_len.assign(e->n_style, 0);
    auto add_style = [&](const std::string& prefix, int sidx, int off, int cin) -> int {
        GET(A, prefix + ".dense.layer.weight");
        GET(Ab, prefix + ".dense.bias");
        REQUIRE(numel(A) == (size_t)cin * L && numel(Ab) == (size_t)cin, GLASS_ERR_ARG, "bad style shape: " + prefix);
        const float coef = 1.0f / sqrtf((float)L);
        for (int i = 0; i < cin; ++i) {
            for (int k = 0; k < L; ++k) swt[(size_t)k * e->S_total + off + i] = A->data[(size_t)i * L + k] * coef;
            sb[off + i] = Ab->data[i];
        }
        e->style_off[sidx] = off;
        e->style_len[sidx] = cin;
        return GLASS_OK;
    };
    {
        GET(cst, "G_synthesis.const");
        const int C0 = c.channels[0];
        REQUIRE(numel(cst) == (size_t)C0 * 16, GLASS_ERR_ARG, "bad shape: G_synthesis.const");
        std::vector<_Float16> h((size_t)16 * C0);
        for (int ch = 0; ch < C0; ++ch)
            for (int p = 0; p < 16; ++p) h[(size_t)p * C0 + ch] = (_Float16)cst->data[(size_t)ch * 16 + p];
        int rc = upload(e, &e->g_const, h);
        if (rc) return rc;
    }
    int gi = 0;
    for (int b = 0; b < c.n_blocks; ++b) {
        const int nl = b == 0 ? 1 : 2;
        for (int l = 0; l < nl; ++l, ++gi) {
            GConv& g = e->gconv[gi];
            snprintf(nm, sizeof nm, "G_synthesis.conv_blocks.%d.conv_block.%d", b, l);
            const std::string p = nm;
            GET(W, p + ".layer.layer.weight");
            GET(bias, p + ".bias");
            GET(ns, p + ".layer.weight");
            REQUIRE(numel(W) == (size_t)g.cout * g.cin * 9 && numel(bias) == (size_t)g.cout && numel(ns) == 1,
                    GLASS_ERR_ARG, "bad conv shape: " + p);
            int rc = add_style(p + ".layer.layer", g.style_idx, g.style_off, g.cin);
            if (rc) return rc;
            std::vector<_Float16> packed;
            if (g.up) glass_fold_upconv(W->data.data(), g.cout, g.cin, packed);
            else glass_pack_conv(W->data.data(), g.cout, g.cin, 3, g.cin, packed);
            rc = upload(e, &g.w, packed);
            if (rc) return rc;
            if (g.up) {
                glass_pack_conv(W->data.data(), g.cout, g.cin, 3, g.cin, packed);
                rc = upload(e, &g.w_up, packed);
                if (rc) return rc;
            }
            // demod table: Wsq[i][o] = sum_taps (W*coef)^2   (modules.py:943-954, SURVEY 8a note 1)
            std::vector<float> wsq((size_t)g.cin * g.cout);
            const float coef2 = 1.0f / ((float)g.cin * 9.f);
            for (int o = 0; o < g.cout; ++o)
                for (int i = 0; i < g.cin; ++i) {
                    const float* w = W->data.data() + ((size_t)o * g.cin + i) * 9;
                    float s = 0.f;
                    for (int t = 0; t < 9; ++t) s += w[t] * w[t];
                    wsq[(size_t)i * g.cout + o] = s * coef2;
                }
            rc = upload(e, &g.wsq, wsq);
            if (rc) return rc;
            rc = upload(e, &g.bias, bias->data);
            if (rc) return rc;
            g.noise_strength = ns->data[0];
        }
        GRgb& r = e->grgb[b];
        snprintf(nm, sizeof nm, "G_synthesis.to_data_layers.%d", b);
        const std::string p = nm;
        GET(W, p + ".layer.weight");
        GET(bias, p + ".bias");
        REQUIRE(numel(W) == (size_t)3 * r.cin && numel(bias) == 3, GLASS_ERR_ARG, "bad toRGB shape: " + p);
        int rc = add_style(p + ".layer", r.style_idx, r.style_off, r.cin);
        if (rc) return rc;
        rc = upload(e, &r.w, scaled(W->data.data(), (size_t)3 * r.cin, 1.0f / sqrtf((float)r.cin)));
        if (rc) return rc;
        rc = upload(e, &r.bias, bias->data);
        if (rc) return rc;
    }
    int rc = upload(e, &e->style_wt, swt);
    if (rc) return rc;
    rc = upload(e, &e->style_b, sb);
    if (rc) return rc;
    rc = upload(e, &e->d_style_off, e->style_off);
    if (rc) return rc;
    rc = upload(e, &e->d_style_len, e->style_len);
    return rc;
}

static int finalize_discriminator(glass_engine* e) {
    const glass_config& c = e->cfg;
    const int n = c.n_blocks;
    char nm[256];
    auto chD = [&](int i) { return c.channels[n - 1 - i]; };  // D order: first (full res) -> last (4x4)
    {
        GET(W, "D.from_data_layers.0.layer.weight");
        GET(b, "D.from_data_layers.0.bias");
        REQUIRE(numel(W) == (size_t)chD(0) * 3 && numel(b) == (size_t)chD(0), GLASS_ERR_ARG, "bad fromRGB shape");
        int rc = upload(e, &e->d_frgb_w, scaled(W->data.data(), numel(W), 1.0f / sqrtf(3.f)));
        if (rc) return rc;
        rc = upload(e, &e->d_frgb_b, b->data);
        if (rc) return rc;
    }
    for (int i = 0; i < n - 1; ++i) {
        DBlock d;
        d.cin = chD(i);
        d.cout = chD(i + 1);
        d.res = e->R >> i;
        snprintf(nm, sizeof nm, "D.conv_blocks.%d", i);
        const std::string p = nm;
        GET(W0, p + ".conv_block.0.layer.weight");
        GET(B0, p + ".conv_block.0.bias");
        GET(W1, p + ".conv_block.1.layer.weight");
        GET(B1, p + ".conv_block.1.bias");
        GET(WS, p + ".projection.weight");
        REQUIRE(numel(W0) == (size_t)d.cin * d.cin * 9 && numel(W1) == (size_t)d.cout * d.cin * 9 &&
                    numel(WS) == (size_t)d.cout * d.cin && numel(B0) == (size_t)d.cin && numel(B1) == (size_t)d.cout,
                GLASS_ERR_ARG, "bad D block shape: " + p);
        std::vector<_Float16> pk;
        glass_pack_conv(W0->data.data(), d.cin, d.cin, 3, d.cin, pk);
        int rc = upload(e, &d.w0, pk);
        if (rc) return rc;
        glass_pack_conv(W1->data.data(), d.cout, d.cin, 3, d.cin, pk);
        rc = upload(e, &d.w1, pk);
        if (rc) return rc;
        glass_pack_conv(WS->data.data(), d.cout, d.cin, 1, d.cin, pk);
        rc = upload(e, &d.wskip, pk);
        if (rc) return rc;
        rc = upload(e, &d.b0, B0->data);
        if (rc) return rc;
        rc = upload(e, &d.b1, B1->data);
        if (rc) return rc;
        e->dblk.push_back(d);
    }
    const int CL = chD(n - 1);
    snprintf(nm, sizeof nm, "D.conv_blocks.%d.1.conv_block.0", n - 1);
    const std::string p = nm;
    GET(WF, p + ".layer.weight");
    GET(BF, p + ".bias");
    REQUIRE(numel(WF) == (size_t)CL * (CL + 1) * 9 && numel(BF) == (size_t)CL, GLASS_ERR_ARG, "bad D final conv shape");
    e->d_final_cpad = ((CL + 1 + 15) / 16) * 16;
    std::vector<_Float16> pk;
    glass_pack_conv(WF->data.data(), CL, CL + 1, 3, e->d_final_cpad, pk);
    int rc = upload(e, &e->d_final_w, pk);
    if (rc) return rc;
    rc = upload(e, &e->d_final_b, BF->data);
    if (rc) return rc;
    GET(W0, "D.dense.0.layer.weight");
    GET(B0, "D.dense.0.bias");
    GET(W1, "D.dense.1.layer.weight");
    GET(B1, "D.dense.1.bias");
    REQUIRE(numel(W0) == (size_t)CL * CL * 16 && numel(B0) == (size_t)CL && numel(W1) == (size_t)CL && numel(B1) == 1,
            GLASS_ERR_ARG, "bad D dense shape");
    // x.view(B,-1) flattens NCHW (models.py:1224): column c*16+p ; our activations are [p][c].
    std::vector<_Float16> d0((size_t)CL * CL * 16);
    const float coef0 = 1.0f / sqrtf((float)CL * 16.f);
    for (int o = 0; o < CL; ++o)
        for (int ch = 0; ch < CL; ++ch)
            for (int px = 0; px < 16; ++px)
                d0[(size_t)o * CL * 16 + (size_t)px * CL + ch] = (_Float16)(W0->data[(size_t)o * CL * 16 + ch * 16 + px] * coef0);
    rc = upload(e, &e->d_dense0_w, d0);
    if (rc) return rc;
    rc = upload(e, &e->d_dense0_b, B0->data);
    if (rc) return rc;
    rc = upload(e, &e->d_dense1_wt, transposed(W1->data.data(), 1, CL, 1.0f / sqrtf((float)CL)));
    if (rc) return rc;
    return upload(e, &e->d_dense1_b, B1->data);
}

static int load_clip_blocks(glass_engine* e, const char* prefix, int layers, int W, std::vector<ClipBlock>& out) {
    char nm[256];
    int rc;
    for (int i = 0; i < layers; ++i) {
        snprintf(nm, sizeof nm, "%s%d.", prefix, i);
        const std::string p = nm;
        ClipBlock b;
        GET(l1g, p + "ln_1.weight");
        GET(l1b, p + "ln_1.bias");
        GET(l2g, p + "ln_2.weight");
        GET(l2b, p + "ln_2.bias");
        GET(wq, p + "attn.in_proj_weight");
        GET(bq, p + "attn.in_proj_bias");
        GET(wo, p + "attn.out_proj.weight");
        GET(bo, p + "attn.out_proj.bias");
        GET(wf, p + "mlp.c_fc.weight");
        GET(bf, p + "mlp.c_fc.bias");
        GET(wp, p + "mlp.c_proj.weight");
        GET(bp, p + "mlp.c_proj.bias");
        REQUIRE(numel(wq) == (size_t)3 * W * W && numel(wo) == (size_t)W * W && numel(wf) == (size_t)4 * W * W &&
                    numel(wp) == (size_t)4 * W * W,
                GLASS_ERR_ARG, "bad CLIP block shapes: " + p);
        if ((rc = upload(e, &b.ln1_g, l1g->data))) return rc;
        if ((rc = upload(e, &b.ln1_b, l1b->data))) return rc;
        if ((rc = upload(e, &b.ln2_g, l2g->data))) return rc;
        if ((rc = upload(e, &b.ln2_b, l2b->data))) return rc;
        if ((rc = upload(e, &b.w_qkv, to_half(wq->data.data(), numel(wq))))) return rc;
        if ((rc = upload(e, &b.w_out, to_half(wo->data.data(), numel(wo))))) return rc;
        if ((rc = upload(e, &b.w_fc, to_half(wf->data.data(), numel(wf))))) return rc;
        if ((rc = upload(e, &b.w_proj, to_half(wp->data.data(), numel(wp))))) return rc;
        if ((rc = upload(e, &b.b_qkv, bq->data))) return rc;
        if ((rc = upload(e, &b.b_out, bo->data))) return rc;
        if ((rc = upload(e, &b.b_fc, bf->data))) return rc;
        if ((rc = upload(e, &b.b_proj, bp->data))) return rc;
        out.push_back(b);
    }
    return GLASS_OK;
}

static int finalize_clip(glass_engine* e) {
    const glass_config& c = e->cfg;
    const int W = c.clip_width, ps = c.clip_patch, G = c.clip_res / ps, T = G * G + 1, E = c.clip_embed;
    const std::string v = "clip.visual.";
    GET(conv1, v + "conv1.weight");
    GET(cls, v + "class_embedding");
    GET(pos, v + "positional_embedding");
    GET(lg, v + "ln_pre.weight");
    GET(lb, v + "ln_pre.bias");
    GET(pg, v + "ln_post.weight");
    GET(pb, v + "ln_post.bias");
    GET(proj, v + "proj");
    REQUIRE(numel(conv1) == (size_t)W * 3 * ps * ps && numel(cls) == (size_t)W && numel(pos) == (size_t)T * W &&
                numel(proj) == (size_t)W * E,
            GLASS_ERR_ARG, "bad CLIP visual shapes");
    int rc = upload(e, &e->c_patch_w, to_half(conv1->data.data(), numel(conv1)));
    if (rc) return rc;
    if ((rc = upload(e, &e->c_cls, cls->data))) return rc;
    if ((rc = upload(e, &e->c_pos, pos->data))) return rc;
    if ((rc = upload(e, &e->c_lnpre_g, lg->data))) return rc;
    if ((rc = upload(e, &e->c_lnpre_b, lb->data))) return rc;
    if ((rc = upload(e, &e->c_lnpost_g, pg->data))) return rc;
    if ((rc = upload(e, &e->c_lnpost_b, pb->data))) return rc;
    if ((rc = upload(e, &e->c_proj, proj->data))) return rc;  // already [K=W][N=E]
    int rc2 = load_clip_blocks(e, "clip.visual.transformer.resblocks.", c.clip_layers, W, e->cblk);
    if (rc2) return rc2;
    // ---- optional text tower (clip/model.py:277-290) ----
    if (find(e, "clip.token_embedding.weight") != nullptr) {
        GET(tok, "clip.token_embedding.weight");
        GET(tpos, "clip.positional_embedding");
        GET(fg, "clip.ln_final.weight");
        GET(fb, "clip.ln_final.bias");
        GET(tp, "clip.text_projection");
        REQUIRE(tok->dims.size() == 2 && tpos->dims.size() == 2 && tpos->dims[1] == tok->dims[1], GLASS_ERR_ARG,
                "bad CLIP text embedding shapes");
        e->t_vocab = (int)tok->dims[0];
        e->t_width = (int)tok->dims[1];
        e->t_ctx = (int)tpos->dims[0];
        REQUIRE(e->t_width % 64 == 0 && numel(tp) == (size_t)e->t_width * E, GLASS_ERR_ARG, "bad CLIP text projection shape");
        int nl = 0;
        char nm2[256];
        for (;; ++nl) {
            snprintf(nm2, sizeof nm2, "clip.transformer.resblocks.%d.ln_1.weight", nl);
            if (!find(e, nm2)) break;
        }
        if ((rc = upload(e, &e->t_tok, tok->data))) return rc;
        if ((rc = upload(e, &e->t_pos, tpos->data))) return rc;
        if ((rc = upload(e, &e->t_lnf_g, fg->data))) return rc;
        if ((rc = upload(e, &e->t_lnf_b, fb->data))) return rc;
        if ((rc = upload(e, &e->t_proj, tp->data))) return rc;
        if ((rc = load_clip_blocks(e, "clip.transformer.resblocks.", nl, e->t_width, e->tblk))) return rc;
    }
    return GLASS_OK;
}

static int alloc_buffers(glass_engine* e) {
    const glass_config& c = e->cfg;
    const int P = c.max_pop, L = c.latent_size, CH = e->chunk;
    int rc;
    if ((rc = dev_alloc(e, &e->d_z, (size_t)P * L))) return rc;
    if ((rc = dev_alloc(e, &e->d_w0, (size_t)P * L))) return rc;
    if ((rc = dev_alloc(e, &e->d_w1, (size_t)P * L))) return rc;
    if ((rc = dev_alloc(e, &e->d_s, (size_t)P * e->S_total))) return rc;
    if ((rc = dev_alloc(e, &e->d_s16, (size_t)P * e->S_total))) return rc;
    if ((rc = dev_alloc(e, &e->d_smax, (size_t)P * e->n_style))) return rc;
    if ((rc = dev_alloc(e, &e->d_trgb_tab, (size_t)P * 32 * 512))) return rc;
    {   // toRGB partial sums of the blocks wider than 128 channels: [C / 128][chunk][3][res][res] fp32, the largest block's
        size_t part = 0;
        for (int b = 0; b < c.n_blocks; ++b)
            if (c.channels[b] > 128 && c.channels[b] % 128 == 0 && c.channels[b] <= 512)
                part = std::max(part, (size_t)(c.channels[b] / 128) * (size_t)P * 3 * ((size_t)4 << b) * ((size_t)4 << b));   // (P: the blocks up to 32 x 32 run the whole population)
        if (part && (rc = dev_alloc(e, &e->d_trgb_part, part))) return rc;
    }
    if (e->cfg.generator != GLASS_GEN_BIGGAN_DEEP && e->cfg.n_blocks > 0) {
        // StyleGAN2's low-resolution layers as im2col + GEMM (conv_gemm.hip): conv grids up to 16 x 16 per candidate, widest
        // channel count (151 + 134 MB at P = 64).  The capacities are per candidate: whether a layer takes this path must not
        // depend on how many candidates a launch carries.  (A few OTHER dispatchers do look at the launch size — conv_stream wants
    // enough tiles per workgroup, gemm_tiled picks its tile by M, gpt2's split-K depth follows M — so chunking / sharding
    // invariance holds to fp16 rounding, not to the bit: tests/test_gpu_engine.py::test_pop512_as_eight_shards_of_64.)
        int cmax = 16;
        for (int i = 0; i < e->cfg.n_blocks; ++i) cmax = std::max(cmax, (int)e->cfg.channels[i]);
        e->cap_a = 256LL * 9 * cmax;
        e->cap_c = 256LL * 4 * cmax;
        if ((rc = dev_alloc(e, &e->ws_a, (size_t)(e->cap_a * P)))) return rc;
        if ((rc = dev_alloc(e, &e->ws_c, (size_t)(e->cap_c * P)))) return rc;
        if (e->cfg.use_discriminator) {   // stream mode 1 runs D on the second stream next to G: its own scratch
            if ((rc = dev_alloc(e, &e->ws_a2, (size_t)(e->cap_a * P)))) return rc;
            if ((rc = dev_alloc(e, &e->ws_c2, (size_t)(e->cap_c * P)))) return rc;
        }
    } else if (e->cfg.generator == GLASS_GEN_BIGGAN_DEEP && !glass_knob("GLASS_BG_NO_CONV_GEMM")) {
        // BigGAN-deep's 4 x 4 .. 16 x 16 layers (3 x 3 on ch * 4 = 512 channels, 1 x 1 up to ch * 16 outputs) on the same path (round 3:
        // they ran on conv_direct at 64 - 220 TFLOP/s); the walk over the population is chunked, so the scratch holds one chunk
        const int cmax = 4 * e->cfg.bg_ch;
        e->cap_a = 256LL * 9 * cmax;
        e->cap_c = 256LL * 4 * cmax;
        if ((rc = dev_alloc(e, &e->ws_a, (size_t)(e->cap_a * P)))) return rc;
        if ((rc = dev_alloc(e, &e->ws_c, (size_t)(e->cap_c * P)))) return rc;
    }
    if ((rc = dev_alloc(e, &e->d_epsrow, (size_t)P * e->n_style))) return rc;
    if ((rc = dev_alloc(e, &e->d_dscale, (size_t)P * e->D_total))) return rc;
    for (auto& g : e->gconv) {
        // per-sample weights where the tensor is tiny and the tiled / fused kernels (one sample per block) apply
        const bool fused_ok = g.up && g.cin % 32 == 0 && g.cout % 32 == 0 && g.res_in >= 16;
        const bool tiled_ok = !g.up && g.cin % 32 == 0 && g.cout % 32 == 0 && g.res_in % 32 == 0;
        g.welems = 9LL * g.cin * g.cout;
        static const long long premod_kb = glass_knob("GLASS_PREMOD_MAX_KB") ? atoll(glass_knob("GLASS_PREMOD_MAX_KB")) : 500;   // A/B knob (round 3: the 256 -> 128 up-conv, 590 KB per sample, runs 4 % faster on the shared-weight image grid, and its consumer 5 % faster on the pre-styled output; 1200 was round 2's value)
        g.premod = (fused_ok || tiled_ok) && g.welems * 2 <= (premod_kb << 10) && !glass_knob("GLASS_NO_PREMOD");
        if (g.premod && (rc = dev_alloc(e, &g.wm, (size_t)P * g.welems))) return rc;
    }
    if (c.noise_mode != 0) {
        const int n_mb = P / c.batch_size;
        for (auto& g : e->gconv) {
            float* p;
            if ((rc = dev_alloc(e, &p, (size_t)n_mb * g.res_out * g.res_out))) return rc;
            e->d_noise.push_back(p);
        }
    }
    size_t maxact = 0;
    for (int b = 0; b < c.n_blocks; ++b) {
        const size_t res = 4u << b;
        size_t ch = c.channels[b];
        if (b > 0) ch = std::max<size_t>(ch, c.channels[b - 1]);
        maxact = std::max(maxact, (res + 1) * (res + 1) * ch);
    }
    if (c.n_blocks > 0) maxact = std::max(maxact, (size_t)16 * (c.channels[0] + 16));
    e->act_elems = maxact * CH;
    for (int i = 0; i < 8 && c.n_blocks > 0; ++i)   // [0..5]: D scratch (D stream), [6..7]: G ping-pong (main stream)
        if (i >= 6 || c.use_discriminator)
            if ((rc = dev_alloc(e, &e->act[i], e->act_elems))) return rc;
    for (int i = 0; i < 4; ++i)     // two sets (chunk parity) of skip-image ping-pong
        if ((rc = dev_alloc(e, &e->ybuf[i], (size_t)CH * 3 * e->R * e->R))) return rc;
    // whole-population buffers for the low-resolution phases (res <= low_res)
    if (c.n_blocks > 0) {
        e->n_low = 0;
        while (e->n_low < c.n_blocks && (4 << e->n_low) <= e->low_res) ++e->n_low;
        if (e->n_low < 1) e->n_low = 1;
        const size_t rl = 4u << (e->n_low - 1);
        size_t cmax = 16;
        for (int b = 0; b < e->n_low; ++b) cmax = std::max<size_t>(cmax, c.channels[b]);
        if (e->n_low < c.n_blocks) cmax = std::max<size_t>(cmax, c.channels[e->n_low]);
        const size_t low_elems = (size_t)P * (rl + 1) * (rl + 1) * (cmax + 16);
        const int n_lowbuf = c.use_discriminator ? 6 : 2;
        for (int i = 0; i < n_lowbuf; ++i)
            if ((rc = dev_alloc(e, &e->low[i], low_elems))) return rc;
        for (int i = 0; i < 2; ++i)
            if ((rc = dev_alloc(e, &e->ylow[i], (size_t)P * 3 * rl * rl))) return rc;
    }
    if ((rc = dev_alloc(e, &e->d_img, (size_t)CH * 3 * e->R * e->R))) return rc;
    const int W = c.clip_width, ps = c.clip_patch, G = c.clip_res / ps, T = G * G + 1;
    if ((rc = dev_alloc(e, &e->d_patches, (size_t)P * G * G * 3 * ps * ps))) return rc;
    if ((rc = dev_alloc(e, &e->d_pe, (size_t)P * G * G * W))) return rc;
    if ((rc = dev_alloc(e, &e->d_x, (size_t)P * T * W))) return rc;
    if ((rc = dev_alloc(e, &e->d_ln16, (size_t)P * T * W))) return rc;
    if ((rc = dev_alloc(e, &e->d_qkv, (size_t)P * T * 3 * W))) return rc;
    if ((rc = dev_alloc(e, &e->d_attn, (size_t)P * T * W))) return rc;
    if ((rc = dev_alloc(e, &e->d_hid, (size_t)P * T * 4 * W))) return rc;
    if ((rc = dev_alloc(e, &e->d_cls, (size_t)P * W))) return rc;
    if ((rc = dev_alloc(e, &e->d_feat, (size_t)P * c.clip_embed))) return rc;
    if ((rc = dev_alloc(e, &e->d_sim, (size_t)P))) return rc;
    if ((rc = dev_alloc(e, &e->d_dis, (size_t)P))) return rc;
    if ((rc = dev_alloc(e, &e->d_F, (size_t)P * 2))) return rc;
    if ((rc = dev_alloc(e, &e->d_target, (size_t)c.clip_embed))) return rc;
    if (c.use_discriminator && c.n_blocks > 0) {
        if ((rc = dev_alloc(e, &e->d_dfin, (size_t)P * 16 * c.channels[0]))) return rc;
        if ((rc = dev_alloc(e, &e->d_dh, (size_t)P * c.channels[0]))) return rc;
        if ((rc = dev_alloc(e, &e->d_dh_part, (size_t)16 * P * c.channels[0]))) return rc;    // split-K slices of D's first dense layer
    }
    {   // descriptor table of the demodulation problems (one launch for all layers)
        std::vector<DenseDesc> dd;
        for (auto& g : e->gconv) {
            DenseDesc q;
            q.x = e->d_s + g.style_off; q.ldx = e->S_total; q.K = g.cin; q.wt = g.wsq; q.N = g.cout; q.bias = nullptr;
            q.out = e->d_dscale + g.ds_off; q.ldo = e->D_total; q.eps_row = e->d_epsrow + g.style_idx; q.eps_stride = e->n_style;
            dd.push_back(q);
            e->demod_max_n = std::max(e->demod_max_n, g.cout);
        }
        DenseDesc* dptr = nullptr;
        if ((rc = upload(e, &dptr, dd))) return rc;
        e->d_demod_desc = dptr;
    }
    GLASS_HIP(hipMemset(e->d_dis, 0, (size_t)P * sizeof(float)));
    e->h_pinned_bytes = std::max((size_t)P * L, (size_t)P * (c.clip_embed + 8)) * sizeof(float);
    GLASS_HIP(hipHostMalloc((void**)&e->h_pinned, e->h_pinned_bytes, hipHostMallocDefault));
    return GLASS_OK;
}

// ------------------------------------------------------------------------------------
// GPT-2 (optional): Conv1D weights are [nx][nf] (gpt2/model.py:30-43) -> transposed to [nf][nx]
// ------------------------------------------------------------------------------------
static int finalize_gpt2(glass_engine* e) {
    if (!find(e, "gpt2.transformer.wte.weight")) return GLASS_OK;
    GET(wte, "gpt2.transformer.wte.weight");
    GET(wpe, "gpt2.transformer.wpe.weight");
    GET(lg, "gpt2.transformer.ln_f.weight");
    GET(lb, "gpt2.transformer.ln_f.bias");
    REQUIRE(wte->dims.size() == 2 && wpe->dims.size() == 2 && wpe->dims[1] == wte->dims[1], GLASS_ERR_ARG, "bad GPT-2 embedding shapes");
    e->g_vocab = (int)wte->dims[0];
    e->g_dim = (int)wte->dims[1];
    e->g_npos = (int)wpe->dims[0];
    const int D = e->g_dim;
    REQUIRE(D % 64 == 0, GLASS_ERR_ARG, "GPT-2 width must be a multiple of the 64-wide head");
    int rc;
    if ((rc = upload(e, &e->g_wte, wte->data))) return rc;
    if ((rc = upload(e, &e->g_wpe, wpe->data))) return rc;
    if ((rc = upload(e, &e->g_lnf_g, lg->data))) return rc;
    if ((rc = upload(e, &e->g_lnf_b, lb->data))) return rc;
    char nm[256];
    auto tr = [](const HostTensor* w, int nx, int nf) {   // [nx][nf] -> [nf][nx]
        std::vector<float> v((size_t)nx * nf);
        for (int i = 0; i < nx; ++i)
            for (int j = 0; j < nf; ++j) v[(size_t)j * nx + i] = w->data[(size_t)i * nf + j];
        return v;
    };
    for (int i = 0;; ++i) {
        snprintf(nm, sizeof nm, "gpt2.transformer.h.%d.", i);
        const std::string p = nm;
        if (!find(e, p + "ln_1.weight")) break;
        glass_engine::Gpt2Block b;
        GET(l1g, p + "ln_1.weight"); GET(l1b, p + "ln_1.bias"); GET(l2g, p + "ln_2.weight"); GET(l2b, p + "ln_2.bias");
        GET(wa, p + "attn.c_attn.weight"); GET(ba, p + "attn.c_attn.bias");
        GET(wo, p + "attn.c_proj.weight"); GET(bo, p + "attn.c_proj.bias");
        GET(wf, p + "mlp.c_fc.weight"); GET(bf, p + "mlp.c_fc.bias");
        GET(wp, p + "mlp.c_proj.weight"); GET(bp, p + "mlp.c_proj.bias");
        REQUIRE(numel(wa) == (size_t)3 * D * D && numel(wo) == (size_t)D * D && numel(wf) == (size_t)4 * D * D &&
                    numel(wp) == (size_t)4 * D * D, GLASS_ERR_ARG, "bad GPT-2 block shapes: " + p);
        if ((rc = upload(e, &b.ln1_g, l1g->data))) return rc;
        if ((rc = upload(e, &b.ln1_b, l1b->data))) return rc;
        if ((rc = upload(e, &b.ln2_g, l2g->data))) return rc;
        if ((rc = upload(e, &b.ln2_b, l2b->data))) return rc;
        if ((rc = upload(e, &b.w_qkv, tr(wa, D, 3 * D)))) return rc;
        if ((rc = upload(e, &b.b_qkv, ba->data))) return rc;
        if ((rc = upload(e, &b.w_o, tr(wo, D, D)))) return rc;
        if ((rc = upload(e, &b.b_o, bo->data))) return rc;
        if ((rc = upload(e, &b.w_fc, tr(wf, D, 4 * D)))) return rc;
        if ((rc = upload(e, &b.b_fc, bf->data))) return rc;
        if ((rc = upload(e, &b.w_pr, tr(wp, 4 * D, D)))) return rc;
        if ((rc = upload(e, &b.b_pr, bp->data))) return rc;
        e->gblk.push_back(b);
    }
    return GLASS_OK;
}

extern "C" int glass_engine_finalize(glass_engine* e) {
    REQUIRE(e, GLASS_ERR_ARG, "null engine");
    REQUIRE(!e->finalized, GLASS_ERR_STATE, "engine already finalized");
    GLASS_HIP(hipSetDevice(e->cfg.device));
    int rc = GLASS_OK;
    if (e->cfg.n_blocks > 0) {
        if ((rc = finalize_generator(e))) return rc;
        if (e->cfg.use_discriminator && (rc = finalize_discriminator(e))) return rc;
    }
    if (e->cfg.generator == GLASS_GEN_BIGGAN_DEEP && (rc = glass_biggan_finalize(e))) return rc;
    if ((rc = finalize_gpt2(e))) return rc;
    if ((rc = finalize_clip(e))) return rc;
    if ((rc = alloc_buffers(e))) return rc;
    e->host.clear();  // host copies no longer needed
    e->finalized = true;
    return GLASS_OK;
}

extern "C" int glass_engine_set_target(glass_engine* e, const float* feat, int32_t n) {
    REQUIRE(e && feat, GLASS_ERR_ARG, "null argument");
    REQUIRE(e->finalized, GLASS_ERR_STATE, "finalize() first");
    REQUIRE(n == e->cfg.clip_embed, GLASS_ERR_ARG, "target feature length != clip_embed");
    GLASS_HIP(hipSetDevice(e->cfg.device));
    GLASS_HIP(hipMemcpy(e->d_target, feat, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    e->has_target = true;
    return GLASS_OK;
}

// ------------------------------------------------------------------------------------
// profiling scopes (hipEvent pair per launch when enabled)
// ------------------------------------------------------------------------------------
void collect_profile(glass_engine* e) {
    std::map<std::string, glass_prof_row> rows;
    std::vector<std::string> order;
#ifdef GLASS_AB_KNOBS
    // developer build: GLASS_TIMELINE=<file> appends "start end name" (ms from the pass's first event) of every instrumented launch —
    // where the second stream's launches sit beside the main stream's (tools/layer_ab.py --mode 2)
    FILE* tl = getenv("GLASS_TIMELINE") ? fopen(getenv("GLASS_TIMELINE"), "a") : nullptr;
    if (tl) fprintf(tl, "# pass\n");
#endif
    for (auto& pe : e->prof_events) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, pe.e0, pe.e1);
#ifdef GLASS_AB_KNOBS
        if (tl) {
            float t0 = 0.f;
            hipEventElapsedTime(&t0, e->ev0, pe.e0);
            fprintf(tl, "%9.3f %9.3f %s\n", t0, t0 + ms, pe.name.c_str());
        }
#endif
        auto it = rows.find(pe.name);
        if (it == rows.end()) {
            glass_prof_row r;
            memset(&r, 0, sizeof r);
            strncpy(r.name, pe.name.c_str(), sizeof(r.name) - 1);
            it = rows.emplace(pe.name, r).first;
            order.push_back(pe.name);
        }
        it->second.launches += 1;
        it->second.total_ms += ms;
        it->second.flops += pe.flops;
        it->second.bytes += pe.bytes;
    }
#ifdef GLASS_AB_KNOBS
    if (tl) fclose(tl);
#endif
    e->prof_rows.clear();
    for (auto& n : order) e->prof_rows.push_back(rows[n]);
    e->prof_events.clear();
    e->event_next = 0;
}

void run_conv(glass_engine* e, const ConvParams& p, const char* tag, double flops, double bytes) {
    if (!(p.out_scale > 0.f)) {   // the epilogues fold the scale into the activation constants, max(v k1, v k2) (common.h act_apply): s > 0 only
        if (e->launch_error.empty()) e->launch_error = std::string("out_scale must be positive: ") + tag;
        return;
    }
    Prof pr(e, tag, flops, bytes);
    const char* k = p.up ? launch_upconv_fused(p, e->cur) : nullptr;
    if (p.rgb_tanh_out) {      // (one kernel family writes this output; the caller asked conv_tiled's dry run first)
        k = launch_conv_tiled(p, e->cur);
        if (!k && e->launch_error.empty()) e->launch_error = std::string("no kernel writes the planar tanh output of layer ") + tag;
        if (!k) k = "(refused)";
        if (pr.on) pr.pe.name = std::string(tag) + "@" + k;
        if (e->profiling) e->tag_kernel[tag] = k;
        return;
    }
    if (!k) k = launch_conv_stream(p, e->cur);
    if (!k) k = launch_conv_glds(p, e->cur);
    if (!k) k = launch_conv_tiled(p, e->cur);
    if (!k) {
        const bool second = e->cur == e->stream_d && e->ws_a2;
        k = launch_conv_gemm(p, second ? e->ws_a2 : e->ws_a, e->cap_a, second ? e->ws_c2 : e->ws_c, e->cap_c, e->cur);
    }
    if (!k) k = launch_conv_direct(p, e->cur);
    if (!k) {   // no kernel family accepts this layer: remember it, the pass returns an error instead of a wrong result
        if (e->launch_error.empty()) e->launch_error = std::string("no kernel accepts layer ") + tag;
        k = "(refused)";
    }
    if (pr.on) pr.pe.name = std::string(tag) + "@" + k;
    if (e->profiling) e->tag_kernel[tag] = k;
}
void run_gemm(glass_engine* e, const GemmParams& p, const char* tag) {
    Prof pr(e, tag, 2.0 * p.M * p.N * p.K, 2.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N));
    const char* k = launch_gemm_tiled(p, e->cur);
    if (!k) k = launch_gemm_direct(p, e->cur);
    if (pr.on) pr.pe.name = std::string(tag) + "@" + k;
    if (e->profiling) e->tag_kernel[tag] = k;
}

ConvParams conv_defaults() {
    ConvParams p;
    memset(&p, 0, sizeof p);
    p.out_scale = 1.f;
    p.batch_size = 1;
    p.stride = 1;
    return p;
}

// ------------------------------------------------------------------------------------
// the pass
// ------------------------------------------------------------------------------------
static int upload_noise(glass_engine* e, int P, int generation, int first_mb, const glass_noise* noise) {
    const glass_config& c = e->cfg;
    const int n_mb = P / c.batch_size;
    if (c.noise_mode == 1) {
        for (size_t l = 0; l < e->gconv.size(); ++l) {
            const int hw = e->gconv[l].res_out * e->gconv[l].res_out;
            Prof pr(e, "noise", 0, (double)n_mb * hw * 4);
            launch_noise(e->d_noise[l], n_mb, hw, (uint32_t)l, (uint32_t)first_mb, (uint32_t)generation, c.noise_seed,
                         e->cur);
        }
    } else if (c.noise_mode == 2) {
        REQUIRE(noise && noise->planes, GLASS_ERR_ARG, "noise_mode 2 requires caller-provided noise planes");
        REQUIRE(noise->n_layers == (int)e->gconv.size() && noise->n_minibatches >= n_mb, GLASS_ERR_ARG,
                "noise: wrong number of layers / minibatches");
        for (int m = 0; m < n_mb; ++m)
            for (size_t l = 0; l < e->gconv.size(); ++l) {
                const size_t hw = (size_t)e->gconv[l].res_out * e->gconv[l].res_out;
                const float* src = noise->planes[(size_t)m * noise->n_layers + l];
                REQUIRE(src, GLASS_ERR_ARG, "noise: null plane");
                GLASS_HIP(hipMemcpyAsync(e->d_noise[l] + (size_t)m * hw, src, hw * sizeof(float), hipMemcpyHostToDevice,
                                         e->cur));
            }
        GLASS_HIP(hipStreamSynchronize(e->cur));  // caller's planes may be freed after return
    }
    return GLASS_OK;
}

static void run_styles(glass_engine* e, int P) {
    const glass_config& c = e->cfg;
    const int L = c.latent_size;
    {
        Prof pr(e, "mapping", 2.0 * P * L * L * c.mapping_layers, 4.0 * L * L * c.mapping_layers);
        static const bool no_fused = glass_knob("GLASS_NO_MAP_FUSE") != nullptr;   // A/B knob
        if (no_fused || c.mapping_layers < 1 ||
            !launch_mapping_fused(e->d_z, e->d_w0, P, L, 1e-8f, e->map_wt.data(), e->map_b.data(), c.mapping_layers, e->cur)) {
            launch_pixelnorm(e->d_z, e->d_w0, P, L, 1e-8f, e->cur);
            float *a = e->d_w0, *b = e->d_w1;
            for (int i = 0; i < c.mapping_layers; ++i) {
                if (L % 64 == 0 && L <= 768) launch_dense_splitk(a, L, P, L, e->map_wt[i], L, e->map_b[i], b, L, 1, e->cur);
                else launch_dense(a, L, P, L, e->map_wt[i], L, e->map_b[i], b, L, 0, 1, nullptr, 0, e->cur);
                std::swap(a, b);
            }
            if (a != e->d_w0)  // result must end in d_w0
                hipMemcpyAsync(e->d_w0, a, (size_t)P * L * sizeof(float), hipMemcpyDeviceToDevice, e->cur);
        }
    }
    {
        Prof pr(e, "styles", 2.0 * P * L * e->S_total, 4.0 * L * e->S_total);
        launch_dense(e->d_w0, L, P, L, e->style_wt, e->S_total, e->style_b, e->d_s, e->S_total, 0, 0, nullptr, 0,
                     e->cur);
        launch_style_norm(e->d_s, e->S_total, P, e->n_style, e->d_style_off, e->d_style_len, e->d_smax, e->d_epsrow,
                          1e-8f, e->cur);
        launch_bg_to_half(e->d_s, e->d_s16, (long long)P * e->S_total, e->cur);   // fp16 table for the LDS-tiled kernels
    }
    {
        Prof pr(e, "demod", 0, 0);
        launch_dense_multi((const DenseDesc*)e->d_demod_desc, (int)e->gconv.size(), e->demod_max_n, P, 1, 2, e->cur);
    }
    {
        Prof pr(e, "premod_weights", 0, 0);
        for (auto& g : e->gconv)
            if (g.premod)
                launch_modulate_weights(g.up ? g.w_up : g.w, g.welems, g.cin, g.cout, e->d_s + g.style_off, e->S_total,
                                        e->d_dscale + g.ds_off, e->D_total, P, g.wm, e->cur);
    }
}

// ------------------------------------------------------------------------------------
// Synthesis blocks [b_lo, b_hi) for candidates [c0, c0+B).  Low-resolution blocks
// (res <= low_res) run once for the whole population (launch-/latency-bound otherwise),
// high-resolution blocks run per chunk so the working set stays near the caches.
//   x/xbs: input feature map (bstride 0 = the learned const); pp[2]: ping-pong outputs;
//   yprev: skip image of the previous block (nullptr for block 0); yb[2]: skip ping-pong.
// Returns the final feature map / skip image through the out parameters.
// ------------------------------------------------------------------------------------
static void run_g_blocks(glass_engine* e, int c0, int B, int b_lo, int b_hi, const half_t* x, long long xbs,
                         half_t* const pp[2], const float* yprev, float* const yb[2], const half_t** x_out,
                         const float** y_out) {
    const glass_config& c = e->cfg;
    char tag[48];
    int gi = b_lo == 0 ? 0 : 1 + 2 * (b_lo - 1);
    int yi = 0;
    static const bool no_trgb_fuse = glass_knob("GLASS_NO_TRGB_FUSE") != nullptr;   // experiment knobs
    static const bool no_trgb_mid = glass_knob("GLASS_NO_TRGB_MID") != nullptr;
    static const bool no_pre_style = glass_knob("GLASS_NO_PRE_STYLE") != nullptr;
    static const bool no_planar = glass_knob("GLASS_NO_PLANAR") != nullptr;      // A/B knob: conv_wreg's input stays pixel-major
    // A block's SEPARATE toRGB pass (blocks wider than 128 channels) is a bandwidth-bound read of the map the next block's
    // up-conv reads too; in the two-stream mode it runs on the second stream next to that (issue-bound) up-conv.  The main
    // stream joins before the next block's last conv: that launch overwrites the map toRGB reads and consumes its skip image.
    // Measured (round 3): 33.77 vs 33.74 ms per population — no gain, co-running kernels share the CUs they would have had
    // anyway (the late-CLIP variant, GLASS_CLIP_LATE, loses 0.3 ms).  Opt-in knob only.
    static const bool rgb_side_on = glass_knob("GLASS_TRGB_SIDE") != nullptr;
    const bool rgb_side = e->clip_overlap && e->cur == e->stream && rgb_side_on;
    bool rgb_pending = false;
    for (int b = b_lo; b < b_hi; ++b) {
        const int nl = b == 0 ? 1 : 2;
        bool rgb_done = false, pre_styled = false, x_planar = false;
        for (int l = 0; l < nl; ++l, ++gi) {
            if (rgb_pending && l == nl - 1) {
                hipStreamWaitEvent(e->stream, e->ev_rgb, 0);
                rgb_pending = false;
            }
            const GConv& g = e->gconv[gi];
            ConvParams p = conv_defaults();
            p.x = x;
            p.x_bstride = xbs;
            p.B = B;
            p.H = p.W = g.res_in;
            p.Cin = g.cin;
            p.Hc = p.Wc = g.res_in;
            p.KS = 3;
            p.pad = 1;
            p.w = g.w;
            p.w_up = g.w_up;
            p.Cout = g.cout;
            p.up = g.up;
            p.Neff = g.up ? 4 * g.cout : g.cout;
            p.Ho = p.Wo = g.res_out;
            p.sn = e->d_s + (size_t)c0 * e->S_total + g.style_off;
            p.sn16 = e->d_s16 + (size_t)c0 * e->S_total + g.style_off;
            p.sn_stride = e->S_total;
            p.dscale = e->d_dscale + (size_t)c0 * e->D_total + g.ds_off;
            p.ds_stride = e->D_total;
            if (c.noise_mode != 0) {
                p.noise = e->d_noise[g.noise_idx] + (size_t)(c0 / c.batch_size) * g.res_out * g.res_out;
                p.noise_strength = g.noise_strength;
            }
            p.batch_size = c.batch_size;
            p.bias = g.bias;
            p.act = 1;
            if (g.premod) {   // weights already carry style and demod of each sample
                p.sn = nullptr;
                p.sn16 = nullptr;
                p.dscale = nullptr;
                p.w_bstride = g.welems;
                if (g.up) { p.w_up = g.wm + (size_t)c0 * g.welems; p.w = nullptr; }
                else p.w = g.wm + (size_t)c0 * g.welems;
            }
            // upconv -> conv link: the up-conv's only consumer is the block's second conv, so (where the fused up-conv kernel
            // runs and that conv modulates on the activation side) its style is applied once, to the up-conv's output
            if (pre_styled) { p.sn = nullptr; p.sn16 = nullptr; pre_styled = false; }
            if (x_planar) { p.x_planar8 = 1; x_planar = false; }     // (a launcher that does not read the layout refuses the layer: run_conv reports it)
            if (g.up && l == 0 && nl == 2 && !no_pre_style && !e->gconv[gi + 1].premod && !e->gconv[gi + 1].up) {
                ConvParams dq = p;
                dq.dry_run = 1;
                dq.y = pp[0];
                if (launch_upconv_fused(dq, e->cur)) {
                    const GConv& g2 = e->gconv[gi + 1];
                    p.post_scale16 = e->d_s16 + (size_t)c0 * e->S_total + g2.style_off;
                    p.post_stride = e->S_total;
                    pre_styled = true;
                }
            }
            half_t* out = pp[(x == pp[0]) ? 1 : 0];
            p.y = out;
            // upconv -> conv_wreg link: that kernel reads its input one 32-channel chunk at a time, so the up-conv writes the map
            // chunk-planar for it (common.h x_planar8) — where the up-conv instance that can runs and the conv has no activation-side style
            if (g.up && l == 0 && nl == 2 && !no_planar && !e->gconv[gi + 1].up && (pre_styled || e->gconv[gi + 1].premod) &&
                conv_wreg_supported(e->gconv[gi + 1].cin, e->gconv[gi + 1].cout, g.res_out, g.res_out)) {
                ConvParams dq = p;
                dq.dry_run = 1;
                dq.y_planar8 = 1;
                if (launch_upconv_fused(dq, e->cur)) { p.y_planar8 = 1; x_planar = true; }
            }
            const double flops = 2.0 * B * (double)g.res_in * g.res_in * 9.0 * g.cin * g.cout;  // reference count
            const double bytes = 2.0 * B * ((double)g.res_in * g.res_in * g.cin + (double)g.res_out * g.res_out * g.cout) +
                                 2.0 * 9 * g.cin * p.Neff;
            snprintf(tag, sizeof tag, "G.%s.r%d.%dx%d", g.up ? "upconv" : "conv", g.res_out, g.cin, g.cout);
            if (l == nl - 1 && !g.up && !no_trgb_fuse) {
                // toRGB of the block fused into its last conv (common.h).  The network's LAST conv feeds toRGB only:
                // conv_stream<torgb> writes just the skip image and the 64-byte-per-pixel feature map never goes to HBM;
                // the mid-resolution blocks still store their map (the next block reads it) but toRGB no longer re-reads it.
                const GRgb& r = e->grgb[b];
                ConvParams q = p;
                q.trgb_w = r.w; q.trgb_b = r.bias;
                q.trgb_sn = e->d_s + (size_t)c0 * e->S_total + r.style_off; q.trgb_sn_stride = e->S_total;
                q.trgb_smax = e->d_smax + (size_t)c0 * e->n_style + r.style_idx; q.trgb_smax_stride = e->n_style;
                q.trgb_yprev = yprev; q.trgb_yout = yb[yi];
                const double tflops = flops + 2.0 * B * (double)r.res * r.res * 3 * r.cin;
                const double ybytes = B * (double)r.res * r.res * (12.0 + (b ? 3.0 : 0.0));
                if (b == c.n_blocks - 1) {
                    q.y = nullptr;
                    if (conv_stream_applies(q)) {
                        Prof pr(e, tag, tflops, 2.0 * B * (double)g.res_in * g.res_in * g.cin + ybytes);
                        const char* k = launch_conv_stream(q, e->cur);
                        if (pr.on) pr.pe.name = std::string(tag) + "@" + k;
                        if (e->profiling) e->tag_kernel[tag] = k;
                        rgb_done = true;
                        x = nullptr;   // not produced
                    }
                }
                if (!rgb_done && !no_trgb_mid && r.cin <= 128) {
                    q.y = out;
                    q.trgb_tab = e->d_trgb_tab + (size_t)c0 * 32 * 128;
                    q.dry_run = 1;
                    const char* k = launch_conv_glds(q, e->cur);
                    if (!k) k = launch_conv_tiled(q, e->cur);
                    if (k) {
                        q.dry_run = 0;
                        Prof pr(e, tag, tflops, bytes + ybytes);
                        launch_trgb_tables(q.trgb_w, q.trgb_sn, q.trgb_sn_stride, q.trgb_smax, q.trgb_smax_stride, B, r.cin,
                                           e->d_trgb_tab + (size_t)c0 * 32 * 128, e->cur);
                        k = launch_conv_glds(q, e->cur);
                        if (!k) k = launch_conv_tiled(q, e->cur);
                        if (pr.on) pr.pe.name = std::string(tag) + "@" + k;
                        if (e->profiling) e->tag_kernel[tag] = k;
                        rgb_done = true;
                        x = out;
                        xbs = (long long)g.res_out * g.res_out * g.cout;
                    }
                }
            }
            if (l == nl - 1 && !g.up && !no_trgb_fuse && !rgb_done && !no_trgb_mid && e->d_trgb_part) {
                // blocks wider than 128 channels (several 128-wide n tiles per pixel): every n tile's conv epilogue writes the toRGB partial sum of
                // its channels, a 3-value-per-pixel pass adds them (+ bias + the upsampled previous image) — the separate toRGB pass read the
                // whole feature map again (0.19 + 0.09 + 0.03 ms at r128 / r64 / r32)
                const GRgb& r = e->grgb[b];
                if (r.cin > 128 && r.cin % 128 == 0 && r.cin <= 512) {
                    ConvParams q = p;
                    q.trgb_w = r.w; q.trgb_b = r.bias;
                    q.trgb_sn = e->d_s + (size_t)c0 * e->S_total + r.style_off; q.trgb_sn_stride = e->S_total;
                    q.trgb_smax = e->d_smax + (size_t)c0 * e->n_style + r.style_idx; q.trgb_smax_stride = e->n_style;
                    q.trgb_tab = e->d_trgb_tab + (size_t)c0 * 32 * 512;
                    q.trgb_part = e->d_trgb_part;
                    q.dry_run = 1;
                    if (launch_conv_glds(q, e->cur)) {
                        q.dry_run = 0;
                        const double tflops = flops + 2.0 * B * (double)r.res * r.res * 3 * r.cin;
                        const double ybytes = B * (double)r.res * r.res * (12.0 * (1 + 2 * (r.cin / 128)) + (b ? 3.0 : 0.0));
                        Prof pr(e, tag, tflops, bytes + ybytes);
                        launch_trgb_tables(q.trgb_w, q.trgb_sn, q.trgb_sn_stride, q.trgb_smax, q.trgb_smax_stride, B, r.cin,
                                           e->d_trgb_tab + (size_t)c0 * 32 * 512, e->cur);
                        const char* k = launch_conv_glds(q, e->cur);
                        launch_trgb_finish(e->d_trgb_part, r.cin / 128, B, r.res, r.bias, yprev, yb[yi], e->cur);
                        if (pr.on) pr.pe.name = std::string(tag) + "@" + k;
                        if (e->profiling) e->tag_kernel[tag] = k;
                        rgb_done = true;
                        x = out;
                        xbs = (long long)g.res_out * g.res_out * g.cout;
                    }
                }
            }
            if (rgb_done) { ++gi; break; }
            run_conv(e, p, tag, flops, bytes);
            x = out;
            xbs = (long long)g.res_out * g.res_out * g.cout;
        }
        const GRgb& r = e->grgb[b];
        if (!rgb_done) {
            snprintf(tag, sizeof tag, "G.torgb.r%d", r.res);
            const bool side = rgb_side && b + 1 < b_hi;       // (the last block's image is consumed right away)
            if (side) {
                hipEventRecord(e->ev_rgb, e->stream);
                hipStreamWaitEvent(e->stream_d, e->ev_rgb, 0);
                e->cur = e->stream_d;
            }
            {
            Prof pr(e, tag, 2.0 * B * (double)r.res * r.res * 3 * r.cin,
                    B * ((double)r.res * r.res * (2.0 * r.cin + 12.0 + (b ? 3.0 : 0.0))));
            if (!launch_torgb(x, B, r.res, r.res, r.cin, r.w, r.bias, e->d_s + (size_t)c0 * e->S_total + r.style_off,
                              e->S_total, e->d_smax + (size_t)c0 * e->n_style + r.style_idx, e->n_style, yprev, yb[yi], e->cur) &&
                e->launch_error.empty())
                e->launch_error = std::string("toRGB width not instantiated: ") + tag;
            }
            if (side) {
                hipEventRecord(e->ev_rgb, e->stream_d);
                e->cur = e->stream;
                rgb_pending = true;
            }
        }
        yprev = yb[yi];
        yi ^= 1;
    }
    if (rgb_pending) hipStreamWaitEvent(e->stream, e->ev_rgb, 0);
    *x_out = x;
    *y_out = yprev;
}

// Discriminator conv blocks [i_lo, i_hi) (D order: block i works at resolution R >> i).
// bufs: six scratch feature maps; X enters in `X`; the result pointer is returned.
static void run_fromrgb(glass_engine* e, int B, const float* y, half_t* X);

// rgb_y != nullptr (only with i_lo == 0): X has NOT been produced yet — the first conv builds the fromRGB map from the
// skip image on the fly and writes it to X as a side output (conv_stream<fromrgb>), or, where that kernel does not
// apply, the separate fromRGB pass runs first.
static half_t* run_d_blocks(glass_engine* e, int B, int i_lo, int i_hi, half_t* X, half_t* const bufs[5],
                            const float* rgb_y = nullptr) {
    char tag[64];
    half_t *Hb = bufs[0], *HB = bufs[1], *XS = bufs[2], *S = bufs[3], *O = bufs[4];
    static const bool no_planar = glass_knob("GLASS_NO_PLANAR") != nullptr;      // A/B knob: conv_wreg's input stays pixel-major
    bool x_planar = false;       // X is chunk-planar (common.h x_planar8): written so by the fused first block for conv_wreg
    for (int i = i_lo; i < i_hi; ++i) {
        const DBlock& d = e->dblk[i];
        const int r = d.res, r2 = r / 2;
        ConvParams p = conv_defaults();
        p.x = X; p.x_bstride = (long long)r * r * d.cin; p.B = B; p.H = p.W = r; p.Cin = d.cin;
        p.Hc = p.Wc = r; p.KS = 3; p.pad = 1; p.w = d.w0; p.Cout = p.Neff = d.cin; p.Ho = p.Wo = r;
        p.bias = d.b0; p.act = 1; p.y = Hb;
        const bool x_was_planar = x_planar;
        if (x_planar) { p.x_planar8 = 1; x_planar = false; }
        bool fused_rgb = false, have_xs = false;
        if (i == 0 && rgb_y) {       // the whole block from the skip image in one kernel (conv_d0.hip): neither x nor h reaches HBM
            snprintf(tag, sizeof tag, "D.block0.r%d.%dx%dx%d", r, d.cin, d.cin, d.cout);
            const double px = (double)B * r * r, px2 = (double)B * r2 * r2;
            Prof pr(e, tag, 2.0 * px * (9.0 * d.cin * d.cin + 3.0 * d.cin) + 2.0 * px2 * 10.0 * d.cin * d.cout, px * 12.0 + px2 * 2.0 * d.cout);
            // the next block's first conv on conv_wreg (with the blur-down by-product: nothing else reads this map): chunk-planar output
            static const bool no_xs_fuse = glass_knob("GLASS_NO_XS_FUSE") != nullptr;
            const bool planar = !no_planar && !no_xs_fuse && i + 1 < i_hi && e->dblk[i + 1].cin == d.cout && conv_wreg_supported(d.cout, d.cout, r2, r2);
            const char* k = launch_dblock0(rgb_y, e->d_frgb_w, e->d_frgb_b, d.w0, d.b0, d.w1, d.wskip, d.b1, O, B, r, d.cin, d.cout, e->cur, planar);
            if (k) {
                x_planar = planar;
                if (pr.on) pr.pe.name = std::string(tag) + "@" + k;
                if (e->profiling) e->tag_kernel[tag] = k;
                std::swap(X, O);
                continue;
            }
            pr.on = false;
        }
        const bool fuse_down = conv_down_supported(r, d.cin, d.cout);   // blur + skip + stride-2 conv + merge as one kernel
        if (i == 0 && rgb_y) {
            static const bool no_fuse = glass_knob("GLASS_NO_FRGB_FUSE") != nullptr;   // A/B knob
            ConvParams q = p;
            q.rgb_y = rgb_y; q.rgb_w = e->d_frgb_w; q.rgb_b = e->d_frgb_b;
            q.rgb_x_out = fuse_down ? nullptr : X;       // the fused second half reads the down-sampled skip input only
            q.rgb_xs_out = fuse_down ? XS : nullptr;
            snprintf(tag, sizeof tag, "D.fromrgb+conv0.r%d.%dx%d", r, d.cin, d.cin);
            const double px = (double)B * r * r;
            Prof pr(e, tag, 2.0 * px * (9.0 * d.cin * d.cin + 3.0 * d.cin), px * (12.0 + 2.0 * d.cin + (fuse_down ? 0.5 : 2.0) * d.cin));
            const char* k = no_fuse ? nullptr : launch_conv_stream(q, e->cur);
            if (k) {
                fused_rgb = true;
                have_xs = fuse_down;
                if (pr.on) pr.pe.name = std::string(tag) + "@" + k;
                if (e->profiling) e->tag_kernel[tag] = k;
            } else {
                pr.on = false;
            }
        }
        if (i == 0 && rgb_y && !fused_rgb) run_fromrgb(e, B, rgb_y, X);
        snprintf(tag, sizeof tag, "D.conv0.r%d.%dx%d", r, d.cin, d.cin);
        if (!fused_rgb) {
            static const bool no_xs = glass_knob("GLASS_NO_XS_FUSE") != nullptr;   // A/B knob
            ConvParams qx = p;
            qx.xs_out = XS; qx.dry_run = 1;
            if (!no_xs && !have_xs && (launch_conv_glds(qx, e->cur) || launch_conv_tiled(qx, e->cur))) {   // the skip branch's blur-down rides in the first conv
                qx.dry_run = 0;
                p = qx;
                have_xs = true;
            }
            run_conv(e, p, tag, 2.0 * B * (double)r * r * 9 * d.cin * d.cin, 4.0 * B * (double)r * r * d.cin + (have_xs ? 0.5 * B * (double)r * r * d.cin : 0.0));
        }
        if (!have_xs && x_was_planar && e->launch_error.empty()) e->launch_error = std::string("chunk-planar block input without the fused blur-down: ") + tag;
        if (!have_xs) {
            snprintf(tag, sizeof tag, "D.blurdown.r%d", r);
            Prof pr(e, tag, 2.0 * B * (double)r2 * r2 * d.cin * 16, 2.5 * B * (double)r * r * d.cin);
            launch_blur_down(X, B, r, r, d.cin, XS, e->cur);
        }
        if (fuse_down) {
            snprintf(tag, sizeof tag, "D.down.r%d.%dx%d", r2, d.cin, d.cout);
            const double px2 = (double)B * r2 * r2;
            Prof pr(e, tag, 2.0 * px2 * (9.0 + 1.0) * d.cin * d.cout, 2.0 * (B * (double)r * r * d.cin + px2 * (d.cin + d.cout)));
            const char* k = launch_conv_down(Hb, XS, d.w1, d.wskip, d.b1, O, B, r, d.cin, d.cout, e->cur);
            if (k) {
                if (pr.on) pr.pe.name = std::string(tag) + "@" + k;
                if (e->profiling) e->tag_kernel[tag] = k;
                std::swap(X, O);
                continue;
            }
            pr.on = false;
        }
        ConvParams q = conv_defaults();
        q.x = HB; q.x_bstride = (long long)(r + 1) * (r + 1) * d.cin; q.B = B; q.H = q.W = r + 1; q.Cin = d.cin;
        q.Hc = q.Wc = r2; q.KS = 3; q.stride = 2; q.pad = 0; q.w = d.w1; q.Cout = q.Neff = d.cout; q.Ho = q.Wo = r2;
        q.bias = d.b1; q.act = 1; q.out_scale = 0.70710678118654752440f; q.y = O;
        // blur -> conv_s2 link: that kernel stages its input one 32-channel chunk per K step, so (where it runs) the blur writes 32-channel planes
        bool hb_planar = false;
        {
            static const bool no_planar = glass_knob("GLASS_NO_PLANAR") != nullptr;      // A/B knob
            static const bool no_skip_fuse0 = glass_knob("GLASS_NO_SKIP_FUSE") != nullptr;
            ConvParams qp = q;
            qp.skip_x = XS; qp.skip_w = d.wskip; qp.x_planar32 = 1; qp.dry_run = 1;
            hb_planar = !no_planar && !no_skip_fuse0 && blur_pad2_planar32_ok(d.cin) && launch_conv_s2(qp, e->cur) != nullptr;
        }
        {
            snprintf(tag, sizeof tag, "D.blur.r%d", r);
            Prof pr(e, tag, 2.0 * B * (double)(r + 1) * (r + 1) * d.cin * 16, 4.0 * B * (double)r * r * d.cin);
            launch_blur_pad2(Hb, B, r, r, d.cin, HB, e->cur, hb_planar);
        }
        snprintf(tag, sizeof tag, "D.conv1.r%d.%dx%d", r2, d.cin, d.cout);
        const double f1 = 2.0 * B * (double)r2 * r2 * 9 * d.cin * d.cout, fs = 2.0 * B * (double)r2 * r2 * d.cin * d.cout;
        {   // skip branch as extra K stages of the stride-2 conv (conv_tiled<3,2,4,N,skip>) where that kernel applies
            static const bool no_skip_fuse = glass_knob("GLASS_NO_SKIP_FUSE") != nullptr;   // A/B knob
            ConvParams qs = q;
            qs.skip_x = XS; qs.skip_w = d.wskip; qs.dry_run = 1;
            qs.x_planar32 = hb_planar;
            if (!no_skip_fuse && (hb_planar ? launch_conv_s2(qs, e->cur) : launch_conv_tiled(qs, e->cur))) {
                qs.dry_run = 0;
                Prof pr(e, tag, f1 + fs, 2.0 * B * ((double)(r + 1) * (r + 1) * d.cin + (double)r2 * r2 * (d.cin + d.cout)));
                const char* k = hb_planar ? launch_conv_s2(qs, e->cur) : launch_conv_tiled(qs, e->cur);
                if (pr.on) pr.pe.name = std::string(tag) + "@" + k;
                if (e->profiling) e->tag_kernel[tag] = k;
                std::swap(X, O);
                continue;
            }
        }
        ConvParams s = conv_defaults();
        s.x = XS; s.x_bstride = (long long)r2 * r2 * d.cin; s.B = B; s.H = s.W = r2; s.Cin = d.cin;
        s.Hc = s.Wc = r2; s.KS = 1; s.pad = 0; s.w = d.wskip; s.Cout = s.Neff = d.cout; s.Ho = s.Wo = r2; s.y = S;
        char stag[64];
        snprintf(stag, sizeof stag, "D.skip.r%d.%dx%d", r2, d.cin, d.cout);
        run_conv(e, s, stag, fs, 2.0 * B * (double)r2 * r2 * (d.cin + d.cout));
        q.res = S;
        run_conv(e, q, tag, f1, 2.0 * B * ((double)(r + 1) * (r + 1) * d.cin + 2.0 * r2 * r2 * d.cout));
        std::swap(X, O);
    }
    return X;
}

static void run_fromrgb(glass_engine* e, int B, const float* y, half_t* X) {
    const glass_config& c = e->cfg;
    const int n = c.n_blocks;
    Prof pr(e, "D.fromrgb", 2.0 * B * (double)e->R * e->R * 3 * c.channels[n - 1],
            B * (double)e->R * e->R * (12.0 + 2.0 * c.channels[n - 1]));
    launch_fromrgb(y, B, e->R, c.channels[n - 1], e->d_frgb_w, e->d_frgb_b, X, e->cur);
}

// mbstd + final conv + dense head for the whole population: X is [P][4][4][C0]
static void run_d_head(glass_engine* e, int P, const half_t* X, half_t* scratch) {
    const glass_config& c = e->cfg;
    const int CL = c.channels[0];
    {
        Prof pr(e, "D.mbstd", 0, 4.0 * P * 16 * CL);
        launch_mbstd(X, P, 16, CL, e->d_final_cpad, c.batch_size, c.mbstd_group, 1e-8f, scratch, e->cur);
    }
    ConvParams p = conv_defaults();
    p.x = scratch; p.x_bstride = 16LL * e->d_final_cpad; p.B = P; p.H = p.W = 4; p.Cin = e->d_final_cpad; p.Hc = p.Wc = 4;
    p.KS = 3; p.pad = 1; p.w = e->d_final_w; p.Cout = p.Neff = CL; p.Ho = p.Wo = 4; p.bias = e->d_final_b; p.act = 1;
    p.y = e->d_dfin;
    run_conv(e, p, "D.final_conv", 2.0 * P * 16 * 9.0 * (CL + 1) * CL, 2.0 * 9 * CL * (CL + 1));
    GemmParams g;
    memset(&g, 0, sizeof g);
    g.a = e->d_dfin; g.w = e->d_dense0_w; g.M = P; g.N = CL; g.K = 16 * CL; g.bias = e->d_dense0_b; g.mode = 4;
    g.out32 = e->d_dh; g.ldo = CL; g.cand_rows = 1;
    // M = P rows, K = 16 CL = 8192: the 128 x 64 tiles are 8 workgroups walking 128 K steps each (97 us for 0.5 GFLOP).  Split K into 16
    // slices (blockIdx.z) with raw partial sums, finished in a fixed order with bias + activation: 128+ workgroups, 8 steps each.
    static const bool no_d0_split = glass_knob("GLASS_NO_DENSE0_SPLIT") != nullptr;   // A/B knob
    const int S0 = 16;
    if (!no_d0_split && e->d_dh_part && g.K % (S0 * 64) == 0 && P <= e->cfg.max_pop) {
        GemmParams q = g;
        q.ld = g.K; q.K = g.K / S0; q.batch = S0; q.a_bs = q.K; q.w_bs = q.K; q.o_bs = (long long)P * CL;
        q.bias = nullptr; q.mode = 3; q.out32 = e->d_dh_part;
        Prof pr(e, "D.dense0", 2.0 * P * (double)g.K * CL, 2.0 * (double)g.K * CL);
        const char* k = launch_gemm_tiled(q, e->cur);
        if (k) {    // finish + the second dense layer (CL -> 1) in one launch
            launch_dense01_finish(e->d_dh_part, S0, q.o_bs, g.bias, e->d_dense1_wt, e->d_dense1_b, e->d_dis, P, CL, e->cur);
            if (pr.on) pr.pe.name = std::string("D.dense0+1@") + k + "+dense01_finish";
            return;
        }
        pr.on = false;
    }
    run_gemm(e, g, "D.dense0");
    {
        Prof pr(e, "D.dense1", 2.0 * P * CL, 0);
        launch_dense(e->d_dh, CL, P, CL, e->d_dense1_wt, 1, e->d_dense1_b, e->d_dis, 1, 0, 0, nullptr, 0, e->cur);
    }
}

void run_clip(glass_engine* e, int P, int l0, int l1);
// layers [l0, end) + the head WITHOUT the patch embedding (it ran with layers [0, l0) on another stream)
static void run_clip_rest(glass_engine* e, int P, int l0) {
    if (l0 == 0) {          // run_clip's l0 == 0 means "with the embedding": walk layer 0 through the general path's layer loop instead
        run_clip(e, P, -1, 1 << 20);
        return;
    }
    run_clip(e, P, l0, 1 << 20);
}
// layers [l0, l1) of the image tower; l0 == 0 also runs the patch embedding (l0 < 0: layers from 0 WITHOUT it), l1 >= the layer count also
// the head (ln_post, projection, cosine)
void run_clip(glass_engine* e, int P, int l0, int l1) {
    const glass_config& c = e->cfg;
    const int W = c.clip_width, ps = c.clip_patch, G = c.clip_res / ps, T = G * G + 1, M = P * T;
    GemmParams g;
    if (l0 == 0) {
        memset(&g, 0, sizeof g);
        g.a = e->d_patches; g.w = e->c_patch_w; g.M = P * G * G; g.N = W; g.K = 3 * ps * ps; g.mode = 3; g.out32 = e->d_pe; g.ldo = W; g.cand_rows = G * G;
        run_gemm(e, g, "clip.patch_embed");
        Prof pr(e, "clip.embed_lnpre", 0, 8.0 * M * W);
        launch_embed_lnpre(e->d_pe, e->c_cls, e->c_pos, e->c_lnpre_g, e->c_lnpre_b, P, T, W, e->d_x, e->cur);
    }
    for (int li = l0 < 0 ? 0 : l0; li < l1 && li < (int)e->cblk.size(); ++li) {
        auto& b = e->cblk[li];
        {
            Prof pr(e, "clip.layernorm", 0, 6.0 * M * W);
            launch_layernorm(e->d_x, W, M, W, b.ln1_g, b.ln1_b, e->d_ln16, nullptr, e->cur);
        }
        memset(&g, 0, sizeof g);
        g.a = e->d_ln16; g.w = b.w_qkv; g.M = M; g.N = 3 * W; g.K = W; g.bias = b.b_qkv; g.mode = 0; g.out16 = e->d_qkv; g.ldo = 3 * W; g.cand_rows = T;
        run_gemm(e, g, "clip.qkv");
        {
            Prof pr(e, "clip.attention", 4.0 * P * c.clip_heads * (double)T * T * 64, 8.0 * M * W);
            launch_attention(e->d_qkv, P, T, c.clip_heads, 64, 0, e->d_attn, e->cur);
        }
        memset(&g, 0, sizeof g);
        g.a = e->d_attn; g.w = b.w_out; g.M = M; g.N = W; g.K = W; g.bias = b.b_out; g.mode = 2; g.out32 = e->d_x; g.ldo = W; g.cand_rows = T;
        run_gemm(e, g, "clip.attn_out");
        {
            Prof pr(e, "clip.layernorm", 0, 6.0 * M * W);
            launch_layernorm(e->d_x, W, M, W, b.ln2_g, b.ln2_b, e->d_ln16, nullptr, e->cur);
        }
        memset(&g, 0, sizeof g);
        g.a = e->d_ln16; g.w = b.w_fc; g.M = M; g.N = 4 * W; g.K = W; g.bias = b.b_fc; g.mode = 1; g.out16 = e->d_hid; g.ldo = 4 * W; g.cand_rows = T;
        run_gemm(e, g, "clip.mlp_fc");
        memset(&g, 0, sizeof g);
        g.a = e->d_hid; g.w = b.w_proj; g.M = M; g.N = W; g.K = 4 * W; g.bias = b.b_proj; g.mode = 2; g.out32 = e->d_x; g.ldo = W; g.cand_rows = T;
        run_gemm(e, g, "clip.mlp_proj");
    }
    if (l1 >= (int)e->cblk.size()) {
        Prof pr(e, "clip.head", 2.0 * P * W * c.clip_embed, 4.0 * W * c.clip_embed);
        launch_layernorm(e->d_x, (long long)T * W, P, W, e->c_lnpost_g, e->c_lnpost_b, nullptr, e->d_cls, e->cur);
        launch_dense(e->d_cls, W, P, W, e->c_proj, c.clip_embed, nullptr, e->d_feat, c.clip_embed, 0, 0, nullptr, 0,
                     e->cur);
        launch_cosine(e->d_feat, e->d_target, P, c.clip_embed, e->d_sim, e->cur);
    }
}

static int run_pass(glass_engine* e, const float* latents, int P, int generation, int first_mb,
                    const glass_noise* noise, float* out_F, float* images) {
    REQUIRE(e && latents, GLASS_ERR_ARG, "null argument");
    REQUIRE(e->finalized, GLASS_ERR_STATE, "finalize() first (weights not loaded)");
    const glass_config& c = e->cfg;
    REQUIRE(P > 0 && P <= c.max_pop, GLASS_ERR_ARG, "population size out of range (max_pop)");
    REQUIRE(P % c.batch_size == 0, GLASS_ERR_ARG,
            "population size must be a multiple of batch_size (reference asserts: models.py:112)");
    const bool biggan = c.generator == GLASS_GEN_BIGGAN_DEEP;
    REQUIRE(e->cfg.n_blocks > 0 || biggan, GLASS_ERR_STATE, "this engine was created without a GAN (n_blocks = 0)");
    if (out_F) REQUIRE(e->has_target, GLASS_ERR_STATE, "set_target() first");
    GLASS_HIP(hipSetDevice(c.device));
    e->launch_error.clear();      // (a pass that returned early through GLASS_HIP must not leave its message to the next one)
    const int L = c.latent_size;
    memcpy(e->h_pinned, latents, (size_t)P * L * sizeof(float));
    GLASS_HIP(hipEventRecord(e->ev0, e->cur));
    GLASS_HIP(hipMemcpyAsync(e->d_z, e->h_pinned, (size_t)P * L * sizeof(float), hipMemcpyHostToDevice, e->cur));
    const int ps = c.clip_patch, G = c.clip_res / ps;
    const size_t img_elems = (size_t)3 * e->R * e->R;
    if (biggan) {
        // BigGAN-deep (models.py:75-86): no noise inputs, no discriminator; candidates are independent, so the
        // reference's minibatch loop has no semantic effect and the population is walked in engine chunks.
        int rc = glass_biggan_prepare(e, P);
        if (rc) return rc;
        for (int c0 = 0; c0 < P; c0 += e->chunk) {
            const int B = std::min(e->chunk, P - c0);
            float* y = e->ybuf[0];
            if ((rc = glass_biggan_chunk(e, c0, B, y))) return rc;
            if (images) {
                launch_finalize_image(y, e->d_img, (long long)B * img_elems, e->cur);
                GLASS_HIP(hipMemcpyAsync(images + (size_t)c0 * img_elems, e->d_img, (size_t)B * img_elems * sizeof(float),
                                         hipMemcpyDeviceToHost, e->cur));
            }
            if (out_F) {
                Prof pr(e, "clip.resize", 0, B * (16.0 * c.clip_res * c.clip_res * 3 + 2.0 * 3 * c.clip_res * c.clip_res));
                launch_resize_patches(y, B, e->R, c.clip_res, ps, e->d_patches + (size_t)c0 * G * G * 3 * ps * ps, e->cur);
            }
        }
        if (out_F) {
            run_clip(e, P, 0, 1 << 20);
            launch_assemble_F(e->d_sim, e->d_dis, P, c.n_obj, e->d_F, e->cur);
            GLASS_HIP(hipMemcpyAsync(e->h_pinned, e->d_F, (size_t)P * c.n_obj * sizeof(float), hipMemcpyDeviceToHost, e->cur));
        }
        GLASS_HIP(hipEventRecord(e->ev1, e->cur));
        GLASS_HIP(hipStreamSynchronize(e->cur));
        GLASS_HIP(hipGetLastError());
        GLASS_HIP(hipEventElapsedTime(&e->last_ms, e->ev0, e->ev1));
        if (out_F) memcpy(out_F, e->h_pinned, (size_t)P * c.n_obj * sizeof(float));
        e->last_P = P;
        if (e->profiling) collect_profile(e);
        return GLASS_OK;
    }
    // device-generated noise planes depend on nothing but (seed, generation, minibatch, layer): their 17 short launches run on the
    // second stream next to the mapping network / style / demodulation chain instead of ahead of it
    static const bool no_noise_ov = glass_knob("GLASS_NO_NOISE_OVERLAP") != nullptr;      // A/B knob, read once
    const bool noise_ov = e->clip_overlap && c.noise_mode == 1 && !no_noise_ov;
    int rc;
    if (noise_ov) {
        e->cur = e->stream_d;
        rc = upload_noise(e, P, generation, first_mb, noise);
        e->cur = e->stream;
        if (rc) return rc;
        GLASS_HIP(hipEventRecord(e->ev_noise, e->stream_d));
        run_styles(e, P);
        GLASS_HIP(hipStreamWaitEvent(e->stream, e->ev_noise, 0));
    } else {
        run_styles(e, P);
        rc = upload_noise(e, P, generation, first_mb, noise);
        if (rc) return rc;
    }
    const bool want_d = out_F && c.use_discriminator && c.n_obj == 2;
    const int n = c.n_blocks;
    const int nlow = std::min(n, e->n_low);           // G blocks 0..nlow-1 run for the whole population
    const int nd = (int)e->dblk.size();               // D conv blocks (n - 1)
    const int d_hi = want_d ? std::max(0, n - nlow) : 0;  // D blocks 0..d_hi-1 (res > low_res) run per chunk
    // ---- phase A: low-resolution synthesis, whole population ------------------------------
    const half_t* xlow = nullptr;
    const float* ylow = nullptr;
    {
        half_t* const pp[2] = {e->low[0], e->low[1]};
        float* const yb[2] = {e->ylow[0], e->ylow[1]};
        run_g_blocks(e, 0, P, 0, nlow, e->g_const, 0, pp, nullptr, yb, &xlow, &ylow);
    }
    const int res_low = 4 << (nlow - 1);
    const long long xlow_bs = (long long)res_low * res_low * c.channels[nlow - 1];
    // D feature map handed from the chunked phase to the whole-population phase
    half_t* dmid = e->low[5];
    const int res_mid = d_hi < nd ? e->dblk[d_hi].res : 4;
    const long long dmid_bs = (long long)res_mid * res_mid * (d_hi < nd ? e->dblk[d_hi].cin : c.channels[0]);
    // ---- phase B: high-resolution synthesis per chunk on the main stream; resize + high-resolution
    // D of chunk k run on the second stream, overlapping the synthesis of chunk k+1 (memory-bound
    // and matrix-bound phases of the two networks interleave on the CUs).
    const int n_chunks = (P + e->chunk - 1) / e->chunk;
    while ((int)e->ev_g.size() < n_chunks) {
        hipEvent_t a, b2;
        GLASS_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        GLASS_HIP(hipEventCreateWithFlags(&b2, hipEventDisableTiming));
        e->ev_g.push_back(a);
        e->ev_d.push_back(b2);
    }
    const bool overlap = e->overlap && out_F;
    const bool clip_ov = e->clip_overlap && !overlap && out_F && want_d;
    static const bool clip_late = glass_knob("GLASS_CLIP_LATE") != nullptr;      // A/B knob
    hipStream_t sd = overlap ? e->stream_d : e->stream;
    for (int c0 = 0, k = 0; c0 < P; c0 += e->chunk, ++k) {
        const int B = std::min(e->chunk, P - c0);
        const int set = k & 1;
        e->cur = e->stream;
        if (overlap && k >= 2) GLASS_HIP(hipStreamWaitEvent(e->stream, e->ev_d[k - 2], 0));  // y set re-use
        const float* y = ylow + (size_t)c0 * 3 * res_low * res_low;
        if (nlow < n) {
            half_t* const pp[2] = {e->act[6], e->act[7]};
            float* const yb[2] = {e->ybuf[2 * set], e->ybuf[2 * set + 1]};
            const half_t* xo;
            run_g_blocks(e, c0, B, nlow, n, xlow + (size_t)c0 * xlow_bs, xlow_bs, pp, y, yb, &xo, &y);
        }
        if (images) {
            launch_finalize_image(y, e->d_img, (long long)B * img_elems, e->cur);
            GLASS_HIP(hipMemcpyAsync(images + (size_t)c0 * img_elems, e->d_img, (size_t)B * img_elems * sizeof(float),
                                     hipMemcpyDeviceToHost, e->cur));
        }
        if (out_F) {
            if (overlap) {
                GLASS_HIP(hipEventRecord(e->ev_g[k], e->stream));
                GLASS_HIP(hipStreamWaitEvent(sd, e->ev_g[k], 0));
            }
            e->cur = sd;
            {
                Prof pr(e, "clip.resize", 0, B * (16.0 * c.clip_res * c.clip_res * 3 + 2.0 * 3 * c.clip_res * c.clip_res));
                launch_resize_patches(y, B, e->R, c.clip_res, ps, e->d_patches + (size_t)c0 * G * G * 3 * ps * ps,
                                      e->cur);
            }
            if (clip_ov && !clip_late && c0 + e->chunk >= P) {   // last chunk's patches are in place: CLIP starts now
                // its first layers on the MAIN stream (alone on the chip), the rest on the second stream beside the discriminator
                static const int serial_layers = glass_knob("GLASS_CLIP_SERIAL") ? atoi(glass_knob("GLASS_CLIP_SERIAL")) : GLASS_CLIP_SERIAL_LAYERS;
                if (serial_layers >= 0) run_clip(e, P, 0, serial_layers);      // (0: the patch embedding alone; -1: nothing)
                GLASS_HIP(hipEventRecord(e->ev_g[0], e->stream));
                GLASS_HIP(hipStreamWaitEvent(e->stream_d, e->ev_g[0], 0));
                e->cur = e->stream_d;
                if (serial_layers >= 0) run_clip_rest(e, P, serial_layers);
                else run_clip(e, P, 0, 1 << 20);
                GLASS_HIP(hipEventRecord(e->ev_d[0], e->stream_d));
                e->cur = e->stream;
            }
            if (want_d) {
                if (d_hi > 0) {
                    half_t* const bufs[5] = {e->act[1], e->act[2], e->act[3], e->act[4], e->act[5]};
                    half_t* Xo = run_d_blocks(e, B, 0, d_hi, e->act[0], bufs, y);   // fromRGB rides in the first conv
                    GLASS_HIP(hipMemcpyAsync(dmid + (size_t)c0 * dmid_bs, Xo, (size_t)B * dmid_bs * sizeof(half_t),
                                             hipMemcpyDeviceToDevice, e->cur));
                } else {
                    run_fromrgb(e, B, y, dmid + (size_t)c0 * dmid_bs);
                }
            }
            if (overlap) GLASS_HIP(hipEventRecord(e->ev_d[k], sd));
        }
    }
    if (clip_ov && clip_late) {   // CLIP's short launches next to the LOW-resolution discriminator (equally short launches) instead of
        GLASS_HIP(hipEventRecord(e->ev_g[0], e->stream));          // next to its chip-filling high-resolution kernels
        GLASS_HIP(hipStreamWaitEvent(e->stream_d, e->ev_g[0], 0));
        e->cur = e->stream_d;
        run_clip(e, P, 0, 1 << 20);
        GLASS_HIP(hipEventRecord(e->ev_d[0], e->stream_d));
        e->cur = e->stream;
    }
    // ---- phase C: low-resolution discriminator + head (second stream) || CLIP (main stream) ------
    if (want_d) {
        e->cur = sd;
        half_t* const bufs[5] = {e->low[0], e->low[1], e->low[2], e->low[3], e->low[4]};
        half_t* Xo = run_d_blocks(e, P, d_hi, nd, dmid, bufs);
        run_d_head(e, P, Xo, Xo == e->low[0] ? e->low[1] : e->low[0]);
    }
    if (out_F) {
        if (overlap) {   // patches (written on the second stream) -> CLIP on the main stream
            GLASS_HIP(hipStreamWaitEvent(e->stream, e->ev_d[n_chunks - 1], 0));
        }
        e->cur = e->stream;
        if (clip_ov) GLASS_HIP(hipStreamWaitEvent(e->stream, e->ev_d[0], 0));   // join: CLIP finished on the second stream
        else run_clip(e, P, 0, 1 << 20);
        if (overlap) {   // join: D head finished
            GLASS_HIP(hipEventRecord(e->ev_g[0], sd));
            GLASS_HIP(hipStreamWaitEvent(e->stream, e->ev_g[0], 0));
        }
        launch_assemble_F(e->d_sim, e->d_dis, P, c.n_obj, e->d_F, e->cur);
        GLASS_HIP(hipMemcpyAsync(e->h_pinned, e->d_F, (size_t)P * c.n_obj * sizeof(float), hipMemcpyDeviceToHost,
                                 e->cur));
    }
    e->cur = e->stream;
    GLASS_HIP(hipEventRecord(e->ev1, e->cur));
    GLASS_HIP(hipStreamSynchronize(e->cur));
    GLASS_HIP(hipGetLastError());
    GLASS_HIP(hipEventElapsedTime(&e->last_ms, e->ev0, e->ev1));
    if (!e->launch_error.empty()) {
        const std::string msg = e->launch_error;
        e->launch_error.clear();
        if (e->profiling) collect_profile(e);
        glass_set_error(msg);
        return GLASS_ERR_STATE;
    }
    if (out_F) memcpy(out_F, e->h_pinned, (size_t)P * c.n_obj * sizeof(float));
    e->last_P = P;
    if (e->profiling) collect_profile(e);
    return GLASS_OK;
}

// CLIP text tower: token+pos embedding -> causal transformer -> ln_final -> EOT row @ text_projection
extern "C" int glass_engine_encode_text(glass_engine* e, const int32_t* tokens, int32_t n_texts, int32_t ctx, float* out_feat) {
    REQUIRE(e && tokens && out_feat && n_texts > 0, GLASS_ERR_ARG, "null argument");
    REQUIRE(e->finalized, GLASS_ERR_STATE, "finalize() first");
    REQUIRE(e->t_tok != nullptr, GLASS_ERR_STATE, "CLIP text tower weights were not loaded (clip.token_embedding.weight ...)");
    REQUIRE(ctx == e->t_ctx && ctx <= 128, GLASS_ERR_ARG, "context length does not match positional_embedding");
    GLASS_HIP(hipSetDevice(e->cfg.device));
    const int W = e->t_width, heads = W / 64, M = n_texts * ctx, E = e->cfg.clip_embed;
    std::vector<int> eot(n_texts);
    for (int n = 0; n < n_texts; ++n) {       // text.argmax(dim=-1): EOT has the highest id (clip/model.py:318)
        int best = 0;
        for (int t = 0; t < ctx; ++t) {
            const int v = tokens[(size_t)n * ctx + t];
            REQUIRE(v >= 0 && v < e->t_vocab, GLASS_ERR_ARG, "token id out of range");
            if (v > tokens[(size_t)n * ctx + best]) best = t;
        }
        eot[n] = best;
    }
    auto& tw = e->twork;
    auto cleanup = [&]() { text_work_free(e); };
    hipError_t err = hipSuccess;
    if (tw.n_texts != n_texts) {          // (re)build the workspace for this batch size
        text_work_free(e);
        err = hipMalloc(&tw.d_tok, (size_t)M * sizeof(int));
        if (err == hipSuccess) err = hipMalloc(&tw.d_rows, (size_t)n_texts * sizeof(int));
        if (err == hipSuccess) err = hipMalloc(&tw.x, (size_t)M * W * sizeof(float));
        if (err == hipSuccess) err = hipMalloc(&tw.cls, (size_t)n_texts * W * sizeof(float));
        if (err == hipSuccess) err = hipMalloc(&tw.feat, (size_t)n_texts * E * sizeof(float));
        if (err == hipSuccess) err = hipMalloc(&tw.ln16, (size_t)M * W * sizeof(half_t));
        if (err == hipSuccess) err = hipMalloc(&tw.qkv, (size_t)M * 3 * W * sizeof(half_t));
        if (err == hipSuccess) err = hipMalloc(&tw.att, (size_t)M * W * sizeof(half_t));
        if (err == hipSuccess) err = hipMalloc(&tw.hid, (size_t)M * 4 * W * sizeof(half_t));
        if (err != hipSuccess) {
            cleanup();
            glass_set_error(std::string("encode_text: hipMalloc failed: ") + hipGetErrorString(err));
            return GLASS_ERR_NOMEM;
        }
        tw.n_texts = n_texts;
    }
    int* d_tok = tw.d_tok;
    float *x = tw.x, *cls = tw.cls, *feat = tw.feat;
    half_t *ln16 = tw.ln16, *qkv = tw.qkv, *att = tw.att, *hid = tw.hid;
    for (int n = 0; n < n_texts; ++n) eot[n] += n * ctx;          // row of each text's EOT token in x
    hipStream_t st = e->stream;
    hipMemcpyAsync(d_tok, tokens, (size_t)M * sizeof(int), hipMemcpyHostToDevice, st);
    hipMemcpyAsync(tw.d_rows, eot.data(), (size_t)n_texts * sizeof(int), hipMemcpyHostToDevice, st);
    launch_embed_text(d_tok, e->t_tok, e->t_pos, M, ctx, W, x, st);
    GemmParams g;
    for (auto& b : e->tblk) {
        launch_layernorm(x, W, M, W, b.ln1_g, b.ln1_b, ln16, nullptr, st);
        memset(&g, 0, sizeof g);
        g.a = ln16; g.w = b.w_qkv; g.M = M; g.N = 3 * W; g.K = W; g.bias = b.b_qkv; g.mode = 0; g.out16 = qkv; g.ldo = 3 * W; g.cand_rows = ctx;
        if (!launch_gemm_tiled(g, st)) launch_gemm_direct(g, st);
        launch_attention(qkv, n_texts, ctx, heads, 64, 1, att, st);
        memset(&g, 0, sizeof g);
        g.a = att; g.w = b.w_out; g.M = M; g.N = W; g.K = W; g.bias = b.b_out; g.mode = 2; g.out32 = x; g.ldo = W; g.cand_rows = ctx;
        if (!launch_gemm_tiled(g, st)) launch_gemm_direct(g, st);
        launch_layernorm(x, W, M, W, b.ln2_g, b.ln2_b, ln16, nullptr, st);
        memset(&g, 0, sizeof g);
        g.a = ln16; g.w = b.w_fc; g.M = M; g.N = 4 * W; g.K = W; g.bias = b.b_fc; g.mode = 1; g.out16 = hid; g.ldo = 4 * W; g.cand_rows = ctx;
        if (!launch_gemm_tiled(g, st)) launch_gemm_direct(g, st);
        memset(&g, 0, sizeof g);
        g.a = hid; g.w = b.w_proj; g.M = M; g.N = W; g.K = 4 * W; g.bias = b.b_proj; g.mode = 2; g.out32 = x; g.ldo = W; g.cand_rows = ctx;
        if (!launch_gemm_tiled(g, st)) launch_gemm_direct(g, st);
    }
    // ln_final on the EOT row of each text only (row-wise op): one launch over the gathered rows (round 4: it was one launch per text)
    launch_layernorm_rows(x, tw.d_rows, n_texts, W, e->t_lnf_g, e->t_lnf_b, cls, st);
    launch_dense(cls, W, n_texts, W, e->t_proj, E, nullptr, feat, E, 0, 0, nullptr, 0, st);
    err = hipMemcpyAsync(out_feat, feat, (size_t)n_texts * E * sizeof(float), hipMemcpyDeviceToHost, st);
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    if (err == hipSuccess) err = hipGetLastError();
    if (err != hipSuccess) {
        cleanup();
        glass_set_error(std::string("encode_text failed: ") + hipGetErrorString(err));
        return GLASS_ERR_HIP;
    }
    return GLASS_OK;
}

extern "C" int glass_engine_encode_image(glass_engine* e, const float* images, int32_t n, float* out_feat) {
    REQUIRE(e && images && out_feat && n > 0, GLASS_ERR_ARG, "null argument");
    REQUIRE(e->finalized, GLASS_ERR_STATE, "finalize() first");
    REQUIRE(n <= e->cfg.max_pop, GLASS_ERR_ARG, "more images than max_pop");
    const glass_config& c = e->cfg;
    GLASS_HIP(hipSetDevice(c.device));
    const size_t elems = (size_t)n * 3 * c.clip_res * c.clip_res;
    float* d_img = nullptr;
    GLASS_HIP(hipMalloc(&d_img, elems * sizeof(float)));
    hipError_t err = hipMemcpyAsync(d_img, images, elems * sizeof(float), hipMemcpyHostToDevice, e->stream);
    e->cur = e->stream;
    launch_image_patches(d_img, n, c.clip_res, c.clip_patch, e->d_patches, e->stream);
    run_clip(e, n, 0, 1 << 20);
    if (err == hipSuccess)
        err = hipMemcpyAsync(out_feat, e->d_feat, (size_t)n * c.clip_embed * sizeof(float), hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    if (err == hipSuccess) err = hipGetLastError();
    hipFree(d_img);
    e->prof_events.clear();
    e->event_next = 0;
    if (err != hipSuccess) {
        glass_set_error(std::string("encode_image failed: ") + hipGetErrorString(err));
        return GLASS_ERR_HIP;
    }
    return GLASS_OK;
}

static void gpt2_work_free(glass_engine* e) {
    auto& w = e->gwork;
    if (w.exec) hipGraphExecDestroy(w.exec);
    if (w.graph) hipGraphDestroy(w.graph);
    hipFree(w.d_tok); hipFree(w.d_gen); hipFree(w.d_state); hipFree(w.x); hipFree(w.ln); hipFree(w.qkv); hipFree(w.att); hipFree(w.hid);
    hipFree(w.last); hipFree(w.logits); hipFree(w.kc); hipFree(w.vc); hipFree(w.part); hipFree(w.stats); hipFree(w.pairs); hipFree(w.pst);
    w = glass_engine::Gpt2Work();
}

static int gpt2_decode_group(glass_engine* e, const int32_t* context, int32_t P, int32_t nctx, int32_t length, int32_t* out_tokens);

// Sequences are independent, and the single-token step kernels hold at most 64 rows: a longer population is decoded in row groups of
// 64, each through exactly the launches a 64-row call makes — a row's tokens do not depend on how many rows the call (or a shard) holds.
extern "C" int glass_engine_gpt2_decode(glass_engine* e, const int32_t* context, int32_t P, int32_t nctx, int32_t length,
                                        int32_t* out_tokens) {
    REQUIRE(e && context && out_tokens && P > 0 && nctx > 0 && length > 0, GLASS_ERR_ARG, "bad argument");
    float total_ms = 0.f;
    for (int g0 = 0; g0 < P; g0 += 64) {
        const int rc = gpt2_decode_group(e, context + (size_t)g0 * nctx, std::min(64, P - g0), nctx, length,
                                         out_tokens + (size_t)g0 * (nctx + length));
        if (rc) return rc;
        total_ms += e->gwork.last_ms;
    }
    e->gwork.last_ms = total_ms;
    e->last_ms = total_ms;
    return GLASS_OK;
}

static int gpt2_decode_group(glass_engine* e, const int32_t* context, int32_t P, int32_t nctx, int32_t length, int32_t* out_tokens) {
    REQUIRE(e && context && out_tokens && P > 0 && nctx > 0 && length > 0, GLASS_ERR_ARG, "bad argument");
    REQUIRE(e->finalized, GLASS_ERR_STATE, "finalize() first");
    REQUIRE(e->g_wte != nullptr, GLASS_ERR_STATE, "GPT-2 weights were not loaded (gpt2.transformer.*)");
    const int D = e->g_dim, V = e->g_vocab, heads = D / 64, Tmax = nctx + length;
    REQUIRE(Tmax <= e->g_npos && Tmax <= 256, GLASS_ERR_ARG, "sequence longer than the position table / 256");
    for (long long i = 0; i < (long long)P * nctx; ++i)
        REQUIRE(context[i] >= 0 && context[i] < V, GLASS_ERR_ARG, "token id out of range");
    GLASS_HIP(hipSetDevice(e->cfg.device));
    const int nl = (int)e->gblk.size();
    const size_t rows = (size_t)P * nctx;
    auto& w = e->gwork;
    hipStream_t st = e->stream;
    auto fits = [&](const glass_engine::Gpt2Work& g) { return g.P == P && g.nctx == nctx && g.length == length; };
    if (!fits(w)) std::swap(e->gwork, e->gwork_alt);   // second slot: a ragged population alternates between a 64-row group and its
                                                       // remainder — each keeps its buffers and its captured step graph
    if (!fits(w)) {      // (re)build the workspace for this geometry
        gpt2_work_free(e);
        // split-K scratch for the single-token steps (M = P <= 64 rows: each weight is streamed once per step, so the number
        // of workgroups pulling on HBM is what matters), device-resident tokens and step state {past length, step index}
        w.part_elems = (size_t)16 * P * 4 * D;
        hipError_t err = hipMalloc(&w.d_tok, rows * sizeof(int));
        if (err == hipSuccess) err = hipMalloc(&w.d_gen, (size_t)P * length * sizeof(int));
        if (err == hipSuccess) err = hipMalloc(&w.d_state, 3 * sizeof(int));      // {past length, step index, ticket counter of the fused step tail}
        if (err == hipSuccess) err = hipMalloc(&w.x, rows * D * sizeof(float));
        if (err == hipSuccess) err = hipMalloc(&w.ln, rows * D * sizeof(float));
        if (err == hipSuccess) err = hipMalloc(&w.qkv, rows * 3 * D * sizeof(float));
        if (err == hipSuccess) err = hipMalloc(&w.att, rows * D * sizeof(float));
        if (err == hipSuccess) err = hipMalloc(&w.hid, rows * 4 * D * sizeof(float));
        if (err == hipSuccess) err = hipMalloc(&w.last, (size_t)P * D * sizeof(float));
        if (err == hipSuccess) err = hipMalloc(&w.logits, (size_t)P * V * sizeof(float));
        if (err == hipSuccess) err = hipMalloc(&w.kc, (size_t)nl * P * Tmax * D * sizeof(float));
        if (err == hipSuccess) err = hipMalloc(&w.vc, (size_t)nl * P * Tmax * D * sizeof(float));
        if (err == hipSuccess) err = hipMalloc(&w.part, w.part_elems * sizeof(float));
        if (err == hipSuccess) err = hipMalloc(&w.pairs, (size_t)2 * P * ((V + 31) / 32) * sizeof(float));   // (max, index) per (row, 32-column block)
        if (err == hipSuccess) err = hipMalloc(&w.stats, (size_t)P * (2 + 2 * 32) * sizeof(float));
        if (err == hipSuccess) err = hipMalloc(&w.pst, (size_t)P * 24 * 2 * sizeof(float));   // row partials (mean, M2) of the complete-output step products   // + the two-stage arg-max's (value, index) pairs
        if (err != hipSuccess) {
            gpt2_work_free(e);
            glass_set_error(std::string("gpt2_decode: hipMalloc failed: ") + hipGetErrorString(err));
            return GLASS_ERR_NOMEM;
        }
        w.P = P; w.nctx = nctx; w.length = length;
    }
    // one transformer pass over `nd` new positions per sequence; step_state != nullptr: single-token step whose past length /
    // step index are read from device memory (the form that is captured into a hipGraph and replayed)
    // single-token steps, fused form (round 3; A/B knob GLASS_GPT2_NO_FUSE): LayerNorm applied on the activation operand of the next
    // product from row statistics, products with complete outputs (no split-K reduce launch) except the MLP's second one, whose
    // split-K slices of the residual products finished together with the residual add and the next statistics: 9 launches per layer
    // instead of 11 (complete-output products — 72 / 24 / 96 workgroups walking three chunks each — were 21 us against 9 + 4: dropped)
    static const bool no_fuse = glass_knob("GLASS_GPT2_NO_FUSE") != nullptr;
    // (the launcher's own shape conditions, asked through the launcher's predicate: a product it refuses returns 0 slices and nothing written)
    const bool fuse_ok = !no_fuse && gemm_f32_step_supported(P, D, D, true) && gemm_f32_step_supported(P, 4 * D, 4 * D, false);
    bool step_refused = false;
    static const bool no_attn_step = glass_knob("GLASS_GPT2_NO_ATTN_STEP") != nullptr;   // A/B knob
    // fused step tail (round 4): the pick also writes the NEXT step's embedding + first LayerNorm statistics and advances the state — a step
    // starts at layer 0's qkv product; the first step's embedding is one eager launch after the prefill.  A/B knob: GLASS_GPT2_NO_TAIL.
    static const bool no_head = glass_knob("GLASS_GPT2_NO_HEAD") != nullptr;   // A/B knob: generic product + two-stage arg-max
    static const bool no_tail = glass_knob("GLASS_GPT2_NO_TAIL") != nullptr;
    const bool tail_fused = fuse_ok && !no_head && !no_tail && D <= 1024 && gpt2_head_supported(P, V, D, D);
    auto pass = [&](int nd, int past, const int* step_state) {
        const int M = P * nd;
        // round 4 (gpt2.hip): the attention output product and the MLP's first product in the complete-output form — no slices, so no
        // gpt2_finalize / splitk_reduce launch behind them: 6 launches per layer instead of 8.  (The qkv product and the MLP's second one
        // stay split: complete, they were 12-14 us against 9.5 and 34 against 12 + 5.)  A/B knob: GLASS_GPT2_NO_ROWBLK.
        static const bool no_rowblk = glass_knob("GLASS_GPT2_NO_ROWBLK") != nullptr;
        const bool rowblk = step_state && fuse_ok && !no_rowblk && D % 32 == 0 && D / 32 <= 24 &&
                            gemm_f32_rowblk_supported(P, D, D, D, false, true) && gemm_f32_rowblk_supported(P, 4 * D, D, D, true, false);
        if (step_state) { if (!tail_fused) launch_gpt2_embed_step(w.d_gen, step_state, P, e->g_wte, e->g_wpe, D, w.x, st, fuse_ok ? w.stats : nullptr); }
        else launch_gpt2_embed(w.d_tok, e->g_wte, e->g_wpe, M, nd, past, D, w.x, st);
        if (step_state && fuse_ok) {      // (the embedding kernel left the first layer's LayerNorm statistics)
            for (int l = 0; l < nl; ++l) {
                const auto& b = e->gblk[l];
                float* kcl = w.kc + (size_t)l * P * Tmax * D;
                float* vcl = w.vc + (size_t)l * P * Tmax * D;
                int S = launch_gemm_f32_step(w.x, b.w_qkv, b.b_qkv, w.qkv, P, 3 * D, D, D, 3 * D, 0, st, w.part, w.part_elems, w.stats, b.ln1_g, b.ln1_b);
                step_refused |= S == 0;
                if (Tmax <= 64 && !no_attn_step) {      // one wave per (sequence, head); it sums the product's slices itself
                    launch_gpt2_attention_step(w.qkv, S > 1 ? w.part : nullptr, S, b.b_qkv, kcl, vcl, P, Tmax, heads, w.att, st, step_state);
                } else {
                    if (S > 1) launch_gpt2_reduce(w.part, S, b.b_qkv, w.qkv, P, 3 * D, 3 * D, 0, st);
                    launch_gpt2_attention(w.qkv, kcl, vcl, P, 1, past, Tmax, heads, w.att, st, step_state);
                }
                if (rowblk) {       // x += att @ Wo + b (row partials of LayerNorm 2 in the epilogue); hid = gelu(LN2(x) @ Wfc + b)
                    bool ok = launch_gemm_f32_rowblk(w.att, b.w_o, b.b_o, w.x, P, D, D, D, D, 2, st, nullptr, 0, nullptr, nullptr, w.pst);
                    ok &= launch_gemm_f32_rowblk(w.x, b.w_fc, b.b_fc, w.hid, P, 4 * D, D, D, 4 * D, 1, st, w.pst, D / 32, b.ln2_g, b.ln2_b, nullptr);
                    step_refused |= !ok;
                } else {
                    S = launch_gemm_f32_step(w.att, b.w_o, b.b_o, w.x, P, D, D, D, D, 2, st, w.part, w.part_elems, nullptr, nullptr, nullptr);
                    step_refused |= S == 0;
                    launch_gpt2_finalize(S > 1 ? w.part : nullptr, S, b.b_o, w.x, P, D, w.stats, st);       // residual + LayerNorm 2 statistics
                    S = launch_gemm_f32_step(w.x, b.w_fc, b.b_fc, w.hid, P, 4 * D, D, D, 4 * D, 1, st, w.part, w.part_elems, w.stats, b.ln2_g, b.ln2_b);
                    step_refused |= S == 0;
                    if (S > 1) launch_gpt2_reduce(w.part, S, b.b_fc, w.hid, P, 4 * D, 4 * D, 1, st);
                }
                S = launch_gemm_f32_step(w.hid, b.w_pr, b.b_pr, w.x, P, D, 4 * D, 4 * D, D, 2, st, w.part, w.part_elems, nullptr, nullptr, nullptr);
                step_refused |= S == 0;
                launch_gpt2_finalize(S > 1 ? w.part : nullptr, S, b.b_pr, w.x, P, D, w.stats, st);      // residual + next LayerNorm's statistics
            }
            if (tail_fused) {
                step_refused |= !launch_gpt2_head_tail(w.x, e->g_wte, P, V, D, D, w.stats, e->g_lnf_g, e->g_lnf_b, w.pairs, w.d_gen, w.d_state, e->g_wte, e->g_wpe,
                                                       w.x, w.stats, st);
                return;
            }
            if (no_head || !launch_gpt2_head(w.x, e->g_wte, P, V, D, D, w.stats, e->g_lnf_g, e->g_lnf_b, nullptr, w.pairs, w.d_gen, w.d_state, st)) {
                // ln_f fused; the real vocabulary (1571 column blocks) is never split, a small one may be
                const int S = launch_gemm_f32_step(w.x, e->g_wte, nullptr, w.logits, P, V, D, D, V, 0, st, w.part, w.part_elems, w.stats, e->g_lnf_g, e->g_lnf_b);
                step_refused |= S == 0;
                if (S > 1) launch_gpt2_reduce(w.part, S, nullptr, w.logits, P, V, V, 0, st);
                launch_argmax(w.logits, P, V, w.d_gen, st, w.d_state, w.stats + 2 * P);
            }
            launch_gpt2_advance(w.d_state, st);
            return;
        }
        for (int l = 0; l < nl; ++l) {
            const auto& b = e->gblk[l];
            float* kcl = w.kc + (size_t)l * P * Tmax * D;
            float* vcl = w.vc + (size_t)l * P * Tmax * D;
            launch_layernorm(w.x, D, M, D, b.ln1_g, b.ln1_b, nullptr, w.ln, st);
            launch_gemm_f32(w.ln, b.w_qkv, b.b_qkv, w.qkv, M, 3 * D, D, D, 3 * D, 0, st, w.part, w.part_elems, nd > 1);
            launch_gpt2_attention(w.qkv, kcl, vcl, P, nd, past, Tmax, heads, w.att, st, step_state);
            launch_gemm_f32(w.att, b.w_o, b.b_o, w.x, M, D, D, D, D, 2, st, w.part, w.part_elems, nd > 1);
            launch_layernorm(w.x, D, M, D, b.ln2_g, b.ln2_b, nullptr, w.ln, st);
            launch_gemm_f32(w.ln, b.w_fc, b.b_fc, w.hid, M, 4 * D, D, D, 4 * D, 1, st, w.part, w.part_elems, nd > 1);
            launch_gemm_f32(w.hid, b.w_pr, b.b_pr, w.x, M, D, 4 * D, 4 * D, D, 2, st, w.part, w.part_elems, nd > 1);
        }
        // ln_f on the last position of each sequence, tied lm_head, greedy pick -> d_gen[step][P]
        launch_layernorm(w.x + (size_t)(nd - 1) * D, (long long)nd * D, P, D, e->g_lnf_g, e->g_lnf_b, nullptr, w.last, st);
        launch_gemm_f32(w.last, e->g_wte, nullptr, w.logits, P, V, D, D, V, 0, st, w.part, w.part_elems);
        launch_argmax(w.logits, P, V, w.d_gen, st, w.d_state, w.stats + 2 * P);
        launch_gpt2_advance(w.d_state, st);
    };
    std::vector<int32_t> gen((size_t)P * length);
    GLASS_HIP(hipEventRecord(e->ev0, st));
    hipMemcpyAsync(w.d_tok, context, rows * sizeof(int), hipMemcpyHostToDevice, st);
    const int state0[3] = {0, 0, 0}, state1[3] = {nctx, 1, 0};
    hipMemcpyAsync(w.d_state, state0, sizeof state0, hipMemcpyHostToDevice, st);
    pass(nctx, 0, nullptr);                                 // prefill = step 0 (writes d_gen[0 .. P))
    hipMemcpyAsync(w.d_state, state1, sizeof state1, hipMemcpyHostToDevice, st);
    if (tail_fused && length > 1) launch_gpt2_embed_step(w.d_gen, w.d_state, P, e->g_wte, e->g_wpe, D, w.x, st, w.stats);   // step 1's embedding (later ones: the step tail)
    hipError_t err = hipSuccess;
    if (length > 1) {
        // the 29 single-token steps are the same ~190 launches each: capture one step once, replay it (launch latency, not work,
        // is what the un-graphed loop spent its time on)
        static const bool no_graph = glass_knob("GLASS_GPT2_NO_GRAPH") != nullptr;   // A/B knob
        if (!w.exec && !no_graph) {
            err = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            if (err == hipSuccess) {
                pass(1, 0, w.d_state);
                err = hipStreamEndCapture(st, &w.graph);
                if (err == hipSuccess) err = hipGraphInstantiate(&w.exec, w.graph, nullptr, nullptr, 0);
            }
            if (err != hipSuccess) {       // capture unavailable: eager steps (same kernels, same device-side state)
                (void)hipGetLastError();
                if (w.exec) { hipGraphExecDestroy(w.exec); w.exec = nullptr; }
                if (w.graph) { hipGraphDestroy(w.graph); w.graph = nullptr; }
                err = hipSuccess;
            }
        }
        for (int step = 1; step < length && err == hipSuccess; ++step) {
            if (w.exec) err = hipGraphLaunch(w.exec, st);
            else pass(1, 0, w.d_state);
        }
    }
    if (err == hipSuccess) err = hipEventRecord(e->ev1, st);
    if (err == hipSuccess) err = hipMemcpyAsync(gen.data(), w.d_gen, (size_t)P * length * sizeof(int), hipMemcpyDeviceToHost, st);
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    if (err == hipSuccess) err = hipGetLastError();
    if (err != hipSuccess) {
        gpt2_work_free(e);
        glass_set_error(std::string("gpt2_decode failed: ") + hipGetErrorString(err));
        return GLASS_ERR_HIP;
    }
    if (step_refused) {        // fuse_ok and the launcher disagreed about a shape: buffers were consumed unwritten — never a silent result
        gpt2_work_free(e);
        glass_set_error("gpt2_decode: a fused step product refused its shape (launch_gemm_f32_step returned 0)");
        return GLASS_ERR_STATE;
    }
    (void)hipEventElapsedTime(&w.last_ms, e->ev0, e->ev1);
    e->last_ms = w.last_ms;
    for (int p = 0; p < P; ++p) {
        for (int t = 0; t < nctx; ++t) out_tokens[(size_t)p * Tmax + t] = context[(size_t)p * nctx + t];
        for (int s2 = 0; s2 < length; ++s2) out_tokens[(size_t)p * Tmax + nctx + s2] = gen[(size_t)s2 * P + p];
    }
    return GLASS_OK;
}

extern "C" int glass_engine_evaluate(glass_engine* e, const float* latents, int32_t P, int32_t generation,
                                     int32_t first_minibatch, const glass_noise* noise, float* out_F) {
    REQUIRE(out_F, GLASS_ERR_ARG, "null out_F");
    return run_pass(e, latents, P, generation, first_minibatch, noise, out_F, nullptr);
}

extern "C" int glass_engine_generate(glass_engine* e, const float* latents, int32_t P, int32_t generation,
                                     int32_t first_minibatch, const glass_noise* noise, float* images) {
    REQUIRE(images, GLASS_ERR_ARG, "null images");
    return run_pass(e, latents, P, generation, first_minibatch, noise, nullptr, images);
}

extern "C" int glass_engine_last_details(glass_engine* e, int32_t P, float* features, float* dis, float* sim) {
    REQUIRE(e && e->finalized, GLASS_ERR_STATE, "engine not ready");
    REQUIRE(P > 0 && P <= e->last_P, GLASS_ERR_ARG, "P exceeds the last evaluated population");
    GLASS_HIP(hipSetDevice(e->cfg.device));
    if (features)
        GLASS_HIP(hipMemcpy(features, e->d_feat, (size_t)P * e->cfg.clip_embed * sizeof(float), hipMemcpyDeviceToHost));
    if (dis) GLASS_HIP(hipMemcpy(dis, e->d_dis, (size_t)P * sizeof(float), hipMemcpyDeviceToHost));
    if (sim) GLASS_HIP(hipMemcpy(sim, e->d_sim, (size_t)P * sizeof(float), hipMemcpyDeviceToHost));
    return GLASS_OK;
}

extern "C" int glass_engine_last_F_device(glass_engine* e, int32_t P, void** dev_ptr) {
    REQUIRE(e && dev_ptr, GLASS_ERR_ARG, "null argument");
    REQUIRE(P > 0 && P == e->last_P && e->d_F, GLASS_ERR_STATE, "no evaluate() of this population size has run");
    *dev_ptr = e->d_F;
    return GLASS_OK;
}

extern "C" int glass_engine_last_gpu_ms(glass_engine* e, float* ms) {
    REQUIRE(e && ms, GLASS_ERR_ARG, "null argument");
    *ms = e->last_ms;
    return GLASS_OK;
}

extern "C" int glass_engine_set_profiling(glass_engine* e, int32_t on) {
    REQUIRE(e, GLASS_ERR_ARG, "null engine");
    e->profiling = on != 0;
    return GLASS_OK;
}

extern "C" int glass_engine_set_overlap(glass_engine* e, int32_t on) {
    REQUIRE(e, GLASS_ERR_ARG, "null engine");
    // 0: one stream; 1: D + CLIP of chunk k under the synthesis of chunk k + 1; 2: CLIP on the second stream next to D (default)
    e->overlap = on == 1;
    e->clip_overlap = on == 2;
    return GLASS_OK;
}

extern "C" int glass_engine_set_biggan_tap(glass_engine* e, int32_t block) {
    REQUIRE(e, GLASS_ERR_ARG, "null engine");
    REQUIRE(e->cfg.generator == GLASS_GEN_BIGGAN_DEEP, GLASS_ERR_STATE, "not a BigGAN-deep engine");
    REQUIRE(block >= -2 && block < (int)e->bg.blocks.size(), GLASS_ERR_ARG, "no such GenBlock");
    e->bg_tap = block;
    e->bg_tap_data.clear();
    return GLASS_OK;
}

extern "C" int glass_engine_get_biggan_tap(glass_engine* e, float* out, int64_t capacity, int32_t dims[4]) {
    REQUIRE(e && dims, GLASS_ERR_ARG, "null argument");
    for (int i = 0; i < 4; ++i) dims[i] = e->bg_tap_dims[i];
    REQUIRE(!e->bg_tap_data.empty(), GLASS_ERR_STATE, "no tap recorded (set_biggan_tap, then evaluate / generate)");
    if (out) {
        REQUIRE(capacity >= (int64_t)e->bg_tap_data.size(), GLASS_ERR_ARG, "tap buffer too small");
        memcpy(out, e->bg_tap_data.data(), e->bg_tap_data.size() * sizeof(float));
    }
    return GLASS_OK;
}

extern "C" int glass_engine_set_profile_filter(glass_engine* e, const char* kernel_substr) {
    REQUIRE(e, GLASS_ERR_ARG, "null engine");
    e->prof_filter = kernel_substr ? kernel_substr : "";
    return GLASS_OK;
}

extern "C" int glass_engine_get_profile(glass_engine* e, glass_prof_row* rows, int32_t max_rows, int32_t* n_rows) {
    REQUIRE(e && n_rows, GLASS_ERR_ARG, "null argument");
    const int n = (int)e->prof_rows.size();
    *n_rows = n;
    if (rows)
        for (int i = 0; i < std::min(n, (int)max_rows); ++i) rows[i] = e->prof_rows[i];
    return GLASS_OK;
}

extern "C" int glass_device_info(int32_t device, char* name, int32_t name_len, int32_t* cus, int64_t* hbm_bytes) {
    hipDeviceProp_t prop;
    GLASS_HIP(hipGetDeviceProperties(&prop, device));
    if (name && name_len > 0)   // the marketing name the driver reports can be generic ("AMD Radeon Graphics"): the ISA name says what it is
        snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return GLASS_OK;
}
