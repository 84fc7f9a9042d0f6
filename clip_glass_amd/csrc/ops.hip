// ops.hip — diagnostic per-kernel C ABI (include/glass_ops.h): host fp32 buffers in/out.
#include <string.h>

#include <string>
#include <vector>

#include "../../include/glass_ops.h"
#include "engine.h"
#include "kernels.h"

namespace {
struct Dev {
    std::vector<void*> ptrs;
    ~Dev() {
        for (void* p : ptrs) hipFree(p);
    }
    template <typename T>
    T* alloc(size_t n) {
        void* q = nullptr;
        if (hipMalloc(&q, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
        ptrs.push_back(q);
        return (T*)q;
    }
    half_t* up16(const float* src, size_t n) {
        if (!src) return nullptr;
        std::vector<_Float16> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = (_Float16)src[i];
        half_t* d = alloc<half_t>(n);
        if (d) hipMemcpy(d, h.data(), n * sizeof(half_t), hipMemcpyHostToDevice);
        return d;
    }
    half_t* up16v(const std::vector<_Float16>& h) {
        half_t* d = alloc<half_t>(h.size());
        if (d) hipMemcpy(d, h.data(), h.size() * sizeof(half_t), hipMemcpyHostToDevice);
        return d;
    }
    float* up32(const float* src, size_t n) {
        if (!src) return nullptr;
        float* d = alloc<float>(n);
        if (d) hipMemcpy(d, src, n * sizeof(float), hipMemcpyHostToDevice);
        return d;
    }
};
int down16(float* dst, const half_t* src, size_t n) {
    std::vector<_Float16> h(n);
    GLASS_HIP(hipMemcpy(h.data(), src, n * sizeof(half_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) dst[i] = (float)h[i];
    return GLASS_OK;
}
int finish() {
    GLASS_HIP(hipDeviceSynchronize());
    GLASS_HIP(hipGetLastError());
    return GLASS_OK;
}
}  // namespace

#define OPREQ(cond, msg)               \
    do {                               \
        if (!(cond)) {                 \
            glass_set_error(msg);      \
            return GLASS_ERR_ARG;      \
        }                              \
    } while (0)

extern "C" int glass_op_conv(int32_t device, const glass_conv_desc* d) {
    OPREQ(d && d->x && d->w && d->y, "null argument");
    OPREQ(d->Cin % 16 == 0, "Cin must be a multiple of 16");
    OPREQ(d->out_scale > 0.f, "out_scale must be positive (the epilogues fold it into the activation constants: max(v k1, v k2))");
    GLASS_HIP(hipSetDevice(device));
    Dev dv;
    ConvParams p;
    memset(&p, 0, sizeof p);
    const size_t xin = (size_t)(d->broadcast_x ? 1 : d->B) * d->H * d->W * d->Cin;
    p.x = dv.up16(d->x, xin);
    p.x_bstride = d->broadcast_x ? 0 : (long long)d->H * d->W * d->Cin;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin;
    p.KS = d->KS; p.stride = d->stride; p.pad = d->pad; p.up = d->up;
    p.Cout = d->Cout; p.Neff = d->up ? 4 * d->Cout : d->Cout;
    p.Ho = d->Ho; p.Wo = d->Wo;
    p.Hc = d->up ? d->H : d->Ho; p.Wc = d->up ? d->W : d->Wo;
    std::vector<_Float16> pk;
    if (d->up) {
        OPREQ(d->KS == 3 && d->stride == 1 && d->pad == 1 && d->Ho == 2 * d->H, "up conv expects KS 3 / pad 1");
        glass_fold_upconv(d->w, d->Cout, d->Cin, pk);
    } else {
        glass_pack_conv(d->w, d->Cout, d->Cin, d->KS, d->Cin, pk);
    }
    p.w = dv.up16v(pk);
    if (d->up) {
        std::vector<_Float16> pk2;
        glass_pack_conv(d->w, d->Cout, d->Cin, 3, d->Cin, pk2);
        p.w_up = dv.up16v(pk2);
    }
    p.sn = dv.up32(d->sn, (size_t)d->B * d->Cin); p.sn_stride = d->Cin;
    p.sn16 = dv.up16(d->sn, (size_t)d->B * d->Cin);
    p.dscale = dv.up32(d->dscale, (size_t)d->B * d->Cout); p.ds_stride = d->Cout;
    p.batch_size = d->batch_size > 0 ? d->batch_size : 1;
    p.noise = dv.up32(d->noise, (size_t)(d->B / p.batch_size) * d->Ho * d->Wo);
    p.noise_strength = d->noise_strength;
    p.bias = dv.up32(d->bias, d->Cout);
    p.act = d->act;
    const size_t nout = (size_t)d->B * d->Ho * d->Wo * d->Cout;
    p.res = dv.up16(d->res, nout);
    p.out_scale = d->out_scale;
    half_t* y = dv.alloc<half_t>(nout);
    p.y = y;
    OPREQ(p.x && p.w && y, "device allocation failed");
    if (d->skip_x) {
        OPREQ(d->skip_w && (d->impl == 2 || d->impl == 5), "fused skip branch: impl 2 / 5 with skip_x and skip_w");
        std::vector<_Float16> pks;
        glass_pack_conv(d->skip_w, d->Cout, d->Cin, 1, d->Cin, pks);
        p.skip_w = dv.up16v(pks);
        p.skip_x = dv.up16(d->skip_x, (size_t)d->B * d->Ho * d->Wo * d->Cin);
    }
    half_t* xs_dev = nullptr;
    const size_t nxs = (size_t)d->B * (d->H / 2) * (d->W / 2) * d->Cin;
    if (d->xs_out) {
        OPREQ(d->impl == 2 || d->impl == 5, "blur-down by-product: impl 2 / 5");
        xs_dev = dv.alloc<half_t>(nxs);
        p.xs_out = xs_dev;
    }
    float* yrgb = nullptr;
    const size_t nrgb = (size_t)d->B * 3 * d->Ho * d->Wo;
    if (d->trgb_yout) {
        OPREQ((d->impl == 4 || d->impl == 2 || d->impl == 5) && d->trgb_w && d->trgb_b && d->trgb_sn && d->trgb_smax,
              "fused toRGB: impl 2 / 4 / 5 and all of w / b / sn / smax");
        p.trgb_w = dv.up32(d->trgb_w, 3 * (size_t)d->Cout); p.trgb_b = dv.up32(d->trgb_b, 3);
        p.trgb_sn = dv.up32(d->trgb_sn, (size_t)d->B * d->Cout); p.trgb_sn_stride = d->Cout;
        p.trgb_smax = dv.up32(d->trgb_smax, d->B); p.trgb_smax_stride = 1;
        p.trgb_yprev = dv.up32(d->trgb_yprev, (size_t)d->B * 3 * (d->Ho / 2) * (d->Wo / 2));
        yrgb = dv.alloc<float>(nrgb);
        p.trgb_yout = yrgb;
        if (d->impl == 4) {
            p.y = nullptr;                       // the streaming form never stores the map
        } else {                                 // tiled / LDS-DMA forms: weight tables from the table kernel
            half_t* tab = dv.alloc<half_t>((size_t)d->B * 32 * d->Cout);
            launch_trgb_tables(p.trgb_w, p.trgb_sn, p.trgb_sn_stride, p.trgb_smax, p.trgb_smax_stride, d->B, d->Cout, tab, 0);
            p.trgb_tab = tab;
        }
    }
    p.x_planar8 = d->x_planar8;
    p.x_planar32 = d->x_planar32;
    if (d->impl == 1) { if (!launch_conv_direct(p, 0)) { glass_set_error("direct conv: unsupported launch"); return GLASS_ERR_ARG; } }
    else if (d->impl == 3) {
        if (!launch_upconv_fused(p, 0)) { glass_set_error("fused up-conv: unsupported shape"); return GLASS_ERR_ARG; }
    } else if (d->impl == 2) {
        if (!launch_conv_tiled(p, 0)) { glass_set_error("tiled conv: unsupported shape"); return GLASS_ERR_ARG; }
    } else if (d->impl == 4) {
        if (!launch_conv_stream(p, 0)) { glass_set_error("streaming conv: unsupported shape"); return GLASS_ERR_ARG; }
    } else if (d->impl == 6) {
        const long long cap_a = (long long)p.Hc * p.Wc * p.KS * p.KS * p.Cin, cap_c = (long long)p.Hc * p.Wc * p.Neff;   // per candidate
        half_t* wa = dv.alloc<half_t>((size_t)(cap_a * p.B));
        float* wc = dv.alloc<float>((size_t)(cap_c * p.B));
        if (!launch_conv_gemm(p, wa, cap_a, wc, cap_c, 0)) { glass_set_error("im2col + GEMM conv: unsupported shape"); return GLASS_ERR_ARG; }
    } else if (d->impl == 5) {
        if (p.skip_x) {
            if (!launch_conv_s2(p, 0, true)) { glass_set_error("LDS-DMA stride-2 conv: unsupported shape"); return GLASS_ERR_ARG; }
        } else if (!launch_conv_glds(p, 0, true)) { glass_set_error("LDS-DMA conv: unsupported shape"); return GLASS_ERR_ARG; }
    } else if (!(d->up && launch_upconv_fused(p, 0)) && !launch_conv_stream(p, 0) && !launch_conv_tiled(p, 0) && !launch_conv_direct(p, 0)) {
        glass_set_error("no kernel accepts this convolution");
        return GLASS_ERR_ARG;
    }
    int rc = finish();
    if (rc) return rc;
    if (yrgb) {
        GLASS_HIP(hipMemcpy(d->trgb_yout, yrgb, nrgb * sizeof(float), hipMemcpyDeviceToHost));
        return p.y ? down16(d->y, y, nout) : GLASS_OK;      // (the forms that also store the feature map hand it back too)
    }
    if (xs_dev && (rc = down16(d->xs_out, xs_dev, nxs))) return rc;
    return down16(d->y, y, nout);
}

extern "C" int glass_op_gemm(int32_t device, int32_t M, int32_t N, int32_t K, const float* a, const float* w,
                             const float* bias, int32_t mode, int32_t impl, float* out) {
    OPREQ(a && w && out && K % 16 == 0, "bad argument");
    GLASS_HIP(hipSetDevice(device));
    Dev dv;
    GemmParams g;
    memset(&g, 0, sizeof g);
    g.a = dv.up16(a, (size_t)M * K); g.w = dv.up16(w, (size_t)N * K); g.M = M; g.N = N; g.K = K;
    g.bias = dv.up32(bias, N); g.mode = mode; g.ldo = N;
    const size_t n = (size_t)M * N;
    half_t* o16 = nullptr; float* o32 = nullptr;
    if (mode <= 1) { o16 = dv.alloc<half_t>(n); g.out16 = o16; }
    else { o32 = mode == 2 ? dv.up32(out, n) : dv.alloc<float>(n); g.out32 = o32; }
    if (impl == 1) launch_gemm_direct(g, 0);
    else if (impl == 2) {
        if (!launch_gemm_tiled(g, 0)) { glass_set_error("tiled gemm: unsupported shape"); return GLASS_ERR_ARG; }
    } else if (!launch_gemm_tiled(g, 0)) launch_gemm_direct(g, 0);
    int rc = finish();
    if (rc) return rc;
    if (o16) return down16(out, o16, n);
    GLASS_HIP(hipMemcpy(out, o32, n * sizeof(float), hipMemcpyDeviceToHost));
    return GLASS_OK;
}

extern "C" int glass_op_dense(int32_t device, int32_t P, int32_t K, int32_t N, const float* x, const float* wt,
                              const float* bias, int32_t in_sq, int32_t mode, const float* eps_row, float* out) {
    OPREQ(x && wt && out, "null argument");
    GLASS_HIP(hipSetDevice(device));
    Dev dv;
    float* dx = dv.up32(x, (size_t)P * K); float* dw = dv.up32(wt, (size_t)K * N); float* db = dv.up32(bias, N);
    float* de = dv.up32(eps_row, P); float* dout = dv.alloc<float>((size_t)P * N);
    launch_dense(dx, K, P, K, dw, N, db, dout, N, in_sq, mode, de, 1, 0);
    int rc = finish();
    if (rc) return rc;
    GLASS_HIP(hipMemcpy(out, dout, (size_t)P * N * sizeof(float), hipMemcpyDeviceToHost));
    return GLASS_OK;
}

extern "C" int glass_op_torgb(int32_t device, int32_t B, int32_t H, int32_t C, const float* x, const float* wrgb,
                              const float* bias, const float* sn, const float* smax, const float* yprev, float* yout) {
    OPREQ(x && wrgb && bias && sn && smax && yout, "null argument");
    GLASS_HIP(hipSetDevice(device));
    Dev dv;
    half_t* dx = dv.up16(x, (size_t)B * H * H * C);
    float* dw = dv.up32(wrgb, 3 * C); float* db = dv.up32(bias, 3); float* ds = dv.up32(sn, (size_t)B * C);
    float* dm = dv.up32(smax, B); float* dp = dv.up32(yprev, (size_t)B * 3 * (H / 2) * (H / 2));
    float* dy = dv.alloc<float>((size_t)B * 3 * H * H);
    OPREQ(launch_torgb(dx, B, H, H, C, dw, db, ds, C, dm, 1, dp, dy, 0), "toRGB: channel width not instantiated");
    int rc = finish();
    if (rc) return rc;
    GLASS_HIP(hipMemcpy(yout, dy, (size_t)B * 3 * H * H * sizeof(float), hipMemcpyDeviceToHost));
    return GLASS_OK;
}

extern "C" int glass_op_blur(int32_t device, int32_t mode, int32_t B, int32_t H, int32_t C, const float* x, float* out) {
    OPREQ(x && out && C % 8 == 0, "bad argument");
    GLASS_HIP(hipSetDevice(device));
    Dev dv;
    half_t* dx = dv.up16(x, (size_t)B * H * H * C);
    OPREQ(mode != 2 || blur_pad2_planar32_ok(C), "blur in 32-channel planes: C must be 32 x a power of two, <= 512");
    const int Ho = mode != 1 ? H + 1 : H / 2;
    half_t* dy = dv.alloc<half_t>((size_t)B * Ho * Ho * C);
    if (mode != 1) launch_blur_pad2(dx, B, H, H, C, dy, 0, mode == 2);
    else launch_blur_down(dx, B, H, H, C, dy, 0);
    int rc = finish();
    if (rc) return rc;
    return down16(out, dy, (size_t)B * Ho * Ho * C);
}

extern "C" int glass_op_dblock_down(int32_t device, int32_t B, int32_t R, int32_t Cin, int32_t Cout, const float* h,
                                    const float* x, const float* w1, const float* wskip, const float* b1, float* y) {
    OPREQ(h && x && w1 && wskip && b1 && y, "bad argument");
    GLASS_HIP(hipSetDevice(device));
    Dev dv;
    half_t* dh = dv.up16(h, (size_t)B * R * R * Cin);
    half_t* dx = dv.up16(x, (size_t)B * R * R * Cin);
    std::vector<_Float16> pk;
    glass_pack_conv(w1, Cout, Cin, 3, Cin, pk);
    half_t* dw1 = dv.up16v(pk);
    glass_pack_conv(wskip, Cout, Cin, 1, Cin, pk);
    half_t* dws = dv.up16v(pk);
    float* db = dv.up32(b1, Cout);
    const int Ro = R / 2;
    half_t* dxs = dv.alloc<half_t>((size_t)B * Ro * Ro * Cin);
    half_t* dy = dv.alloc<half_t>((size_t)B * Ro * Ro * Cout);
    OPREQ(conv_down_supported(R, Cin, Cout), "conv_down: unsupported shape");
    launch_blur_down(dx, B, R, R, Cin, dxs, 0);          // the skip branch's FIR (pad 1) + ::2 (conv_stream<fromrgb> emits it in the engine)
    OPREQ(launch_conv_down(dh, dxs, dw1, dws, db, dy, B, R, Cin, Cout, 0) != nullptr, "conv_down: unsupported shape");
    int rc = finish();
    if (rc) return rc;
    return down16(y, dy, (size_t)B * Ro * Ro * Cout);
}

extern "C" int glass_op_dblock0(int32_t device, int32_t B, int32_t R, int32_t impl, const float* y, const float* frgb_w, const float* frgb_b,
                                const float* w0, const float* b0, const float* w1, const float* wskip, const float* b1, float* out) {
    OPREQ(y && frgb_w && frgb_b && w0 && b0 && w1 && wskip && b1 && out && B > 0 && R > 0, "bad argument");
    GLASS_HIP(hipSetDevice(device));
    Dev dv;
    const int Cin = 32, Cout = 64, Ro = R / 2;
    float* dy = dv.up32(y, (size_t)B * 3 * R * R);
    float* dfw = dv.up32(frgb_w, Cin * 3); float* dfb = dv.up32(frgb_b, Cin);
    std::vector<_Float16> pk;
    glass_pack_conv(w0, Cin, Cin, 3, Cin, pk);
    half_t* dw0 = dv.up16v(pk);
    glass_pack_conv(w1, Cout, Cin, 3, Cin, pk);
    half_t* dw1 = dv.up16v(pk);
    glass_pack_conv(wskip, Cout, Cin, 1, Cin, pk);
    half_t* dws = dv.up16v(pk);
    float* db0 = dv.up32(b0, Cin); float* db1 = dv.up32(b1, Cout);
    half_t* dout = dv.alloc<half_t>((size_t)B * Ro * Ro * Cout);
    OPREQ(dy && dw0 && dw1 && dws && dout, "device allocation failed");
    if (impl == 0 || impl == 2) {          // conv_d0.hip: the whole block in one kernel (2: chunk-planar output)
        OPREQ(launch_dblock0(dy, dfw, dfb, dw0, db0, dw1, dws, db1, dout, B, R, Cin, Cout, 0, impl == 2) != nullptr, "dblock0: unsupported shape");
    } else {                  // the two-kernel form it replaces: conv_stream<fromrgb> (h + the skip input to HBM) + conv_down
        half_t* dh = dv.alloc<half_t>((size_t)B * R * R * Cin);
        half_t* dxs = dv.alloc<half_t>((size_t)B * Ro * Ro * Cin);
        OPREQ(dh && dxs, "device allocation failed");
        ConvParams p = conv_defaults();
        p.x = dh; p.x_bstride = (long long)R * R * Cin; p.B = B; p.H = p.W = R; p.Cin = Cin; p.Hc = p.Wc = R; p.KS = 3; p.pad = 1;
        p.w = dw0; p.Cout = p.Neff = Cin; p.Ho = p.Wo = R; p.bias = db0; p.act = 1; p.y = dh;
        p.rgb_y = dy; p.rgb_w = dfw; p.rgb_b = dfb; p.rgb_xs_out = dxs;
        OPREQ(launch_conv_stream(p, 0) != nullptr, "dblock0 (two-kernel form): conv_stream<fromrgb> does not take this shape");
        OPREQ(launch_conv_down(dh, dxs, dw1, dws, db1, dout, B, R, Cin, Cout, 0) != nullptr, "dblock0 (two-kernel form): conv_down does not take this shape");
    }
    int rc = finish();
    if (rc) return rc;
    return down16(out, dout, (size_t)B * Ro * Ro * Cout);
}

extern "C" int glass_op_fromrgb(int32_t device, int32_t B, int32_t R, int32_t Cout, const float* y, const float* w,
                                const float* bias, float* out) {
    OPREQ(y && w && bias && out && Cout % 8 == 0, "bad argument");
    GLASS_HIP(hipSetDevice(device));
    Dev dv;
    float* dy = dv.up32(y, (size_t)B * 3 * R * R); float* dw = dv.up32(w, Cout * 3); float* db = dv.up32(bias, Cout);
    half_t* dout = dv.alloc<half_t>((size_t)B * R * R * Cout);
    launch_fromrgb(dy, B, R, Cout, dw, db, dout, 0);
    int rc = finish();
    if (rc) return rc;
    return down16(out, dout, (size_t)B * R * R * Cout);
}

extern "C" int glass_op_mbstd(int32_t device, int32_t B, int32_t hw, int32_t C, int32_t Cpad, int32_t batch_size,
                              int32_t group, const float* x, float* out) {
    OPREQ(x && out && B % batch_size == 0 && batch_size % group == 0 && group <= 8, "bad argument");
    GLASS_HIP(hipSetDevice(device));
    Dev dv;
    half_t* dx = dv.up16(x, (size_t)B * hw * C);
    half_t* dy = dv.alloc<half_t>((size_t)B * hw * Cpad);
    launch_mbstd(dx, B, hw, C, Cpad, batch_size, group, 1e-8f, dy, 0);
    int rc = finish();
    if (rc) return rc;
    return down16(out, dy, (size_t)B * hw * Cpad);
}

extern "C" int glass_op_resize(int32_t device, int32_t B, int32_t R, int32_t S, int32_t ps, const float* y, float* patches) {
    OPREQ(y && patches && S % ps == 0, "bad argument");
    GLASS_HIP(hipSetDevice(device));
    Dev dv;
    float* dy = dv.up32(y, (size_t)B * 3 * R * R);
    const size_t n = (size_t)B * 3 * S * S;
    half_t* dp = dv.alloc<half_t>(n);
    launch_resize_patches(dy, B, R, S, ps, dp, 0);
    int rc = finish();
    if (rc) return rc;
    return down16(patches, dp, n);
}

extern "C" int glass_op_layernorm(int32_t device, int32_t M, int32_t D, const float* x, const float* g, const float* b, float* out) {
    OPREQ(x && g && b && out, "null argument");
    GLASS_HIP(hipSetDevice(device));
    Dev dv;
    float* dx = dv.up32(x, (size_t)M * D); float* dg = dv.up32(g, D); float* db = dv.up32(b, D);
    float* dout = dv.alloc<float>((size_t)M * D);
    launch_layernorm(dx, D, M, D, dg, db, nullptr, dout, 0);
    int rc = finish();
    if (rc) return rc;
    GLASS_HIP(hipMemcpy(out, dout, (size_t)M * D * sizeof(float), hipMemcpyDeviceToHost));
    return GLASS_OK;
}

extern "C" int glass_op_attention(int32_t device, int32_t n_img, int32_t L, int32_t heads, int32_t causal,
                                  const float* qkv, float* out) {
    OPREQ(qkv && out && L <= 128, "bad argument");
    GLASS_HIP(hipSetDevice(device));
    Dev dv;
    const int D = heads * 64;
    half_t* dq = dv.up16(qkv, (size_t)n_img * L * 3 * D);
    half_t* dout = dv.alloc<half_t>((size_t)n_img * L * D);
    launch_attention(dq, n_img, L, heads, 64, causal, dout, 0);
    int rc = finish();
    if (rc) return rc;
    return down16(out, dout, (size_t)n_img * L * D);
}

extern "C" int glass_op_noise(int32_t device, int32_t n_mb, int32_t hw, uint32_t layer, uint32_t mb0,
                              uint32_t generation, uint64_t seed, float* out) {
    OPREQ(out, "null argument");
    GLASS_HIP(hipSetDevice(device));
    Dev dv;
    float* d = dv.alloc<float>((size_t)n_mb * hw);
    launch_noise(d, n_mb, hw, layer, mb0, generation, seed, 0);
    int rc = finish();
    if (rc) return rc;
    GLASS_HIP(hipMemcpy(out, d, (size_t)n_mb * hw * sizeof(float), hipMemcpyDeviceToHost));
    return GLASS_OK;
}

__global__ void mfma_probe_kernel(const half_t* a, const half_t* b, float* d) {
    const int lane = threadIdx.x;
    const int r = lane & 31, kh = lane >> 5;
    h8 av, bv;
    for (int j = 0; j < 8; ++j) {
        av[j] = a[r * 16 + kh * 8 + j];       // A[i=r][k]
        bv[j] = b[(kh * 8 + j) * 32 + r];     // B[k][n=r]
    }
    f16x acc;
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    acc = mfma32(av, bv, acc);
    for (int reg = 0; reg < 16; ++reg) d[mfma32_row(reg, lane) * 32 + (lane & 31)] = acc[reg];
}
extern "C" int glass_op_mfma_probe(int32_t device, const float* a, const float* b, float* d) {
    OPREQ(a && b && d, "null argument");
    GLASS_HIP(hipSetDevice(device));
    Dev dv;
    half_t* da = dv.up16(a, 32 * 16); half_t* db = dv.up16(b, 16 * 32); float* dd = dv.alloc<float>(32 * 32);
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, 0, da, db, dd);
    int rc = finish();
    if (rc) return rc;
    GLASS_HIP(hipMemcpy(d, dd, 32 * 32 * sizeof(float), hipMemcpyDeviceToHost));
    return GLASS_OK;
}

extern "C" int glass_host_pack_conv(const float* w, int32_t Cout, int32_t Cin, int32_t KS, int32_t up, float* out) {
    OPREQ(w && out, "null argument");
    std::vector<_Float16> pk;
    if (up) {
        OPREQ(KS == 3, "up conv is 3x3");
        glass_fold_upconv(w, Cout, Cin, pk);
    } else {
        glass_pack_conv(w, Cout, Cin, KS, Cin, pk);
    }
    for (size_t i = 0; i < pk.size(); ++i) out[i] = (float)pk[i];
    return GLASS_OK;
}
