"""ctypes binding of the diagnostic per-kernel ABI (include/glass_ops.h).

Used by tests/ to check each HIP kernel family against the oracle in isolation.
Activations are NHWC float32 on the host; conversion to the kernels' fp16 layouts
happens inside the library.
"""
import ctypes as C

import numpy as np

from .engine import _check, _f32, _fp, load_library


class ConvDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
                ("KS", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("up", C.c_int32),
                ("Ho", C.c_int32), ("Wo", C.c_int32), ("broadcast_x", C.c_int32), ("act", C.c_int32),
                ("batch_size", C.c_int32), ("impl", C.c_int32),
                ("noise_strength", C.c_float), ("out_scale", C.c_float),
                ("x", C.POINTER(C.c_float)), ("w", C.POINTER(C.c_float)), ("sn", C.POINTER(C.c_float)),
                ("dscale", C.POINTER(C.c_float)), ("noise", C.POINTER(C.c_float)), ("bias", C.POINTER(C.c_float)),
                ("res", C.POINTER(C.c_float)), ("y", C.POINTER(C.c_float)),
                ("trgb_w", C.POINTER(C.c_float)), ("trgb_b", C.POINTER(C.c_float)), ("trgb_sn", C.POINTER(C.c_float)),
                ("trgb_smax", C.POINTER(C.c_float)), ("trgb_yprev", C.POINTER(C.c_float)),
                ("trgb_yout", C.POINTER(C.c_float)),
                ("skip_x", C.POINTER(C.c_float)), ("skip_w", C.POINTER(C.c_float)), ("xs_out", C.POINTER(C.c_float)),
                ("x_planar8", C.c_int32), ("x_planar32", C.c_int32)]


def to_planar8(a):
    """[B,H,W,C] -> the chunk-planar layout [B,C/8,H,W,8] conv_wreg's producers write (csrc/common.h x_planar8)."""
    B, H, W, Cc = a.shape
    return np.ascontiguousarray(a.reshape(B, H, W, Cc // 8, 8).transpose(0, 3, 1, 2, 4))


def to_planar32(a):
    """[B,H,W,C] -> 32-channel planes [B,C/32,H,W,32]: the layout the pad-2 blur writes for conv_s2 (csrc/common.h x_planar32)."""
    B, H, W, Cc = a.shape
    return np.ascontiguousarray(a.reshape(B, H, W, Cc // 32, 32).transpose(0, 3, 1, 2, 4))


def from_planar32(a, B, H, W, Cc):
    return np.ascontiguousarray(a.reshape(B, Cc // 32, H, W, 32).transpose(0, 2, 3, 1, 4)).reshape(B, H, W, Cc)


def from_planar8(a, B, H, W, Cc):
    return np.ascontiguousarray(a.reshape(B, Cc // 8, H, W, 8).transpose(0, 2, 3, 1, 4)).reshape(B, H, W, Cc)


def _opt(a):
    if a is None:
        return None, None
    a = _f32(a)
    return a, _fp(a)


def conv(x, w, *, stride=1, pad=None, up=False, sn=None, dscale=None, noise=None, noise_strength=0.0,
         batch_size=1, bias=None, act=False, res=None, out_scale=1.0, impl=0, broadcast_x=False, B=None, device=0,
         torgb=None, skip=None, xs_out=None, planar_x=False, both=False, planar32_x=False):
    """x [B,H,W,Cin] NHWC; w [Cout,Cin,KS,KS] (reference layout).  Returns y [B,Ho,Wo,Cout].
    torgb = dict(w [3,Cout], b [3], sn [B,Cout], smax [B], yprev [B,3,Ho/2,Wo/2] or None) with impl=4: the fused conv + toRGB
    form of the streaming kernel — returns the skip image [B,3,Ho,Wo] instead of y.
    planar_x: the device gets x chunk-planar (the permutation happens here; impl 5, 64 -> 64)."""
    lib = load_library()
    x = _f32(x); w = _f32(w)
    Bx, H, W, Cin = x.shape
    if planar_x:
        x = to_planar8(x)
    if planar32_x:            # conv_s2's input in 32-channel planes (permuted here)
        x = to_planar32(x)
    B = B if B is not None else Bx
    Cout, _, KS, _ = w.shape
    pad = (KS // 2) if pad is None else pad
    if up:
        Ho, Wo = 2 * H, 2 * W
    else:
        Ho, Wo = (H + 2 * pad - KS) // stride + 1, (W + 2 * pad - KS) // stride + 1
    y = np.empty((B, Ho, Wo, Cout), dtype=np.float32)
    d = ConvDesc()
    d.B, d.H, d.W, d.Cin, d.Cout = B, H, W, Cin, Cout
    d.KS, d.stride, d.pad, d.up, d.Ho, d.Wo = KS, stride, pad, int(up), Ho, Wo
    d.broadcast_x, d.act, d.batch_size, d.impl = int(broadcast_x), int(act), batch_size, impl
    d.noise_strength, d.out_scale = noise_strength, out_scale
    d.x_planar8 = int(planar_x)
    d.x_planar32 = int(planar32_x)
    keep = []
    d.x, d.w, d.y = _fp(x), _fp(w), _fp(y)
    for name, val in (("sn", sn), ("dscale", dscale), ("noise", noise), ("bias", bias), ("res", res)):
        a, p = _opt(val)
        keep.append(a)
        if p is not None:
            setattr(d, name, p)
    if skip is not None:      # (skip_x [B,Ho,Wo,Cin], skip_w [Cout,Cin,1,1]): the D block's 1x1 skip conv fused as extra K stages
        for name, val in zip(("skip_x", "skip_w"), skip):
            a, p = _opt(val)
            keep.append(a)
            setattr(d, name, p)
    if xs_out is not None:    # float32 [B,H/2,W/2,Cin] array that receives the blur-down by-product (impl 2, 64 -> 64)
        d.xs_out = _fp(xs_out)
    yrgb = None
    if torgb is not None:
        yrgb = np.empty((B, 3, Ho, Wo), dtype=np.float32)
        for name in ("w", "b", "sn", "smax", "yprev"):
            a, p = _opt(torgb.get(name))
            keep.append(a)
            if p is not None:
                setattr(d, "trgb_" + name, p)
        d.trgb_yout = _fp(yrgb)
    lib.glass_op_conv.argtypes = [C.c_int32, C.POINTER(ConvDesc)]
    _check(lib, lib.glass_op_conv(device, C.byref(d)))
    if both:                  # fused toRGB forms that store the feature map too (impl 2 / 5): (skip image, feature map)
        return yrgb, y
    return yrgb if yrgb is not None else y


def gemm(a, w, bias=None, mode=3, impl=0, acc=None, device=0):
    lib = load_library()
    a = _f32(a); w = _f32(w)
    M, K = a.shape
    N = w.shape[0]
    out = _f32(acc).copy() if acc is not None else np.empty((M, N), dtype=np.float32)
    b, bp = _opt(bias)
    lib.glass_op_gemm.argtypes = [C.c_int32] * 4 + [C.POINTER(C.c_float)] * 3 + [C.c_int32, C.c_int32, C.POINTER(C.c_float)]
    _check(lib, lib.glass_op_gemm(device, M, N, K, _fp(a), _fp(w), bp, mode, impl, _fp(out)))
    return out


def dense(x, wt, bias=None, in_sq=False, mode=0, eps_row=None, device=0):
    lib = load_library()
    x = _f32(x); wt = _f32(wt)
    P, K = x.shape
    N = wt.shape[1]
    out = np.empty((P, N), dtype=np.float32)
    b, bp = _opt(bias)
    e, ep = _opt(eps_row)
    fp = C.POINTER(C.c_float)
    lib.glass_op_dense.argtypes = [C.c_int32] * 4 + [fp, fp, fp, C.c_int32, C.c_int32, fp, fp]
    _check(lib, lib.glass_op_dense(device, P, K, N, _fp(x), _fp(wt), bp, int(in_sq), mode, ep, _fp(out)))
    return out


def torgb(x, wrgb, bias, sn, smax, yprev=None, device=0):
    lib = load_library()
    x = _f32(x)
    B, H, _, Cc = x.shape
    wrgb, bias, sn, smax = _f32(wrgb), _f32(bias), _f32(sn), _f32(smax)
    yp, ypp = _opt(yprev)
    out = np.empty((B, 3, H, H), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib.glass_op_torgb.argtypes = [C.c_int32] * 4 + [fp] * 7
    _check(lib, lib.glass_op_torgb(device, B, H, Cc, _fp(x), _fp(wrgb), _fp(bias), _fp(sn), _fp(smax), ypp, _fp(out)))
    return out


def blur(x, mode, device=0):
    lib = load_library()
    x = _f32(x)
    B, H, _, Cc = x.shape
    Ho = H + 1 if mode != 1 else H // 2      # mode 2: pad 2 written in 32-channel planes (un-permuted here)
    out = np.empty((B, Ho, Ho, Cc), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib.glass_op_blur.argtypes = [C.c_int32] * 5 + [fp, fp]
    _check(lib, lib.glass_op_blur(device, mode, B, H, Cc, _fp(x), _fp(out)))
    return from_planar32(out, B, Ho, Ho, Cc) if mode == 2 else out


def dblock_down(h, x, w1, wskip, b1, device=0):
    """Fused second half of a D block (conv_down.hip).  h, x [B,R,R,Cin] NHWC; w1 [Cout,Cin,3,3]; wskip [Cout,Cin,1,1]."""
    lib = load_library()
    h, x, w1, wskip, b1 = _f32(h), _f32(x), _f32(w1), _f32(wskip), _f32(b1)
    B, R, _, Cin = h.shape
    Cout = w1.shape[0]
    out = np.empty((B, R // 2, R // 2, Cout), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib.glass_op_dblock_down.argtypes = [C.c_int32] * 5 + [fp] * 6
    _check(lib, lib.glass_op_dblock_down(device, B, R, Cin, Cout, _fp(h), _fp(x), _fp(w1), _fp(wskip), _fp(b1), _fp(out)))
    return out


def dblock0(y, frgb_w, frgb_b, w0, b0, w1, wskip, b1, impl=0, device=0):
    """The discriminator's whole full-resolution block (conv_d0.hip): y [B,3,R,R] -> [B,R/2,R/2,64].  impl 1: the two-kernel form;
    2: the fused kernel writing chunk-planar (un-permuted here)."""
    lib = load_library()
    y, frgb_w, frgb_b, w0, b0, w1, wskip, b1 = (_f32(a) for a in (y, frgb_w, frgb_b, w0, b0, w1, wskip, b1))
    B, _, R, _ = y.shape
    out = np.empty((B, R // 2, R // 2, 64), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib.glass_op_dblock0.argtypes = [C.c_int32] * 4 + [fp] * 9
    _check(lib, lib.glass_op_dblock0(device, B, R, impl, _fp(y), _fp(frgb_w), _fp(frgb_b), _fp(w0), _fp(b0), _fp(w1), _fp(wskip),
                                     _fp(b1), _fp(out)))
    return from_planar8(out, B, R // 2, R // 2, 64) if impl == 2 else out


def fromrgb(y, w, bias, device=0):
    lib = load_library()
    y, w, bias = _f32(y), _f32(w), _f32(bias)
    B, _, R, _ = y.shape
    Cout = w.shape[0]
    out = np.empty((B, R, R, Cout), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib.glass_op_fromrgb.argtypes = [C.c_int32] * 4 + [fp] * 4
    _check(lib, lib.glass_op_fromrgb(device, B, R, Cout, _fp(y), _fp(w), _fp(bias), _fp(out)))
    return out


def mbstd(x, Cpad, batch_size, group=4, device=0):
    lib = load_library()
    x = _f32(x)
    B, hw, Cc = x.shape
    out = np.empty((B, hw, Cpad), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib.glass_op_mbstd.argtypes = [C.c_int32] * 7 + [fp, fp]
    _check(lib, lib.glass_op_mbstd(device, B, hw, Cc, Cpad, batch_size, group, _fp(x), _fp(out)))
    return out


def resize(y, S, ps, device=0):
    lib = load_library()
    y = _f32(y)
    B, _, R, _ = y.shape
    G = S // ps
    out = np.empty((B * G * G, 3 * ps * ps), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib.glass_op_resize.argtypes = [C.c_int32] * 5 + [fp, fp]
    _check(lib, lib.glass_op_resize(device, B, R, S, ps, _fp(y), _fp(out)))
    return out


def layernorm(x, g, b, device=0):
    lib = load_library()
    x, g, b = _f32(x), _f32(g), _f32(b)
    out = np.empty_like(x)
    fp = C.POINTER(C.c_float)
    lib.glass_op_layernorm.argtypes = [C.c_int32] * 3 + [fp] * 4
    _check(lib, lib.glass_op_layernorm(device, x.shape[0], x.shape[1], _fp(x), _fp(g), _fp(b), _fp(out)))
    return out


def attention(qkv, n_img, L, heads, causal=False, device=0):
    lib = load_library()
    qkv = _f32(qkv)
    out = np.empty((n_img * L, heads * 64), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib.glass_op_attention.argtypes = [C.c_int32] * 5 + [fp, fp]
    _check(lib, lib.glass_op_attention(device, n_img, L, heads, int(causal), _fp(qkv), _fp(out)))
    return out


def noise(n_mb, hw, layer, mb0, generation, seed, device=0):
    lib = load_library()
    out = np.empty((n_mb, hw), dtype=np.float32)
    lib.glass_op_noise.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                   C.POINTER(C.c_float)]
    _check(lib, lib.glass_op_noise(device, n_mb, hw, layer, mb0, generation, seed, _fp(out)))
    return out


def mfma_probe(a, b, device=0):
    lib = load_library()
    a, b = _f32(a), _f32(b)
    d = np.empty((32, 32), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib.glass_op_mfma_probe.argtypes = [C.c_int32, fp, fp, fp]
    _check(lib, lib.glass_op_mfma_probe(device, _fp(a), _fp(b), _fp(d)))
    return d


def host_pack_conv(w, up=False):
    """finalize()'s weight repacking, host only: returns [KS*KS][Neff][Cin] float32 (fp16-rounded)."""
    lib = load_library()
    w = _f32(w)
    Cout, Cin, KS, _ = w.shape
    out = np.empty((KS * KS, (4 if up else 1) * Cout, Cin), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib.glass_host_pack_conv.argtypes = [fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp]
    _check(lib, lib.glass_host_pack_conv(_fp(w), Cout, Cin, KS, int(up), _fp(out)))
    return out
