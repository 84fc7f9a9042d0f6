"""Per-kernel parity: each HIP kernel family (through the diagnostic C ABI,
include/glass_ops.h) against the oracle's corresponding torch-CPU op."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from clip_glass_amd import synth
from oracle import stylegan2_ref as sg
from util import check, diag, nchw, nhwc, style_tables

pytestmark = pytest.mark.gpu
ops = None


@pytest.fixture(scope="module", autouse=True)
def _lib():
    global ops
    import os
    if os.environ.get("GLASS_EMULATE"):      # CPU dry-run of the tests + kernel math (tests/emu_ops.py)
        import emu_ops as _ops
    else:
        from clip_glass_amd import ops as _ops
    ops = _ops
    yield


def rnd(seed, name, shape, std=1.0):
    return synth.normal(seed, name, shape, std)


def h16(a):
    """Round to fp16 like the device does for activations / weights."""
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


def test_mfma_fragment_layout():
    a = rnd(1, "a", (32, 16)); b = rnd(1, "b", (16, 32))
    d = ops.mfma_probe(a, b)
    check("mfma32x32x16 layout", d, h16(a).astype(np.float64) @ h16(b).astype(np.float64), 1e-3)


@pytest.mark.parametrize("impl,B,H,Cin,Cout", [
    (1, 3, 12, 32, 48), (1, 2, 5, 16, 16), (1, 1, 33, 64, 160),
    (2, 2, 32, 64, 128),    # conv_tiled<3,1,8,128>
    (2, 1, 64, 32, 256),    # conv_tiled<3,1,8,128>, 2 n-tiles, 16 pixel tiles
    (2, 2, 32, 64, 64),     # conv_tiled<3,1,8,64>
    (2, 3, 32, 32, 32),     # conv_tiled<3,1,16,32>
    (2, 1, 64, 96, 96),     # conv_tiled<3,1,16,32>, 3 n-tiles, 3 chunks
])
def test_conv_plain_bias_act(impl, B, H, Cin, Cout):
    x = rnd(2, "x", (B, Cin, H, H)); w = rnd(2, "w", (Cout, Cin, 3, 3)); bias = rnd(2, "b", (Cout,), 0.3)
    ref = sg._bias_act(sg._conv(torch.tensor(h16(x)), torch.tensor(w), padding=1), torch.tensor(bias)).numpy()
    got = ops.conv(nhwc(x), w, bias=bias, act=True, impl=impl)
    check("conv3x3 B%d H%d %d->%d impl%d" % (B, H, Cin, Cout, impl), nchw(got), ref, 4e-3)


def _modconv_case(up, impl, B=4, H=8, Cin=32, Cout=48, L=24, batch_size=2, broadcast=False):
    if impl == 2:
        H, Cout = 32, (32 if up else 64)
    if isinstance(impl, tuple):           # (3, H, Cin, Cout): fused up-conv shapes
        impl, H, Cin, Cout = impl
    x = rnd(3, "x", (1 if broadcast else B, Cin, H, H)); w = rnd(3, "w", (Cout, Cin, 3, 3))
    lat = rnd(3, "lat", (B, L)); A = rnd(3, "A", (Cin, L)); Ab = rnd(3, "Ab", (Cin,), 0.2) + 1
    bias = rnd(3, "b", (Cout,), 0.3); strength = 0.37
    Ho = 2 * H if up else H
    noise = rnd(3, "noise", (B // batch_size, Ho, Ho))
    xt = torch.tensor(h16(x)).expand(B, Cin, H, H)
    ref = sg._mod_conv(xt, torch.tensor(lat), torch.tensor(w), torch.tensor(A), torch.tensor(Ab), demod=True, up=up)
    ref = ref + strength * torch.tensor(noise).repeat_interleave(batch_size, dim=0)[:, None]
    ref = sg._bias_act(ref, torch.tensor(bias)).numpy()
    sn, smax, dscale = style_tables(lat, A, Ab, w, demod=True)
    got = ops.conv(nhwc(x), w, up=up, sn=sn, dscale=dscale, noise=noise, noise_strength=strength,
                   batch_size=batch_size, bias=bias, act=True, impl=impl, broadcast_x=broadcast, B=B)
    return got, ref


def test_conv_stream_matches_tiled_and_direct():
    """conv_stream.hip (persistent, 3 tiles in flight) on a shape with uneven tile ranges per workgroup, image borders
    in every direction, per-sample style + demod + noise + bias + lrelu: same numbers as the tiled / direct kernels."""
    rng = np.random.default_rng(5)
    B, H, W, C = 5, 256, 1024 + 32, 32       # 5 * 32 * 33 = 5280 tiles over 512 workgroups: 11/10-tile ranges, sample switches
    x = rng.standard_normal((B, H, W, C)).astype(np.float16).astype(np.float32)
    w = (rng.standard_normal((C, C, 3, 3)) / math.sqrt(9 * C)).astype(np.float32)
    sn = rng.uniform(0.5, 1.0, (B, C)).astype(np.float32)
    ds = rng.uniform(0.5, 2.0, (B, C)).astype(np.float32)
    noise = rng.standard_normal((B, H, W)).astype(np.float32)
    bias = rng.standard_normal(C).astype(np.float32) * 0.2
    res = rng.standard_normal((B, H, W, C)).astype(np.float16).astype(np.float32)
    kw = dict(sn=sn, dscale=ds, noise=noise, noise_strength=0.3, batch_size=1, bias=bias, act=True, out_scale=0.7)
    got = ops.conv(x, w, impl=4, **kw)
    ref_t = ops.conv(x, w, impl=2, **kw)
    ref_d = ops.conv(x, w, impl=1, **kw)
    # same staging and MFMA order as the tiled kernel; the streaming epilogue rounds the activation INPUT to fp16 and applies
    # lrelu * sqrt2 * scale in packed fp16 (the tiled kernel rounds once, after the activation): outputs differ by an fp16 ulp or two
    check("conv_stream vs direct", got, ref_d, 4e-3)
    d = np.abs(got - ref_t)
    diag("[conv_stream] vs tiled: max |diff| %.3e, %.1f %% of outputs differ" % (d.max(), 100 * (got != ref_t).mean()))
    assert d.max() <= 2.0 ** -7 * max(1.0, float(np.abs(ref_t).max())), float(d.max())


@pytest.mark.parametrize("with_skip", [True, False])
def test_conv_stream_fused_torgb(with_skip):
    """conv_stream<torgb>: the generator's last conv with toRGB (stylegan2/models.py:852-870) + the FIR-upsampled skip image
    (models.py:1004-1013, modules.py:580-602) applied to the tile in the accumulators — against the same conv's stored fp16
    output pushed through a float64 toRGB."""
    rng = np.random.default_rng(11)
    B, H, W, C = 5, 256, 1024 + 32, 32       # uneven tile ranges, sample switches inside a workgroup's range, all four borders
    x = rng.standard_normal((B, H, W, C)).astype(np.float16).astype(np.float32)
    w = (rng.standard_normal((C, C, 3, 3)) / math.sqrt(9 * C)).astype(np.float32)
    sn = rng.uniform(0.5, 1.0, (B, C)).astype(np.float32)
    ds = rng.uniform(0.5, 2.0, (B, C)).astype(np.float32)
    noise = rng.standard_normal((B, H, W)).astype(np.float32)
    bias = rng.standard_normal(C).astype(np.float32) * 0.2
    kw = dict(sn=sn, dscale=ds, noise=noise, noise_strength=0.3, batch_size=1, bias=bias, act=True)
    wrgb = (rng.standard_normal((3, C)) / math.sqrt(C)).astype(np.float32)
    brgb = rng.standard_normal(3).astype(np.float32) * 0.1
    srgb = rng.uniform(0.2, 1.0, (B, C)).astype(np.float32)
    smax = rng.uniform(0.5, 3.0, B).astype(np.float32)
    yprev = rng.standard_normal((B, 3, H // 2, W // 2)).astype(np.float32) if with_skip else None
    got = ops.conv(x, w, impl=4, torgb=dict(w=wrgb, b=brgb, sn=srgb, smax=smax, yprev=yprev), **kw)
    feat = ops.conv(x, w, impl=4, **kw)                 # [B,H,W,C], fp16-rounded as the fused kernel sees it
    ref = _torgb_ref(feat, wrgb, brgb, srgb, smax, yprev)
    check("conv_stream<torgb>", got, ref, 2e-5)


def _stride2_skip_case(B, R, Cin, Cout, impl, label):
    rng = np.random.default_rng(17)
    hb = rng.standard_normal((B, R + 1, R + 1, Cin)).astype(np.float16).astype(np.float32)       # blurred (pad 2) conv0 output
    xs = rng.standard_normal((B, R // 2, R // 2, Cin)).astype(np.float16).astype(np.float32)     # down-sampled block input
    w1 = rng.standard_normal((Cout, Cin, 3, 3)).astype(np.float32)     # un-scaled: the runtime coefficient is applied at packing
    ws = rng.standard_normal((Cout, Cin, 1, 1)).astype(np.float32)
    b1 = rng.standard_normal(Cout).astype(np.float32) * 0.2
    kw = dict(stride=2, pad=0, bias=b1, act=True, out_scale=2.0 ** -0.5)
    got = ops.conv(hb, w1, skip=(xs, ws), impl=impl, **kw)
    s = ops.conv(xs, ws, pad=0, impl=2)
    ref2 = ops.conv(hb, w1, res=s, impl=2, **kw)
    t = lambda a: torch.tensor(np.ascontiguousarray(a.transpose(0, 3, 1, 2)), dtype=torch.float64)
    w1s = torch.tensor(w1 / math.sqrt(9 * Cin)).half().double(); wss = torch.tensor(ws / math.sqrt(Cin)).half().double()
    y = F.leaky_relu(F.conv2d(t(hb), w1s, torch.tensor(b1, dtype=torch.float64), stride=2), 0.2) * math.sqrt(2)
    ref = ((y + F.conv2d(t(xs), wss)) * 2.0 ** -0.5).numpy().transpose(0, 2, 3, 1)
    check(label + " vs float64", got, ref, 2e-3)
    check(label + " vs two passes", got, ref2, 3e-3)
    if impl == 5:     # conv_s2: the blurred input in 32-channel planes (common.h x_planar32, written so by the pad-2 blur): same values
        np.testing.assert_array_equal(got, ops.conv(hb, w1, skip=(xs, ws), impl=impl, planar32_x=True, **kw))


@pytest.mark.parametrize("B,R,Cin,Cout", [(2, 64, 64, 128), (3, 64, 32, 64), (1, 128, 128, 256)])
def test_conv_stride2_fused_skip(B, R, Cin, Cout):
    """conv_tiled<3,2,4,N,skip>: the D block's skip branch (modules.py:1238-1254) as extra K stages after the in-register
    activation, against float64 and the two-pass form (1x1 conv, then the stride-2 conv with it as the residual)."""
    _stride2_skip_case(B, R, Cin, Cout, 2, "stride-2 conv + fused skip")


@pytest.mark.parametrize("B,R,Cin,Cout", [(2, 64, 64, 128),       # 8 work items: one per workgroup, 2 chunks
                                          (1, 64, 32, 256),       # one chunk: prologue -> tail with nothing in between
                                          (1, 128, 128, 256),     # 32 items, 4 chunks
                                          (37, 64, 96, 384),      # 148 pixel tiles (not a multiple of 8) x 3 n tiles > 256 workgroups: the ring
                                                                  # runs on across items, uneven item counts per workgroup, 3 chunks
                                          (3, 256, 64, 128)])     # 4 x 16 tiles per image
def test_conv_s2_dma_ring(B, R, Cin, Cout):
    """conv_s2.hip (LDS-DMA ring: row-parity halves, de-interleaved columns, skip branch as the fourth stage of every chunk) against
    float64 and the two-pass form."""
    _stride2_skip_case(B, R, Cin, Cout, 5, "conv_s2")


@pytest.mark.parametrize("case", ["conv8", "up8", "down17", "skip1x1", "const4", "plain8", "plain4"])
def test_conv_gemm_matches_direct(case):
    """conv_gemm.hip (im2col + gemm_tiled + finishing pass) for the low-resolution layers against conv_direct: modulated 3x3,
    folded up-conv with depth-to-space, stride-2 with residual (gathered patch matrix), 1x1 (one launch), broadcast (learned const) input,
    un-modulated 3x3 (gathered)."""
    rng = np.random.default_rng(23)
    B, C = 6, 128
    kw = {}
    if case == "conv8":
        x = rng.standard_normal((B, 8, 8, C)); w = rng.standard_normal((C, C, 3, 3))
        kw = dict(sn=rng.uniform(0.3, 1.0, (B, C)), dscale=rng.uniform(0.5, 2.0, (B, C)), noise=rng.standard_normal((B // 2, 8, 8)),
                  noise_strength=0.4, batch_size=2, bias=rng.standard_normal(C) * 0.2, act=True)
    elif case == "up8":
        x = rng.standard_normal((B, 8, 8, C)); w = rng.standard_normal((64, C, 3, 3))
        kw = dict(up=True, sn=rng.uniform(0.3, 1.0, (B, C)), dscale=rng.uniform(0.5, 2.0, (B, 64)), noise=rng.standard_normal((B, 16, 16)),
                  noise_strength=0.4, batch_size=1, bias=rng.standard_normal(64) * 0.2, act=True)
    elif case == "down17":
        x = rng.standard_normal((B, 17, 17, C)); w = rng.standard_normal((2 * C, C, 3, 3))
        kw = dict(stride=2, pad=0, bias=rng.standard_normal(2 * C) * 0.2, act=True, res=rng.standard_normal((B, 8, 8, 2 * C)), out_scale=2.0 ** -0.5)
    elif case == "skip1x1":
        x = rng.standard_normal((B, 8, 8, C)); w = rng.standard_normal((2 * C, C, 1, 1))
        kw = dict(pad=0)
    elif case in ("plain8", "plain4"):      # no activation-side style (the D blocks' first convs): the GEMM gathers its patch matrix itself (round 6);
        R = 8 if case == "plain8" else 4    # zero padding on every border, split-K slices at 4 x 4
        x = rng.standard_normal((B, R, R, C)); w = rng.standard_normal((C, C, 3, 3))
        kw = dict(bias=rng.standard_normal(C) * 0.2, act=True)
    else:
        x = rng.standard_normal((1, 4, 4, C)); w = rng.standard_normal((C, C, 3, 3))
        kw = dict(broadcast_x=True, B=16, sn=rng.uniform(0.3, 1.0, (16, C)), dscale=rng.uniform(0.5, 2.0, (16, C)), bias=rng.standard_normal(C) * 0.2, act=True)
    kw = {k: (np.asarray(v, dtype=np.float32) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
    if "res" in kw:
        kw["res"] = kw["res"].astype(np.float16).astype(np.float32)
    x = x.astype(np.float16).astype(np.float32); w = w.astype(np.float32)
    got = ops.conv(x, w, impl=6, **kw)
    ref = ops.conv(x, w, impl=1, **kw)
    check("conv_gemm %s vs direct" % case, got, ref, 2e-3)


def test_conv_tiled_blurdown_byproduct():
    """conv_tiled<3,1,8,64,xs>: the D block's skip-branch input (FIR pad 1 + ::2 of the block input, modules.py:1587-1601) taken
    from the patch the first conv stages anyway — against the separate blur-down pass; the conv output is unchanged."""
    rng = np.random.default_rng(31)
    B, H, W, C = 3, 64, 96, 64
    x = rng.standard_normal((B, H, W, C)).astype(np.float16).astype(np.float32)
    w = (rng.standard_normal((C, C, 3, 3)) / math.sqrt(9 * C)).astype(np.float32)
    bias = (rng.standard_normal(C) * 0.2).astype(np.float32)
    xs = np.empty((B, H // 2, W // 2, C), dtype=np.float32)
    y = ops.conv(x, w, impl=2, bias=bias, act=True, xs_out=xs)
    np.testing.assert_array_equal(y, ops.conv(x, w, impl=2, bias=bias, act=True))
    f = np.array([1, 3, 3, 1], dtype=np.float64) / 8
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
    ref = sum(f[a] * f[b2] * xp[:, a:a + H:2, b2:b2 + W:2] for a in range(4) for b2 in range(4))
    check("blur-down by-product", xs, ref, 2e-3)


def _torgb_ref(feat, wrgb, brgb, srgb, smax, yprev):
    """float64 toRGB (stylegan2/models.py:852-870) + FIR-upsampled skip image (modules.py:580-602) of an NHWC map."""
    B, H, W, _ = feat.shape
    wm = wrgb[None].astype(np.float64) * (srgb.astype(np.float64) * smax[:, None])[:, None, :]    # [B,3,C]
    ref = np.einsum("bhwc,boc->bohw", feat.astype(np.float64), wm) + brgb[None, :, None, None]
    if yprev is not None:
        yp = np.pad(yprev.astype(np.float64), ((0, 0), (0, 0), (1, 0), (1, 0)))       # x[m-1] with zero at m = 0
        a, bq = yp[:, :, :-1], yp[:, :, 1:]                                          # rows m-1, m
        rows = np.empty((B, 3, H, W // 2 + 1))
        rows[:, :, 0::2] = 0.75 * a + 0.25 * bq
        rows[:, :, 1::2] = 0.25 * a + 0.75 * bq
        a, bq = rows[..., :-1], rows[..., 1:]
        up = np.empty((B, 3, H, W))
        up[..., 0::2] = 0.75 * a + 0.25 * bq
        up[..., 1::2] = 0.25 * a + 0.75 * bq
        ref = ref + up
    return ref


@pytest.mark.parametrize("impl,C,with_skip", [(2, 64, True), (2, 64, False), (5, 128, True), (5, 128, False)])
def test_conv_epilogue_fused_torgb(impl, C, with_skip):
    """conv_tiled<3,1,8,64,torgb> / conv_glds<torgb>: a mid-resolution block's last conv with toRGB + skip sum applied to the
    tile in registers — against the same kernel's stored fp16 map pushed through a float64 toRGB."""
    rng = np.random.default_rng(13)
    B, H, W = 3, 32, 96
    x = rng.standard_normal((B, H, W, C)).astype(np.float16).astype(np.float32)
    w = (rng.standard_normal((C, C, 3, 3)) / math.sqrt(9 * C)).astype(np.float32)
    sn = rng.uniform(0.5, 1.0, (B, C)).astype(np.float32)
    ds = rng.uniform(0.5, 2.0, (B, C)).astype(np.float32)
    noise = rng.standard_normal((B, H, W)).astype(np.float32)
    bias = rng.standard_normal(C).astype(np.float32) * 0.2
    kw = dict(sn=sn, dscale=ds, noise=noise, noise_strength=0.3, batch_size=1, bias=bias, act=True)
    wrgb = (rng.standard_normal((3, C)) / math.sqrt(C)).astype(np.float32)
    brgb = rng.standard_normal(3).astype(np.float32) * 0.1
    srgb = rng.uniform(0.2, 1.0, (B, C)).astype(np.float32)
    smax = rng.uniform(0.5, 3.0, B).astype(np.float32)
    yprev = rng.standard_normal((B, 3, H // 2, W // 2)).astype(np.float32) if with_skip else None
    got = ops.conv(x, w, impl=impl, torgb=dict(w=wrgb, b=brgb, sn=srgb, smax=smax, yprev=yprev), **kw)
    feat = ops.conv(x, w, impl=impl, **kw)
    check("fused toRGB impl %d" % impl, got, _torgb_ref(feat, wrgb, brgb, srgb, smax, yprev), 2e-5)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 32, 64, 128, 128), (3, 16, 32, 256, 256), (1, 64, 64, 160, 128), (5, 16, 16, 256, 256),
                                             (2, 32, 16, 128, 384)])
def test_conv_glds_matches_tiled(B, H, W, Cin, Cout):
    """conv_glds.hip (LDS-DMA staged, swizzled dense LDS images, 3-slot weight ring): same arithmetic as the tiled kernel
    (same MFMA order per accumulator), image borders through the zero page, full epilogue."""
    rng = np.random.default_rng(7)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float16).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / math.sqrt(9 * Cin)).astype(np.float32)
    ds = rng.uniform(0.5, 2.0, (B, Cout)).astype(np.float32)
    noise = rng.standard_normal((B, H, W)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32) * 0.2
    res = rng.standard_normal((B, H, W, Cout)).astype(np.float16).astype(np.float32)
    kw = dict(dscale=ds, noise=noise, noise_strength=0.3, batch_size=1, bias=bias, act=True, res=res, out_scale=0.7)
    got = ops.conv(x, w, impl=5, **kw)
    ref_d = ops.conv(x, w, impl=1, **kw)
    ref_t = ops.conv(x, w, impl=2, **kw) if W % 32 == 0 else ref_d       # 16-px-wide images have no conv_tiled instance
    diag("[glds] B%d %dx%d %d->%d max|glds-tiled| %.3e  max|glds-direct| %.3e" % (B, H, W, Cin, Cout, np.abs(got - ref_t).max(),
                                                                                 np.abs(got - ref_d).max()))
    assert np.abs(got - ref_t).max() <= 2.0 ** -8 * max(1.0, float(np.abs(ref_t).max()))
    check("conv_glds vs direct", got, ref_d, 4e-3)
    # style-modulated input: the LDS-DMA kernel multiplies the weight fragments by s (W*s rounded instead of x*s)
    sn = rng.uniform(-1.0, 1.0, (B, Cin)).astype(np.float32)
    got_s = ops.conv(x, w, impl=5, sn=sn, **kw)
    ref_s = ops.conv(x, w, impl=1, sn=sn, **kw)
    check("conv_glds (style on weights) vs direct", got_s, ref_s, 4e-3)


def test_conv_glds_persistent_matches_tiled():
    """The persistent form of conv_glds (>= 2 work items per CU: the DMA ring runs on into the next item, epilogue through one
    patch buffer in 32-channel slices): 512 work items of 8 stages, sample switches, borders, style, residual — the tiled
    kernel's numbers; and the fused toRGB epilogue in that form."""
    rng = np.random.default_rng(29)
    B, H, W, Cin, Cout = 8, 128, 128, 128, 256
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float16).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / math.sqrt(9 * Cin)).astype(np.float32)
    ds = rng.uniform(0.5, 2.0, (B, Cout)).astype(np.float32)
    noise = rng.standard_normal((B, H, W)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32) * 0.2
    res = rng.standard_normal((B, H, W, Cout)).astype(np.float16).astype(np.float32)
    kw = dict(dscale=ds, noise=noise, noise_strength=0.3, batch_size=1, bias=bias, act=True, res=res, out_scale=0.7)
    got = ops.conv(x, w, impl=5, **kw)
    ref_t = ops.conv(x, w, impl=2, **kw)
    assert np.abs(got - ref_t).max() <= 2.0 ** -8 * max(1.0, float(np.abs(ref_t).max()))
    sn = rng.uniform(-1.0, 1.0, (B, Cin)).astype(np.float32)
    check("persistent conv_glds (style on weights) vs direct", ops.conv(x, w, impl=5, sn=sn, **kw), ops.conv(x, w, impl=1, sn=sn, **kw), 4e-3)
    # blur-down of the input as a by-product of the staged patches (two n tiles per pixel tile: chunk c is written by n tile c % 2)
    xs = np.full((B, H // 2, W // 2, Cin), np.nan, dtype=np.float32)
    kw2 = dict(bias=bias, act=True, out_scale=0.7)
    y2 = ops.conv(x, w, impl=5, xs_out=xs, **kw2)
    np.testing.assert_array_equal(y2, ops.conv(x, w, impl=5, **kw2))
    f = np.array([1, 3, 3, 1], dtype=np.float64) / 8
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
    check("persistent conv_glds blur-down by-product", xs,
          sum(f[a] * f[b2] * xp[:, a:a + H:2, b2:b2 + W:2] for a in range(4) for b2 in range(4)), 2e-3)
    # fused toRGB: one n tile, 16 x 8 x 4 = 512 items
    B, C = 16, 128
    x = rng.standard_normal((B, H, W, C)).astype(np.float16).astype(np.float32)
    w = (rng.standard_normal((C, C, 3, 3)) / math.sqrt(9 * C)).astype(np.float32)
    kw = dict(sn=rng.uniform(0.5, 1.0, (B, C)).astype(np.float32), dscale=rng.uniform(0.5, 2.0, (B, C)).astype(np.float32),
              noise=rng.standard_normal((B, H, W)).astype(np.float32), noise_strength=0.3, batch_size=1,
              bias=(rng.standard_normal(C) * 0.2).astype(np.float32), act=True)
    wrgb = (rng.standard_normal((3, C)) / math.sqrt(C)).astype(np.float32)
    brgb = (rng.standard_normal(3) * 0.1).astype(np.float32)
    srgb = rng.uniform(0.2, 1.0, (B, C)).astype(np.float32)
    smax = rng.uniform(0.5, 3.0, B).astype(np.float32)
    yprev = rng.standard_normal((B, 3, H // 2, W // 2)).astype(np.float32)
    got = ops.conv(x, w, impl=5, torgb=dict(w=wrgb, b=brgb, sn=srgb, smax=smax, yprev=yprev), **kw)
    feat = ops.conv(x, w, impl=5, **kw)
    check("persistent conv_glds fused toRGB", got, _torgb_ref(feat, wrgb, brgb, srgb, smax, yprev), 2e-5)


@pytest.mark.parametrize("B,H,W", [(3, 128, 256), (1, 256, 128), (5, 128, 128), (2, 512, 512)])
def test_conv_wreg_matches_tiled(B, H, W):
    """conv_wreg.hip (64 -> 64 channels: the weights in registers, one wave per SIMD, a three-tile LDS-DMA ring of patches): the tiled
    kernel's numbers with the full epilogue, image borders through the zero page, ranges that cross candidates and end unevenly; the
    blur-down by-product; the fused toRGB with and without the skip image; the chunk-planar input layout."""
    rng = np.random.default_rng(41)
    C = 64
    x = rng.standard_normal((B, H, W, C)).astype(np.float16).astype(np.float32)
    w = (rng.standard_normal((C, C, 3, 3)) / math.sqrt(9 * C)).astype(np.float32)
    ds = rng.uniform(0.5, 2.0, (B, C)).astype(np.float32)
    noise = rng.standard_normal((B, H, W)).astype(np.float32)
    bias = rng.standard_normal(C).astype(np.float32) * 0.2
    res = rng.standard_normal((B, H, W, C)).astype(np.float16).astype(np.float32)
    kw = dict(dscale=ds, noise=noise, noise_strength=0.3, batch_size=1, bias=bias, act=True, out_scale=0.7)
    got = ops.conv(x, w, impl=5, **kw)
    ref_t = ops.conv(x, w, impl=2, **kw)
    diag("[wreg] B%d %dx%d max|wreg-tiled| %.3e" % (B, H, W, np.abs(got - ref_t).max()))
    assert np.abs(got - ref_t).max() <= 2.0 ** -8 * max(1.0, float(np.abs(ref_t).max()))
    check("conv_wreg vs direct", got, ops.conv(x, w, impl=1, **kw), 4e-3)
    np.testing.assert_array_equal(got, ops.conv(x, w, impl=5, **kw))          # the ring is deterministic
    np.testing.assert_array_equal(got, ops.conv(x, w, impl=5, planar_x=True, **kw))      # chunk-planar input (common.h x_planar8): same values
    kwp = dict(bias=bias, act=True)                                          # the engine's form: no per-channel scale, no shift (the PLAIN epilogue)
    check("conv_wreg plain epilogue", ops.conv(x, w, impl=5, **kwp), ops.conv(x, w, impl=1, **kwp), 4e-3)
    with pytest.raises(Exception):                                            # a residual input is nobody's on this path: refused, not ignored
        ops.conv(x, w, impl=5, res=res, **kw)
    # blur-down of the input as a by-product of the staged patches
    xs = np.full((B, H // 2, W // 2, C), np.nan, dtype=np.float32)
    kw2 = dict(bias=bias, act=True, out_scale=0.7)
    y2 = ops.conv(x, w, impl=5, xs_out=xs, **kw2)
    np.testing.assert_array_equal(y2, ops.conv(x, w, impl=5, **kw2))
    xs_p = np.full_like(xs, np.nan)
    np.testing.assert_array_equal(y2, ops.conv(x, w, impl=5, xs_out=xs_p, planar_x=True, **kw2))
    np.testing.assert_array_equal(xs, xs_p)
    f = np.array([1, 3, 3, 1], dtype=np.float64) / 8
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
    check("conv_wreg blur-down by-product", xs, sum(f[a] * f[b2] * xp[:, a:a + H:2, b2:b2 + W:2] for a in range(4) for b2 in range(4)), 2e-3)
    xs_t = np.full_like(xs, np.nan)
    ops.conv(x, w, impl=2, xs_out=xs_t, **kw2)
    np.testing.assert_array_equal(xs, xs_t)                                   # the same packed-fp16 FIR as conv_tiled<xs>
    # fused toRGB
    kw3 = dict(dscale=ds, noise=noise, noise_strength=0.3, batch_size=1, bias=bias, act=True)
    wrgb = (rng.standard_normal((3, C)) / math.sqrt(C)).astype(np.float32)
    brgb = (rng.standard_normal(3) * 0.1).astype(np.float32)
    srgb = rng.uniform(0.2, 1.0, (B, C)).astype(np.float32)
    smax = rng.uniform(0.5, 3.0, B).astype(np.float32)
    feat = ops.conv(x, w, impl=5, **kw3)
    for yprev in (rng.standard_normal((B, 3, H // 2, W // 2)).astype(np.float32), None):
        got_rgb = ops.conv(x, w, impl=5, torgb=dict(w=wrgb, b=brgb, sn=srgb, smax=smax, yprev=yprev), **kw3)
        check("conv_wreg fused toRGB", got_rgb, _torgb_ref(feat, wrgb, brgb, srgb, smax, yprev), 2e-5)


@pytest.mark.parametrize("impl", [1, 2])
def test_conv_modulated_demod_noise(impl):
    got, ref = _modconv_case(False, impl)
    check("modconv impl%d" % impl, nchw(got), ref, 5e-3)


@pytest.mark.parametrize("impl", [1, 2, (3, 32, 32, 32), (3, 16, 64, 32), (3, 64, 32, 64), (3, 40, 32, 96), (3, 128, 64, 32), (3, 16, 128, 128)])
def test_conv_modulated_up(impl):
    got, ref = _modconv_case(True, impl)
    check("modconv-up impl%s" % (impl,), nchw(got), ref, 5e-3)


@pytest.mark.parametrize("B,H,Cin,Cout,bs", [
    (5, 16, 32, 32, 1),      # 5 x 1 virtual grid of 16 x 16 images (a tile spans three of them)
    (12, 16, 64, 64, 4),     # 8 x 2 grid, noise planes shared by minibatches of 4
    (9, 32, 32, 32, 3),      # 8 x 2 grid with seven empty slots
    (20, 16, 32, 96, 1),     # 8 x 3 grid, three n tiles
    (70, 16, 32, 32, 2),     # two grids (8 x 8 + 6)
    (3, 64, 96, 64, 1),      # three chunks, rolling segments inside one image row
    (5, 8, 32, 32, 1),       # 8 x 8 images (the r16 layer, round 6): three per virtual row, a tile spans all of them
    (26, 8, 64, 96, 2),      # two grids of 3 x 8 (+ 2), three n tiles
    (64, 8, 32, 32, 4),      # the population's geometry: 3 x 8 grids, the last one with 16 images
])
def test_conv_up_virtual_grid(B, H, Cin, Cout, bs):
    """upfir2_kernel (round 3): tiles and strips run across the candidates of a launch (csrc/upfir.hip)."""
    got, ref = _modconv_case(True, (3, H, Cin, Cout), B=B, batch_size=bs)
    check("modconv-up grid B%d H%d %d->%d" % (B, H, Cin, Cout), nchw(got), ref, 5e-3)


def test_conv_broadcast_const():
    got, ref = _modconv_case(False, 1, B=4, H=4, Cin=32, Cout=32, broadcast=True)
    check("modconv const-input", nchw(got), ref, 5e-3)


@pytest.mark.parametrize("impl,H,Cin,Cout", [(1, 16, 32, 48), (2, 64, 32, 64), (2, 64, 32, 128), (2, 64, 64, 64)])
def test_d_block_pieces(impl, H, Cin, Cout):
    """DiscriminatorConvBlock (modules.py:1587-1601): conv0, FIR+stride-2 conv1, FIR+1x1 skip, (h+s)/sqrt2."""
    B = 2
    x = rnd(4, "x", (B, Cin, H, H)); w0 = rnd(4, "w0", (Cin, Cin, 3, 3)); b0 = rnd(4, "b0", (Cin,), 0.3)
    w1 = rnd(4, "w1", (Cout, Cin, 3, 3)); b1 = rnd(4, "b1", (Cout,), 0.3); ws = rnd(4, "ws", (Cout, Cin, 1, 1))
    xt = torch.tensor(h16(x))
    h = sg._bias_act(sg._conv(xt, torch.tensor(w0), padding=1), torch.tensor(b0))
    hb = sg._filter(h, sg._fir(), 2, 2)
    h1 = sg._bias_act(sg._conv(hb, torch.tensor(w1), stride=2), torch.tensor(b1))
    xs = sg._filter(xt, sg._fir(), 1, 1)[:, :, ::2, ::2]
    s = sg._conv(xs, torch.tensor(ws))
    ref = ((h1 + s) / math.sqrt(2)).numpy()
    g_h = ops.conv(nhwc(x), w0, bias=b0, act=True, impl=impl)
    check("D conv0", nchw(g_h), h.numpy(), 4e-3)
    g_hb = ops.blur(g_h, 0)
    check("D blur pad2", nchw(g_hb), hb.numpy(), 4e-3)
    if g_h.shape[-1] % 32 == 0:
        np.testing.assert_array_equal(g_hb, ops.blur(g_h, 2))       # the same values written in 32-channel planes (for conv_s2)
    g_xs = ops.blur(nhwc(x), 1)
    check("D blur-down", nchw(g_xs), xs.numpy(), 4e-3)
    g_s = ops.conv(g_xs, ws, pad=0, impl=impl)
    check("D skip 1x1", nchw(g_s), s.numpy(), 5e-3)
    g_o = ops.conv(g_hb, w1, stride=2, pad=0, bias=b1, act=True, res=g_s, out_scale=1 / math.sqrt(2), impl=impl)
    check("D conv1 s2 + res", nchw(g_o), ref, 6e-3)


@pytest.mark.parametrize("B,R,Cin,Cout", [
    (2, 64, 32, 64),      # one tile column: both image borders in every window; one tile per workgroup (padded second step)
    (1, 128, 32, 64),     # two tile columns: edge columns inside the image on the left tile
    (3, 192, 32, 64),     # 3 x 3 x 24 tiles: interior tiles (no padding mask), odd counts
    (20, 256, 32, 64),    # 2560 tiles: five tiles per persistent workgroup (odd: padded step), windows refilled two tiles ahead
])
def test_d_block_down_fused(B, R, Cin, Cout):
    """conv_down.hip (+ launch_blur_down for the skip input): FIR pad 2 -> conv3x3 stride 2 + bias + lrelu*sqrt2, FIR pad 1 -> ::2
    -> conv1x1 skip, (a + b)/sqrt2 (modules.py:1204-1254, 1587-1601) vs the oracle's ops on the same fp16-rounded inputs."""
    h = rnd(14, "h", (B, Cin, R, R)); x = rnd(14, "x", (B, Cin, R, R))
    w1 = rnd(14, "w1", (Cout, Cin, 3, 3)); b1 = rnd(14, "b1", (Cout,), 0.3); ws = rnd(14, "ws", (Cout, Cin, 1, 1))
    ht, xt = torch.tensor(h16(h)), torch.tensor(h16(x))
    hb = sg._filter(ht, sg._fir(), 2, 2)
    h1 = sg._bias_act(sg._conv(hb, torch.tensor(w1), stride=2), torch.tensor(b1))
    xs = sg._filter(xt, sg._fir(), 1, 1)[:, :, ::2, ::2]
    ref = ((h1 + sg._conv(xs, torch.tensor(ws))) / math.sqrt(2)).numpy()
    got = ops.dblock_down(nhwc(h), nhwc(x), w1, ws, b1)
    check("D down fused B%d R%d %d->%d" % (B, R, Cin, Cout), nchw(got), ref, 6e-3)
    # image borders carry the zero padding of both FIRs: check them on their own (a wrong mask hides in a global norm)
    for name, sl in (("top", np.s_[:, :, :2, :]), ("bottom", np.s_[:, :, -2:, :]), ("left", np.s_[:, :, :, :2]), ("right", np.s_[:, :, :, -2:])):
        check("D down fused border " + name, nchw(got)[sl], ref[sl], 8e-3)


def _dblock0_case(B, R, seed=21):
    y = rnd(seed, "y", (B, 3, R, R), 0.8); fw = rnd(seed, "fw", (32, 3, 1, 1)); fb = rnd(seed, "fb", (32,), 0.3)
    w0 = rnd(seed, "w0", (32, 32, 3, 3)); b0 = rnd(seed, "b0", (32,), 0.3)
    w1 = rnd(seed, "w1", (64, 32, 3, 3)); b1 = rnd(seed, "b1", (64,), 0.3); ws = rnd(seed, "ws", (64, 32, 1, 1))
    img = ((torch.tensor(y) + 1) / 2).clip(0, 1) * 2 - 1                                   # utils.py:14-21
    x = sg._bias_act(sg._conv(img, torch.tensor(fw)), torch.tensor(fb))                    # fromRGB (models.py:1125-1143)
    h = sg._bias_act(sg._conv(x, torch.tensor(w0), padding=1), torch.tensor(b0))
    h1 = sg._bias_act(sg._conv(sg._filter(h, sg._fir(), 2, 2), torch.tensor(w1), stride=2), torch.tensor(b1))
    xs = sg._filter(x, sg._fir(), 1, 1)[:, :, ::2, ::2]
    ref = ((h1 + sg._conv(xs, torch.tensor(ws))) / math.sqrt(2)).numpy()
    return (y, fw.reshape(32, 3) / math.sqrt(3), fb, w0, b0, w1, ws, b1), ref


@pytest.mark.parametrize("B,R", [
    (1, 64),       # 16 steps for 256 workgroups: one step per workgroup, every step primed; second tile column holds 2 of 30 pixels
    (2, 68),       # R/2 = 34: ragged last tile column (4 of 30), 17 steps per column (R % 8 != 0)
    (3, 192),      # 3 x 4 x 24 = 288 steps: two-step ranges, priming mid-column; interior tiles without any padding mask
    (8, 256),      # 1280 steps: five per workgroup, ranges crossing column and sample boundaries
])
def test_dblock0_fused(B, R):
    """conv_d0.hip: the discriminator's whole full-resolution block in one kernel (skip image -> fromRGB -> conv3x3 -> FIR pad 2 ->
    conv3x3 stride 2, + the 1x1 skip branch of FIR pad 1 [::2] of the fromRGB map; stylegan2/models.py:1125-1143, modules.py:1204-1254,
    1587-1601) against the oracle's UNFUSED ops, and — where the two-kernel form applies — against conv_stream<fromrgb> + conv_down."""
    args, ref = _dblock0_case(B, R)
    got = ops.dblock0(*args)
    check("D block0 fused B%d R%d" % (B, R), nchw(got), ref, 6e-3)
    np.testing.assert_array_equal(got, ops.dblock0(*args, impl=2))     # the chunk-planar output (for conv_wreg): same values, other addresses
    for name, sl in (("top", np.s_[:, :, :2, :]), ("bottom", np.s_[:, :, -2:, :]), ("left", np.s_[:, :, :, :2]), ("right", np.s_[:, :, :, -2:])):
        check("D block0 fused border " + name, nchw(got)[sl], ref[sl], 8e-3)
    if R >= 192 and R % 64 == 0 and ops is not None and hasattr(ops, "load_library"):
        two = ops.dblock0(*args, impl=1)
        # same fromRGB / conv0 arithmetic bit for bit; the FIR runs horizontally first here and vertically first in conv_down (one fp16
        # rounding each way), so outputs agree to a couple of fp16 ulps
        d = np.abs(got - two)
        diag("[dblock0] fused vs conv_stream<fromrgb> + conv_down: max |diff| %.3e, %.1f %% of outputs differ" % (d.max(), 100 * (got != two).mean()))
        assert d.max() <= 2.0 ** -7 * max(1.0, float(np.abs(two).max())), float(d.max())
        check("D block0 two-kernel form B%d R%d" % (B, R), nchw(two), ref, 6e-3)


@pytest.mark.parametrize("impl,M,N,K", [(1, 150, 200, 96), (2, 150, 256, 192), (2, 400, 192, 64), (2, 128, 128, 128)])
def test_gemm_modes(impl, M, N, K):
    a = rnd(5, "a", (M, K)); w = rnd(5, "w", (N, K), K ** -0.5); bias = rnd(5, "b", (N,), 0.2)
    base = h16(a).astype(np.float64) @ h16(w).astype(np.float64).T + bias
    check("gemm f32", ops.gemm(a, w, bias, mode=3, impl=impl), base, 3e-3)
    check("gemm f16", ops.gemm(a, w, bias, mode=0, impl=impl), base, 4e-3)
    check("gemm quickgelu", ops.gemm(a, w, bias, mode=1, impl=impl), base / (1 + np.exp(-1.702 * base)), 4e-3)
    acc = rnd(5, "acc", (M, N))
    check("gemm residual", ops.gemm(a, w, bias, mode=2, impl=impl, acc=acc), acc + base, 3e-3)
    check("gemm lrelu", ops.gemm(a, w, bias, mode=4, impl=impl), np.where(base > 0, base, 0.2 * base) * math.sqrt(2), 3e-3)


def test_dense_modes():
    P, K, N = 37, 200, 150
    x = rnd(6, "x", (P, K)); wt = rnd(6, "wt", (K, N), K ** -0.5); bias = rnd(6, "b", (N,), 0.2)
    v = x.astype(np.float64) @ wt + bias
    check("dense", ops.dense(x, wt, bias), v, 1e-5)
    check("dense lrelu", ops.dense(x, wt, bias, mode=1), np.where(v > 0, v, 0.2 * v) * math.sqrt(2), 1e-5)
    eps = np.abs(rnd(6, "e", (P,))) + 0.1
    wpos = np.abs(wt)
    v2 = (x.astype(np.float64) ** 2) @ wpos
    check("dense sq+rsqrt", ops.dense(x, wpos, None, in_sq=True, mode=2, eps_row=eps), 1 / np.sqrt(v2 + eps[:, None]), 1e-5)


@pytest.mark.parametrize("H", [4, 16])
def test_torgb_skip(H):
    B, Cc, L = 3, 32, 24
    x = rnd(7, "x", (B, Cc, H, H)); w = rnd(7, "w", (3, Cc, 1, 1)); lat = rnd(7, "lat", (B, L))
    A = rnd(7, "A", (Cc, L)); Ab = rnd(7, "Ab", (Cc,), 0.2) + 1; bias = rnd(7, "b", (3,), 0.3)
    yprev = rnd(7, "yp", (B, 3, H // 2, H // 2))
    t = sg._mod_conv(torch.tensor(h16(x)), torch.tensor(lat), torch.tensor(w), torch.tensor(A), torch.tensor(Ab),
                     demod=False, up=False)
    ref = (sg._bias_act(t, torch.tensor(bias), act=False) + sg._upsample_skip(torch.tensor(yprev))).numpy()
    sn, smax, _ = style_tables(lat, A, Ab, w, demod=False)
    got = ops.torgb(nhwc(x), w.reshape(3, Cc) / math.sqrt(Cc), bias, sn, smax[:, 0], yprev)
    check("torgb+skip H%d" % H, got, ref, 2e-3)
    got0 = ops.torgb(nhwc(x), w.reshape(3, Cc) / math.sqrt(Cc), bias, sn, smax[:, 0], None)
    check("torgb first H%d" % H, got0, sg._bias_act(t, torch.tensor(bias), act=False).numpy(), 2e-3)


def test_fromrgb():
    B, R, Cout = 2, 16, 32
    y = rnd(8, "y", (B, 3, R, R), 0.8); w = rnd(8, "w", (Cout, 3, 1, 1)); bias = rnd(8, "b", (Cout,), 0.3)
    img = ((torch.tensor(y) + 1) / 2).clip(0, 1) * 2 - 1
    ref = sg._bias_act(sg._conv(img, torch.tensor(w)), torch.tensor(bias)).numpy()
    got = ops.fromrgb(y, w.reshape(Cout, 3) / math.sqrt(3), bias)
    check("fromrgb", nchw(got), ref, 2e-3)


@pytest.mark.parametrize("batch_size", [4, 8])
def test_mbstd(batch_size):
    B, Cc = 8, 32
    x = h16(rnd(9, "x", (B, Cc, 4, 4)))
    ref = torch.cat([sg.minibatch_std(torch.tensor(x[i:i + batch_size])) for i in range(0, B, batch_size)]).numpy()
    got = ops.mbstd(nhwc(x).reshape(B, 16, Cc), 48, batch_size)
    got = got.reshape(B, 4, 4, 48)
    check("mbstd bs%d features+std" % batch_size, nchw(got[..., :Cc + 1]), ref, 2e-3)
    assert np.all(got[..., Cc + 1:] == 0)


@pytest.mark.parametrize("R,S,ps", [(64, 32, 8), (32, 32, 8), (1024, 224, 32)])
def test_resize_patches(R, S, ps):
    B = 2
    y = rnd(10, "y", (B, 3, R, R), 0.8)
    img = ((torch.tensor(y) + 1) / 2).clip(0, 1)
    ref = F.interpolate(img, size=(S, S), mode="bilinear", align_corners=False)
    G = S // ps
    ref = ref.view(B, 3, G, ps, G, ps).permute(0, 2, 4, 1, 3, 5).reshape(B * G * G, 3 * ps * ps).numpy()
    check("resize %d->%d" % (R, S), ops.resize(y, S, ps), ref, 1e-3)


def test_layernorm_attention():
    M, D = 23, 128
    x = rnd(11, "x", (M, D), 2.0) + 0.5; g = rnd(11, "g", (D,), 0.1) + 1; b = rnd(11, "b", (D,), 0.1)
    ref = F.layer_norm(torch.tensor(x), (D,), torch.tensor(g), torch.tensor(b), 1e-5).numpy()
    check("layernorm", ops.layernorm(x, g, b), ref, 1e-5)
    for L, causal in ((50, False), (17, False), (77, True), (64, True), (33, True)):   # L <= 64: the MFMA kernel
        n_img, heads = 3, 2
        qkv = h16(rnd(12, "qkv%d" % L, (n_img * L, 3 * heads * 64)))
        t = torch.tensor(qkv).view(n_img, L, 3, heads, 64)
        q, k, v = (t[:, :, i].transpose(1, 2) for i in range(3))
        a = (q * 0.125) @ k.transpose(-1, -2)
        if causal:
            a = a + torch.full((L, L), float("-inf")).triu_(1)
        o = (torch.softmax(a, -1) @ v).transpose(1, 2).reshape(n_img * L, heads * 64)
        check("attention L%d causal%d" % (L, causal), ops.attention(qkv, n_img, L, heads, causal), o.numpy(), 2e-3)


def test_noise_matches_numpy_mirror():
    hw = 64 * 64
    got = ops.noise(3, hw, layer=5, mb0=2, generation=7, seed=0x1234567890)
    ref = np.stack([synth.noise_plane(0x1234567890, 7, 2 + m, 5, 64, 64).reshape(-1) for m in range(3)])
    check("philox noise vs numpy", got, ref, 0, atol=2e-5)
    assert abs(got.mean()) < 0.03 and abs(got.std() - 1) < 0.03
