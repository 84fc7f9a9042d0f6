"""CPU check of the round-3 up-conv tile geometry (csrc/upfir.hip: upfir2_kernel): the index-level emulation in emu_ops.upfir2
(virtual image grid, rolling strips, per-image table look-ups) against the oracle's modulated up-convolution.  A wrong index
term in the kernel's design shows up here, without a GPU."""
import numpy as np
import pytest
import torch

import emu_ops
from clip_glass_amd import synth
from oracle import stylegan2_ref as sg
from util import check, nchw, nhwc, style_tables


def rnd(seed, name, shape, std=1.0):
    return synth.normal(seed, name, shape, std)


def h16(a):
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("B,H,W_,Cin,Cout,S,per_sample", [
    (5, 16, 16, 32, 32, None, False),     # 5 x 1 grid of 16 x 16 images: tiles span three images
    (12, 8, 16, 32, 32, 2, False),        # 8 x 2 grid, two-step segments crossing the vertical image boundary
    (3, 24, 40, 32, 64, 3, False),        # odd sizes, two n tiles
    (2, 32, 32, 32, 32, 2, True),         # per-sample weights: one image per grid
    (9, 16, 16, 32, 32, 1, False),        # 8 x 2 grid with seven empty slots, one step per segment
    (7, 8, 8, 32, 32, 2, False),          # 8 x 8 images (the r16 layer): 3 x 3 grid with two empty slots, one tile across the whole row
    (26, 8, 8, 32, 32, 1, False),         # two grids of 3 x 8 (+ 2)
])
def test_upfir2_geometry_emulation(B, H, W_, Cin, Cout, S, per_sample):
    L, bs = 16, 1
    x = rnd(11, "x", (B, Cin, H, W_)); w = rnd(11, "w", (Cout, Cin, 3, 3))
    lat = rnd(11, "lat", (B, L)); A = rnd(11, "A", (Cin, L)); Ab = rnd(11, "Ab", (Cin,), 0.2) + 1
    bias = rnd(11, "b", (Cout,), 0.3); strength = 0.37
    noise = rnd(11, "noise", (B // bs, 2 * H, 2 * W_))
    ps = np.abs(rnd(11, "ps", (B, Cout))) + 0.5
    ref = sg._mod_conv(torch.tensor(h16(x)), torch.tensor(lat), torch.tensor(w), torch.tensor(A), torch.tensor(Ab), demod=True, up=True)
    ref = ref + strength * torch.tensor(noise).repeat_interleave(bs, dim=0)[:, None]
    ref = sg._bias_act(ref, torch.tensor(bias)).numpy() * ps[:, :, None, None]
    sn, smax, dscale = style_tables(lat, A, Ab, w, demod=True)
    geo = emu_ops.upfir2_geometry(B, H, W_, Cout, per_sample_weights=per_sample, S=S)
    got = emu_ops.upfir2(nhwc(x), w, sn=sn, dscale=dscale, noise=noise, noise_strength=strength, batch_size=bs, bias=bias,
                         act=True, post_scale=ps, geo=geo)
    check("upfir2 emulation B%d %dx%d %d->%d S%s" % (B, H, W_, Cin, Cout, S), nchw(got), ref, 6e-3)
