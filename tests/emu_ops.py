"""numpy/torch emulation of the HIP kernels' arithmetic (TEST INFRASTRUCTURE).

Same signatures as clip_glass_amd/ops.py.  It mirrors what each kernel computes —
activation-side modulation, fp16-rounded operands, fp32 accumulation, the C++
weight repacking (taken from libglass.so's host-only glass_host_pack_conv), the
implicit-GEMM + depth-to-space formulation — so the parity tests (and the math of
the kernels) can be exercised on a machine without a GPU:  GLASS_EMULATE=1 pytest
tests/test_gpu_ops.py -m gpu.   Never imported by the product."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from clip_glass_amd import ops as real_ops
from clip_glass_amd import synth


def _h(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def mfma_probe(a, b):
    return (_h(a).astype(np.float64) @ _h(b).astype(np.float64)).astype(np.float32)


def conv(x, w, *, stride=1, pad=None, up=False, sn=None, dscale=None, noise=None, noise_strength=0.0,
         batch_size=1, bias=None, act=False, res=None, out_scale=1.0, impl=0, broadcast_x=False, B=None, device=0):
    x = np.asarray(x, np.float32)
    Bx, H, W, Cin = x.shape
    B = B or Bx
    Cout, _, KS, _ = w.shape
    pad = KS // 2 if pad is None else pad
    pk = real_ops.host_pack_conv(w, up)                      # [taps][Neff][Cin]
    Neff = pk.shape[1]
    xa = np.broadcast_to(_h(x), (B, H, W, Cin)).copy()
    if sn is not None:
        xa = _h(xa * np.asarray(sn, np.float32)[:, None, None, :])
    Hc = H if up else (H + 2 * pad - KS) // stride + 1
    xp = np.pad(xa, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    acc = np.zeros((B, Hc, Hc, Neff), np.float64)
    for ty in range(KS):
        for tx in range(KS):
            sl = xp[:, ty:ty + (Hc - 1) * stride + 1:stride, tx:tx + (Hc - 1) * stride + 1:stride, :]
            acc += sl.astype(np.float64) @ pk[ty * KS + tx].astype(np.float64).T
    if up:
        acc = acc.reshape(B, Hc, Hc, 2, 2, Cout).transpose(0, 1, 3, 2, 4, 5).reshape(B, 2 * Hc, 2 * Hc, Cout)
    v = acc
    if dscale is not None:
        v = v * np.asarray(dscale, np.float64)[:, None, None, :]
    if noise is not None:
        v = v + noise_strength * np.repeat(np.asarray(noise, np.float64), batch_size, axis=0)[..., None]
    if bias is not None:
        v = v + np.asarray(bias, np.float64)
    if act:
        v = np.where(v > 0, v, 0.2 * v) * math.sqrt(2)
    if res is not None:
        v = v + _h(res)
    return _h(v * out_scale)


def gemm(a, w, bias=None, mode=3, impl=0, acc=None, device=0):
    v = _h(a).astype(np.float64) @ _h(w).astype(np.float64).T
    if bias is not None:
        v = v + bias
    if mode == 0:
        return _h(v)
    if mode == 1:
        return _h(v / (1 + np.exp(-1.702 * v)))
    if mode == 2:
        return (acc + v).astype(np.float32)
    if mode == 3:
        return v.astype(np.float32)
    return (np.where(v > 0, v, 0.2 * v) * math.sqrt(2)).astype(np.float32)


def dense(x, wt, bias=None, in_sq=False, mode=0, eps_row=None, device=0):
    x = np.asarray(x, np.float64)
    v = (x * x if in_sq else x) @ np.asarray(wt, np.float64)
    if bias is not None:
        v = v + bias
    if mode == 1:
        v = np.where(v > 0, v, 0.2 * v) * math.sqrt(2)
    elif mode == 2:
        v = 1 / np.sqrt(v + np.asarray(eps_row)[:, None])
    return v.astype(np.float32)


def torgb(x, wrgb, bias, sn, smax, yprev=None, device=0):
    x = _h(x).astype(np.float64)
    wm = np.asarray(wrgb, np.float64)[None] * np.asarray(sn, np.float64)[:, None, :] * np.asarray(smax, np.float64)[:, None, None]
    y = np.einsum("bhwc,bkc->bkhw", x, wm) + np.asarray(bias)[None, :, None, None]
    if yprev is not None:
        yp = np.pad(np.asarray(yprev, np.float64), ((0, 0), (0, 0), (1, 0), (1, 0)))
        h = yprev.shape[2]
        a = np.array([[0.75, 0.25], [0.25, 0.75]])
        up = np.zeros_like(y)
        for py in range(2):
            for px in range(2):
                s = 0
                for dy in range(2):
                    for dx in range(2):
                        s = s + a[py][dy] * a[px][dx] * yp[:, :, dy:dy + h, dx:dx + h]
                up[:, :, py::2, px::2] = s
        y = y + up
    return y.astype(np.float32)


def blur(x, mode, device=0):
    t = torch.tensor(_h(x)).permute(0, 3, 1, 2)
    f = torch.tensor([1., 3., 3., 1.]) / 8
    k = (f[:, None] * f[None, :])[None, None].repeat(t.shape[1], 1, 1, 1)
    if mode == 0:
        y = F.conv2d(F.pad(t, [2, 2, 2, 2]), k, groups=t.shape[1])
    else:
        y = F.conv2d(F.pad(t, [1, 1, 1, 1]), k, groups=t.shape[1])[:, :, ::2, ::2]
    return _h(y.permute(0, 2, 3, 1).numpy())


def fromrgb(y, w, bias, device=0):
    img = np.clip((np.asarray(y, np.float32) + 1) * 0.5, 0, 1) * 2 - 1
    v = np.einsum("bchw,oc->bhwo", img.astype(np.float64), np.asarray(w, np.float64)) + bias
    return _h(np.where(v > 0, v, 0.2 * v) * math.sqrt(2))


def mbstd(x, Cpad, batch_size, group=4, device=0):
    x = _h(x).astype(np.float64)
    B, hw, Cc = x.shape
    out = np.zeros((B, hw, Cpad))
    nsub = batch_size // group
    for mb in range(B // batch_size):
        for j in range(nsub):
            idx = [mb * batch_size + j + g * nsub for g in range(group)]
            v = x[idx]
            d = v - v.mean(0, keepdims=True)
            std = np.sqrt((d ** 2).mean(0) + 1e-8).mean()
            out[idx, :, :Cc] = d
            out[idx, :, Cc] = std
    return _h(out)


def resize(y, S, ps, device=0):
    img = ((torch.tensor(np.asarray(y, np.float32)) + 1) / 2).clip(0, 1)
    B = img.shape[0]
    r = F.interpolate(img, size=(S, S), mode="bilinear", align_corners=False)
    G = S // ps
    return _h(r.view(B, 3, G, ps, G, ps).permute(0, 2, 4, 1, 3, 5).reshape(B * G * G, 3 * ps * ps).numpy())


def layernorm(x, g, b, device=0):
    return F.layer_norm(torch.tensor(x), (x.shape[1],), torch.tensor(g), torch.tensor(b), 1e-5).numpy()


def attention(qkv, n_img, L, heads, causal=False, device=0):
    t = torch.tensor(_h(qkv)).view(n_img, L, 3, heads, 64)
    q, k, v = (t[:, :, i].transpose(1, 2) for i in range(3))
    a = (q * 0.125) @ k.transpose(-1, -2)
    if causal:
        a = a + torch.full((L, L), float("-inf")).triu_(1)
    return _h((torch.softmax(a, -1) @ v).transpose(1, 2).reshape(n_img * L, heads * 64).numpy())


def noise(n_mb, hw, layer, mb0, generation, seed, device=0):
    side = int(round(math.sqrt(hw)))
    return np.stack([synth.noise_plane(seed, generation, mb0 + m, layer, side, side).reshape(-1) for m in range(n_mb)])
