"""Oracle: StyleGAN2 generator + discriminator, torch-CPU fp32, in the
reference's own formulation (per-sample modulated weights + grouped conv,
conv_transpose2d + FIR, FIR + strided conv).  TEST INFRASTRUCTURE — see
oracle/__init__.py.  Every function cites the reference lines it restates
(paths relative to /root/reference).

Weights come in as a dict name -> float32 tensor with the reference's state-dict
keys prefixed `G_mapping.` / `G_synthesis.` / `D.` (clip_glass_amd/synth.py).
"""
import math

import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


def _fir(gain=1.0, up=1):
    # stylegan2/modules.py:169-203 _setup_filter_kernel([1,3,3,1])
    k = torch.tensor([1.0, 3.0, 3.0, 1.0])
    k = k[:, None] * k[None, :]
    k = k / k.sum()
    return (k * (gain * up ** 2)).float()


def _filter(x, kernel, pad0, pad1, stride=1):
    # stylegan2/modules.py:499-523 FilterLayer.forward (depthwise conv)
    c = x.shape[1]
    x = F.pad(x, [pad0, pad1, pad0, pad1])
    return F.conv2d(x, kernel[None, None].repeat(c, 1, 1, 1), stride=stride, groups=c)


def _bias_act(x, bias, bias_coef=1.0, act=True):
    # stylegan2/modules.py:276-297 BiasActivationWrapper.forward
    x = x + (bias * bias_coef).view(1, -1, *([1] * (x.dim() - 2)))
    if act:
        x = F.leaky_relu(x, 0.2) * SQRT2  # modules.py:31 gain sqrt(2)
    return x


def _dense(x, w, lr_mul=1.0):
    # stylegan2/modules.py:786-798 DenseLayer.forward; coef = 1/sqrt(fan_in)*lr_mul (modules.py:103-108)
    coef = lr_mul / math.sqrt(w.shape[1])
    return x.matmul((w * coef).t())


def g_mapping(sd, z, lr_mul=0.01, eps=1e-8):
    """stylegan2/models.py:590-627 GeneratorMapping.forward."""
    x = z * torch.rsqrt(torch.mean(z ** 2, dim=-1, keepdim=True) + eps)
    i = 0
    while "G_mapping.main.%d.layer.weight" % i in sd:
        x = _dense(x, sd["G_mapping.main.%d.layer.weight" % i], lr_mul)
        x = _bias_act(x, sd["G_mapping.main.%d.bias" % i], lr_mul)
        i += 1
    return x


def _mod_conv(x, latent, w, dense_w, dense_b, demod, up, eps=1e-8):
    """stylegan2/modules.py:920-967 ConvLayer.forward_mod (+ :1089-1139 ConvUpLayer._process)."""
    B, I = x.shape[0], x.shape[1]
    O, ks = w.shape[0], w.shape[2]
    w = w * (1.0 / math.sqrt(I * ks * ks))                      # modules.py:978-980 weight_coef
    style = _dense(latent, dense_w) + dense_b                   # modules.py:936 (bias_coef 1, linear)
    wm = w[None] * style.view(B, 1, I, 1, 1)                    # modules.py:940-942
    if demod:
        d = torch.rsqrt((wm.reshape(B, O, -1) ** 2).sum(-1) + eps)   # modules.py:945-954
        wm = wm * d.view(B, O, 1, 1, 1)
    xg = x.reshape(1, B * I, *x.shape[2:])                      # modules.py:960
    if not up:
        y = F.conv2d(xg, wm.reshape(B * O, I, ks, ks), padding=ks // 2, groups=B)  # :985-994
    else:
        wt = wm.transpose(1, 2).reshape(B * I, O, ks, ks)       # modules.py:141-166
        y = F.conv_transpose2d(xg, wt, stride=2, padding=0, groups=B)   # :1117-1123 (pad_once -> padding 0)
        y = _filter(y, _fir(gain=1.0, up=2), 1, 1)              # :1049-1072 pad0=pad1=1, :1131-1132
    return y.reshape(B, O, *y.shape[2:])


def _upsample_skip(y):
    """stylegan2/modules.py:580-602 Upsample.forward (FIR mode): zero-insert x2 then 4x4 FIR, pads [3,1]."""
    c = y.shape[1]
    z = F.conv_transpose2d(y, torch.ones(c, 1, 1, 1), stride=2, groups=c)
    return _filter(z, _fir(gain=1.0, up=2), 3, 1)               # :569-576 pad0=(3+1)//2+1=3, pad1=1


def g_synthesis(sd, dlat, noise=None, n_blocks=None):
    """stylegan2/models.py:969-1014 GeneratorSynthesis.forward.

    dlat: [B, n_latents, L].  noise: list of [H,W] (or [1,1,H,W]) tensors, one per
    noise layer in execution order (static_noise order, models.py:945-959); None -> no noise."""
    B = dlat.shape[0]
    p = "G_synthesis."
    if n_blocks is None:
        n_blocks = 0
        while p + "conv_blocks.%d.conv_block.0.bias" % n_blocks in sd:
            n_blocks += 1
    x = sd[p + "const"][None]
    y = None
    li = 0
    ni = 0
    for b in range(n_blocks):
        nl = 1 if b == 0 else 2
        for l in range(nl):
            q = p + "conv_blocks.%d.conv_block.%d" % (b, l)
            if x.shape[0] != B:
                x = x.expand(B, *x.shape[1:])
            x = _mod_conv(x, dlat[:, li], sd[q + ".layer.layer.weight"],
                          sd[q + ".layer.layer.dense.layer.weight"], sd[q + ".layer.layer.dense.bias"],
                          demod=True, up=(b > 0 and l == 0))
            if noise is not None:                                # modules.py:414-453 (one plane per call)
                x = x + sd[q + ".layer.weight"].view(1, 1, 1, 1) * torch.as_tensor(noise[ni]).reshape(1, 1, *x.shape[2:])
            ni += 1
            x = _bias_act(x, sd[q + ".bias"])
            li += 1
        if y is not None:                                        # models.py:1004-1006
            y = _upsample_skip(y)
        q = p + "to_data_layers.%d" % b                          # models.py:1011-1013, 852-870
        t = _mod_conv(x, dlat[:, li], sd[q + ".layer.weight"], sd[q + ".layer.dense.layer.weight"],
                      sd[q + ".layer.dense.bias"], demod=False, up=False)
        t = _bias_act(t, sd[q + ".bias"], act=False)
        y = t if y is None else y + t
    return y


def generator(sd, z, noise=None):
    """stylegan2/models.py:326-482 Generator.forward (eval mode, truncation inactive — SURVEY 8a note 3)."""
    w = g_mapping(sd, z)
    n_lat = 0
    b = 0
    while "G_synthesis.conv_blocks.%d.conv_block.0.bias" % b in sd:
        n_lat += 1 if b == 0 else 2
        b += 1
    n_lat += 1
    dlat = w[:, None, :].expand(w.shape[0], n_lat, w.shape[1])   # models.py:427-430
    return g_synthesis(sd, dlat, noise)


def _conv(x, w, stride=1, padding=0):
    # stylegan2/modules.py:978-994 ConvLayer.forward (non-modulated)
    coef = 1.0 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3])
    return F.conv2d(x, w * coef, stride=stride, padding=padding)


def minibatch_std(x, group_size=4, eps=1e-8):
    """stylegan2/modules.py:701-747 MinibatchStd.forward."""
    B = x.shape[0]
    g = group_size or B
    y = x.view(g, -1, *x.shape[1:]).float()
    y = y - y.mean(dim=0, keepdim=True)
    # REFERENCE QUIRK (modules.py:726-730): for fp32 inputs `.float()` returns the same
    # storage and `y -= y.mean(...)` is in-place on a view of `input`, so the features
    # that reach torch.cat (:746) are the group-mean-SUBTRACTED ones.  Reproduced here.
    x = y.reshape(B, *x.shape[1:])
    y = torch.sqrt((y ** 2).mean(dim=0) + eps)
    y = y.view(y.shape[0], -1).mean(dim=-1)
    y = y.view(-1, 1, 1, 1).repeat(g, 1, 1, 1).expand(B, 1, *x.shape[2:])
    return torch.cat([x, y], dim=1)


def discriminator(sd, img, mbstd_group=4):
    """stylegan2/models.py:1193-1230 Discriminator.forward (resnet arch, no labels)."""
    p = "D."
    x = _conv(img, sd[p + "from_data_layers.0.layer.weight"])                       # models.py:1125-1143
    x = _bias_act(x, sd[p + "from_data_layers.0.bias"])
    i = 0
    while p + "conv_blocks.%d.projection.weight" % i in sd:                         # modules.py:1587-1601
        q = p + "conv_blocks.%d" % i
        h = _bias_act(_conv(x, sd[q + ".conv_block.0.layer.weight"], padding=1), sd[q + ".conv_block.0.bias"])
        h = _filter(h, _fir(), 2, 2)                                                # :1204-1220 pad=(4-2)+(3-1)=4
        h = _bias_act(_conv(h, sd[q + ".conv_block.1.layer.weight"], stride=2), sd[q + ".conv_block.1.bias"])
        s = _filter(x, _fir(), 1, 1)                                                # projection: pad=(4-2)+(1-1)=2
        s = _conv(s, sd[q + ".projection.weight"], stride=2)
        x = (h + s) * (1.0 / SQRT2)                                                 # :1599-1600
        i += 1
    q = p + "conv_blocks.%d.1" % i
    x = minibatch_std(x, mbstd_group)
    x = _bias_act(_conv(x, sd[q + ".conv_block.0.layer.weight"], padding=1), sd[q + ".conv_block.0.bias"])
    x = x.reshape(x.shape[0], -1)                                                   # models.py:1224
    x = _bias_act(_dense(x, sd[p + "dense.0.layer.weight"]), sd[p + "dense.0.bias"])
    x = _bias_act(_dense(x, sd[p + "dense.1.layer.weight"]), sd[p + "dense.1.bias"], act=False)
    return x
