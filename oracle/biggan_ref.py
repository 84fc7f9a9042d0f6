"""Oracle: BigGAN-deep generator (config C3, `DeepMindBigGAN256/512`).  TEST INFRASTRUCTURE — see
oracle/__init__.py.

PARITY UNPINNED.  The reference calls a third-party package whose source is absent from /root/reference:
`pytorch-pretrained-biggan==0.1.1` (requirements.txt; call sites models.py:69,77,84 and latent.py:9).
This file restates that package's published BigGAN-deep generator (Brock et al. 2019, "deep" variant as
released by DeepMind on TF-Hub and converted by HuggingFace) under the package's own state-dict keys, so
a real `biggan-deep-512` checkpoint goes through the same `load_tensor` calls as the synthetic one:

  BigGAN.forward(z, class_label, truncation):
      embed = embeddings(class_label)                 Linear(num_classes -> z_dim, bias=False)
      cond  = cat(z, embed)                           [B, 2*z_dim]
      h = gen_z(cond).view(B,4,4,16ch).permute(0,3,1,2)       spectral-norm Linear (+bias), TF NHWC order
      for layer in layers: GenBlock(h, cond, truncation) | SelfAttn(h)
      h = tanh(conv_to_rgb(relu(bn(h, truncation))))[:, :3]
  GenBlock (bottleneck, reduction 4): bn0-relu-conv1x1 -> bn1-relu-[nearest x2]-conv3x3 -> bn2-relu-conv3x3
      -> bn3-relu-conv1x1 ; skip = x0[:, :in/2] when in != out, nearest x2 when up ; out = h + skip
  BigGANBatchNorm: running stats stored for n_stats=51 truncation steps, eps 1e-4; conditional:
      out = (x - mean) / sqrt(var + eps) * (1 + scale(cond)) + offset(cond)   (spectral-norm Linears, no bias)
  SelfAttn: theta/phi (C/8), g (C/2) 1x1 convs without bias, 2x2 max-pool on phi and g,
      softmax(theta^T phi) over the pooled positions, o_conv (C/2 -> C), out = x + gamma * o
  Spectral norm at inference: W = weight_orig / (u . (W_mat v)) with the stored u, v (no power iteration in eval).

The reference-side call sites that ARE present are followed line by line: latent.py:20-24 (clip z to
+-2, softmax the class bits), models.py:75-86 (minibatch loop, truncation from config), generator.py:29-34
(biggan_norm).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LAYERS = {   # (up_sample, in_mult, out_mult) of the released configs; attention before layer index 8
    128: [(0, 16, 16), (1, 16, 16), (0, 16, 16), (1, 16, 8), (0, 8, 8), (1, 8, 4), (0, 4, 4), (1, 4, 2), (0, 2, 2), (1, 2, 1)],
    256: [(0, 16, 16), (1, 16, 16), (0, 16, 16), (1, 16, 8), (0, 8, 8), (1, 8, 8), (0, 8, 8), (1, 8, 4), (0, 4, 4), (1, 4, 2),
          (0, 2, 2), (1, 2, 1)],
    512: [(0, 16, 16), (1, 16, 16), (0, 16, 16), (1, 16, 8), (0, 8, 8), (1, 8, 8), (0, 8, 8), (1, 8, 4), (0, 4, 4), (1, 4, 2),
          (0, 2, 2), (1, 2, 1), (0, 1, 1), (1, 1, 1)],
}


def sn_weight(sd, prefix):
    """torch.nn.utils.spectral_norm in eval mode: weight_orig / sigma, sigma = u . (W_mat v)."""
    w = torch.as_tensor(sd[prefix + ".weight_orig"])
    u = torch.as_tensor(sd[prefix + ".weight_u"])
    v = torch.as_tensor(sd[prefix + ".weight_v"])
    sigma = torch.dot(u, torch.mv(w.flatten(1), v))
    return w / sigma


def stat_row(stats, truncation, n_stats):
    """BigGANBatchNorm: pick / blend the running-stat row for this truncation."""
    step = 1.0 / (n_stats - 1)
    coef, start = math.modf(truncation / step)
    start = int(start)
    if coef != 0.0:
        return stats[start] * coef + stats[start + 1] * (1 - coef)
    return stats[start]


def batchnorm(sd, prefix, x, truncation, cond, n_stats, eps):
    mean = stat_row(torch.as_tensor(sd[prefix + ".running_means"]), truncation, n_stats)
    var = stat_row(torch.as_tensor(sd[prefix + ".running_vars"]), truncation, n_stats)
    if cond is not None:
        weight = 1 + F.linear(cond, sn_weight(sd, prefix + ".scale"))[:, :, None, None]
        bias = F.linear(cond, sn_weight(sd, prefix + ".offset"))[:, :, None, None]
        return (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + eps) * weight + bias
    return F.batch_norm(x, mean, var, torch.as_tensor(sd[prefix + ".weight"]), torch.as_tensor(sd[prefix + ".bias"]),
                        training=False, momentum=0.0, eps=eps)


def snconv(sd, prefix, x, padding=0):
    b = sd.get(prefix + ".bias")
    return F.conv2d(x, sn_weight(sd, prefix), None if b is None else torch.as_tensor(b), padding=padding)


def gen_block(sd, p, x, cond, truncation, cin, cout, up, n_stats, eps):
    x0 = x
    x = snconv(sd, p + ".conv_0", F.relu(batchnorm(sd, p + ".bn_0", x, truncation, cond, n_stats, eps)))
    x = F.relu(batchnorm(sd, p + ".bn_1", x, truncation, cond, n_stats, eps))
    if up:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    x = snconv(sd, p + ".conv_1", x, padding=1)
    x = snconv(sd, p + ".conv_2", F.relu(batchnorm(sd, p + ".bn_2", x, truncation, cond, n_stats, eps)), padding=1)
    x = snconv(sd, p + ".conv_3", F.relu(batchnorm(sd, p + ".bn_3", x, truncation, cond, n_stats, eps)))
    if cin != cout:
        x0 = x0[:, :cin // 2]
    if up:
        x0 = F.interpolate(x0, scale_factor=2, mode="nearest")
    return x + x0


def self_attn(sd, p, x):
    B, ch, h, w = x.shape
    theta = snconv(sd, p + ".snconv1x1_theta", x).view(B, ch // 8, h * w)
    phi = F.max_pool2d(snconv(sd, p + ".snconv1x1_phi", x), 2, stride=2).view(B, ch // 8, h * w // 4)
    attn = torch.softmax(torch.bmm(theta.permute(0, 2, 1), phi), dim=-1)
    g = F.max_pool2d(snconv(sd, p + ".snconv1x1_g", x), 2, stride=2).view(B, ch // 2, h * w // 4)
    o = torch.bmm(g, attn.permute(0, 2, 1)).view(B, ch // 2, h, w)
    return x + torch.as_tensor(sd[p + ".gamma"]) * snconv(sd, p + ".snconv1x1_o_conv", o)


def generator(sd, z, class_probs, truncation, layers, attention_pos=8, ch=128, n_stats=51, eps=1e-4, taps=None):
    """BigGAN.forward.  z [B,z_dim], class_probs [B,num_classes] -> [B,3,R,R] in (-1,1)."""
    pre = "biggan."
    embed = F.linear(class_probs, torch.as_tensor(sd[pre + "embeddings.weight"]))
    cond = torch.cat((z, embed), dim=1)
    g = pre + "generator."
    h = F.linear(cond, sn_weight(sd, g + "gen_z"), torch.as_tensor(sd[g + "gen_z.bias"]))
    h = h.view(-1, 4, 4, 16 * ch).permute(0, 3, 1, 2).contiguous()
    m = 0
    for i, (up, cin, cout) in enumerate(layers):
        if i == attention_pos:
            h = self_attn(sd, g + "layers.%d" % m, h)
            if taps is not None:
                taps["attn"] = h
            m += 1
        h = gen_block(sd, g + "layers.%d" % m, h, cond, truncation, ch * cin, ch * cout, up, n_stats, eps)
        if taps is not None:
            taps["block%d" % i] = h
        m += 1
    h = F.relu(batchnorm(sd, g + "bn", h, truncation, None, n_stats, eps))
    h = snconv(sd, g + "conv_to_rgb", h, padding=1)[:, :3]
    return torch.tanh(h)


def latent_forward(x, dim_z):
    """latent.py:16-24: split the population row, clip z to [-2,2], softmax the class bits."""
    x = np.asarray(x)
    z = torch.tensor(x[:, :dim_z].astype(float)).float()
    cl = torch.tensor(x[:, dim_z:].astype(float)).float()
    return torch.clip(z, -2, 2), torch.softmax(cl, dim=1)


def generate(sd, x, dim_z, batch_size, truncation, layers, **kw):
    """generator.py:29-34 + models.py:75-86: minibatch loop, then biggan_norm (utils.py:14-17)."""
    z, cl = latent_forward(x, dim_z)
    assert z.shape[0] % batch_size == 0                                  # models.py:79
    outs = [generator(sd, z[i:i + batch_size], cl[i:i + batch_size], truncation, layers, **kw)
            for i in range(0, z.shape[0], batch_size)]
    img = torch.cat(outs)
    return ((img + 1) / 2.0).clip(0, 1)
