"""Oracle: CLIP ViT visual tower (+ text tower) in torch-CPU fp32.  TEST
INFRASTRUCTURE — see oracle/__init__.py.

dtype decision: weights are the fp16-rounded values the reference
holds after convert_weights (clip/model.py:339-360,397); arithmetic is fp32.
"""
import torch
import torch.nn.functional as F


def _ln(x, w, b):
    # clip/model.py:152-158 (fp32 LayerNorm, eps 1e-5)
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, 1e-5)


def _resblock(x, sd, p, heads, mask=None):
    """clip/model.py:166-187 ResidualAttentionBlock.forward; x is [N, L, D]."""
    N, L, D = x.shape
    h = _ln(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
    qkv = h @ sd[p + "attn.in_proj_weight"].t() + sd[p + "attn.in_proj_bias"]      # nn.MultiheadAttention
    q, k, v = qkv.split(D, dim=-1)
    hd = D // heads
    q = q.view(N, L, heads, hd).transpose(1, 2) * (hd ** -0.5)
    k = k.view(N, L, heads, hd).transpose(1, 2)
    v = v.view(N, L, heads, hd).transpose(1, 2)
    a = q @ k.transpose(-1, -2)
    if mask is not None:
        a = a + mask
    a = torch.softmax(a, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(N, L, D)
    x = x + o @ sd[p + "attn.out_proj.weight"].t() + sd[p + "attn.out_proj.bias"]
    h = _ln(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
    h = h @ sd[p + "mlp.c_fc.weight"].t() + sd[p + "mlp.c_fc.bias"]
    h = h * torch.sigmoid(1.702 * h)                                                # :161-163 QuickGELU
    return x + h @ sd[p + "mlp.c_proj.weight"].t() + sd[p + "mlp.c_proj.bias"]


def encode_image(sd, img, heads=None):
    """clip/model.py:218-235 VisualTransformer.forward; img [N,3,R,R] -> [N,out_dim]."""
    p = "clip.visual."
    w = sd[p + "conv1.weight"]
    width, patch = w.shape[0], w.shape[-1]
    heads = heads or width // 64                                                    # :267
    x = F.conv2d(img.float(), w, stride=patch)
    x = x.reshape(x.shape[0], width, -1).permute(0, 2, 1)
    cls = sd[p + "class_embedding"].view(1, 1, -1).expand(x.shape[0], 1, width)
    x = torch.cat([cls, x], dim=1) + sd[p + "positional_embedding"]
    x = _ln(x, sd[p + "ln_pre.weight"], sd[p + "ln_pre.bias"])
    i = 0
    while p + "transformer.resblocks.%d.ln_1.weight" % i in sd:
        x = _resblock(x, sd, p + "transformer.resblocks.%d." % i, heads)
        i += 1
    x = _ln(x[:, 0, :], sd[p + "ln_post.weight"], sd[p + "ln_post.bias"])
    return x @ sd[p + "proj"]


def cosine_similarity(a, b, eps=1e-8):
    """torch.cosine_similarity(a[P,D], b[1,D]) — generator.py:51."""
    return F.cosine_similarity(a, b, dim=1, eps=eps)


def encode_text(sd, tokens, heads=None):
    """clip/model.py:307-320 CLIP.encode_text; tokens [N, ctx] int64 -> [N, out_dim]."""
    p = "clip."
    x = sd[p + "token_embedding.weight"][tokens] + sd[p + "positional_embedding"]
    width = x.shape[-1]
    heads = heads or width // 64                                                    # :385
    L = x.shape[1]
    mask = torch.full((L, L), float("-inf")).triu_(1)                               # :292-298
    i = 0
    while p + "transformer.resblocks.%d.ln_1.weight" % i in sd:
        x = _resblock(x, sd, p + "transformer.resblocks.%d." % i, heads, mask)
        i += 1
    x = _ln(x, sd[p + "ln_final.weight"], sd[p + "ln_final.bias"])
    return x[torch.arange(x.shape[0]), tokens.argmax(dim=-1)] @ sd[p + "text_projection"]   # :318
