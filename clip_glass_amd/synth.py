"""Deterministic synthetic weights / noise / inputs: value = f(seed, name, shape).

There is no network and no pretrained checkpoint in this environment, so every
test, the oracle and bench.py regenerate the *same* tensors from (seed, name)
with numpy's PCG64.  Names are the reference's own state-dict keys
(stylegan2/models.py:111-132 container: sub-dicts `G_mapping`, `G_synthesis`;
clip/model.py:363-399 for CLIP), prefixed with the sub-model
(`G_mapping.`, `G_synthesis.`, `D.`, `clip.`), so a real checkpoint loaded by
the host shim goes through exactly the same `load_tensor(name, array)` calls.

Distributions follow the reference initialisers so activations have sane
magnitudes (modules.py:87-118 `_get_weight_and_coef`: stored weight ~
N(0, 1/lr_mul), runtime coefficient applied every forward; biases = fill value,
perturbed here so the bias paths are exercised; noise strengths initialise to 0
in the reference (modules.py:326) and are set non-zero here for the same reason).
"""
import zlib
from collections import OrderedDict

import numpy as np

FFHQ_CHANNELS = [32, 64, 128, 256, 512, 512, 512, 512, 512]  # 1024px config-f, G order last->first


def _rng(seed, name):
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence([int(seed), zlib.crc32(name.encode())])))


def normal(seed, name, shape, std=1.0, mean=0.0):
    return (_rng(seed, name).standard_normal(shape, dtype=np.float32) * np.float32(std) + np.float32(mean)).astype(np.float32)


# --------------------------------------------------------------------------
# StyleGAN2 architecture bookkeeping (own restatement of the layer list that
# stylegan2/models.py:771-896 (G) and :1043-1191 (D) build).
# --------------------------------------------------------------------------
def g_layers(channels):
    """Synthesis conv layers in execution order.

    Returns list of dicts(name, cin, cout, res, up, latent_idx) for the 1+2*(n-1)
    modulated 3x3 convs and list of toRGB dicts(name, cin, res, latent_idx).
    `channels` is in the reference's G order (last layer -> first layer)."""
    n = len(channels)
    convs, rgbs = [], []
    res = 4
    li = 0
    for b in range(n):
        cin = channels[-1] if b == 0 else channels[-b]
        cout = channels[-1] if b == 0 else channels[-b - 1]
        nl = 1 if b == 0 else 2
        if b > 0:
            res *= 2
        for l in range(nl):
            convs.append(dict(name="conv_blocks.%d.conv_block.%d" % (b, l),
                              cin=cin if l == 0 else cout, cout=cout, res=res,
                              up=(b > 0 and l == 0), latent_idx=li))
            li += 1
        rgbs.append(dict(name="to_data_layers.%d" % b, cin=cout, res=res, latent_idx=li))
    return convs, rgbs


def stylegan2_g_spec(channels=FFHQ_CHANNELS, latent_size=512, mapping_layers=8, data_channels=3):
    """(name, shape, kind) for every tensor of G (mapping + synthesis)."""
    spec = []
    for i in range(mapping_layers):
        spec.append(("G_mapping.main.%d.layer.weight" % i, (latent_size, latent_size), "w_map"))
        spec.append(("G_mapping.main.%d.bias" % i, (latent_size,), "b_map"))
    spec.append(("G_synthesis.const", (channels[-1], 4, 4), "const"))
    convs, rgbs = g_layers(channels)
    for c in convs:
        p = "G_synthesis." + c["name"]
        spec.append((p + ".layer.layer.weight", (c["cout"], c["cin"], 3, 3), "w"))
        spec.append((p + ".layer.layer.dense.layer.weight", (c["cin"], latent_size), "w"))
        spec.append((p + ".layer.layer.dense.bias", (c["cin"],), "b_style"))
        spec.append((p + ".layer.weight", (1,), "noise_strength"))
        spec.append((p + ".bias", (c["cout"],), "b"))
    for r in rgbs:
        p = "G_synthesis." + r["name"]
        spec.append((p + ".layer.weight", (data_channels, r["cin"], 1, 1), "w"))
        spec.append((p + ".layer.dense.layer.weight", (r["cin"], latent_size), "w"))
        spec.append((p + ".layer.dense.bias", (r["cin"],), "b_style"))
        spec.append((p + ".bias", (data_channels,), "b"))
    return spec


def stylegan2_d_spec(channels=FFHQ_CHANNELS, data_channels=3):
    """(name, shape, kind) for every tensor of D. `channels` in D order (first->last)."""
    n = len(channels)
    spec = [("D.from_data_layers.0.layer.weight", (channels[0], data_channels, 1, 1), "w"),
            ("D.from_data_layers.0.bias", (channels[0],), "b")]
    for i in range(n - 1):
        p = "D.conv_blocks.%d" % i
        spec.append((p + ".conv_block.0.layer.weight", (channels[i], channels[i], 3, 3), "w"))
        spec.append((p + ".conv_block.0.bias", (channels[i],), "b"))
        spec.append((p + ".conv_block.1.layer.weight", (channels[i + 1], channels[i], 3, 3), "w"))
        spec.append((p + ".conv_block.1.bias", (channels[i + 1],), "b"))
        spec.append((p + ".projection.weight", (channels[i + 1], channels[i], 1, 1), "w"))
    p = "D.conv_blocks.%d.1" % (n - 1)
    spec.append((p + ".conv_block.0.layer.weight", (channels[-1], channels[-1] + 1, 3, 3), "w"))
    spec.append((p + ".conv_block.0.bias", (channels[-1],), "b"))
    spec.append(("D.dense.0.layer.weight", (channels[-1], channels[-1] * 16), "w"))
    spec.append(("D.dense.0.bias", (channels[-1],), "b"))
    spec.append(("D.dense.1.layer.weight", (1, channels[-1]), "w"))
    spec.append(("D.dense.1.bias", (1,), "b"))
    return spec


def clip_visual_spec(width=768, layers=12, patch=32, res=224, out_dim=512):
    """(name, shape, kind) of the CLIP visual tower (clip/model.py:201-216,166-178)."""
    n_tok = (res // patch) ** 2 + 1
    s = width ** -0.5
    spec = [("clip.visual.conv1.weight", (width, 3, patch, patch), ("n16", (3 * patch * patch) ** -0.5)),
            ("clip.visual.class_embedding", (width,), ("n", s)),
            ("clip.visual.positional_embedding", (n_tok, width), ("n", s)),
            ("clip.visual.ln_pre.weight", (width,), ("ln_w", 0.1)),
            ("clip.visual.ln_pre.bias", (width,), ("n", 0.1))]
    proj_std = s * (2 * layers) ** -0.5
    for i in range(layers):
        p = "clip.visual.transformer.resblocks.%d." % i
        spec += [(p + "ln_1.weight", (width,), ("ln_w", 0.1)), (p + "ln_1.bias", (width,), ("n", 0.1)),
                 (p + "attn.in_proj_weight", (3 * width, width), ("n16", s)),
                 (p + "attn.in_proj_bias", (3 * width,), ("n16", 0.02)),
                 (p + "attn.out_proj.weight", (width, width), ("n16", proj_std)),
                 (p + "attn.out_proj.bias", (width,), ("n16", 0.02)),
                 (p + "ln_2.weight", (width,), ("ln_w", 0.1)), (p + "ln_2.bias", (width,), ("n", 0.1)),
                 (p + "mlp.c_fc.weight", (4 * width, width), ("n16", (2 * width) ** -0.5)),
                 (p + "mlp.c_fc.bias", (4 * width,), ("n16", 0.02)),
                 (p + "mlp.c_proj.weight", (width, 4 * width), ("n16", proj_std)),
                 (p + "mlp.c_proj.bias", (width,), ("n16", 0.02))]
    spec += [("clip.visual.ln_post.weight", (width,), ("ln_w", 0.1)),
             ("clip.visual.ln_post.bias", (width,), ("n", 0.1)),
             ("clip.visual.proj", (width, out_dim), ("n16", s))]
    return spec


def clip_text_spec(width=512, layers=12, ctx=77, vocab=49408, out_dim=512):
    """(name, shape, kind) of the CLIP text tower (clip/model.py:277-290); init stds follow
    OpenAI's initialize_parameters (not in the vendored file)."""
    s = width ** -0.5
    proj_std = s * (2 * layers) ** -0.5
    spec = [("clip.token_embedding.weight", (vocab, width), ("n", 0.02)),
            ("clip.positional_embedding", (ctx, width), ("n", 0.01))]
    for i in range(layers):
        p = "clip.transformer.resblocks.%d." % i
        spec += [(p + "ln_1.weight", (width,), ("ln_w", 0.1)), (p + "ln_1.bias", (width,), ("n", 0.1)),
                 (p + "attn.in_proj_weight", (3 * width, width), ("n16", s)),
                 (p + "attn.in_proj_bias", (3 * width,), ("n16", 0.02)),
                 (p + "attn.out_proj.weight", (width, width), ("n16", proj_std)),
                 (p + "attn.out_proj.bias", (width,), ("n16", 0.02)),
                 (p + "ln_2.weight", (width,), ("ln_w", 0.1)), (p + "ln_2.bias", (width,), ("n", 0.1)),
                 (p + "mlp.c_fc.weight", (4 * width, width), ("n16", (2 * width) ** -0.5)),
                 (p + "mlp.c_fc.bias", (4 * width,), ("n16", 0.02)),
                 (p + "mlp.c_proj.weight", (width, 4 * width), ("n16", proj_std)),
                 (p + "mlp.c_proj.bias", (width,), ("n16", 0.02))]
    spec += [("clip.ln_final.weight", (width,), ("ln_w", 0.1)), ("clip.ln_final.bias", (width,), ("n", 0.1)),
             ("clip.text_projection", (width, out_dim), ("n16", s)),
             ("clip.logit_scale", (), ("n", 0.0))]
    return spec


def gpt2_spec(n_embd=768, n_layer=12, vocab=50257, n_positions=1024):
    """(name, shape, kind) of GPT2LMHeadModel (gpt2/model.py:126-211) under the keys the reference holds after
    gpt2/utils.py load_weight; Conv1D weights are [nx, nf]; init std 0.02 (model.py:34, config.py:17)."""
    p = "gpt2.transformer."
    spec = [(p + "wte.weight", (vocab, n_embd), ("n", 0.02)), (p + "wpe.weight", (n_positions, n_embd), ("n", 0.01))]
    for i in range(n_layer):
        q = p + "h.%d." % i
        spec += [(q + "ln_1.weight", (n_embd,), ("ln_w", 0.1)), (q + "ln_1.bias", (n_embd,), ("n", 0.1)),
                 (q + "attn.c_attn.weight", (n_embd, 3 * n_embd), ("n", 0.05)), (q + "attn.c_attn.bias", (3 * n_embd,), ("n", 0.02)),
                 (q + "attn.c_proj.weight", (n_embd, n_embd), ("n", 0.02)), (q + "attn.c_proj.bias", (n_embd,), ("n", 0.02)),
                 (q + "ln_2.weight", (n_embd,), ("ln_w", 0.1)), (q + "ln_2.bias", (n_embd,), ("n", 0.1)),
                 (q + "mlp.c_fc.weight", (n_embd, 4 * n_embd), ("n", 0.05)), (q + "mlp.c_fc.bias", (4 * n_embd,), ("n", 0.02)),
                 (q + "mlp.c_proj.weight", (4 * n_embd, n_embd), ("n", 0.02)), (q + "mlp.c_proj.bias", (n_embd,), ("n", 0.02))]
    spec += [(p + "ln_f.weight", (n_embd,), ("ln_w", 0.1)), (p + "ln_f.bias", (n_embd,), ("n", 0.1))]
    return spec


def make_tensor(seed, name, shape, kind, map_lr_mul=0.01):
    if isinstance(kind, tuple):
        k, std = kind
        if k == "ln_w":
            return normal(seed, name, shape, std, 1.0)
        t = normal(seed, name, shape, std)
        if k == "n16":  # convert_weights (clip/model.py:339-360): stored as fp16
            t = t.astype(np.float16).astype(np.float32)
        return t
    if kind == "w":
        return normal(seed, name, shape, 1.0)
    if kind == "w_map":
        return normal(seed, name, shape, 1.0 / map_lr_mul)
    if kind == "b":
        return normal(seed, name, shape, 0.2)
    if kind == "b_map":
        return normal(seed, name, shape, 0.2 / map_lr_mul)
    if kind == "b_style":
        return normal(seed, name, shape, 0.2, 1.0)
    if kind == "noise_strength":
        return normal(seed, name, shape, 0.2)
    if kind == "const":
        return normal(seed, name, shape, 1.0)
    raise ValueError(kind)


def make_state(spec, seed=0):
    return OrderedDict((name, make_tensor(seed, name, shape, kind)) for name, shape, kind in spec)


# --------------------------------------------------------------------------
# Counter-based noise: Philox4x32-10 + Box-Muller.  Numpy mirror of the device
# generator in csrc/noise.hip, so the oracle can be fed bit-compatible noise
# (up to libm differences ~1e-6 in log/cos).  Noise is a pure function of
# (seed, generation, global minibatch index, layer, pixel) -> shard-invariant
# (SURVEY 8(e)).  Reference draws `normal_()` per G call: modules.py:428-452.
# --------------------------------------------------------------------------
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0); k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * _M0
            p1 = c2.astype(np.uint64) * _M1
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32); lo0 = p0.astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32); lo1 = p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def noise_plane(seed, generation, minibatch, layer, h, w):
    """N(0,1) plane [h,w] float32 — device-compatible (csrc/noise.hip)."""
    n = h * w
    nq = (n + 3) // 4
    idx = np.arange(nq, dtype=np.uint32)
    r = philox4x32(idx, np.uint32(layer), np.uint32(minibatch), np.uint32(generation),
                   np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    u = [(x.astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -32) for x in r]
    out = np.empty((nq, 4), dtype=np.float32)
    two_pi = np.float32(6.283185307179586)
    for j in (0, 1):
        rad = np.sqrt(np.float32(-2.0) * np.log(u[2 * j])).astype(np.float32)
        ang = (two_pi * u[2 * j + 1]).astype(np.float32)
        out[:, 2 * j] = rad * np.cos(ang)
        out[:, 2 * j + 1] = rad * np.sin(ang)
    return out.reshape(-1)[:n].reshape(h, w)


def g_noise_planes(seed, generation, minibatch, channels=FFHQ_CHANNELS):
    """The 1+2*(n-1) noise planes of one G call (one minibatch), execution order."""
    convs, _ = g_layers(channels)
    return [noise_plane(seed, generation, minibatch, i, c["res"], c["res"]) for i, c in enumerate(convs)]


def latents(seed, pop, dim=512):
    """RandomState(seed).normal(size=(pop, dim)) float64 — matches NormalRandomSampling
    (operators.py:24-25) and SURVEY 8(d)."""
    return np.random.RandomState(seed).normal(size=(pop, dim))


def make_target(feats, seed=0, spread=0.6):
    """Synthetic target feature giving cosine sims well away from 0 (the 1e-3
    *relative* bar is ill-conditioned near 0): unit(feats[0]) + spread * unit(random)."""
    f0 = np.asarray(feats[0], dtype=np.float64)
    r = normal(seed, "target", f0.shape).astype(np.float64)
    t = f0 / np.linalg.norm(f0) + spread * r / np.linalg.norm(r)
    return t.astype(np.float32)


# --------------------------------------------------------------------------
# BigGAN-deep (config C3).  Keys are pytorch-pretrained-biggan's state-dict keys under "biggan."
# (spectral-norm modules hold weight_orig / weight_u / weight_v); layer tables are the released
# biggan-deep-128/256/512 configs.
# --------------------------------------------------------------------------
BIGGAN_LAYERS = {
    128: [(0, 16, 16), (1, 16, 16), (0, 16, 16), (1, 16, 8), (0, 8, 8), (1, 8, 4), (0, 4, 4), (1, 4, 2), (0, 2, 2), (1, 2, 1)],
    256: [(0, 16, 16), (1, 16, 16), (0, 16, 16), (1, 16, 8), (0, 8, 8), (1, 8, 8), (0, 8, 8), (1, 8, 4), (0, 4, 4), (1, 4, 2),
          (0, 2, 2), (1, 2, 1)],
    512: [(0, 16, 16), (1, 16, 16), (0, 16, 16), (1, 16, 8), (0, 8, 8), (1, 8, 8), (0, 8, 8), (1, 8, 4), (0, 4, 4), (1, 4, 2),
          (0, 2, 2), (1, 2, 1), (0, 1, 1), (1, 1, 1)],
}


def biggan_spec(layers, attention_pos=8, ch=128, z_dim=128, num_classes=1000, n_stats=51):
    """(name, shape, kind) of BigGAN-deep's generator; kind "sn" = spectral-normalised weight_orig
    (make_biggan_state adds the matching weight_u / weight_v)."""
    g = "biggan.generator."
    cd = 2 * z_dim
    spec = [("biggan.embeddings.weight", (z_dim, num_classes), ("n", 1.0)),
            (g + "gen_z.weight_orig", (16 * 16 * ch, cd), "sn"), (g + "gen_z.bias", (16 * 16 * ch,), ("n", 1.0))]

    def bn(p, c, conditional=True):
        out = [(p + ".running_means", (n_stats, c), ("n", 0.3)), (p + ".running_vars", (n_stats, c), "var")]
        if conditional:
            out += [(p + ".scale.weight_orig", (c, cd), "sn"), (p + ".offset.weight_orig", (c, cd), "sn")]
        else:
            out += [(p + ".weight", (c,), ("ln_w", 0.2)), (p + ".bias", (c,), ("n", 0.2))]
        return out

    m = 0
    for i, (up, cin, cout) in enumerate(layers):
        if i == attention_pos:
            p = g + "layers.%d" % m
            c = ch * cin
            spec += [(p + ".snconv1x1_theta.weight_orig", (c // 8, c, 1, 1), "sn"),
                     (p + ".snconv1x1_phi.weight_orig", (c // 8, c, 1, 1), "sn"),
                     (p + ".snconv1x1_g.weight_orig", (c // 2, c, 1, 1), "sn"),
                     (p + ".snconv1x1_o_conv.weight_orig", (c, c // 2, 1, 1), "sn"),
                     (p + ".gamma", (1,), ("gamma", 0.0))]
            m += 1
        p = g + "layers.%d" % m
        ci, co = ch * cin, ch * cout
        mid = ci // 4
        for k, (a, b, ks) in enumerate([(ci, mid, 1), (mid, mid, 3), (mid, mid, 3), (mid, co, 1)]):
            spec += bn(p + ".bn_%d" % k, a)
            spec += [(p + ".conv_%d.weight_orig" % k, (b, a, ks, ks), "sn"), (p + ".conv_%d.bias" % k, (b,), ("n", 0.2))]
        m += 1
    spec += bn(g + "bn", ch, conditional=False)
    spec += [(g + "conv_to_rgb.weight_orig", (ch, ch, 3, 3), "sn"), (g + "conv_to_rgb.bias", (ch,), ("n", 0.2))]
    return spec


def make_biggan_state(spec, seed=0, power_iters=8):
    """Synthetic BigGAN-deep state: Gaussian weight_orig with (u, v) converged by power iteration, as a
    trained checkpoint holds them (random u, v would make weight_orig / sigma explode)."""
    sd = OrderedDict()
    for name, shape, kind in spec:
        if kind == "sn":
            w = normal(seed, name, shape, 1.0)
            wm = w.reshape(shape[0], -1).astype(np.float64)
            u = normal(seed, name + "/u", (shape[0],)).astype(np.float64)
            v = None
            for _ in range(power_iters):
                v = wm.T @ u
                v /= np.linalg.norm(v) + 1e-12
                u = wm @ v
                u /= np.linalg.norm(u) + 1e-12
            sd[name] = w
            base = name[:-len("weight_orig")]
            sd[base + "weight_u"] = u.astype(np.float32)
            sd[base + "weight_v"] = v.astype(np.float32)
        elif kind == "var":
            sd[name] = (0.3 + 0.5 * _rng(seed, name).random(shape, dtype=np.float32)).astype(np.float32)
        elif isinstance(kind, tuple) and kind[0] == "gamma":
            sd[name] = np.asarray([0.7], dtype=np.float32)   # released checkpoints hold a learned non-zero gamma
        else:
            sd[name] = make_tensor(seed, name, shape, kind)
    return sd


def biggan_population(seed, pop, dim_z=128, num_classes=1000, p_class=5 / 1000):
    """Rows [z | class bits] as the reference samples them (operators.py:15,47): truncnorm(-2,2) and
    Bernoulli(5/1000) bits; float64 like pymoo's mixed-variable population after `.astype(float)`."""
    from scipy.stats import truncnorm
    rs = np.random.RandomState(seed)
    z = truncnorm.rvs(-2, 2, size=(pop, dim_z), random_state=rs).astype(np.float32)
    bits = rs.random_sample((pop, num_classes)) < p_class
    return np.concatenate([z.astype(np.float64), bits.astype(np.float64)], axis=1)


# --------------------------------------------------------------------------
# Synthetic BPE assets at the REAL vocabulary sizes (GPT-2: 50257 ids, CLIP: 49408 ids) in the reference's file formats
# (gpt2/encoder.py:107-115 reads encoder.json + vocab.bpe, clip/simple_tokenizer.py:66-72 reads the gzipped merge list).
# The reference's own data files are not on the GPU box; config C5's host stage (decode ids -> text -> CLIP ids) needs SOME
# vocabulary of the right size and shape to be the real path: seeded merges over lower-case words, so decoded texts are plain
# ASCII that clip.tokenize accepts.  Used by `bench.py --config gpt2` and tests/test_gpt2.py.
def write_bpe_assets(out_dir, gpt2_vocab=50257, clip_vocab=49408, seed=0, init_text="the picture of"):
    """Writes encoder.json, vocab.bpe, clip_bpe.txt.gz under out_dir; returns their paths.  GPT-2 ids: the 256 byte symbols, then
    one id per merge, then <|endoftext|> (= gpt2_vocab - 1, config.py:28).  The words of `init_text` are single tokens (their merge
    chains come first), so the decode context has the reference's length (20 latent + 3 prompt tokens, models.py:30,45-60)."""
    import gzip
    import json
    import os
    from .tokenizer import _byte_alphabet
    b2c, order = _byte_alphabet()
    sp = b2c[32]                                         # the byte alphabet's symbol for " " (U+0120)
    rs = np.random.RandomState(seed + 7919)
    letters = [chr(c) for c in range(97, 123)]

    def grow(n_merges, chains, prefix_pool, suffix):
        """n_merges unique merges (a, b) over lower-case letters.  Token classes: word-initial (GPT-2: sp + letters), letter-only,
        word-final (CLIP: letters + '</w>').  Only tokens of <= 4 letters are merged further (the pools stay dense: no rejection
        sampling on length), so no token exceeds 8 letters."""
        nlet = lambda tok: len(tok.replace("</w>", "").replace(sp, ""))
        merges, seen = [], set(letters) | set(prefix_pool)
        mid, head = list(letters), list(prefix_pool)
        tail = [c + suffix for c in letters] if suffix else []
        seen |= set(tail)

        def add(a, b):
            tok = a + b
            if tok in seen:
                return False
            seen.add(tok)
            merges.append((a, b))
            if nlet(tok) <= 4:
                (head if tok.startswith(sp) else tail if suffix and tok.endswith(suffix) else mid).append(tok)
            return True
        for chain in chains:                             # words that must end up as ONE token: left-to-right merge chains
            cur = chain[0]
            for nxt in chain[1:]:
                add(cur, nxt)
                cur = cur + nxt
        while len(merges) < n_merges:
            kind = rs.randint(0, 3)
            if kind == 0 and head:
                add(head[rs.randint(len(head))], mid[rs.randint(len(mid))])
            elif kind == 1 and tail:
                add(mid[rs.randint(len(mid))], tail[rs.randint(len(tail))])
            else:
                add(mid[rs.randint(len(mid))], mid[rs.randint(len(mid))])
        return merges

    os.makedirs(out_dir, exist_ok=True)
    # ---- GPT-2: words carry their leading space (sp + letters) ----
    words = init_text.split(" ")
    chains = []
    for i, w in enumerate(words):
        chains.append(([sp] if i else []) + list(w))
    g_merges = grow(gpt2_vocab - 256 - 1, chains, [sp], None)
    enc = {b2c[b]: i for i, b in enumerate(order)}
    for a, b in g_merges:
        enc[a + b] = len(enc)
    enc["<|endoftext|>"] = len(enc)
    assert len(enc) == gpt2_vocab
    ej, vb, cb = os.path.join(out_dir, "encoder.json"), os.path.join(out_dir, "vocab.bpe"), os.path.join(out_dir, "clip_bpe.txt.gz")
    with open(ej, "w") as f:
        json.dump(enc, f)
    with open(vb, "w", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(a + " " + b for a, b in g_merges) + "\n")
    # ---- CLIP: 256 symbols + 256 word-final symbols + merges + 2 specials; words end in '</w>' ----
    c_merges = grow(clip_vocab - 512 - 2, [], [], "</w>")
    with gzip.open(cb, "wt", encoding="utf-8") as f:
        f.write("#version: synthetic\n" + "\n".join(a + " " + b for a, b in c_merges) + "\n")
    return ej, vb, cb
