"""Oracle: the whole fitness path `GenerationProblem._evaluate` (problem.py:14-29)
for the StyleGAN2 + CLIP txt2img configs.  TEST INFRASTRUCTURE — see
oracle/__init__.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import clip_ref, stylegan2_ref as sg


def generate(sd, x, batch_size, noise_fn=None):
    """latent.py:37-41 + generator.py:29-34 + models.py:108-118: minibatch loop over G, then biggan_norm.

    noise_fn(minibatch_index) -> list of noise planes for that G call (None = no noise)."""
    z = torch.tensor(np.asarray(x).astype(float)).float()              # latent.py:38
    assert z.shape[0] % batch_size == 0                                 # models.py:112
    outs = []
    for i in range(z.shape[0] // batch_size):                           # models.py:114-116
        noise = noise_fn(i) if noise_fn is not None else None
        outs.append(sg.generator(sd, z[i * batch_size:(i + 1) * batch_size], noise))
    img = torch.cat(outs)
    return ((img + 1) / 2.0).clip(0, 1)                                 # utils.py:14-17 biggan_norm


def resize224(img, size=224):
    """generator.py:45 kornia.resize(input,(224,224)) — kornia==0.4.1, source absent:
    restated as bilinear / align_corners=False / no antialias (parity unpinned)."""
    return F.interpolate(img, size=(size, size), mode="bilinear", align_corners=False)


def clip_similarity(sd, img, text_features, size=224):
    """generator.py:43-51 (txt2img branch)."""
    feats = clip_ref.encode_image(sd, resize224(img, size))
    return clip_ref.cosine_similarity(feats, torch.as_tensor(text_features).view(1, -1)), feats


def discriminate(sd, img, batch_size):
    """generator.py:36-38 (biggan_denorm, utils.py:19-21) + models.py:120-129."""
    img = img * 2 - 1
    outs = [sg.discriminator(sd, img[i * batch_size:(i + 1) * batch_size])
            for i in range(img.shape[0] // batch_size)]
    return torch.cat(outs)


def evaluate(sd, x, text_features, batch_size, use_discriminator, noise_fn=None, clip_size=224, detail=None):
    """problem.py:14-29.  Returns F ([P] or [P,2]) float32 and G zeros [P]."""
    with torch.no_grad():
        img = generate(sd, x, batch_size, noise_fn)
        sim, feats = clip_similarity(sd, img, text_features, clip_size)
        sim = sim.numpy()
        if detail is not None:
            detail["image"] = img
            detail["features"] = feats
        if use_discriminator:
            dis = discriminate(sd, img, batch_size)
            if detail is not None:
                detail["dis"] = dis
            hinge = torch.relu(1 - dis).squeeze(1).numpy()
            Fv = np.column_stack((-sim, hinge))
        else:
            Fv = -sim
    return Fv, np.zeros((np.asarray(x).shape[0]))


def evaluate_biggan(sd, x, text_features, dim_z, batch_size, truncation, layers, clip_size=224, detail=None, **kw):
    """problem.py:14-29 for the DeepMindBigGAN configs (n_obj = 1, no discriminator: config.py:38,60).
    The generator itself is pytorch-pretrained-biggan's (source absent; oracle/biggan_ref.py, parity unpinned)."""
    from . import biggan_ref
    with torch.no_grad():
        img = biggan_ref.generate(sd, x, dim_z, batch_size, truncation, layers, **kw)
        sim, feats = clip_similarity(sd, img, text_features, clip_size)
        if detail is not None:
            detail["image"] = img
            detail["features"] = feats
        return -sim.numpy(), np.zeros((np.asarray(x).shape[0],))
