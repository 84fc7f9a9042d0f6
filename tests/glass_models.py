"""Model configurations used by the parity tests (test infrastructure).

`channels` is in the reference's G order (last layer -> first layer, stylegan2/models.py:655-660);
the engine takes them LOW -> HIGH resolution, i.e. reversed."""
import numpy as np

from clip_glass_amd import synth

CONFIGS = {
    # 32 px, everything tiny: runs through the reference/oracle in well under a second
    "mini": dict(channels=[16, 16, 32, 32], latent=32, mapping=2, clip=(64, 2, 1, 8, 32, 32)),
    # 64 px, channel counts that reach the LDS-tiled kernels (W >= 32), resize 64 -> 32
    "mid": dict(channels=[32, 64, 64, 64, 64], latent=64, mapping=3, clip=(128, 2, 2, 8, 32, 64)),
    # the real thing: StyleGAN2 ffhq config-f 1024 px + CLIP ViT-B/32
    "ffhq": dict(channels=synth.FFHQ_CHANNELS, latent=512, mapping=8, clip=(768, 12, 12, 32, 224, 512)),
    # config.py:96-135 (StyleGAN2_church_* 256 px, StyleGAN2_car_* 512 px): config-f channel tables of the smaller networks
    # (convert_from_tf.py:154-175 builds square networks; the car images are letter-boxed 512 x 384 content)
    "church": dict(channels=synth.FFHQ_CHANNELS[2:], latent=512, mapping=8, clip=(768, 12, 12, 32, 224, 512)),
    "car": dict(channels=synth.FFHQ_CHANNELS[1:], latent=512, mapping=8, clip=(768, 12, 12, 32, 224, 512)),
}


def make_state(name, seed=0, with_d=True):
    c = CONFIGS[name]
    sd = synth.make_state(synth.stylegan2_g_spec(c["channels"], c["latent"], c["mapping"]), seed)
    if with_d:
        sd.update(synth.make_state(synth.stylegan2_d_spec(c["channels"]), seed))
    w, layers, heads, patch, res, emb = c["clip"]
    sd.update(synth.make_state(synth.clip_visual_spec(w, layers, patch, res, emb), seed))
    return sd


def make_engine(name, sd, *, batch_size=4, use_discriminator=True, max_pop=8, noise_mode=2, noise_seed=0, chunk=0, device=0):
    from clip_glass_amd.engine import Engine
    c = CONFIGS[name]
    e = Engine(c["channels"][::-1], latent_size=c["latent"], mapping_layers=c["mapping"], batch_size=batch_size,
               use_discriminator=use_discriminator, n_obj=2 if use_discriminator else 1, max_pop=max_pop,
               chunk=chunk, clip=c["clip"], noise_mode=noise_mode, noise_seed=noise_seed, device=device)
    e.load_state(sd)
    e.finalize()
    return e


def noise_planes(name, seed, generation, n_mb, first_mb=0):
    c = CONFIGS[name]
    return [synth.g_noise_planes(seed, generation, first_mb + m, c["channels"]) for m in range(n_mb)]


make_target = synth.make_target


# ---- BigGAN-deep (config C3) -------------------------------------------------------------------------
BIGGAN_CONFIGS = {
    # 64 px, 6 GenBlocks + attention at 16x16; CLIP-mini at 32 px (resize 64 -> 32)
    "bg_mini": dict(layers=[(0, 16, 16), (1, 16, 8), (1, 8, 4), (0, 4, 4), (1, 4, 2), (1, 2, 1)], attention_pos=3, ch=64,
                    z_dim=16, num_classes=24, clip=(128, 2, 2, 8, 32, 64)),
    # the released biggan-deep-256 geometry (config DeepMindBigGAN256) + CLIP ViT-B/32
    "bg256": dict(layers=synth.BIGGAN_LAYERS[256], attention_pos=8, ch=128, z_dim=128, num_classes=1000,
                  clip=(768, 12, 12, 32, 224, 512)),
    # the released biggan-deep-512 geometry + CLIP ViT-B/32
    "bg512": dict(layers=synth.BIGGAN_LAYERS[512], attention_pos=8, ch=128, z_dim=128, num_classes=1000,
                  clip=(768, 12, 12, 32, 224, 512)),
}


def make_biggan_state(name, seed=0):
    c = BIGGAN_CONFIGS[name]
    sd = synth.make_biggan_state(synth.biggan_spec(c["layers"], c["attention_pos"], c["ch"], c["z_dim"], c["num_classes"]), seed)
    w, layers, heads, patch, res, emb = c["clip"]
    sd.update(synth.make_state(synth.clip_visual_spec(w, layers, patch, res, emb), seed))
    return sd


def make_biggan_engine(name, sd, *, batch_size=4, max_pop=8, chunk=0, truncation=1.0):
    from clip_glass_amd.engine import Engine
    c = BIGGAN_CONFIGS[name]
    e = Engine([], batch_size=batch_size, max_pop=max_pop, chunk=chunk, clip=c["clip"],
               biggan=dict(layers=c["layers"], attention_pos=c["attention_pos"], ch=c["ch"], z_dim=c["z_dim"],
                           num_classes=c["num_classes"], truncation=truncation))
    e.load_state(sd)
    e.finalize()
    return e
