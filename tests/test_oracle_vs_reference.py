"""Pin the oracle against the reference itself, imported verbatim (build container only).

The reference has no tests/golden vectors; this is the pinning the tier
asks for: same synthetic weights into the reference's own nn.Modules and into
oracle/, same latents + same injected noise -> outputs must agree to fp32 rounding.
"""
import numpy as np
import pytest
import torch

import ref_harness as rh
from clip_glass_amd import synth
from oracle import clip_ref, fitness_ref, stylegan2_ref as sg

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not rh.available(), reason="/root/reference not present")]

MINI = dict(channels=[16, 16, 32, 32], latent=32, mapping=2)


def _t(sd):
    return {k: torch.as_tensor(v) for k, v in sd.items()}


def _mini_state(seed=3):
    sd = synth.make_state(synth.stylegan2_g_spec(MINI["channels"], MINI["latent"], MINI["mapping"]), seed)
    sd.update(synth.make_state(synth.stylegan2_d_spec(MINI["channels"]), seed))
    return sd


def test_generator_mini_matches_reference():
    sd = _mini_state()
    G = rh.build_ref_G(sd, MINI["channels"], MINI["latent"], MINI["mapping"])
    z = torch.tensor(synth.latents(1, 4, MINI["latent"])).float()
    planes = [torch.tensor(p) for p in synth.g_noise_planes(7, 0, 0, MINI["channels"])]
    with torch.no_grad():
        G(z)  # noise layers learn their shapes
        G.static_noise(noise_tensors=[p[None, None] for p in planes])
        ref = G(z)
        ora = sg.generator(_t(sd), z, planes)
    assert ref.shape == ora.shape == (4, 3, 32, 32)
    np.testing.assert_allclose(ora.numpy(), ref.numpy(), rtol=2e-4, atol=2e-4 * float(ref.abs().max()))


def test_discriminator_mini_matches_reference():
    sd = _mini_state()
    D = rh.build_ref_D(sd, MINI["channels"])
    img = torch.tensor(synth.normal(5, "img", (8, 3, 32, 32)))
    with torch.no_grad():
        ref = D(img)
        ora = sg.discriminator(_t(sd), img)
    assert ref.shape == ora.shape == (8, 1)
    np.testing.assert_allclose(ora.numpy(), ref.numpy(), rtol=2e-4, atol=1e-4)


def test_clip_mini_matches_reference():
    sd = synth.make_state(synth.clip_visual_spec(width=64, layers=2, patch=8, res=32, out_dim=32), 4)
    sd.update(synth.make_state(synth.clip_text_spec(width=64, layers=2, ctx=77, vocab=49408, out_dim=32), 4))
    model = rh.build_ref_clip(sd)
    img = torch.tensor(synth.normal(6, "img", (3, 3, 32, 32))).sigmoid()
    tok = rh.load_reference()["clip_clip"].tokenize(["a wolf at night with the moon in the background", "a cat"])
    with torch.no_grad():
        np.testing.assert_allclose(clip_ref.encode_image(_t(sd), img).numpy(),
                                   model.encode_image(img).numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(clip_ref.encode_text(_t(sd), tok).numpy(),
                                   model.encode_text(tok).numpy(), rtol=1e-4, atol=1e-5)


def test_tokenizer_known_answer():
    tok = rh.load_reference()["clip_clip"].tokenize(["a wolf at night with the moon in the background"])
    assert tok[0, :12].tolist() == [49406, 320, 5916, 536, 930, 593, 518, 3293, 530, 518, 5994, 49407]
