"""Weight ingestion: the product's loaders read what the REFERENCE writes.

Fixtures under tests/golden/containers/ were written by the reference's own code (make_containers.py: Generator.save /
Discriminator.save, build_model(...).state_dict(), an HF-style GPT-2 bin); the loaders must return, tensor for tensor, the
synthetic state those containers were built from.  With /root/reference present the containers are also re-written
fresh and compared.  The -m gpu tests push the loaded tensors through the engine."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from clip_glass_amd import synth
from clip_glass_amd import generator as gen
from clip_glass_amd import models as gm
import glass_models as M

HERE = os.path.dirname(os.path.abspath(__file__))
CONT = os.path.join(HERE, "golden", "containers")
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_containers as mc      # noqa: E402  (constants + clip_mini_state; importing does not touch the reference)
import ref_harness as rh          # noqa: E402


def _same(got, want, keys):
    for k in keys:
        assert k in got, "loader dropped %s" % k
        assert got[k].dtype == np.float32 and got[k].shape == tuple(want[k].shape), k
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)


def _stylegan_cfg(path):
    return types.SimpleNamespace(weights=path, dim_z=M.CONFIGS["mini"]["latent"])


def test_stylegan2_container_fixture():
    """G.pth / D.pth written by stylegan2/models.py:111-132 (.save): nested G_mapping / G_synthesis containers, FIR and
    dlatent_avg buffers dropped, channel list recovered from the weight shapes."""
    c = M.CONFIGS["mini"]
    want = M.make_state("mini", 0)
    m = gm.StyleGAN2(_stylegan_cfg(os.path.join(CONT, "mini")))
    assert m.channels == list(c["channels"])
    sg_keys = [k for k in want if k.startswith(("G_mapping.", "G_synthesis.", "D."))]
    assert len(sg_keys) > 60
    _same(m.state, want, sg_keys)
    assert not [k for k in m.state if k.endswith("filter_kernel") or k.endswith("dlatent_avg")]
    g = torch.load(os.path.join(CONT, "mini", "G.pth"), map_location="cpu", weights_only=False)
    assert set(g) >= {"name", "kwargs", "state_dict", "G_mapping", "G_synthesis"}     # the reference's container layout


def test_stylegan2_missing_weights_exits(tmp_path, capsys):
    with pytest.raises(SystemExit):                                                   # models.py:93-101
        gm.StyleGAN2(_stylegan_cfg(str(tmp_path)))
    assert "Weights not found" in capsys.readouterr().out


def test_clip_checkpoint_fixture_two_towers():
    """ViT state dict with BOTH towers (what clip.load(...).state_dict() holds): the visual geometry must count only
    `visual.` resblocks (clip/model.py:368), every weight of both towers must arrive, fp16 storage -> fp32 values."""
    c = M.CONFIGS["mini"]
    sd = torch.load(os.path.join(CONT, "clip_mini.pt"), map_location="cpu")
    assert sd["visual.conv1.weight"].dtype == torch.float16 and "input_resolution" in sd
    want = mc.clip_mini_state()
    state = gen.clip_state_from_checkpoint(sd, with_text=True)
    keys = [k for k in want if k != "clip.logit_scale"]
    _same(state, want, keys)
    assert set(state) == set(keys)
    assert gen.clip_geometry_from_state(state) == tuple(c["clip"])
    n_text = len([k for k in state if k.startswith("clip.transformer.") and k.endswith(".attn.in_proj_weight")])
    assert n_text == mc.CLIP_TEXT["layers"] and c["clip"][1] == 2        # 2 + 2 blocks in the file, 2 reported
    vis = gen.clip_state_from_checkpoint(sd, with_text=False)
    assert vis and all(k.startswith("clip.visual.") for k in vis)
    cfg = types.SimpleNamespace(clip_weights=os.path.join(CONT, "clip_mini.pt"))
    st2, geom = gen._load_clip_state(cfg, True)
    assert geom == tuple(c["clip"]) and set(st2) == set(keys)


def test_clip_weights_must_be_given():
    with pytest.raises(RuntimeError, match="clip_weights"):
        gen._load_clip_state(types.SimpleNamespace(), True)


def test_gpt2_bin_fixture_remap():
    """gpt2/utils.py:10-51: .g/.b -> .weight/.bias, `transformer.` prefix added, mask buffers h.N.attn.bias dropped —
    and h.N.attn.c_attn.bias (the QKV bias) kept."""
    sd = torch.load(os.path.join(CONT, "gpt2_mini.bin"), map_location="cpu")
    assert "h.0.attn.bias" in sd and "h.0.ln_1.g" in sd and "wte.weight" in sd and not any(k.startswith("transformer.") for k in sd)
    want = synth.make_state(synth.gpt2_spec(**mc.GPT2), 2)
    state = gm.remap_gpt2_state(sd)
    _same(state, want, list(want))
    assert set(state) == set(want)
    for i in range(mc.GPT2["n_layer"]):
        assert "gpt2.transformer.h.%d.attn.c_attn.bias" % i in state
        assert "gpt2.transformer.h.%d.attn.bias" % i not in state
    # already-prefixed keys (a state_dict saved from the reference model) give the same result
    again = gm.remap_gpt2_state({"transformer." + k if not k.startswith("lm_head") else k: v for k, v in sd.items()})
    assert set(again) == set(want)


@pytest.mark.skipif(not rh.available(), reason="needs /root/reference (build container)")
def test_containers_fresh_from_reference(tmp_path):
    """The same loaders on containers written by the reference in THIS process (not the committed copies)."""
    c = M.CONFIGS["mini"]
    want = M.make_state("mini", 1)
    rh.build_ref_G(want, c["channels"], c["latent"], c["mapping"]).save(str(tmp_path / "G.pth"))
    rh.build_ref_D(want, c["channels"]).save(str(tmp_path / "D.pth"))
    m = gm.StyleGAN2(_stylegan_cfg(str(tmp_path)))
    _same(m.state, want, [k for k in want if k.startswith(("G_mapping.", "G_synthesis.", "D."))])
    model = rh.build_ref_clip(mc.clip_mini_state(3), fp32=False)
    state = gen.clip_state_from_checkpoint(model.state_dict(), with_text=True)
    w3 = mc.clip_mini_state(3)
    _same(state, w3, [k for k in w3 if k != "clip.logit_scale"])
    # the reference's own GPT-2 loader accepts the committed bin: same tensors land in its model (gpt2/utils.py load_weight)
    import importlib
    cwd = os.getcwd()
    os.chdir(rh.REF)
    try:
        gu = importlib.import_module("gpt2.utils")
    finally:
        os.chdir(cwd)
    g = synth.make_state(synth.gpt2_spec(**mc.GPT2), 0)
    ref_model, _ = rh.build_ref_gpt2(g, mc.GPT2["n_embd"], mc.GPT2["n_layer"], mc.GPT2["vocab"])
    ref_model = gu.load_weight(ref_model, torch.load(os.path.join(CONT, "gpt2_mini.bin"), map_location="cpu"))
    ours = gm.remap_gpt2_state(torch.load(os.path.join(CONT, "gpt2_mini.bin"), map_location="cpu"))
    for k, v in ref_model.state_dict().items():
        if k.startswith("lm_head") or k.endswith(".attn.bias"):
            continue
        np.testing.assert_array_equal(ours["gpt2." + k], v.numpy(), err_msg=k)


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_engine_from_containers_matches_synthetic_state():
    """GPU box: the container-loaded tensors drive the engine to the same F as the synthetic state they came from; the
    two-tower CLIP checkpoint finalizes (visual layer count = 2, not 4) and its text tower matches the oracle."""
    from oracle import clip_ref
    c = M.CONFIGS["mini"]
    m = gm.StyleGAN2(_stylegan_cfg(os.path.join(CONT, "mini")))
    clip_state, geom = gen._load_clip_state(types.SimpleNamespace(clip_weights=os.path.join(CONT, "clip_mini.pt")), True)
    from clip_glass_amd.engine import Engine
    x = synth.latents(5, 8, c["latent"])
    Fs = []
    for state in (dict(m.state, **clip_state), dict(M.make_state("mini", 0), **mc.clip_mini_state())):
        e = Engine(m.channels[::-1], latent_size=c["latent"], mapping_layers=c["mapping"], batch_size=4, use_discriminator=True,
                   n_obj=2, max_pop=8, clip=geom, noise_mode=1, noise_seed=4)
        e.load_state({k: v for k, v in state.items() if k != "clip.logit_scale"})
        e.finalize()
        rs = np.random.RandomState(0)
        tokens = np.zeros((3, mc.CLIP_TEXT["ctx"]), np.int64)
        for n in range(3):
            L = 4 + 3 * n
            tokens[n, :L] = rs.randint(1, mc.CLIP_TEXT["vocab"] - 1, size=L)
            tokens[n, L] = mc.CLIP_TEXT["vocab"] - 1                   # EOT = highest id (clip/model.py:318)
        tf = e.encode_text(tokens)
        e.set_target(tf[0])
        Fs.append(e.evaluate(x, generation=1))
        e.close()
    np.testing.assert_array_equal(Fs[0], Fs[1])
    tsd = {k: torch.as_tensor(v) for k, v in mc.clip_mini_state().items()}
    ref = clip_ref.encode_text(tsd, torch.tensor(tokens), heads=1).numpy()
    np.testing.assert_allclose(tf, ref, rtol=5e-3, atol=5e-3 * np.abs(ref).max())


@pytest.mark.gpu
def test_gpt2_bin_decodes_like_oracle():
    from oracle import gpt2_ref
    from clip_glass_amd.engine import Engine
    state = gm.remap_gpt2_state(torch.load(os.path.join(CONT, "gpt2_mini.bin"), map_location="cpu"))
    clipc = M.CONFIGS["mini"]["clip"]
    state.update(synth.make_state(synth.clip_visual_spec(clipc[0], clipc[1], clipc[3], clipc[4], clipc[5]), 0))
    e = Engine([], latent_size=4, mapping_layers=0, batch_size=1, use_discriminator=False, n_obj=1, max_pop=8, clip=clipc,
               noise_mode=0)
    e.load_state(state)
    e.finalize()
    ctx = np.random.RandomState(3).randint(0, mc.GPT2["vocab"], size=(8, 23)).astype(np.int64)
    got = e.gpt2_decode(ctx, 12)
    e.close()
    tsd = {k: torch.as_tensor(v) for k, v in state.items() if k.startswith("gpt2.")}
    want = gpt2_ref.sample_sequence(tsd, torch.tensor(ctx), 12)
    np.testing.assert_array_equal(np.asarray(got), np.asarray(want))
