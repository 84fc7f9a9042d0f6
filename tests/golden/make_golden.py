"""Generate the committed golden fixtures by running the REFERENCE itself (imported from
/root/reference in the build container, see ref_harness.py).  Build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixtures hold inputs' seeds and expected outputs only (no reference code, no weights:
weights/latents/noise are regenerated from (seed, name) by clip_glass_amd/synth.py).

  mini_problem.npz : the WHOLE reference path — problem.GenerationProblem._evaluate
                     (problem.py:14-29) with config StyleGAN2_ffhq_d semantics on the
                     "mini" architecture, P=8, batch_size=4, static noise.
  mid_modules.npz  : G / D / CLIP modules called directly, 64 px "mid" architecture,
                     per-minibatch noise planes.
  ffhq_modules.npz : the true 1024 px config-f + ViT-B/32 architecture, P=4.
  clip_text_full.npz : CLIP.encode_text at the real text geometry (512 x 12 x 8 heads, ctx 77), 8 reference-tokenized rows.
"""
import argparse
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import ref_harness as rh  # noqa: E402
import glass_models as M  # noqa: E402
from clip_glass_amd import synth  # noqa: E402


def modules_case(name, P, bs, seed, noise_seed, generation):
    """Reference G/D/CLIP nn.Modules driven like models.py:108-129 + generator.py:29-60."""
    c = M.CONFIGS[name]
    sd = M.make_state(name, seed)
    full = name not in ("mini", "mid")
    sd.update(synth.make_state(synth.clip_text_spec(width=c["clip"][0] if not full else 512,
                                                    layers=2 if not full else 12, out_dim=c["clip"][5]), seed))
    G = rh.build_ref_G(sd, c["channels"], c["latent"], c["mapping"])
    D = rh.build_ref_D(sd, c["channels"])
    clip_model = rh.build_ref_clip(sd)
    x = synth.latents(seed + 1, P, c["latent"])
    z = torch.tensor(x.astype(float)).float()
    imgs, dis = [], []
    with torch.no_grad():
        G(z[:1])  # lets the noise layers learn their shapes
        for i in range(P // bs):
            planes = synth.g_noise_planes(noise_seed, generation, i, c["channels"])
            G.static_noise(noise_tensors=[torch.tensor(p)[None, None] for p in planes])
            imgs.append(G(z[i * bs:(i + 1) * bs]))
        img = torch.cat(imgs)
        img = ((img + 1) / 2.0).clip(0, 1)                                   # utils.py:14-17
        small = torch.nn.functional.interpolate(img, size=(c["clip"][4],) * 2, mode="bilinear", align_corners=False)
        feats = clip_model.encode_image(small)
        for i in range(P // bs):
            dis.append(D(img[i * bs:(i + 1) * bs] * 2 - 1))                  # utils.py:19-21
        dis = torch.cat(dis)
    target = synth.make_target(feats.numpy())
    sim = torch.cosine_similarity(feats, torch.tensor(target)[None]).numpy()
    return dict(config=name, P=P, batch_size=bs, seed=seed, noise_seed=noise_seed, generation=generation,
                target=target, features=feats.numpy(), sim=sim, dis=dis.numpy()[:, 0],
                hinge=np.maximum(1 - dis.numpy()[:, 0], 0),
                image_mean=img.mean(dim=(1, 2, 3)).numpy(), image_std=img.std(dim=(1, 2, 3)).numpy(),
                image_small=small[:2].numpy().astype(np.float16))


def problem_case(seed=0, noise_seed=5, name="mini", config_name="StyleGAN2_ffhq_d", P=8, crafted_target=False):
    """The reference's own problem.py / generator.py / models.py / latent.py driven end to end.
    crafted_target: after construction the text feature is replaced (attribute assignment on the reference's Generator object,
    generator.py:23-24) by a synthetic target built from the population's own image features, so that the similarities sit in
    ~[0.5, 0.9] where the 1e-3 RELATIVE bar is meaningful; the fixture then holds both F's."""
    bs = 4
    c = M.CONFIGS[name]
    full = name not in ("mini", "mid")
    sd = M.make_state(name, seed)
    sd.update(synth.make_state(synth.clip_text_spec(width=512 if full else 64, layers=12 if full else 2, out_dim=c["clip"][5]), seed))
    R = rh.load_reference()
    A = rh.load_author_modules()
    tmp = tempfile.mkdtemp(prefix="glass_golden_")
    rh.build_ref_G(sd, c["channels"], c["latent"], c["mapping"]).save(os.path.join(tmp, "G.pth"))
    rh.build_ref_D(sd, c["channels"]).save(os.path.join(tmp, "D.pth"))
    clip_model = rh.build_ref_clip(sd)
    R["clip_clip"].load = lambda *a, **k: (clip_model, None)                 # no download (clip.py:24-78)
    cfg = types.SimpleNamespace(device="cpu", config=config_name, target="a wolf at night with the moon in the background")
    vars(cfg).update(A["config"].get_config(config_name))
    cfg.weights = tmp
    cfg.dim_z = c["latent"]
    cfg.batch_size = bs
    cfg.problem_args = dict(cfg.problem_args, n_var=c["latent"], n_constr=c["latent"])
    prob = A["problem"].GenerationProblem(cfg)
    text_features = prob.generator.text_features.numpy()[0]
    # kornia.resize stub interpolates to (224,224) as generator.py:45 asks; the mini CLIP takes 32 px
    # images, so give the stub the model's resolution instead (the call site itself is unchanged)
    import kornia
    kornia.resize = lambda x, size: torch.nn.functional.interpolate(x, size=(c["clip"][4],) * 2, mode="bilinear", align_corners=False)
    planes = synth.g_noise_planes(noise_seed, 0, 0, c["channels"])
    with torch.no_grad():
        prob.generator.model.G(torch.zeros(1, c["latent"]))
        prob.generator.model.G.static_noise(noise_tensors=[torch.tensor(p)[None, None] for p in planes])
    x = synth.latents(seed + 1, P, c["latent"])
    out = {}
    prob._evaluate(x, out)
    res = dict(config=name, reference_config=config_name, P=P, batch_size=bs, seed=seed, noise_seed=noise_seed, text_features=text_features,
               F=np.asarray(out["F"], dtype=np.float32), G=np.asarray(out["G"]),
               tokens=R["clip_clip"].tokenize([cfg.target]).numpy()[0])
    if crafted_target:
        with torch.no_grad():
            ls = cfg.latent(cfg)
            ls.set_from_population(x)
            img = prob.generator.generate(ls, minibatch=bs)                     # generator.py:29-34
            feats = prob.generator.CLIP.encode_image(kornia.resize(img, (224, 224)))   # generator.py:45-49
        target = synth.make_target(feats.numpy())
        prob.generator.text_features = torch.tensor(target)[None]
        out2 = {}
        prob._evaluate(x, out2)
        res.update(target=target, F_target=np.asarray(out2["F"], dtype=np.float32), features=feats.numpy())
    return res


def gpt2_case(seed=2, P=8):
    """Reference GPT2LMHeadModel + sample_sequence (gpt2/sample.py) on a small synthetic GPT-2: greedy tokens."""
    geo = dict(n_embd=128, n_layer=2, vocab=2048)
    sd = synth.make_state(synth.gpt2_spec(**geo, n_positions=64), seed)
    model, sample_sequence = rh.build_ref_gpt2(sd, geo["n_embd"], geo["n_layer"], geo["vocab"])
    ctx = np.random.RandomState(1).randint(0, geo["vocab"], size=(P, 23)).astype(np.int64)
    out = sample_sequence(model=model, length=30, context=torch.tensor(ctx), start_token=None, batch_size=P,
                          temperature=0.7, top_k=40, device="cpu", sample=False)
    with torch.no_grad():
        logits, _ = model(torch.tensor(ctx))
    return dict(seed=seed, n_embd=geo["n_embd"], n_layer=geo["n_layer"], vocab=geo["vocab"], context=ctx,
                tokens=np.asarray(out, dtype=np.int64), last_logits=logits[:, -1, :64].numpy())


TEXT_CASE_VISUAL = (128, 2, 8, 32, 512)      # build_model needs a visual tower; the text fixture only uses a small one
TEXT_CASE_TEXTS = ["a wolf at night with the moon in the background",          # run.py:22 default target
                   "the picture of a dog sitting on the grass",               # img2txt-style outputs (config.py:27 init_text)
                   "the picture of", "a", "an astronaut riding a horse in the style of van gogh, highly detailed",
                   "two cats", "the picture of the picture of the picture of a a a", "moon"]


def text_case(seed=0):
    """CLIP.encode_text (clip/model.py:307-320: token + positional embedding, 12 causal blocks of width 512 / 8 heads,
    ln_final, EOT row @ text_projection) at the real ViT-B/32 text geometry, ctx 77, on eight rows tokenized by the
    reference's own clip.tokenize (clip.py:125-138) — the img2txt config's in-loop leg (generator.py:52-59)."""
    sd = synth.make_state(synth.clip_visual_spec(*TEXT_CASE_VISUAL), seed)
    sd.update(synth.make_state(synth.clip_text_spec(), seed))
    R = rh.load_reference()
    model = rh.build_ref_clip(sd)
    tokens = R["clip_clip"].tokenize(TEXT_CASE_TEXTS)
    with torch.no_grad():
        feats = model.encode_text(tokens)
    return dict(seed=seed, tokens=tokens.numpy().astype(np.int64), features=feats.numpy().astype(np.float32),
                texts=np.array(TEXT_CASE_TEXTS))


def r4_cases():
    np.savez_compressed(os.path.join(HERE, "clip_text_full.npz"), **text_case())
    print("clip_text_full.npz")


def r3_cases():
    # BASELINE.json configs[0]: StyleGAN2_ffhq_nod, pop = 8, through the reference's own problem.py at the true 1024 px size
    np.savez_compressed(os.path.join(HERE, "ffhq_nod_problem.npz"),
                        **problem_case(seed=0, noise_seed=5, name="ffhq", config_name="StyleGAN2_ffhq_nod", P=8, crafted_target=True))
    print("ffhq_nod_problem.npz")
    # config.py:96-135: the church (256 px) and car (512 px) config-f channel tables, P = 4 (one minibatch)
    np.savez_compressed(os.path.join(HERE, "church_modules.npz"), **modules_case("church", 4, 4, 0, 11, 2))
    print("church_modules.npz")
    np.savez_compressed(os.path.join(HERE, "car_modules.npz"), **modules_case("car", 4, 4, 0, 11, 2))
    print("car_modules.npz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-ffhq", action="store_true")
    ap.add_argument("--only-ffhq", action="store_true")
    ap.add_argument("--only-r4", action="store_true", help="round-4 addition only: full-size CLIP text tower case")
    ap.add_argument("--only-r3", action="store_true", help="round-3 additions only: full-size C1 problem case, church / car geometry")
    args = ap.parse_args()
    assert rh.available(), "needs /root/reference"
    if args.only_r4:
        r4_cases()
        return
    if args.only_r3:
        r3_cases()
        return
    if args.only_ffhq:
        np.savez_compressed(os.path.join(HERE, "ffhq_modules.npz"), **modules_case("ffhq", 8, 4, 0, 11, 2))
        print("ffhq_modules.npz")
        return
    np.savez_compressed(os.path.join(HERE, "mini_problem.npz"), **problem_case())
    print("mini_problem.npz")
    np.savez_compressed(os.path.join(HERE, "gpt2_mini.npz"), **gpt2_case())
    print("gpt2_mini.npz")
    np.savez_compressed(os.path.join(HERE, "mid_modules.npz"), **modules_case("mid", 8, 4, 0, 11, 2))
    print("mid_modules.npz")
    np.savez_compressed(os.path.join(HERE, "mini_modules.npz"), **modules_case("mini", 8, 4, 0, 11, 2))
    print("mini_modules.npz")
    r4_cases()
    if not args.skip_ffhq:
        r3_cases()
        # P = 8: two minibatches = two shared noise planes per layer and two mbstd groups (SURVEY 8a notes 4-5)
        np.savez_compressed(os.path.join(HERE, "ffhq_modules.npz"), **modules_case("ffhq", 8, 4, 0, 11, 2))
        print("ffhq_modules.npz")


if __name__ == "__main__":
    main()
