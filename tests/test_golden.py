"""Golden fixtures produced by the reference itself (tests/golden/make_golden.py, build
container) vs (a) the oracle on CPU and (b) the HIP engine on the GPU.  Fixtures hold
seeds + expected outputs only; weights / latents / noise are regenerated from the seeds."""
import os

import numpy as np
import pytest
import torch

import glass_models as M
from clip_glass_amd import synth
from oracle import fitness_ref
from util import check, diag

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return dict(np.load(os.path.join(GOLD, name), allow_pickle=False))


def _t(sd):
    return {k: torch.as_tensor(v) for k, v in sd.items()}


def _planes_fn(g, same_for_all):
    name = str(g["config"])
    c = M.CONFIGS[name]
    if same_for_all:   # G.static_noise(): one set of planes for every G call
        planes = synth.g_noise_planes(int(g["noise_seed"]), 0, 0, c["channels"])
        return lambda i: planes
    return lambda i: synth.g_noise_planes(int(g["noise_seed"]), int(g["generation"]), i, c["channels"])


def _oracle(g, target, same_noise):
    name = str(g["config"])
    c = M.CONFIGS[name]
    sd = M.make_state(name, int(g["seed"]))
    x = synth.latents(int(g["seed"]) + 1, int(g["P"]), c["latent"])
    detail = {}
    F, G = fitness_ref.evaluate(_t(sd), x, target, int(g["batch_size"]), True, _planes_fn(g, same_noise),
                                clip_size=c["clip"][4], detail=detail)
    return F, G, detail


def _engine(g, target, same_noise):
    name = str(g["config"])
    c = M.CONFIGS[name]
    P, bs = int(g["P"]), int(g["batch_size"])
    sd = M.make_state(name, int(g["seed"]))
    x = synth.latents(int(g["seed"]) + 1, P, c["latent"])
    fn = _planes_fn(g, same_noise)
    planes = [fn(i) for i in range(P // bs)]
    e = M.make_engine(name, sd, batch_size=bs, use_discriminator=True, max_pop=P, noise_mode=2)
    e.set_target(target)
    F = e.evaluate(x, noise=planes)
    det = e.details(P)
    e.close()
    return F, det


def _cmp_modules(tag, g, sim, dis, feats):
    rel = np.abs(sim - g["sim"]) / np.abs(g["sim"])
    diag("[golden] %s sim rel err %.3e (range %.3f..%.3f)" % (tag, rel.max(), g["sim"].min(), g["sim"].max()))
    assert rel.max() < 1e-3                                   # BASELINE.json north_star tolerance
    check(tag + " features", feats, g["features"], 5e-3)
    check(tag + " D logits", dis, g["dis"], 5e-3, atol=2e-3)


# ------------------------------- oracle (CPU) ---------------------------------------------
def test_oracle_reproduces_reference_problem_evaluate():
    """problem.GenerationProblem._evaluate (problem.py:14-29), mini architecture, P=8."""
    g = _load("mini_problem.npz")
    F, G, _ = _oracle(g, g["text_features"], same_noise=True)
    np.testing.assert_allclose(F, g["F"], rtol=2e-4, atol=2e-5)
    assert G.shape == g["G"].shape and not G.any()            # out["G"] = zeros(P) (problem.py:29)
    assert F.dtype == np.float32 and F.shape == (8, 2)


@pytest.mark.parametrize("fixture", ["mini_modules.npz", "mid_modules.npz", "ffhq_modules.npz"])
def test_oracle_reproduces_reference_modules(fixture):
    g = _load(fixture)
    F, _, d = _oracle(g, g["target"], same_noise=False)
    np.testing.assert_allclose(-F[:, 0], g["sim"], rtol=1e-4)
    np.testing.assert_allclose(d["features"].numpy(), g["features"], rtol=2e-3, atol=2e-4 * np.abs(g["features"]).max())
    np.testing.assert_allclose(d["dis"].numpy()[:, 0], g["dis"], rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(F[:, 1], g["hinge"], rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(d["image"].mean(dim=(1, 2, 3)).numpy(), g["image_mean"], rtol=1e-4)
    small = fitness_ref.resize224(d["image"][:2], M.CONFIGS[str(g["config"])]["clip"][4]).numpy()
    np.testing.assert_allclose(small, g["image_small"].astype(np.float32), atol=2e-3)


def test_clip_tokenizer_known_answer_in_fixture():
    g = _load("mini_problem.npz")
    assert g["tokens"][:12].tolist() == [49406, 320, 5916, 536, 930, 593, 518, 3293, 530, 518, 5994, 49407]


# ------------------------------- HIP engine (GPU) -----------------------------------------
@pytest.mark.gpu
def test_engine_matches_reference_problem_evaluate():
    g = _load("mini_problem.npz")
    F, det = _engine(g, g["text_features"], same_noise=True)
    rel = np.abs(F[:, 0] - g["F"][:, 0]) / np.abs(g["F"][:, 0])
    diag("[golden] mini_problem F[:,0] = -sim in [%.4f, %.4f], max rel err %.3e; F[:,1] max abs err %.3e"
         % (g["F"][:, 0].min(), g["F"][:, 0].max(), rel.max(), np.abs(F[:, 1] - g["F"][:, 1]).max()))
    # real text feature vs random-image features: sims are O(0.01-0.1); the relative bar applies where |sim| is not ~0
    assert np.all(np.abs(F[:, 0] - g["F"][:, 0]) < 1e-3 * np.maximum(np.abs(g["F"][:, 0]), 0.05))
    check("golden mini_problem hinge", F[:, 1], g["F"][:, 1], 5e-3, atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["mini_modules.npz", "mid_modules.npz", "ffhq_modules.npz"])
def test_engine_matches_reference_modules(fixture):
    g = _load(fixture)
    F, det = _engine(g, g["target"], same_noise=False)
    _cmp_modules("golden " + fixture, g, det["sim"], det["dis"], det["features"])
    check("golden %s hinge" % fixture, F[:, 1], g["hinge"], 5e-3, atol=2e-3)
