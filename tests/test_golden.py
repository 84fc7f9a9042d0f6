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
from util import check, check_logits, diag

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


def _oracle(g, target, same_noise, use_d=True):
    name = str(g["config"])
    c = M.CONFIGS[name]
    sd = M.make_state(name, int(g["seed"]), with_d=use_d)
    x = synth.latents(int(g["seed"]) + 1, int(g["P"]), c["latent"])
    detail = {}
    F, G = fitness_ref.evaluate(_t(sd), x, target, int(g["batch_size"]), use_d, _planes_fn(g, same_noise),
                                clip_size=c["clip"][4], detail=detail)
    return F, G, detail


def _engine(g, targets, same_noise, use_d=True):
    """One engine, one evaluate per target (a list) or a single target; returns (F or [F...], details of the last call)."""
    name = str(g["config"])
    c = M.CONFIGS[name]
    P, bs = int(g["P"]), int(g["batch_size"])
    sd = M.make_state(name, int(g["seed"]), with_d=use_d)
    x = synth.latents(int(g["seed"]) + 1, P, c["latent"])
    fn = _planes_fn(g, same_noise)
    planes = [fn(i) for i in range(P // bs)]
    e = M.make_engine(name, sd, batch_size=bs, use_discriminator=use_d, max_pop=P, noise_mode=2)
    many = isinstance(targets, list)
    Fs = []
    for target in (targets if many else [targets]):
        e.set_target(target)
        Fs.append(e.evaluate(x, noise=planes))
    det = e.details(P)
    e.close()
    return (Fs if many else Fs[0]), det


def _cmp_modules(tag, g, sim, dis, feats):
    rel = np.abs(sim - g["sim"]) / np.abs(g["sim"])
    diag("[golden] %s sim rel err %.3e (range %.3f..%.3f)" % (tag, rel.max(), g["sim"].min(), g["sim"].max()))
    assert rel.max() < 1e-3                                   # BASELINE.json north_star tolerance
    check(tag + " features", feats, g["features"], 5e-3)
    check_logits(tag + " D logits", dis, g["dis"], case=str(g["config"]))


# ------------------------------- oracle (CPU) ---------------------------------------------
def test_oracle_reproduces_reference_problem_evaluate():
    """problem.GenerationProblem._evaluate (problem.py:14-29), mini architecture, P=8."""
    g = _load("mini_problem.npz")
    F, G, _ = _oracle(g, g["text_features"], same_noise=True)
    np.testing.assert_allclose(F, g["F"], rtol=2e-4, atol=2e-5)
    assert G.shape == g["G"].shape and not G.any()            # out["G"] = zeros(P) (problem.py:29)
    assert F.dtype == np.float32 and F.shape == (8, 2)


def test_oracle_reproduces_reference_problem_evaluate_full_size_nod():
    """BASELINE.json configs[0]: StyleGAN2_ffhq_nod (config.py:136-155; n_obj = 1, no discriminator, problem.py:26-27), pop = 8,
    through the reference's own problem.py at the true 1024 px / ViT-B/32 size: F with the real text feature and F with a
    crafted target that puts the similarities at ~0.85."""
    g = _load("ffhq_nod_problem.npz")
    assert str(g["reference_config"]) == "StyleGAN2_ffhq_nod" and g["F"].shape == (8,)
    F, G, _ = _oracle(g, g["text_features"], same_noise=True, use_d=False)
    np.testing.assert_allclose(F, g["F"], rtol=2e-4, atol=2e-5)
    assert F.shape == (8,) and G.shape == g["G"].shape and not G.any()
    F2, _, d = _oracle(g, g["target"], same_noise=True, use_d=False)
    np.testing.assert_allclose(F2, g["F_target"], rtol=1e-4)
    np.testing.assert_allclose(d["features"].numpy(), g["features"], rtol=2e-3, atol=2e-4 * np.abs(g["features"]).max())


MODULE_FIXTURES = ["mini_modules.npz", "mid_modules.npz", "ffhq_modules.npz", "church_modules.npz", "car_modules.npz"]


@pytest.mark.parametrize("fixture", MODULE_FIXTURES)
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


TEXT_VISUAL = (128, 2, 2, 8, 32, 512)        # make_golden.TEXT_CASE_VISUAL as an engine `clip` tuple (width, layers, heads, patch, res, embed)


def _text_state(seed):
    sd = synth.make_state(synth.clip_visual_spec(TEXT_VISUAL[0], TEXT_VISUAL[1], TEXT_VISUAL[3], TEXT_VISUAL[4], TEXT_VISUAL[5]), seed)
    sd.update(synth.make_state(synth.clip_text_spec(), seed))
    return sd


def test_oracle_reproduces_reference_text_tower_full_size():
    """CLIP.encode_text (clip/model.py:307-320) at the real ViT-B/32 text geometry (width 512, 12 layers, 8 heads, ctx 77): the
    oracle's restatement vs features the reference's own build_model(...).encode_text produced for eight rows tokenized by the
    reference's clip.tokenize (tests/golden/clip_text_full.npz) — config C5's in-loop CLIP leg (generator.py:52-59)."""
    from oracle import clip_ref
    g = _load("clip_text_full.npz")
    assert g["tokens"].shape == (8, 77) and g["features"].shape == (8, 512)
    assert g["tokens"][0, :12].tolist() == [49406, 320, 5916, 536, 930, 593, 518, 3293, 530, 518, 5994, 49407]
    ref = clip_ref.encode_text(_t(_text_state(int(g["seed"]))), torch.tensor(g["tokens"])).numpy()
    np.testing.assert_allclose(ref, g["features"], rtol=2e-3, atol=2e-4 * np.abs(g["features"]).max())


def test_clip_tokenizer_reproduces_text_fixture_tokens():
    """Own BPE (clip_glass_amd/tokenizer.py) on the fixture's eight texts == the ids the reference tokenizer wrote into it."""
    bpe = "/root/reference/assets/bpe_simple_vocab_16e6.txt.gz"
    if not os.path.exists(bpe):
        pytest.skip("reference BPE asset not present")
    from clip_glass_amd.tokenizer import ClipTokenizer
    g = _load("clip_text_full.npz")
    ids = ClipTokenizer(bpe).tokenize([str(t) for t in g["texts"]])
    np.testing.assert_array_equal(ids, g["tokens"])


# ------------------------------- HIP engine (GPU) -----------------------------------------
@pytest.mark.gpu
def test_engine_text_tower_full_size_matches_reference_fixture():
    """glass_engine_encode_text at the real text geometry (512 x 12 x 8 heads, ctx 77) vs the reference-generated fixture and
    the oracle, at 8 rows and at the C5 population (P = 64 = the 8 rows repeated in a scrambled order: every row must come out
    bitwise as in the 8-row call), then Generator.clip_similarity_texts (generator.py:52-59) over those 64 rows against the
    cosine of the REFERENCE's features: 1e-3 relative (north_star)."""
    from clip_glass_amd.engine import Engine
    from clip_glass_amd.generator import Generator
    from oracle import clip_ref
    g = _load("clip_text_full.npz")
    sd = _text_state(int(g["seed"]))
    e = Engine([], latent_size=4, mapping_layers=0, batch_size=1, use_discriminator=False, n_obj=1, max_pop=64, clip=TEXT_VISUAL,
               noise_mode=0)
    e.load_state(sd)
    e.finalize()
    tokens = g["tokens"].astype(np.int64)
    got8 = e.encode_text(tokens)
    check("text tower (512 x 12, ctx 77) vs reference fixture", got8, g["features"], 3e-3)
    ora = clip_ref.encode_text(_t(sd), torch.tensor(tokens)).numpy()
    check("text tower (512 x 12, ctx 77) vs oracle", got8, ora, 3e-3)
    order = np.random.RandomState(3).permutation(64) % 8
    got64 = e.encode_text(tokens[order])
    np.testing.assert_array_equal(got64, got8[order])
    # the img2txt scoring leg at P = 64 through the host mirror: a stub tokenizer hands the fixture's ids over
    img = synth.make_target(g["features"])              # an "image feature" close to the population's text features
    gen = Generator.__new__(Generator)
    gen.engine, gen.image_features = e, img[None].astype(np.float32)
    gen.tokenizer = type("Tok", (), {"tokenize": staticmethod(lambda texts: tokens[order])})()
    sim = gen.clip_similarity_texts(["row %d" % i for i in order])
    rf = g["features"][order].astype(np.float64)
    ref_sim = (rf @ img.astype(np.float64)) / np.maximum(np.linalg.norm(rf, axis=1) * np.linalg.norm(img.astype(np.float64)), 1e-8)
    rel = np.abs(sim - ref_sim) / np.abs(ref_sim)
    diag("[golden] full-size text tower P=64: sim rel err vs reference features %.3e (sims %.3f..%.3f)" % (rel.max(), ref_sim.min(), ref_sim.max()))
    assert sim.shape == (64,) and rel.max() < 1e-3
    e.close()


@pytest.mark.gpu
def test_engine_matches_reference_problem_evaluate():
    g = _load("mini_problem.npz")
    F, det = _engine(g, g["text_features"], same_noise=True)
    rel = np.abs(F[:, 0] - g["F"][:, 0]) / np.abs(g["F"][:, 0])
    diag("[golden] mini_problem F[:,0] = -sim in [%.4f, %.4f], max rel err %.3e; F[:,1] max abs err %.3e"
         % (g["F"][:, 0].min(), g["F"][:, 0].max(), rel.max(), np.abs(F[:, 1] - g["F"][:, 1]).max()))
    # real text feature vs random-image features: sims are O(0.01-0.1); the relative bar applies where |sim| is not ~0
    assert np.all(np.abs(F[:, 0] - g["F"][:, 0]) < 1e-3 * np.maximum(np.abs(g["F"][:, 0]), 0.05))
    check_logits("golden mini_problem hinge", F[:, 1], g["F"][:, 1], case="mini")


@pytest.mark.gpu
def test_engine_matches_reference_problem_evaluate_full_size_nod():
    """Config C1 on the HIP path: the 1024 px `_nod` engine (no D buffers, n_obj = 1) vs the reference's own _evaluate."""
    g = _load("ffhq_nod_problem.npz")
    (F_text, F_tgt), det = _engine(g, [g["text_features"], g["target"]], same_noise=True, use_d=False)
    assert F_text.shape == (8, 1) and F_tgt.shape == (8, 1)
    rel = np.abs(F_tgt[:, 0] - g["F_target"]) / np.abs(g["F_target"])
    diag("[golden] ffhq_nod_problem: -sim (crafted target) in [%.4f, %.4f] max rel err %.3e; real text feature |F| ~ %.3f max abs err %.3e"
         % (g["F_target"].min(), g["F_target"].max(), rel.max(), np.abs(g["F"]).mean(), np.abs(F_text[:, 0] - g["F"]).max()))
    assert rel.max() < 1e-3                                   # BASELINE.json north_star tolerance
    assert np.all(np.abs(F_text[:, 0] - g["F"]) < 1e-3 * np.maximum(np.abs(g["F"]), 0.05))
    check("golden ffhq_nod features", det["features"], g["features"], 5e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", MODULE_FIXTURES)
def test_engine_matches_reference_modules(fixture):
    g = _load(fixture)
    F, det = _engine(g, g["target"], same_noise=False)
    _cmp_modules("golden " + fixture, g, det["sim"], det["dis"], det["features"])
    check_logits("golden %s hinge" % fixture, F[:, 1], g["hinge"], case=str(g["config"]))
