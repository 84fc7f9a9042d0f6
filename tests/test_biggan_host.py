"""CPU checks of the BigGAN-deep restatement (oracle/biggan_ref.py, synth.biggan_spec) and its host plumbing.

The network's source (pytorch-pretrained-biggan==0.1.1) is absent from /root/reference, so the forward
arithmetic is "parity unpinned"; what CAN be pinned on CPU is pinned here:
  * the layer shapes against the parameter counts the package publishes for its three checkpoints,
  * the spectral-norm inference rule against torch.nn.utils.spectral_norm itself,
  * the reference-side call sites that are present (latent.py:16-24).
"""
import numpy as np
import torch

from clip_glass_amd import synth
from oracle import biggan_ref as bg


def test_parameter_counts_match_the_published_checkpoints():
    # pytorch-pretrained-biggan README: BigGAN-deep-128 50.4M, -256 55.9M, -512 56.2M parameters
    for res, published in ((128, 50.4e6), (256, 55.9e6), (512, 56.2e6)):
        spec = synth.biggan_spec(synth.BIGGAN_LAYERS[res])
        n = sum(int(np.prod(s)) for name, s, _ in spec if "running_" not in name)
        assert abs(n - published) < 0.05e6, (res, n)
    assert synth.BIGGAN_LAYERS == bg.LAYERS


def test_spectral_norm_rule_matches_torch():
    torch.manual_seed(0)
    conv = torch.nn.utils.spectral_norm(torch.nn.Conv2d(12, 20, 3, padding=1), eps=1e-4)
    conv.train()
    for _ in range(3):
        conv(torch.randn(2, 12, 5, 5))          # power iterations update u, v
    conv.eval()
    sd = {"c." + k: v.detach().clone() for k, v in conv.state_dict().items()}
    x = torch.randn(2, 12, 5, 5)
    with torch.no_grad():
        want = conv(x)
        got = bg.snconv(sd, "c", x, padding=1)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-6)


def test_synthetic_state_has_converged_singular_vectors():
    spec = synth.biggan_spec([(0, 16, 16), (1, 16, 8), (1, 8, 4), (1, 4, 2), (1, 2, 1)], attention_pos=2, ch=32, z_dim=8,
                             num_classes=12)
    sd = synth.make_biggan_state(spec, 3)
    w = bg.sn_weight(sd, "biggan.generator.layers.0.conv_1").flatten(1).numpy()
    s = np.linalg.svd(w, compute_uv=False)[0]
    assert 0.95 < s < 1.1, s                    # weight_orig / sigma has spectral norm ~1, as in a trained checkpoint


def test_latent_forward_and_stat_row():
    x = synth.biggan_population(0, 4, 8, 12)
    x[0, :8] = 5.0
    z, c = bg.latent_forward(x, 8)
    assert float(z.max()) == 2.0 and np.allclose(c.sum(1).numpy(), 1.0, atol=1e-6)
    stats = torch.arange(51 * 3, dtype=torch.float32).view(51, 3)
    assert torch.equal(bg.stat_row(stats, 1.0, 51), stats[50])
    r = bg.stat_row(stats, 0.41, 51)            # between rows 20 and 21: rows[20]*coef + rows[21]*(1-coef)
    assert torch.all(r >= stats[20]) and torch.all(r <= stats[21])


def test_host_mirror_biggan_model_and_operators(monkeypatch):
    import types
    from clip_glass_amd import config as gconfig, operators, search
    from clip_glass_amd.latent import DeepMindBigGANLatentSpace
    from clip_glass_amd.models import DeepMindBigGAN
    cfg = types.SimpleNamespace(config="DeepMindBigGAN512", **gconfig.get_config("DeepMindBigGAN512"))
    cfg.weights = "synthetic:0"
    cfg.biggan_geometry = dict(layers=[(0, 16, 16), (1, 16, 8), (1, 8, 4), (1, 4, 2), (1, 2, 1)], attention_pos=2, ch=32)
    cfg.dim_z, cfg.num_classes = 8, 12
    m = DeepMindBigGAN(cfg)
    assert m.geometry["truncation"] == 1.0 and not m.has_discriminator()
    assert "biggan.generator.gen_z.weight_u" in m.state
    ls = DeepMindBigGANLatentSpace(cfg)
    x = synth.biggan_population(1, 6, 8, 12)
    ls.set_from_population(x)
    z, c = ls()
    zo, co = bg.latent_forward(x, 8)
    np.testing.assert_allclose(z, zo.numpy()); np.testing.assert_allclose(c, co.numpy(), atol=1e-7)
    assert ls.population().shape == (6, 20) and ls.population().dtype == np.float32
    # operators: mixed real / bool variables handled by the native driver when pymoo is absent
    ops = operators.get_operators(cfg)
    assert list(ops["mask"]) == ["real"] * 8 + ["bool"] * 12

    class Toy:
        n_var, n_obj, xl, xu = 20, 1, -2, 2
        config = types.SimpleNamespace(batch_size=4)

        def _evaluate(self, X, out):
            out["F"] = ((X[:, :8] - 1.0) ** 2).sum(1) + (X[:, 8:] != (np.arange(12) % 2)).sum(1)
    if not operators.HAVE_PYMOO:
        X0 = ops["sampling"]._do(Toy, 64)
        assert X0.shape == (64, 20) and np.abs(X0[:, :8]).max() <= 2 and set(np.unique(X0[:, 8:])) <= {0.0, 1.0}
        res = search.minimize(Toy(), "ga", 32, 60, ops["sampling"], seed=3, mask=ops["mask"])
        assert set(np.unique(res.X[8:])) <= {0.0, 1.0}
        assert float(np.ravel(res.F)[0]) < 3.0, res.F


def test_hux_and_bitflip_statistics():
    from clip_glass_amd import search
    rng = np.random.default_rng(0)
    a = (rng.random((400, 200)) < 0.5).astype(float)
    b = (rng.random((400, 200)) < 0.5).astype(float)
    ca, cb = search.hux(rng, a, b, prob=1.0)
    diff = a != b
    assert np.array_equal(ca[~diff], a[~diff]) and np.array_equal(ca + cb, a + b)      # bits only exchanged
    frac = ((ca != a) & diff).sum() / diff.sum()
    assert 0.48 < frac < 0.53, frac
    ca2, _ = search.hux(rng, a, b, prob=0.0)
    assert np.array_equal(ca2, a)
    f = search.bitflip(rng, a, 0.01)
    assert 0.005 < (f != a).mean() < 0.015
