"""CPU tests of the native GA / NSGA-II driver (clip_glass_amd/search.py)."""
import types

import numpy as np

from clip_glass_amd import operators, search


class _Toy:
    """ZDT1-like two-objective problem / sphere single objective, pymoo-Problem shaped."""

    def __init__(self, n_var, n_obj):
        self.n_var, self.n_obj, self.xl, self.xu = n_var, n_obj, -10.0, 10.0
        self.config = types.SimpleNamespace(batch_size=4)
        self.calls = []

    def _evaluate(self, x, out, *a, **k):
        assert x.shape[0] % 4 == 0            # the engine's batch_size contract
        self.calls.append(x.shape[0])
        u = (x + 10) / 20
        if self.n_obj == 1:
            out["F"] = (x ** 2).sum(axis=1).astype(np.float32)
        else:
            g = 1 + 9 * u[:, 1:].mean(axis=1)
            out["F"] = np.column_stack([u[:, 0], g * (1 - np.sqrt(u[:, 0] / g))]).astype(np.float32)
        out["G"] = np.zeros(x.shape[0])


def test_non_dominated_sort_and_crowding():
    F = np.array([[1, 5], [2, 3], [3, 1], [2, 4], [4, 4], [3, 3]], float)
    fronts, rank = search.fast_non_dominated_sort(F)
    assert sorted(fronts[0].tolist()) == [0, 1, 2] and rank[3] == 1 and rank[5] == 1 and rank[4] == 2
    cd = search.crowding_distance(F[fronts[0]])
    assert np.isinf(cd).sum() == 2 and np.isfinite(cd).sum() == 1


def test_operators_respect_bounds_and_statistics():
    rng = np.random.default_rng(0)
    xl, xu = np.full(16, -10.0), np.full(16, 10.0)
    a, b = rng.normal(size=(500, 16)), rng.normal(size=(500, 16))
    ca, cb = search.sbx_vectorised(rng, a, b, xl, xu, eta=3.0)
    assert ca.min() >= -10 and cb.max() <= 10
    np.testing.assert_allclose((ca + cb).mean(), (a + b).mean(), atol=0.05)     # SBX preserves the parents' mean
    changed = (ca != a).mean()
    assert 0.4 < changed < 0.6                                                  # prob_per_variable 0.5
    m = search.polynomial_mutation(rng, a, xl, xu, eta=3.0, prob=0.5)
    assert m.min() >= -10 and m.max() <= 10 and 0.4 < (m != a).mean() < 0.6
    # scalar reference form of SBX agrees in distribution with the vectorised one
    c1, _ = search.sbx(np.random.default_rng(1), a[:50], b[:50], xl, xu)
    assert c1.shape == (50, 16) and c1.min() >= -10


def test_ga_and_nsga2_converge_on_toy_problems():
    ops = operators.get_operators(types.SimpleNamespace(config="StyleGAN2_ffhq_d"))
    p1 = _Toy(8, 1)
    r1 = search.minimize(p1, "ga", 16, 60, ops["sampling"], seed=3)
    # eta 3 / per-variable mutation prob 0.5 (the reference's settings) is a very disruptive operator set:
    # expect steady improvement, not fast convergence (random start: E[F] = 8)
    assert float(np.ravel(r1.F)[0]) < 3.0 and np.atleast_2d(r1.X).shape == (1, 8)
    p2 = _Toy(6, 2)
    r2 = search.minimize(p2, "nsga2", 16, 60, ops["sampling"], seed=3)
    assert r2.F.shape[1] == 2 and r2.F.shape[0] >= 4
    assert np.all(r2.F[:, 1] < 3.0)                     # pushed towards the g = 1 front
    assert all(c % 4 == 0 for c in p2.calls)
    i = search.pseudo_weights_choice(r2.F, [0, 1])
    assert 0 <= i < r2.F.shape[0] and r2.F[i, 1] == r2.F[:, 1].min()
