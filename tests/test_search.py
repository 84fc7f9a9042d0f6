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


# ---- known answers: the operators' published closed forms on INJECTED uniforms -------------------------------
# pymoo 0.4.2.1 is not in the image (parity of f2 is unpinned against pymoo itself), but SBX (Deb & Agrawal 1995; the
# bounded form of Deb's NSGA-II code that pymoo's real_sbx restates), polynomial mutation (Deb & Goyal 1996, bounded form), HUX,
# bit-flip, fast non-dominated sort + crowding distance (Deb et al. 2002) and the pseudo-weight decision rule (run.py:103-113) are
# closed forms.  The expected numbers below were worked out by hand from those formulas; `_FakeRng` feeds the operators the
# uniforms the hand calculation assumed, in the order the operator asks for them.
class _FakeRng:
    def __init__(self, randoms=(), perms=()):
        self.randoms, self.perms = list(randoms), list(perms)

    def random(self, shape=None):
        a = np.asarray(self.randoms.pop(0), float)
        if shape is not None:
            a = np.broadcast_to(a, shape if isinstance(shape, tuple) else (shape,)).copy()
        return a

    def permutation(self, x):
        p = np.asarray(self.perms.pop(0))
        x = np.arange(x) if np.isscalar(x) else np.asarray(x)
        assert sorted(p.tolist()) == sorted(x.tolist())
        return p


def test_sbx_known_answers():
    """eta = 3, parents (-1, 1), bounds +-10: beta = 1 + 2 * 9 / 2 = 10, alpha = 2 - 10^-4 = 1.9999.
    u = 0.25 <= 1/alpha: betaq = (0.25 * 1.9999)^(1/4) = 0.840886 -> children -+0.840886 (contracting);
    u = 0.75 >  1/alpha: betaq = (1 / (2 - 0.75 * 1.9999))^(1/4) = 1.189162 -> children -+1.189162 (expanding).
    Asymmetric bounds (xl = -1.5): child 1 uses beta = 1 + 2 * 0.5 / 2 = 1.5, alpha = 2 - 1.5^-4 = 1.802469;
    u = 0.9 > 1/alpha = 0.554795: betaq = (1 / (2 - 0.9 * 1.802469))^(1/4) = (1 / 0.377778)^(1/4) = 2.647059^(1/4) = 1.275531 -> c1 = -1.275531."""
    xl, xu = np.array([-10.0, -10.0, -1.5]), np.array([10.0, 10.0, 10.0])
    pa, pb = np.array([[-1.0, 1.0, -1.0]]), np.array([[1.0, -1.0, 1.0]])
    # vectorised form draws: u [n, nv], swap [n, nv], mating prob [n, 1], per-variable prob [n, nv]
    rng = _FakeRng(randoms=[[[0.25, 0.75, 0.9]], [[0.9, 0.9, 0.9]], [[0.0]], [[0.0, 0.0, 0.0]]])
    ca, cb = search.sbx_vectorised(rng, pa, pb, xl, xu, eta=3.0)
    np.testing.assert_allclose(ca[0], [-0.840886, -1.189162, -1.275531], atol=2e-6)      # (no swap: child 1 is the lower one)
    np.testing.assert_allclose(cb[0, :2], [0.840886, 1.189162], atol=2e-6)
    beta2 = 1 + 2 * 9 / 2.0
    alpha2 = 2 - beta2 ** -4
    np.testing.assert_allclose(cb[0, 2], 0.5 * (0 + (1 / (2 - 0.9 * alpha2)) ** 0.25 * 2), atol=1e-12)
    # swap draw < 0.5 exchanges the two children; per-variable draw >= 0.5 leaves the parents' values in place
    rng = _FakeRng(randoms=[[[0.25, 0.75, 0.9]], [[0.1, 0.9, 0.9]], [[0.0]], [[0.0, 0.7, 0.0]]])
    ca, cb = search.sbx_vectorised(rng, pa, pb, xl, xu, eta=3.0)
    np.testing.assert_allclose([ca[0, 0], cb[0, 0]], [0.840886, -0.840886], atol=2e-6)
    assert ca[0, 1] == pa[0, 1] and cb[0, 1] == pb[0, 1]
    # the scalar form (same formulas, one draw at a time) gives the same children for the same uniforms
    rng = _FakeRng(randoms=[[0.0], 0.0, 0.25, 0.9, 0.0, 0.75, 0.9, 0.0, 0.9, 0.9])
    sa, sb = search.sbx(rng, pa, pb, xl, xu, eta=3.0)
    np.testing.assert_allclose(sa[0], [-0.840886, -1.189162, -1.275531], atol=2e-6)
    # children of identical parents are the parents
    rng = _FakeRng(randoms=[[[0.3]], [[0.9]], [[0.0]], [[0.0]]])
    ca, cb = search.sbx_vectorised(rng, np.array([[2.0]]), np.array([[2.0]]), xl[:1], xu[:1])
    assert ca[0, 0] == 2.0 and cb[0, 0] == 2.0


def test_polynomial_mutation_known_answers():
    """eta = 3, x = 0 in [-10, 10]: delta1 = delta2 = 0.5, (1 - delta)^4 = 0.0625.
    u = 0.25: deltaq = (0.5 + 0.5 * 0.0625)^(1/4) - 1 = 0.53125^(1/4) - 1 = -0.146262 -> x' = -2.925235
    u = 0.50: deltaq = 0; u = 0.75: deltaq = +0.146262 -> x' = +2.925235.
    x = 9 (delta2 = 0.05, 0.95^4 = 0.81450625), u = 0.99: deltaq = 1 - (0.02 + 0.98 * 0.81450625)^(1/4) = 1 - 0.81821613^(1/4)
    = 1 - 0.951080 = 0.048920 -> x' = 9.978402; the per-variable draw >= prob keeps x."""
    xl, xu = np.full(5, -10.0), np.full(5, 10.0)
    x = np.array([[0.0, 0.0, 0.0, 9.0, 3.0]])
    rng = _FakeRng(randoms=[[[0.25, 0.5, 0.75, 0.99, 0.3]], [[0.1, 0.1, 0.1, 0.1, 0.6]]])
    y = search.polynomial_mutation(rng, x, xl, xu, eta=3.0, prob=0.5)
    np.testing.assert_allclose(y[0], [-2.925235, 0.0, 2.925235, 9.978402, 3.0], atol=3e-6)
    # int variant (GPT2 config, operators.py:69-70): mutated value rounded to the nearest integer inside the bounds
    ops = operators.get_operators(types.SimpleNamespace(config="GPT2", dim_z=2))
    assert ops["mask"] == ["int", "int"]
    rng = _FakeRng(randoms=[[[0.5, 0.5]], [[0.9, 0.9]], [[0.0]], [[0.9, 0.9]]], perms=[[0, 1]])
    off = search.vary(_Chain(rng, [[[0.25, 0.75]], [[0.0, 0.0]]]), np.array([[100.0, 100.0]]), np.array([[100.0, 100.0]]),
                      np.zeros(2), np.full(2, 50256.0), np.array(["int", "int"]), 3.0, 3.0, 0.5, 2)
    # x = 100 of [0, 50256]: delta1 = 100 / 50256; u = 0.25: deltaq = (0.5 + 0.5 (1 - d1)^4)^(1/4) - 1 = -0.000995... * 50256 = -50.03 -> 50
    d1 = 100 / 50256.0
    lo = (0.5 + 0.5 * (1 - d1) ** 4) ** 0.25 - 1
    hi = 1 - (0.5 + 0.5 * (1 - (1 - d1)) ** 4) ** 0.25
    np.testing.assert_array_equal(off[:, 0], np.rint(100 + lo * 50256))
    np.testing.assert_array_equal(off[:, 1], np.rint(100 + hi * 50256))
    assert off[0, 0] == 50.0 and off.dtype == float and np.all(off == np.rint(off))


class _Chain:
    """First rng's queue, then a second list of uniforms (vary() = crossover draws, permutation, mutation draws)."""

    def __init__(self, first, more):
        self.first, self.more = first, _FakeRng(randoms=more)

    def random(self, shape=None):
        return (self.first if self.first.randoms else self.more).random(shape)

    def permutation(self, x):
        return self.first.permutation(x)


def test_hux_and_bitflip_known_answers():
    """HUX (pymoo bin_hux): the parents differ in bits {0, 2, 3, 5, 6}: ceil(0.5 * 5) = 3 of them are exchanged — the first three
    of the (injected) permutation [5, 0, 6, 2, 3]; a mating whose draw >= prob is copied.  Bit-flip: bits with draw < 0.01 flip."""
    pa = np.array([[1, 1, 0, 0, 1, 1, 0, 1], [1, 0, 1, 0, 1, 0, 1, 0]], float)
    pb = np.array([[0, 1, 1, 1, 1, 0, 1, 1], [0, 1, 0, 1, 0, 1, 0, 1]], float)
    rng = _FakeRng(randoms=[[0.1, 0.9]], perms=[[5, 0, 6, 2, 3]])
    ca, cb = search.hux(rng, pa, pb, prob=0.2)
    np.testing.assert_array_equal(ca[0], [0, 1, 0, 0, 1, 0, 1, 1])     # bits 5, 0, 6 taken from the other parent
    np.testing.assert_array_equal(cb[0], [1, 1, 1, 1, 1, 1, 0, 1])
    np.testing.assert_array_equal(ca[1], pa[1])
    np.testing.assert_array_equal(cb[1], pb[1])
    rng = _FakeRng(randoms=[[[0.5, 0.009, 0.01, 0.0, 0.99, 0.5, 0.5, 0.5]]])
    np.testing.assert_array_equal(search.bitflip(rng, pa[:1], prob=0.01)[0], [1, 0, 0, 1, 1, 1, 0, 1])


def test_sort_crowding_survival_and_decision_known_answers():
    """Deb et al. 2002 on a hand-checked example.  Points (minimisation):
         a (0,10) b (1,5) c (4,4) d (8,1) e (10,0) | f (2,6) g (5,5) | h (9,9)
    front 0 = {a,b,c,d,e}; f is dominated by b only, g by c (and b): front 1 = {f,g}; h by everything: front 2.
    Crowding in front 0 (objective spans 10 and 10): a, e = inf; b = (4-0)/10 + (10-4)/10 = 1.0; c = (8-1)/10 + (5-1)/10 = 1.1;
    d = (10-4)/10 + (4-0)/10 = 1.0."""
    F = np.array([[0, 10], [1, 5], [4, 4], [8, 1], [10, 0], [2, 6], [5, 5], [9, 9]], float)
    fronts, rank = search.fast_non_dominated_sort(F)
    assert [sorted(f.tolist()) for f in fronts] == [[0, 1, 2, 3, 4], [5, 6], [7]]
    assert rank.tolist() == [0, 0, 0, 0, 0, 1, 1, 2]
    cd = search.crowding_distance(F[:5])
    assert np.isinf(cd[0]) and np.isinf(cd[4])
    np.testing.assert_allclose(cd[1:4], [1.0, 1.1, 1.0], atol=1e-12)
    assert np.isinf(search.crowding_distance(F[5:7])).all()
    # pseudo-weights (run.py:103-113, weights [0, 1]): w_i = ((f_i^max - f_i) / span_i) / sum; the point with pseudo-weight closest
    # to (0, 1) is the one with the best SECOND objective = e; with weights [1, 0] it is a
    assert search.pseudo_weights_choice(F[:5], [0, 1]) == 4 and search.pseudo_weights_choice(F[:5], [1, 0]) == 0
    # middle point c: pseudo-weights ((10-4)/10, (10-4)/10) / 1.2 = (0.5, 0.5)
    assert search.pseudo_weights_choice(F[:5], [0.5, 0.5]) == 2

    # rank-and-crowding survival to 4 of the 8 through the driver: front 0 is cut by crowding distance (a, e, then c)
    class P:
        n_var, n_obj, xl, xu = 2, 2, -100.0, 100.0

        def _evaluate(self, x, out, *a, **k):
            out["F"] = x.copy()

    class S:
        def _do(self, problem, n):
            return F.copy()
    res = search.minimize(P(), "nsga2", 8, 0, S(), seed=0)
    assert sorted(map(tuple, res.F.tolist())) == sorted(map(tuple, F[:5].tolist()))
