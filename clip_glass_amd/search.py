"""Native search driver: GA (single objective) and NSGA-II (two objectives) with the operator
set the reference configures through pymoo 0.4.2.1 (run.py:59-76, operators.py:37-81):
simulated binary crossover (eta 3, prob 1.0), polynomial mutation (eta 3, per-variable prob 0.5),
binary tournament mating, duplicate elimination, fitness / rank-and-crowding survival.

pymoo is not vendored by the reference and is not installable here (its 0.4.2.1 release breaks
on numpy >= 1.24), so this is a restatement of the published algorithms (Deb et al. 2002; Deb &
Agrawal 1995), statistically — not bit — equivalent to pymoo's (parity unpinned).
It lets `python -m clip_glass_amd.run` work end-to-end without pymoo and gives every rank of a
multi-GPU job the same population (same seed => identical GA state, no broadcast needed).
"""
import numpy as np


# ----------------------------- variation operators --------------------------------------
def sbx(rng, pa, pb, xl, xu, eta=3.0, prob=1.0, prob_var=0.5):
    """Bounded simulated binary crossover of parent matrices pa, pb [n, n_var] -> two children each."""
    n, nv = pa.shape
    ca, cb = pa.copy(), pb.copy()
    do_cross = rng.random(n) < prob
    for i in np.nonzero(do_cross)[0]:
        for j in range(nv):
            if rng.random() > prob_var or abs(pa[i, j] - pb[i, j]) < 1e-14:
                continue
            y1, y2 = min(pa[i, j], pb[i, j]), max(pa[i, j], pb[i, j])
            u = rng.random()

            def child(beta_bound):
                alpha = 2.0 - beta_bound ** -(eta + 1.0)
                if u <= 1.0 / alpha:
                    return (u * alpha) ** (1.0 / (eta + 1.0))
                return (1.0 / (2.0 - u * alpha)) ** (1.0 / (eta + 1.0))
            b1 = child(1.0 + 2.0 * (y1 - xl[j]) / (y2 - y1))
            b2 = child(1.0 + 2.0 * (xu[j] - y2) / (y2 - y1))
            c1 = min(max(0.5 * ((y1 + y2) - b1 * (y2 - y1)), xl[j]), xu[j])
            c2 = min(max(0.5 * ((y1 + y2) + b2 * (y2 - y1)), xl[j]), xu[j])
            if rng.random() < 0.5:
                c1, c2 = c2, c1
            ca[i, j], cb[i, j] = c1, c2
    return ca, cb


def sbx_vectorised(rng, pa, pb, xl, xu, eta=3.0, prob=1.0, prob_var=0.5):
    """Same operator, whole-population numpy form (used by the driver)."""
    n, nv = pa.shape
    y1, y2 = np.minimum(pa, pb), np.maximum(pa, pb)
    d = np.maximum(y2 - y1, 1e-300)
    u = rng.random((n, nv))

    def betaq(bound):
        alpha = 2.0 - bound ** -(eta + 1.0)
        return np.where(u <= 1.0 / alpha, (u * alpha) ** (1.0 / (eta + 1.0)),
                        (1.0 / np.maximum(2.0 - u * alpha, 1e-300)) ** (1.0 / (eta + 1.0)))
    c1 = np.clip(0.5 * ((y1 + y2) - betaq(1.0 + 2.0 * (y1 - xl) / d) * d), xl, xu)
    c2 = np.clip(0.5 * ((y1 + y2) + betaq(1.0 + 2.0 * (xu - y2) / d) * d), xl, xu)
    swap = rng.random((n, nv)) < 0.5
    c1, c2 = np.where(swap, c2, c1), np.where(swap, c1, c2)
    active = (rng.random((n, 1)) < prob) & (rng.random((n, nv)) < prob_var) & ((y2 - y1) > 1e-14)
    return np.where(active, c1, pa), np.where(active, c2, pb)


def polynomial_mutation(rng, x, xl, xu, eta=3.0, prob=0.5):
    """Bounded polynomial mutation, per-variable probability `prob`."""
    n, nv = x.shape
    span = xu - xl
    d1, d2 = (x - xl) / span, (xu - x) / span
    u = rng.random((n, nv))
    mpow = 1.0 / (eta + 1.0)
    lo = (2.0 * u + (1.0 - 2.0 * u) * (1.0 - d1) ** (eta + 1.0)) ** mpow - 1.0
    hi = 1.0 - (2.0 * (1.0 - u) + 2.0 * (u - 0.5) * (1.0 - d2) ** (eta + 1.0)) ** mpow
    y = np.clip(x + np.where(u <= 0.5, lo, hi) * span, xl, xu)
    return np.where(rng.random((n, nv)) < prob, y, x)


def hux(rng, pa, pb, prob=0.2, prob_hux=0.5):
    """Half-uniform crossover on boolean matrices (pymoo "bin_hux"): with probability `prob` per mating,
    exchange ceil(prob_hux * #differing) randomly chosen differing bits."""
    ca, cb = pa.copy(), pb.copy()
    for i in np.nonzero(rng.random(pa.shape[0]) < prob)[0]:
        diff = np.nonzero(pa[i] != pb[i])[0]
        n = int(np.ceil(len(diff) * prob_hux))
        if n:
            sw = rng.permutation(diff)[:n]
            ca[i, sw], cb[i, sw] = pb[i, sw], pa[i, sw]
    return ca, cb


def bitflip(rng, x, prob=0.01):
    """pymoo "bin_bitflip": flip every bit independently with probability `prob`."""
    flip = rng.random(x.shape) < prob
    return np.where(flip, 1.0 - x, x)


def vary(rng, A, B, xl, xu, mask, eta_c, eta_m, prob_m, n_off):
    """Crossover + mutation for a (possibly mixed-variable) population: real / int columns get SBX + polynomial
    mutation (int: rounded, as pymoo's int_sbx / int_pm do), bool columns HUX + bit-flip
    (operators.py:38-63 MixedVariable* with the reference's probabilities)."""
    mask = np.asarray(mask)
    num = mask != "bool"
    ca, cb = A.copy(), B.copy()
    if num.any():
        ca[:, num], cb[:, num] = sbx_vectorised(rng, A[:, num], B[:, num], xl[num], xu[num], eta_c)
    if (~num).any():
        ca[:, ~num], cb[:, ~num] = hux(rng, A[:, ~num], B[:, ~num], 0.2)
    off = np.concatenate([ca, cb])[rng.permutation(2 * A.shape[0])[:n_off]]
    if num.any():
        off[:, num] = polynomial_mutation(rng, off[:, num], xl[num], xu[num], eta_m, prob_m)
    if (~num).any():
        off[:, ~num] = bitflip(rng, off[:, ~num], 10 / 1000)
    ints = mask == "int"
    if ints.any():
        off[:, ints] = np.clip(np.rint(off[:, ints]), xl[ints], xu[ints])
    return off


# ----------------------------- NSGA-II machinery ------------------------------------------
def fast_non_dominated_sort(F):
    n = F.shape[0]
    dom = (np.all(F[:, None, :] <= F[None, :, :], axis=2) & np.any(F[:, None, :] < F[None, :, :], axis=2))
    n_dominating = dom.sum(axis=0)          # how many dominate j
    rank = np.full(n, -1)
    fronts, current, r = [], np.nonzero(n_dominating == 0)[0], 0
    while current.size:
        rank[current] = r
        fronts.append(current)
        n_dominating = n_dominating - dom[current].sum(axis=0)
        n_dominating[rank >= 0] = -1
        current = np.nonzero(n_dominating == 0)[0]
        r += 1
    return fronts, rank


def crowding_distance(F):
    n, m = F.shape
    if n <= 2:
        return np.full(n, np.inf)
    cd = np.zeros(n)
    for k in range(m):
        order = np.argsort(F[:, k], kind="mergesort")
        f = F[order, k]
        span = f[-1] - f[0]
        cd[order[0]] = cd[order[-1]] = np.inf
        if span > 0:
            cd[order[1:-1]] += (f[2:] - f[:-2]) / span
    return cd


def _eliminate_duplicates(X, ref=None, eps=1e-16):
    keep = np.ones(X.shape[0], bool)
    for i in range(X.shape[0]):
        if not keep[i]:
            continue
        d = np.abs(X[i + 1:] - X[i]).max(axis=1) if i + 1 < X.shape[0] else np.zeros(0)
        keep[i + 1:] &= d > eps
    if ref is not None and ref.size:
        keep &= np.abs(X[:, None, :] - ref[None, :, :]).max(axis=2).min(axis=1) > eps
    return keep


class Individual:
    def __init__(self, X, F):
        self.X, self.F = X, F


class Result:
    pass


def minimize(problem, algorithm, pop_size, n_gen, sampling, seed=1, callback=None, eta_c=3.0, eta_m=3.0,
             prob_m=0.5, verbose=False, mask=None):
    """pymoo.optimize.minimize(problem, algorithm, ("n_gen", n_gen)) stand-in.

    problem: has n_var, n_obj, xl, xu and _evaluate(x, out) (GenerationProblem);
    algorithm: "ga" | "nsga2"; sampling: object with _do(problem, n) (operators.get_operators);
    mask: per-variable type "real" | "int" | "bool" (None = all real) — the BigGAN configs mix real z with
    boolean class bits (operators.py:39), the GPT2 config is all-int."""
    rng = np.random.default_rng(seed)
    np.random.seed(seed)   # the reference's Sampling classes draw from numpy's global RNG (operators.py:24-25)
    xl = np.broadcast_to(np.asarray(problem.xl, float), (problem.n_var,))
    xu = np.broadcast_to(np.asarray(problem.xu, float), (problem.n_var,))
    nsga = algorithm == "nsga2"
    mask = np.asarray(["real"] * problem.n_var if mask is None else list(mask))

    def evaluate(X):
        out = {}
        problem._evaluate(X, out)
        F = np.asarray(out["F"], dtype=float)
        return F.reshape(X.shape[0], -1)

    def survive(X, F, n):
        if not nsga:
            order = np.argsort(F[:, 0], kind="mergesort")[:n]
            return X[order], F[order], np.zeros(len(order), int), -F[order, 0]
        fronts, rank = fast_non_dominated_sort(F)
        chosen, cds = [], []
        for fr in fronts:
            cd = crowding_distance(F[fr])
            if len(chosen) + len(fr) > n:
                order = np.argsort(-cd, kind="mergesort")[: n - len(chosen)]
                fr, cd = fr[order], cd[order]
            chosen.extend(fr.tolist())
            cds.extend(cd.tolist())
            if len(chosen) >= n:
                break
        chosen = np.array(chosen)
        return X[chosen], F[chosen], rank[chosen], np.array(cds)

    X = np.asarray(sampling._do(problem, pop_size), dtype=float)
    X = X[_eliminate_duplicates(X)]
    pad = (-X.shape[0]) % getattr(getattr(problem, "config", None), "batch_size", 1)
    if pad:   # SURVEY 8a note 8: the engine needs whole minibatches; refill instead of asserting
        X = np.concatenate([X, np.asarray(sampling._do(problem, pad), dtype=float)])
    F = evaluate(X)
    X, F, rank, cd = survive(X, F, pop_size)
    algo = Result()
    algo.problem = problem
    for gen in range(1, n_gen + 1):
        n = X.shape[0]
        # binary tournament: lower rank wins, then larger crowding distance (GA: better fitness)
        a, b = rng.integers(0, n, (2, pop_size)), rng.integers(0, n, (2, pop_size))
        better = (rank[a] < rank[b]) | ((rank[a] == rank[b]) & (cd[a] >= cd[b]))
        parents = np.where(better, a, b)
        off = vary(rng, X[parents[0]], X[parents[1]], xl, xu, mask, eta_c, eta_m, prob_m, pop_size)
        off = off[_eliminate_duplicates(off, X)]
        bs = getattr(getattr(problem, "config", None), "batch_size", 1)
        if off.shape[0] % bs:
            extra = bs - off.shape[0] % bs
            pick = rng.integers(0, n, (2, extra))
            off = np.concatenate([off, vary(rng, X[pick[0]], X[pick[1]], xl, xu, mask, eta_c, eta_m, 1.0, extra)])
        Fo = evaluate(off)
        X, F, rank, cd = survive(np.concatenate([X, off]), np.concatenate([F, Fo]), pop_size)
        algo.pop = [Individual(x, f if nsga else f[0]) for x, f in zip(X, F)]
        algo.n_gen = gen
        if verbose:
            print("gen %4d | n_eval %6d | best %s" % (gen, gen * pop_size, np.array2string(F.min(axis=0), precision=5)))
        if callback is not None:
            callback(algo)
    res = Result()
    res.pop = [Individual(x, f if nsga else f[0]) for x, f in zip(X, F)]
    if nsga:
        front = fast_non_dominated_sort(F)[0][0]
        res.X, res.F = X[front], F[front]
    else:
        best = int(np.argmin(F[:, 0]))
        res.X, res.F = X[best], F[best]
    res.G = np.zeros(np.atleast_2d(res.X).shape[0])
    res.CV = np.zeros(np.atleast_2d(res.X).shape[0])
    return res


def pseudo_weights_choice(F, weights=(0.0, 1.0)):
    """pymoo get_decision_making("pseudo-weights", w).do(F): index of the point whose pseudo-weight vector is closest to w."""
    F = np.asarray(F, float)
    span = F.max(axis=0) - F.min(axis=0)
    span[span == 0] = 1.0
    pw = (F.max(axis=0) - F) / span
    pw = pw / np.maximum(pw.sum(axis=1, keepdims=True), 1e-300)
    return int(np.argmin(np.linalg.norm(pw - np.asarray(weights, float), axis=1)))
