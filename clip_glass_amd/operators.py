"""Operators — mirror of /root/reference/operators.py (the API surface `run.py` uses:
`get_operators(config) -> dict(sampling, crossover, mutation)`).

The three custom Sampling classes are restated with builtin dtypes (np.float / np.bool are
gone from numpy >= 1.24, which breaks the reference as written: operators.py:10,18,34).
Crossover / mutation come from pymoo's factory exactly as in the reference when pymoo is
importable; without pymoo the factory names are returned as specs and the native driver
(search.py) applies the same operators per variable type from the `mask` entry.
"""
import numpy as np

try:
    from pymoo.model.sampling import Sampling
    from pymoo.factory import get_crossover, get_mutation, get_sampling
    HAVE_PYMOO = True
except ImportError:
    HAVE_PYMOO = False

    class Sampling:
        def __init__(self):
            pass

        def do(self, problem, n_samples, **kwargs):
            return self._do(problem, n_samples, **kwargs)

    def _spec(kind):
        def make(name, **kw):
            return dict(kind=kind, name=name, **kw)
        return make
    get_crossover, get_mutation, get_sampling = _spec("crossover"), _spec("mutation"), _spec("sampling")


class TruncatedNormalRandomSampling(Sampling):
    """operators.py:9-15"""

    def __init__(self, var_type=float):
        super().__init__()
        self.var_type = var_type

    def _do(self, problem, n_samples, **kwargs):
        from scipy.stats import truncnorm
        return truncnorm.rvs(-2, 2, size=(n_samples, problem.n_var)).astype(np.float32)


class NormalRandomSampling(Sampling):
    """operators.py:17-25"""

    def __init__(self, mu=0, std=1, var_type=float):
        super().__init__()
        self.mu, self.std, self.var_type = mu, std, var_type

    def _do(self, problem, n_samples, **kwargs):
        return np.random.normal(self.mu, self.std, size=(n_samples, problem.n_var))


class BinaryRandomSampling(Sampling):
    """operators.py:27-34"""

    def __init__(self, prob=0.5):
        super().__init__()
        self.prob = prob

    def _do(self, problem, n_samples, **kwargs):
        val = np.random.random((n_samples, problem.n_var))
        return (val < self.prob).astype(bool)


class MixedVariableSampling(Sampling):
    """pymoo.operators.mixed_variable_operator.MixedVariableSampling restated (used when pymoo is absent):
    each variable type is drawn by its own Sampling over the columns the mask assigns to it."""

    def __init__(self, mask, process):
        super().__init__()
        self.mask, self.process = np.asarray(mask), process

    def _do(self, problem, n_samples, **kwargs):
        X = np.empty((n_samples, len(self.mask)), dtype=float)
        for kind, sampling in self.process.items():
            cols = np.nonzero(self.mask == kind)[0]

            class _Sub:                      # the sub-problem pymoo hands to each per-type operator
                n_var = len(cols)
                xl = np.asarray(problem.xl)[cols] if np.ndim(problem.xl) else problem.xl
                xu = np.asarray(problem.xu)[cols] if np.ndim(problem.xu) else problem.xu
            X[:, cols] = np.asarray(sampling._do(_Sub, n_samples), dtype=float)
        return X


class IntegerRandomSampling(Sampling):
    """pymoo "int_random": uniform integers in [xl, xu]."""

    def _do(self, problem, n_samples, **kwargs):
        xl = np.broadcast_to(np.asarray(problem.xl, float), (problem.n_var,))
        xu = np.broadcast_to(np.asarray(problem.xu, float), (problem.n_var,))
        return np.floor(xl + np.random.random((n_samples, problem.n_var)) * (xu - xl + 1)).astype(float)


def get_operators(config):
    """operators.py:37-81.  The extra "mask" entry (variable types) is what the native driver (search.py)
    needs to apply the same per-type operators when pymoo is absent."""
    if config.config.split("_")[0] == "StyleGAN2":
        return dict(sampling=NormalRandomSampling(),
                    crossover=get_crossover("real_sbx", prob=1.0, eta=3.0),
                    mutation=get_mutation("real_pm", prob=0.5, eta=3.0))
    if config.config in ("DeepMindBigGAN256", "DeepMindBigGAN512"):
        mask = ["real"] * config.dim_z + ["bool"] * config.num_classes
        process = {"real": TruncatedNormalRandomSampling(), "bool": BinaryRandomSampling(prob=5 / 1000)}
        if not HAVE_PYMOO:
            return dict(sampling=MixedVariableSampling(mask, process), mask=mask,
                        crossover=dict(kind="crossover", name="mixed",
                                       real=get_crossover("real_sbx", prob=1.0, eta=3.0), bool=get_crossover("bin_hux", prob=0.2)),
                        mutation=dict(kind="mutation", name="mixed",
                                      real=get_mutation("real_pm", prob=0.5, eta=3.0), bool=get_mutation("bin_bitflip", prob=10 / 1000)))
        from pymoo.operators.mixed_variable_operator import MixedVariableCrossover, MixedVariableMutation
        from pymoo.operators.mixed_variable_operator import MixedVariableSampling as PymooMixedSampling
        return dict(
            sampling=PymooMixedSampling(mask, process), mask=mask,
            crossover=MixedVariableCrossover(mask, {"real": get_crossover("real_sbx", prob=1.0, eta=3.0),
                                                    "bool": get_crossover("bin_hux", prob=0.2)}),
            mutation=MixedVariableMutation(mask, {"real": get_mutation("real_pm", prob=0.5, eta=3.0),
                                                  "bool": get_mutation("bin_bitflip", prob=10 / 1000)}))
    if config.config == "GPT2":
        return dict(sampling=get_sampling("int_random") if HAVE_PYMOO else IntegerRandomSampling(),
                    mask=["int"] * config.dim_z,
                    crossover=get_crossover("int_sbx", prob=1.0, eta=3.0),
                    mutation=get_mutation("int_pm", prob=0.5, eta=3.0))
    raise Exception("Unknown config")
