"""Fitness problem — drop-in for /root/reference/problem.py.

Same class name, constructor and `_evaluate(x, out)` contract: `out["F"]` float32 [P] or
[P,2] = column_stack(-sim, relu(1-D)), `out["G"]` zeros [P] (problem.py:14-29).  The base
class is pymoo's Problem when pymoo is importable (0.4.2.1 layout first), otherwise a
minimal stand-in with the attributes pymoo reads.
"""
import warnings

import numpy as np

try:
    from pymoo.model.problem import Problem            # pymoo==0.4.2.1 (requirements.txt:29)
except ImportError:
    try:
        from pymoo.core.problem import Problem          # newer pymoo
    except ImportError:
        class Problem:                                  # attribute bag (pymoo not installed)
            def __init__(self, n_var=-1, n_obj=-1, n_constr=0, xl=None, xu=None, **kwargs):
                self.n_var, self.n_obj, self.n_constr = n_var, n_obj, n_constr
                self.xl = np.full(n_var, xl, dtype=float) if np.isscalar(xl) else xl
                self.xu = np.full(n_var, xu, dtype=float) if np.isscalar(xu) else xu

            def evaluate(self, x, *args, **kwargs):
                out = {}
                self._evaluate(np.atleast_2d(x), out, *args, **kwargs)
                return out

from .generator import Generator


class GenerationProblem(Problem):
    def __init__(self, config, dist=None):
        """dist: an initialised torch.distributed module (one process per GPU; backend "nccl" = RCCL), or None = use the default
        process group if one is initialised with more than one rank (generator.py).  Every rank calls _evaluate with the SAME x;
        each scores its contiguous shard and one all-gather returns all rows (parallel.py)."""
        self.generator = Generator(config, dist=dist)
        self.config = config
        super().__init__(**self.config.problem_args)

    def _evaluate(self, x, out, *args, **kwargs):
        ls = self.config.latent(self.config)
        x = np.asarray(x)
        P = x.shape[0]
        if self.config.task == "txt2img" and P % self.config.batch_size:
            # The reference asserts P % batch_size == 0 inside generate (models.py:112) — and pymoo's duplicate elimination
            # (run.py:65) can shrink generation 0 below pop_size, which kills the reference run.  SURVEY 8a note 8: pad instead.
            # The last row is repeated up to the next minibatch boundary and the padding rows' fitness is dropped; the real rows of
            # that last minibatch share its noise plane / minibatch-stddev group with the copies.
            pad = (-P) % self.config.batch_size
            if not getattr(self, "_pad_warned", False):
                self._pad_warned = True
                warnings.warn("GenerationProblem._evaluate: population of %d rows is not a multiple of batch_size %d; padding with %d "
                              "copies of the last row (the reference asserts here, models.py:112).  The real rows of the last "
                              "minibatch share its minibatch-stddev group with the copies, so their hinge objective is not "
                              "comparable with a full minibatch's." % (P, self.config.batch_size, pad), RuntimeWarning, stacklevel=2)
            x = np.concatenate([x, np.repeat(x[-1:], pad, axis=0)])
        ls.set_from_population(x)
        F = self.generator.evaluate(ls)[:P]
        if self.config.problem_args["n_obj"] == 2 and self.config.use_discriminator:
            out["F"] = F                                # column_stack((-sim, hinge)) (problem.py:25)
        else:
            out["F"] = F[:, 0]                          # -sim (problem.py:27)
        out["G"] = np.zeros((P))
