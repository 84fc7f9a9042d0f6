"""CLI — mirror of /root/reference/run.py (same flags and outputs) on the HIP engine.

    python -m clip_glass_amd.run --config StyleGAN2_ffhq_d --target "a wolf at night ..." \\
        --generations 50 --tmp-folder ./tmp [--weights synthetic:0 --clip-weights synthetic:0]

The search runs on the native GA / NSGA-II driver in search.py (pymoo 0.4.2.1 does not import under numpy >= 1.24).
Outputs (run.py:29-51, 79-125): genetic-it-*.{jpg,txt} every --save-each generations,
genetic_result (pickle: X, F, G, CV), F.jpg (Pareto scatter, two objectives), ls_result, output.{jpg,txt}.
"""
import argparse
import os
import pickle

import numpy as np

from .config import get_config
from .operators import get_operators
from .problem import GenerationProblem
from . import search


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--device", type=str, default="cuda")                       # run.py:17
    p.add_argument("--config", type=str, default="StyleGAN2_ffhq_d")
    p.add_argument("--generations", type=int, default=500)
    p.add_argument("--save-each", type=int, default=50)
    p.add_argument("--tmp-folder", type=str, default="./tmp")
    p.add_argument("--target", type=str, default="a wolf at night with the moon in the background")
    # additions (no checkpoints / vocab in this environment)
    p.add_argument("--weights", type=str, default=None, help="directory with G.pth/D.pth, or synthetic:<seed>")
    p.add_argument("--clip-weights", type=str, default=None, help="ViT-B-32.pt, or synthetic:<seed>")
    p.add_argument("--bpe-path", type=str, default=None)
    p.add_argument("--pop-size", type=int, default=None)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--dist", action="store_true",
                   help="multi-GPU: run under `torchrun --nproc-per-node N -m clip_glass_amd.run --dist ...` (one process per GPU, "
                        "RCCL); every rank runs the same seeded search, scores its shard of each population, and one all-gather "
                        "returns all fitness rows; rank 0 writes the outputs")
    return p


def main(argv=None, extra_config=None):
    config = build_parser().parse_args(argv)
    over = {k: v for k, v in vars(config).items() if v is not None}
    vars(config).update(get_config(config.config))                             # run.py:25
    for k in ("weights", "clip_weights", "bpe_path", "pop_size"):
        if k in over:
            setattr(config, k, over[k])
    if extra_config:
        vars(config).update(extra_config)
    state = dict(iteration=0)
    dist, rank0 = None, True
    if getattr(config, "dist", False):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = os.environ.get("GLASS_DIST_BACKEND", "nccl")     # "nccl" IS RCCL on ROCm (gloo: CPU tests)
            if backend == "nccl":
                torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group(backend)
        rank0 = dist.get_rank() == 0

    def save_callback(algorithm):                                              # run.py:29-51
        state["iteration"] += 1
        it = state["iteration"]
        if rank0 and (it % config.save_each == 0 or it == config.generations):
            if config.problem_args["n_obj"] == 1:
                X = np.stack([p.X for p in sorted(algorithm.pop, key=lambda p: p.F)])
            else:
                X = np.stack([p.X for p in algorithm.pop])
            ls = config.latent(config)
            ls.set_from_population(X)
            generated = algorithm.problem.generator.generate(ls, minibatch=config.batch_size)
            ext = "txt" if config.task == "img2txt" else "jpg"                   # run.py:46-50
            name = "genetic-it-%d.%s" % (it, ext) if it < config.generations else "genetic-it-final.%s" % ext
            algorithm.problem.generator.save(generated, os.path.join(config.tmp_folder, name))

    problem = GenerationProblem(config, dist=dist)
    operators = get_operators(config)
    os.makedirs(config.tmp_folder, exist_ok=True)
    # (same seed on every rank => identical GA state everywhere: no broadcast of the population is needed)
    res = search.minimize(problem, config.algorithm, config.pop_size, config.generations, operators["sampling"],
                          seed=config.seed, callback=save_callback, verbose=rank0, mask=operators.get("mask"))
    if not rank0:
        return res
    with open(os.path.join(config.tmp_folder, "genetic_result"), "wb") as f:   # run.py:79-84
        pickle.dump(dict(X=res.X, F=res.F, G=res.G, CV=res.CV), f)
    if config.problem_args["n_obj"] == 2:
        try:                                                                   # run.py:86-89
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            Fp = np.atleast_2d(res.F)
            plt.figure()
            plt.scatter(Fp[:, 0], Fp[:, 1], color="red")
            plt.xlabel("similarity"); plt.ylabel("discriminator")
            plt.savefig(os.path.join(config.tmp_folder, "F.jpg"))
            plt.close()
        except Exception as ex:   # plotting is cosmetic
            print("Warning: could not write F.jpg (%s)" % ex)
    if config.problem_args["n_obj"] == 1:
        X = np.stack([p.X for p in sorted(res.pop, key=lambda p: p.F)])
    else:
        X = np.stack([p.X for p in res.pop])
    ls = config.latent(config)
    ls.set_from_population(X)
    with open(os.path.join(config.tmp_folder, "ls_result"), "wb") as f:        # run.py:101 (same file name, npz payload)
        np.savez(f, **ls.state_dict())
    if config.problem_args["n_obj"] == 1:
        X = np.atleast_2d(res.X)
    else:
        X = np.atleast_2d(np.atleast_2d(res.X)[search.pseudo_weights_choice(np.atleast_2d(res.F), [0, 1])])  # run.py:103-113
    ls.set_from_population(X)
    generated = problem.generator.generate(ls)                                 # run.py:118
    ext = "txt" if config.task == "img2txt" else "jpg"                         # run.py:120-125
    problem.generator.save(generated, os.path.join(config.tmp_folder, "output.%s" % ext))
    return res


if __name__ == "__main__":
    main()
