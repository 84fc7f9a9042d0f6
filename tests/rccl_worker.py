"""Worker of tests/test_gpu_engine.py::test_rccl_two_ranks_all_gather — one process per GPU under torch.distributed.run, backend
"nccl" (= RCCL over xGMI): every rank scores its contiguous shard of the SAME seeded population on its own GPU, one
all_gather_into_tensor of [P/N, n_obj] returns all rows (clip_glass_amd/parallel.py), and rank 0 compares them with the
whole population scored by its own engine in one call — bitwise (device noise = f(seed, generation, GLOBAL minibatch, layer)).
Also covers the ragged split (3 minibatches over 2 ranks) and bench.py's weak-scaling form (evaluate_local)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    import glass_models as M
    from clip_glass_amd import synth
    from clip_glass_amd.parallel import ShardedEvaluator
    name, bs = "mini", 4
    P = 8 * world
    c = M.CONFIGS[name]
    sd = M.make_state(name, 0)
    x = synth.latents(7, P, c["latent"])
    e = M.make_engine(name, sd, batch_size=bs, use_discriminator=True, max_pop=P, noise_mode=1, noise_seed=99, device=local)
    e.set_target(M.make_target(synth.normal(3, "feat", (8, c["clip"][5]))))
    ev = ShardedEvaluator(e, dist, rank, world, bs)
    assert ev.gather_device() == torch.device("cuda", local)
    F_sharded = ev.evaluate_global(x, generation=5)                       # even shards: one all_gather_into_tensor
    F_ragged = ev.evaluate_global(x[:P - bs], generation=5)               # ragged shards: padded gather, trimmed
    lo = rank * 8
    F_weak = ev.evaluate_local(x[lo:lo + 8], generation=5)                # bench.py's form
    F_whole = e.evaluate(x, generation=5)
    F_whole_r = e.evaluate(x[:P - bs], generation=5)
    ok = bool(np.array_equal(F_sharded, F_whole) and np.array_equal(F_ragged, F_whole_r) and np.array_equal(F_weak, F_whole))
    flag = torch.tensor([1 if ok else 0], device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    e.close()
    if rank == 0:
        print(json.dumps(dict(ok=bool(flag.item()), world=world, backend=dist.get_backend(), rows=int(F_sharded.shape[0]),
                              max_abs_diff=float(np.abs(F_sharded - F_whole).max()))))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
