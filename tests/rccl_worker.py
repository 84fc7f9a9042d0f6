"""Worker of tests/test_gpu_engine.py::test_rccl_two_ranks_all_gather — one process per GPU under torch.distributed.run, backend
"nccl" (= RCCL over xGMI): every rank scores its contiguous shard of the SAME seeded population on its own GPU, one
all_gather_into_tensor of [P/N, n_obj] returns all rows (clip_glass_amd/parallel.py), and rank 0 compares them with the
whole population scored by its own engine in one call — bitwise (device noise = f(seed, generation, GLOBAL minibatch, layer)).
Also covers the ragged split (3 minibatches over 2 ranks) and bench.py's weak-scaling form (evaluate_local), and then runs that
weak-scaling loop for 3 steps at the headline geometry (per-rank milliseconds in the JSON line).  Started with ONE rank (a 1-GPU box) it
still drives `all_gather_into_tensor` on nccl (ShardedEvaluator(force_collective=True)) from the engine's device-resident rows."""
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
    ev = ShardedEvaluator(e, dist, rank, world, bs, force_collective=True)     # (world 1 on a 1-GPU box: the nccl all-gather itself still runs)
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
    src = ev.last_gather_source
    e.close()
    # bench.py's weak-scaling loop at the HEADLINE geometry (ffhq 1024 px G + D + CLIP ViT-B/32, 64 candidates per rank), 3 timed steps:
    # the first multi-GPU lease yields per-rank milliseconds and a verified all_gather_into_tensor in one shot
    weak = None
    if os.environ.get("GLASS_RCCL_WORKER_BENCH", "1") != "0":
        import time
        from clip_glass_amd.engine import Engine
        ch, lat, clip = synth.FFHQ_CHANNELS, 512, (768, 12, 12, 32, 224, 512)
        sdh = synth.make_state(synth.stylegan2_g_spec(ch, lat, 8), 0)
        sdh.update(synth.make_state(synth.stylegan2_d_spec(ch), 0))
        sdh.update(synth.make_state(synth.clip_visual_spec(clip[0], clip[1], clip[3], clip[4], clip[5]), 0))
        eh = Engine(ch[::-1], latent_size=lat, mapping_layers=8, batch_size=4, use_discriminator=True, n_obj=2, max_pop=64, clip=clip,
                    noise_mode=1, noise_seed=1234, device=local)
        eh.load_state(sdh)
        eh.finalize()
        eh.set_target(np.ones(clip[5], np.float32))
        evh = ShardedEvaluator(eh, dist, rank, world, 4, force_collective=True)
        pops = [synth.latents(1000 * rank + s, 64, lat) for s in range(4)]
        Fh = evh.evaluate_local(pops[0], generation=0)                   # warm-up
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(1, 4):
            Fh = evh.evaluate_local(pops[s], generation=s)
        dist.barrier()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        tt = torch.tensor([ms], device="cuda")
        allms = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allms, tt)
        weak = dict(ms_per_step_per_rank=[round(float(a.item()), 3) for a in allms], rows_gathered=int(Fh.shape[0]),
                    candidates_per_s=round(64 * world / (max(float(a.item()) for a in allms) / 1e3), 1), finite=bool(np.isfinite(Fh).all()),
                    gather_source=evh.last_gather_source)
        eh.close()
    if rank == 0:
        print(json.dumps(dict(ok=bool(flag.item()), world=world, backend=dist.get_backend(), rows=int(F_sharded.shape[0]),
                              max_abs_diff=float(np.abs(F_sharded - F_whole).max()), gather_source=src, weak_scaling=weak)))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
