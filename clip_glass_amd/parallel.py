"""Population sharding across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" == RCCL on ROCm).  Candidates
are independent except inside a minibatch (shared noise plane, mbstd group), so the
population is cut into contiguous blocks that are multiples of batch_size; weights and
the target feature are replicated; the only collective on the data path is ONE
all-gather of the [P_local, n_obj] float32 fitness rows per generation (512 B per rank
at 64 x 2 — latency-bound on xGMI).  Noise is a pure function of (seed, generation,
GLOBAL minibatch index, layer) and every kernel dispatcher decides on the layer geometry alone (launch-size thresholds are
evaluated at a nominal population, csrc/common.h), so a sharded run reproduces the single-GPU rows BIT FOR BIT
(tests/test_gpu_engine.py: test_pop512_as_eight_shards_of_64, test_full_size_offset_shards).
"""
import numpy as np


def shard_bounds(P, world, batch_size):
    """Contiguous shard [lo, hi) per rank, each a multiple of batch_size (as even as possible)."""
    assert P % batch_size == 0
    n_mb = P // batch_size
    base, extra = divmod(n_mb, world)
    bounds, lo = [], 0
    for r in range(world):
        n = (base + (1 if r < extra else 0)) * batch_size
        bounds.append((lo, lo + n))
        lo += n
    return bounds


class ShardedEvaluator:
    def __init__(self, engine, dist, rank, world, batch_size, device=None):
        """device: the GPU this rank's engine runs on (default: the engine's own).  The RCCL gather tensors are staged THERE — not on
        torch's current device, which is cuda:0 on every rank unless the launcher called torch.cuda.set_device (a duplicate-GPU
        error or a hang in the collective otherwise)."""
        self.engine, self.dist, self.rank, self.world, self.batch_size = engine, dist, rank, world, batch_size
        if device is None:
            device = getattr(getattr(engine, "cfg", None), "device", 0)
        self.device = int(device)

    def gather_device(self):
        """torch device of the all-gather staging tensors: this rank's GPU under nccl (= RCCL), host memory under gloo."""
        import torch
        return torch.device("cuda", self.device) if self.dist.get_backend() == "nccl" else torch.device("cpu")

    def evaluate_local(self, x_local, generation=0, noise=None):
        """Weak-scaling form: every rank brings its own P_local rows; returns all ranks' F [P_local*world, n_obj]."""
        P = x_local.shape[0]
        first_mb = self.rank * (P // self.batch_size)
        F = self.engine.evaluate(x_local, generation=generation, first_minibatch=first_mb, noise=noise)
        return self.all_gather(F)

    def evaluate_global(self, x, generation=0):
        """Strong form used by GenerationProblem: same x on every rank, each evaluates its shard."""
        lo, hi = shard_bounds(x.shape[0], self.world, self.batch_size)[self.rank]
        F = np.zeros((0, self.engine.cfg.n_obj), np.float32)
        if hi > lo:
            F = self.engine.evaluate(x[lo:hi], generation=generation, first_minibatch=lo // self.batch_size)
        return self.all_gather(F, sizes=[b - a for a, b in shard_bounds(x.shape[0], self.world, self.batch_size)])

    def all_gather(self, F, sizes=None):
        if self.dist is None or self.world == 1:
            return F
        import torch
        dev = self.gather_device()
        n_obj = F.shape[1]
        if sizes is None or len(set(sizes)) == 1:
            t = torch.from_numpy(np.ascontiguousarray(F)).to(dev)
            out = torch.empty((self.world * t.shape[0], n_obj), dtype=t.dtype, device=dev)
            self.dist.all_gather_into_tensor(out, t)
            return out.cpu().numpy()
        m = max(sizes)  # ragged shards: pad to the largest, gather, trim
        pad = np.zeros((m, n_obj), np.float32)
        pad[:F.shape[0]] = F
        t = torch.from_numpy(pad).to(dev)
        out = torch.empty((self.world * m, n_obj), dtype=t.dtype, device=dev)
        self.dist.all_gather_into_tensor(out, t)
        out = out.cpu().numpy().reshape(self.world, m, n_obj)
        return np.concatenate([out[r, :sizes[r]] for r in range(self.world)])
