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
    def __init__(self, engine, dist, rank, world, batch_size, device=None, force_collective=False):
        """device: the GPU this rank's engine runs on (default: the engine's own).  The RCCL gather tensors are staged THERE — not on
        torch's current device, which is cuda:0 on every rank unless the launcher called torch.cuda.set_device (a duplicate-GPU
        error or a hang in the collective otherwise).  force_collective: run the all-gather at world size 1 too (tests/rccl_worker.py
        on a 1-GPU box: the `nccl` call itself is then exercised)."""
        self.engine, self.dist, self.rank, self.world, self.batch_size = engine, dist, rank, world, batch_size
        self.force_collective = bool(force_collective)
        self.last_gather_source = None          # "device" / "host": where the last all-gather read this rank's rows from
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

    def _local_rows(self, F, dev):
        """This rank's rows as a tensor on the gather device.  Under `nccl` they are taken where evaluate() left them — the engine's own
        device buffer (glass_engine_last_F_device), no D2H -> H2D bounce; `F` (the host copy evaluate() returned) is the gloo / fallback
        source and must hold the same rows."""
        import torch
        if dev.type == "cuda" and F.shape[0] > 0 and hasattr(self.engine, "last_F_device"):
            try:
                t = self.engine.last_F_device(F.shape[0])
                self.last_gather_source = "device"
                return t
            except Exception:
                pass
        self.last_gather_source = "host"
        return torch.from_numpy(np.ascontiguousarray(F)).to(dev)

    def all_gather(self, F, sizes=None):
        if self.dist is None or (self.world == 1 and not self.force_collective):
            return F
        import torch
        dev = self.gather_device()
        n_obj = F.shape[1]
        t = self._local_rows(F, dev)
        if sizes is None or len(set(sizes)) == 1:
            out = torch.empty((self.world * t.shape[0], n_obj), dtype=t.dtype, device=dev)
            self.dist.all_gather_into_tensor(out, t.contiguous())
            return out.cpu().numpy()
        m = max(sizes)  # ragged shards: pad to the largest (on the gather device), gather, trim
        pad = torch.zeros((m, n_obj), dtype=torch.float32, device=dev)
        pad[:F.shape[0]] = t
        out = torch.empty((self.world * m, n_obj), dtype=pad.dtype, device=dev)
        self.dist.all_gather_into_tensor(out, pad)
        out = out.cpu().numpy().reshape(self.world, m, n_obj)
        return np.concatenate([out[r, :sizes[r]] for r in range(self.world)])
