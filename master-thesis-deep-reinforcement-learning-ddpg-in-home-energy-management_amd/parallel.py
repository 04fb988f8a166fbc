"""Data-parallel glue (host logic only; device-agnostic so it is testable with gloo on CPU).

The reference has no collective at all: it launches 40 independent `julia` processes on 2 GPUs
(RL-SHEMS_bs_scheduler_1179_08_on_01-98.sh:67-87).  Here one process drives one MI355X; env shards
are independent (no cross-env term anywhere in shems_LU1.jl) and the replicas of the learner stay
identical by summing gradients over RCCL (`torch.distributed` backend "nccl" on ROCm) between the
backward kernels and the ADAM kernel -- twice per update, because the actor gradient is taken
through the already-updated critic (DDPG.jl:137-140).  Messages are 516 KB each (129 001 / 129 002
f32): latency-class, one all-reduce per network, no bucketing needed.
"""
from __future__ import annotations


def shard_envs(total_envs, rank, world):
    """Contiguous shard [offset, offset + count) of `total_envs` for `rank`."""
    base, rem = divmod(int(total_envs), int(world))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


class GradSync:
    """Keeps learner replicas identical.  `dist` is `torch.distributed` (initialised) or None."""

    def __init__(self, dist=None):
        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size() > 1) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0

    @property
    def grad_scale(self):
        """Factor ADAM applies to the summed gradient: mean over replicas = gradient of the loss
        averaged over the union of the replicas' minibatches."""
        return 1.0 / self.world

    def broadcast(self, *tensors, src=0):
        if self.dist:
            for t in tensors:
                self.dist.broadcast(t, src=src)

    def sum_(self, t):
        if self.dist:
            self.dist.all_reduce(t)
        return t

    def sum_async_(self, t):
        """Start the sum over replicas and return the work handle (None without replicas); handle.wait() orders the CURRENT
        stream behind the collective without blocking the host (NCCL/RCCL) -- whatever is enqueued between the two runs under it."""
        if self.dist:
            return self.dist.all_reduce(t, async_op=True)
        return None

    def minmax_(self, t_min, t_max):
        if self.dist:
            self.dist.all_reduce(t_min, op=self.dist.ReduceOp.MIN)
            self.dist.all_reduce(t_max, op=self.dist.ReduceOp.MAX)
        return t_min, t_max

    def mean_scalar(self, value, weight=1.0):
        """Weighted mean of a python scalar over replicas (episode scores)."""
        if not self.dist:
            return float(value)
        import torch
        t = torch.tensor([float(value) * weight, float(weight)], dtype=torch.float64)
        backend = self.dist.get_backend()
        if backend == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t)
        return float(t[0].item() / t[1].item())
