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


def native_comm(dist, timeout_s=120.0, log=None):
    """A shems_dp communicator (RCCL called in the update's own stream from native code, csrc/shems_dp.hip) for the ranks of `dist`, or
    None when it cannot be had -- then the caller keeps torch.distributed's all_reduce.  Collective: every rank calls it.

    Rank 0 draws the 128-byte id and torch.distributed broadcasts it; ncclCommInitRank runs in a helper thread so that a rendezvous that
    never completes cannot hang the job (after `timeout_s` the attempt is marked dead: a thread that completes later destroys its own
    communicator); the ranks then agree (MIN over a flag) whether ALL of
    them have a communicator and a tiny all-reduce through it gave the right sum -- one rank without it and everybody falls back.
    SHEMS_DP=torch skips the attempt."""
    import ctypes as C
    import os
    import threading
    import torch
    from . import _capi
    if dist is None or not dist.is_initialized() or dist.get_world_size() < 1:
        return None
    if os.environ.get("SHEMS_DP", "native") == "torch":
        return None
    L = _capi.lib()
    L.shems_dp_unique_id.argtypes = [C.c_char_p]
    L.shems_dp_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.shems_dp_destroy.argtypes = [C.c_void_p]
    L.shems_dp_allreduce_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    for fn in ("shems_dp_unique_id", "shems_dp_create", "shems_dp_destroy", "shems_dp_allreduce_sum"):
        getattr(L, fn).restype = C.c_int
    rank, world = dist.get_rank(), dist.get_world_size()
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    buf = C.create_string_buffer(128)
    ok = 1.0
    if rank == 0 and L.shems_dp_unique_id(buf) != 0:
        ok = 0.0
    idt = torch.tensor(list(buf.raw), dtype=torch.uint8, device=dev)
    dist.broadcast(idt, src=0)
    flag = torch.tensor([ok], device=dev)
    dist.broadcast(flag, src=0)
    handle = C.c_void_p()
    # The helper thread works on its OWN handle and hands it over under the lock; an attempt given up on is marked dead, and a rendezvous
    # that completes after that destroys its communicator itself instead of leaving a live one nobody owns.
    state = {"rc": None, "dead": False, "handle": None}
    lock = threading.Lock()
    mine = 0.0
    try:                                            # whatever goes wrong on this rank, it still reaches the vote below
        if float(flag.item()) > 0.5:
            idb = bytes(idt.cpu().tolist())
            cur = torch.cuda.current_device()

            def init():
                torch.cuda.set_device(cur)          # HIP's current device is per thread
                h = C.c_void_p()
                rc = L.shems_dp_create(idb, rank, world, C.byref(h))
                with lock:
                    if state["dead"]:
                        if rc == 0 and h.value:
                            L.shems_dp_destroy(h)
                    else:
                        state["rc"], state["handle"] = rc, h
            th = threading.Thread(target=init, daemon=True)
            th.start()
            th.join(timeout_s)
            with lock:
                if state["handle"] is None:
                    state["dead"] = True            # timed out (or still running): whatever it produces later is its own to destroy
                else:
                    handle = state["handle"]
        mine = 1.0 if state["rc"] == 0 and handle.value else 0.0
        if mine:
            # self-test: sum of (rank + 1) over the replicas through the new communicator, in this thread's current stream
            t = torch.full((1024,), float(rank + 1), dtype=torch.float32, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            if L.shems_dp_allreduce_sum(handle, C.c_void_p(t.data_ptr()), t.numel(), C.c_void_p(st)) != 0:
                mine = 0.0
            else:
                torch.cuda.synchronize()
                mine = 1.0 if bool((t == world * (world + 1) / 2).all()) else 0.0
    except Exception as e:                          # noqa: BLE001 -- reported below, the job continues on torch.distributed
        mine = 0.0
        if log:
            log(f"rank {rank}: native communicator attempt raised {e!r}")
    vote = torch.tensor([mine], device=dev)
    dist.all_reduce(vote, op=dist.ReduceOp.MIN)
    if float(vote.item()) < 0.5:
        if log:
            log(f"rank {rank}: no native RCCL communicator ({L.shems_last_error().decode('utf-8', 'replace') if not mine else 'another rank failed'}): "
                "gradients go through torch.distributed")
        if handle.value and state["rc"] == 0:
            L.shems_dp_destroy(handle)
        return None
    return handle


def direct_comm(dist, log=None):
    """A shems_dp record that exchanges gradients DIRECTLY through peer-mapped inboxes (csrc/shems_dp.hip: shems_dp_create_direct; the
    exchange itself is the ADAM sweep k_adam_xchg) -- no RCCL, no collective launch.  Opt-in (SHEMS_DP=direct).  Collective: every rank
    calls it; the 128-byte IPC handle blocks travel by all_gather_object; the ranks vote, and if any of them could not map every peer the
    attempt is dropped on every rank (returns None).  Works for xGMI peers of one node and -- the rehearsal form -- for several processes on
    ONE device.  At most 8 replicas."""
    import ctypes as C
    import torch
    from . import _capi
    if dist is None or not dist.is_initialized() or dist.get_world_size() < 2:
        return None
    L = _capi.lib()
    L.shems_dp_create_direct.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.shems_dp_direct_handles.argtypes = [C.c_void_p, C.c_char_p]
    L.shems_dp_direct_connect.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
    L.shems_dp_destroy.argtypes = [C.c_void_p]
    for fn in ("shems_dp_create_direct", "shems_dp_direct_handles", "shems_dp_direct_connect", "shems_dp_destroy"):
        getattr(L, fn).restype = C.c_int
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    handle = C.c_void_p()
    mine, blob = 0.0, None
    try:
        buf = C.create_string_buffer(128)
        if world <= 8 and L.shems_dp_create_direct(rank, world, C.byref(handle)) == 0 and L.shems_dp_direct_handles(handle, buf) == 0:
            blob = bytes(buf.raw)
    except Exception as e:                          # noqa: BLE001
        if log:
            log(f"rank {rank}: direct exchange attempt raised {e!r}")
    blobs = [None] * world
    dist.all_gather_object(blobs, blob)
    try:
        if blob is not None and all(b is not None for b in blobs):
            mine = 1.0
            for q, b in enumerate(blobs):
                if q != rank and L.shems_dp_direct_connect(handle, q, b) != 0:
                    mine = 0.0
                    break
    except Exception as e:                          # noqa: BLE001
        mine = 0.0
        if log:
            log(f"rank {rank}: mapping the peers' inboxes raised {e!r}")
    vote = torch.tensor([mine], device=dev)
    dist.all_reduce(vote, op=dist.ReduceOp.MIN)       # also the barrier: nobody updates before everybody has mapped everybody
    if float(vote.item()) < 0.5:
        if log:
            log(f"rank {rank}: no direct exchange ({L.shems_last_error().decode('utf-8', 'replace') if not mine else 'another rank failed'})")
        if handle.value:
            L.shems_dp_destroy(handle)
        return None
    return handle


class GradSync:
    """Keeps learner replicas identical.  `dist` is `torch.distributed` (initialised) or None.  `native`: a shems_dp communicator
    (native_comm) -- the gradient all-reduces of replay() are then RCCL calls in the update's own stream, issued from native code
    (shems_ddpg_update_dp / shems_train_steps), and torch.distributed only carries the rendezvous, the broadcasts and the scalars."""

    def __init__(self, dist=None, native=None, direct=False):
        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size() > 1) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        self.native = native
        self.direct = bool(direct) and native is not None      # `native` exchanges through peer-mapped inboxes (direct_comm), not RCCL

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
