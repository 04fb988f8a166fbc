"""Helper of tests/test_bench_gpu.py: the data-parallel replay() on REAL RCCL streams, as far as one GPU allows.

RCCL refuses two ranks on one device ("Duplicate GPU detected"), so the 2-rank rehearsals run on gloo, whose collectives are
host-synchronous: they cannot show a missing stream dependency.  A ONE-rank NCCL group can: its all-reduce is the identity, but it is
enqueued on RCCL's own stream, `async_op=True` returns a work handle, and `wait()` is a stream-to-stream dependency -- exactly what the
N-GPU run relies on for ordering the collectives against kernels this package launches through ctypes on torch's current stream.
Three learners advance from the same state with the world-2 arithmetic (grad_scale 1/2):
  A  critic all-reduce asynchronous, the actor's E products launched under it, then wait()   (Agent.dp_overlap = True)
  B  every collective in program order                                                        (dp_overlap = False)
  C  no collective at all (the same launches, gradients left as they are)
and must end bit-identical; a lost dependency would let ADAM read a gradient buffer the collective has not released yet.
Prints one JSON line."""
import importlib
import json
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

PKG = "master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"
S = importlib.import_module(PKG)
D = importlib.import_module(PKG + ".ddpg")
P = importlib.import_module(PKG + ".parallel")

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))


class HalfWorld(P.GradSync):
    """The collective call path of a 2-replica run on a 1-rank group: sums are identities, the mean divides by 2."""
    def __init__(self, collective):
        self.dist = dist if collective else None
        self.world, self.rank = 2, 0


times = {}


def run(overlap, collective, steps=40, n=4096):
    wl = D.TrainWorkload(S, torch, n, seed=11, updates=1)
    wl.agent.sync = HalfWorld(collective)
    wl.agent.dp_overlap = overlap
    wl.agent.fused = False
    for _ in range(steps):
        wl.step()
    wl.finish()
    torch.cuda.synchronize()
    # replay() alone, HIP events over groups of 8 (timing.py): what the two collectives cost on this stack before any byte crosses a link
    T = importlib.import_module(PKG + ".timing")
    snap = wl.agent.snapshot()
    times[(overlap, collective)] = T.time_launches(torch, lambda i: wl.agent.replay(wl.ring), 96)[0]
    wl.agent.restore(snap)
    crc = 0
    for name in ("actor", "critic", "actor_t", "critic_t", "m_actor", "v_critic"):
        crc = zlib.crc32(getattr(wl.agent, name).detach().cpu().numpy().tobytes(), crc)
    return crc


out = {"backend": dist.get_backend(), "async_overlap": run(True, True), "in_order": run(False, True), "no_collective": run(False, False)}
out["replay_us"] = {"async_overlap": times[(True, True)], "in_order": times[(False, True)], "no_collective": times[(False, False)]}
print(json.dumps(out), flush=True)
dist.destroy_process_group()
