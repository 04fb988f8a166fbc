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
Round 4 adds the NATIVE exchange (csrc/shems_dp.hip: ncclAllReduce in the update's own stream, issued from shems_ddpg_update_dp /
shems_train_steps) on a one-rank shems_dp communicator, whose grad_scale is 1 / its own world = 1:
  D  Agent.replay() -> shems_ddpg_update_dp, host loop          E  the same inside the native loop (shems_train_loop.dp)
  F  a single replica (fused update, native loop)
D, E and F must end bit-identical (the split form with grad_scale 1 is the fused form's arithmetic).
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
    """The collective call path of a 2-replica run on a 1-rank group: sums are identities, the mean divides by 2 (torch path) or by
    the communicator's own world = 1 (native path)."""
    def __init__(self, collective, native=None):
        self.dist = dist if collective else None
        self.world, self.rank = 2, 0
        self.native = native
        self.direct = False


NATIVE = P.native_comm(dist)


times = {}


def run(overlap, collective, steps=40, n=4096, native=None, loop="host", single=False):
    wl = D.TrainWorkload(S, torch, n, seed=11, updates=1, loop=loop)
    if not single:
        wl.agent.sync = HalfWorld(collective, native)
        wl.agent.dp_overlap = overlap
        wl.agent.fused = False
    wl.steps(steps)
    wl.finish()
    torch.cuda.synchronize()
    # replay() alone, HIP events over groups of 8 (timing.py): what the two collectives cost on this stack before any byte crosses a link
    T = importlib.import_module(PKG + ".timing")
    snap = wl.agent.snapshot()
    times[(overlap, collective, native is not None, single)] = T.time_launches(torch, lambda i: wl.agent.replay(wl.ring), 96)[0]
    wl.agent.restore(snap)
    crc = 0
    for name in ("actor", "critic", "actor_t", "critic_t", "m_actor", "v_critic"):
        crc = zlib.crc32(getattr(wl.agent, name).detach().cpu().numpy().tobytes(), crc)
    return crc


out = {"backend": dist.get_backend(), "async_overlap": run(True, True), "in_order": run(False, True), "no_collective": run(False, False)}
out["native_communicator"] = NATIVE is not None
if NATIVE is not None:
    out["native_host_loop"] = run(False, False, native=NATIVE, loop="host")
    out["native_native_loop"] = run(False, False, native=NATIVE, loop="native")
    out["single_replica"] = run(False, False, loop="native", single=True)
out["replay_us"] = {"async_overlap": times[(True, True, False, False)], "in_order": times[(False, True, False, False)],
                    "no_collective": times[(False, False, False, False)], "native_in_stream": times.get((False, False, True, False)),
                    "single_replica_fused": times.get((False, False, False, True))}
print(json.dumps(out), flush=True)
dist.destroy_process_group()
