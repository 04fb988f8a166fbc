"""Vector-step time of the data-parallel call sequence on real RCCL streams (one-rank NCCL group: the collectives move no bytes but cost
their launches and stream dependencies), asynchronous critic all-reduce (dp_overlap) against everything in program order, at sizes where the
GPU (65 536 envs) or possibly the host (8 192) is the bound."""
import importlib, json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
PKG = "master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"
S = importlib.import_module(PKG); D = importlib.import_module(PKG + ".ddpg"); P = importlib.import_module(PKG + ".parallel")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29548")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
class HalfWorld(P.GradSync):
    def __init__(self, collective, native=None):
        self.dist = dist if collective else None
        self.world, self.rank = 2, 0
        self.native = native
        self.direct = False
native = P.native_comm(dist)                                     # a ONE-rank shems_dp communicator: RCCL in the update's own stream, from native code
assert native is not None, "no native communicator"
out = {}
for n in (65536, 8192, 4096):
    for label, overlap, coll, nat, loop in (("torch_async_overlap", True, True, None, "host"), ("torch_in_order", False, True, None, "host"),
                                            ("native_in_stream_host_loop", False, False, native, "host"), ("native_in_stream_native_loop", False, False, native, "native"),
                                            ("no_collective_split_form", False, False, None, "host"), ("single_replica_fused_native_loop", None, False, None, "native")):
        wl = D.TrainWorkload(S, torch, n, seed=11, updates=1, loop=loop)
        if overlap is not None:
            wl.agent.sync = HalfWorld(coll, nat); wl.agent.dp_overlap = overlap; wl.agent.fused = False
        wl.steps(2000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wl.steps(1440)
        torch.cuda.synchronize()
        out[f"{n}_{label}_us_per_step"] = (time.perf_counter() - t0) / 1440 * 1e6
        print(n, label, round(out[f"{n}_{label}_us_per_step"], 2), flush=True)
        del wl
print(json.dumps(out, indent=1))
dist.destroy_process_group()
