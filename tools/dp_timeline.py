"""Kernel timeline of the data-parallel vector step on a one-rank communicator (8 192 envs), for rocprofv3 --kernel-trace:
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/dp_timeline.py torch|native
torch: torch.distributed all_reduce on ProcessGroupNCCL's stream; native: ncclAllReduce in the update's stream (csrc/shems_dp.hip)."""
import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
PKG = "master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"
S = importlib.import_module(PKG); D = importlib.import_module(PKG + ".ddpg"); P = importlib.import_module(PKG + ".parallel")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29549")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
how = sys.argv[1] if len(sys.argv) > 1 else "native"
class HalfWorld(P.GradSync):
    def __init__(self, collective, native=None):
        self.dist = dist if collective else None
        self.world, self.rank = 2, 0
        self.native = native
        self.direct = False
native = P.native_comm(dist) if how == "native" else None
wl = D.TrainWorkload(S, torch, 8192, seed=11, updates=1, loop="native" if how == "native" else "host")
wl.agent.sync = HalfWorld(how == "torch", native); wl.agent.fused = False
wl.steps(400)
torch.cuda.synchronize()
dist.destroy_process_group()
