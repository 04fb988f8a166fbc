"""Host-side cost of one vector step: how long the Python loop needs to ENQUEUE a step (no synchronisation) in the three forms of
replay() -- fused (one C call), split (the data-parallel call sequence without collectives) and split with real RCCL all-reduce calls on a
one-rank group (identity, but the full host path of torch.distributed).  If the enqueue time exceeds the GPU time of a step, the host is the bound."""
import os, sys, time, importlib
sys.path.insert(0, "/root/repo")
import torch, torch.distributed as dist
PKG = "master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"
S = importlib.import_module(PKG); D = importlib.import_module(PKG + ".ddpg"); P = importlib.import_module(PKG + ".parallel")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
for n in (8192, 65536):
    wl = D.TrainWorkload(S, torch, n, seed=7, updates=1)
    for form in ("fused", "split", "split+nccl"):
        wl.agent.fused = form == "fused"
        if form == "split+nccl":
            class OneRankSync(P.GradSync):                  # takes the collective path on one rank: a 1-rank all-reduce is the identity
                def __init__(self):
                    self.dist = dist; self.world = 1; self.rank = 0
            wl.agent.sync = OneRankSync()
        for _ in range(20): wl.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(60): wl.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"n={n} {form:10s}: host enqueue {1e6*(t1-t0)/60:7.1f} us/step, with drain {1e6*(t2-t0)/60:7.1f} us/step")
x = torch.zeros(129002, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): dist.all_reduce(x)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"dist.all_reduce(129002 f32), 1 rank: host {1e6*(t1-t0)/200:.1f} us/call, with drain {1e6*(t2-t0)/200:.1f}")
