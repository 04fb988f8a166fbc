"""replay() of the wide path alone, for a rocprofv3 --kernel-trace --stats pass (which kernels the 0.5 ms are made of)."""
import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
PKG = "master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"
S = importlib.import_module(PKG); D = importlib.import_module(PKG + ".ddpg")
tab = S.tables.synthetic_table("train", 98)
env = S.ShemsBatch(4096, 72, [tab], [S.make_config(98, 0, tab.shape[0])]).use_torch_stream()
ag = D.Agent(seed=1231, hidden=(300, 600))
ring = D.ReplayRing(24000)
ag.populate_memory(env, ring, seed=1)
ag.min_max_buffer(ring, 24000, seed=1)
for i in range(300):
    ag.replay(ring)
torch.cuda.synchronize()
