"""The wide path's vector step alone (65 536 envs, (300, 600)), for a rocprofv3 --kernel-trace --stats pass."""
import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
PKG = "master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"
S = importlib.import_module(PKG); D = importlib.import_module(PKG + ".ddpg")
tab = S.tables.synthetic_table("train", 98)
n = 65536
env = S.ShemsBatch(n, 72, [tab], [S.make_config(98, 0, tab.shape[0])]).use_torch_stream()
ag = D.Agent(seed=1231, hidden=(300, 600))
ring = D.ReplayRing(24000)
ag.populate_memory(env, ring, seed=1)
ag.min_max_buffer(ring, 24000, seed=1)
env.reset_(1, episode=1)
for t in range(60 * 5):
    if t % 60 == 0:
        env.reset_(1, episode=1 + t // 60)
    ag.act_step(env, train=True, tick=t, ring=ring, window=D.RingWindow(ring.pos, 333, (t * 333) % n))
    ring.pushed += 333
torch.cuda.synchronize()
