"""Time one replay() in its two forms on one GPU: fused (shems_ddpg_update, 5 launches) and split (the calls the data-parallel
path makes, 7 launches, no collective here).  HIP events over back-to-back updates."""
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

PKG = "master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"
S = importlib.import_module(PKG)
D = importlib.import_module(PKG + ".ddpg")
wl = D.TrainWorkload(S, torch, 8192, seed=7, updates=1)
out = {}
for name, fused in (("fused_5_launches", True), ("split_7_launches", False)):
    wl.agent.fused = fused
    for _ in range(20):
        wl.agent.replay(wl.ring)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(200):
        wl.agent.replay(wl.ring)
    e1.record()
    torch.cuda.synchronize()
    out[name + "_us"] = e0.elapsed_time(e1) * 1e3 / 200
print(json.dumps(out))
