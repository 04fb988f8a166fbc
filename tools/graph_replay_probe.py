"""Probe: does replaying the five launches of replay() (and the whole vector step) from a captured hipGraph shorten the gaps between the
dependent kernels?  The captured arguments (tick, ADAM powers) are frozen, so the graph's RESULTS are not those of a training run -- this
measures launch structure only."""
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

PKG = "master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"
S = importlib.import_module(PKG)
D = importlib.import_module(PKG + ".ddpg")


def timed(fn, reps=200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


out = {}
for n in (8192, 65536):
    wl = D.TrainWorkload(S, torch, n, seed=7, updates=1)
    for _ in range(20):
        wl.step()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(5):
            wl.agent.replay(wl.ring)
        torch.cuda.synchronize()
        g_upd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_upd, stream=side):
            wl.agent.replay(wl.ring)
        g_step = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_step, stream=side):
            wl.step()
        torch.cuda.synchronize()
        out[f"{n}_update_direct_us"] = timed(lambda: wl.agent.replay(wl.ring))
        out[f"{n}_update_graph_us"] = timed(g_upd.replay)
        out[f"{n}_step_direct_us"] = timed(wl.step)
        out[f"{n}_step_graph_us"] = timed(g_step.replay)
print(json.dumps(out))
