"""Distribution of bench.py's roofline-pass group times (groups of 8 vector steps; groups of 8 replay()): why the average of the
difference reads 3 % above its median and above rocprofv3's per-kernel average."""
import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
PKG = "master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"
S = importlib.import_module(PKG); D = importlib.import_module(PKG + ".ddpg")
wl = D.TrainWorkload(S, torch, 65536, seed=1231, updates=1)
for _ in range(6000): wl.step()
torch.cuda.synchronize()
def groups(fn, ng, group=8):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(ng)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        for _ in range(group): fn()
        b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) * 1e3 / group for a, b in ev]
for name, fn in (("step", wl.step), ("replay", lambda: wl.agent.replay(wl.ring)), ("step", wl.step), ("replay", lambda: wl.agent.replay(wl.ring))):
    g = groups(fn, 25)
    print(name, "in order:", " ".join(f"{x:.1f}" for x in g))
    s = sorted(g)
    print(name, "avg %.2f median %.2f min %.2f max %.2f" % (sum(g) / len(g), s[len(s) // 2], s[0], s[-1]))
