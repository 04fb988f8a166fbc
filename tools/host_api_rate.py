"""PCIe-inclusive rate of the handle API (host arrays in, host arrays out: shems_step with H2D actions and D2H reward/obs every
call), for DESIGN.md.  The bench's `value` never includes this path: it keeps everything resident in HBM."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

PKG = "master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"
S = importlib.import_module(PKG)
tab = S.tables.synthetic_table("train", 98)
for n in (1, 4096, 65536):
    env = S.ShemsBatch(n, 72, [tab], [S.make_config(98, 0, tab.shape[0])])
    rng = np.random.default_rng(0)
    acts = [rng.random((n, 2)).astype(np.float32) for _ in range(8)]
    for rep in range(2):                       # first pass warms up
        env.reset_(123, episode=rep)
        t0 = time.perf_counter()
        for t in range(71):
            env.step_(None, acts[t % 8])
        dt = time.perf_counter() - t0
    print(f"n={n:6d}: {dt / 71 * 1e6:8.1f} us per step! call, {n * 71 / dt / 1e6:8.3f} M env-steps/s "
          f"({n * (8 + 8 + 36) / 1e6:.2f} MB over PCIe per call)")
    env.close()
