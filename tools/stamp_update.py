"""In-kernel phase stamps of one DDPG update (diagnostic build, never the product library).

    python <package>/_build.py --stamp                      # libshems_hip_stamp.so (-DSHEMS_STAMP)
    SHEMS_HIP_LIB=build/libshems_hip_stamp.so python tools/stamp_update.py [out.json]

Thread 0 of every workgroup of the five launches records (s_memtime, s_memrealtime) at its phase boundaries.  The table printed
below is, per launch and workgroup role, the median over workgroups of each phase in shader cycles, and the launch's span on the
100 MHz real-time clock (first workgroup's first stamp -> last workgroup's last stamp, and the spread of the start stamps)."""
import ctypes as C
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

PKG = "master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"
S = importlib.import_module(PKG)
D = importlib.import_module(PKG + ".ddpg")
L = S._capi.lib()
assert hasattr(L, "shems_debug_set_stamps"), "not a stamp build: set SHEMS_HIP_LIB to libshems_hip_stamp.so"

wl = D.TrainWorkload(S, torch, 8192, seed=7, updates=1)
buf = torch.zeros(5 * 1024 * 16 * 2, dtype=torch.int64, device="cuda")
for _ in range(30):
    wl.agent.replay(wl.ring)
torch.cuda.synchronize()
S._capi.check(L.shems_debug_set_stamps(C.c_void_p(buf.data_ptr())))
nrep = 20
acc = []
for _ in range(nrep):
    buf.zero_()
    wl.agent.replay(wl.ring)
    torch.cuda.synchronize()
    acc.append(buf.cpu().numpy().reshape(5, 1024, 16, 2).copy())
S._capi.check(L.shems_debug_set_stamps(None))

names = ["K1 k_fwd x3", "K2 k_mid", "K3 k_grad critic", "K4 k_fwd QG", "K5 k_grad actor"]
k5 = lambda w: w                                  # (round 3's merged K4 + K5 launch, whose K5 ids started at 128, was removed in round 4)
roles = {0: lambda w: "tile" if (w % 70) < 64 else "duty", 1: lambda w: "fwd" if w < 64 else "E", 2: lambda w: "W" if w < 128 else "G" if w < 144 else "R",
         3: lambda w: "tile", 4: lambda w: "W" if k5(w) < 128 else "G" if k5(w) < 144 else "R"}
out = {}
a = np.stack(acc)                                  # [rep][launch][wg][stamp][2]
for k in range(5):
    ak = a[:, k]                                   # [rep][wg][stamp][2]
    used = ak[0, :, 0, 1] != 0
    wgs = np.where(used)[0]
    print(f"\n== {names[k]}: {len(wgs)} workgroups")
    rt0 = ak[:, wgs][:, :, 0, 1].astype(np.float64)                      # start stamps (10 ns ticks)
    last = ak[:, wgs][:, :, :, 1].max(-1).astype(np.float64)
    span = (last.max(1) - rt0.min(1)) * 0.01
    skew = (rt0.max(1) - rt0.min(1)) * 0.01
    print(f"   span first-start -> last-end: median {np.median(span):.2f} us; start skew {np.median(skew):.2f} us")
    out[names[k]] = {"span_us": float(np.median(span)), "start_skew_us": float(np.median(skew)), "roles": {}}
    # where the start skew sits: median start offset (us after the launch's first workgroup) per block of 32 workgroup ids
    rel = np.median(rt0 - rt0.min(1, keepdims=True), 0) * 0.01
    print("   start offset by workgroup id (blocks of 32): " + " ".join(f"{rel[i:i + 32].mean():.2f}" for i in range(0, len(wgs), 32)))
    for role in sorted(set(roles[k](int(w)) for w in wgs)):
        sel = [i for i, w in enumerate(wgs) if roles[k](int(w)) == role]
        t = ak[:, wgs[sel]][:, :, :, 0].astype(np.float64)               # shader cycles
        valid = t[0, 0] != 0
        idx = np.where(valid)[0]
        idx = idx[np.argsort(np.median(t[:, :, idx], (0, 1)))]             # in time order (stamp 10 = "first burst staged, now waiting" sits between 0 and 1)
        ph = []
        for i0, i1 in zip(idx[:-1], idx[1:]):
            dt = t[:, :, i1] - t[:, :, i0]
            ph.append((int(i0), int(i1), float(np.median(dt)), float(np.percentile(dt, 95))))
        tot = t[:, :, idx[-1]] - t[:, :, idx[0]]
        rdur = (ak[:, wgs[sel]][:, :, :, 1].max(-1) - ak[:, wgs[sel]][:, :, 0, 1]).astype(np.float64) * 0.01
        print(f"   {role:5s} ({len(sel):3d} wgs): total {np.median(tot):7.0f} cyc = {np.median(rdur):5.2f} us (p95 {np.percentile(rdur, 95):5.2f}); " +
              " ".join(f"[{i0}-{i1}] {m:.0f}" for i0, i1, m, _ in ph))
        out[names[k]]["roles"][role] = {"wgs": len(sel), "total_cycles": float(np.median(tot)), "us": float(np.median(rdur)),
                                        "phases": [{"from": i0, "to": i1, "median_cycles": m, "p95_cycles": p} for i0, i1, m, p in ph]}
# whole update on the real-time clock: K1 first start -> K5 last end
allrt = a[:, :, :, :, 1].astype(np.float64)
allrt[allrt == 0] = np.nan
tot = (np.nanmax(allrt[:, 4].reshape(nrep, -1), 1) - np.nanmin(allrt[:, 0].reshape(nrep, -1), 1)) * 0.01
print(f"\nupdate, first stamp of K1 -> last stamp of K5: median {np.median(tot):.2f} us")
gaps = []
for k in range(4):
    g = (np.nanmin(allrt[:, k + 1, :, 0], 1) - np.nanmax(allrt[:, k].reshape(nrep, -1), 1)) * 0.01
    gaps.append(float(np.median(g)))
print("boundary (last stamp of a launch -> first start stamp of the next):", ["%.2f" % g for g in gaps], "us")
out["update_us"] = float(np.median(tot)); out["boundaries_us"] = gaps
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
