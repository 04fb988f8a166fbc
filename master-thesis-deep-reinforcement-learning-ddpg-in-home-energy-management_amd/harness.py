"""Evaluation / tracking harness and on-disk formats.

Reference: inference() (memory_plotting_saving.jl:62-89), write_to_results_file (:167-190),
write_to_tracker_file (:193-212).  File names, column headers and column order are the reference's, so the
thesis notebooks that read `out/tracker/*.csv` and `out/Tracker_Charger.csv` keep working.
"""
from __future__ import annotations

import csv
import ctypes as C
import datetime
import os

import numpy as np

from . import _capi

RESULTS_HEADER = ["index", "c_ev", "EV_target", "EV", "Soc_ev", "rewards", "profit", "discomfort", "penalty", "PV_DE",
                  "B_DE", "GR_DE", "PV_B", "PV_GR", "PV_EV", "B_EV", "GR_EV", "EX_EV", "GR_B", "B_GR", "B", "B_tar", "Soc_b"]
TRACKER_HEADER = ["time", "NUM_EP", "L1", "L2", "BATCH_SIZE", "MEM_SIZE", "MIN_EXP_SIZE", "season", "run", "Job_ID", "seed",
                  "case", "best", "idx", "rewards", "profit", "discomfort", "penalty", "filename"]


def inference(env, agent=None, track=1, num_steps=None, which=0):
    """inference(env; track != 0): one deterministic pass over the data set from reset!(rng = -1)
    (MPS:66-71 -> episode!(..., train=false, track, rng_ep=-1), DDPG.jl:186-242).  track > 0: the actor's
    actions (no noise); track < 0: the rule-based controller action(env, track).  Every env of the batch runs the
    same pass; returns (sum of rewards [N], results [steps][23] float64 of env `which`).

    ONE launch (shems_track_dev: all hours inside the kernel, one workgroup per env) and ONE device-to-host copy of the
    results rows -- no launch, synchronisation or copy per hour."""
    if track > 0 and agent is None:
        raise ValueError("track > 0 needs an agent")
    total, res = _track(env, None if track < 0 else agent.actor, None if track < 0 else agent.s_min, None if track < 0 else agent.s_max,
                        0, track, num_steps, which, hidden=agent.hidden if track > 0 and agent.wide else None)
    return total, res[0]


def inference_many(env, actors, s_min, s_max, num_steps=None, hidden=None):
    """The job's tracking passes in one launch (MAIN:87-105: 40 seeds x {last, best} actor, each a full pass over the data set):
    env e of the batch (len(actors) envs on the same table) runs the pass with actors[e].  actors: [P][129002] float32 (numpy or a
    CUDA tensor); s_min / s_max: [9] (shared) or [P][9].  Returns (sum of rewards [P], results [P][steps][23] float64).
    hidden: (L1, L2) of a network wider than (250, 500) -- the actors then hold that network's own parameter count (ddpg.is_wide)."""
    import torch
    from .ddpg import is_wide, net_size
    wide = hidden is not None and is_wide(hidden)
    na = net_size(9, 2, hidden) if wide else 129002
    dev = torch.device("cuda", torch.cuda.current_device())
    A = torch.as_tensor(np.asarray(actors, np.float32) if not torch.is_tensor(actors) else actors, dtype=torch.float32, device=dev)
    P = A.shape[0]
    if A.dim() != 2 or A.shape[1] != na or env.n != P:
        raise ValueError(f"actors must be [P][{na}] with one env of the batch per actor")
    # one slab row per pass: actor | pad to 16 B | s_min[9] | s_max[9] | pad -- env e finds all three at the same byte stride
    row = -(-na // 4) * 4 + 32
    slab = torch.zeros((P, row), dtype=torch.float32, device=dev)
    slab[:, :na] = A
    o_min, o_max = row - 32, row - 16
    for off, val in ((o_min, s_min), (o_max, s_max)):
        t = torch.as_tensor(np.asarray(val, np.float32) if not torch.is_tensor(val) else val, dtype=torch.float32, device=dev)
        slab[:, off:off + 9] = t if t.dim() == 2 else t[None, :]
    return _track(env, slab[0, :na], slab[0, o_min:o_min + 9], slab[0, o_max:o_max + 9], row * 4, 1, num_steps, -1, keep=slab,
                  hidden=hidden if wide else None)


def _track(env, actor, s_min, s_max, stride, track, num_steps, which, keep=None, hidden=None):
    import torch
    from .ddpg import ActParams, _declare
    _declare()
    n = env.n
    num_steps = env.maxsteps if num_steps is None else int(num_steps)
    env.use_torch_stream()
    env.reset_(-1)
    L = _capi.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    rows = n if which < 0 else 1
    res = torch.empty((rows, num_steps, _capi.NRESULT), dtype=torch.float64, device=dev)
    total = torch.empty(n, dtype=torch.float64, device=dev)
    v = env.view()
    p = None
    if track > 0:
        p = ActParams(actor.data_ptr(), s_min.data_ptr(), s_max.data_ptr(), 0.0, 0.0, 0, 0, 0, 0, 0.0, 0.0, 0.0, None, None)
    if track > 0 and hidden is not None:                        # a wide actor (or one forced onto the wide path): its own flat layout
        L.shems_wide_track_dev.argtypes = [C.POINTER(_capi.View), C.POINTER(ActParams), C.c_int32, C.c_int32, C.c_int64, C.c_int32,
                                           C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.shems_wide_track_dev.restype = C.c_int
        _capi.check(L.shems_wide_track_dev(C.byref(v), C.byref(p), int(hidden[0]), int(hidden[1]), int(stride), num_steps,
                                           C.c_void_p(res.data_ptr()), int(which), C.c_void_p(total.data_ptr()), env._stream()))
    else:
        _capi.check(L.shems_track_dev(C.byref(v), C.byref(p) if p is not None else None, int(stride), 1 if track > 0 else -1, num_steps,
                                      C.c_void_p(res.data_ptr()), int(which), C.c_void_p(total.data_ptr()), env._stream()))
    out, tot = res.cpu().numpy(), total.cpu().numpy()           # the pass's one synchronisation
    env.check_error()
    return tot, out


def results_file_name(job_id, run, ep_len, num_ep, l1, l2, case, rng, idx, best=False, out_dir="out/tracker"):
    """File names of write_to_results_file (MPS:170-187)."""
    if best:
        return os.path.join(out_dir, f"{job_id}_{run}_results_charger_v1_{ep_len}_{num_ep}_{l1}_{l2}_{case}_{rng}_best.csv")
    if isinstance(idx, (int, np.integer)) and idx == num_ep:
        return os.path.join(out_dir, f"{job_id}_{run}_results_charger_v1_{ep_len}_{num_ep}_{l1}_{l2}_{case}_{rng}_{idx}.csv")
    return os.path.join(out_dir, f"{job_id}_{run}_results_{case}_rule_{idx}.csv")


def write_to_results_file(results, path):
    """CSV.write(path, DataFrame(results), header = the 23 names) (MPS:167-190)."""
    results = np.asarray(results, np.float64)
    if results.ndim != 2 or results.shape[1] != len(RESULTS_HEADER):
        raise ValueError("results must be [steps][23]")
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(RESULTS_HEADER)
        for row in results:
            w.writerow([repr(float(x)) for x in row])
    return path


def write_to_tracker_file(results_path, tracker_path="out/Tracker_Charger.csv", *, num_ep, l1=250, l2=500, batch_size=120,
                          mem_size=24000, min_exp_size=24000, season="all", run="eval", job_id=0, seed=0, case="", best=False,
                          idx=0, now=None):
    """Append one row of KPI sums to the overall tracker (MPS:193-212): sums of the rewards / profit / discomfort /
    penalty columns of a results file."""
    with open(results_path, newline="") as fh:
        rd = csv.DictReader(fh)
        sums = {k: 0.0 for k in ("rewards", "profit", "discomfort", "penalty")}
        for row in rd:
            for k in sums:
                sums[k] += float(row[k])
    rows = []
    if os.path.exists(tracker_path):
        with open(tracker_path, newline="") as fh:
            rows = list(csv.reader(fh))[1:]
    now = datetime.datetime.now().isoformat(timespec="milliseconds") if now is None else now
    rows.append([now, num_ep, l1, l2, batch_size, mem_size, min_exp_size, season, run, job_id, seed, case, str(bool(best)).lower(), idx,
                 repr(sums["rewards"]), repr(sums["profit"]), repr(sums["discomfort"]), repr(sums["penalty"]), results_path])
    os.makedirs(os.path.dirname(tracker_path) or ".", exist_ok=True)
    with open(tracker_path, "w", newline="") as fh:            # the reference rewrites the whole file as well
        w = csv.writer(fh)
        w.writerow(TRACKER_HEADER)
        w.writerows(rows)
    return sums
