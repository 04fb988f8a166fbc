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
    same pass; returns (sum of rewards [N], results [steps][23] float64 of env `which`)."""
    import torch
    n = env.n
    num_steps = env.maxsteps if num_steps is None else int(num_steps)
    env.use_torch_stream()
    env.reset_(-1)
    L = _capi.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    a = torch.empty((n, 2), dtype=torch.float32, device=dev)
    scaled = torch.empty((n, 2), dtype=torch.float32, device=dev)
    rew = torch.empty(n, dtype=torch.float64, device=dev)
    res = torch.empty((n, _capi.NRESULT), dtype=torch.float64, device=dev)
    total = torch.zeros(n, dtype=torch.float64, device=dev)
    out = np.empty((num_steps, _capi.NRESULT), np.float64)
    v = env.view()
    for t in range(num_steps):
        if track > 0:
            if agent is None:
                raise ValueError("track > 0 needs an agent")
            # act(normalize(s); train=false) on the resident observations, then scale_action, then step!(track = 1)
            agent.act((int(v.obs), n), train=False, out=a)
            _capi.check(L.shems_scale_action_dev(C.c_void_p(a.data_ptr()), n, C.c_void_p(scaled.data_ptr()), env._stream()))
            env.step_dev(scaled, 1, rewards=rew, results=res)
        else:
            _capi.check(L.shems_action_dev(C.byref(v), None, 1, C.c_void_p(a.data_ptr()), env._stream()))
            env.step_dev(a, -1, rewards=rew, results=res)
        total += rew
        out[t] = res[which].cpu().numpy()
    env.check_error()
    return total.cpu().numpy(), out


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
