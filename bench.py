#!/usr/bin/env python3
"""bench.py -- env-steps/sec (+ DDPG updates/sec) of the batched shems_LU1 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one VECTOR step of the hot path over all envs of a rank:
  mode "train" (default once the DDPG path is built): actor forward + Gaussian noise + scale_action +
       step! + replay insert for every env, plus `--updates` DDPG updates (BATCH=120) -- the body of
       the reference's episode! loop (DDPG.jl:195-234) for 65 536 households at once;
  mode "env": step! only, on pre-generated random targets (populate_memory's inner loop, MPS:12-24).
Every 72 steps the episode ends and reset! runs (inside the timed region).
Workload = BASELINE.json configs[2]: 65 536 parallel shems_LU1 envs per GPU, Charger98 synthetic table,
72-step episodes (weak scaling: each rank owns its own 65 536-env shard; `--envs` overrides).
Inputs are resident in HBM before the timed region starts.  Prints ONE JSON line on rank 0.

Order of a run: build the workload -> device pre-warm (`--prewarm-s`, default 1.0 s of the same workload, untimed, reported as
`prewarm_steps`; 0 switches it off) -> W warm-up steps -> barrier + synchronize -> EXACTLY K timed steps -> barrier + synchronize.
The pre-warm exists because a GPU that was idle a moment ago runs its first milliseconds below its sustained state: `--steps 20
--warmup 5` straight after start-up read 347 M env-steps/s on a build that sustains 385 M (same command with the pre-warm: 383-385 M).
"""
from __future__ import annotations

import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA
BYTES_PER_ENV_STEP = 92        # obs 36 + action 8 + idx 4 in; obs' 36 + reward 4 + idx 4 out
EP_LEN = 72
# bytes one replay() has to move: 258 003 parameters x (28 B ADAM + 12 B soft update) + one pass over the four networks' weights
UPDATE_ALGORITHMIC_BYTES = 258003 * 40 + 2 * 4 * (129002 + 129001)
BATCH_FOR_UPDATE = 120


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=720)
    ap.add_argument("--warmup", type=int, default=72)
    ap.add_argument("--prewarm-s", type=float, default=5.5,
                    help="run the workload untimed for this many seconds BEFORE the W warm-up steps, so that a short run (--steps 20) is "
                         "measured at the sustained clock and not on a GPU that was idle a moment ago (0 = off)")
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--mode", default="auto", choices=["auto", "train", "policy", "env", "group"])
    ap.add_argument("--group-form", default=None, choices=["throughput", "latency"], help="group mode: the form of the grouped replay() (default: throughput from 16 learners up; csrc/shems_gupd.hip / csrc/shems_ddpg.hip)")
    ap.add_argument("--learners", type=int, default=32, help="group mode: independent learners per GPU (the thesis protocol of many seeds x chargers, SURVEY 8(f) rank 4); --envs must be learners x a multiple of 32")
    ap.add_argument("--updates", type=int, default=1, help="DDPG updates per vector step (train mode)")
    ap.add_argument("--overlap", nargs="?", const="pipelined", default=None, choices=["pipelined", "exact"],
                    help="train mode: run replay() on a second stream under the act/step launch (DESIGN.md 5b; not the headline configuration).  "
                         "pipelined: replay(t) samples the ring as it stood before step t's inserts; exact: the inserting envs are stepped first and "
                         "replay(t) runs under the rest of the batch -- the ordered loop's bytes")
    ap.add_argument("--loop", default=None, choices=["native", "host"], help="train mode: native = k vector steps enqueued by one shems_train_steps call "
                    "(default for a single replica on the tuned kernels), host = one foreign call per launch from Python")
    ap.add_argument("--mixed", action="store_true", help="train mode: BASELINE config 5 (10 charger profiles x discomfort-weight sweep, per-env configs)")
    ap.add_argument("--scaled-replay", action="store_true", help="train mode: SURVEY 8(d)'s optional replay mode: ring capacity 72 x envs, every env's transition inserted each step (177 B per env-step) instead of MEM_SIZE = 24 000 with a rotating window of 333 envs")
    ap.add_argument("--hidden", default="250x500", help="train mode: Dense widths L1xL2 of actor and critic; the headline is the tuned 250x500, 300x600 is the reference grids' wider point (layer-by-layer path, csrc/shems_wide.hip), smaller ones run zero-padded")
    ap.add_argument("--group-window", type=int, default=None, help="group mode: transitions each learner remembers per vector step.  Default: the rotating window of "
                    "min(envs per learner, MEM_SIZE / 72) households; 1 = the reference's ratio, ONE remembered transition per replay() (DDPG.jl:229-233), household 0 of "
                    "the learner's block")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--also", default="auto", choices=["auto", "on", "off"], help="after the headline run, time the other configurations BASELINE.json names and the "
                    "learner-group shapes on the same box and report them as compact sub-records under \"also\" (auto: only for the default headline workload -- "
                    "train mode, 65 536 envs, one GPU)")
    ap.add_argument("--also-which", default=None, help="comma-separated subset of the sub-record names (tests)")
    return ap.parse_args()


# The other workloads this repo quotes numbers for, timed by the default run so that every claimed figure stands under the driver's clock
# (a sixth record added for the thesis-exact ratio): BASELINE configs[1] (4 096 envs), the configs[3] shard (8 192 envs per GPU), configs[4] on one GPU (65 536
# envs x 10 charger profiles x discomfort-weight sweep), and the learner groups of SURVEY 8(f) rank 4 at the thesis protocol's width
# (40 seeds x 10 chargers, RL-SHEMS_bs_scheduler_1179_08_on_01-98.sh:67-87) and at 32 learners x 2 048 households.
ALSO = [
    ("config2_4096_envs", dict(kind="train", envs=4096, steps=720, warmup=72)),
    ("config4_shard_8192_envs", dict(kind="train", envs=8192, steps=720, warmup=72)),
    ("config5_mixed_65536_envs", dict(kind="train", envs=65536, mixed=True, steps=360, warmup=72)),
    ("group_400x128", dict(kind="group", learners=400, envs=51200, mixed=True, steps=72, warmup=8)),
    ("group_32x2048", dict(kind="group", learners=32, envs=65536, mixed=True, steps=144, warmup=16)),
    # the thesis protocol at the reference's update-to-data ratio: ONE remembered transition per learner-update (DDPG.jl:229-233) -- household 0 of a
    # 32-household block (the smallest tile of the fused kernel; the reference's learner owns one household)
    ("group_400x32_one_transition_per_update", dict(kind="group", learners=400, envs=12800, mixed=True, window=1, steps=72, warmup=8)),
]


def run_also(S, torch, which=None, prewarm_s=0.7, log=None):
    """Each entry of ALSO as a compact record: the same order as the headline (build -> untimed pre-warm -> W warm-up steps -> synchronize ->
    exactly K steps -> synchronize -> roofline pass), one after the other on this process's GPU, every workload freed before the next is built.
    A workload that fails leaves {"error": ...} in its place and the others still run."""
    out = {}
    for name, spec in ALSO:
        if which is not None and name not in which:
            continue
        t_build = time.perf_counter()
        wl = None
        try:
            if spec["kind"] == "train":
                wl = importlib.import_module(PKG + ".ddpg").TrainWorkload(S, torch, spec["envs"], seed=1231, updates=1, mixed=spec.get("mixed", False))
                upd_per_step = 1
            else:
                wl = importlib.import_module(PKG + ".group").GroupWorkload(S, torch, spec["envs"], spec["learners"], seed=1231, mixed=spec.get("mixed", False),
                                                                           window=spec.get("window"))
                upd_per_step = spec["learners"]
            many = getattr(wl, "steps", None)
            run = many if many is not None else (lambda k: [wl.step() for _ in range(k)])
            torch.cuda.synchronize()
            setup_s = time.perf_counter() - t_build
            tp, pre = time.perf_counter(), 0
            while time.perf_counter() - tp < prewarm_s:
                run(24)
                pre += 24
                torch.cuda.synchronize()
            run(spec["warmup"])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(spec["steps"])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            wl.finish()
            k = wl.kernel_pass(64)
            ach = k["algorithmic"] / (k["avg_us"] * 1e-6) / (1e9 if k["unit"] == "GB/s" else 1e12)
            rec = {"workload": (f"{spec['learners']} learners x {spec['envs'] // spec['learners']} households" if spec["kind"] == "group" else f"{spec['envs']} envs")
                               + (", 10 charger profiles" if spec.get("mixed") else ", Charger98") + f", mode={spec['kind']}",
                   "value": spec["envs"] * spec["steps"] / dt, "unit": "env-steps/s", "ms_per_step": dt / spec["steps"] * 1e3, "steps": spec["steps"],
                   "warmup": spec["warmup"], "prewarm_steps": pre, "updates_per_sec": upd_per_step * spec["steps"] / dt,
                   "roofline": {"kernel": k["kernel"], "kernel_avg_us": k["avg_us"], "bound": k["bound"], "achieved": ach, "peak": k["peak"], "unit": k["unit"],
                                "frac": ach / k["peak"]},
                   "setup_s": setup_s, "wall_s": time.perf_counter() - t_build}
            if k.get("other_kernel") is not None:
                rec["roofline"]["other_kernel"] = k["other_kernel"]
            if k.get("back_to_back_us") is not None:
                rec["roofline"]["kernel_back_to_back_us"] = k["back_to_back_us"]
            upd_us = getattr(wl, "update_us", None)
            if upd_us:
                rec["update_us"] = upd_us
            if spec["kind"] == "group":
                rec["update_form"] = wl.group.form
                rec["w2_layout"] = "tiled" if wl.group.tiled else "flux"
                rec["replay_window_envs_per_step"] = wl.win
            out[name] = rec
        except Exception as e:                      # noqa: BLE001 -- the headline line must still be printed
            out[name] = {"error": f"{type(e).__name__}: {e}"[:400], "wall_s": time.perf_counter() - t_build}
        finally:
            if wl is not None:
                try:
                    if hasattr(wl, "close"):
                        wl.close()
                    wl.env.close()
                except Exception:                   # noqa: BLE001
                    pass
            wl = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
        if log:
            log(f"also: {name}: {json.dumps(out[name])[:300]}")
    return out


class EnvWorkload:
    """step! only: random SoC targets, one k_step launch per vector step."""

    name = "env"
    dtype = "f32/f64"

    def __init__(self, S, torch, n, seed):
        self.S, self.torch, self.n = S, torch, n
        self.tab = S.tables.synthetic_table("train", 98)
        self.env = S.ShemsBatch(n, EP_LEN, [self.tab], [S.make_config(98, 0, self.tab.shape[0])],
                                device=torch.cuda.current_device()).use_torch_stream()
        g = torch.Generator(device="cuda").manual_seed(seed)
        self.actions = [torch.rand((n, 2), generator=g, device="cuda", dtype=torch.float32) for _ in range(8)]
        self.rew32 = torch.empty(n, dtype=torch.float32, device="cuda")
        self.seed = seed
        self.t = 0
        self.episode = 0
        self.env.reset_(seed, episode=0)

    def step(self):
        if self.t and self.t % EP_LEN == 0:
            self.episode += 1
            v = self.env.view()
            self.S._capi.check(self.S._capi.lib().shems_reset_seeded_dev(C.byref(v), self.seed, self.episode, self.env._stream()))
        self.env.step_dev(self.actions[self.t % 8], 0, rewards_f32=self.rew32)
        self.t += 1

    def finish(self):
        self.env.check_error()

    def _reset_dev(self, episode):
        v = self.env.view()
        self.S._capi.check(self.S._capi.lib().shems_reset_seeded_dev(C.byref(v), self.seed, episode, self.env._stream()))

    def kernel_pass(self, reps):
        """HIP-event timing of the dominant kernel alone (events on the launch stream)."""
        T = importlib.import_module(PKG + ".timing")
        reset = lambda g, i: self._reset_dev(1000 + i) if g % 8 == 0 else None      # 64 launches < one episode; enqueued, no host sync
        avg, med, reps = T.time_launches(self.torch, lambda i: self.env.step_dev(self.actions[i % 8], 0, rewards_f32=self.rew32),
                                         reps, before_group=reset)
        return dict(kernel="shems::k_step", avg_us=avg, median_us=med, launches=reps,
                    bound="hbm", algorithmic=BYTES_PER_ENV_STEP * self.n, unit="GB/s", peak=HBM_PEAK_GBS)

    def extra(self):
        return {}


class PolicyWorkload(EnvWorkload):
    """act + step! + remember fused (no DDPG update): one k_act launch per vector step."""

    name = "policy"
    dtype = "f32"

    def __init__(self, S, torch, n, seed):
        super().__init__(S, torch, n, seed)
        D = importlib.import_module(PKG + ".ddpg")
        self.D = D
        self.agent = D.Agent(seed=seed)
        st = self.env.state
        self.agent.set_norm(st.min(0), st.max(0))
        self.ring = D.ReplayRing(D.MEM_SIZE)
        self.win_count = min(n, D.MEM_SIZE // EP_LEN)          # rotating window of 333 envs
        self.pushed = 0

    def _launch(self, tick):
        win = self.D.RingWindow(self.pushed % self.ring.capacity, self.win_count, (tick * self.win_count) % self.n)
        self.agent.act_step(self.env, train=True, tick=tick, rewards_f32=self.rew32, ring=self.ring, window=win)
        self.pushed += self.win_count

    def step(self):
        if self.t and self.t % EP_LEN == 0:
            self.episode += 1
            v = self.env.view()
            self.S._capi.check(self.S._capi.lib().shems_reset_seeded_dev(C.byref(v), self.seed, self.episode, self.env._stream()))
        self._launch(self.t)
        self.t += 1

    def kernel_pass(self, reps):
        T = importlib.import_module(PKG + ".timing")
        reset = lambda g, i: self._reset_dev(1000 + i) if g % 8 == 0 else None
        avg, med, reps = T.time_launches(self.torch, self._launch, min(reps, 200), before_group=reset)
        flops = 2 * (9 * 250 + 250 * 500 + 500 * 2) * self.n       # 256 500 FLOP per env-step
        return dict(kernel=self.D.act_kernel_name(self.n), avg_us=avg, median_us=med, launches=reps,
                    bound="mfma", algorithmic=flops, unit="TFLOP/s", peak=MFMA_F32_PEAK_TFLOPS,
                    algorithmic_bytes=self.D.act_algorithmic_bytes(self.n, self.win_count))


def cpu_baseline(n_envs, mode, updates, scale=1.0):
    """The CPU oracle (`kind: port`: C restatement of shems_LU1.jl + restatement of the DDPG learner) timed on this box's host
    cores, on bounded samples of the same workload (about 20 s in all).  No timed region contains a per-env foreign call: the
    env batch is stepped, reset and read with ONE C call each (orc_batch_step[_omp] with obs_out, orc_batch_reset).

      value / cores            the FULL vector step (act + Gaussian noise + scale_action + step! + remember [+ `updates` x replay()])
                               on ALL usable host cores: one OpenMP region per step takes blocks of 32 households through actor,
                               noise, scale_action and step! (oracle/shems_policy_omp.c); NumPy learner with the BLAS thread
                               count that is fastest for B = 120 on this box (measured here, reported)
      one_thread_value         the same step on ONE thread (C env, NumPy actor + learner, BLAS capped at 1) on a 4096-env slice
      env_only_*               step! alone, one thread / all cores (whole episodes per OpenMP region)
      update_only_per_sec      replay() alone at each BLAS thread count tried
    scale shrinks every leg's time budget (tests)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_c
    import ddpg_oracle as DO
    from threadpoolctl import threadpool_limits
    S = importlib.import_module(PKG)
    tab = S.tables.synthetic_table("train", 98)
    rng = np.random.default_rng(0)
    L = oracle_c.lib()
    cores = oracle_c.usable_cpus()                 # affinity mask / cgroup quota, not the CPUs merely visible
    oracle_c.set_threads(cores)

    def env_batch(n):
        b = oracle_c.Batch(n, EP_LEN, tab, oracle_c.profile(98))
        idx0 = rng.integers(1, tab.shape[0] - EP_LEN + 1, n)
        soc0 = (rng.random(n) * 6.75).astype(np.float32)
        return b, idx0, soc0

    # (a) step! only, 1 thread and all cores (OpenMP) -- the env half of the path
    n = min(n_envs, 16384)
    b, idx0, soc0 = env_batch(n)
    acts = [rng.random((n, 2)).astype(np.float32) for _ in range(4)]
    rew = np.empty(n)
    steps, t_step = 0, 0.0
    while t_step < 2.0 * scale:
        b.reset(False, idx0, soc0)                  # one C call, not timed
        t0 = time.perf_counter()
        for t in range(EP_LEN):
            L.orc_batch_step(b.ptr, n, acts[t % 4].ctypes.data, 0, rew.ctypes.data, None, None)
        t_step += time.perf_counter() - t0
        steps += n * EP_LEN
    env_one = steps / t_step
    # all host cores: each OpenMP thread carries its envs through whole 72-step episodes inside one parallel region
    # (orc_batch_episode_omp); one untimed episode warms the thread pool
    sets = np.stack(acts)
    b.reset(False, idx0, soc0)
    b.episode_omp(sets, EP_LEN)
    steps_all, t_all = 0, 0.0
    while t_all < 1.5 * scale:
        b.reset(False, idx0, soc0)
        t0 = time.perf_counter()
        rc, _ = b.episode_omp(sets, EP_LEN)
        t_all += time.perf_counter() - t0
        steps_all += n * EP_LEN
        assert rc == 0
    env_all = steps_all / t_all
    # BASELINE config 1 / B1: ONE env, rule-based 72-step episodes (reset!(env; rng=-1), action(env, track), step!), one thread
    one = oracle_c.Batch(1, EP_LEN, tab, oracle_c.profile(98))
    nrep, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 0.5 * scale:
        one.reset(True)
        one.rule_episode(0, EP_LEN)
        nrep += 1
    rule_us = (time.perf_counter() - t0) / (nrep * EP_LEN) * 1e6
    out = {"env_only_value": env_one, "env_only_all_cores_value": env_all, "all_cores": cores,
           "config1_rule_episode_us_per_step": rule_us}
    if mode == "env":
        out.update({"value": env_all, "unit": "env-steps/s", "cores": cores, "kind": "port", "one_thread_value": env_one,
                    "sample": f"oracle/shems_oracle.c step! only, {n} envs x {EP_LEN}-step episodes: {steps_all} env-steps in {t_all:.1f} s "
                              f"on all cores (OpenMP), {steps} in {t_step:.1f} s on one thread"})
        return out

    cap = 24000
    actor, critic = DO.init_params(1231, 9, 2, 0), DO.init_params(1231, 11, 1, 1)

    def make_ring(st):
        ring = dict(s=np.zeros((cap, 9), np.float32), a=np.zeros((cap, 2), np.float32), r=np.zeros(cap, np.float32),
                    s2=np.zeros((cap, 9), np.float32), d=np.zeros(cap, bool))
        ring["s"][:] = st[rng.integers(0, len(st), cap)]
        ring["s2"][:] = ring["s"]
        return ring

    def upd_rate(learner, ring, nrep):
        t1 = time.perf_counter()
        for u in range(nrep):
            i = DO.sample_indices(2, u, 120, cap)
            learner.replay(ring["s"][i], ring["a"][i], ring["r"][i], ring["s2"][i], ring["d"][i])
        return nrep / (time.perf_counter() - t1)

    def full_step_loop(n, budget, env_step, policy, blas_threads, fused=None):
        """`budget` seconds of whole vector steps on an n-env batch; returns (vector steps, seconds)."""
        b, idx0, soc0 = env_batch(n)
        b.reset(False, idx0, soc0)
        s, s2 = b.state(), np.empty((n, 9), np.float32)
        s_min, s_max = s.min(0), s.max(0)
        learner = DO.Learner(actor, critic, s_min, s_max)
        ring = make_ring(s)
        win = min(n, cap // EP_LEN)
        rew = np.empty(n)
        sc = np.empty((n, 2), np.float32)
        with threadpool_limits(limits=blas_threads):
            nstep, pos, t0 = 0, 0, time.perf_counter()
            while time.perf_counter() - t0 < budget or nstep < 2:
                if fused is not None:
                    a = sc
                    fused(b.ptr, n, learner.actor, s, s_min, s_max, nstep, a, rew, s2)             # everything up to s', one call
                else:
                    a = policy(learner.actor, s, s_min, s_max, nstep)                              # actor + Gaussian noise + clamp
                    L.orc_scale_actions(a.ctypes.data, a.size, sc.ctypes.data)                     # scale_action, one call
                    env_step(b.ptr, n, sc.ctypes.data, 0, rew.ctypes.data, s2.ctypes.data)         # step!, s' written by the same call
                sl = (pos + np.arange(win)) % cap                                                  # remember(): rotating window
                ring["s"][sl], ring["a"][sl], ring["r"][sl], ring["s2"][sl] = s[:win], a[:win], rew[:win], s2[:win]
                pos += win
                if mode == "train":
                    for u in range(updates):
                        i = DO.sample_indices(1, nstep * 8 + u, 120, cap)
                        learner.replay(ring["s"][i], ring["a"][i], ring["r"][i], ring["s2"][i], ring["d"][i])
                s, s2 = s2, s
                nstep += 1
                if nstep % (EP_LEN - 1) == 0:
                    b.reset(False, idx0, soc0)                                                     # one C call
                    b.state(out=s)
            return nstep, time.perf_counter() - t0, learner, ring

    # (b) ONE thread: C env, NumPy actor and learner, BLAS capped at 1, 4096-env slice
    np_policy = lambda act_p, s, lo, hi, k: DO.act(act_p, s, lo, hi, True, seed=1, tick=k)
    step1 = lambda ptr, n, a, tm, r, o: L.orc_batch_step(ptr, n, a, tm, r, o, None)
    n1 = min(n_envs, 4096)
    nstep1, dt1, learner, ring = full_step_loop(n1, 6.0 * scale, step1, np_policy, 1)
    one_thread = n1 * nstep1 / dt1
    if mode == "train":
        # replay() alone at several BLAS thread counts: at B = 120 the matrices are small and more threads are
        # not faster; the all-cores leg below uses whichever count wins here
        rates = {}
        for th in sorted({1, min(4, cores), min(16, cores)}):
            with threadpool_limits(limits=th):
                upd_rate(learner, ring, 5)
                rates[th] = upd_rate(learner, ring, max(3, int(30 * scale)))
        best_blas = max(rates, key=rates.get)
        out["update_only_per_sec"] = rates[1]
        out["update_only_per_sec_by_blas_threads"] = {str(k): v for k, v in rates.items()}
    else:
        best_blas = 1

    # (c) ALL cores: the whole vector step (normalize + actor + noise + clamp + scale_action + step!) in ONE OpenMP region per step
    # (oracle/shems_policy_omp.c: a thread takes blocks of 32 households through all of it), learner as above with the best BLAS
    # thread count.  `policy` = None tells the loop that the env step is part of the fused call.
    def fused_step(ptr, n, learner_actor, s, lo, hi, k, a_out, rew, s2):
        return L.orc_policy_step_omp(ptr, n, learner_actor.ctypes.data, lo.ctypes.data, hi.ctypes.data, 0.1, 1, k & 0xFFFFFFFF, 1,
                                     s.ctypes.data, a_out.ctypes.data, rew.ctypes.data, s2.ctypes.data)
    # Two ways to spend the cores, both timed, the faster one reported as `value`: (i) the fused C / OpenMP step above (hand-written
    # AVX2 dense layers, W2 resident in each core's L2); (ii) the library route -- actor forward + noise on torch's CPU thread pool
    # (its BLAS picks the widest vectors the host has: AVX-512 on the GPU boxes), then scale_action and the OpenMP env step.
    import torch
    torch.set_num_threads(cores)

    def torch_policy(act_p, s, lo, hi, k):
        with torch.no_grad():
            W1, b1, W2, b2, W3, b3 = [torch.from_numpy(x) for x in DO.split(act_p, 9, 2)]
            x = (torch.from_numpy(s) - torch.from_numpy(lo)) / ((torch.from_numpy(hi) - torch.from_numpy(lo)) + 1e-8)
            h = torch.relu_(torch.addmm(b1, x, W1))
            h = torch.relu_(torch.addmm(b2, h, W2))
            a = torch.tanh_(torch.addmm(b3, h, W3))
            a.add_(torch.randn_like(a), alpha=0.1).clamp_(-1.0, 1.0)
            return a.numpy()
    stepN = lambda ptr, n, a, tm, r, o: L.orc_batch_step_omp(ptr, n, a, tm, r, o)
    nN = n_envs
    legs = {}
    legs["fused C/OpenMP step (shems_policy_omp.c)"] = full_step_loop(nN, 3.5 * scale, None, None, best_blas, fused=fused_step)[:2]
    legs["torch-CPU actor + OpenMP C env"] = full_step_loop(nN, 3.5 * scale, stepN, torch_policy, best_blas)[:2]
    rate = {k: nN * v[0] / v[1] for k, v in legs.items()}
    best = max(rate, key=rate.get)
    nstepN, dtN = legs[best]
    all_cores = rate[best]
    out.update({"value": all_cores, "unit": "env-steps/s", "cores": cores, "kind": "port",
                "updates_per_sec": (updates * nstepN / dtN) if mode == "train" else None,
                "one_thread_value": one_thread, "one_thread_updates_per_sec": (updates * nstep1 / dt1) if mode == "train" else None,
                "visible_cpus": os.cpu_count(), "learner_blas_threads": best_blas, "all_cores_route": best,
                "all_cores_routes": {k: float(v) for k, v in rate.items()},
                "sample": f"CPU oracle, the same vector step (mode={mode}, {updates} update/step).  All {cores} usable host threads, faster of two "
                          f"routes = {best} + NumPy learner (BLAS x{best_blas}) on {nN} envs: {nstepN} vector "
                          f"steps = {nN * nstepN} env-steps in {dtN:.1f} s.  One thread: C env + NumPy actor/learner on a {n1}-env slice: "
                          f"{nstep1} vector steps = {n1 * nstep1} env-steps in {dt1:.1f} s.  No per-env foreign call in any timed loop."})
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` with no rendezvous in the environment: this process becomes the launcher.  It has made NO
    GPU call (torch is not even imported here) and starts N fresh children of this same script, one rank per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, exactly what `-m torch.distributed.run` would set.
    Rank 0's single JSON line goes to this process's stdout; the exit code is non-zero if any rank failed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this host driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    rc = 0
    try:
        pending = set(range(args.gpus))
        while pending:
            for r in list(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    rc = rc or code
                    print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr, flush=True)
                    for q in pending:                            # a failed rank leaves the others in a collective: stop them
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    json_fd = None
    if world > 1:
        # RCCL prints its version banner to STDOUT at communicator creation (seen on this image: "RCCL version : ...", "Librccl path : ...").
        # The contract is ONE JSON line on stdout, so everything any library writes to descriptor 1 goes to stderr for the life of the
        # process, and rank 0 writes its line to the saved descriptor.
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
    import torch
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Rehearsal knobs for a one-GPU box (never set by the driver): every rank on device 0, gloo instead of RCCL.
    if os.environ.get("SHEMS_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SHEMS_BENCH_BACKEND", "nccl")             # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    S = importlib.import_module(PKG)

    mode = args.mode
    train_mod = None
    if mode in ("auto", "train"):
        try:
            train_mod = importlib.import_module(PKG + ".ddpg")
            if not hasattr(train_mod, "TrainWorkload"):
                train_mod = None
        except ImportError:
            train_mod = None
        if train_mod is None and mode == "train":
            raise SystemExit("train mode requested but the DDPG path is not built")
        mode = "train" if train_mod is not None else "env"
    if mode == "group":
        wl = importlib.import_module(PKG + ".group").GroupWorkload(S, torch, args.envs, args.learners, seed=1231 + 1000 * rank, mixed=args.mixed, form=args.group_form, window=args.group_window)
    elif mode == "policy":
        wl = PolicyWorkload(S, torch, args.envs, seed=123 + rank)
    elif mode == "train":
        wl = train_mod.TrainWorkload(S, torch, args.envs, seed=1231 + rank, updates=args.updates, dist=dist, overlap=args.overlap or False, loop=args.loop, mixed=args.mixed, scaled_replay=args.scaled_replay,
                                     hidden=tuple(int(x) for x in args.hidden.lower().split("x")))
    else:
        wl = EnvWorkload(S, torch, args.envs, seed=123 + rank)

    def run_steps(k):
        # the train workload enqueues k vector steps with ONE foreign call (shems_train_steps: the hour loop of episode! in native code);
        # the other workloads are one call per launch
        many = getattr(wl, "steps", None)
        if many is not None:
            return many(k)
        for _ in range(k):
            wl.step()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    t_gpu_section_start = time.perf_counter()
    # Device pre-warm (untimed, reported as prewarm_steps): a GPU that has been idle runs its first milliseconds well below the
    # sustained state (clock ramp, cold instruction / TLB / L2 state): 20 steps right after start-up measured 353 M env-steps/s where
    # the same build sustains 388 M.  The contract's W warm-up steps and K timed steps follow unchanged.  Ranks agree on the count.
    prewarm_steps = 0
    if args.prewarm_s > 0:
        tp = time.perf_counter()
        while True:
            run_steps(50)
            prewarm_steps += 50
            torch.cuda.synchronize()
            more = time.perf_counter() - tp < args.prewarm_s
            if dist is not None:
                flag = torch.tensor([1.0 if more else 0.0], device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                more = bool(flag.item() > 0.5)
            if not more:
                break
    run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    wl.finish()
    t_gpu_section_end = time.perf_counter()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # Who took part: one entry per rank (device index, uuid / PCI id, host, pid) and the size of the collective group, so that "did
    # RCCL see N distinct GPUs" is answerable from the JSON line alone.
    census = None
    if dist is not None:
        import socket
        pr = torch.cuda.get_device_properties(torch.cuda.current_device())
        me = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device_index": torch.cuda.current_device(),
              "uuid": str(getattr(pr, "uuid", None)), "pci_bus_id": getattr(pr, "pci_bus_id", None), "name": pr.name,
              "host": socket.gethostname(), "pid": os.getpid()}
        gathered = [None] * world
        dist.all_gather_object(gathered, me)
        census = {"backend": dist.get_backend(), "rccl_ranks": dist.get_world_size(), "ranks": gathered,
                  "distinct_devices": len({(g["host"], g["uuid"], g["pci_bus_id"], g["device_index"]) for g in gathered})}

    roof = None
    cpu = None
    all_ranks_pass = dist is not None and mode == "train"      # the data-parallel pass issues collectives: every rank runs it
    t_pass0 = time.perf_counter()
    if rank == 0 or all_ranks_pass:
        k = wl.kernel_pass(max(200, min(args.steps, 500)))
    t_pass = time.perf_counter() - t_pass0
    t_cpu = 0.0
    also, t_also = None, 0.0
    headline_default = (mode == "train" and world == 1 and args.envs == 65536 and args.updates == 1 and not args.mixed and not args.scaled_replay
                        and not args.overlap and args.hidden.lower() == "250x500")
    if args.also == "on" or (args.also == "auto" and headline_default):
        if world != 1:
            raise SystemExit("--also on: the sub-records are one-GPU workloads")
        # the headline workload's buffers are kept (its learner checksum is read below); the sub-records fit beside them (< 8 GB)
        t_also0 = time.perf_counter()
        also = run_also(S, torch, which=None if args.also_which is None else set(args.also_which.split(",")), log=lambda m: print(m, file=sys.stderr, flush=True))
        t_also = time.perf_counter() - t_also0
    if rank == 0:
        achieved = k["algorithmic"] / (k["avg_us"] * 1e-6) / (1e9 if k["unit"] == "GB/s" else 1e12)
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")       # rocprofv3 --pmc passes, see profiles/README.md
        if os.path.exists(pmc):
            rec = json.load(open(pmc)).get(mode, {})
            if mode == "group":
                # counters of the grouped update (all eight launches of one grouped replay()), only for the shape they were collected on
                if (rec.get("learners") == args.learners and rec.get("envs_per_gpu") == args.envs and k["kernel"].startswith("grouped replay()")
                        and rec.get("form") == getattr(wl.group, "form", None)
                        and rec.get("w2_layout", "flux") == ("tiled" if getattr(wl.group, "tiled", False) else "flux")):
                    traffic = rec.get("bytes_as_read")
                    traffic_src = ("profiles/pmc_traffic.json (static: FETCH_SIZE + WRITE_SIZE as read over the eight launches of one grouped replay(), separate "
                                   "rocprofv3 --pmc passes of this command, " + str(rec.get("round")) + "; with the guide's 2x FETCH correction: "
                                   + str(round(rec.get("bytes_fetch_x2", 0) / 1e9, 2)) + " GB)")
            elif rec.get("envs_per_gpu") == args.envs and args.hidden.lower() == "250x500":      # counters of the headline kernel only
                traffic = rec.get("hbm_bytes_per_launch")
                # PMC counters cannot be read from inside this process: the figure is the committed result of separate
                # `rocprofv3 --pmc` passes over this same command, not a measurement of the run that prints it
                traffic_src = traffic_src or "profiles/pmc_traffic.json (static: separate rocprofv3 --pmc passes of this command, " + str(rec.get("round", "r01")) + ")"
        roof = {"bound": k["bound"], "achieved": achieved, "peak": k["peak"], "unit": k["unit"],
                "frac": achieved / k["peak"], "traffic": traffic, "traffic_source": traffic_src, "kernel": k["kernel"],
                "kernel_avg_us": k["avg_us"], "kernel_median_us": k["median_us"], "launches": k["launches"],
                "algorithmic_per_launch": k["algorithmic"],
                "timing": k.get("method", "HIP events over back-to-back groups of 8 launches")}
        # north_star asks for "rocprof HBM GB/s": the counter bytes per launch over the launch's time; next to it the bytes the launch
        # HAS to move and their ratio (well above 1 = wasted re-reads; an MFMA-bound kernel sits far below the HBM roof either way)
        roof["algorithmic_bytes"] = k.get("algorithmic_bytes")
        roof["hbm_gbs"] = traffic / (k["avg_us"] * 1e-6) / 1e9 if traffic else None
        roof["traffic_ratio"] = traffic / k["algorithmic_bytes"] if traffic and k.get("algorithmic_bytes") else None
        if k.get("avg_us_is"):
            roof["kernel_avg_us_is"] = k["avg_us_is"]
        if k.get("back_to_back_us") is not None:
            roof["kernel_back_to_back_us"] = k["back_to_back_us"]      # the same launch timed directly, back to back: the cross-check of the difference
            roof["frac_back_to_back"] = k["algorithmic"] / (k["back_to_back_us"] * 1e-6) / 1e12 / k["peak"] if k["unit"] == "TFLOP/s" else None
        for extra in ("hbm_frac_of_8tbs", "algorithmic_gbs", "per_learner_update_us", "launches_per_update", "other_kernel"):
            if k.get(extra) is not None:
                roof[extra] = k[extra]
        # the learner's update next to the fused kernel (train mode): five dependent launches of 5-8 us at B = 120, latency-bound -- its
        # MFMA fraction is reported all the same (north_star: "MFMA utilisation vs gfx950 peak"), with the counter bytes of the
        # committed rocprofv3 --pmc passes against the bytes an update has to move (10.3 MB of ADAM + soft-update traffic
        # + one pass over the four networks' weights)
        upd_us = getattr(wl, "update_us", None)
        if mode == "train" and upd_us and args.hidden.lower() == "250x500":
            mflop = 2.565 * BATCH_FOR_UPDATE                 # 2.565 MFLOP x B
            ur = {"update_us": upd_us, "launches": 5 if world == 1 else 7, "mflop": mflop, "achieved_tflops": mflop * 1e6 / (upd_us * 1e-6) / 1e12,
                  "peak_tflops": 157.3, "frac": mflop * 1e6 / (upd_us * 1e-6) / 1e12 / 157.3, "bound": "latency (five dependent launches; MFMA nominally)",
                  "algorithmic_bytes": UPDATE_ALGORITHMIC_BYTES, "counter_bytes": None, "counter_bytes_fetch_x2": None, "traffic_ratio": None,
                  "traffic_source": None}
            if os.path.exists(pmc):
                urec = json.load(open(pmc)).get("update", {})
                if urec:
                    ur["counter_bytes"] = urec.get("bytes_as_read")
                    ur["counter_bytes_fetch_x2"] = urec.get("bytes_fetch_x2")
                    ur["traffic_ratio"] = [urec["bytes_as_read"] / UPDATE_ALGORITHMIC_BYTES, urec["bytes_fetch_x2"] / UPDATE_ALGORITHMIC_BYTES] if urec.get("bytes_as_read") else None
                    ur["traffic_source"] = "profiles/pmc_traffic.json (static: FETCH_SIZE + WRITE_SIZE of k_fwd x2, k_mid, k_grad x2 per update, as read / with the guide's 2x FETCH correction, " + str(urec.get("round")) + ")"
            roof["update_roofline"] = ur
        if world == 1 and not args.no_cpu_baseline and args.hidden.lower() == "250x500":      # (the CPU port is timed at the headline architecture)
            t_cpu0 = time.perf_counter()
            cpu = cpu_baseline(args.envs, "train" if mode == "group" else mode, args.learners if mode == "group" else args.updates)
            t_cpu = time.perf_counter() - t_cpu0
    if dist is not None:
        dist.barrier()
    if rank == 0:
        total_env_steps = args.envs * world * args.steps
        out = {
            "metric": "env-steps/sec",
            "value": total_env_steps / dt,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm_steps": prewarm_steps,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": wl.dtype,
            "data": "synthetic",
            "config": {"workload": f"{args.envs} parallel shems_LU1 envs per GPU, Charger98 synthetic train table "
                                   f"(4320 rows), {EP_LEN}-step episodes, mode={mode}" + ("" if args.hidden.lower() == "250x500" else f", networks {args.hidden}"),
                       "envs_per_gpu": args.envs, "episode_len": EP_LEN, "mode": mode, "mixed_profiles": bool(args.mixed)},
            "roofline": roof,
            "cpu_baseline": cpu,
            # where the run's wall time went, so that a sampled GPU-utilisation figure can be reconciled with the line: the GPU is busy
            # during gpu_section_s (pre-warm + warm-up + timed steps) and roofline_pass_s only; cpu_baseline_s is host cores alone
            "gpu_section_s": t_gpu_section_end - t_gpu_section_start,
            "roofline_pass_s": t_pass,
            "cpu_baseline_s": t_cpu,
            "also_s": t_also,
        }
        if also is not None:
            out["also"] = also
        out.update(wl.extra())
        if census is not None:
            out["rccl_ranks"] = census["rccl_ranks"]
            out["collective_backend"] = census["backend"]
            out["distinct_devices"] = census["distinct_devices"]
            out["rank_census"] = census["ranks"]
        if mode == "train":
            out["updates_per_sec"] = args.updates * args.steps / dt      # complete replay() equivalents (B = 120)
        if mode == "group":
            out["updates_per_sec"] = args.learners * world * args.steps / dt   # one replay() per learner per vector step
        if json_fd is None:
            print(json.dumps(out), flush=True)
        else:
            os.write(json_fd, (json.dumps(out) + "\n").encode())
    if hasattr(wl, "close"):
        if dist is not None:
            dist.barrier()                        # nobody tears its end of a communicator down while a peer still exchanges
        wl.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
