"""Learner groups: many INDEPENDENT DDPG learners advanced by the same launches.

The thesis protocol is 40 seeds x 10 charger profiles, each an OS process of its own running the batch-1 loop of
DDPG_reinforce_charger_v1.jl:10-47 (run scripts `RL-SHEMS/*.sh`).  Here learner l owns one slab of device memory
(networks, targets, ADAM moments, gradients, workspace, normalisation, replay ring -- all carved identically) and the
env block [l * E, (l + 1) * E) of one ShemsBatch; `shems_act_step_group_dev` / `shems_ddpg_group_*` take learner 0's
pointers plus the slab stride and run every learner in the same grid (grid z = learner for the update kernels).
Learners never exchange anything, so a group shards over GPUs as plain replicas (no collective).

Per learner the results are bit-identical to the single-learner `Agent` driven on that learner's views
(tests/test_group_gpu.py); `LearnerGroup.learners[l]` / `.rings[l]` ARE such single-learner objects on the slab.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from .ddpg import (ACTION, BATCH_SIZE, MEM_SIZE, N_ACTOR, N_CRITIC, NOISE_SIGMA, STATE, Agent, RingWindow, _declare, act_kernel_name)
from .replay import ReplayRing


class Group(C.Structure):              # shems_group
    _fields_ = [("count", C.c_int32), ("reserved", C.c_int32), ("stride_bytes", C.c_int64), ("envs_per_learner", C.c_int64)]


class GroupW2T(C.Structure):           # shems_group_w2t
    _fields_ = [("actor", C.c_void_p), ("critic", C.c_void_p)]


W2T_FLOATS = 32 * 4 * 64 * 64          # SHEMS_W2T_FLOATS: [4 k-tiles][8 n-tiles][m | v | p | target][64][64] per network


def _declare_group():
    L = _declare()
    if getattr(L, "_group_declared", False):
        return L
    vp, i64 = C.c_void_p, C.c_int64
    from .ddpg import ActParams, DdpgArgs
    PD, PG, PR, PT = C.POINTER(DdpgArgs), C.POINTER(Group), C.POINTER(_capi.Replay), C.POINTER(GroupW2T)
    L.shems_act_step_group_dev.argtypes = [C.POINTER(_capi.View), C.POINTER(ActParams), PG, vp, vp, PR, C.POINTER(RingWindow), vp]
    L.shems_act_step_group_tiled_dev.argtypes = [C.POINTER(_capi.View), C.POINTER(ActParams), PG, PT, vp, vp, PR, C.POINTER(RingWindow), vp]
    dbl = C.c_double
    L.shems_ddpg_group_update_tiled.argtypes = [PD, PR, PG, PT, i64, C.c_uint64, C.c_uint32, dbl, dbl, dbl, dbl, dbl, dbl, C.c_int32, vp]
    L.shems_group_w2_to_tiled.argtypes = [PD, PG, PT, vp]
    L.shems_group_w2_to_flux.argtypes = [PD, PG, PT, vp]
    L.shems_act_step_group_kernel.argtypes = [i64, i64, C.c_int, C.c_char_p, C.c_int32]
    for fn in ("shems_act_step_group_tiled_dev", "shems_ddpg_group_update_tiled", "shems_group_w2_to_tiled", "shems_group_w2_to_flux", "shems_act_step_group_kernel"):
        getattr(L, fn).restype = C.c_int
    L.shems_ddpg_group_update.argtypes = [PD, PR, PG, i64, C.c_uint64, C.c_uint32, dbl, dbl, dbl, dbl, dbl, dbl, vp]
    L.shems_ddpg_group_update.restype = C.c_int
    L.shems_ddpg_group_update_tp.argtypes = [PD, PR, PG, i64, C.c_uint64, C.c_uint32, dbl, dbl, dbl, dbl, dbl, dbl, C.c_int32, vp]
    L.shems_ddpg_group_update_tp.restype = C.c_int
    L.shems_ddpg_group_critic_grad.argtypes = [PD, PR, PG, i64, C.c_uint64, C.c_uint32, vp]
    L.shems_ddpg_group_critic_apply.argtypes = [PD, PG, C.c_double, C.c_double, C.c_double, vp]
    L.shems_ddpg_group_actor_grad.argtypes = [PD, PG, vp]
    L.shems_ddpg_group_actor_apply.argtypes = [PD, PG, C.c_double, C.c_double, C.c_double, vp]
    L.shems_minmax_group_dev.argtypes = [PR, PG, i64, i64, C.c_uint64, vp, vp, vp]
    for fn in ("shems_act_step_group_dev", "shems_ddpg_group_critic_grad", "shems_ddpg_group_critic_apply",
               "shems_ddpg_group_actor_grad", "shems_ddpg_group_actor_apply", "shems_minmax_group_dev"):
        getattr(L, fn).restype = C.c_int
    L._group_declared = True
    return L


def group_act_kernel_name(n_envs, envs_per_learner, tiled):
    """The kernel the fused-step dispatcher runs for a learner group, by its profiler name (shems_act_step_group_kernel)."""
    L = _declare_group()
    buf = C.create_string_buffer(96)
    _capi.check(L.shems_act_step_group_kernel(int(n_envs), int(envs_per_learner), 1 if tiled else 0, buf, 96))
    return buf.value.decode()


def _pad4(n):
    return (int(n) + 3) & ~3


class LearnerGroup:
    """`count` independent learners, learner l seeded with (seed + l) for its network initialisation and (rng_seed + l)
    for its minibatch stream; exploration noise is keyed per env, so it differs between learners by construction."""

    # replay() of a group: "throughput" = csrc/shems_gupd.hip (eight launches shaped for hundreds of learners: small tiles, four workgroups
    # resident per CU, plain back-propagation), "latency" = the single-learner kernels with grid z = learner (five launches, per learner
    # bit-identical to Agent.replay).  Default: throughput from TP_MIN_LEARNERS learners up.
    TP_MIN_LEARNERS = 16

    def __init__(self, count, envs_per_learner, seed=1231, rng_seed=None, capacity=MEM_SIZE, sigma=NOISE_SIGMA, device=None, form=None, tiled=None):
        """tiled (throughput form only; default on, SHEMS_GROUP_TILED=0 switches it off): the layer-2 state of both networks (W2, its
        ADAM moments, the target's W2) is kept in the TILED working layout (shems_group_w2t, include/shems_hip.h) while the group trains:
        one contiguous 64 KB piece per 64 x 64 tile for the update's W2-gradient / ADAM launches (4.78 against 3.85 TB/s), read from
        there by the forward / D1 launches and by the fused act/step launch.  The Flux-order blocks -- what `learners[l].actor` etc. ARE --
        keep everything else and are the API's view of W2: `flux_()` brings their W2 ranges up to date (call it before reading a learner's
        tensors, evaluating a learner through its Agent, or saving), `Agent.set_params` on a learner of the group is noticed by itself,
        any other write into those tensors must be followed by `flux_changed()`."""
        import os
        import torch
        self.torch = torch
        self.L = _declare_group()
        self.count, self.envs_per_learner, self.capacity = int(count), int(envs_per_learner), int(capacity)
        self.form = form if form is not None else ("throughput" if self.count >= self.TP_MIN_LEARNERS else "latency")
        if self.form not in ("throughput", "latency"):
            raise ValueError("form must be 'throughput' or 'latency'")
        self.store_grad = False                    # throughput form: also leave the gradients in grad_actor / grad_critic (tests)
        self.tiled = (self.form == "throughput" and os.environ.get("SHEMS_GROUP_TILED", "1") != "0") if tiled is None else bool(tiled)
        if self.tiled and self.form != "throughput":
            raise ValueError("the tiled working layout belongs to the throughput form")
        self._flux_valid, self._tiled_valid = True, False          # which copy of the layer-2 state is current (both may be)
        if self.count < 1 or self.envs_per_learner < 32 or self.envs_per_learner % 32 != 0:
            raise ValueError("a learner group needs count >= 1 and envs_per_learner a multiple of 32")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.seed = int(seed)
        self.rng_seed = self.seed if rng_seed is None else int(rng_seed)
        nws = C.c_int64(0)
        _capi.check(self.L.shems_ddpg_workspace_floats(C.byref(nws)))
        # one slab per learner; every block starts on a 16-byte boundary (float offsets are multiples of 4)
        layout, off = {}, 0
        for name, n in (("actor", N_ACTOR), ("critic", N_CRITIC), ("actor_t", N_ACTOR), ("critic_t", N_CRITIC),
                        ("m_actor", N_ACTOR), ("v_actor", N_ACTOR), ("m_critic", N_CRITIC), ("v_critic", N_CRITIC),
                        ("w2t_actor", W2T_FLOATS if self.tiled else 0), ("w2t_critic", W2T_FLOATS if self.tiled else 0),
                        ("grad_actor", N_ACTOR), ("grad_critic", N_CRITIC), ("s_min", STATE), ("s_max", STATE), ("losses", 2),
                        ("ws", nws.value), ("ring_s", self.capacity * STATE), ("ring_a", self.capacity * ACTION),
                        ("ring_r", self.capacity), ("ring_s2", self.capacity * STATE), ("ring_done", (self.capacity + 3) // 4)):
            layout[name] = (off, int(n))
            off = _pad4(off + n)
        self.layout, self.slab_floats = layout, off
        self.slab = torch.zeros((self.count, self.slab_floats), dtype=torch.float32, device=self.device)
        self.learners, self.rings = [], []
        for l in range(self.count):
            v = lambda name: self.slab[l, layout[name][0]:layout[name][0] + layout[name][1]]
            tens = {k: v(k) for k in ("actor", "critic", "actor_t", "critic_t", "m_actor", "v_actor", "m_critic", "v_critic",
                                      "grad_actor", "grad_critic", "s_min", "s_max", "ws", "losses")}
            self.learners.append(Agent(seed=self.seed + l, rng_seed=self.rng_seed + l, sigma=sigma, device=self.device, tensors=tens))
            if self.tiled:
                self.learners[-1]._before_param_write = self._before_flux_write
            done = v("ring_done").view(torch.uint8)[:self.capacity]
            self.rings.append(ReplayRing(self.capacity, tensors=(v("ring_s").view(self.capacity, STATE), v("ring_a").view(self.capacity, ACTION),
                                                                 v("ring_r"), v("ring_s2").view(self.capacity, STATE), done)))
        self.updates = 0
        self.tick = 0

    # ------------------------------------------------------------------
    def struct(self):
        return Group(self.count, 0, self.slab_floats * 4, self.envs_per_learner)

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    # ---- the tiled working layout of the layer-2 state -----------------------------------------------------------------------------
    def w2t_struct(self):
        base = self.slab.data_ptr()
        return GroupW2T(base + 4 * self.layout["w2t_actor"][0], base + 4 * self.layout["w2t_critic"][0])

    def _use_tiled(self):
        """Before a launch that reads / writes the tiled regions: bring them up to date from the Flux-order blocks if those were written."""
        if self.tiled and self.form != "throughput":
            self.flux_()                               # the form was switched on a tiled group: the latency form speaks Flux order
            self._tiled_valid = False
            return False
        if not self.tiled:
            if not self._flux_valid:                   # (the flag was switched off on a group whose current layer-2 state is in the tiles)
                raise RuntimeError("this group's Flux-order blocks are stale: call flux_() while it is still tiled")
            return False
        if not self._tiled_valid:
            d, g, t = self.learners[0]._ddpg_args(), self.struct(), self.w2t_struct()
            _capi.check(self.L.shems_group_w2_to_tiled(C.byref(d), C.byref(g), C.byref(t), self._stream()))
            self._tiled_valid = True
        return True

    def flux_(self):
        """Bring the W2 ranges of the Flux-order blocks (learners[l].actor / critic / actor_t / critic_t / m_* / v_*) up to date."""
        if self.tiled and not self._flux_valid:
            d, g, t = self.learners[0]._ddpg_args(), self.struct(), self.w2t_struct()
            _capi.check(self.L.shems_group_w2_to_flux(C.byref(d), C.byref(g), C.byref(t), self._stream()))
            self._flux_valid = True
        return self

    def flux_changed(self):
        """The caller wrote into Flux-order tensors of the group (after flux_()): the tiled regions are re-made before their next use."""
        if self.tiled and not self._flux_valid:
            raise RuntimeError("flux_changed(): the Flux-order blocks were stale when they were written -- call flux_() before writing into them")
        self._tiled_valid = False
        return self

    def _before_flux_write(self):
        self.flux_()
        self._tiled_valid = False

    @property
    def n_envs(self):
        return self.count * self.envs_per_learner

    def populate_memory(self, env, seed=None):
        """populate_memory (MPS:9-29) per learner on its own env block (learner l: seed + l)."""
        seed = self.rng_seed if seed is None else int(seed)
        E = self.envs_per_learner
        for l, (ag, ring) in enumerate(zip(self.learners, self.rings)):
            ag.populate_memory(env.slice(l * E, E), ring, seed=seed + l)
        return self

    def min_max_buffer(self, count=None, seed=None):
        """min_max_buffer (MPS:50-53) for every learner in one launch (learner l: Philox key seed + l)."""
        g, r0 = self.struct(), self.rings[0].struct()
        n = len(self.rings[0])
        a0 = self.learners[0]
        _capi.check(self.L.shems_minmax_group_dev(C.byref(r0), C.byref(g), n, n if count is None else int(count),
                                                  self.rng_seed if seed is None else int(seed), C.c_void_p(a0.s_min.data_ptr()),
                                                  C.c_void_p(a0.s_max.data_ptr()), self._stream()))
        return self

    def act_step(self, env, train=True, tick=None, a_out=None, returns_acc=None, window=None):
        """One fused vector step for all learners: env i acts with learner i // E's actor; with `window` = (pos, count, offset)
        each learner stores `count` transitions of its own env block into its own ring."""
        if env.n != self.n_envs:
            raise ValueError("the env batch must hold count * envs_per_learner envs")
        env.use_torch_stream()
        v, g = env.view(), self.struct()
        a0 = self.learners[0]
        p = a0._act_params(train, self.tick if tick is None else tick)
        ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        r0 = self.rings[0].struct()
        w = RingWindow(*window) if window is not None else None
        if self._use_tiled():
            t = self.w2t_struct()
            _capi.check(self.L.shems_act_step_group_tiled_dev(C.byref(v), C.byref(p), C.byref(g), C.byref(t), ptr(a_out), ptr(returns_acc),
                                                              C.byref(r0) if w is not None else None, C.byref(w) if w is not None else None,
                                                              self._stream()))
        else:
            _capi.check(self.L.shems_act_step_group_dev(C.byref(v), C.byref(p), C.byref(g), ptr(a_out), ptr(returns_acc),
                                                        C.byref(r0) if w is not None else None, C.byref(w) if w is not None else None,
                                                        self._stream()))
        if w is not None:
            for ring in self.rings:
                ring.pushed += int(window[1])

    def replay(self, tick=None):
        """replay() (DDPG.jl:121-145) for every learner.  form "throughput": eight launches of csrc/shems_gupd.hip; "latency": 5 launches
        in total, grid z = learner (fused = False: the split calls, 7 launches -- the same bits)."""
        a0, g, r0 = self.learners[0], self.struct(), self.rings[0].struct()
        d = a0._ddpg_args()
        st = self._stream()
        tick = self.updates if tick is None else tick
        if self._use_tiled():
            t = self.w2t_struct()
            _capi.check(self.L.shems_ddpg_group_update_tiled(C.byref(d), C.byref(r0), C.byref(g), C.byref(t), len(self.rings[0]), self.rng_seed, int(tick) & 0xFFFFFFFF,
                                                             a0.eta_crit, a0.bp_critic[0], a0.bp_critic[1], a0.eta_act, a0.bp_actor[0], a0.bp_actor[1],
                                                             1 if self.store_grad else 0, st))
            self._flux_valid = False
        elif self.form == "throughput":
            _capi.check(self.L.shems_ddpg_group_update_tp(C.byref(d), C.byref(r0), C.byref(g), len(self.rings[0]), self.rng_seed, int(tick) & 0xFFFFFFFF,
                                                          a0.eta_crit, a0.bp_critic[0], a0.bp_critic[1], a0.eta_act, a0.bp_actor[0], a0.bp_actor[1],
                                                          1 if self.store_grad else 0, st))
        elif getattr(self, "fused", True):
            _capi.check(self.L.shems_ddpg_group_update(C.byref(d), C.byref(r0), C.byref(g), len(self.rings[0]), self.rng_seed, int(tick) & 0xFFFFFFFF,
                                                       a0.eta_crit, a0.bp_critic[0], a0.bp_critic[1], a0.eta_act, a0.bp_actor[0], a0.bp_actor[1], st))
        else:
            _capi.check(self.L.shems_ddpg_group_critic_grad(C.byref(d), C.byref(r0), C.byref(g), len(self.rings[0]), self.rng_seed,
                                                            int(tick) & 0xFFFFFFFF, st))
            _capi.check(self.L.shems_ddpg_group_critic_apply(C.byref(d), C.byref(g), a0.eta_crit, a0.bp_critic[0], a0.bp_critic[1], st))
            _capi.check(self.L.shems_ddpg_group_actor_grad(C.byref(d), C.byref(g), st))
            _capi.check(self.L.shems_ddpg_group_actor_apply(C.byref(d), C.byref(g), a0.eta_act, a0.bp_actor[0], a0.bp_actor[1], st))
        for ag in self.learners:                   # the learners advance in lockstep: shared beta powers / update count
            ag.bp_critic = [ag.bp_critic[0] * 0.9, ag.bp_critic[1] * 0.999]
            ag.bp_actor = [ag.bp_actor[0] * 0.9, ag.bp_actor[1] * 0.999]
            ag.updates += 1
        self.updates += 1

    def ring_window(self, num_steps=72, window_count=None):
        """(count, offset) of the transitions each learner stores at the CURRENT tick.
        window_count None: the vectorised default (SURVEY 8(d) replay-capacity note): a rotating window of min(E, capacity / num_steps)
        of the learner's E households per step, so the ring spans about one episode of each.
        window_count 1: the reference's ratio -- episode! remembers ONE transition per replay() (DDPG.jl:229-233, push order
        [s, a, r, s', done] MPS:46-47): always household 0 of the learner's block, so that consecutive ring entries are one household's
        trajectory (s' of entry t = s of entry t + 1 inside an episode), as the reference's single env fills its CircularBuffer.
        Any other count: a rotating window of that many households."""
        E = self.envs_per_learner
        wc = min(E, max(1, self.capacity // int(num_steps))) if window_count is None else int(window_count)
        if not 1 <= wc <= E:
            raise ValueError("window_count must be in 1 .. envs_per_learner")
        return wc, (0 if wc == 1 else (self.tick * wc) % E)

    def episode_(self, env, train=True, num_steps=None, rng_ep=0, episode=0, window_count=None):
        """episode! for all learners at once (cf. Agent.episode_).  Returns the per-env episode returns [count * E].
        window_count: see ring_window (1 = one remembered transition per update, the thesis protocol's update-to-data ratio)."""
        t = self.torch
        num_steps = env.maxsteps if num_steps is None else int(num_steps)
        env.reset_(rng_ep, episode=episode) if rng_ep != -1 else env.reset_(-1)
        returns = t.zeros(env.n, dtype=t.float64, device=self.device)
        for step in range(num_steps):
            tick = (int(episode) * 4096 + step) & 0xFFFFFFFF
            win = (self.rings[0].pos, *self.ring_window(num_steps, window_count)) if train else None
            self.act_step(env, train=train, tick=tick, returns_acc=returns, window=win)
            if train:
                self.replay()
            self.tick += 1
        return returns


class GroupWorkload:
    """bench.py's "group" step: one fused vector step of all learners' envs + one replay() of every learner."""

    name = "group"
    dtype = "f32"
    EP_LEN = 72

    def __init__(self, S, torch, n, learners, seed, mixed=False, form=None, window=None):
        """window: transitions each learner remembers per vector step (None: min(E, MEM_SIZE / 72), the rotating window; 1: the reference's
        one transition per replay(), LearnerGroup.ring_window)."""
        self.S, self.torch, self.n, self.count = S, torch, int(n), int(learners)
        if self.n % self.count or (self.n // self.count) % 32:
            raise ValueError("--envs must be learners x a multiple of 32")
        E = self.n // self.count
        if mixed:                       # the thesis grid: learner l trains on charger profile l mod 10 (ids 1-9, 98; LU1:47-58)
            ids = (1, 2, 3, 4, 5, 6, 7, 8, 9, 98)
            tabs = [S.tables.synthetic_table("train", c) for c in ids]
            row0 = np.cumsum([0] + [t.shape[0] for t in tabs])
            cfgs = [S.make_config(c, row0[k], tabs[k].shape[0]) for k, c in enumerate(ids)]
            co = ((np.arange(self.n) // E) % len(ids)).astype(np.uint16)
            self.env = S.ShemsBatch(self.n, self.EP_LEN, tabs, cfgs, co, device=torch.cuda.current_device()).use_torch_stream()
        else:
            self.tab = S.tables.synthetic_table("train", 98)
            self.env = S.ShemsBatch(self.n, self.EP_LEN, [self.tab], [S.make_config(98, 0, self.tab.shape[0])],
                                    device=torch.cuda.current_device()).use_torch_stream()
        self.env_seed = int(seed)
        self.group = LearnerGroup(self.count, E, seed=1231, rng_seed=self.env_seed, form=form)
        self.group.populate_memory(self.env, seed=self.env_seed)
        self.group.min_max_buffer()
        self.window = None if window is None else int(window)
        self.win = self.group.ring_window(self.EP_LEN, self.window)[0]
        self.t, self.episode = 0, 1
        self.env.reset_(self.env_seed, episode=self.episode)

    def step(self):
        if self.t and self.t % self.EP_LEN == 0:
            self.episode += 1
            v = self.env.view()
            _capi.check(_capi.lib().shems_reset_seeded_dev(C.byref(v), self.env_seed, self.episode, self.env._stream()))
        g = self.group
        g.tick = self.t
        g.act_step(self.env, train=True, tick=self.t, window=(g.rings[0].pos, *g.ring_window(self.EP_LEN, self.window)))
        g.replay()
        self.t += 1

    def finish(self):
        self.torch.cuda.synchronize()
        self.env.check_error()
        end = self.group.layout["ws"][0]               # networks, targets, moments, gradients, normalisation, losses (the workspace
        if not bool(self.torch.isfinite(self.group.slab[:, :end]).all()):      # keeps int32 slots whose -1 pad reads as NaN)
            raise RuntimeError("non-finite learner state after the timed steps")

    def kernel_pass(self, reps):
        """HIP-event timing of the fused act/step launch (all learners' envs) and of one grouped replay()."""
        torch = self.torch
        from .timing import time_launches
        g = self.group
        reset = lambda k, i: self.env.reset_(self.env_seed, episode=100000 + i) if k % 8 == 0 else None
        def act(i):
            g.tick = i
            g.act_step(self.env, train=True, tick=i, window=(g.rings[0].pos, *g.ring_window(self.EP_LEN, self.window)))
        avg_us, med_us, reps = time_launches(torch, act, min(reps, 96), before_group=reset)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nup = 20
        e0.record()
        for _ in range(nup):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        self.update_us = e0.elapsed_time(e1) * 1e3 / nup
        flops = 2 * (9 * 250 + 250 * 500 + 500 * 2) * self.n
        act = dict(kernel=group_act_kernel_name(self.n, g.envs_per_learner, g.tiled), avg_us=avg_us, median_us=med_us, launches=reps,
                   bound="mfma", algorithmic=flops, unit="TFLOP/s", peak=157.3)
        if self.update_us <= avg_us:
            return act
        # the grouped replay() is the larger share of the step: it is the kernel the roofline object describes.
        # Work per learner-update: 2.565 MFLOP x B = 307.8 MFLOP, and 12.4 MB that have to move (258 003 parameters x
        # (28 B ADAM + 12 B soft update) + one pass over the four networks' weights) -- at 400 learners 0.78 ms at the fp32-MFMA peak and
        # 0.62 ms at 8 TB/s: the matrix pipe is the binding roof, HBM within 25 % of it.
        upd_bytes = 258003 * 40 + 2 * 4 * (129002 + 129001)
        tp = g.form == "throughput"
        name = ("grouped replay(), throughput form: k_tp_prep + k_tp_fwd x3 + k_tp_d1 x2 + k_tp_gw2 x2 (csrc/shems_gupd.hip)" if tp else
                "grouped replay(), latency form: k_fwd x2 + k_mid + k_grad x2 with grid z = learner (csrc/shems_ddpg.hip)")
        t_s = self.update_us * 1e-6
        return dict(kernel=name, avg_us=self.update_us, median_us=self.update_us, launches=nup, bound="mfma",
                    algorithmic=2.565e6 * BATCH_SIZE * self.count, unit="TFLOP/s", peak=157.3,
                    algorithmic_bytes=upd_bytes * self.count, algorithmic_gbs=upd_bytes * self.count / t_s / 1e9,
                    hbm_frac_of_8tbs=upd_bytes * self.count / t_s / 8e12, per_learner_update_us=self.update_us / self.count,
                    launches_per_update=8 if tp else 5,
                    other_kernel={"kernel": act["kernel"], "avg_us": avg_us, "frac": flops / (avg_us * 1e-6) / 1e12 / 157.3},
                    method=f"HIP events around {nup} grouped replay() calls back to back ({8 if tp else 5} launches each, all learners per launch)")

    def extra(self):
        return {"learners": self.count, "envs_per_learner": self.n // self.count, "updates_per_step": self.count,
                "batch_size": BATCH_SIZE, "mem_size": MEM_SIZE, "replay_window_envs_per_step": self.win,
                "group_update_us": getattr(self, "update_us", None), "update_mflop": 307.8, "update_form": self.group.form,
                "w2_layout": "tiled (shems_group_w2t: one 64 KB piece per 64 x 64 tile of W2 | m | v | target)" if self.group.tiled else "Flux order"}
