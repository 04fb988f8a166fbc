"""Host-side mirror of the reference's environment interface, over the C ABI.

Reference interface (Julia, Reinforce.jl generics extended in shems_LU1.jl):

    Shems(maxsteps, path)                 LU1:203      ->  Shems(maxsteps, path)          (N = 1)
    reset!(env; rng=0)                    LU1:206      ->  reset_(env, rng=0)
    step!(env, s, a; track=0)             LU1:343      ->  step_(env, s, a, track=0)
    action(env, a::ShemsAction)           LU1:283      ->  action(env, a)
    action(env, track)                    LU1:318      ->  action(env, track)   (a negative number)
    finished(env, s')                     LU1:487      ->  finished(env, s2)
    env.state / .reward / .a / .step / .idx / .maxsteps / .path    LU1:169-177

Python has no `!` in identifiers, hence the trailing underscore.  `ShemsBatch` is the same
interface for N parallel instances (arrays gain a leading N axis: obs [N][9] is Julia's 9xN).
Every method runs on the GPU through libshems_hip.so; there is no CPU implementation here.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _capi
from . import tables as _tables
from ._capi import BoundsError, Config, ShemsError  # noqa: F401

f32 = np.float32


def make_config(charger_id=98, table_row0=0, nrow=0, disc_weight=0.01, disc_pot=2.0, penalty_weight=0.1):
    """Module globals of shems_LU1.jl:40-59, 92-99 for one charger id (+ per-env reward weights).
    soc_max is the Float32 product written in the capacities Dict (e.g. 7.5f0 * 0.9f0); the Market
    weights are Float64(Float32 literal) exactly as `Market(0.2f0, w, pot)` stores them."""
    cap, nominal, rate = _tables.CHARGER_PROFILES[int(charger_id)]
    soc_max = f32(nominal) * f32(0.9)
    return Config(float(f32(cap)), float(soc_max), float(rate), float(f32(disc_weight)), float(f32(disc_pot)),
                  float(f32(penalty_weight)), int(table_row0), int(nrow), 0)


class EnvSlice:
    """Envs [start, start + count) of a ShemsBatch through the device-chaining entry points only (view, seeded reset,
    rollout): what one learner of a learner group needs for populate_memory and what the parity tests drive.  Inside the
    slice env indices -- and with them the Philox counters of reset / random actions / noise -- start at 0."""

    def __init__(self, parent, start, count):
        start, count = int(start), int(count)
        if start < 0 or count < 1 or start + count > parent.n:
            raise ValueError("slice outside the batch")
        self.parent, self.start, self.n, self.maxsteps = parent, start, count, parent.maxsteps
        self.device_index = parent.device_index
        self._L = parent._L

    def view(self):
        v = self.parent.view()
        v.n_envs = self.n
        v.obs += self.start * _capi.NSTATE * 4
        v.idx += self.start * 4
        v.step += self.start * 4
        if v.cfg_of_env:
            v.cfg_of_env += self.start * 2
        return v

    def use_torch_stream(self):
        self.parent.use_torch_stream()
        return self

    def _stream(self):
        return self.parent._stream()

    def reset_(self, rng=0, episode=0):
        v = self.view()
        _capi.check(self._L.shems_reset_seeded_dev(C.byref(v), int(rng) & ((1 << 64) - 1), int(episode), self._stream()))
        return self



def mixed_profile_setup(n_envs, charger_ids=(1, 2, 3, 4, 5, 6, 7, 8, 9, 98), sweep=((0.01, 2.0), (0.04, 2.0), (0.1, 2.0), (0.01, 1.0), (0.04, 1.0), (0.1, 1.0)),
                        split="train", prefer_real=True):
    """BASELINE config 5: one table per charger profile -- the real exogenous series where the reference holds one (train:
    Chargers 01/03/04/05/08/09, tables.real_series), the seeded synthetic generator for the rest -- and one config per
    (profile, discomfort weight, power) point of the sweep (values of shems_LU1.jl:20-41 / shems_LU1_input0607.jl); env i uses
    config i mod n_cfg.  Returns (tables, configs, cfg_of_env)."""
    tabs = [_tables.profile_table(c, split, prefer_real) for c in charger_ids]
    row0 = np.cumsum([0] + [t.shape[0] for t in tabs])
    cfgs = [make_config(c, row0[p], tabs[p].shape[0], w, pot) for p, c in enumerate(charger_ids) for (w, pot) in sweep]
    return tabs, cfgs, (np.arange(n_envs) % len(cfgs)).astype(np.uint16)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class ShemsBatch:
    """N parallel `Shems` environments resident on one MI355X."""

    def __init__(self, n_envs, maxsteps, tables, configs=None, cfg_of_env=None, device=0):
        L = _capi.lib()
        self._L = L
        self._h = C.c_void_p()
        self.n = int(n_envs)
        self.maxsteps = int(maxsteps)
        self.device_index = int(device)
        _capi.check(L.shems_create(self.n, self.maxsteps, int(device), C.byref(self._h)))
        tabs = tables if isinstance(tables, (list, tuple)) else [tables]
        tabs = [np.ascontiguousarray(t, dtype=np.float32) for t in tabs]
        for t in tabs:
            if t.ndim != 2 or t.shape[1] != _capi.NCOL:
                raise ValueError("a table must be [nrow][8] float32")
        self.table_row0 = np.cumsum([0] + [t.shape[0] for t in tabs])[:-1].astype(np.int64)
        self.table_nrow = np.array([t.shape[0] for t in tabs], np.int64)
        rows = np.ascontiguousarray(np.concatenate(tabs, 0))
        _capi.check(L.shems_set_tables(self._h, _ptr(rows), rows.shape[0]))
        if configs is None:
            configs = [make_config(98, 0, tabs[0].shape[0])]
        self.configs = list(configs)
        arr = (Config * len(self.configs))(*self.configs)
        co = None
        if cfg_of_env is not None:
            co = np.ascontiguousarray(cfg_of_env, dtype=np.uint16)
            if co.shape != (self.n,):
                raise ValueError("cfg_of_env must have shape (n_envs,)")
        self.cfg_of_env = co
        _capi.check(L.shems_set_configs(self._h, arr, len(self.configs), _ptr(co)))
        self.reward = np.zeros(self.n, np.float64)
        self.a = np.tile(np.array([0.7, 1.0], np.float32), (self.n, 1))     # ShemsAction() LU1:151

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.shems_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reference API ----------------------------------------------------
    def reset_(self, rng=0, idx0=None, soc_b0=None, episode=0):
        """reset!(env; rng).  rng == -1: deterministic start (idx = 1, Soc_b = mid).  Otherwise the two
        MersenneTwister draws of LU1:224-225 are either given (idx0, soc_b0) or drawn on the device
        from Philox keyed by `rng` (and `episode`)."""
        if rng == -1:
            _capi.check(self._L.shems_reset(self._h, 1, None, None))
        elif idx0 is not None:
            i0 = np.ascontiguousarray(idx0, dtype=np.int32)
            s0 = np.ascontiguousarray(soc_b0, dtype=np.float32)
            if i0.shape != (self.n,) or s0.shape != (self.n,):
                raise ValueError("idx0 / soc_b0 must have shape (n_envs,)")
            _capi.check(self._L.shems_reset(self._h, 0, _ptr(i0), _ptr(s0)))
        else:
            _capi.check(self._L.shems_reset_seeded(self._h, int(rng) & ((1 << 64) - 1), int(episode)))
        self.reward[:] = 0.0
        self.a[:] = np.array([0.7, 1.0], np.float32)
        return self

    def step_(self, s, a, track=0):
        """step!(env, s, a; track).  `s` is ignored exactly as in the reference (LU1:344 reads env.state).
        Returns (r [N] f64, s' [N][9] f32) and, for track != 0, the [N][23] Float64 results rows."""
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(self.n, 2)
        mode = _capi.TRACK_OFF if track == 0 else (_capi.TRACK_DRL if track > 0 else _capi.TRACK_RULE)
        r = np.empty(self.n, np.float64)
        s2 = np.empty((self.n, _capi.NSTATE), np.float32)
        res = np.empty((self.n, _capi.NRESULT), np.float64) if track != 0 else None
        _capi.check(self._L.shems_step(self._h, _ptr(a), mode, _ptr(r), _ptr(s2), _ptr(res)))
        self.reward = r
        self.a = a.copy() if track >= 0 else np.zeros((self.n, 2), np.float32)   # LU1:349, 351-353
        if track == 0:
            return r, s2
        return r, s2, res

    def action(self, a_or_track=-1):
        """action(env, a::ShemsAction) for an [N][2] array of targets, action(env, track) for a number."""
        out = np.empty((self.n, 2), np.float32)
        if np.isscalar(a_or_track):
            _capi.check(self._L.shems_rule_action(self._h, _ptr(out)))
        else:
            t = np.ascontiguousarray(a_or_track, dtype=np.float32).reshape(self.n, 2)
            _capi.check(self._L.shems_action(self._h, _ptr(t), _ptr(out)))
        return out

    def finished(self, s2=None):
        done = np.empty(self.n, np.uint8)
        _capi.check(self._L.shems_finished(self._h, _ptr(done)))
        return done.astype(bool)

    # -- fields -----------------------------------------------------------
    @property
    def state(self):
        s = np.empty((self.n, _capi.NSTATE), np.float32)
        _capi.check(self._L.shems_get_state(self._h, _ptr(s), None, None))
        return s

    @state.setter
    def state(self, value):
        s = np.ascontiguousarray(value, dtype=np.float32).reshape(self.n, _capi.NSTATE)
        _capi.check(self._L.shems_set_state(self._h, _ptr(s), None, None))

    @property
    def idx(self):
        i = np.empty(self.n, np.int32)
        _capi.check(self._L.shems_get_state(self._h, None, _ptr(i), None))
        return i

    @idx.setter
    def idx(self, value):
        i = np.ascontiguousarray(value, dtype=np.int32).reshape(self.n)
        _capi.check(self._L.shems_set_state(self._h, None, _ptr(i), None))

    @property
    def step(self):
        i = np.empty(self.n, np.int32)
        _capi.check(self._L.shems_get_state(self._h, None, None, _ptr(i)))
        return i

    @step.setter
    def step(self, value):
        i = np.ascontiguousarray(value, dtype=np.int32).reshape(self.n)
        _capi.check(self._L.shems_set_state(self._h, None, None, _ptr(i)))

    # -- device chaining --------------------------------------------------
    def view(self):
        """Device pointers of this handle (shems_view) for the zero-copy entry points."""
        v = _capi.View()
        _capi.check(self._L.shems_get_view(self._h, C.byref(v)))
        return v

    def use_torch_stream(self):
        """Run the handle's kernels on PyTorch's current HIP stream so they order with tensor ops and with the policy /
        DDPG kernels.  Idempotent; every device-chaining method calls it, so mixing the host-array API and the device API
        on one handle is safe."""
        import torch
        ptr = torch.cuda.current_stream().cuda_stream
        if getattr(self, "_bound_stream", None) != ptr:
            self.torch_device = torch.device("cuda", torch.cuda.current_device())
            self.set_stream(ptr)                      # synchronises the stream used so far, then switches
            self._bound_stream = ptr
        return self

    def _stream(self):
        self.use_torch_stream()
        return C.c_void_p(self._bound_stream)

    def rollout(self, policy, nsteps, seed=0, ring=None, ring_envs=0):
        """`nsteps` x { a = policy(env); step! } in one launch (shems_rollout_dev).  policy: "rule"
        (action(env, track), track < 0) or "random" (populate_memory's uniform actions).  Returns the
        per-env episode returns as a float64 device tensor; pushes into `ring` if given."""
        import torch
        v = self.view()
        ret = torch.empty(self.n, dtype=torch.float64, device="cuda")
        pol = {"rule": _capi.ROLLOUT_RULE, "random": _capi.ROLLOUT_RANDOM}[policy]
        rs = ring.struct() if ring is not None else None
        _capi.check(self._L.shems_rollout_dev(C.byref(v), pol, int(nsteps), int(seed) & ((1 << 64) - 1),
                                              C.c_void_p(ret.data_ptr()), C.byref(rs) if rs is not None else None,
                                              ring.pos if ring is not None else 0, int(ring_envs), self._stream()))
        if ring is not None:
            ring.pushed += (self.n if ring_envs <= 0 else min(self.n, int(ring_envs))) * int(nsteps)
        return ret

    def step_dev(self, actions, track=0, rewards=None, rewards_f32=None, results=None, block_reward=None):
        """step! on device tensors (zero-copy): actions [N][2] float32 cuda tensor."""
        v = self.view()
        mode = _capi.TRACK_OFF if track == 0 else (_capi.TRACK_DRL if track > 0 else _capi.TRACK_RULE)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        _capi.check(self._L.shems_step_dev(C.byref(v), p(actions), mode, p(rewards), p(rewards_f32), p(results),
                                           p(block_reward), self._stream()))

    def set_stream(self, stream_ptr):
        _capi.check(self._L.shems_set_stream(self._h, C.c_void_p(stream_ptr)))

    def slice(self, start, count):
        """Envs [start, start + count) as a device view of their own (see EnvSlice)."""
        return EnvSlice(self, start, count)

    def check_error(self):
        _capi.check(self._L.shems_check_error(self._h))


EnvSlice.rollout = ShemsBatch.rollout          # same launch, on the slice's view


class Shems:
    """`Shems(maxsteps, path)` of the reference: one household, scalar fields, same names."""

    def __init__(self, maxsteps, path, charger_id=None, device=None, table=None, **weights):
        if charger_id is None:
            job = os.environ.get("JOB_ID")           # LU1:17, 45: third and fourth last digits of JOB_ID
            charger_id = (int(job) // 100) % 100 if job else 98
        if device is None:
            device = int(os.environ.get("GPU_ID", "0"))   # DDPG_reinforce_charger_v1.jl:12-14
        self.path = path
        self.maxsteps = int(maxsteps)
        tab = _tables.load_csv(path) if table is None else np.asarray(table, np.float32)
        cfg = make_config(charger_id, 0, tab.shape[0], **weights)
        self._b = ShemsBatch(1, maxsteps, [tab], [cfg], None, device)
        self._b.state = np.array([[0, 0, -1, 0, 0, 0, 1, 0, 1]], np.float32)      # ShemsState() LU1:115
        self._b.idx = np.array([1], np.int32)
        self.reward = 0.0
        self.a = np.array([0.7, 1.0], np.float32)

    @property
    def state(self):
        return self._b.state[0]

    @property
    def idx(self):
        return int(self._b.idx[0])

    @property
    def step(self):
        return int(self._b.step[0])

    def __len__(self):
        return 7                                      # Base.size(::Shems) = (7,)  LU1:179


def reset_(env, rng=0, **kw):
    if isinstance(env, Shems):
        if "idx0" in kw:
            kw = dict(kw, idx0=[kw["idx0"]], soc_b0=[kw["soc_b0"]])
        env._b.reset_(rng, **kw)
        env.reward = 0.0
        env.a = np.array([0.7, 1.0], np.float32)
        return env
    return env.reset_(rng, **kw)


def step_(env, s, a, track=0):
    if isinstance(env, Shems):
        out = env._b.step_(s, np.asarray(a, np.float32).reshape(1, 2), track)
        env.reward = float(out[0][0])
        env.a = env._b.a[0]
        if track == 0:
            return float(out[0][0]), out[1][0]
        return float(out[0][0]), out[1][0], out[2]          # results is a 1x23 Matrix{Float64}
    return env.step_(s, a, track)


def action(env, a_or_track=-1):
    if isinstance(env, Shems):
        if np.isscalar(a_or_track):
            return env._b.action(a_or_track)[0]
        return env._b.action(np.asarray(a_or_track, np.float32).reshape(1, 2))[0]
    return env.action(a_or_track)


def finished(env, s2=None):
    if isinstance(env, Shems):
        return False
    return env.finished(s2)
