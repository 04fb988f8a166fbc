"""ctypes binding of oracle/libshems_oracle.so (the C restatement of shems_LU1.jl).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libshems_oracle.so")


class Profile(C.Structure):
    _fields_ = [("cap_ev", C.c_float), ("soc_max", C.c_float), ("rate_max", C.c_double),
                ("disc_weight", C.c_double), ("disc_pot", C.c_double), ("penalty_weight", C.c_float)]


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("shems_oracle.c", "shems_policy_omp.c", "shems_oracle.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "libshems_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    vp, i64, fp, dp = C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_double)
    L.orc_profile_for_charger.argtypes = [C.c_int, C.POINTER(Profile)]
    L.orc_env_init.argtypes = [vp, i64, vp, i64, C.POINTER(Profile)]
    L.orc_env_init.restype = None
    L.orc_resolve_start.argtypes = [vp, i64, i64, i64, C.POINTER(C.c_int)]
    L.orc_resolve_start.restype = i64
    L.orc_reset.argtypes = [vp, C.c_int, i64, C.c_float]
    L.orc_action_drl.argtypes = [vp, C.c_float, C.c_float, fp]
    L.orc_action_drl.restype = None
    L.orc_action_rule.argtypes = [vp, fp]
    L.orc_action_rule.restype = None
    L.orc_step.argtypes = [vp, fp, C.c_int, dp, dp]
    L.orc_scale_action.argtypes = [C.c_float]
    L.orc_scale_action.restype = C.c_float
    L.orc_batch_step.argtypes = [vp, i64, vp, C.c_int, vp, vp, vp]
    L.orc_batch_step_omp.argtypes = [vp, i64, vp, C.c_int, vp, vp]
    L.orc_batch_episode_omp.argtypes = [vp, i64, vp, i64, i64, C.c_int, vp]
    L.orc_rule_episode.argtypes = [vp, i64, vp]
    L.orc_rule_episode.restype = C.c_double
    L.orc_batch_alloc.argtypes = [i64]
    L.orc_batch_alloc.restype = vp
    L.orc_batch_free.argtypes = [vp]
    L.orc_batch_free.restype = None
    L.orc_batch_at.argtypes = [vp, i64]
    L.orc_batch_at.restype = vp
    L.orc_env_idx.argtypes = [vp]
    L.orc_env_idx.restype = i64
    L.orc_env_step.argtypes = [vp]
    L.orc_env_step.restype = i64
    L.orc_env_get_state.argtypes = [vp, vp]
    L.orc_env_get_state.restype = None
    L.orc_env_set_state.argtypes = [vp, vp, i64, i64]
    L.orc_env_set_state.restype = None
    L.orc_batch_init.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp]
    L.orc_batch_init.restype = None
    L.orc_batch_reset.argtypes = [vp, i64, C.c_int, vp, vp]
    L.orc_batch_get_state.argtypes = [vp, i64, vp]
    L.orc_batch_get_state.restype = None
    L.orc_batch_set_state.argtypes = [vp, i64, vp, vp, vp]
    L.orc_batch_set_state.restype = None
    L.orc_batch_get_idx.argtypes = [vp, i64, vp, vp]
    L.orc_batch_get_idx.restype = None
    L.orc_batch_action_drl.argtypes = [vp, i64, vp, vp]
    L.orc_batch_action_drl.restype = None
    L.orc_batch_action_rule.argtypes = [vp, i64, vp]
    L.orc_batch_action_rule.restype = None
    L.orc_scale_actions.argtypes = [vp, i64, vp]
    L.orc_scale_actions.restype = None
    L.orc_set_threads.argtypes = [C.c_int]
    L.orc_set_threads.restype = None
    L.orc_policy_step_omp.argtypes = [vp, i64, vp, vp, vp, C.c_float, C.c_uint64, C.c_uint32, C.c_int, vp, vp, vp, vp]
    _lib = L
    return L


def usable_cpus():
    """CPUs this process may actually run on: the affinity mask, cut by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def set_threads(n):
    lib().orc_set_threads(int(n))


def profile(charger_id=98, disc_weight=None, disc_pot=None, penalty_weight=None):
    p = Profile()
    if lib().orc_profile_for_charger(charger_id, C.byref(p)) != 0:
        raise KeyError(charger_id)
    if disc_weight is not None:
        p.disc_weight = float(np.float32(disc_weight))
    if disc_pot is not None:
        p.disc_pot = float(np.float32(disc_pot))
    if penalty_weight is not None:
        p.penalty_weight = float(np.float32(penalty_weight))
    return p


class Batch:
    """n scalar oracle envs sharing tables (kept alive here).  Every accessor is ONE foreign call for the whole batch."""

    def __init__(self, n, maxsteps, tables, profiles, table_of_env=None, profile_of_env=None):
        L = lib()
        self.n = int(n)
        self.maxsteps = int(maxsteps)
        self.tables = [np.ascontiguousarray(t, dtype=np.float32) for t in (tables if isinstance(tables, (list, tuple)) else [tables])]
        self.profiles = list(profiles) if isinstance(profiles, (list, tuple)) else [profiles]
        self.ptr = L.orc_batch_alloc(self.n)
        tptr = (C.c_void_p * len(self.tables))(*[t.ctypes.data for t in self.tables])
        nrows = np.array([t.shape[0] for t in self.tables], np.int64)
        profs = (Profile * len(self.profiles))(*self.profiles)
        to = None if table_of_env is None else np.ascontiguousarray(table_of_env, dtype=np.int64)
        po = None if profile_of_env is None else np.ascontiguousarray(profile_of_env, dtype=np.int64)
        L.orc_batch_init(self.ptr, self.n, self.maxsteps, tptr, nrows.ctypes.data, None if to is None else to.ctypes.data,
                         profs, None if po is None else po.ctypes.data)

    def __del__(self):
        try:
            lib().orc_batch_free(self.ptr)
        except Exception:
            pass

    def at(self, i):
        return lib().orc_batch_at(self.ptr, i)

    def reset(self, rng_is_minus1=True, idx0=None, soc_b0=None):
        i0 = None if idx0 is None else np.ascontiguousarray(idx0, dtype=np.int64)
        s0 = None if soc_b0 is None else np.ascontiguousarray(soc_b0, dtype=np.float32)
        assert (i0 is None or i0.shape == (self.n,)) and (s0 is None or s0.shape == (self.n,))
        return lib().orc_batch_reset(self.ptr, self.n, 1 if rng_is_minus1 else 0, None if i0 is None else i0.ctypes.data,
                                     None if s0 is None else s0.ctypes.data)

    def set_state(self, obs, idx, step=None):
        obs = np.ascontiguousarray(obs, dtype=np.float32)
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        st = None if step is None else np.ascontiguousarray(step, dtype=np.int64)
        assert obs.shape == (self.n, 9) and idx.shape == (self.n,)
        lib().orc_batch_set_state(self.ptr, self.n, obs.ctypes.data, idx.ctypes.data, None if st is None else st.ctypes.data)

    def state(self, out=None):
        out = np.empty((self.n, 9), np.float32) if out is None else out
        lib().orc_batch_get_state(self.ptr, self.n, out.ctypes.data)
        return out

    def idx(self):
        out = np.empty(self.n, np.int64)
        lib().orc_batch_get_idx(self.ptr, self.n, out.ctypes.data, None)
        return out

    def steps(self):
        out = np.empty(self.n, np.int64)
        lib().orc_batch_get_idx(self.ptr, self.n, None, out.ctypes.data)
        return out

    def action_drl(self, targets):
        targets = np.ascontiguousarray(targets, dtype=np.float32)
        assert targets.shape == (self.n, 2)
        out = np.empty((self.n, 2), np.float32)
        lib().orc_batch_action_drl(self.ptr, self.n, targets.ctypes.data, out.ctypes.data)
        return out

    def action_rule(self):
        out = np.empty((self.n, 2), np.float32)
        lib().orc_batch_action_rule(self.ptr, self.n, out.ctypes.data)
        return out

    def step(self, actions, track_mode=0, want_results=False, omp=False):
        L = lib()
        actions = np.ascontiguousarray(actions, dtype=np.float32)
        assert actions.shape == (self.n, 2)
        rewards = np.empty(self.n, np.float64)
        obs = np.empty((self.n, 9), np.float32)
        if omp:
            rc = L.orc_batch_step_omp(self.ptr, self.n, actions.ctypes.data, track_mode,
                                      rewards.ctypes.data, obs.ctypes.data)
            return rc, rewards, obs, None
        res = np.empty((self.n, 23), np.float64) if want_results else None
        rc = L.orc_batch_step(self.ptr, self.n, actions.ctypes.data, track_mode, rewards.ctypes.data,
                              obs.ctypes.data, res.ctypes.data if want_results else None)
        return rc, rewards, obs, res

    def episode_omp(self, action_sets, nsteps, track_mode=0):
        """nsteps x step! for every env inside one OpenMP region (all host cores); action_sets [nsets][n][2], set t % nsets at step t."""
        a = np.ascontiguousarray(action_sets, dtype=np.float32)
        assert a.ndim == 3 and a.shape[1:] == (self.n, 2)
        ret = np.empty(self.n, np.float64)
        rc = lib().orc_batch_episode_omp(self.ptr, self.n, a.ctypes.data, a.shape[0], int(nsteps), track_mode, ret.ctypes.data)
        return rc, ret

    def policy_step_omp(self, actor, s_min, s_max, obs, sigma=0.1, seed=0, tick=0, train=True):
        """The whole vector step (normalize + actor + noise + clamp + scale_action + step!) on all host cores in one OpenMP region
        (shems_policy_omp.c; throughput baseline).  Returns (rc, a [n][2] unscaled, rewards [n], obs' [n][9])."""
        actor = np.ascontiguousarray(actor, dtype=np.float32)
        lo, hi = np.ascontiguousarray(s_min, dtype=np.float32), np.ascontiguousarray(s_max, dtype=np.float32)
        obs = np.ascontiguousarray(obs, dtype=np.float32)
        assert actor.size == 129002 and obs.shape == (self.n, 9)
        a = np.empty((self.n, 2), np.float32)
        rew = np.empty(self.n, np.float64)
        out = np.empty((self.n, 9), np.float32)
        rc = lib().orc_policy_step_omp(self.ptr, self.n, actor.ctypes.data, lo.ctypes.data, hi.ctypes.data, float(sigma), int(seed),
                                       int(tick) & 0xFFFFFFFF, 1 if train else 0, obs.ctypes.data, a.ctypes.data, rew.ctypes.data,
                                       out.ctypes.data)
        return rc, a, rew, out

    def rule_episode(self, i, steps, want_results=False):
        L = lib()
        res = np.empty((steps, 23), np.float64) if want_results else None
        tot = L.orc_rule_episode(self.at(i), steps, res.ctypes.data if want_results else None)
        return tot, res


def resolve_start(table, maxsteps, idx0):
    t = np.ascontiguousarray(table, dtype=np.float32)
    it = C.c_int(0)
    r = lib().orc_resolve_start(t.ctypes.data, t.shape[0], maxsteps, int(idx0), C.byref(it))
    return int(r), it.value


def scale_action(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty_like(a)
    lib().orc_scale_actions(a.ctypes.data, a.size, out.ctypes.data)
    return out
