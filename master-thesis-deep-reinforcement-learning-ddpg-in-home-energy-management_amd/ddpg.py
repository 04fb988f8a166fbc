"""Host side of the DDPG learner path: mirrors the reference's algorithm interface over the C ABI.

Reference names (RL-SHEMS/algorithms/DDPG.jl, src/memory_plotting_saving.jl) -> here:
    actor / critic / actor_target / critic_target (DDPG.jl:30-46)  -> Agent.actor / .critic / .actor_t / .critic_t
    act(s_norm; train)            DDPG.jl:148-176   -> Agent.act(obs, train)
    scale_action(a)               DDPG.jl:178-184   -> fused into the step kernel (shems_core.h)
    remember / getData            MPS:31-47         -> ReplayRing + device sampler
    replay(; rng_rpl)             DDPG.jl:121-145   -> Agent.replay()
    populate_memory / min_max_buffer  MPS:9-29, 50-53 -> Agent.populate_memory() / .min_max_buffer()
    episode! / run_episodes       DDPG.jl:186-298   -> Agent.episode_() / .run_episodes()
All device memory is PyTorch tensors (allocator + streams + torch.distributed only); every numeric
kernel is a hand-written HIP kernel in libshems_hip.so.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from .replay import ReplayRing

L1, L2, STATE, ACTION = 250, 500, 9, 2
N_ACTOR, N_CRITIC = 129002, 129001
f32 = np.float32

# TUNED hyper-parameters (input_templates/input09_08_on_01-09_eval.jl:64-91, README.md:69-87)
GAMMA, TAU, ETA_ACT, ETA_CRIT = 0.99, 1e-3, 1e-4, 1e-3
BATCH_SIZE, MEM_SIZE, NOISE_SIGMA, EP_LENGTH_TRAIN = 120, 24000, 0.1, 72


class ActParams(C.Structure):          # shems_act_params
    _fields_ = [("actor", C.c_void_p), ("s_min", C.c_void_p), ("s_max", C.c_void_p),
                ("noise_mu", C.c_float), ("noise_sigma", C.c_float), ("train", C.c_int32),
                ("tick", C.c_uint32), ("seed", C.c_uint64)]


class DdpgArgs(C.Structure):           # shems_ddpg
    _fields_ = [(n, C.c_void_p) for n in ("actor", "critic", "actor_t", "critic_t", "m_actor", "v_actor", "m_critic",
                                          "v_critic", "grad_actor", "grad_critic", "s_min", "s_max", "ws", "losses")] + \
               [("gamma", C.c_float), ("tau", C.c_float), ("batch", C.c_int32), ("reserved", C.c_int32)]


class RingWindow(C.Structure):         # shems_ring_window
    _fields_ = [("pos", C.c_int64), ("count", C.c_int64), ("offset", C.c_int64)]


def _declare():
    L = _capi.lib()
    if getattr(L, "_ddpg_declared", False):
        return L
    vp, i64 = C.c_void_p, C.c_int64
    L.shems_actor_forward_dev.argtypes = [C.POINTER(ActParams), vp, i64, vp, vp]
    L.shems_actor_forward_dev.restype = C.c_int
    L.shems_act_step_dev.argtypes = [C.POINTER(_capi.View), C.POINTER(ActParams), vp, vp, vp, vp,
                                     C.POINTER(_capi.Replay), C.POINTER(RingWindow), vp]
    L.shems_act_step_dev.restype = C.c_int
    L.shems_act_step_grid.argtypes = [i64, C.POINTER(i64)]
    L.shems_act_step_grid.restype = C.c_int
    PD = C.POINTER(DdpgArgs)
    L.shems_ddpg_workspace_floats.argtypes = [C.POINTER(i64)]
    L.shems_ddpg_critic_grad.argtypes = [PD, C.POINTER(_capi.Replay), i64, C.c_uint64, C.c_uint32, vp]
    L.shems_ddpg_critic_apply.argtypes = [PD, C.c_double, C.c_double, C.c_double, C.c_double, vp]
    L.shems_ddpg_actor_grad.argtypes = [PD, vp]
    L.shems_ddpg_actor_apply.argtypes = [PD, C.c_double, C.c_double, C.c_double, C.c_double, vp]
    L.shems_ddpg_sample_indices.argtypes = [C.c_uint64, C.c_uint32, C.c_int32, i64, vp]
    L.shems_minmax_dev.argtypes = [C.POINTER(_capi.Replay), i64, i64, C.c_uint64, vp, vp, vp]
    for fn in ("shems_ddpg_workspace_floats", "shems_ddpg_critic_grad", "shems_ddpg_critic_apply", "shems_ddpg_actor_grad",
               "shems_ddpg_actor_apply", "shems_ddpg_sample_indices", "shems_minmax_dev"):
        getattr(L, fn).restype = C.c_int
    L._ddpg_declared = True
    return L


# ----------------------------------------------------------- host Philox --
def _philox(c0, c1, c2, c3, k0, k1):
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & 0xFFFFFFFF for c in np.broadcast_arrays(c0, c1, c2, c3))
    mask = np.uint64(0xFFFFFFFF)
    k0, k1 = np.uint64(k0 & 0xFFFFFFFF), np.uint64(k1 & 0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(M0) * c0, np.uint64(M1) * c2
        c0, c1, c2, c3 = (((p1 >> np.uint64(32)) ^ c1 ^ k0) & mask, p1 & mask,
                          ((p0 >> np.uint64(32)) ^ c3 ^ k1) & mask, p0 & mask)
        k0, k1 = (k0 + np.uint64(W0)) & mask, (k1 + np.uint64(W1)) & mask
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


_STREAM_INIT = 0x494E4954


def init_params(seed, in_dim, out_dim, which):
    """Network initialisation of DDPG.jl:21-46 in the flat Flux layout: glorot_uniform for the two
    hidden layers, U(-3e-3, 3e-3) for the last, zero biases.  The uniforms come from Philox (the
    reference's shared MersenneTwister(rng_run) stream is not reproducible outside Julia)."""
    out = []
    for li, (fan_in, fan_out) in enumerate([(in_dim, L1), (L1, L2), (L2, out_dim)]):
        n = fan_in * fan_out
        q = np.arange((n + 3) // 4, dtype=np.uint64)
        xs = _philox(q, li, which, _STREAM_INIT, seed & 0xFFFFFFFF, seed >> 32)
        u = (np.stack(xs, 1).reshape(-1)[:n] >> np.uint32(8)).astype(f32) * f32(1.0 / 16777216.0)
        if li < 2:
            w = (u - f32(0.5)) * f32(np.sqrt(f32(24.0) / f32(fan_in + fan_out)))
        else:
            w = f32(6e-3) * u - f32(3e-3)
        out += [w.astype(f32), np.zeros(fan_out, f32)]
    return np.concatenate(out)


class Agent:
    """The DDPG learner state on one GPU (one replica under data parallelism)."""

    def __init__(self, seed=1231, device=None, sigma=NOISE_SIGMA, mu=0.0):
        import torch
        self.torch = torch
        self.L = _declare()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.seed = int(seed)
        self.sigma, self.mu = float(sigma), float(mu)
        a = init_params(self.seed, STATE, ACTION, 0)
        c = init_params(self.seed, STATE + ACTION, 1, 1)
        assert a.size == N_ACTOR and c.size == N_CRITIC
        self.actor = torch.from_numpy(a).to(self.device)
        self.critic = torch.from_numpy(c).to(self.device)
        self.actor_t = self.actor.clone()          # deepcopy(actor), DDPG.jl:38
        self.critic_t = self.critic.clone()
        self.s_min = torch.zeros(STATE, dtype=torch.float32, device=self.device)
        self.s_max = torch.ones(STATE, dtype=torch.float32, device=self.device)
        self.tick = 0
        # learner state: ADAM moments (opt_act = ADAM(eta_act), opt_crit = ADAM(eta_crit), input.jl:126-127)
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=self.device)
        self.m_actor, self.v_actor, self.m_critic, self.v_critic = z(N_ACTOR), z(N_ACTOR), z(N_CRITIC), z(N_CRITIC)
        self.grad_actor, self.grad_critic = z(N_ACTOR), z(N_CRITIC)
        nws = C.c_int64(0)
        _capi.check(self.L.shems_ddpg_workspace_floats(C.byref(nws)))
        self.ws = z(nws.value)
        self.losses = z(2)
        self.gamma, self.tau = GAMMA, TAU
        self.eta_act, self.eta_crit = float(f32(ETA_ACT)), float(f32(ETA_CRIT))
        self.batch = BATCH_SIZE
        self.bp_actor = [0.9, 0.999]               # Flux ADAM state beta^t (Float64), advanced after every step
        self.bp_critic = [0.9, 0.999]
        self.updates = 0
        self.dist = None                           # torch.distributed module when replicas exchange gradients
        self.world = 1

    # ------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def set_params(self, actor=None, critic=None, sync_targets=True):
        t = self.torch
        if actor is not None:
            self.actor.copy_(t.as_tensor(np.asarray(actor, f32)))
            if sync_targets:
                self.actor_t.copy_(self.actor)
        if critic is not None:
            self.critic.copy_(t.as_tensor(np.asarray(critic, f32)))
            if sync_targets:
                self.critic_t.copy_(self.critic)

    def set_norm(self, s_min, s_max):
        t = self.torch
        self.s_min.copy_(t.as_tensor(np.asarray(s_min, f32)))
        self.s_max.copy_(t.as_tensor(np.asarray(s_max, f32)))

    def _act_params(self, train, tick, actor=None):
        a = self.actor if actor is None else actor
        return ActParams(a.data_ptr(), self.s_min.data_ptr(), self.s_max.data_ptr(), self.mu, self.sigma,
                         1 if train else 0, int(tick) & 0xFFFFFFFF, self.seed)

    def act(self, obs, train=True, tick=None, out=None):
        """act(normalize(s); train): obs [M][9] cuda float32 -> a [M][2] in [-1, 1] (unscaled)."""
        t = self.torch
        m = obs.shape[0]
        if out is None:
            out = t.empty((m, ACTION), dtype=t.float32, device=self.device)
        p = self._act_params(train, self.tick if tick is None else tick)
        _capi.check(self.L.shems_actor_forward_dev(C.byref(p), C.c_void_p(obs.data_ptr()), m,
                                                   C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def act_step(self, env, train=True, tick=None, a_out=None, rewards=None, rewards_f32=None, block_reward=None,
                 ring=None, window=None):
        """One fused vector step: s = env.state; a = act(s); step!(env, s, scale_action(a)); remember(...)."""
        v = env.view()
        p = self._act_params(train, self.tick if tick is None else tick)
        ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        rs = ring.struct() if ring is not None else None
        _capi.check(self.L.shems_act_step_dev(C.byref(v), C.byref(p), ptr(a_out), ptr(rewards), ptr(rewards_f32),
                                              ptr(block_reward), C.byref(rs) if rs is not None else None,
                                              C.byref(window) if window is not None else None, self._stream()))

    # ---------------------------------------------------------------- learner
    def _ddpg_args(self):
        return DdpgArgs(self.actor.data_ptr(), self.critic.data_ptr(), self.actor_t.data_ptr(), self.critic_t.data_ptr(),
                        self.m_actor.data_ptr(), self.v_actor.data_ptr(), self.m_critic.data_ptr(), self.v_critic.data_ptr(),
                        self.grad_actor.data_ptr(), self.grad_critic.data_ptr(), self.s_min.data_ptr(), self.s_max.data_ptr(),
                        self.ws.data_ptr(), self.losses.data_ptr(), self.gamma, self.tau, self.batch, 0)

    def enable_data_parallel(self, dist):
        """Replicas (one per GPU, each with its own env shard and ring) all-reduce gradients over RCCL."""
        self.dist = dist
        self.world = dist.get_world_size()
        for t in (self.actor, self.critic, self.actor_t, self.critic_t):
            dist.broadcast(t, src=0)               # identical initial weights on every replica

    def _allreduce(self, g):
        if self.dist is not None and self.world > 1:
            self.dist.all_reduce(g)                # sum over replicas; the 1/world is folded into ADAM's grad_scale

    def replay(self, ring, tick=None):
        """replay(; rng_rpl) (DDPG.jl:121-145): one DDPG update from `ring`."""
        d = self._ddpg_args()
        st = self._stream()
        rs = ring.struct()
        tick = self.updates if tick is None else tick
        _capi.check(self.L.shems_ddpg_critic_grad(C.byref(d), C.byref(rs), len(ring), self.seed, int(tick) & 0xFFFFFFFF, st))
        self._allreduce(self.grad_critic)
        gs = 1.0 / self.world
        _capi.check(self.L.shems_ddpg_critic_apply(C.byref(d), self.eta_crit, self.bp_critic[0], self.bp_critic[1], gs, st))
        self.bp_critic = [self.bp_critic[0] * 0.9, self.bp_critic[1] * 0.999]
        _capi.check(self.L.shems_ddpg_actor_grad(C.byref(d), st))
        self._allreduce(self.grad_actor)
        _capi.check(self.L.shems_ddpg_actor_apply(C.byref(d), self.eta_act, self.bp_actor[0], self.bp_actor[1], gs, st))
        self.bp_actor = [self.bp_actor[0] * 0.9, self.bp_actor[1] * 0.999]
        self.updates += 1

    def sample_indices(self, tick, ring_len):
        out = np.empty(self.batch, np.int64)
        _capi.check(self.L.shems_ddpg_sample_indices(self.seed, int(tick) & 0xFFFFFFFF, self.batch, int(ring_len),
                                                     out.ctypes.data_as(C.c_void_p)))
        return out

    def min_max_buffer(self, ring, count=None, seed=None):
        """s_min, s_max = min_max_buffer(MIN_EXP_SIZE) (MPS:50-53, main script :30): extrema of s over a
        bootstrap sample (with replacement) of `count` ring entries; with several replicas the 9-float
        extrema are all-reduced (min / max)."""
        rs = ring.struct()
        count = len(ring) if count is None else int(count)
        _capi.check(self.L.shems_minmax_dev(C.byref(rs), len(ring), count, self.seed if seed is None else int(seed),
                                            C.c_void_p(self.s_min.data_ptr()), C.c_void_p(self.s_max.data_ptr()), self._stream()))
        if self.dist is not None and self.world > 1:
            self.dist.all_reduce(self.s_min, op=self.dist.ReduceOp.MIN)
            self.dist.all_reduce(self.s_max, op=self.dist.ReduceOp.MAX)
        return self.s_min, self.s_max

    def act_step_blocks(self, n):
        out = C.c_int64(0)
        _capi.check(self.L.shems_act_step_grid(int(n), C.byref(out)))
        return out.value
