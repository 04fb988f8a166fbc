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
from .parallel import GradSync, shard_envs  # noqa: F401
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
                ("tick", C.c_uint32), ("seed", C.c_uint64), ("noise_kind", C.c_int32), ("ou_theta", C.c_float),
                ("ou_dt", C.c_float), ("eps", C.c_float), ("ou_state", C.c_void_p), ("noise_acc", C.c_void_p)]


NOISE_KINDS = {"gn": 0, "ou": 1, "en": 2, "pn": 0}   # noise_type strings of the reference (DDPG.jl:152-161); "pn" runs the perturbed actor with train = 0
NOISE_ACT = 0.1                                  # noise_act, input.jl:231 (ParamNoise.sigma_target)
EPS_ZETA, EPS_XI0, EPS_XI_MIN = float(np.float32(0.0005)), 0.5, float(np.float32(0.1))   # input.jl:226-228 (Float32 literals)
SEED_INI = 123                                   # input.jl:134; evaluation seeds are "123" * test_ep (DDPG.jl:275)


def eps_schedule(current_episode, mem_size=24000, ep_length=72, zeta=EPS_ZETA, xi_min=EPS_XI_MIN):
    """sample_noise(en::EpsNoise) (DDPG.jl:69-72): xi = Float32(max(0.5 - zeta * (current_episode - MEM_SIZE / EP_LENGTH), xi_min)),
    evaluated in Float64 (0.5 and the quotient are Float64) and rounded once.  No upper clamp: episode 1 gives 0.666."""
    return float(np.float32(max(0.5 - zeta * (current_episode - mem_size / ep_length), xi_min)))


class DdpgArgs(C.Structure):           # shems_ddpg
    _fields_ = [(n, C.c_void_p) for n in ("actor", "critic", "actor_t", "critic_t", "m_actor", "v_actor", "m_critic",
                                          "v_critic", "grad_actor", "grad_critic", "s_min", "s_max", "ws", "losses")] + \
               [("gamma", C.c_float), ("tau", C.c_float), ("batch", C.c_int32), ("flags", C.c_int32)]


class RingWindow(C.Structure):         # shems_ring_window
    _fields_ = [("pos", C.c_int64), ("count", C.c_int64), ("offset", C.c_int64)]


class TrainLoop(C.Structure):          # shems_train_loop: the hour loop of episode! enqueued natively (shems_train_steps)
    _fields_ = [("view", _capi.View), ("act", ActParams), ("ddpg", DdpgArgs), ("ring", _capi.Replay),
                ("rewards_f32", C.c_void_p), ("actor_pub", C.c_void_p * 2), ("window", C.c_int64), ("ring_pushed", C.c_int64),
                ("t", C.c_int64), ("updates", C.c_int64), ("env_seed", C.c_uint64), ("sample_seed", C.c_uint64),
                ("episode", C.c_uint32), ("ep_len", C.c_int32), ("updates_per_step", C.c_int32), ("mode", C.c_int32),
                ("eta_crit", C.c_double), ("bp_crit", C.c_double * 2), ("eta_act", C.c_double), ("bp_act", C.c_double * 2),
                ("sync", C.c_void_p), ("dp", C.c_void_p)]


LOOP_ORDERED, LOOP_PIPELINED, LOOP_PIPELINED_EXACT = 0, 1, 2


def _declare():
    L = _capi.lib()
    if getattr(L, "_ddpg_declared", False):
        return L
    vp, i64 = C.c_void_p, C.c_int64
    L.shems_train_steps.argtypes = [C.POINTER(TrainLoop), i64, vp, vp]
    L.shems_train_loop_join.argtypes = [C.POINTER(TrainLoop), vp, vp]
    L.shems_train_loop_release.argtypes = [C.POINTER(TrainLoop)]
    L.shems_act_step_range_dev.argtypes = [C.POINTER(_capi.View), C.POINTER(ActParams), i64, i64, vp, C.POINTER(_capi.Replay),
                                           C.POINTER(RingWindow), vp]
    L.shems_ddpg_update_dp.argtypes = [C.POINTER(DdpgArgs), C.POINTER(_capi.Replay), i64, C.c_uint64, C.c_uint32, i64, i64, C.c_double, C.c_double,
                                       C.c_double, C.c_double, C.c_double, C.c_double, vp, vp, vp]
    L.shems_ddpg_update_dp.restype = C.c_int
    for fn in ("shems_train_steps", "shems_train_loop_join", "shems_train_loop_release", "shems_act_step_range_dev"):
        getattr(L, fn).restype = C.c_int
    L.shems_actor_forward_dev.argtypes = [C.POINTER(ActParams), vp, i64, vp, vp]
    L.shems_actor_forward_dev.restype = C.c_int
    L.shems_act_step_dev.argtypes = [C.POINTER(_capi.View), C.POINTER(ActParams), vp, vp, vp, vp, vp,
                                     C.POINTER(_capi.Replay), C.POINTER(RingWindow), vp]
    L.shems_act_step_dev.restype = C.c_int
    L.shems_act_step_grid.argtypes = [i64, C.POINTER(i64)]
    L.shems_act_step_grid.restype = C.c_int
    L.shems_act_step_kernel.argtypes = [i64, C.c_int, C.c_char_p, C.c_int32]
    L.shems_act_step_kernel.restype = C.c_int
    PD = C.POINTER(DdpgArgs)
    L.shems_ddpg_workspace_floats.argtypes = [C.POINTER(i64)]
    dbl = C.c_double
    L.shems_ddpg_update.argtypes = [PD, C.POINTER(_capi.Replay), i64, C.c_uint64, C.c_uint32, i64, i64, dbl, dbl, dbl, dbl, dbl, dbl, vp, vp]
    L.shems_ddpg_update.restype = C.c_int
    L.shems_ddpg_critic_grad.argtypes = [PD, C.POINTER(_capi.Replay), i64, C.c_uint64, C.c_uint32, vp]
    L.shems_ddpg_critic_grad_ex.argtypes = [PD, C.POINTER(_capi.Replay), i64, C.c_uint64, C.c_uint32, i64, i64, vp]
    L.shems_ddpg_critic_grad_ex.restype = C.c_int
    L.shems_ddpg_actor_apply_pub.argtypes = [PD, C.c_double, C.c_double, C.c_double, C.c_double, vp, vp]
    L.shems_ddpg_actor_apply_pub.restype = C.c_int
    L.shems_ddpg_critic_apply.argtypes = [PD, C.c_double, C.c_double, C.c_double, C.c_double, vp]
    L.shems_ddpg_actor_grad.argtypes = [PD, vp]
    L.shems_ddpg_actor_prepare.argtypes = [PD, vp]
    L.shems_ddpg_actor_prepare.restype = C.c_int
    L.shems_ddpg_actor_apply.argtypes = [PD, C.c_double, C.c_double, C.c_double, C.c_double, vp]
    L.shems_ddpg_sample_indices.argtypes = [C.c_uint64, C.c_uint32, C.c_int32, i64, vp]
    L.shems_minmax_dev.argtypes = [C.POINTER(_capi.Replay), i64, i64, C.c_uint64, vp, vp, vp]
    L.shems_ddpg_perturb_dev.argtypes = [vp, vp, i64, C.c_float, vp]
    L.shems_ddpg_combine_dev.argtypes = [vp, vp, i64, C.c_float, C.c_float, vp]
    L.shems_ddpg_combine_dev.restype = C.c_int
    L.shems_ddpg_batch_obs_dev.argtypes = [PD, C.POINTER(_capi.Replay), vp, vp]
    L.shems_action_distance_dev.argtypes = [vp, vp, i64, vp, vp]
    for fn in ("shems_ddpg_perturb_dev", "shems_ddpg_batch_obs_dev", "shems_action_distance_dev"):
        getattr(L, fn).restype = C.c_int
    for fn in ("shems_ddpg_workspace_floats", "shems_ddpg_critic_grad", "shems_ddpg_critic_apply", "shems_ddpg_actor_grad",
               "shems_ddpg_actor_apply", "shems_ddpg_sample_indices", "shems_minmax_dev"):
        getattr(L, fn).restype = C.c_int
    i32 = C.c_int32
    L.shems_wide_params.argtypes = [i32, i32, C.POINTER(i64), C.POINTER(i64)]
    L.shems_wide_workspace_floats.argtypes = [i32, i32, C.POINTER(i64)]
    L.shems_wide_act_workspace_floats.argtypes = [i32, i32, i64, C.POINTER(i64)]
    L.shems_wide_actor_forward_dev.argtypes = [C.POINTER(ActParams), i32, i32, vp, i64, vp, vp, vp]
    L.shems_wide_act_step_dev.argtypes = [C.POINTER(_capi.View), C.POINTER(ActParams), i32, i32, vp, vp, vp, vp, vp,
                                          C.POINTER(_capi.Replay), C.POINTER(RingWindow), vp]
    L.shems_wide_critic_grad_ex.argtypes = [PD, i32, i32, C.POINTER(_capi.Replay), i64, C.c_uint64, C.c_uint32, i64, i64, vp]
    L.shems_wide_critic_apply.argtypes = [PD, i32, i32, dbl, dbl, dbl, dbl, vp]
    L.shems_wide_actor_grad.argtypes = [PD, i32, i32, vp]
    L.shems_wide_actor_apply_pub.argtypes = [PD, i32, i32, dbl, dbl, dbl, dbl, vp, vp]
    L.shems_wide_batch_slots.argtypes = [PD, i32, i32, vp, vp]
    for fn in ("shems_wide_params", "shems_wide_workspace_floats", "shems_wide_act_workspace_floats", "shems_wide_actor_forward_dev",
               "shems_wide_act_step_dev", "shems_wide_critic_grad_ex", "shems_wide_critic_apply", "shems_wide_actor_grad",
               "shems_wide_actor_apply_pub", "shems_wide_batch_slots"):
        getattr(L, fn).restype = C.c_int
    L._ddpg_declared = True
    return L


def act_kernel_name(n_envs, grouped=False):
    """The kernel the fused-step dispatcher runs for n_envs envs, by its profiler name (shems_act_step_kernel: the dispatcher's own
    decision, environment overrides included) -- bench.py's roofline.kernel.  grouped: False / True = a learner group in Flux order /
    2 = a learner group on the tiled working layout."""
    buf = C.create_string_buffer(96)
    _capi.check(_declare().shems_act_step_kernel(int(n_envs), int(grouped), buf, 96))
    return buf.value.decode()


def act_algorithmic_bytes(n_envs, inserted):
    """HBM bytes one fused vector step has to move: 92 B per env-step (obs 36 + action 8 + idx 4 in, obs' 36 +
    reward 4 + idx 4 out), the actor's parameter block once, 85 B per transition pushed into the ring."""
    return 92 * int(n_envs) + 4 * N_ACTOR + 85 * int(inserted)


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
_STREAM_PERTURB = 0x50455254


def perturb_shift(seed, tick, mu, sigma):
    """sample_noise(pn, rng) (DDPG.jl:63-67): one scalar Normal(mu, sigma_current) draw per (seed, tick), Float32.
    Box-Muller on two Philox words (stream PERT); the reference's Random.seed!(rng) stream is Julia-only."""
    x = _philox(int(tick) & 0xFFFFFFFF, 0, 0, _STREAM_PERTURB, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u1 = (float(x[0] >> np.uint32(8)) + 0.5) / 16777216.0
    u2 = (float(x[1] >> np.uint32(8)) + 0.5) / 16777216.0
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return float(f32(mu + sigma * z))


def net_size(in_dim, out_dim, hidden=(L1, L2)):
    h1, h2 = hidden
    return in_dim * h1 + h1 + h1 * h2 + h2 + h2 * out_dim + out_dim


def _blocks(flat, in_dim, out_dim, hidden):
    h1, h2 = hidden
    o = np.cumsum([0, in_dim * h1, h1, h1 * h2, h2, h2 * out_dim, out_dim])
    shapes = [(in_dim, h1), (h1,), (h1, h2), (h2,), (h2, out_dim), (out_dim,)]
    return [np.asarray(flat[o[i]:o[i + 1]]).reshape(sh) for i, sh in enumerate(shapes)]


def is_wide(hidden):
    """Hidden sizes the (250, 500) tile maps cannot hold (the reference grids' (300, 600)): such a network keeps its own flat layout
    and runs through the shems_wide_* entry points (csrc/shems_wide.hip), layer by layer."""
    return int(hidden[0]) > L1 or int(hidden[1]) > L2


def pad_net(flat, in_dim, out_dim, hidden):
    """A (in -> h1 -> h2 -> out) network, h1 <= 250, h2 <= 500, in the flat Flux layout -> the (250, 500) layout the kernels are built
    for, the extra hidden units carrying ZERO weights and biases.  The padded network computes the same function, and it stays padded
    under training: an extra unit's pre-activation is exactly 0, relu and its mask (pre > 0) give 0, so every gradient entry that
    touches it is exactly 0 and ADAM (m = v = 0 -> step 0 / (0 + eps)) and the soft update leave the zeros in place.  This is how the
    (200, 400) and (150, 300) points of the reference's grids (input09_08_on_01-09_eval.jl:62-66, input.jl:58-66) run on this build."""
    h1, h2 = hidden
    if (h1, h2) == (L1, L2):
        return np.asarray(flat, f32).copy()
    if not (1 <= h1 <= L1 and 1 <= h2 <= L2):
        raise NotImplementedError(f"hidden sizes {hidden} do not fit the ({L1}, {L2}) layout: a wider network is not padded, it keeps its own "
                                  "layout (is_wide / Agent(hidden=...) -> shems_wide_*)")
    W1, b1, W2, b2, W3, b3 = _blocks(np.asarray(flat, f32), in_dim, out_dim, hidden)
    P1, q1, P2, q2, P3 = (np.zeros(sh, f32) for sh in ((in_dim, L1), (L1,), (L1, L2), (L2,), (L2, out_dim)))
    P1[:, :h1], q1[:h1], P2[:h1, :h2], q2[:h2], P3[:h2] = W1, b1, W2, b2, W3
    return np.concatenate([P1.ravel(), q1, P2.ravel(), q2, P3.ravel(), b3.astype(f32)])


def unpad_net(flat, in_dim, out_dim, hidden):
    """The inverse of pad_net (what a checkpoint of the smaller network holds)."""
    h1, h2 = hidden
    if (h1, h2) == (L1, L2):
        return np.asarray(flat, f32).copy()
    W1, b1, W2, b2, W3, b3 = _blocks(np.asarray(flat, f32), in_dim, out_dim, (L1, L2))
    return np.concatenate([W1[:, :h1].ravel(), b1[:h1], W2[:h1, :h2].ravel(), b2[:h2], W3[:h2].ravel(), b3])


def init_params(seed, in_dim, out_dim, which, hidden=(L1, L2)):
    """Network initialisation of DDPG.jl:21-46 in the flat Flux layout: glorot_uniform for the two
    hidden layers, U(-3e-3, 3e-3) for the last, zero biases.  The uniforms come from Philox (the
    reference's shared MersenneTwister(rng_run) stream is not reproducible outside Julia).  hidden != (250, 500): the network
    of that size (true fan-in / fan-out in the glorot bound), returned in ITS OWN flat layout -- pad_net embeds it."""
    out = []
    h1, h2 = hidden
    for li, (fan_in, fan_out) in enumerate([(in_dim, h1), (h1, h2), (h2, out_dim)]):
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

    def __init__(self, seed=1231, device=None, sigma=NOISE_SIGMA, mu=0.0, rng_seed=None, noise_type="gn", theta=0.15,
                 dt=1e-2, eps=0.5, tensors=None, hidden=(L1, L2), wide=None):
        """tensors: dict of float32 device views (actor, critic, actor_t, critic_t, m_actor, v_actor, m_critic, v_critic,
        grad_actor, grad_critic, s_min, s_max, ws, losses) in memory the caller owns -- a learner group's slab -- instead
        of buffers allocated here.  wide: None = by size (is_wide); True = the layer-by-layer path whatever the size (tests hold the two
        implementations against each other at (250, 500))."""
        import torch
        self.torch = torch
        self.L = _declare()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.seed = int(seed)                      # network initialisation (identical on every replica)
        self.rng_seed = self.seed if rng_seed is None else int(rng_seed)   # noise / minibatch streams (per replica)
        self.sigma, self.mu = float(sigma), float(mu)
        self.noise_type, self.theta, self.dt, self.eps = noise_type, float(theta), float(dt), float(eps)
        self.ou_state = None                      # OUNoise.X per env, allocated on first use
        # pn = ParamNoise(mu, sigma, noise_act, 1.01), input.jl:237
        self.pn_sigma, self.pn_target, self.pn_adoption, self.pn_shift = float(f32(sigma)), NOISE_ACT, 1.01, 0.0
        self.actor_perturb = None                  # actor_perturb = deepcopy(actor), DDPG.jl:39
        self.hidden = (int(hidden[0]), int(hidden[1]))        # Dense widths; smaller than (250, 500): zero-padded (pad_net); larger: wide path
        self.wide = is_wide(self.hidden) if wide is None else bool(wide)
        if is_wide(self.hidden) and not self.wide:
            raise NotImplementedError(f"hidden sizes {self.hidden} do not fit the ({L1}, {L2}) kernels")
        if self.hidden != (L1, L2) and not self.wide and noise_type == "pn":
            raise NotImplementedError("parameter noise adds one scalar to EVERY parameter (DDPG.jl:89-96): it would un-zero the padding of a smaller network")
        nws = C.c_int64(0)
        if self.wide:
            if tensors is not None:
                raise NotImplementedError("learner groups run the (250, 500) kernels: no slabs of wide networks")
            a = init_params(self.seed, STATE, ACTION, 0, self.hidden)
            c = init_params(self.seed, STATE + ACTION, 1, 1, self.hidden)
            _capi.check(self.L.shems_wide_workspace_floats(*self.hidden, C.byref(nws)))
        else:
            a = pad_net(init_params(self.seed, STATE, ACTION, 0, self.hidden), STATE, ACTION, self.hidden)
            c = pad_net(init_params(self.seed, STATE + ACTION, 1, 1, self.hidden), STATE + ACTION, 1, self.hidden)
            _capi.check(self.L.shems_ddpg_workspace_floats(C.byref(nws)))
        N_ACTOR, N_CRITIC = a.size, c.size                    # (250, 500) layout: 129 002 / 129 001; a wide network: its own counts
        self.n_actor, self.n_critic = N_ACTOR, N_CRITIC
        assert self.wide or (N_ACTOR, N_CRITIC) == (globals()["N_ACTOR"], globals()["N_CRITIC"])
        self._act_ws = None                                   # wide path: scratch of the layer-by-layer forward (normalised obs, hidden layers)
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=self.device)
        if tensors is None:
            tensors = dict(actor=z(N_ACTOR), critic=z(N_CRITIC), actor_t=z(N_ACTOR), critic_t=z(N_CRITIC),
                           m_actor=z(N_ACTOR), v_actor=z(N_ACTOR), m_critic=z(N_CRITIC), v_critic=z(N_CRITIC),
                           grad_actor=z(N_ACTOR), grad_critic=z(N_CRITIC), s_min=z(STATE), s_max=z(STATE), ws=z(nws.value), losses=z(2))
        for k, n in (("actor", N_ACTOR), ("critic", N_CRITIC), ("actor_t", N_ACTOR), ("critic_t", N_CRITIC), ("m_actor", N_ACTOR),
                     ("v_actor", N_ACTOR), ("m_critic", N_CRITIC), ("v_critic", N_CRITIC), ("grad_actor", N_ACTOR),
                     ("grad_critic", N_CRITIC), ("s_min", STATE), ("s_max", STATE), ("ws", nws.value), ("losses", 2)):
            t = tensors[k]
            assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n, k
            setattr(self, k, t)
        self.device = self.actor.device
        self.actor.copy_(torch.from_numpy(a))
        self.critic.copy_(torch.from_numpy(c))
        self.actor_t.copy_(self.actor)             # deepcopy(actor), DDPG.jl:38
        self.critic_t.copy_(self.critic)
        self.s_min.zero_()
        self.s_max.fill_(1.0)
        self.tick = 0
        # learner state: ADAM moments (opt_act = ADAM(eta_act), opt_crit = ADAM(eta_crit), input.jl:126-127)
        for k in ("m_actor", "v_actor", "m_critic", "v_critic", "grad_actor", "grad_critic", "ws", "losses"):
            getattr(self, k).zero_()
        self.gamma, self.tau = GAMMA, TAU
        self.eta_act, self.eta_crit = float(f32(ETA_ACT)), float(f32(ETA_CRIT))
        self.batch = BATCH_SIZE
        self.bp_actor = [0.9, 0.999]               # Flux ADAM state beta^t (Float64), advanced after every step
        self.bp_critic = [0.9, 0.999]
        self.updates = 0
        self.sync = GradSync(None)                 # replicas exchange gradients through this (RCCL)
        self.dp_overlap = False                    # data parallel: True = the critic's gradient all-reduce runs asynchronously under the actor's E
                                                   # products (shems_ddpg_actor_prepare); False = everything in program order (same bits).
                                                   # Off by default since it was measured on real RCCL streams (one-rank group,
                                                   # tools/dp_one_rank_steps.py): the asynchronous form costs 20 us per step MORE than
                                                   # program order (207 against 187 us at 65 536 envs) to hide 4 us of E products
        self.fused = True                          # single replica: replay() = ONE call, shems_ddpg_update (5 launches, ADAM inside the
                                                   # gradient launches); False = the split calls the data-parallel path uses (same bits)

    # ------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def _same_device(self, env=None, ring=None):
        """Every buffer a launch dereferences must live on this learner's GPU (one process per GPU: a ring or env
        created on another device would be unmapped peer memory inside the kernels)."""
        if ring is not None and ring.s.device != self.device:
            raise ValueError(f"replay ring lives on {ring.s.device}, the learner on {self.device}")
        if env is not None and getattr(env, "device_index", self.device.index) != self.device.index:
            raise ValueError(f"env batch lives on cuda:{env.device_index}, the learner on {self.device}")

    _LEARNER_TENSORS = ("actor", "critic", "actor_t", "critic_t", "m_actor", "v_actor", "m_critic", "v_critic", "grad_actor",
                        "grad_critic", "losses")

    def snapshot(self):
        """Everything replay() mutates (device clones + the host-side ADAM powers / counters)."""
        return ({k: getattr(self, k).clone() for k in self._LEARNER_TENSORS},
                dict(bp_actor=list(self.bp_actor), bp_critic=list(self.bp_critic), updates=self.updates, tick=self.tick,
                     pn_sigma=self.pn_sigma, fused=self.fused))

    def restore(self, snap):
        hook = getattr(self, "_before_param_write", None)
        if hook is not None:
            hook()
        tensors, host = snap
        for k, v in tensors.items():
            getattr(self, k).copy_(v)
        for k, v in host.items():
            setattr(self, k, list(v) if isinstance(v, list) else v)

    def export_actor(self, tensor=None):
        """The actor (or `tensor`, e.g. the target) as the flat Flux-layout vector of ITS network size (what a checkpoint holds)."""
        src = self.actor if tensor is None else tensor
        if self.wide:
            return src.detach().cpu().numpy().copy()
        return unpad_net(src.detach().cpu().numpy(), STATE, ACTION, self.hidden)

    def export_critic(self, tensor=None):
        src = self.critic if tensor is None else tensor
        if self.wide:
            return src.detach().cpu().numpy().copy()
        return unpad_net(src.detach().cpu().numpy(), STATE + ACTION, 1, self.hidden)

    def set_params(self, actor=None, critic=None, sync_targets=True):
        """actor / critic: flat Flux-layout vectors, either of this learner's network size (padded here) or already in the (250, 500)
        layout."""
        t = self.torch
        hook = getattr(self, "_before_param_write", None)      # a learner of a group on the tiled working layout (group.py)
        if hook is not None:
            hook()
        for name, vec, n_lay, n_own in (("actor", actor, self.n_actor, net_size(STATE, ACTION, self.hidden)),
                                        ("critic", critic, self.n_critic, net_size(STATE + ACTION, 1, self.hidden))):
            if vec is not None and np.asarray(vec).size not in (n_lay, n_own):
                raise ValueError(f"set_params: {name} has {np.asarray(vec).size} parameters; a {self.hidden} network holds {n_own}"
                                 + ("" if n_lay == n_own else f" ({n_lay} in the kernels' padded layout)"))
        if actor is not None and np.asarray(actor).size != self.n_actor:
            actor = pad_net(actor, STATE, ACTION, self.hidden)
        if critic is not None and np.asarray(critic).size != self.n_critic:
            critic = pad_net(critic, STATE + ACTION, 1, self.hidden)
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

    def _act_params(self, train, tick, actor=None, noise_acc=None):
        a = self.actor if actor is None else actor
        return ActParams(a.data_ptr(), self.s_min.data_ptr(), self.s_max.data_ptr(), self.mu, self.sigma,
                         1 if train else 0, int(tick) & 0xFFFFFFFF, self.rng_seed, NOISE_KINDS[self.noise_type], self.theta,
                         self.dt, self.eps, self.ou_state.data_ptr() if self.ou_state is not None else None,
                         noise_acc.data_ptr() if noise_acc is not None else None)

    def _ensure_ou(self, n):
        if self.noise_type == "ou" and (self.ou_state is None or self.ou_state.shape[0] != n):
            self.ou_state = self.torch.zeros((n, ACTION), dtype=self.torch.float32, device=self.device)   # X = zeros(Float32, 2)

    def add_perturb_(self, rng):
        """add_perturb!(rng) (DDPG.jl:89-96): actor_perturb = actor .+ one scalar N(mu, sigma_current) draw."""
        if self.actor_perturb is None:
            self.actor_perturb = self.torch.empty_like(self.actor)
        self.pn_shift = perturb_shift(self.rng_seed, rng, self.mu, self.pn_sigma)
        _capi.check(self.L.shems_ddpg_perturb_dev(C.c_void_p(self.actor.data_ptr()), C.c_void_p(self.actor_perturb.data_ptr()),
                                                  self.n_actor, self.pn_shift, self._stream()))
        return self.actor_perturb

    def adapt_param_noise_(self, ring, rng):
        """adapt_param_noise!(s_norm, rng_rpl) (DDPG.jl:74-87) on the minibatch the running replay() sampled: sigma_current
        shrinks / grows by `adoption` when the perturbed actor's actions are farther / nearer than sigma_target."""
        t = self.torch
        d = self._ddpg_args()
        rs = ring.struct()
        st = self._stream()
        if self.wide:
            slots = np.empty(self.batch, np.int32)
            _capi.check(self.L.shems_wide_batch_slots(C.byref(d), *self.hidden, slots.ctypes.data_as(C.c_void_p), st))
            obs = ring.s[t.as_tensor(slots.astype(np.int64), device=self.device)].contiguous()
        else:
            obs = t.empty((self.batch, STATE), dtype=t.float32, device=self.device)
            _capi.check(self.L.shems_ddpg_batch_obs_dev(C.byref(d), C.byref(rs), C.c_void_p(obs.data_ptr()), st))
        a = self.act(obs, train=False)
        self.add_perturb_(rng)
        a_p = self.act(obs, train=False, actor=self.actor_perturb)
        dist = t.empty(1, dtype=t.float32, device=self.device)
        _capi.check(self.L.shems_action_distance_dev(C.c_void_p(a.data_ptr()), C.c_void_p(a_p.data_ptr()), a.numel(),
                                                     C.c_void_p(dist.data_ptr()), st))
        distance = float(dist.item())
        if distance > self.pn_target:
            self.pn_sigma /= self.pn_adoption
        else:
            self.pn_sigma *= self.pn_adoption
        return distance

    def _explore(self, train, tick):
        """(train flag, actor block) the policy kernel runs with: "pn" evaluates the freshly perturbed copy without action
        noise (DDPG.jl:152-156), every other noise type the actor itself."""
        if train and self.noise_type == "pn":
            return False, self.add_perturb_(tick)
        return train, None

    def act(self, obs, train=True, tick=None, out=None, actor=None):
        """act(normalize(s); train): obs [M][9] cuda float32 -> a [M][2] in [-1, 1] (unscaled)."""
        t = self.torch
        if isinstance(obs, tuple):                       # (device pointer, rows): e.g. the env handle's resident observations
            ptr, m = int(obs[0]), int(obs[1])
        else:
            ptr, m = obs.data_ptr(), obs.shape[0]
        if out is None:
            out = t.empty((m, ACTION), dtype=t.float32, device=self.device)
        self._ensure_ou(m)
        tick = self.tick if tick is None else tick
        if actor is None:
            train, actor = self._explore(train, tick)
        p = self._act_params(train, tick, actor)
        if self.wide:
            _capi.check(self.L.shems_wide_actor_forward_dev(C.byref(p), *self.hidden, C.c_void_p(ptr), m, C.c_void_p(out.data_ptr()),
                                                            C.c_void_p(self._wide_act_ws(m).data_ptr()), self._stream()))
            return out
        _capi.check(self.L.shems_actor_forward_dev(C.byref(p), C.c_void_p(ptr), m,
                                                   C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def _wide_act_ws(self, m):
        """Scratch of the wide path's forward for m observations (normalised observations, both hidden layers, pre-activation outputs):
        m x (9 + l1 + l2 + 2) floats in HBM, kept between calls."""
        need = C.c_int64(0)
        _capi.check(self.L.shems_wide_act_workspace_floats(*self.hidden, int(m), C.byref(need)))
        if self._act_ws is None or self._act_ws.numel() < need.value:
            self._act_ws = self.torch.empty(need.value, dtype=self.torch.float32, device=self.device)
        return self._act_ws

    def act_step(self, env, train=True, tick=None, a_out=None, rewards=None, rewards_f32=None, block_reward=None,
                 returns_acc=None, ring=None, window=None, noise_acc=None):
        """One fused vector step: s = env.state; a = act(s); step!(env, s, scale_action(a)); remember(...)."""
        self._same_device(env, ring)
        env.use_torch_stream()
        v = env.view()
        self._ensure_ou(env.n)
        tick = self.tick if tick is None else tick
        train, actor = self._explore(train, tick)
        p = self._act_params(train, tick, actor, noise_acc=noise_acc)
        ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        rs = ring.struct() if ring is not None else None
        if self.wide:
            if block_reward is not None:
                raise NotImplementedError("per-workgroup reward sums are a by-product of the fused (250, 500) kernel")
            _capi.check(self.L.shems_wide_act_step_dev(C.byref(v), C.byref(p), *self.hidden, C.c_void_p(self._wide_act_ws(env.n).data_ptr()),
                                                       ptr(a_out), ptr(rewards), ptr(rewards_f32), ptr(returns_acc),
                                                       C.byref(rs) if rs is not None else None,
                                                       C.byref(window) if window is not None else None, self._stream()))
            return
        _capi.check(self.L.shems_act_step_dev(C.byref(v), C.byref(p), ptr(a_out), ptr(rewards), ptr(rewards_f32),
                                              ptr(block_reward), ptr(returns_acc), C.byref(rs) if rs is not None else None,
                                              C.byref(window) if window is not None else None, self._stream()))

    # ---------------------------------------------------------------- learner
    MAX_PASS_BATCH = 128                           # minibatch columns one update pass holds (csrc/shems_ddpg.hip: BP)

    def _ddpg_args(self, sub=None):
        """sub: a sub-batch record (its own workspace, gradient buffers, losses, size) of a minibatch wider than one pass."""
        ga, gc, ws, ls, b = ((self.grad_actor, self.grad_critic, self.ws, self.losses, self.batch) if sub is None else
                             (sub["ga"], sub["gc"], sub["ws"], sub["losses"], sub["batch"]))
        defer = self.sync.world > 1 and self.dp_overlap and sub is None and self.batch <= self.MAX_PASS_BATCH and not self.wide
        return DdpgArgs(self.actor.data_ptr(), self.critic.data_ptr(), self.actor_t.data_ptr(), self.critic_t.data_ptr(),
                        self.m_actor.data_ptr(), self.v_actor.data_ptr(), self.m_critic.data_ptr(), self.v_critic.data_ptr(),
                        ga.data_ptr(), gc.data_ptr(), self.s_min.data_ptr(), self.s_max.data_ptr(),
                        ws.data_ptr(), ls.data_ptr(), self.gamma, self.tau, b,
                        1 if defer else 0)                                                    # SHEMS_DDPG_DEFER_ACTOR_E

    def sub_batches(self):
        """BATCH_SIZE > 128: the near-equal sub-batch sizes replay() runs (150 -> 75 + 75, 200 -> 100 + 100), each with its own
        workspace and gradient buffers (allocated on first use)."""
        k = -(-self.batch // self.MAX_PASS_BATCH)
        sizes = [self.batch // k + (1 if i < self.batch % k else 0) for i in range(k)]
        cur = getattr(self, "_subs", None)
        if cur is None or [x["batch"] for x in cur] != sizes:
            t = self.torch
            z = lambda n: t.zeros(n, dtype=t.float32, device=self.device)
            self._subs = [dict(batch=b, ws=z(self.ws.numel()), gc=z(self.n_critic), ga=z(self.n_actor), losses=z(2)) for b in sizes]
        return self._subs

    # the four split-form calls of replay(), on the tuned kernels or -- a network wider than (250, 500) -- layer by layer (shems_wide_*)
    def _critic_grad_ex(self, d, rs, ring_len, tick, ex_pos, ex_cnt, st):
        if self.wide:
            return _capi.check(self.L.shems_wide_critic_grad_ex(C.byref(d), *self.hidden, C.byref(rs), ring_len, self.rng_seed,
                                                                int(tick) & 0xFFFFFFFF, ex_pos, ex_cnt, st))
        _capi.check(self.L.shems_ddpg_critic_grad_ex(C.byref(d), C.byref(rs), ring_len, self.rng_seed, int(tick) & 0xFFFFFFFF, ex_pos, ex_cnt, st))

    def _critic_apply(self, d, gs, st):
        if self.wide:
            return _capi.check(self.L.shems_wide_critic_apply(C.byref(d), *self.hidden, self.eta_crit, self.bp_critic[0], self.bp_critic[1], gs, st))
        _capi.check(self.L.shems_ddpg_critic_apply(C.byref(d), self.eta_crit, self.bp_critic[0], self.bp_critic[1], gs, st))

    def _actor_grad(self, d, st):
        if self.wide:
            return _capi.check(self.L.shems_wide_actor_grad(C.byref(d), *self.hidden, st))
        _capi.check(self.L.shems_ddpg_actor_grad(C.byref(d), st))

    def _actor_apply_pub(self, d, gs, publish, st):
        pub = C.c_void_p(publish.data_ptr()) if publish is not None else None
        if self.wide:
            return _capi.check(self.L.shems_wide_actor_apply_pub(C.byref(d), *self.hidden, self.eta_act, self.bp_actor[0], self.bp_actor[1], gs, pub, st))
        _capi.check(self.L.shems_ddpg_actor_apply_pub(C.byref(d), self.eta_act, self.bp_actor[0], self.bp_actor[1], gs, pub, st))

    def _replay_wide(self, ring, tick, ex_pos, ex_cnt, publish):
        """replay() for BATCH_SIZE > 128: the loss is a mean over the minibatch, so its gradient is the size-weighted mean of the
        sub-batches' gradients (shems_ddpg_combine_dev); ONE ADAM step per network, in the reference's order (critic first, the actor
        through the updated critic, DDPG.jl:134-140).  Sub-batch i draws its indices with sampler tick `tick * 8 + i`."""
        st = self._stream()
        rs = ring.struct()
        subs = self.sub_batches()
        vp = lambda x: C.c_void_p(x.data_ptr())
        for i, sb in enumerate(subs):
            d = self._ddpg_args(sb)
            self._critic_grad_ex(d, rs, len(ring), int(tick) * 8 + i, ex_pos, ex_cnt, st)
            w = sb["batch"] / self.batch
            _capi.check(self.L.shems_ddpg_combine_dev(vp(self.grad_critic), vp(sb["gc"]), self.n_critic, 0.0 if i == 0 else 1.0, w, st))
            _capi.check(self.L.shems_ddpg_combine_dev(vp(self.losses), vp(sb["losses"]), 1, 0.0 if i == 0 else 1.0, w, st))
        self._allreduce(self.grad_critic)
        gs = self.sync.grad_scale
        # the ADAM / soft-update sweeps work on the learner's own (combined) gradient buffers; they do not look at `batch`
        dm = self._ddpg_args(dict(ga=self.grad_actor, gc=self.grad_critic, ws=self.ws, losses=self.losses, batch=subs[0]["batch"]))
        self._critic_apply(dm, gs, st)
        self.bp_critic = [self.bp_critic[0] * 0.9, self.bp_critic[1] * 0.999]
        for i, sb in enumerate(subs):
            d = self._ddpg_args(sb)
            self._actor_grad(d, st)
            w = sb["batch"] / self.batch
            _capi.check(self.L.shems_ddpg_combine_dev(vp(self.grad_actor), vp(sb["ga"]), self.n_actor, 0.0 if i == 0 else 1.0, w, st))
            _capi.check(self.L.shems_ddpg_combine_dev(vp(self.losses[1:]), vp(sb["losses"][1:]), 1, 0.0 if i == 0 else 1.0, w, st))
        self._allreduce(self.grad_actor)
        self._actor_apply_pub(dm, gs, publish, st)
        self.bp_actor = [self.bp_actor[0] * 0.9, self.bp_actor[1] * 0.999]
        self.updates += 1

    def enable_data_parallel(self, dist, native=None, direct=False):
        """Replicas (one per GPU, each with its own env shard and ring) all-reduce gradients over RCCL.  native: a shems_dp communicator
        (parallel.native_comm) -- the all-reduces then run in the update's own stream, from native code; direct: that record exchanges
        through peer-mapped inboxes instead (parallel.direct_comm)."""
        self.sync = GradSync(dist, native=native, direct=direct)
        self.sync.broadcast(self.actor, self.critic, self.actor_t, self.critic_t)   # identical initial weights

    def _allreduce(self, g):
        self.sync.sum_(g)                          # sum over replicas; the 1/world is folded into ADAM's grad_scale

    def replay(self, ring, tick=None, exclude=None, publish=None):
        """replay(; rng_rpl) (DDPG.jl:121-145): one DDPG update from `ring`.  exclude = (pos, count): ring slots another
        stream is writing right now (pipelined mode); publish = tensor that also receives the updated actor."""
        self._same_device(None, ring)
        d = self._ddpg_args()
        st = self._stream()
        rs = ring.struct()
        tick = self.updates if tick is None else tick
        ex_pos, ex_cnt = (0, 0) if exclude is None else (int(exclude[0]) % ring.capacity, int(exclude[1]))
        if self.batch > self.MAX_PASS_BATCH:
            if self.noise_type == "pn":
                raise NotImplementedError("parameter-noise adaptation with BATCH_SIZE > 128")
            return self._replay_wide(ring, tick, ex_pos, ex_cnt, publish)
        if self.fused and self.sync.world == 1 and self.noise_type != "pn" and not self.wide:
            # one replica, nothing to exchange and no parameter-noise adaptation between getData and the updates: the whole
            # replay() is one call (K1..K5, csrc/shems_ddpg.hip)
            _capi.check(self.L.shems_ddpg_update(C.byref(d), C.byref(rs), len(ring), self.rng_seed, int(tick) & 0xFFFFFFFF, ex_pos, ex_cnt,
                                                 self.eta_crit, self.bp_critic[0], self.bp_critic[1], self.eta_act, self.bp_actor[0],
                                                 self.bp_actor[1], C.c_void_p(publish.data_ptr()) if publish is not None else None, st))
            self.bp_critic = [self.bp_critic[0] * 0.9, self.bp_critic[1] * 0.999]
            self.bp_actor = [self.bp_actor[0] * 0.9, self.bp_actor[1] * 0.999]
            self.updates += 1
            return
        if self.sync.native is not None and self.sync.world > 1 and self.noise_type != "pn" and not self.wide:
            # replicas with a native communicator: the split form and both all-reduces in ONE call, everything in this stream
            d.flags = 0                             # (no deferred E products: nothing runs beside an in-stream exchange)
            _capi.check(self.L.shems_ddpg_update_dp(C.byref(d), C.byref(rs), len(ring), self.rng_seed, int(tick) & 0xFFFFFFFF, ex_pos, ex_cnt,
                                                    self.eta_crit, self.bp_critic[0], self.bp_critic[1], self.eta_act, self.bp_actor[0],
                                                    self.bp_actor[1], C.c_void_p(publish.data_ptr()) if publish is not None else None,
                                                    self.sync.native, st))
            self.bp_critic = [self.bp_critic[0] * 0.9, self.bp_critic[1] * 0.999]
            self.bp_actor = [self.bp_actor[0] * 0.9, self.bp_actor[1] * 0.999]
            self.updates += 1
            return
        self._critic_grad_ex(d, rs, len(ring), tick, ex_pos, ex_cnt, st)
        if self.noise_type == "pn":                # DDPG.jl:126-128 (the actor is still the pre-update one here)
            self.adapt_param_noise_(ring, tick)
        if d.flags & 1:
            # the all-reduce is enqueued on the collective's own stream behind the gradient kernel; the actor's E products -- which
            # need nothing the critic update produces -- run on the update stream meanwhile; wait() makes the update stream wait for
            # the collective (no host synchronisation)
            work = self.sync.sum_async_(self.grad_critic)
            _capi.check(self.L.shems_ddpg_actor_prepare(C.byref(d), st))
            if work is not None:
                work.wait()
        else:
            self._allreduce(self.grad_critic)
        gs = self.sync.grad_scale
        self._critic_apply(d, gs, st)
        self.bp_critic = [self.bp_critic[0] * 0.9, self.bp_critic[1] * 0.999]
        self._actor_grad(d, st)
        self._allreduce(self.grad_actor)
        self._actor_apply_pub(d, gs, publish, st)
        self.bp_actor = [self.bp_actor[0] * 0.9, self.bp_actor[1] * 0.999]
        self.updates += 1

    def sample_indices(self, tick, ring_len, batch=None):
        batch = self.batch if batch is None else int(batch)
        out = np.empty(batch, np.int64)
        _capi.check(self.L.shems_ddpg_sample_indices(self.rng_seed, int(tick) & 0xFFFFFFFF, batch, int(ring_len),
                                                     out.ctypes.data_as(C.c_void_p)))
        return out

    def min_max_buffer(self, ring, count=None, seed=None):
        """s_min, s_max = min_max_buffer(MIN_EXP_SIZE) (MPS:50-53, main script :30): extrema of s over a
        bootstrap sample (with replacement) of `count` ring entries; with several replicas the 9-float
        extrema are all-reduced (min / max)."""
        self._same_device(None, ring)
        rs = ring.struct()
        count = len(ring) if count is None else int(count)
        _capi.check(self.L.shems_minmax_dev(C.byref(rs), len(ring), count, self.rng_seed if seed is None else int(seed),
                                            C.c_void_p(self.s_min.data_ptr()), C.c_void_p(self.s_max.data_ptr()), self._stream()))
        self.sync.minmax_(self.s_min, self.s_max)
        return self.s_min, self.s_max

    def act_step_blocks(self, n):
        out = C.c_int64(0)
        _capi.check(self.L.shems_act_step_grid(int(n), C.byref(out)))
        return out.value


    # ------------------------------------------------------------- training loop
    def populate_memory(self, env, ring, seed=None):
        """populate_memory (MPS:9-29): fill the ring to capacity with uniform random actions.  The
        reference runs ceil(MEM/72) sequential 72-step episodes; here that many envs of the batch run
        them in ONE launch (shems_rollout_dev) and push in the reference's episode-major order."""
        self._same_device(env, ring)
        seed = self.rng_seed if seed is None else int(seed)
        nsteps = env.maxsteps
        n_ep = -(-ring.capacity // nsteps)                    # episodes until length(memory) >= MIN_EXP_SIZE
        while len(ring) < ring.capacity:
            env.reset_(seed, episode=0x7FFF0000 + ring.pushed // max(1, nsteps))
            env.rollout("random", nsteps, seed=seed + ring.pushed, ring=ring, ring_envs=min(env.n, n_ep))
        return ring

    def episode_(self, env, ring=None, train=True, num_steps=None, rng_ep=0, episode=0, updates_per_step=1,
                 window_count=None, current_episode=None):
        """episode!(env; train, rng_ep) (DDPG.jl:186-242) for every env of the batch at once.  Returns the
        per-env episode returns (float64 device tensor).  train=True: exploration noise, replay insert
        of a rotating window of envs and `updates_per_step` x replay() per vector step."""
        t = self.torch
        num_steps = env.maxsteps if num_steps is None else int(num_steps)
        if train and self.noise_type == "en":                                   # `global current_episode = i` (DDPG.jl:250)
            self.eps = eps_schedule(episode if current_episode is None else current_episode, ring.capacity if ring is not None
                                    else MEM_SIZE, EP_LENGTH_TRAIN)
        if rng_ep == -1:
            env.reset_(-1)
        else:
            env.reset_(rng_ep, episode=episode)
        returns = t.zeros(env.n, dtype=t.float64, device=self.device)
        self.last_noise = t.zeros(env.n, dtype=t.float32, device=self.device) if train else None     # noise_eps of episode! (DDPG.jl:224)
        if train and ring is None:
            raise ValueError("training episodes need a replay ring")
        if window_count is None and ring is not None:
            window_count = min(env.n, max(1, ring.capacity // num_steps))       # SURVEY.md 8(d) replay-capacity note
        for step in range(num_steps):
            tick = (int(episode) * 4096 + step) & 0xFFFFFFFF
            win = None
            if train:
                win = RingWindow(ring.pos, window_count, (self.tick * window_count) % env.n)
            self.act_step(env, train=train, tick=tick, returns_acc=returns, ring=ring if train else None, window=win,
                          noise_acc=self.last_noise)
            if train and self.noise_type == "pn":
                self.last_noise += float(self.pn_sigma)                           # act() returns pn.sigma_current as the "noise" (DDPG.jl:155)
            if train:
                ring.pushed += window_count
                for _ in range(updates_per_step):
                    self.replay(ring)
                self.tick += 1                                                  # rotates the replay window: training steps only
        return returns

    def run_episodes(self, env_train, env_eval, ring, num_ep, test_every=100, test_runs=100, seed=None,
                     updates_per_step=1, on_eval=None, on_best=None):
        """run_episodes (DDPG.jl:244-298): train episodes, an evaluation sweep every `test_every` episodes
        (when i % test_every == 1) and a snapshot of the best-scoring actor.  Returns
        (total_reward [num_ep], score_mean [ceil(num_ep/test_every)], best_run, best_actor)."""
        seed = self.seed if seed is None else int(seed)
        total_reward = np.zeros(num_ep, np.float32)
        self.noise_mean = np.zeros(num_ep, np.float32)                              # noise_mean[i] (DDPG.jl:255), mean over the batch's envs
        score_mean = np.zeros(-(-num_ep // test_every), np.float64)
        best_score, best_run, best_actor = -100000.0, 0, None
        for i in range(1, num_ep + 1):
            ret = self.episode_(env_train, ring, train=True, rng_ep=seed, episode=i, updates_per_step=updates_per_step)
            total_reward[i - 1] = self.sync.mean_scalar(ret.mean().item(), env_train.n)
            self.noise_mean[i - 1] = self.sync.mean_scalar(self.last_noise.mean().item(), env_train.n)
            if i % test_every == 1:
                idx = -(-i // test_every)
                # DDPG.jl:273-277: every evaluation sweep runs the SAME test_runs seeds "123" * test_ep, so best-score
                # snapshots compare like with like -> fixed (seed_ini, episode 0) reset key, env j = test_ep j + 1.  The eval
                # env has nrow - maxsteps = 1 => every test episode starts at idx 1 and differs in Soc_b only.
                score = self.episode_(env_eval, None, train=False, num_steps=EP_LENGTH_TRAIN, rng_ep=SEED_INI, episode=0)
                score_mean[idx - 1] = self.sync.mean_scalar(score.mean().item(), env_eval.n)
                if score_mean[idx - 1] > best_score:
                    best_score, best_run = score_mean[idx - 1], i
                    best_actor = self.export_actor()
                    if on_best:                                             # saveBSON(...; idx=i, path="temp"), DDPG.jl:282-286
                        on_best(i, best_actor, total_reward, score_mean)
                if on_eval:
                    on_eval(i, total_reward[i - 1], score_mean[idx - 1])
        return total_reward, score_mean, best_run, best_actor


class TrainWorkload:
    """bench.py's "train" step: the body of the reference's episode! loop for all envs of a rank at once --
    act + noise + scale_action + step! + remember (one fused launch) and `updates` x replay() (5 launches
    each; 7 and 2 gradient all-reduces when several GPUs train one model)."""

    name = "train"
    dtype = "f32"
    EP_LEN = EP_LENGTH_TRAIN

    def __init__(self, S, torch, n, seed, updates=1, dist=None, overlap=False, mixed=False, scaled_replay=False, hidden=(L1, L2), loop=None):
        """overlap: False = the reference's order; True / "pipelined" = replay(t) under the fused launch of step t, sampling the ring as
        it stood before step t's inserts; "exact" = the window's envs stepped first, replay(t) under the rest of the batch: the ordered
        loop's bytes.  loop: "native" = the steps are enqueued by shems_train_steps (one foreign call for k steps),
        "host" = one foreign call per launch from this class; None = native wherever the learner is a single replica on the tuned
        kernels (data parallel replicas exchange gradients through torch.distributed between launches, wide networks take their own path)."""
        self.S, self.torch, self.n, self.updates = S, torch, int(n), int(updates)
        self.overlap_mode = {False: LOOP_ORDERED, None: LOOP_ORDERED, True: LOOP_PIPELINED, "pipelined": LOOP_PIPELINED,
                             "exact": LOOP_PIPELINED_EXACT}[overlap]
        overlap = self.overlap_mode != LOOP_ORDERED
        # SURVEY.md 8(d), replay-capacity note: the default ring holds MEM_SIZE = 24 000 transitions and every vector step inserts a rotating
        # window of 333 envs; the optional "scaled" mode holds one 72-step episode of EVERY env (capacity 72 N) and inserts all N each step
        self.scaled_replay = bool(scaled_replay)
        self.mem_size = self.EP_LEN * self.n if self.scaled_replay else MEM_SIZE
        self.overlap = bool(overlap)
        if mixed:                                            # BASELINE config 5: 10 profiles x discomfort-weight sweep
            tabs, cfgs, co = S.mixed_profile_setup(self.n)
            self.env = S.ShemsBatch(self.n, self.EP_LEN, tabs, cfgs, co, device=torch.cuda.current_device()).use_torch_stream()
        else:
            self.tab = S.tables.synthetic_table("train", 98)
            self.env = S.ShemsBatch(self.n, self.EP_LEN, [self.tab], [S.make_config(98, 0, self.tab.shape[0])],
                                    device=torch.cuda.current_device()).use_torch_stream()
        self.env_seed = int(seed)
        self.hidden = (int(hidden[0]), int(hidden[1]))       # other than (250, 500): another point of the reference's grids (bench.py --hidden)
        self.agent = Agent(seed=1231, rng_seed=self.env_seed, hidden=self.hidden)   # same initial weights on every rank (config: seed 1231)
        if dist is not None:
            from .parallel import direct_comm, native_comm
            import os
            import sys
            log = lambda m: print(m, file=sys.stderr, flush=True)
            # How the two gradient all-reduces of an update travel (SHEMS_DP):
            #   torch  (default)  torch.distributed all_reduce on the gradient buffers between the launches ("nccl" = RCCL on ROCm) -- the one
            #                     path whose collective library every multi-GPU PyTorch job on this image exercises.  It is the default because
            #                     NO path of this code has exchanged a byte between two GPUs yet: until a SCALE record exists the
            #                     default is the form with the fewest unknowns, not the fastest projected one.
            #   native            RCCL called in the update's own stream from native code (csrc/shems_dp.hip; a second communicator, made by
            #                     vote, self-tested, falling back to torch on any rank's failure): 63.9 us against 80.6-88.1 us per step with a
            #                     one-rank communicator on one GPU.
            #   direct            no collective: peer-mapped inboxes inside the ADAM sweeps (k_adam_xchg).
            native, direct = None, False
            how = os.environ.get("SHEMS_DP", "torch")
            if how not in ("torch", "native", "direct"):
                raise ValueError("SHEMS_DP must be torch, native or direct")
            if not self.agent.wide:
                if how == "direct":
                    native = direct_comm(dist, log=log)
                    direct = native is not None
                if how == "native" or (how == "direct" and native is None):
                    native = native_comm(dist, log=log)
            self.dp_requested = how
            self.agent.enable_data_parallel(dist, native=native, direct=direct)
            if os.environ.get("SHEMS_DP_OVERLAP") in ("0", "1"):   # A/B knob: "1" = the critic's all-reduce asynchronous, under the actor's E products
                self.agent.dp_overlap = os.environ["SHEMS_DP_OVERLAP"] == "1"
        self.ring = ReplayRing(self.mem_size)
        self.agent.populate_memory(self.env, self.ring, seed=self.env_seed)          # MAIN:28
        self.agent.min_max_buffer(self.ring, self.mem_size, seed=self.env_seed)      # MAIN:30
        self.win = min(self.n, self.mem_size // self.EP_LEN)
        self.rew32 = torch.empty(self.n, dtype=torch.float32, device="cuda")
        self.t = 0
        self.episode = 1
        self.env.reset_(self.env_seed, episode=self.episode)
        native_ok = (dist is None or self.agent.sync.world == 1 or self.agent.sync.native is not None) and not self.agent.wide and self.agent.noise_type != "pn" and self.agent.batch <= self.agent.MAX_PASS_BATCH
        if loop is None:
            loop = "native" if native_ok else "host"
        if loop == "native" and not native_ok:
            raise NotImplementedError("the native loop drives one replica on the tuned kernels (no gradient exchange, no wide network, no parameter noise)")
        if self.overlap_mode == LOOP_PIPELINED_EXACT and loop != "native":
            raise NotImplementedError("the order-exact pipelined mode exists in the native loop only (shems_train_steps)")
        if self.overlap and loop == "native" and self.agent.sync.world > 1:
            raise NotImplementedError("data-parallel replicas run in program order (the gradient exchange sits in the update's own stream)")
        if self.overlap and self.agent.wide:
            raise NotImplementedError("pipelined modes run the tuned (250, 500) kernels: a wide network's actor copies are not in their layout")
        self.loop = loop
        self._native = None
        if self.overlap:
            # Pipelined mode: replay(t) runs on a second stream while the fused act/step kernel of step t runs on the main
            # one.  act(t) still uses the actor produced by replay(t-1), exactly as in the sequential loop; the one deviation
            # is that replay(t) cannot sample the transitions step t is inserting (333 of 24 000 slots are excluded).
            self.upd_stream = torch.cuda.Stream()
            self.actor_pub = [self.agent.actor.clone(), self.agent.actor.clone()]
            self.ev_act = [torch.cuda.Event(), torch.cuda.Event()]
            self.ev_upd = [torch.cuda.Event(), torch.cuda.Event()]
            for e in self.ev_act + self.ev_upd:
                e.record()

    def _act(self, tick, actor=None):
        w = RingWindow(self.ring.pos, self.win, (tick * self.win) % self.n)
        if actor is None:
            self.agent.act_step(self.env, train=True, tick=tick, rewards_f32=self.rew32, ring=self.ring, window=w)
        else:
            v = self.env.view()
            p = self.agent._act_params(True, tick, actor=actor)
            rs = self.ring.struct()
            _capi.check(self.agent.L.shems_act_step_dev(C.byref(v), C.byref(p), None, None, C.c_void_p(self.rew32.data_ptr()), None,
                                                        None, C.byref(rs), C.byref(w), self.agent._stream()))
        self.ring.pushed += self.win

    def _native_loop(self):
        """The shems_train_loop record of this workload, its in/out fields loaded from the Python-side state."""
        ag = self.agent
        if self._native is None:
            L = TrainLoop()
            L.view = self.env.view()
            L.act = ag._act_params(True, 0)
            L.ddpg = ag._ddpg_args()
            L.ddpg.flags = 0                     # the native data-parallel step runs both exchanges in program order: K2 keeps the actor's E
                                                 # products (the DEFER_ACTOR_E form belongs to Agent.replay's asynchronous torch path only)
            L.ring = self.ring.struct()
            L.rewards_f32 = self.rew32.data_ptr()
            if self.overlap:
                L.actor_pub[0], L.actor_pub[1] = self.actor_pub[0].data_ptr(), self.actor_pub[1].data_ptr()
            L.window, L.env_seed, L.sample_seed = self.win, self.env_seed, ag.rng_seed
            L.ep_len, L.updates_per_step, L.mode = self.EP_LEN, self.updates, self.overlap_mode
            L.eta_crit, L.eta_act = ag.eta_crit, ag.eta_act
            L.dp = ag.sync.native if ag.sync.world > 1 else None
            self._native = L
        L = self._native
        L.ring_pushed, L.t, L.updates, L.episode = self.ring.pushed, self.t, ag.updates, self.episode
        L.bp_crit[0], L.bp_crit[1], L.bp_act[0], L.bp_act[1] = ag.bp_critic[0], ag.bp_critic[1], ag.bp_actor[0], ag.bp_actor[1]
        return L

    def steps(self, k):
        """k vector steps.  Native loop: ONE foreign call enqueues all of them."""
        if self.loop != "native":
            for _ in range(int(k)):
                self.step()
            return
        ag = self.agent
        L = self._native_loop()
        st2 = C.c_void_p(self.upd_stream.cuda_stream) if self.overlap else None
        try:
            _capi.check(ag.L.shems_train_steps(C.byref(L), int(k), ag._stream(), st2))
        finally:
            self.ring.pushed, self.t, self.episode = L.ring_pushed, L.t, L.episode
            ag.updates = L.updates
            ag.bp_critic, ag.bp_actor = [L.bp_crit[0], L.bp_crit[1]], [L.bp_act[0], L.bp_act[1]]

    def step(self):
        if self.loop == "native":
            return self.steps(1)
        torch = self.torch
        if self.t and self.t % self.EP_LEN == 0:
            self.episode += 1
            v = self.env.view()
            _capi.check(_capi.lib().shems_reset_seeded_dev(C.byref(v), self.env_seed, self.episode, self.env._stream()))
        if not self.overlap:
            self._act(self.t)
            for _ in range(self.updates):
                self.agent.replay(self.ring)
            self.t += 1
            return
        t = self.t
        main = torch.cuda.current_stream()
        main.wait_event(self.ev_upd[(t - 1) & 1])                  # actor_pub[t & 1] was published by replay(t - 1)
        win_pos = self.ring.pos
        self._act(t, actor=self.actor_pub[t & 1])
        self.ev_act[t & 1].record(main)
        self.upd_stream.wait_event(self.ev_act[(t - 1) & 1])       # ring complete through step t-1; actor_pub[(t+1)&1] no longer read
        with torch.cuda.stream(self.upd_stream):
            for u in range(self.updates):
                self.agent.replay(self.ring, exclude=(win_pos, self.win),
                                  publish=self.actor_pub[(t + 1) & 1] if u == self.updates - 1 else None)
            self.ev_upd[t & 1].record(self.upd_stream)
        self.t += 1

    def finish(self):
        if self.overlap:
            self.torch.cuda.current_stream().wait_stream(self.upd_stream)
            self.torch.cuda.synchronize()
        if self._native is not None:
            _capi.check(self.agent.L.shems_train_loop_release(C.byref(self._native)))
        # Every rank-local verdict is collected FIRST and exchanged in ONE collective together with the learner's checksum; only then does
        # anybody raise -- a rank that sees an exchange timeout (normally only some do) must not leave its peers blocked in a collective.
        problems = []
        if getattr(self.agent.sync, "direct", False):
            n = C.c_int64(0)
            self.agent.L.shems_dp_direct_timeouts.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]
            self.agent.L.shems_dp_direct_timeouts.restype = C.c_int
            _capi.check(self.agent.L.shems_dp_direct_timeouts(self.agent.sync.native, C.byref(n), self.agent._stream()))
            if n.value or self.dp_poisoned():
                problems.append(f"{n.value} waits of the direct gradient exchange gave up: a peer never delivered, the replicas have diverged")
        try:
            self.env.check_error()
        except Exception as e:                      # noqa: BLE001
            problems.append(f"env error: {e}")
        if not bool(self.torch.isfinite(self.agent.actor).all()) or not bool(self.torch.isfinite(self.agent.critic).all()):
            problems.append("non-finite network parameters after the timed steps")
        # replicas: every rank's learner must hold the same bytes (they started identical and added the same gradients in the same order);
        # gathered here, where every rank passes, and reported by rank 0 (`replica_crc32_distinct`: 1 = identical)
        self.replica_crcs = None
        sync = self.agent.sync
        if sync.dist is not None:
            got = [None] * sync.dist.get_world_size()
            sync.dist.all_gather_object(got, (self._learner_crc(), problems))
            self.replica_crcs = [g[0] for g in got]
            for r, g in enumerate(got):
                if r != sync.rank and g[1]:
                    problems.append(f"rank {r}: " + "; ".join(g[1]))
        if problems:
            raise RuntimeError("; ".join(problems))

    def dp_poisoned(self):
        """True once a wait of the direct gradient exchange has given up on this rank (sticky; no synchronisation)."""
        sync = self.agent.sync
        if not getattr(sync, "direct", False) or sync.native is None:
            return False
        out = C.c_int32(0)
        self.agent.L.shems_dp_direct_poisoned.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        self.agent.L.shems_dp_direct_poisoned.restype = C.c_int
        _capi.check(self.agent.L.shems_dp_direct_poisoned(sync.native, C.byref(out)))
        return bool(out.value)

    def close(self):
        """End of the run: give the native communicator / the direct exchange's mappings back (every rank calls it; idempotent)."""
        sync = self.agent.sync
        h = getattr(sync, "native", None)
        if h is not None:
            self.torch.cuda.synchronize()
            self.agent.L.shems_dp_destroy.argtypes = [C.c_void_p]
            self.agent.L.shems_dp_destroy.restype = C.c_int
            sync.native, sync.direct = None, False
            if self._native is not None:
                self._native.dp = None
            self.agent.L.shems_dp_destroy(h)

    def kernel_pass(self, reps):
        """HIP-event timing of the dominant kernel (the fused actor/step launch) IN THE LOOP IT RUNS IN.

        The train loop alternates k_act with the launches of replay(), and the clock the chip holds in that mix is not the clock of
        k_act launched back to back: one rocprofv3 trace of `bench.py` shows 133.3 us per k_act inside the loop and 139.6 us for the
        same kernel in a back-to-back pass.  So the pass times groups of 8 whole vector steps -- the timed region's own step(), episode
        boundaries included -- and, separately, groups of 8 x replay() alone; the difference per step is the act launch together with the
        gap in front of it.  A pair of events around every single launch would add the command processor's hand-off (3-5 us) to each
        reading (timing.py).  (Round 3: the pass used to call its own [act, replay] pair with small tick numbers; groups of that pair read
        171 us where groups of the real step() read 167.5 us = the main loop's 167.2 us, tools/pass_probe.py.)

        Data parallel (world > 1): EVERY rank runs this pass, with the gradient exchange in place (the same number of collectives on
        every rank: the group counts below do not depend on the rank), and it additionally times (i) replay() with its two all-reduces
        (`update_us_dp`), (ii) the same split launches without the exchange (`update_us_split_local`) and (iii) each all-reduce alone on
        a scratch tensor of the gradient's size, back to back (`allreduce_critic_us`, `allreduce_actor_us`); the figures are the MAX over
        ranks.  The learner state is put back afterwards on every rank, so the replicas stay identical."""
        torch = self.torch
        from .timing import time_launches
        reps = min(reps, 200)
        snap = self.agent.snapshot()
        t_saved, ep_saved, pushed_saved = self.t, self.episode, self.ring.pushed
        sync = self.agent.sync
        world = sync.world
        self.agent.fused = world == 1                   # the launch structure the benchmarked world size runs

        def vector_step(i):                              # the timed region's own step (episode boundaries and their device reset included)
            if not self.overlap:
                return self.step()
            self._act(i)                                 # pipelined mode keeps its second stream out of this pass: act + replay in order
            for _ in range(self.updates):
                self.agent.replay(self.ring)

        def updates_only(i):
            for _ in range(self.updates):
                self.agent.replay(self.ring)

        self.dp = None
        try:
            # settle into the loop's regime before the first timed group.  The pass starts after finish()'s synchronisation and host-side
            # checks, i.e. from an idle device: tools/pass_dist.py shows groups of 8 steps reading 178, 174, 173, 171, 172, 170, 169, 168 us
            # ... before they are back at the loop's 166 us after ~100 steps (clock / power state), which put the pass's average 3 % above its
            # median and above rocprofv3's per-kernel average.  600 steps (0.1 s at 65 536 envs; the same count on every rank).
            for i in range(600):
                vector_step(i)
            step_avg, step_med, n = time_launches(torch, vector_step, reps)
            if self.updates:
                upd_avg, upd_med, _ = time_launches(torch, updates_only, reps)
            else:
                upd_avg = upd_med = 0.0
            # cross-check of the difference above: the fused launch alone, back to back (no update between two of them: the chip holds a
            # lower clock under the uninterrupted MFMA load, so this reads a few per cent ABOVE the in-loop figure -- rocprofv3 showed 139.6
            # against 133.3 us inside one trace in round 2 -- but it is a direct timing of the kernel with nothing subtracted)
            alone_avg = time_launches(torch, lambda i: self._act(i), min(reps, 96))[0] if not self.agent.wide else None
            if world > 1:
                self.agent.sync = GradSync(None)         # same 7 launches, no exchange
                self.agent.fused = False
                loc_avg = time_launches(torch, updates_only, reps)[0] if self.updates else 0.0
                self.agent.sync = sync
                gc, ga = torch.zeros_like(self.agent.grad_critic), torch.zeros_like(self.agent.grad_actor)
                if getattr(sync, "direct", False):  # no all-reduce exists on its own: the exchange is inside the ADAM sweeps
                    ar = lambda g: None
                elif sync.native is not None:       # the exchange as the update issues it: RCCL in this stream, from native code
                    Lc = self.agent.L
                    Lc.shems_dp_allreduce_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
                    Lc.shems_dp_allreduce_sum.restype = C.c_int
                    ar = lambda g: _capi.check(Lc.shems_dp_allreduce_sum(sync.native, C.c_void_p(g.data_ptr()), g.numel(), self.agent._stream()))
                else:
                    ar = sync.sum_
                for g in (gc, ga):
                    ar(g)
                ar_c = time_launches(torch, lambda i: ar(gc), 64)[0]
                ar_a = time_launches(torch, lambda i: ar(ga), 64)[0]
                fig = torch.tensor([step_avg, step_med, upd_avg, upd_med, loc_avg, ar_c, ar_a], dtype=torch.float64, device=self.agent.device)
                sync.dist.all_reduce(fig, op=sync.dist.ReduceOp.MAX)
                step_avg, step_med, upd_avg, upd_med, loc_avg, ar_c, ar_a = (float(x) for x in fig.tolist())
                per = max(1, self.updates)
                self.dp = {"update_us_dp": upd_avg / per, "update_us_split_local": loc_avg / per, "allreduce_critic_us": ar_c,
                           "allreduce_actor_us": ar_a, "allreduce_bytes": [4 * N_CRITIC, 4 * N_ACTOR],
                           "exchange_us_in_update": (upd_avg - loc_avg) / per,
                           "k_act_us_at_shard": step_avg - upd_avg, "figures": "max over ranks; HIP events over groups of 8"}
        finally:
            self.agent.sync = sync
            self.agent.restore(snap)
            self.t, self.episode, self.ring.pushed = t_saved, ep_saved, pushed_saved
        self.update_us = upd_avg / self.updates if self.updates else None
        self.step_us_in_pass = step_avg
        h1, h2 = self.hidden
        flops = 2 * (9 * h1 + h1 * h2 + h2 * 2) * self.n                     # 256 500 FLOP / env-step at (250, 500)
        kname = "shems::k_wgemm x3 + k_act_tail (wide network, csrc/shems_wide.hip)" if self.agent.wide else act_kernel_name(self.n)
        return dict(kernel=kname, avg_us=step_avg - upd_avg, median_us=step_med - upd_med, launches=n,
                    bound="mfma", algorithmic=flops, unit="TFLOP/s", peak=157.3,
                    algorithmic_bytes=None if self.agent.wide else act_algorithmic_bytes(self.n, self.win),
                    back_to_back_us=alone_avg,
                    avg_us_is="vector step minus replay(): the fused launch together with the gap in front of it, not a timing of the kernel alone (rocprofv3's per-kernel average in profiles/ is the cross-check)",
                    method="HIP events over groups of 8 vector steps minus groups of 8 replay() alone (the kernel inside its loop, launch gap included)"
                           + ("; data parallel: the gradient exchange is inside both, figures are the max over ranks" if world > 1 else ""))

    def _learner_crc(self):
        import zlib
        return zlib.crc32(self.agent.actor.detach().cpu().numpy().tobytes()) ^ zlib.crc32(self.agent.critic_t.detach().cpu().numpy().tobytes())

    def extra(self):
        crc = self._learner_crc()
        rc = getattr(self, "replica_crcs", None)
        return {"replica_crc32_distinct": None if rc is None else len(set(rc)),
                "updates_per_step": self.updates, "batch_size": BATCH_SIZE, "mem_size": self.mem_size,
                "replay_mode": "scaled (capacity 72 N, every env inserts)" if self.scaled_replay else "window (MEM_SIZE = 24 000, 333 envs insert per step)",
                "overlap": {LOOP_ORDERED: False, LOOP_PIPELINED: "pipelined", LOOP_PIPELINED_EXACT: "exact"}[self.overlap_mode],
                "loop": self.loop,
                "learner_crc32": crc, "dp_overlap": bool(self.agent.dp_overlap and self.agent.sync.world > 1 and self.loop != "native"),   # (the native loop exchanges in program order)
                "dp_exchange": None if self.agent.sync.world == 1 else (
                    "direct exchange through peer-mapped inboxes inside the ADAM sweeps (k_adam_xchg), no collective launch" if getattr(self.agent.sync, "direct", False) else
                    "RCCL all-reduce in the update's own stream, issued from native code (shems_ddpg_update_dp)" if self.agent.sync.native is not None else
                    "torch.distributed all_reduce (its own stream)"),
                "dp_requested": getattr(self, "dp_requested", None),
                "replay_window_envs_per_step": self.win, "update_us": getattr(self, "update_us", None),
                "update_mflop": 307.8 if self.hidden == (L1, L2) else 20 * (10 * self.hidden[0] + self.hidden[0] * self.hidden[1] + 1.5 * self.hidden[1]) * BATCH_SIZE / 1e6,
                "hidden": list(self.hidden), "data_parallel": getattr(self, "dp", None)}


def smoke():
    """One tiny train iteration on cuda:0: populate, normalise, 3 fused steps + updates; finite + oracle-checked."""
    import importlib
    import torch
    pkg = importlib.import_module(__name__.rsplit(".", 1)[0])
    wl = TrainWorkload(pkg, torch, 2048, seed=7, updates=1)
    before = wl.agent.actor.clone()
    for _ in range(3):
        wl.step()
    wl.finish()
    assert not torch.equal(before, wl.agent.actor), "actor did not move"
    assert len(wl.ring) == MEM_SIZE
