"""CPU restatement (NumPy) of the reference's DDPG learner path.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/RL-SHEMS/algorithms/DDPG.jl (nets :30-46, soft_update! :99-103, losses
:114-119, replay :121-145, act :148-176, scale_action :178-184) and
src/memory_plotting_saving.jl (normalize :55-57, getData :31-42, min_max_buffer :50-53).

PARITY STATUS: "parity unpinned".  The arithmetic below lives in un-vendored dependencies of the
reference (Flux 0.12.1 Dense/mse/ADAM, Zygote 0.6.12 gradients, CUDA.jl/CUBLAS reduction order --
Manifest.toml pins) and the reference has no test or fixture for it; the formulas are restated from
those packages' published definitions and cross-checked against PyTorch autograd in tests/.
Random streams (glorot init, exploration noise, minibatch indices) use this repo's Philox
counters, not Julia's MersenneTwister.

Parameter layout (one flat float32 vector per network, Flux.params order, each W the C-order view
[in][out] of Julia's column-major out x in matrix):  W1 b1 W2 b2 W3 b3.
"""
from __future__ import annotations

import numpy as np

L1, L2 = 250, 500
STATE, ACTION = 9, 2
GAMMA = np.float32(0.99)
TAU = np.float32(1e-3)
ETA_ACT, ETA_CRIT = np.float32(1e-4), np.float32(1e-3)
BETA = (0.9, 0.999)
EPS = 1e-8

f32 = np.float32


def sizes(in_dim, out_dim):
    return [in_dim * L1, L1, L1 * L2, L2, L2 * out_dim, out_dim]


def n_params(in_dim, out_dim):
    return sum(sizes(in_dim, out_dim))


def split(p, in_dim, out_dim):
    o = np.cumsum([0] + sizes(in_dim, out_dim))
    return (p[o[0]:o[1]].reshape(in_dim, L1), p[o[1]:o[2]], p[o[2]:o[3]].reshape(L1, L2), p[o[3]:o[4]],
            p[o[4]:o[5]].reshape(L2, out_dim), p[o[5]:o[6]])


def normalize(s, s_min, s_max):
    """MPS:55-57  (s .- s_min) ./ (s_max .- s_min .+ 1f-8)"""
    s = np.asarray(s, f32)
    return ((s - s_min) / ((s_max - s_min) + f32(1e-8))).astype(f32)


def mlp_forward(p, x, in_dim, out_dim, final_tanh, keep=False, dtype=np.float32):
    """Chain(Dense(in,250,relu), Dense(250,500,relu), Dense(500,out[,tanh]))  DDPG.jl:30-44."""
    W1, b1, W2, b2, W3, b3 = (a.astype(dtype) for a in split(p, in_dim, out_dim))
    x = np.asarray(x, dtype)
    z1 = x @ W1 + b1
    h1 = np.maximum(z1, 0)
    z2 = h1 @ W2 + b2
    h2 = np.maximum(z2, 0)
    z3 = h2 @ W3 + b3
    y = np.tanh(z3) if final_tanh else z3
    if keep:
        return y, (x, z1, h1, z2, h2, z3)
    return y


def actor_forward(p, s_norm, dtype=np.float32):
    return mlp_forward(p, s_norm, STATE, ACTION, True, dtype=dtype)


def critic_forward(p, s_norm, a, dtype=np.float32):
    return mlp_forward(p, np.concatenate([s_norm, a], 1), STATE + ACTION, 1, False, dtype=dtype)[:, 0]


def mlp_backward(p, cache, dy, in_dim, out_dim, final_tanh, y=None, dtype=np.float32):
    """Gradient of sum(dy * y) wrt the flat parameters and wrt the input x (arithmetic in `dtype`; float64 = the arbiter
    the per-block parity tests hold both this restatement and the kernels to)."""
    W1, b1, W2, b2, W3, b3 = (a.astype(dtype) for a in split(p, in_dim, out_dim))
    dy = np.asarray(dy, dtype)
    x, z1, h1, z2, h2, z3 = cache
    d3 = dy * (1 - np.tanh(z3) ** 2) if final_tanh else dy
    gW3 = h2.T @ d3
    gb3 = d3.sum(0)
    d2 = (d3 @ W3.T) * (z2 > 0)
    gW2 = h1.T @ d2
    gb2 = d2.sum(0)
    d1 = (d2 @ W2.T) * (z1 > 0)
    gW1 = x.T @ d1
    gb1 = d1.sum(0)
    dx = d1 @ W1.T
    g = np.concatenate([gW1.ravel(), gb1, gW2.ravel(), gb2, gW3.ravel(), gb3]).astype(dtype)
    return g, dx.astype(dtype)


def blocks(in_dim, out_dim):
    """[(name, lo, hi)] of the six Flux.params blocks in the flat layout."""
    o = np.cumsum([0] + sizes(in_dim, out_dim))
    return [(n, int(o[i]), int(o[i + 1])) for i, n in enumerate(("W1", "b1", "W2", "b2", "W3", "b3"))]


class Adam:
    """Flux 0.12.1 ADAM(eta, (0.9, 0.999)), eps = 1e-8: scalars in Float64, arrays Float32.
        mt = b1*mt + (1-b1)*g ; vt = b2*vt + (1-b2)*g^2   (g^2 = literal_pow -> g*g in Float32, then promoted)
        delta = mt / (1 - bp1) / (sqrt(vt / (1 - bp2)) + eps) * eta ; bp .*= beta ; p .-= delta"""

    def __init__(self, n, eta):
        self.m = np.zeros(n, f32)
        self.v = np.zeros(n, f32)
        self.bp = [BETA[0], BETA[1]]
        self.eta = float(f32(eta))

    def step(self, p, g):
        g64 = g.astype(np.float64)
        self.m = (BETA[0] * self.m.astype(np.float64) + (1 - BETA[0]) * g64).astype(f32)
        g2 = (g.astype(f32) * g.astype(f32)).astype(np.float64)          # Float32 square (Flux optimisers.jl ADAM: `Δ^2`)
        self.v = (BETA[1] * self.v.astype(np.float64) + (1 - BETA[1]) * g2).astype(f32)
        delta = (self.m.astype(np.float64) / (1 - self.bp[0]) /
                 (np.sqrt(self.v.astype(np.float64) / (1 - self.bp[1])) + EPS) * self.eta).astype(f32)
        self.bp = [self.bp[0] * BETA[0], self.bp[1] * BETA[1]]
        return (p - delta).astype(f32)


def soft_update(target, model, tau=TAU):
    """DDPG.jl:99-103  p_t .= (1f0 - tau) * p_t .+ tau * p_m"""
    return ((f32(1) - tau) * target + tau * model).astype(f32)


# ------------------------------------------------------------------ RNG --
M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
STREAM_NOISE, STREAM_SAMPLE, STREAM_INIT = 0x4E4F4953, 0x53414D50, 0x494E4954


def philox(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & 0xFFFFFFFF for c in np.broadcast_arrays(c0, c1, c2, c3))
    mask = np.uint64(0xFFFFFFFF)
    k0, k1 = np.uint64(k0 & 0xFFFFFFFF), np.uint64(k1 & 0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(M0) * c0, np.uint64(M1) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & mask, p1 & mask, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & mask, p0 & mask
        k0, k1 = (k0 + np.uint64(W0)) & mask, (k1 + np.uint64(W1)) & mask
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def gauss_noise(seed, tick, n):
    """The device's exploration noise: Box-Muller on one Philox block per env (float32)."""
    i = np.arange(n, dtype=np.uint64)
    x, y, _, _ = philox(i & 0xFFFFFFFF, i >> np.uint64(32), tick, STREAM_NOISE, seed & 0xFFFFFFFF, seed >> 32)
    u1 = ((x >> np.uint32(8)).astype(f32) + f32(1)) * f32(1.0 / 16777216.0)
    u2 = (y >> np.uint32(8)).astype(f32) * f32(1.0 / 16777216.0)
    r = np.sqrt(f32(-2) * np.log(u1)).astype(f32)
    ang = (f32(6.28318530717958647692) * u2).astype(f32)
    return np.stack([r * np.cos(ang), r * np.sin(ang)], 1).astype(f32)


def sample_indices(seed, tick, batch, length):
    """Minibatch indices WITH replacement (StatsBase.sample(rng, memory, n), MPS:33): one Philox
    block yields 4 indices; index = x mod length."""
    q = np.arange((batch + 3) // 4, dtype=np.uint64)
    xs = philox(q, 0, tick, STREAM_SAMPLE, seed & 0xFFFFFFFF, seed >> 32)
    flat = np.stack(xs, 1).reshape(-1)[:batch]
    return (flat % np.uint32(length)).astype(np.int64)


def init_params(seed, in_dim, out_dim, which):
    """Flux.glorot_uniform for the two hidden layers ((rand - 0.5) * sqrt(24 / (fan_in + fan_out))),
    U(-3e-3, 3e-3) for the last (w_init, DDPG.jl:21-22), zero biases -- with Philox uniforms.
    `which` (0 actor, 1 critic) separates the streams."""
    out = []
    ctr = 0
    for li, (fan_in, fan_out) in enumerate([(in_dim, L1), (L1, L2), (L2, out_dim)]):
        n = fan_in * fan_out
        q = np.arange((n + 3) // 4, dtype=np.uint64)
        xs = philox(q, li, which, STREAM_INIT, seed & 0xFFFFFFFF, seed >> 32)
        u = (np.stack(xs, 1).reshape(-1)[:n] >> np.uint32(8)).astype(f32) * f32(1.0 / 16777216.0)
        if li < 2:
            w = (u - f32(0.5)) * f32(np.sqrt(f32(24.0) / f32(fan_in + fan_out)))
        else:
            w = f32(6e-3) * u - f32(3e-3)
        out += [w.astype(f32), np.zeros(fan_out, f32)]
        ctr += 1
    return np.concatenate(out)


# ------------------------------------------------------------ act / replay --
def act(actor, s, s_min, s_max, train, seed=0, tick=0, mu=0.0, sigma=0.1, dtype=np.float32, noise="gn", ou_state=None,
        theta=0.15, dt=1e-2, eps=0.5):
    """act(): the exploration branches of DDPG.jl:148-176 (gn / ou / en).  `ou_state` [n][2] is updated in place."""
    a = actor_forward(actor, normalize(s, s_min, s_max), dtype=dtype).astype(f32)
    n = len(a)
    if train and noise == "en":
        i = np.arange(n, dtype=np.uint64)
        x, y, z, _ = philox(i & 0xFFFFFFFF, i >> np.uint64(32), tick, STREAM_NOISE, seed & 0xFFFFFFFF, seed >> 32)
        explore = ~(((z >> np.uint32(8)).astype(f32) * f32(1.0 / 16777216.0)) > f32(eps))
        uni = np.stack([(x.astype(np.float64) / 4294967296.0 * 2 - 1), (y.astype(np.float64) / 4294967296.0 * 2 - 1)], 1).astype(f32)
        return np.where(explore[:, None], uni, a).astype(f32)
    if train:
        zn = gauss_noise(seed, tick, n)
        if noise == "ou":
            sdt = f32(sigma) * np.sqrt(f32(dt))
            ou_state += (f32(theta) * (f32(mu) - ou_state) * f32(dt) + sdt * zn).astype(f32)
            a = a + ou_state
        else:
            a = a + (f32(mu) + f32(sigma) * zn)
    return np.clip(a, f32(-1), f32(1)).astype(f32)


STREAM_PERTURB = 0x50455254


def perturb_shift(seed, tick, mu, sigma):
    """sample_noise(pn::ParamNoise, rng) (DDPG.jl:63-67): ONE scalar Normal(mu, sigma_current) draw, Float32."""
    x = philox(int(tick) & 0xFFFFFFFF, 0, 0, STREAM_PERTURB, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u1 = (float(x[0] >> np.uint32(8)) + 0.5) / 16777216.0
    u2 = (float(x[1] >> np.uint32(8)) + 0.5) / 16777216.0
    return float(f32(mu + sigma * np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)))


def add_perturb(actor, shift):
    """add_perturb! (DDPG.jl:89-96): `p_t .= p_t .+ sample_noise(pn, rng)` re-seeds before each draw, so all six parameter
    arrays of the copy receive the same scalar."""
    return (actor.astype(f32) + f32(shift)).astype(f32)


def act_param_noise(actor, s, s_min, s_max, shift, dtype=np.float32):
    """act(...; train=true) with noise_type == "pn" (DDPG.jl:152-156): clamp(actor_perturb(s_norm), -1, 1)."""
    a = actor_forward(add_perturb(actor, shift), normalize(s, s_min, s_max), dtype=dtype).astype(f32)
    return np.clip(a, f32(-1), f32(1)).astype(f32)


def adapt_param_noise(actor, s, s_min, s_max, shift, sigma_current, sigma_target=0.1, adoption=1.01, dtype=np.float32):
    """adapt_param_noise! (DDPG.jl:74-87) -> (distance, new sigma_current)."""
    sn = normalize(s, s_min, s_max)
    a = actor_forward(actor, sn, dtype=dtype).astype(np.float64)
    ap = actor_forward(add_perturb(actor, shift), sn, dtype=dtype).astype(np.float64)
    distance = float(np.sqrt(np.mean((a - ap) ** 2)))
    return distance, (sigma_current / adoption if distance > sigma_target else sigma_current * adoption)


class Learner:
    """actor/critic/targets + two ADAMs; replay() = one DDPG update (DDPG.jl:121-145)."""

    def __init__(self, actor, critic, s_min, s_max):
        self.actor, self.critic = actor.copy(), critic.copy()
        self.actor_t, self.critic_t = actor.copy(), critic.copy()      # deepcopy, DDPG.jl:38, 46
        self.opt_a = Adam(len(actor), ETA_ACT)
        self.opt_c = Adam(len(critic), ETA_CRIT)
        self.s_min, self.s_max = np.asarray(s_min, f32), np.asarray(s_max, f32)

    def targets(self, r, s2, done):
        s2n = normalize(s2, self.s_min, self.s_max)
        a2 = actor_forward(self.actor_t, s2n)
        q2 = critic_forward(self.critic_t, s2n, a2)
        return (r + GAMMA * (f32(1) - done.astype(f32)) * q2).astype(f32)          # DDPG.jl:133

    def critic_grad(self, s, a, y, dtype=np.float32):
        sn = normalize(s, self.s_min, self.s_max)
        q, cache = mlp_forward(self.critic, np.concatenate([sn, a], 1), STATE + ACTION, 1, False, keep=True, dtype=dtype)
        B = len(y)
        dq = (dtype(2) * (q[:, 0] - y) / dtype(B)).astype(dtype)[:, None]           # Flux.mse
        g, _ = mlp_backward(self.critic, cache, dq, STATE + ACTION, 1, False, dtype=dtype)
        loss = float(np.mean((q[:, 0] - y) ** 2))
        return g, loss

    def actor_grad(self, s, dtype=np.float32):
        sn = normalize(s, self.s_min, self.s_max)
        a, ca = mlp_forward(self.actor, sn, STATE, ACTION, True, keep=True, dtype=dtype)
        q, cc = mlp_forward(self.critic, np.concatenate([sn, a], 1), STATE + ACTION, 1, False, keep=True, dtype=dtype)
        B = len(s)
        dq = np.full((B, 1), dtype(-1.0 / B), dtype)                                # -mean(q)
        _, dx = mlp_backward(self.critic, cc, dq, STATE + ACTION, 1, False, dtype=dtype)
        g, _ = mlp_backward(self.actor, ca, dx[:, STATE:], STATE, ACTION, True, dtype=dtype)
        return g, float(-np.mean(q))

    def replay(self, s, a, r, s2, done, allreduce=None):
        y = self.targets(r, s2, done)
        gc, lc = self.critic_grad(s, a, y)
        if allreduce:
            gc = allreduce(gc)
        self.critic = self.opt_c.step(self.critic, gc)
        ga, la = self.actor_grad(s)
        if allreduce:
            ga = allreduce(ga)
        self.actor = self.opt_a.step(self.actor, ga)
        self.actor_t = soft_update(self.actor_t, self.actor)
        self.critic_t = soft_update(self.critic_t, self.critic)
        return lc, la
