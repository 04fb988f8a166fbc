"""CPU tests of the DDPG oracle (NumPy restatement of DDPG.jl / MPS learner arithmetic) against
PyTorch autograd + a hand-written Flux-0.12 Adam -- "parity unpinned" by the reference (un-vendored
Flux/Zygote arithmetic, no reference tests); these tests pin the oracle to an independent
implementation of the same published formulas."""
import numpy as np
import pytest

import util as U  # noqa: F401  (sys.path)
import ddpg_oracle as DO
import philox_np


def test_philox_streams_agree_and_noise_is_standard_normal():
    assert [int(v) for v in DO.philox(0, 0, 0, 0, 0, 0)] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert [int(v) for v in philox_np.philox4x32_10(0, 0, 0, 0, 0, 0)] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    z = DO.gauss_noise(1231, 5, 200000)
    assert z.shape == (200000, 2) and abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.01 and np.isfinite(z).all()
    assert not (z == DO.gauss_noise(1231, 6, 200000)).all()
    idx = DO.sample_indices(7, 3, 120, 24000)
    assert idx.shape == (120,) and idx.min() >= 0 and idx.max() < 24000 and len(np.unique(idx)) > 100


def _torch_nets(actor, critic):
    import torch

    def mk(p, i, o):
        W1, b1, W2, b2, W3, b3 = DO.split(p, i, o)
        ts = [torch.tensor(np.array(a), dtype=torch.float64, requires_grad=True) for a in (W1, b1, W2, b2, W3, b3)]
        return ts

    def fwd(ts, x, tanh):
        W1, b1, W2, b2, W3, b3 = ts
        h = torch.relu(x @ W1 + b1)
        h = torch.relu(h @ W2 + b2)
        y = h @ W3 + b3
        return torch.tanh(y) if tanh else y
    return mk(actor, 9, 2), mk(critic, 11, 1), fwd


def test_gradients_match_torch_autograd():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(0)
    actor, critic = DO.init_params(1231, 9, 2, 0), DO.init_params(1231, 11, 1, 1)
    critic[128250:128750] *= 50       # make q sensitive so gradients are not ~1e-6
    actor[128000:129000] *= 50
    B = 120
    s = rng.random((B, 9)).astype(np.float32) * 3
    a = (rng.random((B, 2)).astype(np.float32) * 2 - 1)
    r = rng.normal(size=B).astype(np.float32)
    s2 = rng.random((B, 9)).astype(np.float32) * 3
    done = np.zeros(B, bool)
    s_min, s_max = s.min(0), s.max(0)
    L = DO.Learner(actor, critic, s_min, s_max)
    y = L.targets(r, s2, done)
    gc, lc = L.critic_grad(s, a, y)
    ga, la = L.actor_grad(s)

    ta, tc, fwd = _torch_nets(actor, critic)
    sn = torch.tensor(DO.normalize(s, s_min, s_max), dtype=torch.float64)
    s2n = torch.tensor(DO.normalize(s2, s_min, s_max), dtype=torch.float64)
    with torch.no_grad():
        q2 = fwd(tc, torch.cat([s2n, fwd(ta, s2n, True)], 1), False)[:, 0]
        yt = torch.tensor(r, dtype=torch.float64) + 0.99 * q2
    np.testing.assert_allclose(y, yt.numpy(), rtol=2e-5, atol=2e-6)
    loss_c = ((fwd(tc, torch.cat([sn, torch.tensor(a, dtype=torch.float64)], 1), False)[:, 0] - yt) ** 2).mean()
    gt = torch.autograd.grad(loss_c, tc)
    gt = np.concatenate([g.numpy().ravel() for g in gt])
    assert abs(lc - loss_c.item()) < 1e-5 * max(1, abs(lc))
    assert np.abs(gc - gt).max() < 1e-5 * max(1e-3, np.abs(gt).max())
    loss_a = -fwd(tc, torch.cat([sn, fwd(ta, sn, True)], 1), False).mean()
    gta = np.concatenate([g.numpy().ravel() for g in torch.autograd.grad(loss_a, ta)])
    assert np.abs(ga - gta).max() < 1e-5 * max(1e-3, np.abs(gta).max()) and np.abs(gta).max() > 1e-4


def test_adam_is_flux_0_12_form_and_soft_update():
    rng = np.random.default_rng(1)
    p = rng.normal(size=1000).astype(np.float32)
    opt = DO.Adam(1000, 1e-3)
    m = np.zeros(1000); v = np.zeros(1000); bp = [0.9, 0.999]; q = p.astype(np.float64)
    for _ in range(5):
        g = rng.normal(size=1000).astype(np.float32)
        p = opt.step(p, g)
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g.astype(np.float64) ** 2
        q = q - m / (1 - bp[0]) / (np.sqrt(v / (1 - bp[1])) + 1e-8) * float(np.float32(1e-3))
        bp = [bp[0] * 0.9, bp[1] * 0.999]
    np.testing.assert_allclose(p, q, rtol=0, atol=5e-7)
    # first step moves every parameter by ~eta (bias correction by (1 - beta) on step 1)
    o2 = DO.Adam(3, 1e-3)
    out = o2.step(np.zeros(3, np.float32), np.array([1.0, -2.0, 1e-3], np.float32))
    np.testing.assert_allclose(out, [-1e-3, 1e-3, -1e-3], rtol=1e-4)
    t = DO.soft_update(np.ones(4, np.float32), np.zeros(4, np.float32))
    assert (t == np.float32(1) - np.float32(1e-3)).all()


def test_replay_update_reduces_critic_loss_and_moves_targets_slowly():
    rng = np.random.default_rng(2)
    actor, critic = DO.init_params(5, 9, 2, 0), DO.init_params(5, 11, 1, 1)
    B = 120
    s = rng.random((B, 9)).astype(np.float32)
    a = (rng.random((B, 2)).astype(np.float32) * 2 - 1)
    r = (s[:, 0] - a[:, 1]).astype(np.float32)
    s2 = rng.random((B, 9)).astype(np.float32)
    L = DO.Learner(actor, critic, np.zeros(9, np.float32), np.ones(9, np.float32))
    losses = [L.replay(s, a, r, s2, np.zeros(B, bool))[0] for _ in range(30)]
    assert losses[-1] < 0.5 * losses[0]
    assert 0 < np.abs(L.actor_t - actor).max() < np.abs(L.actor - actor).max()
    assert np.abs(L.critic_t - critic).max() < 0.1 * np.abs(L.critic - critic).max()


def test_param_noise_restatement():
    """ParamNoise (input.jl:210-215, DDPG.jl:63-96): scalar shift on every array, adaptation by the adoption factor; the product's
    host draw (ddpg.perturb_shift) and the oracle's agree bit for bit."""
    import importlib
    import util as U
    D = importlib.import_module(U.PKG_NAME + ".ddpg")
    for seed, tick in ((1, 0), (13, 5), (2 ** 40 + 7, 123456)):
        assert D.perturb_shift(seed, tick, 0.0, 0.1) == DO.perturb_shift(seed, tick, 0.0, 0.1)
    z = np.array([DO.perturb_shift(3, t, 0.0, 1.0) for t in range(4000)])
    assert abs(z.mean()) < 0.06 and abs(z.std() - 1.0) < 0.05
    p = DO.init_params(5, 9, 2, 0)
    q = DO.add_perturb(p, 0.25)
    assert q.dtype == np.float32 and np.all(q == (p + np.float32(0.25)).astype(np.float32))
    rng = np.random.default_rng(0)
    s = rng.random((120, 9)).astype(np.float32)
    d0, s0 = DO.adapt_param_noise(p, s, np.zeros(9, np.float32), np.ones(9, np.float32), 0.0, 0.1)
    assert d0 == 0.0 and abs(s0 - 0.1 * 1.01) < 1e-15                  # identical actors: distance 0 -> sigma grows
    d1, s1 = DO.adapt_param_noise(p, s, np.zeros(9, np.float32), np.ones(9, np.float32), 0.5, 0.1)
    assert d1 > 0.1 and abs(s1 - 0.1 / 1.01) < 1e-15                   # far apart -> sigma shrinks
