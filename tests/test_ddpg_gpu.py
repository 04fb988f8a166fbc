"""GPU tests of the DDPG update kernels (replay(), DDPG.jl:121-145) against the NumPy oracle.

"Parity unpinned" by the reference for this arithmetic (Flux/Zygote/CUBLAS, un-vendored; no
reference tests): the oracle is pinned to PyTorch autograd in tests/test_ddpg_oracle.py, and the
kernels are held to it here.  Tolerances: gradients 2e-4 of the largest |g| of the tensor (fp32
accumulation order differs between MFMA tiles, wave reductions and BLAS); ADAM / soft update are
checked element-wise at 1e-7 by feeding the kernel's own gradient through the oracle's formulas."""
import importlib

import numpy as np
import pytest

import util as U
import ddpg_oracle as DO
import philox_np

pytestmark = pytest.mark.gpu


def _setup(seed=11, cap=24000, boost=30.0):
    torch = pytest.importorskip("torch")
    S = U.pkg()
    D = importlib.import_module(U.PKG_NAME + ".ddpg")
    rng = np.random.default_rng(seed)
    tab = S.tables.synthetic_table("train", 98)
    ring = D.ReplayRing(cap)
    rows = tab[rng.integers(0, tab.shape[0] - 1, cap)]
    s = np.empty((cap, 9), np.float32)
    s[:, 0] = rng.random(cap) * 6.75
    s[:, 1:] = rows[:, [1, 0, 2, 3, 4, 5, 6, 7]]
    s2 = s.copy()
    s2[:, 0] = np.clip(s[:, 0] + rng.normal(0, 1, cap), 0, 6.75)
    s2[:, 3:5] = rng.random((cap, 2)) * 5
    a = (rng.random((cap, 2)) * 2 - 1).astype(np.float32)
    r = (rng.normal(-1, 2, cap)).astype(np.float32)
    done = np.zeros(cap, np.uint8)
    done[rng.random(cap) < 0.05] = 1           # the reference never sets done, the formula still has the term
    for t, v in ((ring.s, s), (ring.a, a), (ring.r, r), (ring.s2, s2), (ring.done, done)):
        t.copy_(torch.from_numpy(v))
    ring.pushed = cap
    ag = D.Agent(seed=seed)
    pa, pc = D.init_params(seed, 9, 2, 0), D.init_params(seed, 11, 1, 1)
    pa[128000:129000] *= boost                 # lift the 3e-3 heads so every gradient path is exercised
    pc[128250:128750] *= boost
    pa[2250:2500] = rng.normal(0, 0.05, 250); pc[2750:3000] = rng.normal(0, 0.05, 250)
    ag.set_params(actor=pa, critic=pc)
    s_min, s_max = s.min(0), s.max(0)
    ag.set_norm(s_min, s_max)
    host = dict(s=s, a=a, r=r, s2=s2, done=done, s_min=s_min, s_max=s_max, pa=pa, pc=pc)
    return torch, S, D, ag, ring, host


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


# Per-block gradient bound: every Flux.params block (W1, b1, W2, b2, W3, b3) of a gradient is held, on its own, to the float64
# evaluation of the same formulas, normalised by THAT block's max-abs (a concatenated vector normalised by its global max lets a
# percent-level error in a small-magnitude block through).  fp32 accumulation over K <= 500, B <= 128 in a different order than
# BLAS gives ~1e-7 of the block max; measured on MI355X (profiles/r02_gradient_block_errors.txt): worst HIP block 1.8e-7 (actor W2),
# worst block of the f32 NumPy oracle 5.1e-7 (critic W2) -> bound = 4 x the worst measured = 2e-6.
BLOCK_TOL = 2e-6


def _block_errs(g, g64, in_dim, out_dim):
    return {n: float(np.abs(g[lo:hi] - g64[lo:hi]).max() / max(np.abs(g64[lo:hi]).max(), 1e-30)) for n, lo, hi in DO.blocks(in_dim, out_dim)}


def _assert_blocks(g, g64, in_dim, out_dim, what, tol=BLOCK_TOL):
    errs = _block_errs(g, g64, in_dim, out_dim)
    for n, lo, hi in DO.blocks(in_dim, out_dim):
        assert np.abs(g64[lo:hi]).max() > 0, (what, n, "reference block is all zero: the case does not exercise it")
    bad = {n: e for n, e in errs.items() if not e < tol}
    assert not bad, (what, errs)
    return errs


def test_sampler_matches_oracle_philox():
    torch, S, D, ag, ring, h = _setup()
    for tick in (0, 3, 999):
        idx = ag.sample_indices(tick, 24000)
        assert (idx == DO.sample_indices(ag.seed, tick, 120, 24000)).all()
    assert len(np.unique(np.concatenate([ag.sample_indices(t, 24000) for t in range(50)]))) > 5000


def test_one_update_matches_oracle():
    torch, S, D, ag, ring, h = _setup()
    tick = 3
    idx = ag.sample_indices(tick, len(ring))
    L = DO.Learner(h["pa"], h["pc"], h["s_min"], h["s_max"])
    s, a, r, s2, done = (h[k][idx] for k in ("s", "a", "r", "s2", "done"))
    y = L.targets(r, s2, done.astype(bool))
    gc_ref, lc_ref = L.critic_grad(s, a, y)

    ag.replay(ring, tick=tick)
    torch.cuda.synchronize()
    ws = ag.ws.cpu().numpy()
    # gathered + normalised minibatch
    XT = ws[0:9 * 128].reshape(9, 128)
    np.testing.assert_allclose(XT[:, :120].T, DO.normalize(s, h["s_min"], h["s_max"]), rtol=0, atol=1e-6)
    assert not XT[:, 120:].any()
    Y = ws[128 * 22:128 * 23][:120]           # WS_Y = (9 + 9 + 2 + 1 + 1) * 128
    np.testing.assert_allclose(Y, y, rtol=2e-5, atol=2e-5)
    gc = ag.grad_critic.cpu().numpy()
    assert _rel(gc, gc_ref) < 2e-4 and np.abs(gc_ref).max() > 1e-3
    gc64, _ = L.critic_grad(s, a, y, dtype=np.float64)
    errs = {"critic_hip": _assert_blocks(gc, gc64, 11, 1, "critic gradient, HIP vs float64"),
            "critic_oracle_f32": _assert_blocks(gc_ref, gc64, 11, 1, "critic gradient, f32 oracle vs float64")}
    losses = ag.losses.cpu().numpy()
    assert abs(losses[0] - lc_ref) < 1e-4 * max(1.0, abs(lc_ref))

    # ADAM + soft update, element-wise, from the kernel's own gradient
    opt = DO.Adam(len(gc), DO.ETA_CRIT)
    pc1 = opt.step(h["pc"], gc)
    crit = ag.critic.cpu().numpy()
    np.testing.assert_allclose(crit, pc1, rtol=0, atol=1e-7)
    np.testing.assert_allclose(ag.critic_t.cpu().numpy(), DO.soft_update(h["pc"], crit), rtol=0, atol=1e-7)
    np.testing.assert_allclose(ag.m_critic.cpu().numpy(), opt.m, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(ag.v_critic.cpu().numpy(), opt.v, rtol=1e-6, atol=1e-15)

    # actor gradient is taken through the UPDATED critic (DDPG.jl:137-140): give the oracle the kernel's critic
    L.critic = crit
    ga_ref, la_ref = L.actor_grad(s)
    ga = ag.grad_actor.cpu().numpy()
    assert _rel(ga, ga_ref) < 2e-4 and np.abs(ga_ref).max() > 1e-5
    ga64, _ = L.actor_grad(s, dtype=np.float64)
    errs["actor_hip"] = _assert_blocks(ga, ga64, 9, 2, "actor gradient, HIP vs float64")
    errs["actor_oracle_f32"] = _assert_blocks(ga_ref, ga64, 9, 2, "actor gradient, f32 oracle vs float64")
    print("per-block gradient errors (fraction of the block's max-abs):", errs)
    assert abs(losses[1] - la_ref) < 1e-4 * max(1.0, abs(la_ref))
    opt_a = DO.Adam(len(ga), DO.ETA_ACT)
    pa1 = opt_a.step(h["pa"], ga)
    act = ag.actor.cpu().numpy()
    np.testing.assert_allclose(act, pa1, rtol=0, atol=1e-7)
    np.testing.assert_allclose(ag.actor_t.cpu().numpy(), DO.soft_update(h["pa"], act), rtol=0, atol=1e-7)
    # every gradient entry is written each update (no stale region): second update with a fresh batch changes all blocks
    g_before = ga.copy()
    ag.replay(ring, tick=tick + 1)
    g_after = ag.grad_actor.cpu().numpy()
    for lo, hi in ((0, 2250), (2250, 2500), (2500, 127500), (127500, 128000), (128000, 129000), (129000, 129002)):
        assert (g_after[lo:hi] != g_before[lo:hi]).mean() > 0.3, (lo, hi)


def test_second_step_uses_advanced_beta_powers():
    torch, S, D, ag, ring, h = _setup(seed=5)
    opt = DO.Adam(D.N_CRITIC, DO.ETA_CRIT)
    p = h["pc"].copy()
    for t in range(3):
        ag.replay(ring, tick=t)
        g = ag.grad_critic.cpu().numpy()
        p_prev = p
        p = opt.step(p, g)
        # the kernel's critic was produced from the kernel's own previous critic: compare the step
        crit = ag.critic.cpu().numpy()
        np.testing.assert_allclose(crit, p, rtol=0, atol=2e-7)
        p = crit
    assert ag.updates == 3 and abs(ag.bp_critic[0] - 0.9 ** 4) < 1e-15


def test_training_on_fixed_ring_reduces_critic_loss():
    torch, S, D, ag, ring, h = _setup(seed=2, boost=1.0)
    # a learnable reward (the random one of _setup has an irreducible variance of 4)
    r = (0.4 * (h["s"][:, 4] - h["s"][:, 3]) + h["a"][:, 1] - 0.5 * h["s"][:, 0]).astype(np.float32)
    ring.r.copy_(torch.from_numpy(r))
    first = last = None
    n_up = 160
    for t in range(n_up):
        ag.replay(ring, tick=t % 4)            # revisit 4 minibatches
        if t < 4:
            first = (first or 0) + ag.losses[0].item() / 4
        if t >= n_up - 4:
            last = (last or 0) + ag.losses[0].item() / 4
    assert np.isfinite(last) and last < 0.5 * first, (first, last)
    assert torch.isfinite(ag.actor).all() and torch.isfinite(ag.critic).all()


def test_min_max_buffer_bootstrap():
    torch, S, D, ag, ring, h = _setup(seed=9, cap=5000)
    ag.min_max_buffer(ring, count=5000, seed=77)
    q = np.arange((5000 + 3) // 4, dtype=np.uint64)
    xs = philox_np.philox4x32_10(q & 0xFFFFFFFF, q >> np.uint64(32), 0xFFFFFFFF, philox_np.STREAM_SAMPLE, 77, 0)
    idx = (np.stack(xs, 1).reshape(-1)[:5000] % np.uint32(5000)).astype(np.int64)
    assert (ag.s_min.cpu().numpy() == h["s"][idx].min(0)).all()
    assert (ag.s_max.cpu().numpy() == h["s"][idx].max(0)).all()
    assert len(np.unique(idx)) < 5000           # with replacement


def test_pipelined_replay_excludes_the_window_being_written_and_publishes_the_actor():
    torch, S, D, ag, ring, h = _setup(seed=21)
    pub = torch.zeros_like(ag.actor)
    pos, cnt = 23900, 333                         # wraps around the end of the ring
    seen = set()
    for t in range(40):
        ag.fused = bool(t & 1)                    # both forms take the exclusion window and publish the actor
        ag.replay(ring, tick=t, exclude=(pos, cnt), publish=pub)
        idx = ag.ws[30 * 128:31 * 128].cpu().numpy().view(np.int32)[:120]      # WS_IDX
        assert ((idx >= 0) & (idx < 24000)).all()
        rel = (idx - pos) % 24000
        assert (rel >= cnt).all(), "sampled a slot inside the excluded window"
        seen.update(idx.tolist())
        assert torch.equal(pub, ag.actor)
    assert len(seen) > 3000
    with pytest.raises(S.ShemsError):
        ag.replay(ring, tick=0, exclude=(0, 24000))


def test_parameter_noise_act_and_adaptation():
    """noise_type "pn" (input.jl:210-215, DDPG.jl:63-96, 126-128, 152-156): act() evaluates actor .+ one scalar draw with no action
    noise; replay() adapts sigma_current from the distance of the two actors on the sampled minibatch."""
    torch, S, D, ag0, ring, h = _setup(seed=13)
    ag = D.Agent(seed=13, noise_type="pn", sigma=0.02)
    ag.set_params(actor=h["pa"], critic=h["pc"]); ag.set_norm(h["s_min"], h["s_max"])
    obs = h["s"][:3000]
    dev = torch.from_numpy(obs).cuda()
    out = ag.act(dev, train=True, tick=5).cpu().numpy()
    shift = ag.pn_shift
    assert shift == DO.perturb_shift(13, 5, 0.0, ag.pn_sigma) and shift != 0.0
    assert np.abs(ag.actor_perturb.cpu().numpy() - DO.add_perturb(h["pa"], shift)).max() == 0.0
    ref = DO.act_param_noise(h["pa"], obs, h["s_min"], h["s_max"], shift, dtype=np.float64)
    assert np.abs(out - ref).max() < 2e-5
    clean = ag.act(dev, train=False).cpu().numpy()
    assert np.abs(clean - DO.act(h["pa"], obs, h["s_min"], h["s_max"], False, dtype=np.float64)).max() < 1e-5
    assert np.abs(out - clean).max() > 1e-3                       # the perturbation reaches the actions
    # the fused vector step takes the same branch: its actions equal act()'s on the env's observations
    tab = S.tables.synthetic_table("train", 98)
    env = S.ShemsBatch(2048, 72, [tab], [S.make_config(98, 0, tab.shape[0])]).use_torch_stream()
    env.reset_(3, episode=1)
    s_before = np.array(env.state, np.float32, copy=True)
    a_out = torch.empty((2048, 2), dtype=torch.float32, device="cuda")
    ag.act_step(env, train=True, tick=9, a_out=a_out)
    ref = DO.act_param_noise(h["pa"], s_before, h["s_min"], h["s_max"], DO.perturb_shift(13, 9, 0.0, ag.pn_sigma), dtype=np.float64)
    assert np.abs(a_out.cpu().numpy() - ref).max() < 2e-5
    # adaptation inside replay(): sigma moves by the adoption factor in the direction the oracle's distance says
    sig = ag.pn_sigma
    idx = DO.sample_indices(13, 0, 120, len(ring))
    dist, sig_ref = DO.adapt_param_noise(h["pa"], h["s"][idx], h["s_min"], h["s_max"], DO.perturb_shift(13, 0, 0.0, sig), sig, dtype=np.float64)
    assert abs(dist - 0.1) > 1e-3                                  # not a knife-edge case
    ag.replay(ring)
    assert abs(ag.pn_sigma - sig_ref) < 1e-12 and ag.pn_sigma != sig
    torch.cuda.synchronize()


def test_fused_update_is_bit_identical_to_the_split_calls():
    """shems_ddpg_update (single replica: ADAM + soft update applied inside the gradient launches, 5 launches) against the split
    calls the data-parallel path uses (critic_grad -> critic_apply -> actor_grad -> actor_apply, 7 launches): the same gradient
    kernels and the same ADAM function, so every buffer must agree bit for bit after several updates."""
    out = []
    for fused in (True, False):
        torch, S, D, ag, ring, h = _setup(seed=17)
        ag.fused = fused
        for t in range(3):
            ag.replay(ring, tick=t)
        torch.cuda.synchronize()
        out.append({k: getattr(ag, k).clone() for k in ("actor", "critic", "actor_t", "critic_t", "m_actor", "v_actor", "m_critic", "v_critic",
                                                        "grad_actor", "grad_critic", "losses")})
    for k in out[0]:
        assert torch.equal(out[0][k], out[1][k]), k
    assert float(out[0]["grad_critic"][:3000].abs().max()) > 0 and float(out[0]["grad_actor"][:2500].abs().max()) > 0


@pytest.mark.parametrize("B", [1, 17, 128])
def test_other_batch_sizes_match_oracle(B):
    """BATCH_SIZE other than the tuned 120 (1 <= B <= 128): the unused columns of the 128-wide tiles must stay out of every sum."""
    torch, S, D, ag, ring, h = _setup(seed=23)
    ag.batch = B
    idx = DO.sample_indices(23, 4, B, len(ring))
    assert (ag.sample_indices(4, len(ring)) == idx).all()
    L = DO.Learner(h["pa"], h["pc"], h["s_min"], h["s_max"])
    s, a, r, s2, done = (h[k][idx] for k in ("s", "a", "r", "s2", "done"))
    y = L.targets(r, s2, done.astype(bool))
    gc_ref, lc_ref = L.critic_grad(s, a, y)
    ag.replay(ring, tick=4)
    torch.cuda.synchronize()
    gc = ag.grad_critic.cpu().numpy()
    assert _rel(gc, gc_ref) < 2e-4
    _assert_blocks(gc, L.critic_grad(s, a, y, dtype=np.float64)[0], 11, 1, f"critic gradient B={B}")
    losses = ag.losses.cpu().numpy()
    assert abs(losses[0] - lc_ref) < 1e-4 * max(1.0, abs(lc_ref))
    L.critic = ag.critic.cpu().numpy()
    ga_ref, la_ref = L.actor_grad(s)
    ga = ag.grad_actor.cpu().numpy()
    assert _rel(ga, ga_ref) < 2e-4
    _assert_blocks(ga, L.actor_grad(s, dtype=np.float64)[0], 9, 2, f"actor gradient B={B}")
    assert abs(losses[1] - la_ref) < 1e-4 * max(1.0, abs(la_ref))
    # one update pass holds at most 128 columns: the C ABI refuses more (Agent.replay splits wider minibatches into sub-batches,
    # tests/test_grid_points.py)
    import ctypes as C
    d = ag._ddpg_args(dict(ga=ag.grad_actor, gc=ag.grad_critic, ws=ag.ws, losses=ag.losses, batch=129))
    rs = ring.struct()
    with pytest.raises(S.ShemsError):
        S._capi.check(ag.L.shems_ddpg_critic_grad_ex(C.byref(d), C.byref(rs), len(ring), 1, 5, 0, 0, ag._stream()))
