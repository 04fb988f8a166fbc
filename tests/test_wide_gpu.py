"""Networks wider than the (250, 500) the tuned kernels are built for -- the (300, 600) point of the reference's grids
(input09_08_on_01-09_eval.jl:62-66, input.jl:58-66) -- on the layer-by-layer path (csrc/shems_wide.hip, shems_wide_*): act(), the fused
vector step, replay(), the tracking pass and the entry script, each held to the same oracle and the same bounds as the tuned path
(tests/test_policy_gpu.py, test_ddpg_gpu.py, test_harness.py); and the two implementations against each other at (250, 500).
"Parity unpinned" like the rest of the learner: the oracle is the NumPy restatement run at that size."""
import csv
import importlib
import os

import numpy as np
import pytest

import util as U
from util import oracle_c
import ddpg_oracle as DO

pytestmark = pytest.mark.gpu
HID = (300, 600)
ATOL = 1e-5
BLOCK_TOL = 2e-6


@pytest.fixture
def wide_oracle(monkeypatch):
    monkeypatch.setattr(DO, "L1", HID[0])
    monkeypatch.setattr(DO, "L2", HID[1])
    return DO


def _mods():
    torch = pytest.importorskip("torch")
    return torch, U.pkg(), importlib.import_module(U.PKG_NAME + ".ddpg")


def _boosted(D, seed, hid):
    """Initial networks with the 3e-3 heads lifted and non-zero biases, so tanh and every gradient path are exercised."""
    rng = np.random.default_rng(seed)
    pa, pc = D.init_params(seed, 9, 2, 0, hid), D.init_params(seed, 11, 1, 1, hid)
    h1, h2 = hid
    pa[-(2 * h2 + 2):-2] *= 40.0
    pa[-2:] = [0.3, -0.2]
    pc[-(h2 + 1):-1] *= 30.0
    pa[9 * h1:10 * h1] = rng.normal(0, 0.05, h1)
    pc[11 * h1:12 * h1] = rng.normal(0, 0.05, h1)
    return pa, pc


def _rand_obs(rng, n):
    tab = U.tables_mod().synthetic_table("train", 98)
    rows = tab[rng.integers(0, tab.shape[0], n)]
    obs = np.empty((n, 9), np.float32)
    obs[:, 0] = rng.random(n) * 6.75
    obs[:, 1:] = rows[:, [1, 0, 2, 3, 4, 5, 6, 7]]
    return obs


@pytest.mark.parametrize("m", [1, 777, 20000])
def test_wide_act_matches_the_float64_evaluation(wide_oracle, m):
    torch, S, D = _mods()
    rng = np.random.default_rng(m)
    ag = D.Agent(seed=1231, hidden=HID)
    assert ag.wide and ag.actor.numel() == D.net_size(9, 2, HID) == DO.n_params(9, 2) and ag.critic.numel() == DO.n_params(11, 1)
    pa, _ = _boosted(D, 1231, HID)
    ag.set_params(actor=pa)
    assert (ag.export_actor() == pa).all()
    obs = _rand_obs(rng, m)
    s_min, s_max = obs.min(0) - 0.01, obs.max(0) + 0.5
    s_max[5] = s_min[5]                                       # constant feature: the 1f-8 denominator
    ag.set_norm(s_min, s_max)
    out = ag.act(torch.from_numpy(obs).cuda(), train=False).cpu().numpy()
    ref = DO.act(pa, obs, s_min.astype(np.float32), s_max.astype(np.float32), False, dtype=np.float64)
    assert out.shape == (m, 2) and np.abs(out - ref).max() < ATOL and np.abs(out).max() > 0.3
    outn = ag.act(torch.from_numpy(obs).cuda(), train=True, tick=17).cpu().numpy()
    refn = DO.act(pa, obs, s_min.astype(np.float32), s_max.astype(np.float32), True, seed=1231, tick=17, dtype=np.float64)
    assert np.abs(outn - refn).max() < 5e-6 + ATOL and outn.min() >= -1 and outn.max() <= 1 and (m == 1 or np.abs(outn - out).max() > 0.05)


def test_wide_fused_step_equals_act_then_oracle_step_and_fills_ring(wide_oracle):
    torch, S, D = _mods()
    n, nsteps = 3000 + 5, 3
    tab = S.tables.synthetic_table("train", 98)
    env = S.ShemsBatch(n, 72, [tab], [S.make_config(98, 0, tab.shape[0])]).use_torch_stream()
    ref = oracle_c.Batch(n, 72, tab, oracle_c.profile(98))
    ag = D.Agent(seed=77, hidden=HID)
    pa, _ = _boosted(D, 77, HID)
    ag.set_params(actor=pa)
    env.reset_(5, episode=0)
    st0 = env.state
    ag.set_norm(st0.min(0), st0.max(0))
    ref.set_state(st0, env.idx)
    ring = D.ReplayRing(5000)
    a_out = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    rew = torch.empty(n, dtype=torch.float64, device="cuda")
    rew32 = torch.empty(n, dtype=torch.float32, device="cuda")
    ret = torch.zeros(n, dtype=torch.float64, device="cuda")
    pos, tot = 0, np.zeros(n)
    for t in range(nsteps):
        pre = env.state
        win = D.RingWindow(pos % ring.capacity, 333, (t * 333) % n)
        ag.act_step(env, train=True, tick=t, a_out=a_out, rewards=rew, rewards_f32=rew32, returns_acc=ret, ring=ring, window=win)
        env.check_error()
        a = a_out.cpu().numpy()
        want = DO.act(pa, pre, st0.min(0), st0.max(0), True, seed=77, tick=t, dtype=np.float64)
        assert np.abs(a - want).max() < 5e-6 + ATOL                      # (1) the action is act() of the pre-step observation
        rc, r_ref, o_ref, _ = ref.step(oracle_c.scale_action(a), 0)      # (2) given that action the transition is the oracle's, bit for bit
        assert rc == 0
        r = rew.cpu().numpy()
        tot += r
        assert (U.bits64(r) == U.bits64(r_ref)).all() and (U.bits32(env.state) == U.bits32(o_ref)).all()
        assert (rew32.cpu().numpy() == r_ref.astype(np.float32)).all()
        rel = (np.arange(n) - (t * 333) % n) % n                          # (3) the ring window
        sel = np.where(rel < 333)[0]
        slots = (pos + rel[sel]) % ring.capacity
        assert (U.bits32(ring.s.cpu().numpy()[slots]) == U.bits32(pre[sel])).all()
        assert (U.bits32(ring.s2.cpu().numpy()[slots]) == U.bits32(o_ref[sel])).all()
        assert (U.bits32(ring.a.cpu().numpy()[slots]) == U.bits32(a[sel])).all()
        assert (ring.r.cpu().numpy()[slots] == r_ref[sel].astype(np.float32)).all()
        pos += 333
    assert (env.idx == ref.idx()).all() and (env.step == nsteps).all() and (ret.cpu().numpy() == tot).all()
    env.close()


def _ring(torch, S, D, rng, cap=24000):
    tab = S.tables.synthetic_table("train", 98)
    ring = D.ReplayRing(cap)
    rows = tab[rng.integers(0, tab.shape[0] - 1, cap)]
    s = np.empty((cap, 9), np.float32); s[:, 0] = rng.random(cap) * 6.75; s[:, 1:] = rows[:, [1, 0, 2, 3, 4, 5, 6, 7]]
    s2 = s.copy(); s2[:, 0] = np.clip(s[:, 0] + rng.normal(0, 1, cap), 0, 6.75); s2[:, 3:5] = rng.random((cap, 2)) * 5
    a = (rng.random((cap, 2)) * 2 - 1).astype(np.float32)
    r = rng.normal(-1, 2, cap).astype(np.float32)
    done = np.zeros(cap, np.uint8); done[rng.random(cap) < 0.05] = 1
    for t, v in ((ring.s, s), (ring.a, a), (ring.r, r), (ring.s2, s2), (ring.done, done)):
        t.copy_(torch.from_numpy(v))
    ring.pushed = cap
    return ring, dict(s=s, a=a, r=r, s2=s2, done=done, s_min=s.min(0), s_max=s.max(0))


def _assert_blocks(g, g64, in_dim, out_dim, what):
    errs = {}
    for name, lo, hi in DO.blocks(in_dim, out_dim):
        assert np.abs(g64[lo:hi]).max() > 0, (what, name, "reference block is all zero")
        errs[name] = float(np.abs(g[lo:hi] - g64[lo:hi]).max() / np.abs(g64[lo:hi]).max())
    assert all(e < BLOCK_TOL for e in errs.values()), (what, errs)
    return errs


@pytest.mark.parametrize("B", [120, 17])
def test_wide_update_matches_the_oracle(wide_oracle, B):
    """replay() at (300, 600): every gradient block against the float64 evaluation (the tuned kernels' bound, 2e-6 of the block's
    max-abs), ADAM + soft update element-wise from the kernel's own gradient, the actor's gradient through the UPDATED critic."""
    torch, S, D = _mods()
    ag = D.Agent(seed=11, hidden=HID)
    ag.batch = B
    ring, h = _ring(torch, S, D, np.random.default_rng(11))
    pa, pc = _boosted(D, 11, HID)
    ag.set_params(actor=pa, critic=pc)
    ag.set_norm(h["s_min"], h["s_max"])
    tick = 3
    idx = ag.sample_indices(tick, len(ring))
    assert (idx == DO.sample_indices(ag.seed, tick, B, 24000)).all()
    L = DO.Learner(pa, pc, h["s_min"], h["s_max"])
    s, a, r, s2, done = (h[k][idx] for k in ("s", "a", "r", "s2", "done"))
    y = L.targets(r, s2, done.astype(bool))
    gc_ref, lc_ref = L.critic_grad(s, a, y)
    gc64, _ = L.critic_grad(s, a, y, dtype=np.float64)
    ag.replay(ring, tick=tick)
    torch.cuda.synchronize()
    gc = ag.grad_critic.cpu().numpy()
    errs = {"critic": _assert_blocks(gc, gc64, 11, 1, "critic gradient, wide path vs float64")}
    losses = ag.losses.cpu().numpy()
    assert abs(losses[0] - lc_ref) < 1e-4 * max(1.0, abs(lc_ref))
    opt = DO.Adam(len(gc), DO.ETA_CRIT)
    pc1 = opt.step(pc, gc)
    crit = ag.critic.cpu().numpy()
    np.testing.assert_allclose(crit, pc1, rtol=0, atol=1e-7)
    np.testing.assert_allclose(ag.critic_t.cpu().numpy(), DO.soft_update(pc, crit), rtol=0, atol=1e-7)
    np.testing.assert_allclose(ag.m_critic.cpu().numpy(), opt.m, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(ag.v_critic.cpu().numpy(), opt.v, rtol=1e-6, atol=1e-15)
    L.critic = crit
    ga_ref, la_ref = L.actor_grad(s)
    ga64, _ = L.actor_grad(s, dtype=np.float64)
    ga = ag.grad_actor.cpu().numpy()
    errs["actor"] = _assert_blocks(ga, ga64, 9, 2, "actor gradient, wide path vs float64")
    print("per-block gradient errors, (300, 600):", errs)
    assert abs(losses[1] - la_ref) < 1e-4 * max(1.0, abs(la_ref))
    opt_a = DO.Adam(len(ga), DO.ETA_ACT)
    pa1 = opt_a.step(pa, ga)
    act = ag.actor.cpu().numpy()
    np.testing.assert_allclose(act, pa1, rtol=0, atol=1e-7)
    np.testing.assert_allclose(ag.actor_t.cpu().numpy(), DO.soft_update(pa, act), rtol=0, atol=1e-7)
    # a second update (advanced beta powers), the whole learner against the oracle's replay()
    L2 = DO.Learner(pa, pc, h["s_min"], h["s_max"])
    ag2 = D.Agent(seed=11, hidden=HID)
    ag2.batch = B
    ag2.set_params(actor=pa, critic=pc)
    ag2.set_norm(h["s_min"], h["s_max"])
    for tk in (5, 6):
        i2 = ag2.sample_indices(tk, len(ring))
        L2.replay(h["s"][i2], h["a"][i2], h["r"][i2], h["s2"][i2], h["done"][i2].astype(bool))
        ag2.replay(ring, tick=tk)
    torch.cuda.synchronize()
    for name, got, want in (("critic", ag2.critic, L2.critic), ("actor", ag2.actor, L2.actor), ("critic_t", ag2.critic_t, L2.critic_t),
                            ("actor_t", ag2.actor_t, L2.actor_t)):
        assert np.abs(got.cpu().numpy() - want).max() < 3e-6, name
    # the run is reproducible bit for bit (fixed summation orders)
    ag3 = D.Agent(seed=11, hidden=HID)
    ag3.batch = B
    ag3.set_params(actor=pa, critic=pc)
    ag3.set_norm(h["s_min"], h["s_max"])
    for tk in (5, 6):
        ag3.replay(ring, tick=tk)
    assert torch.equal(ag3.actor, ag2.actor) and torch.equal(ag3.critic_t, ag2.critic_t)


def test_wide_path_on_sizes_that_divide_nothing(monkeypatch):
    """(257, 513): one more than the tuned widths, odd, no multiple of any tile or vector width -- the matrix products' edge handling
    (clamped loads, zero-filled stage tails, partial tiles), the column sums, the ADAM sweep's tail and the tracking kernel's loops."""
    torch, S, D = _mods()
    hid = (257, 513)
    monkeypatch.setattr(DO, "L1", hid[0]); monkeypatch.setattr(DO, "L2", hid[1])
    assert D.is_wide(hid)
    ring, h = _ring(torch, S, D, np.random.default_rng(2))
    pa, pc = _boosted(D, 13, hid)
    ag = D.Agent(seed=13, hidden=hid)
    ag.batch = 97
    ag.set_params(actor=pa, critic=pc)
    ag.set_norm(h["s_min"], h["s_max"])
    obs = h["s"][:1001]
    got = ag.act(torch.from_numpy(obs).cuda(), train=False).cpu().numpy()
    want = DO.act(pa, obs, h["s_min"], h["s_max"], False, dtype=np.float64)
    assert np.abs(got - want).max() < ATOL and np.abs(want).max() > 0.3
    got1 = ag.act(torch.from_numpy(obs[:3]).cuda(), train=False).cpu().numpy()       # the small-M kernel on the same rows
    assert np.abs(got1 - want[:3]).max() < ATOL
    L = DO.Learner(pa, pc, h["s_min"], h["s_max"])
    idx = ag.sample_indices(9, len(ring))
    s, a, r, s2, done = (h[k][idx] for k in ("s", "a", "r", "s2", "done"))
    y = L.targets(r, s2, done.astype(bool))
    gc64, _ = L.critic_grad(s, a, y, dtype=np.float64)
    ag.replay(ring, tick=9)
    torch.cuda.synchronize()
    _assert_blocks(ag.grad_critic.cpu().numpy(), gc64, 11, 1, "critic gradient at (257, 513)")
    L.critic = ag.critic.cpu().numpy()
    ga64, _ = L.actor_grad(s, dtype=np.float64)
    _assert_blocks(ag.grad_actor.cpu().numpy(), ga64, 9, 2, "actor gradient at (257, 513)")
    L3 = DO.Learner(pa, pc, h["s_min"], h["s_max"])
    L3.replay(s, a, r, s2, done.astype(bool))
    for name, got_t, want_v in (("critic", ag.critic, L3.critic), ("actor", ag.actor, L3.actor), ("critic_t", ag.critic_t, L3.critic_t),
                                ("actor_t", ag.actor_t, L3.actor_t)):
        assert np.abs(got_t.cpu().numpy() - want_v).max() < 3e-6, name
    H = importlib.import_module(U.PKG_NAME + ".harness")
    ev = S.tables.synthetic_table("eval", 98)
    env = S.ShemsBatch(1, 1439, [ev], [S.make_config(98, 0, ev.shape[0])])
    ag.set_params(actor=pa)
    total, res = H.inference(env, ag, track=1, num_steps=50)
    ref = oracle_c.Batch(1, 1439, ev, oracle_c.profile(98)); ref.reset(True)
    for t in range(50):
        act = DO.act(pa, ref.state(), h["s_min"], h["s_max"], False, dtype=np.float64)
        tgt = res[t, [21, 2]].astype(np.float32)[None]
        assert np.abs(oracle_c.scale_action(act) - tgt).max() < 1e-5
        rc, r_, o_, rr = ref.step(tgt, 1, want_results=True)
        assert rc == 0 and (U.bits64(rr[0]) == U.bits64(res[t])).all()
    env.close()


def test_wide_path_agrees_with_the_tuned_kernels_at_250_500():
    """Two independent implementations of the same functions: the layer-by-layer path forced onto the tuned size against the fused
    kernels -- actions, both gradients, and the learner after two updates."""
    torch, S, D = _mods()
    ring, h = _ring(torch, S, D, np.random.default_rng(3))
    pa, pc = _boosted(D, 9, (250, 500))
    ags = []
    for wide in (False, True):
        ag = D.Agent(seed=9, wide=wide)
        assert ag.wide is wide
        ag.set_params(actor=pa, critic=pc)
        ag.set_norm(h["s_min"], h["s_max"])
        ags.append(ag)
    obs = torch.from_numpy(h["s"][:5000]).cuda()
    a0, a1 = (ag.act(obs, train=True, tick=4).cpu().numpy() for ag in ags)
    assert np.abs(a0 - a1).max() < 2e-6 and np.abs(a0).max() > 0.3
    for tk in (1, 2):
        for ag in ags:
            ag.replay(ring, tick=tk)
        torch.cuda.synchronize()
        if tk == 1:
            for name, i, o in (("grad_critic", 11, 1), ("grad_actor", 9, 2)):
                g0, g1 = (getattr(ag, name).cpu().numpy() for ag in ags)
                for bn, lo, hi in DO.blocks(i, o):
                    assert np.abs(g0[lo:hi] - g1[lo:hi]).max() < 4e-6 * np.abs(g0[lo:hi]).max(), (name, bn)
    for name in ("actor", "critic", "actor_t", "critic_t"):
        assert np.abs(getattr(ags[0], name).cpu().numpy() - getattr(ags[1], name).cpu().numpy()).max() < 3e-6, name


def test_wide_minibatch_of_150_and_parameter_noise(wide_oracle):
    """BATCH_SIZE 150 on a wide network (sub-batches + one ADAM step) against the oracle on the whole batch; parameter noise runs
    (no padding to un-zero in a wide network's own layout)."""
    torch, S, D = _mods()
    ring, h = _ring(torch, S, D, np.random.default_rng(8))
    pa, pc = _boosted(D, 31, HID)
    ag = D.Agent(seed=31, hidden=HID)
    ag.batch = 150
    ag.set_params(actor=pa, critic=pc)
    ag.set_norm(h["s_min"], h["s_max"])
    sizes = [sb["batch"] for sb in ag.sub_batches()]
    assert sizes == [75, 75]
    L = DO.Learner(pa, pc, h["s_min"], h["s_max"])
    idx = np.concatenate([ag.sample_indices(5 * 8 + i, len(ring), batch=b) for i, b in enumerate(sizes)])
    L.replay(h["s"][idx], h["a"][idx], h["r"][idx], h["s2"][idx], h["done"][idx].astype(bool))
    ag.replay(ring, tick=5)
    torch.cuda.synchronize()
    for name, got, want in (("critic", ag.critic, L.critic), ("actor", ag.actor, L.actor), ("critic_t", ag.critic_t, L.critic_t),
                            ("actor_t", ag.actor_t, L.actor_t)):
        assert np.abs(got.cpu().numpy() - want).max() < 3e-6, name
    pn = D.Agent(seed=31, hidden=HID, noise_type="pn", sigma=0.05)
    pn.set_params(actor=pa, critic=pc)
    pn.set_norm(h["s_min"], h["s_max"])
    sig0 = pn.pn_sigma
    pn.replay(ring, tick=1)
    torch.cuda.synchronize()
    assert pn.pn_sigma != sig0 and np.isfinite(pn.actor.cpu().numpy()).all()
    shifted = pn.actor_perturb.cpu().numpy() - pa                      # add_perturb! ran on the pre-update actor: one scalar everywhere
    assert np.abs(shifted - shifted[0]).max() < 1e-6 and abs(shifted[0]) > 0


def test_wide_tracking_pass(wide_oracle):
    """inference(env; track = 1) with a (300, 600) actor: one launch (k_track<true>), the targets each results row holds within 1e-5 of the
    float64 evaluation, and the oracle reproducing every row bit for bit from those targets; several actors in one launch."""
    torch, S, D = _mods()
    H = importlib.import_module(U.PKG_NAME + ".harness")
    ev = S.tables.synthetic_table("eval", 98)
    env = S.ShemsBatch(2, 1439, [ev], [S.make_config(98, 0, ev.shape[0])])
    st = np.concatenate([ev[:, [1, 1, 0, 2, 3, 4, 5, 6, 7]]]); st[:, 0] = np.linspace(0, 6.75, len(st))
    lo, hi = st.min(0), st.max(0)
    ag = D.Agent(seed=4, hidden=HID)
    pa, _ = _boosted(D, 4, HID)
    ag.set_params(actor=pa)
    ag.set_norm(lo, hi)
    steps = 200
    total, res = H.inference(env, ag, track=1, num_steps=steps)
    assert res.shape == (steps, 23)
    ref = oracle_c.Batch(1, 1439, ev, oracle_c.profile(98)); ref.reset(True)
    for t in range(steps):
        a = DO.act(pa, ref.state(), lo, hi, False, dtype=np.float64)
        tgt = res[t, [21, 2]].astype(np.float32)[None]
        assert np.abs(oracle_c.scale_action(a) - tgt).max() < 1e-5
        rc, r, o, rr = ref.step(tgt, 1, want_results=True)
        assert rc == 0 and (U.bits64(rr[0]) == U.bits64(res[t])).all()
    assert abs(total[0] - res[:, 5].sum()) < 1e-9 * max(1.0, abs(res[:, 5].sum()))
    env.close()
    actors = [pa, _boosted(D, 5, HID)[0], _boosted(D, 6, HID)[0]]
    many = S.ShemsBatch(len(actors), 1439, [ev], [S.make_config(98, 0, ev.shape[0])])
    tot, resm = H.inference_many(many, np.stack(actors), lo, hi, num_steps=steps, hidden=HID)
    many.close()
    assert resm.shape == (3, steps, 23) and (U.bits64(resm[0]) == U.bits64(res)).all() and len({float(x) for x in tot}) == 3


def test_entry_script_on_the_300_600_grid_point(tmp_path):
    """JOB_ID suffix 00 = ternary 0000 of the tuned template: (L1, L2) = (300, 600) (input09_08_on_01-09_eval.jl:62-66).  Trains on the
    wide path, writes 300 x 600 chains under the reference's names, runs the tracking passes."""
    pytest.importorskip("torch")
    M = importlib.import_module(U.PKG_NAME + ".main")
    env = {"JOB_ID": "1179800", "TASK_ID": "1", "GPU_ID": "0", "SHEMS_NUM_EP": "2", "SHEMS_NUM_SEEDS": "1", "SHEMS_NUM_ENVS": "64",
           "SHEMS_SYNTHETIC_DATA": "1"}
    cwd0 = os.getcwd()
    try:
        cfg, written = M.main(env, cwd=str(tmp_path), log=lambda *_: None)
    finally:
        os.chdir(cwd0)
    assert (cfg.L1, cfg.L2) == (300, 600) and len(written) == 2 and all("_300_600_" in w for w in written)
    B = importlib.import_module(U.PKG_NAME + ".bson_chain")
    stem = f"DDPG_Shems_Charger_v1_72_2_300_600_{cfg.case}_1231"
    a = B.load_chain(str(tmp_path / f"out/bson/{stem}_actor_2.bson"), hidden=HID)
    assert a.size == 9 * 300 + 300 + 300 * 600 + 600 + 600 * 2 + 2 and np.isfinite(a).all() and np.count_nonzero(a) > 180000
    rows = list(csv.reader(open(tmp_path / written[0])))
    assert len(rows) == 1 + 1439 and np.isfinite(np.array(rows[1:], float)).all()


def test_entry_script_input_template_300_600_ou_noise_batch_200(tmp_path):
    """The untuned template (input.jl:58-100) at its heaviest code: (300, 600), BATCH_SIZE 200 (two sub-batches of 100, one ADAM step),
    MEM_SIZE 30 000, OU noise -- the wide path with every host-side branch that differs from the tuned run."""
    pytest.importorskip("torch")
    M = importlib.import_module(U.PKG_NAME + ".main")
    env = {"JOB_ID": "1179802", "TASK_ID": "1", "GPU_ID": "0", "SHEMS_INPUT_TEMPLATE": "input", "SHEMS_NUM_EP": "2", "SHEMS_NUM_SEEDS": "1",
           "SHEMS_NUM_ENVS": "64", "SHEMS_SYNTHETIC_DATA": "1"}
    cwd0 = os.getcwd()
    try:
        cfg, written = M.main(env, cwd=str(tmp_path), log=lambda *_: None)
    finally:
        os.chdir(cwd0)
    assert (cfg.L1, cfg.L2, cfg.BATCH_SIZE, cfg.MEM_SIZE, cfg.noise_type) == (300, 600, 200, 30000, "ou")
    assert len(written) == 2 and all("_300_600_" in w for w in written)
    rows = list(csv.reader(open(tmp_path / written[0])))
    assert len(rows) > 100 and np.isfinite(np.array(rows[1:], float)).all()
