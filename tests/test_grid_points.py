"""The other points of the reference's hyper-parameter grids (input09_08_on_01-09_eval.jl:62-66, input.jl:58-66) that run on this build:
smaller networks -- (200, 400), (150, 300) -- on the (250, 500) kernels by exact zero padding (ddpg.pad_net), and BATCH_SIZE above the
128 columns of one update pass -- 150, 200 -- as size-weighted sub-batches with ONE ADAM step (ddpg.Agent._replay_wide).  "Parity unpinned" like the rest of the learner:
the oracle here is the NumPy restatement run at the smaller size."""
import importlib
import os

import numpy as np
import pytest

import util as U
import ddpg_oracle as DO


def _D():
    return importlib.import_module(U.PKG_NAME + ".ddpg")


@pytest.fixture
def small_oracle(monkeypatch):
    monkeypatch.setattr(DO, "L1", 200)
    monkeypatch.setattr(DO, "L2", 400)
    return DO


def test_pad_roundtrip_and_same_function(small_oracle):
    D = _D()
    hid = (200, 400)
    for in_dim, out_dim, which in ((9, 2, 0), (11, 1, 1)):
        p = D.init_params(5, in_dim, out_dim, which, hid)
        assert p.size == D.net_size(in_dim, out_dim, hid) == DO.n_params(in_dim, out_dim)
        assert (p == DO.init_params(5, in_dim, out_dim, which)).all()              # same Philox draws, true fan sizes in the glorot bound
        q = D.pad_net(p, in_dim, out_dim, hid)
        assert q.size == D.net_size(in_dim, out_dim) and (D.unpad_net(q, in_dim, out_dim, hid) == p).all()
        assert np.count_nonzero(q) == np.count_nonzero(p)
    with pytest.raises(NotImplementedError):
        D.pad_net(np.zeros(D.net_size(9, 2, (300, 600)), np.float32), 9, 2, (300, 600))
    # the padded network computes the same function
    rng = np.random.default_rng(0)
    x = rng.random((64, 9)).astype(np.float32)
    p = D.init_params(5, 9, 2, 0, hid)
    p[-802:-2] *= 40
    y_small = DO.actor_forward(p, x, dtype=np.float64)
    q = D.pad_net(p, 9, 2, hid)
    small_oracle_L = (DO.L1, DO.L2)
    DO.L1, DO.L2 = 250, 500
    try:
        y_pad = DO.actor_forward(q, x, dtype=np.float64)
    finally:
        DO.L1, DO.L2 = small_oracle_L
    assert np.abs(y_small - y_pad).max() < 1e-12 and np.abs(y_small).max() > 0.1


def test_padding_survives_oracle_training():
    """The zero padding is a fixed point of replay(): relu'(0) = 0 kills every gradient entry that touches an extra unit, ADAM's step is
    0 / (0 + eps), the soft update mixes zeros.  (NumPy restatement at the padded size; the GPU twin is test_small_network_on_the_gpu.)"""
    D = _D()
    hid = (200, 400)
    rng = np.random.default_rng(1)
    pa = D.pad_net(D.init_params(3, 9, 2, 0, hid), 9, 2, hid)
    pc = D.pad_net(D.init_params(3, 11, 1, 1, hid), 11, 1, hid)
    za, zc = pa == 0, pc == 0
    L = DO.Learner(pa, pc, np.zeros(9, np.float32), np.ones(9, np.float32))
    for _ in range(3):
        s, s2 = rng.random((120, 9)).astype(np.float32), rng.random((120, 9)).astype(np.float32)
        a = (rng.random((120, 2)) * 2 - 1).astype(np.float32)
        L.replay(s, a, rng.normal(-1, 1, 120).astype(np.float32), s2, np.zeros(120, bool))
    # biases start at zero in the real part as well and move; what must stay zero is the PADDING
    pad_a = D.pad_net(np.ones(D.net_size(9, 2, hid), np.float32), 9, 2, hid) == 0
    pad_c = D.pad_net(np.ones(D.net_size(11, 1, hid), np.float32), 11, 1, hid) == 0
    assert not L.actor[pad_a].any() and not L.critic[pad_c].any() and not L.actor_t[pad_a].any() and not L.critic_t[pad_c].any()
    assert (L.actor != pa)[~pad_a].any() and za.sum() >= pad_a.sum() and zc.sum() >= pad_c.sum()


def test_checkpoint_of_a_small_network(tmp_path):
    CK = importlib.import_module(U.PKG_NAME + ".checkpoint")
    D = _D()
    hid = (200, 400)
    actor = D.init_params(1231, 9, 2, 0, hid)
    tr, sm, nm = np.arange(5, dtype=np.float32), np.array([1.5, -2.0]), np.zeros(5, np.float32)
    st = CK.save(actor, tr, sm, 3, nm, idx=7, l1=200, l2=400, case="c", rng=1231, out_dir=str(tmp_path))
    assert st.endswith("DDPG_Shems_Charger_v1_72_1001_200_400_c_1231")                  # the reference's stem carries L1, L2
    a, *_ = CK.load(idx=7, l1=200, l2=400, case="c", rng=1231, out_dir=str(tmp_path))
    assert (a == actor).all()
    with pytest.raises(ValueError):
        CK.save(D.pad_net(actor, 9, 2, hid), tr, sm, 3, nm, idx=8, l1=200, l2=400, case="c", rng=1231, out_dir=str(tmp_path))


def test_job_ids_of_smaller_networks_are_accepted():
    M = importlib.import_module(U.PKG_NAME + ".main")
    # tuned template, ternary digit 3 (of 4) = 1 -> (200, 400): code 0010 (base 3) = 3
    cfg = M.config_from_env({"JOB_ID": "11709803", "TASK_ID": "1", "GPU_ID": "0"})
    assert (cfg.L1, cfg.L2) == (200, 400)
    M._check_supported(cfg)
    cfg = M.config_from_env({"JOB_ID": "11709800", "TASK_ID": "1", "GPU_ID": "0"})      # code 0 -> (300, 600): larger than the tuned kernels,
    assert (cfg.L1, cfg.L2) == (300, 600)                                                # runs layer by layer (tests/test_wide_gpu.py)
    M._check_supported(cfg)
    D = _D()
    assert D.is_wide((300, 600)) and not D.is_wide((250, 500)) and not D.is_wide((200, 400)) and D.is_wide((250, 501))
    for code in range(81):                                                               # every code of the tuned template's grid is accepted
        M._check_supported(M.config_from_env({"JOB_ID": "117098%02d" % code, "TASK_ID": "1", "GPU_ID": "0"}))
    for code in range(27):                                                               # and every code of input.jl's
        M._check_supported(M.config_from_env({"JOB_ID": "117098%02d" % code, "TASK_ID": "1", "GPU_ID": "0", "SHEMS_INPUT_TEMPLATE": "input"}))


@pytest.mark.gpu
def test_small_network_on_the_gpu(small_oracle):
    """A (200, 400) learner on the (250, 500) kernels: the fused step's actions and one whole replay() agree with the oracle run at the
    smaller size, and after updates and steps the padding is still exactly zero in every learner tensor."""
    torch = pytest.importorskip("torch")
    S = U.pkg()
    D = _D()
    hid = (200, 400)
    rng = np.random.default_rng(4)
    ag = D.Agent(seed=21, hidden=hid)
    pa, pc = D.init_params(21, 9, 2, 0, hid), D.init_params(21, 11, 1, 1, hid)
    pa[-802:-2] *= 30; pc[-401:-1] *= 30                      # lift the 3e-3 heads so every gradient path is exercised
    ag.set_params(actor=pa, critic=pc)
    assert (ag.export_actor() == pa).all() and (ag.export_critic() == pc).all()
    cap = 24000
    tab = S.tables.synthetic_table("train", 98)
    ring = D.ReplayRing(cap)
    rows = tab[rng.integers(0, tab.shape[0] - 1, cap)]
    s = np.empty((cap, 9), np.float32); s[:, 0] = rng.random(cap) * 6.75; s[:, 1:] = rows[:, [1, 0, 2, 3, 4, 5, 6, 7]]
    s2 = s.copy(); s2[:, 0] = np.clip(s[:, 0] + rng.normal(0, 1, cap), 0, 6.75)
    a = (rng.random((cap, 2)) * 2 - 1).astype(np.float32)
    r = rng.normal(-1, 2, cap).astype(np.float32)
    for t, v in ((ring.s, s), (ring.a, a), (ring.r, r), (ring.s2, s2)):
        t.copy_(torch.from_numpy(v))
    ring.pushed = cap
    lo, hi = s.min(0), s.max(0)
    ag.set_norm(lo, hi)
    # act(): the padded actor on the MFMA path against the small oracle in float64
    obs = torch.from_numpy(s[:777]).cuda()
    got = ag.act(obs, train=False).cpu().numpy()
    want = DO.act(pa, s[:777], lo, hi, False, dtype=np.float64)
    assert np.abs(got - want).max() < 1e-5 and np.abs(want).max() > 0.1
    # one replay()
    tick = 2
    idx = ag.sample_indices(tick, len(ring))
    L = DO.Learner(pa, pc, lo, hi)
    L.replay(s[idx], a[idx], r[idx], s2[idx], np.zeros(len(idx), bool))
    ag.replay(ring, tick=tick)
    torch.cuda.synchronize()
    for name, got_t, want_v in (("critic", ag.export_critic(), L.critic), ("actor", ag.export_actor(), L.actor),
                                ("critic_t", ag.export_critic(ag.critic_t), L.critic_t), ("actor_t", ag.export_actor(ag.actor_t), L.actor_t)):
        assert np.abs(got_t - want_v).max() < 2e-6, name
    assert np.abs(ag.export_critic() - pc).max() > 1e-5
    # more updates and fused steps; then the padding of every learner tensor is still exactly zero
    env = S.ShemsBatch(2048, 72, [tab], [S.make_config(98, 0, tab.shape[0])]).use_torch_stream()
    env.reset_(3, episode=0)
    for t in range(10):
        ag.act_step(env, train=True, tick=t, ring=ring, window=D.RingWindow(ring.pos, 28, 0))
        ring.pushed += 28
        ag.replay(ring)
    torch.cuda.synchronize()
    pad_a = D.pad_net(np.ones(D.net_size(9, 2, hid), np.float32), 9, 2, hid) == 0
    pad_c = D.pad_net(np.ones(D.net_size(11, 1, hid), np.float32), 11, 1, hid) == 0
    for name in ("actor", "actor_t", "m_actor", "v_actor", "grad_actor"):
        assert not getattr(ag, name).cpu().numpy()[pad_a].any(), name
    for name in ("critic", "critic_t", "m_critic", "v_critic", "grad_critic"):
        assert not getattr(ag, name).cpu().numpy()[pad_c].any(), name
    env.check_error()
    env.close()


def test_job_ids_of_wide_batches_are_accepted():
    M = importlib.import_module(U.PKG_NAME + ".main")
    cfg = M.config_from_env({"JOB_ID": "11709862", "TASK_ID": "1", "GPU_ID": "0"})      # 62 = 2022 (base 3): BATCH 150, (250, 500)
    assert cfg.BATCH_SIZE == 150 and (cfg.L1, cfg.L2) == (250, 500)
    M._check_supported(cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [150, 200])
def test_wide_minibatch_update_matches_the_oracle_on_the_whole_batch(batch):
    """BATCH_SIZE = 150 (tuned template) / 200 (input.jl): replay() runs two gradient passes of 75 / 100 columns and combines them
    before ONE ADAM step per network.  The oracle takes the SAME transitions as one minibatch of 150 / 200 (Flux.mse and -mean(q) over
    all of them, DDPG.jl:134-140): parameters, targets and moments must agree as for the one-pass update."""
    torch = pytest.importorskip("torch")
    S = U.pkg()
    D = _D()
    rng = np.random.default_rng(8)
    ag = D.Agent(seed=31)
    ag.batch = batch
    pa, pc = D.init_params(31, 9, 2, 0), D.init_params(31, 11, 1, 1)
    pa[128000:129000] *= 30; pc[128250:128750] *= 30
    ag.set_params(actor=pa, critic=pc)
    cap = 24000
    tab = S.tables.synthetic_table("train", 98)
    ring = D.ReplayRing(cap)
    rows = tab[rng.integers(0, tab.shape[0] - 1, cap)]
    s = np.empty((cap, 9), np.float32); s[:, 0] = rng.random(cap) * 6.75; s[:, 1:] = rows[:, [1, 0, 2, 3, 4, 5, 6, 7]]
    s2 = s.copy(); s2[:, 0] = np.clip(s[:, 0] + rng.normal(0, 1, cap), 0, 6.75)
    a = (rng.random((cap, 2)) * 2 - 1).astype(np.float32)
    r = rng.normal(-1, 2, cap).astype(np.float32)
    for t, v in ((ring.s, s), (ring.a, a), (ring.r, r), (ring.s2, s2)):
        t.copy_(torch.from_numpy(v))
    ring.pushed = cap
    lo, hi = s.min(0), s.max(0)
    ag.set_norm(lo, hi)
    sizes = [sb["batch"] for sb in ag.sub_batches()]
    assert sizes == [batch // 2, batch // 2]
    L = DO.Learner(pa, pc, lo, hi)
    for tick in (5, 6):                                                     # two updates: the second with advanced beta powers
        idx = np.concatenate([ag.sample_indices(tick * 8 + i, len(ring), batch=b) for i, b in enumerate(sizes)])
        assert idx.shape == (batch,)
        lc, la = L.replay(s[idx], a[idx], r[idx], s2[idx], np.zeros(batch, bool))
        ag.replay(ring, tick=tick)
        torch.cuda.synchronize()
        for name, got, want in (("critic", ag.critic, L.critic), ("actor", ag.actor, L.actor), ("critic_t", ag.critic_t, L.critic_t),
                                ("actor_t", ag.actor_t, L.actor_t)):
            assert np.abs(got.cpu().numpy() - want).max() < 3e-6, (name, tick)
        np.testing.assert_allclose(ag.m_critic.cpu().numpy(), L.opt_c.m, rtol=2e-4, atol=1e-9)
        np.testing.assert_allclose(ag.m_actor.cpu().numpy(), L.opt_a.m, rtol=2e-4, atol=1e-10)
        losses = ag.losses.cpu().numpy()
        assert abs(losses[0] - lc) < 1e-4 * max(1.0, abs(lc)) and abs(losses[1] - la) < 1e-4 * max(1.0, abs(la))
    assert np.abs(ag.critic.cpu().numpy() - pc).max() > 1e-4
