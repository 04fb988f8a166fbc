"""GPU parity tests of the batched environment, through the C ABI, against the CPU oracle.

Bar (BASELINE.json north_star): bit-exact idx/step/c_ev bookkeeping; Float32 state and Float64
reward compared BIT-EXACT here (the kernel reproduces the reference's mixed f32/f64 sequence, so
the 1e-5 relative tolerance of the north star is met with zero slack).
"""
import os

import numpy as np
import pytest

import util as U
from util import oracle_c
import philox_np

pytestmark = pytest.mark.gpu

GOLD = os.path.join(U.ROOT, "tests", "golden")


def _S():
    return U.pkg()


def _hash_actions(n, t):
    k = np.arange(n, dtype=np.int64) * 72 + t
    return np.stack([((k * 2654435761) % 1000) / 999.0, ((k * 40503 + 7) % 1000) / 999.0], 1).astype(np.float32)


def _profile_cfg(S, cid, row0, nrow, w=0.01, pot=2.0, pen=0.1):
    return S.make_config(cid, row0, nrow, w, pot, pen), oracle_c.profile(cid, w, pot, pen)


@pytest.mark.parametrize("cid,w,pot,pen", [(98, 0.01, 2.0, 0.1), (4, 0.04, 2.0, 0.1), (6, 1.0, 1.0, 0.0)])
def test_single_step_cases_bit_exact(cid, w, pot, pen):
    """Adversarial single steps: every env sits on row 1 of its own 2-row table (4096 configs)."""
    S = _S()
    n = 4096
    cases = U.single_step_cases(1000 + cid, n)
    prof = oracle_c.profile(cid, w, pot, pen)
    r_ref, o_ref, res_ref = U.run_oracle_c(cases, prof)
    rows = np.empty((2 * n, 8), np.float32)
    rows[0::2], rows[1::2] = cases["row_cur"], cases["row_next"]
    cfgs = [S.make_config(cid, 2 * i, 2, w, pot, pen) for i in range(n)]
    for mode in (0, -1):
        env = S.ShemsBatch(n, 1, [rows], cfgs, np.arange(n, dtype=np.uint16))
        env.state = cases["obs"]
        env.idx = np.ones(n, np.int32)
        sel = cases["mode"] == mode
        # envs of the other mode still step (with these actions) but are not compared
        r, s2, res = env.step_(None, cases["act"], track=(1 if mode == 0 else -0.5))
        assert (U.bits64(r[sel]) == U.bits64(r_ref[sel])).all()
        assert (U.bits32(s2[sel]) == U.bits32(o_ref[sel])).all()
        assert (U.bits64(res[sel]) == U.bits64(res_ref[sel])).all()
        assert (env.idx == 2).all() and (env.step == 1).all()
        assert (U.bits32(env.state[sel]) == U.bits32(o_ref[sel])).all()
        # action(env, a) / action(env, track) agree with step!'s B, EV columns
        env.state = cases["obs"]
        if mode == 0:
            be = env.action(cases["act"])
            assert (U.bits32(be[sel]) == U.bits32(res_ref[sel][:, [20, 3]].astype(np.float32))).all()
        env.close()


def test_episode_72_steps_vs_oracle():
    S = _S()
    T = S.tables
    n = 1024
    tab = T.synthetic_table("train", 98)
    cfg, prof = _profile_cfg(S, 98, 0, tab.shape[0])
    env = S.ShemsBatch(n, 72, [tab], [cfg])
    ref = oracle_c.Batch(n, 72, tab, prof)
    rng = np.random.default_rng(3)
    idx0 = rng.integers(1, tab.shape[0] - 72 + 1, n).astype(np.int32)
    soc0 = (rng.random(n) * 6.75).astype(np.float32)
    env.reset_(0, idx0=idx0, soc_b0=soc0)
    assert ref.reset(False, idx0, soc0) == 0
    assert (env.idx == ref.idx()).all() and (env.step == 0).all()
    assert (U.bits32(env.state) == U.bits32(ref.state())).all()
    tot = np.zeros(n)
    for t in range(72):
        a = _hash_actions(n, t)
        r, s2 = env.step_(None, a)
        rc, r_ref, o_ref, _ = ref.step(a, 0)
        assert rc == 0
        assert (U.bits64(r) == U.bits64(r_ref)).all(), f"reward mismatch at step {t}"
        assert (U.bits32(s2) == U.bits32(o_ref)).all(), f"state mismatch at step {t}"
        tot += r
    assert (env.idx == ref.idx()).all() and (env.step == 72).all()
    assert not env.finished(s2).any()
    env.close()


def test_reset_variants_and_seeded_draws():
    S = _S()
    T = S.tables
    tabs = [T.synthetic_table("train", c) for c in (98, 4)]
    cfgs, profs = zip(*[_profile_cfg(S, c, r0, t.shape[0]) for c, r0, t in zip((98, 4), (0, 4320), tabs)])
    n = 2048
    co = (np.arange(n) % 2).astype(np.uint16)
    env = S.ShemsBatch(n, 72, tabs, list(cfgs), co)
    ref = oracle_c.Batch(n, 72, tabs, list(profs), co, co)
    # rng == -1
    env.reset_(-1)
    ref.reset(True)
    assert (env.idx == 1).all() and (U.bits32(env.state) == U.bits32(ref.state())).all()
    assert env.state[0, 0] == np.float32(3.375) and env.state[1, 0] == np.float32(0.5 * float(np.float32(11) * np.float32(0.9)))
    # seeded: draws must equal the NumPy Philox, resolution must equal the oracle loop
    seed, ep = 123, 7
    env.reset_(seed, episode=ep)
    i0 = np.empty(n, np.int32); s0 = np.empty(n, np.float32)
    for c in (0, 1):
        a, b = philox_np.reset_draws(seed, ep, n, tabs[c].shape[0], 72, cfgs[c].soc_max)
        i0[co == c], s0[co == c] = a[co == c], b[co == c]
    ref.reset(False, i0, s0)
    assert (env.idx == ref.idx()).all()
    assert (U.bits32(env.state) == U.bits32(ref.state())).all()
    assert len(np.unique(env.idx)) > n // 4                    # draws are spread
    # every possible first draw resolves like the reference loop (extension loop on the device)
    m = tabs[0].shape[0] - 72
    env2 = S.ShemsBatch(m, 72, [tabs[0]], [cfgs[0]])
    env2.reset_(0, idx0=np.arange(1, m + 1, dtype=np.int32), soc_b0=np.zeros(m, np.float32))
    assert (env2.idx == T.episode_start_table(tabs[0], 72)).all()
    env.close(); env2.close()


def test_config1_rule_based_episode_single_env(tmp_path):
    """BASELINE config 1 through the N = 1 drop-in wrapper: Shems(maxsteps, path), reset!(rng = -1),
    72 x { a = action(env, track); step!(env, s, a, track = -0.5) } against the committed golden."""
    S = _S()
    g = np.load(os.path.join(GOLD, "oracle_golden.npz"))
    path = str(tmp_path / "Charger98_all_train_fix.csv")
    S.tables.save_csv(path, S.tables.synthetic_table("train", 98))
    env = S.Shems(72, path, charger_id=98)
    assert len(env.state) == 9 and len(env.a) == 2 and env.idx == 1 and env.step == 0
    S.reset_(env, rng=-1)
    total = 0.0
    for t in range(72):
        s = env.state.copy()
        a = S.action(env, -0.5)
        r, s2, results = S.step_(env, s, a, track=-0.5)
        assert results.shape == (1, 23)
        assert (U.bits64(results[0]) == U.bits64(g["rule_results"][t])).all(), f"step {t}"
        total += r
        assert S.finished(env, s2) is False
    assert total == float(g["rule_total"]) and env.idx == 73 and env.step == 72


def test_mixed_profiles_and_weight_sweep():
    """BASELINE config 5 shape: 10 charger profiles x discomfort-weight sweep, one table per profile."""
    S = _S()
    T = S.tables
    ids = U.CHARGER_IDS
    tabs = [T.synthetic_table("train", c, nrow=600) for c in ids]
    sweep = [(0.01, 2.0), (0.04, 2.0), (0.1, 2.0), (0.01, 1.0), (0.04, 1.0), (0.1, 1.0)]
    cfgs, profs, tab_of = [], [], []
    for p, c in enumerate(ids):
        for (w, pot) in sweep:
            k, o = _profile_cfg(S, c, 600 * p, 600, w, pot)
            cfgs.append(k); profs.append(o); tab_of.append(p)
    n = 3000
    co = (np.arange(n) % len(cfgs)).astype(np.uint16)
    env = S.ShemsBatch(n, 72, tabs, cfgs, co)
    ref = oracle_c.Batch(n, 72, tabs, profs, np.asarray(tab_of)[co], co)
    rng = np.random.default_rng(11)
    idx0 = rng.integers(1, 600 - 72 + 1, n).astype(np.int32)
    soc0 = (rng.random(n) * 6.0).astype(np.float32)
    env.reset_(0, idx0=idx0, soc_b0=soc0)
    ref.reset(False, idx0, soc0)
    for t in range(40):
        a = rng.random((n, 2)).astype(np.float32)
        r, s2 = env.step_(None, a)
        _, r_ref, o_ref, _ = ref.step(a, 0)
        assert (U.bits64(r) == U.bits64(r_ref)).all() and (U.bits32(s2) == U.bits32(o_ref)).all()
    env.close()


def test_stepping_past_the_table_is_a_bounds_error():
    S = _S()
    tab = np.zeros((3, 8), np.float32)
    tab[:, 0] = -1
    tab[:, 1] = 1
    env = S.ShemsBatch(5, 1, [tab], [S.make_config(98, 0, 3)])
    env.reset_(-1)
    a = np.zeros((5, 2), np.float32)
    env.step_(None, a); env.step_(None, a)
    before = env.state.copy()
    with pytest.raises(S.BoundsError):
        env.step_(None, a)                                       # idx + 1 = 4 > nrow = 3
    assert (env.idx == 3).all() and (env.step == 2).all() and (env.state == before).all()
    with pytest.raises(S.ShemsError):
        S.ShemsBatch(5, 3, [tab], [S.make_config(98, 0, 3)])     # nrow <= maxsteps
    with pytest.raises(S.ShemsError):
        S.ShemsBatch(5, 1, [tab], [S.make_config(98, 2, 3)])     # config addresses rows past the upload
    env.close()


def test_rollout_kernel_rule_and_random_with_replay_ring():
    torch = pytest.importorskip("torch")
    S = _S()
    from importlib import import_module
    ReplayRing = import_module(U.PKG_NAME + ".replay").ReplayRing
    T = S.tables
    tab = T.synthetic_table("train", 98)
    cfg, prof = _profile_cfg(S, 98, 0, tab.shape[0])
    n = 777                                                     # ragged: not a multiple of 256
    env = S.ShemsBatch(n, 72, [tab], [cfg]).use_torch_stream()
    # rule-based, whole episode in one launch
    env.reset_(-1)
    ret = env.rollout("rule", 72).cpu().numpy()
    g = np.load(os.path.join(GOLD, "oracle_golden.npz"))
    assert (ret == float(g["rule_total"])).all() and (env.idx == 73).all() and (env.step == 72).all()
    # random policy + ring, vs the oracle driven with the same Philox actions
    seed = 99
    env.reset_(5, episode=1)
    ref = oracle_c.Batch(n, 72, tab, prof)
    ref.set_state(env.state, env.idx)
    ring = ReplayRing(n * 72 + 10)
    ret = env.rollout("random", 72, seed=seed, ring=ring).cpu().numpy()
    tot = np.zeros(n)
    S_, A_, R_, S2_ = (t.cpu().numpy() for t in (ring.s, ring.a, ring.r, ring.s2))
    for t in range(72):
        raw = philox_np.random_actions(seed, t, n)
        pre = ref.state()
        _, r, o, _ = ref.step(oracle_c.scale_action(raw), 0)
        tot += r
        slots = np.arange(n) * 72 + t
        assert (U.bits32(S_[slots]) == U.bits32(pre)).all() and (U.bits32(S2_[slots]) == U.bits32(o)).all()
        assert (U.bits32(A_[slots]) == U.bits32(raw)).all()
        assert (U.bits32(R_[slots]) == U.bits32(r.astype(np.float32))).all()
    assert (U.bits64(ret) == U.bits64(tot)).all()
    assert ring.pushed == n * 72 and not ring.done.any().item()
    assert (U.bits32(env.state) == U.bits32(ref.state())).all()
    env.close()


def test_full_size_65536_envs_properties_and_sampled_parity():
    """BASELINE config 3 size.  Size-independent properties for all envs + oracle parity on a sample."""
    torch = pytest.importorskip("torch")
    S = _S()
    T = S.tables
    n = 65536
    tab = T.synthetic_table("train", 98)
    cfg, prof = _profile_cfg(S, 98, 0, tab.shape[0])
    env = S.ShemsBatch(n, 72, [tab], [cfg]).use_torch_stream()
    env.reset_(123, episode=0)
    idx_start = env.idx.copy()
    st0 = env.state.copy()
    assert ((idx_start >= 1) & (idx_start <= tab.shape[0] - 72)).all()
    assert (U.bits32(st0[:, 2:]) == U.bits32(tab[idx_start - 1][:, [0, 2, 3, 4, 5, 6, 7]])).all()
    sample = np.random.default_rng(0).choice(n, 512, replace=False)
    ref = oracle_c.Batch(512, 72, tab, prof)
    ref.set_state(st0[sample], idx_start[sample])
    act = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    rew = torch.empty(n, dtype=torch.float64, device="cuda")
    blk = torch.empty((n + 255) // 256, dtype=torch.float64, device="cuda")
    total = np.zeros(n)
    for t in range(72):
        a = philox_np.random_actions(7, t, n)
        a = oracle_c.scale_action(a[sample]) if False else ((a.astype(np.float64) + 1.0) * 0.5).astype(np.float32)
        act.copy_(torch.from_numpy(a))
        env.step_dev(act, 0, rewards=rew, block_reward=blk)
        r = rew.cpu().numpy()
        total += r
        # wavefront/block reduction equals the sum of the per-env rewards of that workgroup
        np.testing.assert_allclose(blk.cpu().numpy(), r.reshape(-1, 256).sum(1), rtol=1e-12, atol=1e-12)
        _, r_ref, o_ref, _ = ref.step(a[sample], 0)
        assert (U.bits64(r[sample]) == U.bits64(r_ref)).all()
    env.check_error()
    st = env.state
    assert (env.idx == idx_start + 72).all() and (env.step == 72).all()          # integer bookkeeping
    assert (st[:, 2] == tab[idx_start + 72 - 1, 0]).all()                          # c_ev = h_countdown[idx]
    assert (st[:, 0] >= 0).all() and (st[:, 0] <= np.float32(6.75) * (1 + 1e-6)).all()
    assert (st[:, 1] <= 1.0 + 1e-6).all() and np.isfinite(total).all()
    assert (U.bits32(st[sample]) == U.bits32(ref.state())).all()
    env.close()


def test_four_million_envs_sampled_parity():
    """64 x BASELINE config 3: 4 194 304 households in one launch (the size at which the env-only kernel reaches its HBM-bound regime,
    profiles/r01_env_scaling.txt).  Index bookkeeping for every env, oracle parity bit for bit on a sample spread over the whole range
    (first, last and random envs), the per-workgroup reward sums."""
    torch = pytest.importorskip("torch")
    S = _S()
    n = 1 << 22
    tab = S.tables.synthetic_table("train", 98)
    cfg, prof = _profile_cfg(S, 98, 0, tab.shape[0])
    env = S.ShemsBatch(n, 72, [tab], [cfg]).use_torch_stream()
    env.reset_(321, episode=2)
    idx_start = env.idx.copy()
    st0 = env.state.copy()
    assert ((idx_start >= 1) & (idx_start <= tab.shape[0] - 72)).all()
    assert len(np.unique(idx_start)) > 2000 and st0[:, 0].std() > 1.0                # the draws differ from env to env (the start resolver, LU1:227-246, admits ~2 600 of the 4 248 rows)
    sample = np.unique(np.concatenate([[0, 1, n - 2, n - 1], np.random.default_rng(1).choice(n, 1020, replace=False)]))
    ref = oracle_c.Batch(len(sample), 72, tab, prof)
    ref.set_state(st0[sample], idx_start[sample])
    act = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    rew = torch.empty(n, dtype=torch.float64, device="cuda")
    blk = torch.empty((n + 255) // 256, dtype=torch.float64, device="cuda")
    for t in range(5):
        a = ((philox_np.random_actions(9, t, n).astype(np.float64) + 1.0) * 0.5).astype(np.float32)
        act.copy_(torch.from_numpy(a))
        env.step_dev(act, 0, rewards=rew, block_reward=blk)
        r = rew.cpu().numpy()
        np.testing.assert_allclose(blk.cpu().numpy(), r.reshape(-1, 256).sum(1), rtol=1e-12, atol=1e-12)
        _, r_ref, o_ref, _ = ref.step(a[sample], 0)
        assert (U.bits64(r[sample]) == U.bits64(r_ref)).all()
    env.check_error()
    assert (env.idx == idx_start + 5).all() and (env.step == 5).all()
    assert (U.bits32(env.state[sample]) == U.bits32(ref.state())).all()
    env.close()


def test_real_series_mixed_profiles_reset_and_steps_bit_exact():
    """BASELINE config 5 on the REAL exogenous series (tables.real_series: Chargers 01/03/04/05/08/09 train from the reference's MPC
    result files, synthetic for 02/06/07/98): seeded reset (Philox draws + the LU1:227-246 extension loop on real transaction
    patterns, every first draw of every real table) and 72 steps of the per-env-config batch, bit for bit against the oracle."""
    S = _S()
    T = S.tables
    tabs, cfgs, co = S.mixed_profile_setup(6000)
    assert [t.shape[0] for t in tabs] == [4319, 4320, 4319, 4319, 4319, 4320, 4320, 4319, 4319, 4320]
    sweep = [(0.01, 2.0), (0.04, 2.0), (0.1, 2.0), (0.01, 1.0), (0.04, 1.0), (0.1, 1.0)]
    profs = [oracle_c.profile(c, w, pot) for c in U.CHARGER_IDS for (w, pot) in sweep]
    tab_of = np.repeat(np.arange(10), 6)
    n = 6000
    env = S.ShemsBatch(n, 72, tabs, cfgs, co)
    ref = oracle_c.Batch(n, 72, tabs, profs, tab_of[co], co)
    seed, ep = 1231, 3
    env.reset_(seed, episode=ep)
    i0 = np.empty(n, np.int32); s0 = np.empty(n, np.float32)
    for k, cfg in enumerate(cfgs):
        a, b = philox_np.reset_draws(seed, ep, n, tabs[tab_of[k]].shape[0], 72, cfg.soc_max)
        i0[co == k], s0[co == k] = a[co == k], b[co == k]
    ref.reset(False, i0, s0)
    assert (env.idx == ref.idx()).all() and (U.bits32(env.state) == U.bits32(ref.state())).all()
    rng = np.random.default_rng(5)
    for t in range(72):
        a = rng.random((n, 2)).astype(np.float32)
        r, s2 = env.step_(None, a)
        rc, r_ref, o_ref, _ = ref.step(a, 0)
        assert rc == 0 and (U.bits64(r) == U.bits64(r_ref)).all() and (U.bits32(s2) == U.bits32(o_ref)).all()
    assert (env.idx == ref.idx()).all() and (env.step == 72).all()
    env.close()
    # every possible first draw of every real train table resolves like the reference loop (device extension loop)
    for c in (1, 3, 4, 5, 8, 9):
        tab = T.real_series(c, "train")
        m = tab.shape[0] - 72
        e2 = S.ShemsBatch(m, 72, [tab], [S.make_config(c, 0, tab.shape[0])])
        e2.reset_(0, idx0=np.arange(1, m + 1, dtype=np.int32), soc_b0=np.zeros(m, np.float32))
        assert (e2.idx == T.episode_start_table(tab, 72)).all(), c
        e2.close()
