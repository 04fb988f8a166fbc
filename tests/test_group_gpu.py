"""GPU tests of learner groups (the thesis protocol of many independent seeds / chargers): L learners
advanced by the same launches must leave, per learner, exactly the bytes the single-learner entry points leave when they are run
on that learner's buffers -- networks, targets, ADAM moments, gradients, workspace, losses and replay ring (one slab each)."""
import importlib

import numpy as np
import pytest

import util as U
import ddpg_oracle as DO

pytestmark = pytest.mark.gpu


def _mods():
    torch = pytest.importorskip("torch")
    S = U.pkg()
    D = importlib.import_module(U.PKG_NAME + ".ddpg")
    G = importlib.import_module(U.PKG_NAME + ".group")
    return torch, S, D, G


def _setup(L=3, E=256, cap=2400, mixed=False):
    torch, S, D, G = _mods()
    if mixed:                                       # the thesis grid: learner l trains on charger profile l mod 10 (ids 1-9, 98; LU1:47-58)
        ids = (1, 2, 3, 4, 5, 6, 7, 8, 9, 98)
        tabs = [S.tables.synthetic_table("train", c) for c in ids]
        row0 = np.cumsum([0] + [t.shape[0] for t in tabs])
        cfgs = [S.make_config(c, row0[k], tabs[k].shape[0]) for k, c in enumerate(ids)]
        co = ((np.arange(L * E) // E) % len(ids)).astype(np.uint16)
        env = S.ShemsBatch(L * E, 72, tabs, cfgs, co).use_torch_stream()
    else:
        tab = S.tables.synthetic_table("train", 98)
        env = S.ShemsBatch(L * E, 72, [tab], [S.make_config(98, 0, tab.shape[0])]).use_torch_stream()
    grp = G.LearnerGroup(L, E, seed=21, rng_seed=77, capacity=cap)
    grp.populate_memory(env, seed=5)
    grp.min_max_buffer()
    env.reset_(9, episode=1)
    return torch, S, D, G, env, grp


def test_learners_are_independent_and_populated():
    torch, S, D, G, env, grp = _setup()
    assert all(len(r) == 2400 for r in grp.rings)
    assert not torch.equal(grp.learners[0].actor, grp.learners[1].actor)          # seed + l initialisation
    assert not torch.equal(grp.rings[0].s, grp.rings[1].s)                        # each learner filled its ring from its own envs
    for l, (ag, ring) in enumerate(zip(grp.learners, grp.rings)):                  # group min_max == the single-learner launch
        ref = D.Agent(seed=1)
        ref.rng_seed = ag.rng_seed
        ref.min_max_buffer(ring)
        assert torch.equal(ref.s_min, ag.s_min) and torch.equal(ref.s_max, ag.s_max)
        assert float((ag.s_max - ag.s_min).max()) > 0.5


# 8 learners: the grouped update switches to its wider forward tile; (40, 128, mixed): the thesis protocol's shape -- 40 seeds, each on one
# of the 10 charger profiles (RL-SHEMS_bs_scheduler_1179_08_on_01-98.sh:67-87 runs 40 seeds x 10 chargers = 400 learners; bench.py
# --mode group --learners 400 --envs 51200 --mixed is that width, profiles/r04_group400_bench.json)
@pytest.mark.parametrize("L,E,mixed", [(3, 256, False), (8, 128, False), (40, 128, True)])
def test_group_step_and_update_match_single_learner_calls_bitwise(L, E, mixed):
    torch, S, D, G, env, grp = _setup(L=L, E=E, mixed=mixed)
    L, E, n = grp.count, grp.envs_per_learner, grp.n_envs
    snap = grp.slab.clone()
    st0, idx0, step0 = env.state, env.idx, env.step
    host = [(r.pushed, list(a.bp_actor), list(a.bp_critic), a.updates) for r, a in zip(grp.rings, grp.learners)]
    pos = grp.rings[0].pos
    a_g = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    ret_g = torch.zeros(n, dtype=torch.float64, device="cuda")
    for rep in range(2):                                                   # two rounds: the second uses advanced beta powers
        grp.act_step(env, train=False, tick=3 + rep, a_out=a_g, returns_acc=ret_g, window=(grp.rings[0].pos, 40, 7 + rep))
        grp.replay()
    torch.cuda.synchronize()
    slab_g, state_g, a_gh, ret_gh = grp.slab.clone(), env.state, a_g.cpu().numpy(), ret_g.cpu().numpy()
    assert not torch.equal(slab_g, snap)
    # the same work through the single-learner API on each learner's views
    grp.slab.copy_(snap)
    env.state, env.idx, env.step = st0, idx0, step0
    for (r, a), (pushed, bpa, bpc, upd) in zip(zip(grp.rings, grp.learners), host):
        r.pushed, a.bp_actor, a.bp_critic, a.updates = pushed, bpa, bpc, upd
    a_s = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    ret_s = torch.zeros(n, dtype=torch.float64, device="cuda")
    for l, (ag, ring) in enumerate(zip(grp.learners, grp.rings)):
        sub = env.slice(l * E, E)
        for rep in range(2):
            ag.act_step(sub, train=False, tick=3 + rep, a_out=a_s[l * E:(l + 1) * E], returns_acc=ret_s[l * E:(l + 1) * E], ring=ring,
                        window=D.RingWindow(ring.pos, 40, 7 + rep))
            ring.pushed += 40
            ag.replay(ring, tick=rep)
    torch.cuda.synchronize()
    assert np.array_equal(a_s.cpu().numpy(), a_gh) and np.array_equal(ret_s.cpu().numpy(), ret_gh)
    assert np.array_equal(env.state, state_g)
    for name, (off, cnt) in grp.layout.items():
        # bit patterns, not float values: the workspace keeps int32 ring slots (-1 for the 8 pad columns reads as NaN)
        if name == "ws":
            cnt -= 96        # the workspace ends with 96 bookkeeping words (timeout count of the pipelined loop's device-side waits): not results
        assert torch.equal(grp.slab[:, off:off + cnt].contiguous().view(torch.int32), slab_g[:, off:off + cnt].contiguous().view(torch.int32)), name
    env.check_error()


def test_group_exploration_noise_is_keyed_by_the_global_env_index():
    torch, S, D, G, env, grp = _setup()
    L, E, n = grp.count, grp.envs_per_learner, grp.n_envs
    obs = env.state
    a = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    grp.act_step(env, train=True, tick=5, a_out=a)
    a = a.cpu().numpy()
    zn = DO.gauss_noise(grp.rng_seed, 5, n)
    for l, ag in enumerate(grp.learners):
        sl = slice(l * E, (l + 1) * E)
        clean = DO.act(ag.actor.cpu().numpy(), obs[sl], ag.s_min.cpu().numpy(), ag.s_max.cpu().numpy(), False, dtype=np.float64)
        ref = np.clip(clean + np.float32(0.1) * zn[sl], -1, 1)
        assert np.abs(a[sl] - ref).max() < 2e-5


def test_group_argument_checks():
    torch, S, D, G, env, grp = _setup()
    with pytest.raises(ValueError):
        G.LearnerGroup(2, 100)
    bad = G.LearnerGroup(2, 256, capacity=720)
    with pytest.raises(ValueError):
        bad.act_step(env)                                                   # 768 envs != 2 x 256
    g = grp.struct()
    g.envs_per_learner = 128                                                # count x envs_per_learner != n_envs
    import ctypes as C
    v = env.view()
    p = grp.learners[0]._act_params(False, 0)
    rc = grp.L.shems_act_step_group_dev(C.byref(v), C.byref(p), C.byref(g), None, None, None, None, grp._stream())
    assert rc != 0
