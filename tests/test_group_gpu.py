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


def _setup(L=3, E=256, cap=2400, mixed=False, form="latency", tiled=None):
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
    grp = G.LearnerGroup(L, E, seed=21, rng_seed=77, capacity=cap, form=form, tiled=tiled)
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
# (12, 32) / (6, 96): env blocks that are multiples of 32 but not of 128 / 64 (round 6: the thesis-exact protocol needs ONE household per learner;
# 32 is the smallest tile) -- the dispatcher must pick tiles that never straddle two learners
@pytest.mark.parametrize("L,E,mixed", [(3, 256, False), (8, 128, False), (40, 128, True), (12, 32, False), (6, 96, False)])
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
        grp.act_step(env, train=False, tick=3 + rep, a_out=a_g, returns_acc=ret_g, window=(grp.rings[0].pos, min(40, E), 7 + rep))
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
                        window=D.RingWindow(ring.pos, min(40, E), 7 + rep))
            ring.pushed += min(40, E)
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


def test_default_form_follows_the_group_width():
    torch, S, D, G = _mods()
    assert G.LearnerGroup(3, 128, capacity=720).form == "latency"
    assert G.LearnerGroup(G.LearnerGroup.TP_MIN_LEARNERS, 128, capacity=720).form == "throughput"
    with pytest.raises(ValueError):
        G.LearnerGroup(3, 128, capacity=720, form="fast")


# The THROUGHPUT form of the grouped update (csrc/shems_gupd.hip: eight launches for the whole group, plain back-propagation on small
# tiles) sums in another order than the single-learner kernels, so it is held to what those are held to in tests/test_ddpg_gpu.py
# instead of to their bits: every Flux.params block of every learner's two gradients within BLOCK_TOL of ITS max-abs of the float64
# evaluation (DDPG.jl:121-145 restated in oracle/ddpg_oracle.py), ADAM / soft update element-wise at 1e-7 from the kernel's own
# gradient, targets and losses at the latency form's tolerances.  batch 120 (8 pad columns) and 128 (none); two updates in a row
# (the second with advanced beta powers, non-zero moments and targets that have moved).
# A relu whose float64 pre-activation lies within fp32 accumulation error of zero is on in one evaluation order and off in another:
# the gradient then differs by that unit's whole contribution (seen in this very test: unit 496 of an actor at 1.5e-8 for one sample
# moved gb2[496] by 1.2e-3 of the block's max while every other element agreed to 4e-7).  Nothing is forgiven for that: when a block
# comparison fails, the float64 evaluation is REPEATED with every such unit (|z| < TIE for some sample) decided the other way -- its bias
# moved by -+ 4 TIE, which changes no other relu decision and no gradient element by more than ~1e-6 of its block -- and the kernel's
# gradient must meet BLOCK_TOL against one of those evaluations; with no such unit the failure stands.  Seeds, shapes and summation orders
# are fixed, so WHICH comparisons needed the second evaluation is deterministic: each case names its exact list (EXPECTED_TIES) and the
# test asserts equality with it -- an unexpected entry fails, and so does an expected one that no longer occurs.
TIE = 1e-7                      # ~5 sigma of the fp32 accumulation error of a layer-2 pre-activation (K = 250 terms of ~3e-2)
# (learners, batch) -> [(network, learner, tick)] of the committed seeds
EXPECTED_TIES = {(5, 120): [], (3, 128): [("actor", 0, 4)], (2, 17): [], (11, 120): [], (48, 120): [], (400, 120): []}


def _tied_units(p, x, in_dim, out_dim):
    """[(index of the unit's bias in the flat parameter vector, sign of its closest-to-zero float64 pre-activation)]"""
    _, (_, z1, _, z2, _, _) = DO.mlp_forward(p, x, in_dim, out_dim, out_dim == 2, keep=True, dtype=np.float64)
    out = []
    for z, off in ((z1, in_dim * 250), (z2, in_dim * 250 + 250 + 250 * 500)):
        for u in np.unique(np.where(np.abs(z) < TIE)[1]):
            col = z[:, u]
            out.append((off + int(u), float(np.sign(col[np.argmin(np.abs(col))]) or 1.0)))
    return out


def _assert_blocks_or_the_other_relu_decision(TD, g, evaluate, nets, in_dim, out_dim, what):
    """evaluate(params by net name) -> float64 gradient.  nets: {name: (float32 params, inputs, in_dim, out_dim)} of the networks whose relus
    the gradient passes through.  Returns (per-block errors, tie used?)."""
    base = {k: v[0] for k, v in nets.items()}
    try:
        return TD._assert_blocks(g, evaluate(base), in_dim, out_dim, what), False
    except AssertionError:
        tied = [(name, i, sg) for name, (p, x, i_d, o_d) in nets.items() for i, sg in _tied_units(p, x, i_d, o_d)]
        if not tied or len(tied) > 3:
            raise
        flipped = {k: v.astype(np.float64) for k, v in base.items()}
        for name, i, sg in tied:                    # every tied unit decided the other way
            flipped[name][i] -= sg * 4 * TIE
        return TD._assert_blocks(g, evaluate(flipped), in_dim, out_dim, what + " (tied relus decided the other way)"), True


def _throughput_vs_float64(L, batch, check=None, ticks=(3, 4), tiled=None):
    """Two grouped updates in a row of an L-learner group (throughput form, store_grad on); the learners in `check` (default: all) are
    held, per Flux.params block, to the float64 evaluation of DDPG.jl:121-145.  Returns (worst per-block errors, ties set aside, the
    group).  tiled: the working layout of the layer-2 state (None = the default, tiled); results are read in Flux order (flux_())."""
    import test_ddpg_gpu as TD
    ties = []
    torch, S, D, G, env, grp = _setup(L=L, E=128, cap=2400, form="throughput", tiled=tiled)
    assert grp.tiled == (True if tiled is None else tiled)
    grp.store_grad = True
    rng = np.random.default_rng(5)
    check = list(range(L)) if check is None else sorted(set(int(l) for l in check))
    host = {}
    for l, ag in enumerate(grp.learners):
        ag.batch = batch
        pa, pc = ag.actor.cpu().numpy().copy(), ag.critic.cpu().numpy().copy()
        pa[128000:129000] *= 30.0                 # lift the 3e-3 heads so every gradient path is exercised (as test_ddpg_gpu._setup)
        pc[128250:128750] *= 30.0
        pa[2250:2500] = rng.normal(0, 0.05, 250); pc[2750:3000] = rng.normal(0, 0.05, 250)
        ag.set_params(actor=pa, critic=pc)
        ring = grp.rings[l]
        ring.done.copy_(torch.from_numpy((rng.random(ring.capacity) < 0.05).astype(np.uint8)))   # the formula's (1 - done) term
        if l in check:
            host[l] = dict(pa=pa, pc=pc, pat=pa.copy(), pct=pc.copy(), s=ring.s.cpu().numpy(), a=ring.a.cpu().numpy(), r=ring.r.cpu().numpy(),
                           s2=ring.s2.cpu().numpy(), done=ring.done.cpu().numpy(), s_min=ag.s_min.cpu().numpy(), s_max=ag.s_max.cpu().numpy(),
                           opt_c=DO.Adam(len(pc), DO.ETA_CRIT), opt_a=DO.Adam(len(pa), DO.ETA_ACT))
    worst = {}
    for tick in ticks:
        grp.replay(tick=tick)
        grp.flux_()
        torch.cuda.synchronize()
        for l in check:
            ag, h = grp.learners[l], host[l]
            idx = DO.sample_indices(grp.rng_seed + l, tick, batch, len(grp.rings[l]))
            Lr = DO.Learner(h["pa"], h["pc"], h["s_min"], h["s_max"])
            Lr.actor_t, Lr.critic_t = h["pat"], h["pct"]
            s, a, r, s2, done = (h[k][idx] for k in ("s", "a", "r", "s2", "done"))
            y = Lr.targets(r, s2, done.astype(bool))
            gc64, lc64 = Lr.critic_grad(s, a, y, dtype=np.float64)
            gc = ag.grad_critic.cpu().numpy()
            sn = DO.normalize(s, h["s_min"], h["s_max"])

            def crit_eval(P):
                Lq = DO.Learner(h["pa"], P["critic"], h["s_min"], h["s_max"])
                return Lq.critic_grad(s, a, y, dtype=np.float64)[0]
            e, tie = _assert_blocks_or_the_other_relu_decision(TD, gc, crit_eval, {"critic": (h["pc"], np.concatenate([sn, a], 1), 11, 1)}, 11, 1,
                                                               f"critic gradient of learner {l}, tick {tick}: throughput form vs float64")
            if tie:
                ties.append(("critic", l, tick))
            losses = ag.losses.cpu().numpy()
            assert abs(losses[0] - lc64) < 1e-4 * max(1.0, abs(lc64)), (l, tick)
            pc1 = h["opt_c"].step(h["pc"], gc)                                  # ADAM + soft update from the kernel's own gradient
            crit = ag.critic.cpu().numpy()
            np.testing.assert_allclose(crit, pc1, rtol=0, atol=1e-7)
            pct1 = DO.soft_update(h["pct"], crit)
            np.testing.assert_allclose(ag.critic_t.cpu().numpy(), pct1, rtol=0, atol=1e-7)
            np.testing.assert_allclose(ag.m_critic.cpu().numpy(), h["opt_c"].m, rtol=1e-6, atol=1e-12)
            np.testing.assert_allclose(ag.v_critic.cpu().numpy(), h["opt_c"].v, rtol=1e-6, atol=1e-15)
            Lr.critic = crit                                                    # the actor gradient goes through the UPDATED critic
            ga64, la64 = Lr.actor_grad(s, dtype=np.float64)
            ga = ag.grad_actor.cpu().numpy()
            a_pi = DO.actor_forward(h["pa"], sn, dtype=np.float64)

            def act_eval(P):
                Lq = DO.Learner(P["actor"], P["critic"], h["s_min"], h["s_max"])
                return Lq.actor_grad(s, dtype=np.float64)[0]
            e2, tie = _assert_blocks_or_the_other_relu_decision(TD, ga, act_eval, {"actor": (h["pa"], sn, 9, 2), "critic": (crit, np.concatenate([sn, a_pi], 1), 11, 1)},
                                                                9, 2, f"actor gradient of learner {l}, tick {tick}: throughput form vs float64")
            if tie:
                ties.append(("actor", l, tick))
            assert abs(losses[1] - la64) < 1e-4 * max(1.0, abs(la64)), (l, tick)
            pa1 = h["opt_a"].step(h["pa"], ga)
            act = ag.actor.cpu().numpy()
            np.testing.assert_allclose(act, pa1, rtol=0, atol=1e-7)
            pat1 = DO.soft_update(h["pat"], act)
            np.testing.assert_allclose(ag.actor_t.cpu().numpy(), pat1, rtol=0, atol=1e-7)
            np.testing.assert_allclose(ag.m_actor.cpu().numpy(), h["opt_a"].m, rtol=1e-6, atol=1e-12)
            np.testing.assert_allclose(ag.v_actor.cpu().numpy(), h["opt_a"].v, rtol=1e-6, atol=1e-15)
            for k, v in list(e.items()) + [("a_" + k, v) for k, v in e2.items()]:
                worst[k] = max(worst.get(k, 0.0), v)
            # carry the DEVICE state forward: the next update starts from exactly these bytes
            h["pa"], h["pc"], h["pat"], h["pct"] = act, crit, ag.actor_t.cpu().numpy(), ag.critic_t.cpu().numpy()
            h["opt_c"].m, h["opt_c"].v = ag.m_critic.cpu().numpy().astype(h["opt_c"].m.dtype), ag.v_critic.cpu().numpy().astype(h["opt_c"].v.dtype)
            h["opt_a"].m, h["opt_a"].v = ag.m_actor.cpu().numpy().astype(h["opt_a"].m.dtype), ag.v_actor.cpu().numpy().astype(h["opt_a"].v.dtype)
    # every learner of the group, checked or not: finite state after the two updates
    end = grp.layout["grad_actor"][0]
    assert bool(torch.isfinite(grp.slab[:, :end]).all())
    print(f"throughput form, {L} learners, batch {batch}: worst per-block gradient error (fraction of the block's max-abs):", worst,
          "comparisons that needed a tied relu decided the other way:", ties)
    return worst, ties, grp


@pytest.mark.parametrize("L,batch", [(5, 120), (3, 128), (2, 17), (11, 120), (48, 120)])      # < 48 learners: the narrow launch shapes; 48: the wide ones
def test_throughput_form_matches_float64_oracle_per_block(L, batch):
    worst, ties, grp = _throughput_vs_float64(L, batch)
    assert ties == EXPECTED_TIES[(L, batch)], ties
    # the same two updates on the Flux-order layout (round 5's form, SHEMS_GROUP_TILED=0): the tiled working layout changes where the
    # layer-2 state lives, not one operation on it -- every learner's networks, targets, moments and gradients bit for bit
    _, _, grp_f = _throughput_vs_float64(L, batch, check=(), tiled=False)
    torch = grp.torch
    for name in ("actor", "critic", "actor_t", "critic_t", "m_actor", "v_actor", "m_critic", "v_critic", "grad_actor", "grad_critic", "losses"):
        (o, n), (of, nf) = grp.layout[name], grp_f.layout[name]
        assert n == nf and torch.equal(grp.slab[:, o:o + n].contiguous().view(torch.int32), grp_f.slab[:, of:of + nf].contiguous().view(torch.int32)), name


def test_throughput_form_at_the_benched_width_400_learners():
    """The shape `bench.py --mode group --learners 400 --envs 51200` and tools/group_protocol_demo.py run -- 40 seeds x 10 chargers
    (RL-SHEMS_bs_scheduler_1179_08_on_01-98.sh:67-87) -- with its grid y = 400, the XCD remap of P3 / P6 over 50 groups of 8 learners and
    the wide launch shapes: first, second, middle, and the last two learners plus five drawn ones against the float64 oracle per block,
    batch 120, two updates in a row."""
    L = 400
    pick = [0, 1, 199, 398, 399] + [int(x) for x in np.random.default_rng(400).choice(np.arange(2, 398), 5, replace=False)]
    worst, ties, grp = _throughput_vs_float64(L, 120, check=pick)
    assert ties == EXPECTED_TIES[(L, 120)], ties


def test_one_remembered_transition_per_update_in_reference_push_order():
    """window_count = 1: the reference's update-to-data ratio.  episode! (DDPG.jl:186-242) remembers ONE transition per replay()
    (DDPG.jl:229-233), pushed as [s, a, r, s', done] (MPS:46-47) into the CircularBuffer; here each learner's ring receives exactly one
    transition per vector step -- household 0 of the learner's block, whose consecutive entries therefore chain (s' of entry t is s of
    entry t + 1), written at the ring's push position with the UNSCALED action -- every other slot of every ring keeps its bytes, and the
    update that follows samples DO.sample_indices(rng_seed + l, tick, 120, len(ring)) from the ring that already holds the new entry."""
    torch, S, D, G, env, grp = _setup(L=16, E=128, cap=2400, form="throughput")
    L, E, n = grp.count, grp.envs_per_learner, grp.n_envs
    assert grp.ring_window(72, 1) == (1, 0) and grp.ring_window(72, None) == (33, 0) and grp.ring_window(72, 128)[0] == 128
    with pytest.raises(ValueError):
        grp.ring_window(72, 129)
    a = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    TP_IDX = 2 * 12 * 128 + 2 * 128                 # csrc/shems_gupd.hip: [BP] int32 sampled ring slots behind the two input blocks, r, done
    prev_s2 = None
    pushed0 = grp.rings[0].pushed                   # populate_memory's count (whole rollouts: >= capacity)
    assert all(r.pushed == pushed0 for r in grp.rings) and all(len(r) == 2400 for r in grp.rings)
    for step in range(5):
        pre = env.state
        pos = grp.rings[0].pos
        assert all(r.pos == pos for r in grp.rings)
        before = [tuple(t.clone() for t in (r.s, r.a, r.r, r.s2, r.done)) for r in grp.rings]
        ret = torch.zeros(n, dtype=torch.float64, device="cuda")
        grp.tick = step
        wc, off = grp.ring_window(72, 1)
        grp.act_step(env, train=True, tick=step, a_out=a, returns_acc=ret, window=(pos, wc, off))
        torch.cuda.synchronize()
        post, ah, rh = env.state, a.cpu().numpy(), ret.cpu().numpy()
        for l, ring in enumerate(grp.rings):
            assert ring.pushed == pushed0 + step + 1                                            # exactly one per step
            e = l * E                                                                         # the learner's household 0
            assert (U.bits32(ring.s[pos].cpu().numpy()) == U.bits32(pre[e])).all()
            assert (U.bits32(ring.a[pos].cpu().numpy()) == U.bits32(ah[e])).all() and np.abs(ah[e]).max() <= 1.0      # as act() returned it, not scale_action's
            assert ring.r[pos].item() == np.float32(rh[e])                                   # r: Float64 -> Float32 at upload
            assert (U.bits32(ring.s2[pos].cpu().numpy()) == U.bits32(post[e])).all() and ring.done[pos].item() == 0
            for cur, old in zip((ring.s, ring.a, ring.r, ring.s2, ring.done), before[l]):    # nothing else moved
                keep = torch.ones(ring.capacity, dtype=torch.bool, device="cuda")
                keep[pos] = False
                assert torch.equal(cur[keep], old[keep])
            if prev_s2 is not None:                                                           # one household's trajectory
                assert (U.bits32(ring.s[pos].cpu().numpy()) == U.bits32(prev_s2[l])).all()
        prev_s2 = [ring.s2[pos].cpu().numpy().copy() for ring in grp.rings]
        grp.replay(tick=step)
        torch.cuda.synchronize()
        for l, (ag, ring) in enumerate(zip(grp.learners, grp.rings)):
            idx = DO.sample_indices(grp.rng_seed + l, step, 120, len(ring))
            got = ag.ws[TP_IDX:TP_IDX + 128].view(torch.int32).cpu().numpy()
            assert (got[:120] == idx).all() and (got[120:] == -1).all()
            XT = ag.ws[0:9 * 128].view(9, 128).cpu().numpy()
            want = DO.normalize(ring.s.cpu().numpy()[idx], ag.s_min.cpu().numpy(), ag.s_max.cpu().numpy())
            np.testing.assert_allclose(XT[:, :120].T, want, rtol=0, atol=1e-6)
    # episode_ drives the same mode: 72 steps = 72 entries per ring
    p0 = grp.rings[0].pushed
    grp.episode_(env, train=True, rng_ep=3, episode=2, window_count=1)
    assert all(r.pushed == p0 + 72 for r in grp.rings)
    env.check_error()


def test_throughput_and_latency_forms_agree_and_leave_no_gradient_unless_asked():
    """Same group, same minibatches: after one update the two forms' networks agree to ADAM's step size (the gradients differ in the
    last bits only, and ADAM normalises them), and the throughput form does not touch the gradient buffers unless store_grad is set."""
    torch, S, D, G, env, grp = _setup(L=4, E=128, cap=2400, form="throughput")
    snap = grp.slab.clone()
    grp.replay(tick=2)
    grp.flux_()                                    # (tiled working layout: the W2 ranges of the Flux-order blocks are made current)
    torch.cuda.synchronize()
    tp = grp.slab.clone()
    for name in ("grad_actor", "grad_critic"):
        off, cnt = grp.layout[name]
        assert torch.equal(tp[:, off:off + cnt], snap[:, off:off + cnt]), name
    grp.slab.copy_(snap)
    grp.flux_changed()                             # the Flux-order blocks were written behind the group's back
    for ag in grp.learners:
        ag.bp_critic, ag.bp_actor, ag.updates = [0.9, 0.999], [0.9, 0.999], 0
    grp.updates = 0
    grp.form = "latency"
    grp.replay(tick=2)
    torch.cuda.synchronize()
    for name, tol in (("critic", 2.1e-3), ("actor", 2.1e-4), ("critic_t", 2.1e-6), ("actor_t", 2.1e-7)):   # eta (1e-3 / 1e-4), x tau for the targets
        off, cnt = grp.layout[name]
        d = (tp[:, off:off + cnt] - grp.slab[:, off:off + cnt]).abs()
        assert float(d.max()) <= tol, (name, float(d.max()))
        assert float((d > tol * 1e-2).float().mean()) < 0.05, (name, float((d > tol * 1e-2).float().mean()))   # (a first step is +-eta: sign flips of ~0 gradients only)
    for name in ("losses",):
        off, cnt = grp.layout[name]
        assert torch.allclose(tp[:, off:off + cnt], grp.slab[:, off:off + cnt], rtol=1e-4, atol=1e-5)


def test_tiled_layout_round_trip_and_fused_step_reads_the_same_weights():
    """shems_group_w2_to_tiled / _to_flux are inverse on the W2 ranges and touch nothing else; pad rows / columns of the tiled regions are
    zero; the fused act/step launch reading every learner's actor W2 from its tiled region (the free-running k_act forms) leaves the bytes
    the Flux-order launch leaves: actions, env state, returns, ring."""
    import ctypes as C
    torch, S, D, G, env, grp = _setup(L=16, E=128, cap=2400, form="throughput")
    assert grp.tiled
    rng = np.random.default_rng(3)
    for name in ("actor_t", "critic_t", "m_actor", "v_actor", "m_critic", "v_critic"):       # distinct values in every array
        o, n = grp.layout[name]
        grp.slab[:, o:o + n] = torch.from_numpy(rng.normal(0, 0.05, (grp.count, n)).astype(np.float32)).cuda()
    grp.flux_changed()
    before = grp.slab.clone()
    assert grp._use_tiled() and grp._tiled_valid
    torch.cuda.synchronize()
    o, n = grp.layout["w2t_actor"]
    reg = grp.slab[:, o:o + n].view(grp.count, 4, 8, 4, 64, 64).cpu().numpy()                # [learner][kt][nt][m | v | p | target][row][col]
    for a, name in enumerate(("m_actor", "v_actor", "actor", "actor_t")):
        fo, _ = grp.layout[name]
        w2 = before[:, fo + 2500:fo + 2500 + 125000].view(grp.count, 250, 500).cpu().numpy()
        full = np.zeros((grp.count, 256, 512), np.float32)
        full[:, :250, :500] = w2
        want = full.reshape(grp.count, 4, 64, 8, 64).transpose(0, 1, 3, 2, 4)
        assert np.array_equal(reg[:, :, :, a], want), name
    # scribble over the Flux-order W2 ranges, then bring them back from the tiles: everything as before
    for name in ("actor", "critic", "actor_t", "critic_t", "m_actor", "v_actor", "m_critic", "v_critic"):
        fo, _ = grp.layout[name]
        w2o = 2500 if "actor" in name else 3000
        grp.slab[:, fo + w2o:fo + w2o + 125000] = -7.0
    grp._flux_valid = False
    grp.flux_()
    torch.cuda.synchronize()
    for name, (fo, n) in grp.layout.items():
        if not name.startswith("w2t_"):
            assert torch.equal(grp.slab[:, fo:fo + n].contiguous().view(torch.int32), before[:, fo:fo + n].contiguous().view(torch.int32)), name
    # fused step: tiled against Flux order
    n_envs = grp.n_envs
    outs = []
    st0, idx0, step0 = env.state, env.idx, env.step
    ring0 = [tuple(t.clone() for t in (r.s, r.a, r.r, r.s2, r.done)) for r in grp.rings]
    pushed0 = [r.pushed for r in grp.rings]
    for tiled in (True, False):
        env.state, env.idx, env.step = st0, idx0, step0
        for r, old, pu in zip(grp.rings, ring0, pushed0):
            for cur, o_ in zip((r.s, r.a, r.r, r.s2, r.done), old):
                cur.copy_(o_)
            r.pushed = pu
        grp.tiled = tiled
        a = torch.empty((n_envs, 2), dtype=torch.float32, device="cuda")
        ret = torch.zeros(n_envs, dtype=torch.float64, device="cuda")
        for t in range(3):
            grp.tick = t
            grp.act_step(env, train=True, tick=t, a_out=a, returns_acc=ret, window=(grp.rings[0].pos, *grp.ring_window(72, None)))
        torch.cuda.synchronize()
        outs.append((a.cpu().numpy(), ret.cpu().numpy(), env.state.copy(), [tuple(t.cpu().numpy() for t in (r.s, r.a, r.r, r.s2)) for r in grp.rings]))
    grp.tiled = True
    (a1, r1, s1, g1), (a2, r2, s2, g2) = outs
    assert np.array_equal(a1, a2) and np.array_equal(r1, r2) and np.array_equal(s1, s2)
    for x, y in zip(g1, g2):
        assert all(np.array_equal(u, v) for u, v in zip(x, y))
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
