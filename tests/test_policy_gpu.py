"""GPU tests of the fused actor kernel (fp32 MFMA) against the NumPy oracle (float64 reference of
the same network) and, for the fused step, against the C env oracle driven by the emitted actions.

Tolerance: the reference's act() runs Flux/CUBLAS fp32 GEMMs whose summation order is not pinned
("parity unpinned" for network arithmetic); the kernel's fp32 MFMA result must agree with a float64
evaluation of the same weights to 1e-5 absolute on actions in [-1, 1] (north_star: 1e-5 relative)."""
import importlib

import numpy as np
import pytest

import util as U
from util import oracle_c
import ddpg_oracle as DO

pytestmark = pytest.mark.gpu
ATOL = 1e-5


def _mods():
    torch = pytest.importorskip("torch")
    S = U.pkg()
    D = importlib.import_module(U.PKG_NAME + ".ddpg")
    return torch, S, D


def _rand_obs(rng, n):
    tab = U.tables_mod().synthetic_table("train", 98)
    rows = tab[rng.integers(0, tab.shape[0], n)]
    obs = np.empty((n, 9), np.float32)
    obs[:, 0] = rng.random(n) * 6.75
    obs[:, 1] = rows[:, 1]; obs[:, 2] = rows[:, 0]; obs[:, 3] = rows[:, 2]; obs[:, 4] = rows[:, 3]
    obs[:, 5] = rows[:, 4]; obs[:, 6] = rows[:, 5]; obs[:, 7] = rows[:, 6]; obs[:, 8] = rows[:, 7]
    return obs


def test_init_matches_oracle_and_layout():
    torch, S, D = _mods()
    for which, (i, o) in enumerate([(9, 2), (11, 1)]):
        a = D.init_params(1231, i, o, which)
        assert (a == DO.init_params(1231, i, o, which)).all() and a.size == DO.n_params(i, o)
        W1, b1, W2, b2, W3, b3 = DO.split(a, i, o)
        lim = np.sqrt(24.0 / (i + 250)) / 2
        assert abs(W1).max() <= lim * 1.0001 and abs(W1).max() > 0.9 * lim and not b1.any()
        assert abs(W3).max() <= 3e-3 + 1e-9 and W3.std() > 1e-3


@pytest.mark.parametrize("m", [1, 31, 120, 8192, 16384 + 77, 65536])
def test_actor_forward_matches_float64_reference(m):
    torch, S, D = _mods()
    rng = np.random.default_rng(m)
    ag = D.Agent(seed=1231)
    # make the last layer big enough that tanh is exercised over its range
    p = D.init_params(1231, 9, 2, 0)
    p[128000:129000] *= 40.0
    p[129000:] = [0.3, -0.2]
    p[2250:2500] = rng.normal(0, 0.1, 250)        # non-zero biases
    p[127500:128000] = rng.normal(0, 0.1, 500)
    ag.set_params(actor=p)
    obs = _rand_obs(rng, m)
    s_min, s_max = obs.min(0) - 0.01, obs.max(0) + 0.5
    s_max[5] = s_min[5]                            # constant feature (p_buy): denominator = 1f-8 path
    ag.set_norm(s_min, s_max)
    out = ag.act(torch.from_numpy(obs).cuda(), train=False).cpu().numpy()
    ref = DO.act(p, obs, s_min.astype(np.float32), s_max.astype(np.float32), False, dtype=np.float64)
    assert out.shape == (m, 2) and np.isfinite(out).all()
    assert np.abs(out - ref).max() < ATOL
    assert np.abs(out).max() > 0.3                # not a degenerate all-zero check
    # train=True adds the Philox/Box-Muller noise and clamps
    outn = ag.act(torch.from_numpy(obs).cuda(), train=True, tick=17).cpu().numpy()
    refn = DO.act(p, obs, s_min.astype(np.float32), s_max.astype(np.float32), True, seed=1231, tick=17, dtype=np.float64)
    assert np.abs(outn - refn).max() < 5e-6 + ATOL and outn.min() >= -1 and outn.max() <= 1
    assert np.abs(outn - out).max() > 0.05         # the noise is really there (its N(0,1) shape: test_ddpg_oracle.py)


# 64-, 32- and 128-env tiles (the first two ragged) with per-tile reward sums; (8192, .., False) / (4096, .., False): BASELINE config 4's shard
# and config 2 exactly as the training loop launches them -- NO per-tile sums, hence the two-workgroups-per-tile forms k_actg<1, 4, 2, 2> and
# k_actg<1, 4, 2, 3> (with sums the dispatcher takes the 8-wave form): the shapes the bench times meet the oracle directly.
@pytest.mark.parametrize("n,nsteps,sums", [(20000, 6, True), (2048 + 5, 3, True), (65536, 2, True), (8192, 2, False), (4096, 2, False),
                                           (6005, 2, False)])
def test_fused_act_step_equals_act_then_oracle_step_and_fills_ring(n, nsteps, sums):
    torch, S, D = _mods()
    if not sums:
        assert D.act_kernel_name(n) == ("shems::k_actg<1, 4, 2, 3>" if n <= 4096 else "shems::k_actg<1, 4, 2, 2>")
    ReplayRing = importlib.import_module(U.PKG_NAME + ".replay").ReplayRing
    T = S.tables
    tab = T.synthetic_table("train", 98)
    env = S.ShemsBatch(n, 72, [tab], [S.make_config(98, 0, tab.shape[0])]).use_torch_stream()
    ref = oracle_c.Batch(n, 72, tab, oracle_c.profile(98))
    ag = D.Agent(seed=77)
    p = D.init_params(77, 9, 2, 0)
    p[128000:129000] *= 60.0
    ag.set_params(actor=p)
    env.reset_(5, episode=0)
    st0 = env.state
    ag.set_norm(st0.min(0), st0.max(0))
    ref.set_state(st0, env.idx)
    ring = ReplayRing(5000)
    a_out = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    rew = torch.empty(n, dtype=torch.float64, device="cuda")
    rew32 = torch.empty(n, dtype=torch.float32, device="cuda")
    blk = torch.zeros(ag.act_step_blocks(n), dtype=torch.float64, device="cuda") if sums else None
    pos = 0
    for t in range(nsteps):
        pre = env.state
        win = D.RingWindow(pos % ring.capacity, 333, (t * 333) % n)
        ag.act_step(env, train=True, tick=t, a_out=a_out, rewards=rew, rewards_f32=rew32, block_reward=blk, ring=ring, window=win)
        env.check_error()
        a = a_out.cpu().numpy()
        # (1) the action is act() of the pre-step observation
        want = DO.act(p, pre, st0.min(0), st0.max(0), True, seed=77, tick=t, dtype=np.float64)
        assert np.abs(a - want).max() < 5e-6 + ATOL
        # (2) given that action, the transition is the oracle's, bit for bit
        rc, r_ref, o_ref, _ = ref.step(oracle_c.scale_action(a), 0)
        assert rc == 0
        r = rew.cpu().numpy()
        assert (U.bits64(r) == U.bits64(r_ref)).all() and (U.bits32(env.state) == U.bits32(o_ref)).all()
        assert (rew32.cpu().numpy() == r_ref.astype(np.float32)).all()
        if sums:
            assert abs(blk.sum().item() - r.sum()) < 1e-9 * max(1.0, abs(r).sum())
        # (3) the ring window
        rel = (np.arange(n) - (t * 333) % n) % n
        sel = np.where(rel < 333)[0]
        slots = (pos + rel[sel]) % ring.capacity
        assert (U.bits32(ring.s.cpu().numpy()[slots]) == U.bits32(pre[sel])).all()
        assert (U.bits32(ring.s2.cpu().numpy()[slots]) == U.bits32(o_ref[sel])).all()
        assert (U.bits32(ring.a.cpu().numpy()[slots]) == U.bits32(a[sel])).all()
        assert (ring.r.cpu().numpy()[slots] == r_ref[sel].astype(np.float32)).all()
        pos += 333
    assert (env.idx == ref.idx()).all() and (env.step == nsteps).all()
    env.close()


def test_fused_step_at_four_times_config_3_ragged():
    """262 144 + 37 households (4 x BASELINE config 3, ragged against every tile size): the fused kernel's grid arithmetic beyond the
    benchmarked size.  Actions against the float64 evaluation and transitions against the env oracle on a sample spread over the whole
    range (first, last, the ragged tail, random envs); index bookkeeping for every env."""
    torch, S, D = _mods()
    n = 262144 + 37
    tab = S.tables.synthetic_table("train", 98)
    env = S.ShemsBatch(n, 72, [tab], [S.make_config(98, 0, tab.shape[0])]).use_torch_stream()
    ag = D.Agent(seed=5)
    p = D.init_params(5, 9, 2, 0)
    p[128000:129000] *= 60.0
    ag.set_params(actor=p)
    env.reset_(9, episode=3)
    st0, idx0 = env.state, env.idx.copy()
    lo, hi = st0.min(0), st0.max(0)
    ag.set_norm(lo, hi)
    sample = np.unique(np.concatenate([[0, 1, 63, 64, n - 38, n - 37, n - 2, n - 1], np.random.default_rng(2).choice(n, 1500, replace=False)]))
    ref = oracle_c.Batch(len(sample), 72, tab, oracle_c.profile(98))
    ref.set_state(st0[sample], idx0[sample])
    a_out = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    rew = torch.empty(n, dtype=torch.float64, device="cuda")
    for t in range(2):
        pre = env.state[sample]
        ag.act_step(env, train=True, tick=t, a_out=a_out, rewards=rew)
        env.check_error()
        a = a_out.cpu().numpy()
        assert np.isfinite(a).all() and a.min() >= -1 and a.max() <= 1
        # act() of the sampled envs: the float64 forward + the noise rows of THOSE env ids (the stream is indexed by env)
        det = DO.actor_forward(p, DO.normalize(pre, lo, hi), dtype=np.float64)
        want = np.clip(det + (0.0 + 0.1 * DO.gauss_noise(5, t, n)[sample].astype(np.float64)), -1.0, 1.0)
        rc, r_ref, o_ref, _ = ref.step(oracle_c.scale_action(a[sample]), 0)
        assert rc == 0 and (U.bits64(rew.cpu().numpy()[sample]) == U.bits64(r_ref)).all()
        assert (U.bits32(env.state[sample]) == U.bits32(o_ref)).all()
        assert np.abs(a[sample] - want).max() < 5e-6 + ATOL
    # deterministic actions on the sample against the float64 evaluation (no noise stream to index)
    got = ag.act(torch.from_numpy(st0[sample]).cuda(), train=False).cpu().numpy()
    assert np.abs(got - DO.act(p, st0[sample], lo, hi, False, dtype=np.float64)).max() < ATOL
    assert (env.idx == idx0 + 2).all() and (env.step == 2).all()
    env.close()


def test_ou_and_epsilon_noise_branches():
    """noise_type "ou" (persistent per-env OUNoise.X) and "en" (epsilon-greedy uniform actions), DDPG.jl:49-72, 157-170."""
    torch, S, D = _mods()
    rng = np.random.default_rng(5)
    m = 5000
    obs = _rand_obs(rng, m)
    p = D.init_params(9, 9, 2, 0); p[128000:129000] *= 30
    s_min, s_max = obs.min(0), obs.max(0) + 0.1
    ag = D.Agent(seed=9, noise_type="ou", sigma=0.3, theta=0.15, dt=1e-2)
    ag.set_params(actor=p); ag.set_norm(s_min, s_max)
    X = np.zeros((m, 2), np.float32)
    dev = torch.from_numpy(obs).cuda()
    for t in range(4):
        out = ag.act(dev, train=True, tick=t).cpu().numpy()
        ref = DO.act(p, obs, s_min, s_max, True, seed=9, tick=t, sigma=0.3, noise="ou", ou_state=X, dtype=np.float64)
        assert np.abs(out - ref).max() < 2e-5
    assert np.abs(ag.ou_state.cpu().numpy() - X).max() < 1e-5 and np.abs(X).std() > 0.03      # the state accumulated
    clean = ag.act(dev, train=False).cpu().numpy()
    assert np.abs(clean - DO.act(p, obs, s_min, s_max, False, dtype=np.float64)).max() < ATOL
    ag2 = D.Agent(seed=9, noise_type="en", eps=0.3)
    ag2.set_params(actor=p); ag2.set_norm(s_min, s_max)
    out = ag2.act(dev, train=True, tick=7).cpu().numpy()
    ref = DO.act(p, obs, s_min, s_max, True, seed=9, tick=7, noise="en", eps=0.3, dtype=np.float64)
    assert np.abs(out - ref).max() < ATOL
    frac = (np.abs(out - clean).max(1) > 1e-4).mean()
    assert 0.25 < frac < 0.35


def test_noise_mean_accumulator_is_acts_second_return_value():
    """act() returns (action, mean(noise)) and episode! sums the second value into noise_eps / noise_mean[i] (DDPG.jl:172-175, 224, 255).
    shems_act_params.noise_acc receives that value per env: for Gaussian noise 0.5 * (n0 + n1) -- recovered here from the unclamped
    actions (tanh of a freshly initialised actor is ~0: nothing clamps) -- and exactly 0 in evaluation mode."""
    torch, S, D = _mods()
    tab = U.tables_mod().synthetic_table("train", 98)
    n = 3000
    env = S.ShemsBatch(n, 72, [tab], [S.make_config(98, 0, tab.shape[0])]).use_torch_stream()
    env.reset_(5, episode=1)
    ag = D.Agent(seed=21)
    st = env.state
    ag.set_norm(st.min(0), st.max(0))
    dev = torch.from_numpy(st).cuda()
    clean = ag.act(dev, train=False).cpu().numpy()
    acc = torch.zeros(n, dtype=torch.float32, device="cuda")
    a_out = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    ag.act_step(env, train=True, tick=4, a_out=a_out, noise_acc=acc)
    noisy = a_out.cpu().numpy()
    assert np.abs(noisy).max() < 1.0                                       # nothing clamped
    want = 0.5 * ((noisy[:, 0] - clean[:, 0]) + (noisy[:, 1] - clean[:, 1]))
    got = acc.cpu().numpy()
    assert np.abs(got - want).max() < 2e-6 and 0.05 < got.std() < 0.09      # sigma_act / sqrt(2) = 0.0707
    ag.act_step(env, train=False, tick=5, a_out=a_out, noise_acc=acc)      # evaluation: adds 0
    assert (acc.cpu().numpy() == got).all()
    env.close()


_FORM_SCRIPT = r"""
import sys, zlib, importlib
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import util as U
S = U.pkg(); D = importlib.import_module(U.PKG_NAME + ".ddpg"); R = importlib.import_module(U.PKG_NAME + ".replay")
import torch
out = []
for n, mixed in ((1, False), (2048 + 13, False), (6000 + 5, False), (20000, False), (40000, False), (3000, True), (12000, True),
                 (40000 + 7, True)):                       # 32-, 64- and 128-env tiles; one tile, odd tile counts, ragged last tiles
    if mixed:                                            # ten charger profiles x weight sweep: a config index per env
        tabs, cfgs, co = S.mixed_profile_setup(n)
        env = S.ShemsBatch(n, 72, tabs, cfgs, co).use_torch_stream()
    else:
        tab = S.tables.synthetic_table("train", 98)
        env = S.ShemsBatch(n, 72, [tab], [S.make_config(98, 0, tab.shape[0])]).use_torch_stream()
    ag = D.Agent(seed=7)
    env.reset_(3, episode=0)
    st = env.state; ag.set_norm(st.min(0), st.max(0))
    ring = R.ReplayRing(24000)
    for t in range(3):
        ag.act_step(env, train=True, tick=t, ring=ring)
    torch.cuda.synchronize()
    out.append(zlib.crc32(env.state.tobytes()) ^ zlib.crc32(ring.s2.cpu().numpy().tobytes()) ^ zlib.crc32(ring.a.cpu().numpy().tobytes()))
# the GROUPED call (shems_act_step_group_dev: env i acts with learner i / E's actor, every learner pushes into its own ring): 32-, 64- and
# 128-env tiles, Flux order (the forced form applies) and the tiled working layout (always the free-running forms, W2 from the tiles)
G = importlib.import_module(U.PKG_NAME + ".group")
pairs = dict()
for L, E, tiled in ((5, 256, False), (20, 1024, False), (40, 1024, False), (16, 128, True), (20, 1024, True), (40, 1024, True), (48, 32, False), (48, 32, True),
                    (300, 96, False), (300, 96, True), (520, 64, False), (520, 64, True)):   # env blocks of 32 / 96 / 64: tiles never straddle two learners (64: 64-env tiles where the batch alone would take 128)
    n = L * E
    tab = S.tables.synthetic_table("train", 98)
    env = S.ShemsBatch(n, 72, [tab], [S.make_config(98, 0, tab.shape[0])]).use_torch_stream()
    grp = G.LearnerGroup(L, E, seed=7, rng_seed=11, capacity=720, form="throughput" if tiled else "latency")
    assert grp.tiled == tiled
    env.reset_(3, episode=0)
    st = env.state
    for ag in grp.learners:
        ag.set_norm(st.min(0), st.max(0))
    for t in range(3):
        grp.tick = t
        grp.act_step(env, train=True, tick=t, window=(grp.rings[0].pos, *grp.ring_window(72, None)))
    torch.cuda.synchronize()
    env.check_error()
    c = zlib.crc32(env.state.tobytes())
    for r in grp.rings:
        c ^= zlib.crc32(r.s2.cpu().numpy().tobytes()) ^ zlib.crc32(r.a.cpu().numpy().tobytes())
    out.append(c)
    pairs.setdefault((L, E), []).append(c)
    env.close()
assert all(len(set(v)) == 1 for v in pairs.values()), pairs          # tiled == Flux order at the same shape
print("FORMS", *out)
"""


def test_every_form_of_the_act_kernel_writes_the_same_bytes():
    """k_act exists in several forms (shared W2 stream / free-running waves with private rings, ring depth 2 or 3, half-resident layer 1
    for 128-env tiles; the two-workgroups-per-CU kernel k_act2 on 64-env tiles; for small batches the column-group kernel k_actg with 8
    waves per env tile, or with TWO workgroups per env tile whose second arriver finishes the tile), chosen by batch size;
    SHEMS_ACT_FORM / SHEMS_ACT_FORM4 force the others.  All of them follow
    the canonical column order (csrc/shems_policy.hip, act_col), so three fused steps (actions, next states, ring contents) must agree
    bit for bit -- whichever half of a split tile arrives second.  Run in child processes, the form is read once."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = _FORM_SCRIPT.format(root=root, tests=os.path.join(root, "tests"))
    got = {}
    # default: k_act2 (two workgroups per CU) above 8 192 envs, the column-group forms below
    forms = {"default": {}, "shared": {"SHEMS_ACT_FORM": "0", "SHEMS_ACT_FORM4": "0"}, "free": {"SHEMS_ACT_FORM": "2", "SHEMS_ACT_FORM4": "1"},
             "ring3": {"SHEMS_ACT_FORM": "3", "SHEMS_ACT_FORM4": "1"}, "group8": {"SHEMS_ACT_FORM": "8"}, "split": {"SHEMS_ACT_FORM": "9"},
             "two_per_cu_everywhere": {"SHEMS_ACT_FORM": "12"}, "split_ring2": {"SHEMS_ACT_FORM": "10"}}
    for name, env in forms.items():
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", script], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        got[name] = [ln for ln in r.stdout.splitlines() if ln.startswith("FORMS")][-1]
    assert len(set(got.values())) == 1, got


_LOAD_SCRIPT = r"""
import sys, zlib, importlib
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import util as U
S = U.pkg(); D = importlib.import_module(U.PKG_NAME + ".ddpg"); R = importlib.import_module(U.PKG_NAME + ".replay")
import torch
n = 4096
tab = S.tables.synthetic_table("train", 98)
env = S.ShemsBatch(n, 72, [tab], [S.make_config(98, 0, tab.shape[0])]).use_torch_stream()
ag = D.Agent(seed=7)
env.reset_(3, episode=0)
st = env.state; ag.set_norm(st.min(0), st.max(0))
ring = R.ReplayRing(24000)
ret = torch.zeros(n, dtype=torch.float64, device="cuda")
crc = 0
for t in range(216):                                   # three episodes, 256 workgroups per launch, no host sync in between
    if t and t % 72 == 0:
        env.reset_(3, episode=t // 72)
    w = D.RingWindow(ring.pos, 333, (t * 333) % n)
    ag.act_step(env, train=True, tick=t, ring=ring, window=w, returns_acc=ret)
    ring.pushed += 333
torch.cuda.synchronize()
env.check_error()
for a in (env.state, env.idx, ring.s2.cpu().numpy(), ring.a.cpu().numpy(), ring.r.cpu().numpy(), ret.cpu().numpy()):
    crc = zlib.crc32(np.ascontiguousarray(a).tobytes(), crc)
print("LOAD", crc)
"""


def test_split_tiles_give_the_same_bytes_whichever_half_arrives_second():
    """The column-split form (two workgroups per env tile, one 8-byte exchange per env, the finisher is whoever arrives second) over 216
    back-to-back launches of 256 workgroups: arrival order varies from launch to launch and tile to tile, the result may not.  Three
    runs of the split form and one of the 8-wave form (no hand-off at all) must end with identical env state, returns and ring."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = _LOAD_SCRIPT.format(root=root, tests=os.path.join(root, "tests"))
    got = []
    for form in ("9", "9", "9", "8"):
        e = dict(os.environ); e["SHEMS_ACT_FORM"] = form
        r = subprocess.run([sys.executable, "-c", script], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        got.append([ln for ln in r.stdout.splitlines() if ln.startswith("LOAD")][-1])
    assert len(set(got)) == 1, got
