"""Empty and malformed inputs through the C ABI: every entry point refuses them with SHEMS_ERR_ARG / SHEMS_ERR_STATE and a message
(shems_last_error), launches nothing, and leaves the handles usable -- the reference's counterparts are Julia MethodErrors / BoundsErrors
at the call site (shems_LU1.jl:203-262, DDPG.jl:121-176)."""
import ctypes as C
import importlib

import numpy as np
import pytest

import util as U

pytestmark = pytest.mark.gpu


def _mods():
    torch = pytest.importorskip("torch")
    return torch, U.pkg(), importlib.import_module(U.PKG_NAME + ".ddpg")


def test_empty_and_malformed_batches_are_refused():
    torch, S, D = _mods()
    tab = S.tables.synthetic_table("train", 98)
    cfg = S.make_config(98, 0, tab.shape[0])
    for n in (0, -3):
        with pytest.raises(S.ShemsError) as ei:
            S.ShemsBatch(n, 72, [tab], [cfg])
        assert ei.value.code == S._capi.ERR_ARG
    with pytest.raises((S.ShemsError, ValueError)):
        S.ShemsBatch(8, 72, [tab[:1]], [S.make_config(98, 0, 1)])            # a one-row table cannot hold an episode
    env = S.ShemsBatch(8, 72, [tab], [cfg]).use_torch_stream()
    ag = D.Agent(seed=3)
    L = D._declare()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    obs = torch.zeros((4, 9), device="cuda")
    out = torch.zeros((4, 2), device="cuda")
    p = ag._act_params(False, 0)
    # act() on zero observations / NULL buffers
    for m, o_ptr, a_ptr in ((0, obs.data_ptr(), out.data_ptr()), (4, None, out.data_ptr()), (4, obs.data_ptr(), None)):
        rc = L.shems_actor_forward_dev(C.byref(p), C.c_void_p(o_ptr), m, C.c_void_p(a_ptr), st)
        assert rc == S._capi.ERR_ARG and b"shems_actor_forward_dev" in S._capi.lib().shems_last_error()
    # unknown noise kind, OU noise without its state
    bad = ag._act_params(True, 0); bad.noise_kind = 7
    assert L.shems_actor_forward_dev(C.byref(bad), C.c_void_p(obs.data_ptr()), 4, C.c_void_p(out.data_ptr()), st) == S._capi.ERR_ARG
    bad = ag._act_params(True, 0); bad.noise_kind = 1; bad.ou_state = None
    assert L.shems_actor_forward_dev(C.byref(bad), C.c_void_p(obs.data_ptr()), 4, C.c_void_p(out.data_ptr()), st) == S._capi.ERR_ARG
    # ring window larger than the batch / the ring, offset outside the batch
    env.reset_(1, episode=0)
    ring = D.ReplayRing(100)
    for win in (D.RingWindow(0, 9, 0), D.RingWindow(0, 4, 8), D.RingWindow(-1, 4, 0)):
        with pytest.raises(S.ShemsError):
            ag.act_step(env, train=True, tick=0, ring=ring, window=win)
    small = D.ReplayRing(4)
    with pytest.raises(S.ShemsError):
        ag.act_step(env, train=True, tick=0, ring=small, window=D.RingWindow(0, 8, 0))
    # the handle is still good
    ag.act_step(env, train=True, tick=0, ring=ring, window=D.RingWindow(0, 8, 0))
    env.check_error()
    assert (env.step == 1).all()
    env.close()


def test_malformed_updates_are_refused():
    torch, S, D = _mods()
    ag = D.Agent(seed=3)
    ring = D.ReplayRing(1000)
    with pytest.raises(S.ShemsError):                       # an empty ring has nothing to sample
        ag.replay(ring, tick=0)
    ring.pushed = 500
    for b in (0, 129, -1):
        ag.batch = b
        if b > 128:
            continue                                        # above 128: sub-batches (tests/test_grid_points.py)
        with pytest.raises(S.ShemsError):
            ag.replay(ring, tick=0)
    ag.batch = 120
    with pytest.raises(S.ShemsError):                       # an exclusion window needs a full ring
        ag.replay(ring, tick=0, exclude=(0, 10))
    L = D._declare()
    d = ag._ddpg_args()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for bp1, bp2 in ((0.0, 0.999), (0.9, 1.0), (1.5, 0.5)):                       # beta powers outside (0, 1)
        assert L.shems_ddpg_critic_apply(C.byref(d), 1e-3, bp1, bp2, 1.0, st) == S._capi.ERR_ARG
    d.ws = None
    rs = ring.struct()
    assert L.shems_ddpg_critic_grad_ex(C.byref(d), C.byref(rs), 500, 1, 0, 0, 0, st) == S._capi.ERR_ARG
    ag.replay(ring, tick=0)                                 # and a well-formed call goes through
    torch.cuda.synchronize()
    assert np.isfinite(ag.critic.cpu().numpy()).all()


def test_malformed_wide_calls_are_refused():
    torch, S, D = _mods()
    L = D._declare()
    n = C.c_int64(0)
    for l1, l2 in ((0, 600), (300, 0), (5000, 600), (300, -1)):
        assert L.shems_wide_params(l1, l2, C.byref(n), None) == S._capi.ERR_ARG
        assert L.shems_wide_workspace_floats(l1, l2, C.byref(n)) == S._capi.ERR_ARG
    assert L.shems_wide_act_workspace_floats(300, 600, 0, C.byref(n)) == S._capi.ERR_ARG
    assert L.shems_wide_params(300, 600, C.byref(n), None) == 0 and n.value == D.net_size(9, 2, (300, 600))
    ag = D.Agent(seed=2, hidden=(300, 600))
    ring = D.ReplayRing(1000)
    with pytest.raises(S.ShemsError):
        ag.replay(ring, tick=0)                             # empty ring
    with pytest.raises(NotImplementedError):
        D.Agent(seed=2, hidden=(300, 600), wide=False)      # does not fit the tuned kernels
    tab = S.tables.synthetic_table("train", 98)
    env = S.ShemsBatch(8, 72, [tab], [S.make_config(98, 0, tab.shape[0])]).use_torch_stream()
    env.reset_(1, episode=0)
    with pytest.raises(NotImplementedError):
        ag.act_step(env, train=False, block_reward=torch.zeros(1, dtype=torch.float64, device="cuda"))
    with pytest.raises(S.ShemsError):
        ag.act_step(env, train=True, tick=0, ring=ring, window=D.RingWindow(0, 9, 0))
    ag.act_step(env, train=False)
    env.check_error()
    env.close()
