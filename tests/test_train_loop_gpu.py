"""GPU tests of the native training loop (shems_train_steps: the hour loop of episode!, DDPG.jl:195-234, enqueued by one foreign call)
and of the range form of the fused step it is built on (shems_act_step_range_dev).

Parity statements: the native loop in program order leaves the bytes of the host loop over shems_act_step_dev / shems_ddpg_update (which
tests/test_train_gpu.py, test_policy_gpu.py and test_ddpg_gpu.py hold against the oracle); the order-exact pipelined mode leaves the bytes
of the ordered loop; the plain pipelined mode (replay(t) does not see step t's inserts) leaves the bytes of the host-side pipelined loop."""
import ctypes as C
import importlib

import numpy as np
import pytest

import util as U

pytestmark = pytest.mark.gpu

_KEYS = ("actor", "critic", "actor_t", "critic_t", "m_actor", "v_actor", "m_critic", "v_critic", "losses", "grad_actor", "grad_critic")


def _mods():
    torch = pytest.importorskip("torch")
    return torch, U.pkg(), importlib.import_module(U.PKG_NAME + ".ddpg")


def _end_state(torch, wl):
    wl.finish()
    ag = wl.agent
    out = {k: getattr(ag, k).clone() for k in _KEYS}
    out.update(ring_s=wl.ring.s.clone(), ring_a=wl.ring.a.clone(), ring_r=wl.ring.r.clone(), ring_s2=wl.ring.s2.clone(), rew=wl.rew32.clone())
    host = dict(t=wl.t, episode=wl.episode, pushed=wl.ring.pushed, updates=ag.updates, bp_c=tuple(ag.bp_critic), bp_a=tuple(ag.bp_actor))
    return out, (wl.env.state.copy(), wl.env.idx.copy(), wl.env.step.copy()), host


def _same(torch, a, b, what):
    (ta, ea, ha), (tb, eb, hb) = a, b
    for k in ta:
        x, y = ta[k], tb[k]
        assert torch.equal(x.view(torch.int32), y.view(torch.int32)), f"{what}: {k} differs"
    for i, name in enumerate(("obs", "idx", "step")):
        assert np.array_equal(ea[i].view(np.uint32) if ea[i].dtype == np.float32 else ea[i], eb[i].view(np.uint32) if eb[i].dtype == np.float32 else eb[i]), f"{what}: env {name}"
    assert ha == hb, (what, ha, hb)


def _run(n, steps, chunks, **kw):
    torch, S, D = _mods()
    wl = D.TrainWorkload(S, torch, n, seed=777, updates=kw.pop("updates", 1), **kw)
    done = 0
    for c in chunks:
        wl.steps(c)
        done += c
    assert done == steps
    return torch, _end_state(torch, wl)


@pytest.mark.parametrize("n", [333, 2048, 9000])
def test_native_ordered_loop_leaves_the_bytes_of_the_host_loop(n):
    # 150 steps cross two episode boundaries (72-step episodes); 9 000 envs: the window wraps around the batch; 333 envs: the window is the batch
    torch, host = _run(n, 150, [150], loop="host")
    _, nat = _run(n, 150, [150], loop="native")
    _same(torch, host, nat, "native vs host")
    _, nat2 = _run(n, 150, [1, 70, 2, 77], loop="native")               # the in/out counters carry over between calls
    _same(torch, host, nat2, "native in four calls vs host")


def test_native_loop_with_two_updates_per_step_and_none():
    torch, host = _run(1024, 40, [40], loop="host", updates=2)
    _, nat = _run(1024, 40, [40], loop="native", updates=2)
    _same(torch, host, nat, "two updates per step")
    _, host0 = _run(1024, 80, [80], loop="host", updates=0)
    _, nat0 = _run(1024, 80, [80], loop="native", updates=0)
    _same(torch, host0, nat0, "no learning")


@pytest.mark.parametrize("n", [512, 4096, 8192 + 37, 20000])
def test_order_exact_pipelined_loop_leaves_the_bytes_of_the_ordered_loop(n):
    torch, ordered = _run(n, 150, [150], loop="native")
    _, exact = _run(n, 150, [150], loop="native", overlap="exact")
    _same(torch, ordered, exact, "exact pipelined vs ordered")
    _, exact2 = _run(n, 150, [50, 1, 99], loop="native", overlap="exact")
    _same(torch, ordered, exact2, "exact pipelined in three calls vs ordered")


@pytest.mark.parametrize("n", [2048, 4096, 8192, 16384, 20000])
def test_native_pipelined_loop_leaves_the_bytes_of_the_host_pipelined_loop(n):
    """Up to 16 384 envs the native pipelined loop has NO queue-level dependency: the launches wait in the kernel for counts of finished
    producer workgroups (DevSync).  A lost or late dependency shows as different bytes against the host loop, whose two streams are
    ordered by events."""
    torch, host = _run(n, 300, [300], loop="host", overlap="pipelined")
    _, nat = _run(n, 300, [300], loop="native", overlap="pipelined")
    _same(torch, host, nat, "pipelined native vs host")
    _, nat3 = _run(n, 300, [7, 200, 93], loop="native", overlap="pipelined")
    _same(torch, host, nat3, "pipelined native in three calls vs host")
    _, ordered = _run(n, 300, [300], loop="native")
    assert not torch.equal(ordered[0]["actor"], nat[0]["actor"])         # the documented deviation: replay(t) does not see step t's inserts


def test_range_launches_leave_the_bytes_of_one_launch():
    """shems_act_step_range_dev: a batch stepped range by range, in any order, = one shems_act_step_dev (noise keyed by the env's index
    in the view, ring window defined on the whole batch)."""
    torch, S, D = _mods()
    n = 5000

    def fresh():
        wl = D.TrainWorkload(S, torch, n, seed=99, updates=0, loop="host")
        return wl

    a, b = fresh(), fresh()
    L = a.agent.L
    for t in range(5):
        w = D.RingWindow(a.ring.pos, a.win, (t * a.win) % n)
        a._act(t)
        v = b.env.view()
        p = b.agent._act_params(True, t)
        rs = b.ring.struct()
        for lo, cnt in ((4097, 903), (0, 31), (31, 1), (32, 4065)):        # unaligned pieces, out of order
            S._capi.check(L.shems_act_step_range_dev(C.byref(v), C.byref(p), lo, cnt, C.c_void_p(b.rew32.data_ptr()), C.byref(rs), C.byref(w),
                                                     b.agent._stream()))
        b.ring.pushed += b.win
    torch.cuda.synchronize()
    for x, y in ((a.ring.s, b.ring.s), (a.ring.a, b.ring.a), (a.ring.r, b.ring.r), (a.ring.s2, b.ring.s2), (a.rew32, b.rew32)):
        assert torch.equal(x.view(torch.int32), y.view(torch.int32))
    assert np.array_equal(a.env.state.view(np.uint32), b.env.state.view(np.uint32)) and np.array_equal(a.env.idx, b.env.idx)
    v = b.env.view()
    p = b.agent._act_params(True, 0)
    for lo, cnt in ((-1, 5), (0, 0), (4990, 11)):
        with pytest.raises(S.ShemsError):
            S._capi.check(L.shems_act_step_range_dev(C.byref(v), C.byref(p), lo, cnt, None, None, None, b.agent._stream()))


def test_train_loop_refuses_bad_records():
    torch, S, D = _mods()
    wl = D.TrainWorkload(S, torch, 1024, seed=5, updates=1, loop="native")
    L = wl._native_loop()
    st = wl.agent._stream()
    for field, val in (("window", 2000), ("ep_len", 0), ("mode", 7), ("updates_per_step", -1), ("t", -3)):
        old = getattr(L, field)
        setattr(L, field, val)
        with pytest.raises(S.ShemsError):
            S._capi.check(wl.agent.L.shems_train_steps(C.byref(L), 1, st, None))
        setattr(L, field, old)
    L.mode = D.LOOP_PIPELINED                                              # no actor_pub buffers, one stream
    with pytest.raises(S.ShemsError):
        S._capi.check(wl.agent.L.shems_train_steps(C.byref(L), 1, st, st))
    L.mode = D.LOOP_ORDERED
    S._capi.check(wl.agent.L.shems_train_steps(C.byref(L), 0, st, None))   # k = 0: nothing
    wl.steps(3)                                                            # the record is still usable
    wl.finish()


def test_pipelined_modes_refuse_a_wide_network():
    """a pipelined step used to hand the tuned kernel a clone of a (300, 600) actor in the wide layout (silently wrong
    actions).  Pipelined modes run the tuned kernels only."""
    torch, S, D = _mods()
    for ov in ("pipelined", "exact"):
        with pytest.raises(NotImplementedError):
            D.TrainWorkload(S, torch, 1024, seed=3, updates=1, overlap=ov, hidden=(300, 600))
    wl = D.TrainWorkload(S, torch, 1024, seed=3, updates=1, hidden=(300, 600))      # ordered: runs (host loop: the wide path has its own entry points)
    assert wl.loop == "host"
    wl.steps(3)
    wl.finish()


def test_fused_step_on_more_than_32_streams():
    """the two-workgroups-per-tile form keeps one exchange slab per (device, stream) in a 32-entry table; the 33rd stream
    used to fail every launch of <= 4 096 envs for the rest of the process.  It now runs the one-workgroup-per-tile form (same bytes)."""
    torch, S, D = _mods()
    ag = D.Agent(seed=2)
    obs = torch.rand((300, 9), device="cuda")
    ref = ag.act(obs, train=False)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(40)]
    for st in streams:
        with torch.cuda.stream(st):
            out = ag.act(obs, train=False)
        st.synchronize()
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32))


def test_mode_change_on_a_live_loop_record():
    """A caller may flip shems_train_loop.mode between calls (the record is caller-owned): pipelined -> order-exact -> pipelined on one
    record (the two signal words carry both modes' dependencies; round 4's event and in-kernel forms were removed in round 5)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys, importlib
sys.path.insert(0, %r); sys.path.insert(0, %r)
import util as U, torch
S = U.pkg(); D = importlib.import_module(U.PKG_NAME + ".ddpg")
wl = D.TrainWorkload(S, torch, 4096, seed=5, updates=1, loop="native", overlap="pipelined")
wl.steps(20)
L = wl._native_loop(); L.mode = D.LOOP_PIPELINED_EXACT; wl.overlap_mode = D.LOOP_PIPELINED_EXACT
wl.steps(20)
L = wl._native_loop(); L.mode = D.LOOP_PIPELINED; wl.overlap_mode = D.LOOP_PIPELINED
wl.steps(20)
wl.finish()
assert wl.t == 60 and wl.agent.updates == 60 and bool(torch.isfinite(wl.agent.actor).all())
print("MODES ok")
""" % (root, os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "MODES ok" in r.stdout, r.stderr[-2000:]
