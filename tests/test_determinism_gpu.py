"""Run-to-run determinism of the whole training step on the GPU: two workloads built from the same seeds and stepped the same
number of times must end with identical bytes everywhere (networks, targets, ADAM moments, replay ring, env state).  Every
reduction in the kernels has a fixed order, so any difference here is a race."""
import importlib

import numpy as np
import pytest

import util as U

pytestmark = pytest.mark.gpu


def _run(n_envs, steps):
    torch = pytest.importorskip("torch")
    S = U.pkg()
    D = importlib.import_module(U.PKG_NAME + ".ddpg")
    wl = D.TrainWorkload(S, torch, n_envs, seed=4242, updates=1)
    for _ in range(steps):
        wl.step()
    wl.finish()
    ag = wl.agent
    out = {k: getattr(ag, k).clone() for k in ("actor", "critic", "actor_t", "critic_t", "m_actor", "v_actor", "m_critic", "v_critic", "losses")}
    out.update(ring_s=wl.ring.s.clone(), ring_a=wl.ring.a.clone(), ring_r=wl.ring.r.clone(), ring_s2=wl.ring.s2.clone())
    return torch, out, wl.env.state, wl.env.idx, wl.env.step


@pytest.mark.parametrize("n_envs", [2048, 65536])
def test_training_step_is_bitwise_reproducible(n_envs):
    torch, a, sa, ia, ta = _run(n_envs, 150)                 # crosses an episode boundary (72-step episodes, seeded resets)
    _, b, sb, ib, tb = _run(n_envs, 150)
    for k in a:
        assert torch.equal(a[k].view(torch.int32) if a[k].dtype == torch.float32 else a[k], b[k].view(torch.int32) if b[k].dtype == torch.float32 else b[k]), k
    assert np.array_equal(sa.view(np.uint32), sb.view(np.uint32)) and np.array_equal(ia, ib) and np.array_equal(ta, tb)
    assert float(a["losses"][0]) > 0 and torch.isfinite(a["actor"]).all()
