"""Host-side logic that needs no GPU: the epsilon-noise schedule, bench.py's rank launcher."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

import util as U


def test_eps_noise_schedule_follows_the_reference_formula():
    """sample_noise(en::EpsNoise), DDPG.jl:69-72 with input.jl:226-228 (zeta = 0.0005f0, xi_0 = 0.5f0, xi_min = 0.1f0):
    xi = Float32(max(0.5 - zeta * (current_episode - MEM_SIZE / EP_LENGTH["train"]), xi_min)).  "Parity unpinned": the reference
    holds no fixture for it; the values below are the formula evaluated by hand in Float64."""
    D = importlib.import_module(U.PKG_NAME + ".ddpg")
    z = float(np.float32(0.0005))
    for ep in (1, 100, 333, 334, 1001, 1133, 1134, 5000):
        want = np.float32(max(0.5 - z * (ep - 24000 / 72), float(np.float32(0.1))))
        assert D.eps_schedule(ep) == float(want), ep
    assert D.eps_schedule(1) > 0.66                      # no upper clamp in the reference: 0.5 + zeta * 332.33
    assert abs(D.eps_schedule(334) - 0.5) < 1e-3         # reaches xi_0 once the pre-filled memory's worth of episodes has passed
    assert D.eps_schedule(1134) == float(np.float32(0.1)) and D.eps_schedule(10 ** 6) == float(np.float32(0.1))
    xs = [D.eps_schedule(e) for e in range(1, 1300)]
    assert all(a >= b for a, b in zip(xs, xs[1:]))      # monotone decay


def test_bench_launcher_reports_a_failed_rank():
    """`python bench.py --gpus 2` with no rendezvous in the environment becomes the launcher (it makes no GPU call itself).
    In the GPU-less container every rank fails at device selection; the launcher must return non-zero and name the rank
    rather than hang or print a JSON line.  (The success path runs on the GPU box: tests/test_bench_gpu.py.)"""
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    e["HIP_VISIBLE_DEVICES"] = ""                       # no device even if the box has one
    e["CUDA_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         cwd=U.ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "rank" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_rejects_mismatched_world_size():
    e = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=U.ROOT, env=e,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=3" in out.stderr


def test_cpu_baseline_times_the_port_not_python():
    """bench.py's cpu_baseline leg: both figures (all cores / one thread) with the thread counts stated, and no per-env foreign
    call in a timed loop (the batch accessors of oracle_c are single C calls)."""
    import importlib.util
    import inspect
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(U.ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    src = inspect.getsource(bench.cpu_baseline)
    assert "b.at(" not in src and "orc_env_get_state" not in src and "orc_reset(" not in src
    import oracle_c
    for name in ("reset", "state", "set_state", "idx", "steps", "action_drl", "action_rule"):
        body = inspect.getsource(getattr(oracle_c.Batch, name))
        assert "for i in range" not in body, name
    out = bench.cpu_baseline(2048, "train", 1, scale=0.08)
    assert out["kind"] == "port" and out["cores"] == oracle_c.usable_cpus() and out["value"] > 0 and out["one_thread_value"] > 0
    assert out["updates_per_sec"] > 0 and out["one_thread_updates_per_sec"] > 0 and "learner_blas_threads" in out
    assert out["env_only_all_cores_value"] > 0 and out["env_only_value"] > 0
    e = bench.cpu_baseline(2048, "env", 0, scale=0.08)
    assert e["value"] == e["env_only_all_cores_value"] and e["one_thread_value"] == e["env_only_value"]


def test_all_cores_fused_step_agrees_with_the_oracle():
    """oracle/shems_policy_omp.c (the all-cores leg of bench.py's cpu_baseline) is a throughput baseline, but it must compute the
    same vector step: its actions within float rounding of ddpg_oracle.act (same Philox noise stream), and -- given those actions --
    rewards and next states bit-identical to the scalar oracle's step!.  A ragged batch size exercises the block tails."""
    import oracle_c
    import ddpg_oracle as DO
    T = U.tables_mod()
    tab = T.synthetic_table("train", 98)
    n = 1000 + 27
    rng = np.random.default_rng(3)
    a, b = (oracle_c.Batch(n, 72, tab, oracle_c.profile(98)) for _ in range(2))
    idx0 = rng.integers(1, tab.shape[0] - 72 + 1, n)
    soc0 = (rng.random(n) * 6.75).astype(np.float32)
    assert a.reset(False, idx0, soc0) == 0 and b.reset(False, idx0, soc0) == 0
    actor = DO.init_params(77, 9, 2, 0)
    actor[-1004:-2] *= 60.0                                   # W3 of the fresh init is +-3e-3: make the outputs span [-1, 1]
    s = a.state()
    lo, hi = s.min(0), s.max(0)
    for tick in range(3):
        rc, act, rew, s2 = a.policy_step_omp(actor, lo, hi, s, sigma=0.1, seed=9, tick=tick, train=True)
        want = DO.act(actor, s, lo, hi, True, seed=9, tick=tick)
        assert rc == 0 and np.abs(act - want).max() < 2e-6 and np.abs(act).max() <= 1.0 and np.abs(act).mean() > 0.1
        rc2, rew2, s2b, _ = b.step(oracle_c.scale_action(act), 0)
        assert rc2 == 0 and (rew.view(np.uint64) == rew2.view(np.uint64)).all() and (s2.view(np.uint32) == s2b.view(np.uint32)).all()
        s = s2
    rc, act, _, _ = a.policy_step_omp(actor, lo, hi, s, train=False)
    assert np.abs(act - DO.act(actor, s, lo, hi, False)).max() < 2e-6


def test_group_ring_window_and_layout_bookkeeping_without_a_gpu():
    """LearnerGroup's host-side decisions (no device call): how many transitions a learner remembers per vector step and from which
    household -- window_count 1 is the reference's one transition per replay() (DDPG.jl:229-233), always household 0 -- and which copy
    of the layer-2 state is current on the tiled working layout."""
    import importlib
    import types
    G = importlib.import_module(U.PKG_NAME + ".group")
    g = types.SimpleNamespace(envs_per_learner=128, capacity=24000, tick=0)
    rw = lambda *a: G.LearnerGroup.ring_window(g, *a)
    assert rw(72, None) == (128, 0)                     # min(E, 24 000 / 72 = 333)
    g.tick = 5
    assert rw(72, 1) == (1, 0) and rw(72, 32) == (32, (5 * 32) % 128)
    g.envs_per_learner, g.capacity = 2048, 24000
    assert rw(72, None) == (333, (5 * 333) % 2048)
    for bad in (0, 2049):
        with pytest.raises(ValueError):
            rw(72, bad)
    # validity flags: a write into Flux-order tensors while they are stale must be refused, not silently lose the tiles' state
    s = types.SimpleNamespace(tiled=True, _flux_valid=False, _tiled_valid=True)
    with pytest.raises(RuntimeError):
        G.LearnerGroup.flux_changed(s)
    s._flux_valid = True
    G.LearnerGroup.flux_changed(s)
    assert s._tiled_valid is False
    assert G.W2T_FLOATS == 32 * 4 * 64 * 64
    hdr = open(os.path.join(U.ROOT, "include", "shems_hip.h")).read()
    assert "SHEMS_W2T_FLOATS = 32 * 4 * 64 * 64" in hdr
