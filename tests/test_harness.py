"""Results / tracker file formats (CPU) and the tracking inference pass (GPU) -- SURVEY.md 8f rank 2."""
import csv
import importlib
import os

import numpy as np
import pytest

import util as U
from util import oracle_c


def _H():
    return importlib.import_module(U.PKG_NAME + ".harness")


def test_results_and_tracker_file_formats(tmp_path):
    H = _H()
    g = np.load(os.path.join(U.ROOT, "tests", "golden", "oracle_golden.npz"))
    res = g["rule_results"]
    name = H.results_file_name(11709800, "eval", 72, 1001, 250, 500, "Charger98_rule_based_-0.5", -0.5, -0.5, out_dir=str(tmp_path / "out" / "tracker"))
    assert name.endswith("11709800_eval_results_Charger98_rule_based_-0.5_rule_-0.5.csv")     # the reference's file name
    H.write_to_results_file(res, name)
    rows = list(csv.reader(open(name)))
    assert rows[0] == ["index", "c_ev", "EV_target", "EV", "Soc_ev", "rewards", "profit", "discomfort", "penalty", "PV_DE", "B_DE",
                       "GR_DE", "PV_B", "PV_GR", "PV_EV", "B_EV", "GR_EV", "EX_EV", "GR_B", "B_GR", "B", "B_tar", "Soc_b"]
    back = np.array(rows[1:], dtype=np.float64)
    assert back.shape == (72, 23) and (back.view(np.uint64) == res.view(np.uint64)).all()        # round-trips exactly
    assert H.results_file_name(1, "eval", 72, 1001, 250, 500, "c", 1231, 1001).endswith("1_eval_results_charger_v1_72_1001_250_500_c_1231_1001.csv")
    assert H.results_file_name(1, "eval", 72, 1001, 250, 500, "c", 1231, 301, best=True).endswith("_c_1231_best.csv")
    trk = str(tmp_path / "out" / "Tracker_Charger.csv")
    for k in range(2):
        sums = H.write_to_tracker_file(name, trk, num_ep=1001, job_id=11709800, seed=-0.5, case="Charger98_rule_based_-0.5", idx=-0.5, now=f"t{k}")
    rows = list(csv.reader(open(trk)))
    assert rows[0] == ["time", "NUM_EP", "L1", "L2", "BATCH_SIZE", "MEM_SIZE", "MIN_EXP_SIZE", "season", "run", "Job_ID", "seed",
                       "case", "best", "idx", "rewards", "profit", "discomfort", "penalty", "filename"]
    assert len(rows) == 3 and rows[1][0] == "t0" and rows[2][0] == "t1" and rows[2][-1] == name
    assert float(rows[2][14]) == pytest.approx(res[:, 5].sum(), rel=1e-15) and sums["profit"] == pytest.approx(res[:, 6].sum(), rel=1e-15)
    with pytest.raises(ValueError):
        H.write_to_results_file(np.zeros((3, 22)), str(tmp_path / "x.csv"))


@pytest.mark.gpu
def test_inference_rule_based_and_drl_tracking():
    torch = pytest.importorskip("torch")
    H = _H()
    S = U.pkg()
    D = importlib.import_module(U.PKG_NAME + ".ddpg")
    import ddpg_oracle as DO
    ev = S.tables.synthetic_table("eval", 98)
    env = S.ShemsBatch(3, 1439, [ev], [S.make_config(98, 0, ev.shape[0])])
    # rule based, the whole evaluation set (BASELINE config 1 shape at EP_LENGTH["all","eval"] = 1439)
    total, res = H.inference(env, track=-0.5)
    ref = oracle_c.Batch(1, 1439, ev, oracle_c.profile(98))
    tot_ref, res_ref = ref.rule_episode(0, 1439, want_results=True)
    assert res.shape == (1439, 23) and (U.bits64(res) == U.bits64(res_ref)).all() and (total == tot_ref).all()
    # DRL tracking: the actor's deterministic actions, results rows consistent with the oracle driven by them
    ag = D.Agent(seed=4)
    p = D.init_params(4, 9, 2, 0); p[128000:129000] *= 50
    ag.set_params(actor=p)
    st = np.concatenate([ev[:, [1, 1, 0, 2, 3, 4, 5, 6, 7]]]); st[:, 0] = np.linspace(0, 6.75, len(st))
    ag.set_norm(st.min(0), st.max(0))
    total, res = H.inference(env, ag, track=1, num_steps=200)
    ref = oracle_c.Batch(1, 1439, ev, oracle_c.profile(98)); ref.reset(True)
    for t in range(200):
        a = DO.act(p, ref.state(), st.min(0), st.max(0), False, dtype=np.float64)
        tgt = res[t, [21, 2]].astype(np.float32)[None]                 # B_tar, EV_target the kernel used
        assert np.abs(oracle_c.scale_action(a) - tgt).max() < 1e-5
        rc, r, o, rr = ref.step(tgt, 1, want_results=True)
        assert (U.bits64(rr[0]) == U.bits64(res[t])).all()
    env.close()


@pytest.mark.gpu
def test_tracking_passes_of_a_job_in_one_launch():
    """MAIN:87-105 runs 2 x num_seeds tracking passes; harness.inference_many runs them as one launch, pass p with actor p.  Each pass
    must equal the single-pass call with that actor bit for bit (same kernel, same arithmetic), differ from the others, and leave
    results rows the oracle reproduces from the targets they hold."""
    torch = pytest.importorskip("torch")
    H = _H()
    S = U.pkg()
    D = importlib.import_module(U.PKG_NAME + ".ddpg")
    ev = S.tables.synthetic_table("eval", 98)
    st = np.concatenate([ev[:, [1, 1, 0, 2, 3, 4, 5, 6, 7]]]); st[:, 0] = np.linspace(0, 6.75, len(st))
    lo, hi = st.min(0), st.max(0)
    actors = []
    for seed in (4, 5, 6, 7, 8):
        p = D.init_params(seed, 9, 2, 0); p[128000:129000] *= 50
        actors.append(p)
    steps = 300
    many = S.ShemsBatch(len(actors), 1439, [ev], [S.make_config(98, 0, ev.shape[0])])
    tot, res = H.inference_many(many, np.stack(actors), lo, hi, num_steps=steps)
    many.close()
    assert res.shape == (len(actors), steps, 23) and tot.shape == (len(actors),)
    one = S.ShemsBatch(1, 1439, [ev], [S.make_config(98, 0, ev.shape[0])])
    ag = D.Agent(seed=4)
    ag.set_norm(lo, hi)
    for k, p in enumerate(actors):
        ag.set_params(actor=p)
        t1, r1 = H.inference(one, ag, track=1, num_steps=steps)
        assert (U.bits64(r1) == U.bits64(res[k])).all() and t1[0] == tot[k]
        assert tot[k] == res[k][:, 5].sum() or abs(tot[k] - res[k][:, 5].sum()) < 1e-9
    assert len({float(t) for t in tot}) == len(actors)                   # five different actors, five different passes
    ref = oracle_c.Batch(1, 1439, ev, oracle_c.profile(98)); ref.reset(True)
    for t in range(steps):
        tgt = res[2][t, [21, 2]].astype(np.float32)[None]
        rc, r, o, rr = ref.step(tgt, 1, want_results=True)
        assert rc == 0 and (U.bits64(rr[0]) == U.bits64(res[2][t])).all()
    one.close()


@pytest.mark.gpu
def test_host_array_tracking_pass_of_the_handle_api():
    """shems_track (what julia/ShemsEnv_LU1.jl's track_pass binds): host arrays in and out, the same launch underneath -- its rows and
    returns must be the bytes of the device-pointer path, for the actor and for the rule-based controller."""
    import ctypes as C
    H = _H()
    S = U.pkg()
    D = importlib.import_module(U.PKG_NAME + ".ddpg")
    ev = S.tables.synthetic_table("eval", 98)
    env = S.ShemsBatch(2, 1439, [ev], [S.make_config(98, 0, ev.shape[0])])
    ag = D.Agent(seed=4)
    p = D.init_params(4, 9, 2, 0); p[128000:129000] *= 50
    ag.set_params(actor=p)
    lo, hi = np.zeros(9, np.float32), np.linspace(1, 9, 9).astype(np.float32)
    ag.set_norm(lo, hi)
    L = S._capi.lib()
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    for track, steps in ((1, 300), (-0.5, 1439)):
        tot_dev, res_dev = H.inference(env, ag if track > 0 else None, track=track, num_steps=steps)
        env.reset_(-1)
        res = np.empty((steps, 23), np.float64)
        ret = np.empty(2, np.float64)
        S._capi.check(L.shems_track(env._h, ptr(p) if track > 0 else None, ptr(lo) if track > 0 else None, ptr(hi) if track > 0 else None,
                                    1 if track > 0 else -1, steps, ptr(res), ptr(ret)))
        assert (U.bits64(res) == U.bits64(res_dev)).all() and (ret == tot_dev).all()
    env.close()


@pytest.mark.gpu
def test_tracking_pass_past_the_end_of_the_table_is_a_bounds_error():
    """next_state! reads row idx + 1: one hour more than the table holds is Julia's BoundsError in the reference (LU1:265-279).  The
    one-launch pass stops at that hour, raises the sticky error, and everything up to it is intact (bit-exact with the oracle)."""
    H = _H()
    S = U.pkg()
    ev = S.tables.synthetic_table("eval", 98)                 # 1440 rows: 1439 hours are possible from row 1
    env = S.ShemsBatch(2, 1439, [ev], [S.make_config(98, 0, ev.shape[0])])
    with pytest.raises(S._capi.BoundsError):
        H.inference(env, track=-0.5, num_steps=1445)
    assert (env.idx == 1440).all() and (env.step == 1439).all()      # stopped at the last row, 1439 hours done
    total, res = H.inference(env, track=-0.5, num_steps=1439)         # the handle works on after the error was read
    ref = oracle_c.Batch(1, 1439, ev, oracle_c.profile(98))
    tot_ref, res_ref = ref.rule_episode(0, 1439, want_results=True)
    assert (U.bits64(res) == U.bits64(res_ref)).all() and (total == tot_ref).all()
    env.close()


def test_checkpoint_roundtrip_and_reference_file_stems(tmp_path):
    CK = importlib.import_module(U.PKG_NAME + ".checkpoint")
    import ddpg_oracle as DO
    actor = DO.init_params(1231, 9, 2, 0)
    tr, sm, nm = np.arange(5, dtype=np.float32), np.array([1.5, -2.0]), np.zeros(5, np.float32)
    st = CK.save(actor, tr, sm, 3, nm, idx=1001, case="Charger98_x", rng=1231, out_dir=str(tmp_path / "out" / "bson"), path="temp")
    assert st.endswith(os.path.join("out", "bson", "temp", "DDPG_Shems_Charger_v1_72_1001_250_500_Charger98_x_1231"))
    a, tr2, sm2, best, nm2 = CK.load(idx=1001, case="Charger98_x", rng=1231, out_dir=str(tmp_path / "out" / "bson"), path="temp")
    assert (a == actor).all() and (tr2 == tr).all() and (sm2 == sm).all() and best == 3 and (nm2 == nm).all()
    assert len(CK.load(idx=1001, scores_only=True, case="Charger98_x", rng=1231, out_dir=str(tmp_path / "out" / "bson"), path="temp")) == 4
    with pytest.raises(ValueError):
        CK.save(actor[:10], tr, sm, 3, nm, idx=1, out_dir=str(tmp_path))
