"""The R18 entry point (DDPG_reinforce_charger_v1.jl:10-47, 87-110) on the GPU path: env-var contract, MAIN's order, the reference's
file names and headers.  The decoding rules of the job's input file are host logic and are tested without a GPU."""
import csv
import importlib
import os

import numpy as np
import pytest

import util as U

M = importlib.import_module(U.PKG_NAME + ".main")


def test_job_id_decoding_and_case_string():
    """TUNED = input_templates/input09_08_on_01-09_eval.jl: JOB_ID ...9808 -> Charger98, suffix 08 = ternary 0022 -> BATCH 120,
    noise_act 0.1, (250, 500), (1f-4, 1f-3); TASK_ID 3 -> rng_run 1233."""
    c = M.config_from_env({"JOB_ID": "1179808", "TASK_ID": "3", "GPU_ID": "1"})
    assert (c.charger_id, c.Charger_ID, c.seed_run, c.rng_run, c.gpu_id) == (98, "Charger98", 3, 1233, 1)
    assert (c.BATCH_SIZE, c.noise_act, c.L1, c.L2, c.eta_act, c.eta_crit, c.NUM_EP, c.MEM_SIZE, c.noise_type) == \
           (120, 0.1, 250, 500, 1e-4, 1e-3, 1001, 24000, "gn")
    assert c.case == ("Charger98_dw0.01_p0.1_B120_M24000_gn-o0.1_th0.15_Y0.99_tau0.001_lract0.0001_lrcrit0.001_nact0.1_ntrg0.2")
    # other digits of the grid
    c2 = M.config_from_env({"JOB_ID": "1170143", "TASK_ID": "12", "GPU_ID": "0"})      # 43 = 1121 (base 3)
    assert (c2.charger_id, c2.BATCH_SIZE, c2.noise_act, (c2.L1, c2.L2), (c2.eta_act, c2.eta_crit), c2.rng_run) == \
           (1, 100, 0.2, (250, 500), (5e-4, 5e-3), 12312)
    c3 = M.config_from_env({"JOB_ID": "1170100", "TASK_ID": "1", "GPU_ID": "0"})       # 00 -> (300, 600): wider than the tuned kernels, runs
    assert (c3.L1, c3.L2, c3.eta_act) == (300, 600, 1e-5)                               # layer by layer (csrc/shems_wide.hip)
    M._check_supported(c3)
    with pytest.raises(KeyError):
        M.config_from_env({"JOB_ID": "1179808", "GPU_ID": "0"})
    # Julia's printing of the Float32 / Float64 values that go into file names
    for x, f32, want in ((1e-4, True, "0.0001"), (1e-5, True, "1.0e-5"), (5e-3, True, "0.005"), (0.99, True, "0.99"), (0.999, True, "0.999"),
                         (0.01, False, "0.01"), (24000.0, False, "24000.0"), (-0.5, False, "-0.5"), (2.0, True, "2.0")):
        assert M.julia_float(x, f32) == want, (x, f32)
    # the untuned template of input.jl decodes three digits
    c4 = M.config_from_env({"JOB_ID": "1179826", "TASK_ID": "2", "GPU_ID": "0", "SHEMS_INPUT_TEMPLATE": "input"})    # 26 = 222
    assert (c4.MEM_SIZE, c4.BATCH_SIZE, c4.L1, c4.L2, c4.gamma, c4.sigma, c4.theta, c4.noise_type, c4.NUM_EP) == \
           (24000, 120, 300, 600, 0.99, 0.2, 0.2, "ou", 101)
    assert c4.case.startswith("Charger98_disw2_pen0.5_BATCH120_MEM24000_ou-noise_om0.2_th0.2_Y0.99_tau0.001_nact0.0001_ncrit0.001_smart-trainEP")


def test_unsupported_parameter_noise_jobs_are_refused_before_anything_runs():
    """noise_type "pn" (DDPG.jl:74-96) cannot run on a zero-padded smaller network nor with BATCH_SIZE > 128: both used to surface only
    partway through a job (Agent.__init__ / the first replay()); _check_supported names the JOB_ID up front.  The templates themselves
    never select "pn" (tuned: "gn", input.jl: "ou"); a caller building the RunConfig can."""
    import dataclasses
    c = M.config_from_env({"JOB_ID": "1179808", "TASK_ID": "1", "GPU_ID": "0"})
    ok = dataclasses.replace(c, noise_type="pn")
    M._check_supported(ok)                                                   # (250, 500), BATCH 120: runs
    M._check_supported(dataclasses.replace(c, noise_type="pn", L1=300, L2=600))   # the wide path carries parameter noise
    for bad, word in ((dataclasses.replace(c, noise_type="pn", L1=200, L2=400), "zero-padded"), (dataclasses.replace(c, noise_type="pn", BATCH_SIZE=150), "BATCH_SIZE = 150"),
                      (dataclasses.replace(c, noise_type="xx"), "noise_type")):
        with pytest.raises(NotImplementedError) as ei:
            M._check_supported(bad)
        assert "1179808" in str(ei.value) and word in str(ei.value)


@pytest.mark.gpu
def test_set_params_refuses_a_vector_of_the_wrong_size():
    pytest.importorskip("torch")
    D = importlib.import_module(U.PKG_NAME + ".ddpg")
    for hidden in ((250, 500), (200, 400), (300, 600)):
        ag = D.Agent(seed=1, hidden=hidden)
        n = D.net_size(9, 2, hidden)
        ag.set_params(actor=np.zeros(n, np.float32))
        with pytest.raises(ValueError) as ei:
            ag.set_params(actor=np.zeros(n - 1, np.float32))
        assert "set_params" in str(ei.value) and str(n) in str(ei.value)
        with pytest.raises(ValueError):
            ag.set_params(critic=np.zeros(7, np.float32))


@pytest.mark.gpu
def test_entry_script_runs_mains_order_and_writes_the_reference_files(tmp_path):
    torch = pytest.importorskip("torch")
    env = {"JOB_ID": "1179808", "TASK_ID": "1", "GPU_ID": "0", "SHEMS_NUM_EP": "2", "SHEMS_NUM_SEEDS": "1", "SHEMS_NUM_ENVS": "64",
           "SHEMS_SYNTHETIC_DATA": "1"}
    logs = []
    cwd0 = os.getcwd()
    try:
        cfg, written = M.main(env, cwd=str(tmp_path), log=logs.append)
    finally:
        os.chdir(cwd0)
    case = cfg.case
    stem = f"DDPG_Shems_Charger_v1_72_2_250_500_{case}_1231"
    for f in (f"out/bson/{stem}_actor_2.bson", f"out/bson/{stem}_scores_2.bson", f"out/bson/temp/{stem}_actor_1.bson",
              f"out/bson/temp/{stem}_scores_1.bson", "data/Charger98_all_train_fix.csv", "data/Charger98_all_eval_fix.csv"):
        assert (tmp_path / f).exists(), f
    last = tmp_path / f"out/tracker/1179808_eval_results_charger_v1_72_2_250_500_{case}_1231_2.csv"
    best = tmp_path / f"out/tracker/1179808_eval_results_charger_v1_72_2_250_500_{case}_1231_best.csv"
    assert [os.path.basename(w) for w in written] == [last.name, best.name]
    H = importlib.import_module(U.PKG_NAME + ".harness")
    for f in (last, best):
        rows = list(csv.reader(open(f)))
        assert rows[0] == H.RESULTS_HEADER and len(rows) == 1 + 1439                  # EP_LENGTH["all", "eval"] steps of the eval pass
        a = np.array(rows[1:], float)
        assert (a[:, 0] == np.arange(2, 1441)).all() and np.isfinite(a).all()         # `index` column = env.idx after the step (LU1:453, 476)
    tr = list(csv.reader(open(tmp_path / "out/Tracker_Charger.csv")))
    assert tr[0] == H.TRACKER_HEADER and len(tr) == 3
    assert [r[12] for r in tr[1:]] == ["false", "true"] and [r[13] for r in tr[1:]] == ["2", "1"] and tr[1][9] == "1179808" and tr[1][10] == "1231"
    assert any("Starting script with JOB_ID: 1179808, TASK_ID: 1 for charger Charger98" in l for l in logs)
    assert any("is done!" in l for l in logs)
    # scores file: total_reward [NUM_EP], score_mean [ceil(NUM_EP / test_every)], best_run
    C = importlib.import_module(U.PKG_NAME + ".checkpoint")
    os.chdir(tmp_path)
    try:
        tr_, sm_, best_run, nm_ = C.load(idx=2, scores_only=True, ep_len=72, num_ep=2, case=case, rng=1231)
    finally:
        os.chdir(cwd0)
    assert tr_.shape == (2,) and sm_.shape == (1,) and best_run == 1 and np.isfinite(sm_).all()


@pytest.mark.gpu
def test_entry_script_on_a_charger_with_real_series(tmp_path):
    """Charger04: train and eval tables are the series reconstructed from the reference's MPC results (4 319 / 1 439 rows, one per MPC
    decision).  The 1 439-step tracking pass reads row 1 440, which those files do not hold: the demo data writer pads it (tables.pad_rows)."""
    pytest.importorskip("torch")
    env = {"JOB_ID": "1170408", "TASK_ID": "1", "GPU_ID": "0", "SHEMS_NUM_EP": "2", "SHEMS_NUM_SEEDS": "1", "SHEMS_NUM_ENVS": "64",
           "SHEMS_SYNTHETIC_DATA": "1"}
    logs = []
    cwd0 = os.getcwd()
    try:
        cfg, written = M.main(env, cwd=str(tmp_path), log=logs.append)
    finally:
        os.chdir(cwd0)
    assert cfg.Charger_ID == "Charger04" and len(written) == 2
    T = U.tables_mod()
    real = T.real_series(4, "eval")
    got = T.load_csv(str(tmp_path / "data/Charger04_all_eval_fix.csv"))
    assert real.shape == (1439, 8) and got.shape == (1440, 8) and (got[:1439] == real).all() and (got[1439] == real[1415]).all()
    assert (T.load_csv(str(tmp_path / "data/Charger04_all_train_fix.csv")) == T.real_series(4, "train")).all()
    assert any("padded with the rows 24 h earlier" in l for l in logs)
    rows = list(csv.reader(open(tmp_path / written[0])))
    assert len(rows) == 1 + 1439 and np.isfinite(np.array(rows[1:], float)).all()


@pytest.mark.gpu
def test_entry_script_on_the_200_400_grid_point(tmp_path):
    """JOB_ID suffix 03 = ternary 0010 of the tuned template: (L1, L2) = (200, 400) (input09_08_on_01-09_eval.jl:62-66).  The learner runs
    zero-padded on the (250, 500) kernels (ddpg.pad_net); the checkpoints hold the 200 x 400 chain and the file names say so."""
    pytest.importorskip("torch")
    env = {"JOB_ID": "1179803", "TASK_ID": "1", "GPU_ID": "0", "SHEMS_NUM_EP": "2", "SHEMS_NUM_SEEDS": "1", "SHEMS_NUM_ENVS": "64",
           "SHEMS_SYNTHETIC_DATA": "1"}
    cwd0 = os.getcwd()
    try:
        cfg, written = M.main(env, cwd=str(tmp_path), log=lambda *_: None)
    finally:
        os.chdir(cwd0)
    assert (cfg.L1, cfg.L2) == (200, 400) and len(written) == 2 and all("_200_400_" in w for w in written)
    B = importlib.import_module(U.PKG_NAME + ".bson_chain")
    stem = f"DDPG_Shems_Charger_v1_72_2_200_400_{cfg.case}_1231"
    a = B.load_chain(str(tmp_path / f"out/bson/{stem}_actor_2.bson"), hidden=(200, 400))
    assert a.size == 9 * 200 + 200 + 200 * 400 + 400 + 400 * 2 + 2 and np.isfinite(a).all() and np.count_nonzero(a) > 80000
    rows = list(csv.reader(open(tmp_path / written[0])))
    assert len(rows) == 1 + 1439 and np.isfinite(np.array(rows[1:], float)).all()
