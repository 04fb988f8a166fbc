"""The reference's entry point, RL-SHEMS/DDPG_reinforce_charger_v1.jl, on the GPU-resident path.

Process contract kept from the reference (MAIN = DDPG_reinforce_charger_v1.jl, INPUT = input.jl / the job's input template):
    JOB_ID    MAIN:10   digits 3-4 from the right = charger id (INPUT:38, LU1:45); last two digits = hyper-parameter code
              (ternary, `set_hyperparameters`, input_templates/input09_08_on_01-09_eval.jl:62-106)
    TASK_ID   INPUT:35-36  seed_run; rng_run = parse(Int, "123" * TASK_ID) (INPUT:135-136)
    GPU_ID    MAIN:12-14   device index
    data      data/<Charger_ID>_<season>_<split>_<price>.csv (INPUT:162-164), relative to the working directory
    outputs   out/bson/[temp/]DDPG_Shems_Charger_v1_<EP>_<NUM_EP>_<L1>_<L2>_<case>_<rng>_{actor,scores}_<idx>   (MPS:263-268)
              out/tracker/<Job_ID>_<run>_results_charger_v1_<EP>_<NUM_EP>_<L1>_<L2>_<case>_<rng>_<idx|best>.csv (MPS:167-190)
              out/Tracker_Charger.csv                                                                          (MPS:193-212)
Order of MAIN:28-110: populate_memory -> min_max_buffer -> run_episodes (best-score snapshots under out/bson/temp) -> saveBSON ->
(the last seed of the job) inference + write_to_results_file + write_to_tracker_file for every seed's last and best actor.

Differences that follow from the platform, all explicit:
  * the job's Julia input file (out/input/$JOB_ID--input.jl) cannot be executed; its decoding rules are restated here for the two
    templates the thesis used last (TUNED = input09_08_on_01-09_eval.jl and input.jl), selected with SHEMS_INPUT_TEMPLATE;
  * the kernels are built for the tuned architecture (L1, L2) = (250, 500) and 128 minibatch columns; the grids' other points run
    too: smaller networks zero-padded, (300, 600) layer by layer (csrc/shems_wide.hip), BATCH_SIZE 150 / 200 as sub-batches;
  * snapshots are BSON files under the reference's names, laid out as BSON.jl lowers a Chain (bson_chain.py; parity unpinned:
    the reference ships no real .bson to compare with);
  * SHEMS_NUM_ENVS (default 1 = the reference's protocol) trains that many households at once;
  * random streams are Philox counters keyed by the same seeds (Julia's MersenneTwister streams do not exist outside Julia).
"""
from __future__ import annotations

import dataclasses
import os
import sys
import time

import numpy as np

SEED_INI = 123                                   # INPUT:134
EP_LENGTH = {"train": 72, ("all", "eval"): 1439, ("all", "test"): 2999, ("summer", "eval"): 359, ("summer", "test"): 767,
             ("winter", "eval"): 359, ("winter", "test"): 719, ("both", "eval"): 719, ("both", "test"): 1487}   # INPUT:154-160


def julia_float(x, f32=True):
    """How Julia's string interpolation prints a Float32 / Float64 value (`$(x)`): the shortest digits that round-trip, in fixed
    notation when the decimal exponent of that shortest form is in [-4, 21) (always with a decimal point: 0.0001, 24000.0), otherwise
    as d.ddde-N (1.0e-5).  Needed for the `case` string in every file name."""
    v = np.float32(x) if f32 else np.float64(x)
    if v == 0:
        return "0.0"
    mant, exp = np.format_float_scientific(v, unique=True, trim="0", exp_digits=1).split("e")
    if -4 <= int(exp) < 21:
        r = np.format_float_positional(v, unique=True, trim="0")
        return r if "." in r else r + ".0"
    if "." not in mant:
        mant += ".0"
    return f"{mant}e{int(exp)}"


@dataclasses.dataclass
class RunConfig:
    job_id: str
    task_id: str
    gpu_id: int
    template: str = "tuned"
    train: int = 1
    track: float = 1                             # 0 off, 1 DRL, < 0 rule-based (INPUT:47)
    run: str = "eval"
    season: str = "all"
    price: str = "fix"
    num_seeds: int = 40
    test_every: int = 100
    test_runs: int = 100
    # hyper-parameters (filled by set_hyperparameters)
    L1: int = 250
    L2: int = 500
    gamma: float = 0.99
    tau: float = 1e-3
    eta_act: float = 1e-4
    eta_crit: float = 1e-3
    sigma: float = 0.1
    theta: float = 0.15
    noise_act: float = 0.1
    noise_trg: float = 0.2
    DISCOMFORT_WEIGHT_EV: float = 0.01
    penalty: float = 0.1
    TRAIN_EP_LENGTH: int = 72
    NUM_EP: int = 1001
    BATCH_SIZE: int = 120
    MEM_SIZE: int = 24000
    noise_type: str = "gn"

    @property
    def charger_id(self):
        return (int(self.job_id) // 100) % 100          # INPUT:38, LU1:45

    @property
    def Charger_ID(self):
        return "Charger%02d" % self.charger_id

    @property
    def seed_run(self):
        return int(self.task_id)

    @property
    def rng_run(self):
        return int(str(SEED_INI) + str(self.seed_run))   # INPUT:136

    @property
    def case(self):
        if self.track < 0:                               # INPUT:143-144
            return f"{self.Charger_ID}_rule_based_{julia_float(self.track, f32=False) if self.track != int(self.track) else int(self.track)}"
        f = julia_float
        if self.template == "tuned":                     # TUNED:155
            return (f"{self.Charger_ID}_dw{f(self.DISCOMFORT_WEIGHT_EV, False)}_p{f(self.penalty, False)}_B{self.BATCH_SIZE}_M{self.MEM_SIZE}_"
                    f"{self.noise_type}-o{f(self.sigma)}_th{f(self.theta)}_Y{f(self.gamma)}_tau{f(self.tau)}_lract{f(self.eta_act)}_"
                    f"lrcrit{f(self.eta_crit)}_nact{f(self.noise_act)}_ntrg{f(self.noise_trg)}")
        dw = self.DISCOMFORT_WEIGHT_EV                   # INPUT:146 (Int 2 / Float64 0.5 in that template)
        return (f"{self.Charger_ID}_disw{dw if isinstance(dw, int) else f(dw, False)}_pen{f(self.penalty, False)}_BATCH{self.BATCH_SIZE}_"
                f"MEM{self.MEM_SIZE}_{self.noise_type}-noise_om{f(self.sigma)}_th{f(self.theta)}_Y{f(self.gamma)}_tau{f(self.tau)}_"
                f"nact{f(self.eta_act)}_ncrit{f(self.eta_crit)}_smart-trainEP")


def set_hyperparameters(cfg: RunConfig):
    """`set_hyperparameters(Job_ID)`: the last two digits of JOB_ID in base 3 pick one of three alternatives per hyper-parameter."""
    code = int(cfg.job_id[-2:])
    if cfg.template == "tuned":                          # TUNED:62-106: 4 ternary digits -> BATCH, noise_act, (L1, L2), (eta_act, eta_crit)
        alt = {1: (120, 100, 150), 2: (0.1, 0.2, 0.3), 3: ((300, 600), (200, 400), (250, 500)), 4: ((1e-5, 1e-4), (5e-4, 5e-3), (1e-4, 1e-3))}
        cfg.L1, cfg.L2 = alt[3][0]
        cfg.eta_act, cfg.eta_crit = alt[4][0]
        cfg.BATCH_SIZE, cfg.noise_act = alt[1][0], alt[2][0]
        cfg.gamma, cfg.tau, cfg.sigma, cfg.theta = 0.99, 1e-3, 0.1, 0.15
        cfg.DISCOMFORT_WEIGHT_EV, cfg.penalty, cfg.NUM_EP, cfg.MEM_SIZE, cfg.noise_type, cfg.noise_trg = 0.01, 0.1, 1001, 24000, "gn", 0.2
        digits = np.base_repr(code, 3).zfill(4)
        if len(digits) > 4:
            raise ValueError(f"JOB_ID suffix {code} has more than four ternary digits")
        for i, ch in enumerate(digits, start=1):
            d = int(ch)
            if i == 4:
                cfg.eta_act, cfg.eta_crit = alt[4][d]
            elif i == 3:
                cfg.L1, cfg.L2 = alt[3][d]
            elif i == 2:
                cfg.noise_act = alt[2][d]
            else:
                cfg.BATCH_SIZE = alt[1][d]
        cfg.num_seeds = 40                               # TUNED:55
    elif cfg.template == "input":                        # INPUT:58-100: 3 ternary digits -> MEM_SIZE, BATCH_SIZE, (L1, L2, gamma, sigma, theta)
        alt = {1: (30000, 20000, 24000), 2: (200, 50, 120),
               3: ((150, 300, 0.99, 0.1, 0.15), (300, 600, 0.999, 0.1, 0.15), (300, 600, 0.99, 0.2, 0.2))}
        cfg.L1, cfg.L2, cfg.gamma, cfg.sigma, cfg.theta = alt[3][0]
        cfg.tau, cfg.eta_act, cfg.eta_crit = 1e-3, 1e-4, 1e-3
        cfg.DISCOMFORT_WEIGHT_EV, cfg.penalty, cfg.NUM_EP, cfg.noise_type = 2, 0.5, 101, "ou"
        cfg.BATCH_SIZE, cfg.MEM_SIZE = alt[2][0], alt[1][0]
        digits = np.base_repr(code, 3).zfill(3)
        for i, ch in enumerate(digits, start=1):
            d = int(ch)
            if i == 3:
                cfg.L1, cfg.L2, cfg.gamma, cfg.sigma, cfg.theta = alt[3][d]
            elif i == 2:
                cfg.BATCH_SIZE = alt[2][d]
            else:
                cfg.MEM_SIZE = alt[1][d]
        cfg.noise_act, cfg.noise_trg = 0.1, 0.2          # INPUT:230-231
        cfg.num_seeds = 2                                # INPUT:52
    else:
        raise ValueError(f"unknown input template {cfg.template!r} (tuned | input)")
    return cfg


def config_from_env(environ=os.environ):
    """MAIN:10-14 + INPUT:34-52.  SHEMS_* variables are this build's additions (test-sized runs, batch width, template)."""
    for k in ("JOB_ID", "TASK_ID", "GPU_ID"):
        if k not in environ:
            raise KeyError(f"{k} is not set (the reference reads ENV[\"{k}\"], DDPG_reinforce_charger_v1.jl:10-14 / input.jl:34-36)")
    cfg = RunConfig(job_id=str(environ["JOB_ID"]), task_id=str(environ["TASK_ID"]), gpu_id=int(environ["GPU_ID"]),
                    template=environ.get("SHEMS_INPUT_TEMPLATE", "tuned"))
    set_hyperparameters(cfg)
    for name, cast in (("NUM_EP", int), ("num_seeds", int), ("test_every", int), ("test_runs", int), ("track", float), ("train", int),
                       ("run", str)):
        v = environ.get("SHEMS_" + name.upper())
        if v is not None:
            setattr(cfg, name, cast(v))
    return cfg


def _check_supported(cfg):
    # (250, 500): the tuned kernels; smaller: zero-padded into them (ddpg.pad_net); larger -- the grids' (300, 600) --: the layer-by-layer
    # wide path (csrc/shems_wide.hip, ddpg.is_wide)
    if not (1 <= cfg.L1 <= 4096 and 1 <= cfg.L2 <= 4096):
        raise NotImplementedError(f"JOB_ID {cfg.job_id} selects (L1, L2) = ({cfg.L1}, {cfg.L2})")
    if not 1 <= cfg.BATCH_SIZE <= 1024:             # above 128: replay() runs the gradient passes per sub-batch (ddpg.Agent._replay_wide)
        raise NotImplementedError(f"JOB_ID {cfg.job_id} selects BATCH_SIZE = {cfg.BATCH_SIZE}")
    if cfg.noise_type not in ("gn", "ou", "en", "pn"):
        raise NotImplementedError(f"JOB_ID {cfg.job_id} selects noise_type = {cfg.noise_type!r} (DDPG.jl:152-161 knows gn, ou, en, pn)")
    if cfg.noise_type == "pn":
        # parameter noise adds one scalar to EVERY parameter (DDPG.jl:89-96): it would un-zero the padding a smaller network runs with,
        # and its adaptation step reads the minibatch of ONE update pass (128 columns) -- refuse here, before populate_memory, not at the
        # first replay()
        from .ddpg import L1 as _L1, L2 as _L2, is_wide
        if (cfg.L1, cfg.L2) != (_L1, _L2) and not is_wide((cfg.L1, cfg.L2)):
            raise NotImplementedError(f"JOB_ID {cfg.job_id} selects parameter noise with (L1, L2) = ({cfg.L1}, {cfg.L2}): a network smaller than "
                                      f"({_L1}, {_L2}) runs zero-padded, which parameter noise would break")
        if cfg.BATCH_SIZE > 128:
            raise NotImplementedError(f"JOB_ID {cfg.job_id} selects parameter noise with BATCH_SIZE = {cfg.BATCH_SIZE} > 128 (adapt_param_noise! on sub-batches)")


def data_path(cfg, split, data_dir="data", charger=None):
    return os.path.join(data_dir, f"{charger or cfg.Charger_ID}_{cfg.season}_{split}_{cfg.price}.csv")      # INPUT:162-164


def main(environ=os.environ, cwd=".", log=print):
    import torch
    from . import checkpoint, harness, tables
    from . import ddpg as D
    from .env import ShemsBatch, make_config

    cfg = config_from_env(environ)
    _check_supported(cfg)
    os.chdir(cwd)
    torch.cuda.set_device(cfg.gpu_id)                                    # CUDA.device!(gpu_id), MAIN:12-14
    if cfg.seed_run == 1:
        log(f"Using bash scheduler.\n\tMax steps: {EP_LENGTH['train']} | Max episodes: {cfg.NUM_EP} | Layer 1: {cfg.L1} nodes | "
            f"Layer 2: {cfg.L2} nodes |\n\tCase: {cfg.case}")
    log(f"Starting script with JOB_ID: {cfg.job_id}, TASK_ID: {cfg.task_id} for charger {cfg.Charger_ID} on GPU: {cfg.gpu_id}!")

    if environ.get("SHEMS_SYNTHETIC_DATA") == "1":                       # demo / tests: the per-charger CSVs are not public (README.md:12)
        os.makedirs("data", exist_ok=True)
        for split in ("train", "eval", "test"):
            p = data_path(cfg, split)
            if not os.path.exists(p):
                t = tables.profile_table(cfg.charger_id, split)
                need = 1 + (EP_LENGTH["train"] if split == "train" else EP_LENGTH[cfg.season, split])    # a pass of n steps reads row n + 1
                if t.shape[0] < need:                                      # a real series with one row per MPC decision: see tables.pad_rows
                    log(f"{os.path.basename(p)}: {t.shape[0]} rows in the reconstructed series, {need} needed; padded with the rows 24 h earlier")
                    t = tables.pad_rows(t, need)
                tables.save_csv(p, t)
    tabs = {split: tables.load_csv(data_path(cfg, split)) for split in ("train", "eval", "test")}       # env_dict, INPUT:162-164
    n_envs = int(environ.get("SHEMS_NUM_ENVS", "1"))
    # the env's reward weights are module constants of shems_LU1.jl:40-43 (0.01f0, 2f0, 0.1f0) whatever the input file says ("REMEMBER TO
    # ADJUST THIS IN ENV", INPUT:80-81): the input's values only enter the `case` string
    mk = lambda n, steps, tab: ShemsBatch(n, steps, [tab], [make_config(cfg.charger_id, 0, tab.shape[0])], device=cfg.gpu_id).use_torch_stream()
    env_train = mk(n_envs, EP_LENGTH["train"], tabs["train"])
    env_eval = mk(cfg.test_runs, EP_LENGTH[cfg.season, "eval"], tabs["eval"])
    env_track = mk(1, EP_LENGTH[cfg.season, cfg.run], tabs[cfg.run])

    D.GAMMA, D.TAU = cfg.gamma, cfg.tau
    agent = D.Agent(seed=cfg.rng_run, sigma=cfg.noise_act if cfg.noise_type == "gn" else cfg.sigma, noise_type=cfg.noise_type, theta=cfg.theta,
                    hidden=(cfg.L1, cfg.L2))
    agent.gamma, agent.tau, agent.batch = float(np.float32(cfg.gamma)), float(np.float32(cfg.tau)), cfg.BATCH_SIZE
    agent.eta_act, agent.eta_crit = float(np.float32(cfg.eta_act)), float(np.float32(cfg.eta_crit))
    ring = D.ReplayRing(cfg.MEM_SIZE)
    ck = dict(ep_len=EP_LENGTH["train"], num_ep=cfg.NUM_EP, l1=cfg.L1, l2=cfg.L2, case=cfg.case)

    # ---- Memory Buffer (MAIN:27-30) ----
    agent.populate_memory(env_train, ring, seed=cfg.rng_run)
    agent.min_max_buffer(ring, cfg.MEM_SIZE, seed=cfg.rng_run)

    noise_mean = np.zeros(cfg.NUM_EP, np.float32)
    best_eval = 0
    if cfg.train:
        t0 = time.time()
        log(f", Training run: {cfg.rng_run}")

        def on_best(i, actor, total_reward, score_mean):                 # saveBSON(...; idx=i, path="temp", rng=rng_run), DDPG.jl:282-286
            checkpoint.save(actor, total_reward, score_mean, i, agent.noise_mean, idx=i, rng=cfg.rng_run, path="temp", **ck)

        total_reward, score_mean, best_eval, _ = agent.run_episodes(env_train, env_eval, ring, cfg.NUM_EP, test_every=cfg.test_every,
                                                                    test_runs=cfg.test_runs, seed=cfg.rng_run, on_best=on_best)
        noise_mean = agent.noise_mean
        checkpoint.save(agent.export_actor(), total_reward, score_mean, best_eval, noise_mean, idx=cfg.NUM_EP, rng=cfg.rng_run, **ck)   # MAIN:45-46
        log(f"trained {cfg.NUM_EP} episodes in {time.time() - t0:.1f} s; best evaluation at episode {best_eval}")

    # ---- track evaluation (MAIN:87-110) ----
    written = []
    tk = dict(num_ep=cfg.NUM_EP, l1=cfg.L1, l2=cfg.L2, batch_size=cfg.BATCH_SIZE, mem_size=cfg.MEM_SIZE, min_exp_size=cfg.MEM_SIZE,
              season=cfg.season, run=cfg.run, job_id=cfg.job_id, case=cfg.case)
    if cfg.track == 1 and cfg.seed_run == cfg.num_seeds:
        log(f"Evaluation/Testing for TASK_IDs of {cfg.job_id}.")
        # every seed's last and best actor (MAIN:90-103) -> ONE launch: pass p of the batch runs the whole data set with actor p
        # (harness.inference_many / shems_track_dev); the files are then written in the reference's order.  s_min / s_max are this
        # process's, as in the reference (module globals of the evaluating task).
        passes = []                                                      # (test_rng_run, best, idx, actor)
        for i in range(1, cfg.num_seeds + 1):
            test_rng_run = int(str(SEED_INI) + str(i))
            try:
                ac, _, _, best_i, _ = checkpoint.load(idx=cfg.NUM_EP, rng=test_rng_run, **ck)
            except FileNotFoundError:
                log(f"  seed {i}: no snapshot of run {test_rng_run} (that task has not finished): skipped")
                continue
            passes.append((test_rng_run, False, cfg.NUM_EP, ac))
            passes.append((test_rng_run, True, best_i, checkpoint.load(idx=best_i, rng=test_rng_run, path="temp", **ck)[0]))
        if passes:
            env_many = mk(len(passes), EP_LENGTH[cfg.season, cfg.run], tabs[cfg.run])
            hid = (cfg.L1, cfg.L2)                                       # checkpoints hold the network's own size: pad into the kernels' layout
            lay = (lambda a: np.asarray(a, np.float32)) if D.is_wide(hid) else (lambda a: D.pad_net(a, 9, 2, hid))      # (a wide one keeps its own)
            _, results = harness.inference_many(env_many, np.stack([lay(p[3]) for p in passes]), agent.s_min, agent.s_max, hidden=hid)
            env_many.close()
            for (test_rng_run, best, idx, _), res in zip(passes, results):
                path = harness.results_file_name(cfg.job_id, cfg.run, EP_LENGTH["train"], cfg.NUM_EP, cfg.L1, cfg.L2, cfg.case, test_rng_run,
                                                 cfg.NUM_EP, best=best)
                harness.write_to_results_file(res, path)
                harness.write_to_tracker_file(path, seed=test_rng_run, best=best, idx=idx, **tk)
                written.append(path)
        log(f"Evaluation/Testing for TASK_IDs of {cfg.job_id} is finished.")
    elif cfg.track < 0:                                                  # rule-based
        _, results = harness.inference(env_track, None, track=cfg.track)
        idx = cfg.track if cfg.track != int(cfg.track) else int(cfg.track)
        path = harness.results_file_name(cfg.job_id, cfg.run, EP_LENGTH["train"], cfg.NUM_EP, cfg.L1, cfg.L2, cfg.case, cfg.track, idx)
        harness.write_to_results_file(results, path)
        harness.write_to_tracker_file(path, seed=idx, best=False, idx=idx, **tk)
        written.append(path)
    for e in (env_train, env_eval, env_track):
        e.close()
    log(f"Script with JOB_ID: {cfg.job_id} & TASK_ID: {cfg.task_id} is done!")
    return cfg, written


if __name__ == "__main__":
    main()
