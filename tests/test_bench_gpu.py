"""bench.py contract on the GPU box: the JSON line, and a 2-rank rehearsal of the data-parallel flow (both ranks on the one
GPU, gloo instead of RCCL) so that a rank-0-only collective or a mismatched collective count shows up as a hang here and not
on the 8-GPU node."""
import json
import os
import subprocess
import sys

import pytest

import util as U

pytestmark = pytest.mark.gpu


def _run(cmd, env=None, timeout=420):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run(cmd, cwd=U.ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_json_contract_small():
    d = _run([sys.executable, "bench.py", "--steps", "16", "--warmup", "4", "--envs", "8192", "--no-cpu-baseline", "--prewarm-s", "0.2"])
    assert d["prewarm_steps"] >= 50 and d["steps"] == 16 and d["warmup"] == 4          # the pre-warm is extra, untimed and reported
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 16 and d["warmup"] == 4 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["config"]["mode"] == "train" and d["config"]["envs_per_gpu"] == 8192 and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["value"] > 1e6 and d["updates_per_sec"] > 100
    # the line says what ran: ONE kernel name (the dispatcher's), the bytes the launch has to move, what kernel_avg_us is, and where the
    # wall time went (HBM rate / traffic ratio only where committed counters exist for this size: 65 536 envs)
    assert r["kernel"] == "shems::k_actg<1, 4, 2, 2>" and "|" not in r["kernel"]
    assert r["algorithmic_bytes"] == 92 * 8192 + 4 * 129002 + 85 * 333 and "hbm_gbs" in r and "traffic_ratio" in r
    assert "minus replay" in r["kernel_avg_us_is"]
    assert 0.7 * r["kernel_avg_us"] < r["kernel_back_to_back_us"] < 1.5 * r["kernel_avg_us"]      # the direct timing agrees with the difference
    assert d["gpu_section_s"] > 0 and d["roofline_pass_s"] > 0 and d["cpu_baseline_s"] == 0.0
    assert d["overlap"] is False and d["loop"] == "native"
    # round 5: every dominant kernel group is on the line -- the direct timing's fraction next to the in-loop one, and the learner's
    # update with its own (latency-bound, tiny) MFMA fraction and bytes: 307.8 MFLOP, 12.4 MB algorithmic, counter bytes where committed
    assert abs(r["frac_back_to_back"] - r["algorithmic_per_launch"] / (r["kernel_back_to_back_us"] * 1e-6) / 1e12 / 157.3) < 1e-9
    u = r["update_roofline"]
    assert abs(u["update_us"] - d["update_us"]) < 1e-9 and u["launches"] == 5 and abs(u["mflop"] - 307.8) < 1e-9
    assert abs(u["frac"] - 307.8e6 / (u["update_us"] * 1e-6) / 1e12 / 157.3) < 1e-9 and 0.01 < u["frac"] < 0.2
    assert u["algorithmic_bytes"] == 258003 * 40 + 8 * 258003
    assert u["counter_bytes"] is None or (u["counter_bytes"] > u["algorithmic_bytes"] and len(u["traffic_ratio"]) == 2)


def test_bench_also_records():
    """The default run's "also" object (here switched on for a small headline and two of its six entries): compact sub-records of the other
    configurations and of the learner groups, each with value, ms_per_step, steps, updates_per_sec and roofline{kernel, kernel_avg_us, frac}."""
    d = _run([sys.executable, "bench.py", "--steps", "16", "--warmup", "4", "--envs", "8192", "--no-cpu-baseline", "--prewarm-s", "0.2",
              "--also", "on", "--also-which", "config2_4096_envs,group_32x2048"])
    assert d["config"]["envs_per_gpu"] == 8192 and d["value"] > 1e6                      # the headline is untouched
    also = d["also"]
    assert sorted(also) == ["config2_4096_envs", "group_32x2048"] and d["also_s"] > 0
    for name, rec in also.items():
        assert "error" not in rec, rec
        for k in ("workload", "value", "unit", "ms_per_step", "steps", "warmup", "updates_per_sec", "roofline", "setup_s", "wall_s"):
            assert k in rec, (name, k)
        r = rec["roofline"]
        assert r["kernel"] and r["kernel_avg_us"] > 0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert abs(rec["value"] - (4096 if name.startswith("config2") else 65536) * rec["steps"] / (rec["ms_per_step"] * 1e-3 * rec["steps"])) < 1e-3 * rec["value"]
    c2, g = also["config2_4096_envs"], also["group_32x2048"]
    assert c2["roofline"]["kernel"] == "shems::k_actg<1, 4, 2, 3>" and abs(c2["updates_per_sec"] - c2["steps"] / (c2["ms_per_step"] * 1e-3 * c2["steps"])) < 1e-3 * c2["updates_per_sec"]
    assert g["update_form"] == "throughput" and abs(g["updates_per_sec"] - 32 * 1e3 / g["ms_per_step"]) < 1e-3 * g["updates_per_sec"]
    assert g["roofline"]["kernel"].startswith("grouped replay(), throughput form") and g["roofline"]["other_kernel"]["kernel"].startswith("shems::k_act")
    # without the switch a non-default headline carries no sub-records
    d0 = _run([sys.executable, "bench.py", "--steps", "8", "--warmup", "2", "--envs", "4096", "--no-cpu-baseline", "--prewarm-s", "0.1"])
    assert "also" not in d0 and d0["also_s"] == 0.0


def test_bench_group_window_one():
    """--group-window 1: one remembered transition per learner per update (the reference's ratio, DDPG.jl:229-233)."""
    d = _run([sys.executable, "bench.py", "--mode", "group", "--learners", "16", "--envs", "2048", "--steps", "12", "--warmup", "2", "--no-cpu-baseline",
              "--prewarm-s", "0.1", "--group-window", "1"])
    assert d["replay_window_envs_per_step"] == 1 and d["learners"] == 16 and d["updates_per_sec"] > 1600
    # ... on 32 households per learner (the smallest env block: the reference's learner owns ONE household, the other 31 only act)
    d = _run([sys.executable, "bench.py", "--mode", "group", "--learners", "16", "--envs", "512", "--steps", "12", "--warmup", "2", "--no-cpu-baseline",
              "--prewarm-s", "0.1", "--group-window", "1"])
    assert d["replay_window_envs_per_step"] == 1 and d["envs_per_learner"] == 32 and d["updates_per_sec"] > 1600


def test_group_mode_shards_as_plain_replicas_over_ranks():
    """Learner groups shard over GPUs as plain replicas: every rank advances its own learners on its own env shard, no collective on the
    data path; `value` / `updates_per_sec` are whole-job aggregates.  Rehearsal form: two rank processes on device 0, gloo for the
    barriers and the MAX over ranks."""
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--mode", "group", "--learners", "16", "--envs", "2048", "--steps", "12", "--warmup", "2",
              "--prewarm-s", "0.1", "--no-cpu-baseline"], env={"SHEMS_BENCH_ONE_DEVICE": "1", "SHEMS_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["mode"] == "group" and d["learners"] == 16 and d["rccl_ranks"] == 2
    assert abs(d["value"] - 2 * 2048 * 12 / (d["ms_per_step"] * 1e-3 * 12)) < 1e-3 * d["value"]
    assert abs(d["updates_per_sec"] - 2 * 16 * 1e3 / d["ms_per_step"]) < 1e-3 * d["updates_per_sec"]
    assert len({c["pid"] for c in d["rank_census"]}) == 2


def test_bench_kernel_name_follows_the_dispatcher():
    import importlib
    D = importlib.import_module(U.PKG_NAME + ".ddpg")
    assert D.act_kernel_name(65536) == "shems::k_act2" and D.act_kernel_name(4096) == "shems::k_actg<1, 4, 2, 3>"
    assert D.act_kernel_name(8192) == "shems::k_actg<1, 4, 2, 2>" and D.act_kernel_name(65536, grouped=True) == "shems::k_act<4, 4, 2>"


def test_bench_scaled_replay_mode():
    """SURVEY 8(d)'s optional replay mode: ring capacity 72 x envs, every env's transition inserted every step."""
    d = _run([sys.executable, "bench.py", "--steps", "16", "--warmup", "4", "--envs", "4096", "--no-cpu-baseline", "--prewarm-s", "0.1", "--scaled-replay"])
    assert d["mem_size"] == 72 * 4096 and d["replay_mode"].startswith("scaled") and d["replay_window_envs_per_step"] == 4096
    assert d["value"] > 1e6 and d["updates_per_sec"] > 100
    d0 = _run([sys.executable, "bench.py", "--steps", "16", "--warmup", "4", "--envs", "4096", "--no-cpu-baseline", "--prewarm-s", "0.1"])
    assert d0["mem_size"] == 24000 and d0["replay_mode"].startswith("window") and d0["replay_window_envs_per_step"] == 333


def test_bench_mixed_profiles():
    """BASELINE config 5: 10 charger profiles x 6 (discomfort weight, power) points, per-env configs."""
    d = _run([sys.executable, "bench.py", "--steps", "16", "--warmup", "4", "--envs", "8192", "--no-cpu-baseline", "--mixed"])
    assert d["config"]["mixed_profiles"] is True and d["value"] > 1e6


def test_bench_group_mode_with_charger_grid():
    """--mode group: independent learners, learner l on charger profile l mod 10."""
    d = _run([sys.executable, "bench.py", "--mode", "group", "--learners", "16", "--envs", "8192", "--mixed", "--steps", "12", "--warmup", "2",
              "--no-cpu-baseline"])
    assert d["config"]["mode"] == "group" and d["learners"] == 16 and d["envs_per_learner"] == 512
    assert d["updates_per_sec"] > 16 * 100 and d["value"] > 1e6
    # the roofline object describes the DOMINANT kernel of the step: the grouped replay() when it is the larger share
    assert d["update_form"] == "throughput"
    r = d["roofline"]
    if d["group_update_us"] > r.get("other_kernel", {}).get("avg_us", float("inf")):
        assert r["kernel"].startswith("grouped replay(), throughput form") and r["launches_per_update"] == 8
        assert abs(r["kernel_avg_us"] - d["group_update_us"]) < 1e-6 and abs(r["per_learner_update_us"] - d["group_update_us"] / 16) < 1e-9
        assert abs(r["frac"] - 16 * 307.8e6 / (d["group_update_us"] * 1e-6) / 1e12 / 157.3) < 1e-9
        assert r["algorithmic_bytes"] == 16 * (258003 * 40 + 8 * 258003) and r["hbm_frac_of_8tbs"] > 0
        assert r["other_kernel"]["kernel"].startswith("shems::k_act")
    else:
        assert r["kernel"].startswith("shems::k_act")
    dl = _run([sys.executable, "bench.py", "--mode", "group", "--learners", "16", "--envs", "8192", "--steps", "12", "--warmup", "2",
               "--no-cpu-baseline", "--group-form", "latency"])
    assert dl["update_form"] == "latency"


def test_two_rank_data_parallel_rehearsal():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", "29541", "bench.py", "--gpus", "2", "--steps", "24", "--warmup", "4", "--envs", "4096"],
             env={"SHEMS_BENCH_ONE_DEVICE": "1", "SHEMS_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["cpu_baseline"] is None and d["value"] > 0
    _check_data_parallel_fields(d, 2)


def _check_data_parallel_fields(d, world):
    """The N > 1 line must explain itself: who took part, what the exchange cost, what the shard-sized k_act does."""
    assert d["rccl_ranks"] == world and d["collective_backend"] in ("nccl", "gloo")
    cen = d["rank_census"]
    assert sorted(c["rank"] for c in cen) == list(range(world)) and all("uuid" in c and "device_index" in c and "pid" in c for c in cen)
    assert len({c["pid"] for c in cen}) == world                       # one process per rank
    assert d["distinct_devices"] == 1                                   # the rehearsal puts every rank on device 0 -- and the line says so
    dp = d["data_parallel"]
    for k in ("update_us_dp", "update_us_split_local", "allreduce_critic_us", "allreduce_actor_us", "k_act_us_at_shard", "exchange_us_in_update"):
        assert dp[k] is not None and dp[k] == dp[k], k
    assert dp["update_us_dp"] > dp["update_us_split_local"] > 10.0     # the exchange is inside the first and not the second
    assert dp["allreduce_critic_us"] > 0 and dp["allreduce_actor_us"] > 0 and dp["allreduce_bytes"] == [516004, 516008]
    # (gloo on one device is host-synchronous: a 700 us update next to a 25 us k_act -- the difference of two such group times is
    # noise here and may come out below zero; on RCCL the collectives are stream-ordered and it is the k_act time at the shard size)
    assert abs(d["update_us"] - dp["update_us_dp"]) < 1e-6 and abs(d["roofline"]["kernel_avg_us"]) < 1e5      # (seen: 10.2 ms at four gloo ranks on a busy host)
    # how the gradients travel: the line says which path was asked for and which ran.  Round 6: torch.distributed is the DEFAULT at
    # world > 1 (no path of this code has moved a byte between two GPUs yet; the native RCCL-in-stream form and the direct exchange are
    # opt-in, SHEMS_DP=native / direct)
    assert d["dp_requested"] == "torch" and d["dp_exchange"].startswith("torch.distributed") and d["loop"] == "host"
    assert d["replica_crc32_distinct"] == 1                             # every rank's learner ended with the same bytes (gathered in finish())


def test_bench_starts_its_own_ranks_without_torchrun():
    """`python bench.py --gpus 2` (the driver's call: no torchrun, no WORLD_SIZE in the environment) must start two ranks
    itself and report n_gpus == 2.  Rehearsal form for the one-GPU box: both ranks on device 0, gloo instead of RCCL.
    BASELINE config 4's call is `python bench.py --gpus 8 --envs 8192`."""
    env = {"SHEMS_BENCH_ONE_DEVICE": "1", "SHEMS_BENCH_BACKEND": "gloo"}
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        assert k not in os.environ or True
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    e.update(env)
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "24", "--warmup", "4", "--envs", "8192"],
                         cwd=U.ROOT, env=e, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    assert [l for l in out.stdout.splitlines() if l.strip()] == lines, "nothing but the JSON line on stdout (library banners go to stderr)"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["envs_per_gpu"] == 8192 and d["cpu_baseline"] is None and d["value"] > 0
    assert d["updates_per_sec"] > 0 and d["roofline"]["frac"] == d["roofline"]["frac"]      # (present; its sign is noise under gloo, see below)
    _check_data_parallel_fields(d, 2)


def test_async_gradient_exchange_gives_the_same_bytes():
    """Data parallel: the critic's gradient all-reduce issued asynchronously with the actor's E products running under it
    (shems_ddpg_actor_prepare) against everything in program order: the learner must end up bit-identical (2-rank rehearsal on one
    device; gloo's collectives are host-synchronous -- the stream ordering on real RCCL is test_rccl_stream_ordering_on_a_one_rank_group).  Program
    order is the default since the asynchronous form measured slower (tools/dp_one_rank_steps.py)."""
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--envs", "4096", "--prewarm-s", "0"],
             env={"SHEMS_BENCH_ONE_DEVICE": "1", "SHEMS_BENCH_BACKEND": "gloo"})
    assert d["dp_overlap"] is False
    out = []
    for knob in ("1", "0"):
        # --prewarm-s 0: the pre-warm runs for a wall-clock time, i.e. a different number of updates from run to run
        d = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "12", "--warmup", "2", "--envs", "4096", "--prewarm-s", "0"],
                 env={"SHEMS_BENCH_ONE_DEVICE": "1", "SHEMS_BENCH_BACKEND": "gloo", "SHEMS_DP_OVERLAP": knob})
        assert d["n_gpus"] == 2 and d["dp_overlap"] is (knob == "1")
        out.append(d["learner_crc32"])
    assert out[0] == out[1]


def test_direct_gradient_exchange_two_ranks_on_one_device():
    """The direct exchange (SHEMS_DP=direct: peer-mapped inboxes, per-slice epoch flags, the sum in rank order inside the ADAM sweep -- no
    collective launch) rehearsed as far as one GPU allows: two rank processes on device 0 map each other's inboxes with hipIpcOpenMemHandle
    (profiles/r04_ipc_probe.json, tools/ipc_kernel_pingpong.hip: kernels of two processes do run concurrently there and see each other's
    stores).  With two replicas the rank-order sum IS the all-reduce's sum (a + b), so the learner must end with the bytes of the
    torch.distributed (gloo) path; no exchange wait may have given up (TrainWorkload.finish raises otherwise)."""
    out = {}
    for how in ("torch", "direct", "native"):
        d = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "48", "--warmup", "6", "--envs", "4096", "--prewarm-s", "0"],
                 env={"SHEMS_BENCH_ONE_DEVICE": "1", "SHEMS_BENCH_BACKEND": "gloo", "SHEMS_DP": how})
        assert d["n_gpus"] == 2 and d["value"] > 0 and d["dp_requested"] == how
        out[how] = d
    # SHEMS_DP=native on this rehearsal: two ranks on ONE device cannot have an RCCL communicator ("Duplicate GPU detected"), so the attempt
    # must fall back -- on every rank, by vote -- to torch.distributed; the line says what was asked for and what ran
    assert out["native"]["dp_exchange"].startswith("torch.distributed") and out["native"]["loop"] == "host"
    assert out["native"]["learner_crc32"] == out["torch"]["learner_crc32"]
    assert out["direct"]["dp_exchange"].startswith("direct exchange") and out["direct"]["loop"] == "native"
    assert out["torch"]["dp_exchange"].startswith("torch.distributed") and out["torch"]["loop"] == "host"
    assert out["direct"]["learner_crc32"] == out["torch"]["learner_crc32"], (out["direct"]["learner_crc32"], out["torch"]["learner_crc32"])
    assert out["direct"]["replica_crc32_distinct"] == 1 and out["torch"]["replica_crc32_distinct"] == 1
    # with SHEMS_DP_OVERLAP=1 Agent._ddpg_args() sets DEFER_ACTOR_E (K2 leaves the actor's E products to
    # shems_ddpg_actor_prepare).  The native data-parallel step never issues that call: the flag must not reach it (shems_ddpg_update_dp
    # clears it on its copy, the native loop's record carries flags = 0) -- same bytes as the torch path, and the line says program order.
    dov = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "48", "--warmup", "6", "--envs", "4096", "--prewarm-s", "0"],
               env={"SHEMS_BENCH_ONE_DEVICE": "1", "SHEMS_BENCH_BACKEND": "gloo", "SHEMS_DP": "direct", "SHEMS_DP_OVERLAP": "1"})
    assert dov["loop"] == "native" and dov["dp_overlap"] is False and dov["learner_crc32"] == out["torch"]["learner_crc32"]
    # four replicas (four rank processes on the one device): the rank-order sum of four gradients is no longer gloo's order, so only the
    # replicas are held to each other -- all four learners bit-identical after 54 exchanged updates, no wait gave up
    d4 = _run([sys.executable, "bench.py", "--gpus", "4", "--steps", "48", "--warmup", "6", "--envs", "2048", "--prewarm-s", "0"],
              env={"SHEMS_BENCH_ONE_DEVICE": "1", "SHEMS_BENCH_BACKEND": "gloo", "SHEMS_DP": "direct"})
    assert d4["n_gpus"] == 4 and d4["dp_exchange"].startswith("direct exchange") and d4["replica_crc32_distinct"] == 1


def test_direct_exchange_late_peer_waits_or_fails_loudly_on_every_rank():
    """a peer that is late must not produce silently diverged replicas.  tests/dp_direct_late_peer.py: two rank
    processes on device 0, rank 1 stalls on the host in the middle of the run.  (a) the stall (1 s) is shorter than the wait bound
    (5 s, the default): rank 0's sweeps wait in the kernel, both ranks finish with the SAME learner bytes.  (b) the stall (2 s) is
    longer than the bound (set to 300 ms): rank 0's wait gives up and poisons its record -- its enqueued sweeps apply nothing, its next
    call returns SHEMS_ERR_STATE -- and rank 1, whose next exchange never gets rank 0's slice, ends the same way: BOTH ranks fail.
    The script exits non-zero for any third outcome (one rank finishing, different checksums)."""
    script = os.path.join(U.ROOT, "tests", "dp_direct_late_peer.py")
    a = subprocess.run([sys.executable, script, "5000", "1.0"], cwd=U.ROOT, capture_output=True, text=True, timeout=400)
    assert a.returncode == 0 and "LATE ok" in a.stdout, (a.stdout[-500:], a.stderr[-1500:])
    crcs = a.stdout.split("LATE ok")[1].split()
    assert crcs[0] == crcs[1]
    b = subprocess.run([sys.executable, script, "300", "2.0"], cwd=U.ROOT, capture_output=True, text=True, timeout=400)
    assert b.returncode == 0 and "LATE poisoned" in b.stdout, (b.stdout[-500:], b.stderr[-1500:])


def test_driver_command_rehearsal_at_the_widest_world_one_box_allows():
    """The driver's multi-GPU call is `python bench.py --gpus N --steps K --warmup W` (BASELINE config 4: N = 8, --envs 8192).  A one-GPU box
    admits at most six processes on the card at once (this pytest process is one of them), so the rehearsal inside the suite is N = 4
    rank processes on device 0 over gloo; the N = 6 run of the same command outside pytest is profiles/r05_dp_world6_rehearsal.json.
    Every data-parallel field, one census entry per rank, replicas identical."""
    d = _run([sys.executable, "bench.py", "--gpus", "4", "--envs", "8192", "--steps", "20", "--warmup", "5", "--prewarm-s", "0"],
             env={"SHEMS_BENCH_ONE_DEVICE": "1", "SHEMS_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 4 and d["config"]["envs_per_gpu"] == 8192 and len(d["rank_census"]) == 4
    assert abs(d["value"] - 4 * 8192 * 20 / (d["ms_per_step"] * 1e-3 * 20)) < 1e-3 * d["value"]      # whole-job aggregate
    _check_data_parallel_fields(d, 4)


def test_rccl_stream_ordering_on_a_one_rank_group():
    """The data-parallel replay() on real RCCL streams (a one-rank NCCL group: RCCL refuses two ranks on one device): asynchronous critic
    all-reduce with the actor's E products under it + wait(), everything in program order, and no collective at all must leave
    bit-identical learners (tests/dp_rccl_one_rank.py)."""
    out = subprocess.run([sys.executable, os.path.join(U.ROOT, "tests", "dp_rccl_one_rank.py")], cwd=U.ROOT, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["backend"] == "nccl"
    assert d["async_overlap"] == d["in_order"] == d["no_collective"], d
    # round 4: the native exchange (RCCL in the update's own stream, csrc/shems_dp.hip) on a one-rank communicator = a single replica
    assert d["native_communicator"] is True
    assert d["native_host_loop"] == d["native_native_loop"] == d["single_replica"], d
