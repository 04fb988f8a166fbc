"""End-to-end demonstration run (not a benchmark): populate_memory -> min_max_buffer -> run_episodes on 4 096 households, with
the deterministic evaluation sweep every `test_every` episodes, then the same for a learner group of 8 independent seeds.
Writes one JSON document (learning curves, rule-based reference score) to the path given as argv[1].
argv[3] = "L1xL2" (e.g. 300x600) trains the single learner with those hidden sizes (wider than (250, 500): the layer-by-layer path; the learner
group, which runs the tuned kernels only, is skipped then).  argv[4] = households (default 4 096; 65 536 = BASELINE config 3 as written:
`run_episodes` with NUM_EP = 1001, DDPG.jl:244-298 -- the learner group is skipped for any other size than 4 096), argv[5] = evaluation
cadence in episodes (default 25; the reference's test_every is 100, input09_08_on_01-09_eval.jl:64-91)."""
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

PKG = "master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"
S = importlib.import_module(PKG)
D = importlib.import_module(PKG + ".ddpg")
G = importlib.import_module(PKG + ".group")

out_path = sys.argv[1] if len(sys.argv) > 1 else "learning_curve.json"
num_ep = int(sys.argv[2]) if len(sys.argv) > 2 else 300
hidden = tuple(int(x) for x in sys.argv[3].split("x")) if len(sys.argv) > 3 else (250, 500)
n_envs = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
test_every = int(sys.argv[5]) if len(sys.argv) > 5 else 25
tab, ev = S.tables.synthetic_table("train", 98), S.tables.synthetic_table("eval", 98)
env = S.ShemsBatch(n_envs, 72, [tab], [S.make_config(98, 0, tab.shape[0])]).use_torch_stream()
env_eval = S.ShemsBatch(100, 1439, [ev], [S.make_config(98, 0, ev.shape[0])]).use_torch_stream()
env_eval.reset_(123, episode=1)
rule = env_eval.rollout("rule", 72).mean().item()

ag = D.Agent(seed=1231, hidden=hidden)
ring = D.ReplayRing(D.MEM_SIZE)
ag.populate_memory(env, ring)
ag.min_max_buffer(ring)
curve, snaps = [], []
t0 = time.perf_counter()
tr, sm, best_run, best_actor = ag.run_episodes(env, env_eval, ring, num_ep=num_ep, test_every=test_every, test_runs=100,
                                               on_eval=lambda i, r, s: curve.append({"episode": i, "train_return": float(r), "eval_score": float(s)}),
                                               on_best=lambda i, a, trw, smn: snaps.append(int(i)))
torch.cuda.synchronize()
import zlib
crc = zlib.crc32(ag.actor.detach().cpu().numpy().tobytes()) ^ zlib.crc32(ag.critic_t.detach().cpu().numpy().tobytes())
single = {"hidden": list(hidden), "episodes": num_ep, "envs": n_envs, "test_every": test_every, "updates": ag.updates, "env_steps": num_ep * 72 * n_envs,
          "wall_s": time.perf_counter() - t0, "best_run": int(best_run), "best_actor_snapshots_at": snaps,
          "best_actor_crc32": zlib.crc32(np.asarray(best_actor, np.float32).tobytes()) if best_actor is not None else None,
          "learner_crc32": crc, "noise_mean_first10": float(ag.noise_mean[:10].mean()), "noise_mean_abs_max": float(np.abs(ag.noise_mean).max()),
          "score_mean": [float(x) for x in sm],
          "curve": curve, "train_return_first10": float(tr[:10].mean()), "train_return_last10": float(tr[-10:].mean())}

if D.is_wide(hidden) or n_envs != 4096:
    json.dump({"rule_based_eval_score": rule, "single_learner": single}, open(out_path, "w"), indent=1)
    print(json.dumps({"rule": rule, "hidden": list(hidden), "envs": n_envs, "learner_crc32": crc, "eval_first": curve[0]["eval_score"], "eval_last": curve[-1]["eval_score"],
                      "best": max(c["eval_score"] for c in curve), "wall_s": single["wall_s"], "updates": ag.updates}))
    sys.exit(0)
# learner group: 8 independent seeds x 512 households, same protocol, evaluation per learner at the end
L, E = 8, 512
envg = S.ShemsBatch(L * E, 72, [tab], [S.make_config(98, 0, tab.shape[0])]).use_torch_stream()
grp = G.LearnerGroup(L, E, seed=1231, rng_seed=99)
grp.populate_memory(envg)
grp.min_max_buffer()
t0 = time.perf_counter()
first = last = None
gep = min(num_ep, 120)
for ep in range(1, gep + 1):
    ret = grp.episode_(envg, train=True, rng_ep=7, episode=ep).view(L, E).mean(1).cpu().numpy()
    first = ret if ep == 1 else first
    last = ret
torch.cuda.synchronize()
wall = time.perf_counter() - t0
grp.flux_()
scores = [float(a.episode_(env_eval, None, train=False, num_steps=72, rng_ep=123, episode=1).mean().item()) for a in grp.learners]
group = {"learners": L, "envs_per_learner": E, "episodes": gep, "updates_per_learner": grp.updates, "wall_s": wall,
         "train_return_first": [float(x) for x in first], "train_return_last": [float(x) for x in last], "eval_score_per_learner": scores}
json.dump({"rule_based_eval_score": rule, "single_learner": single, "learner_group": group}, open(out_path, "w"), indent=1)
print(json.dumps({"rule": rule, "eval_first": curve[0]["eval_score"], "eval_last": curve[-1]["eval_score"], "best": max(c["eval_score"] for c in curve),
                  "group_scores": scores}))
