"""The thesis protocol at its real width, end to end (not a benchmark): 40 seeds x 10 charger profiles = 400 independent DDPG learners
(RL-SHEMS_bs_scheduler_1179_08_on_01-98.sh:67-87; learner l trains on charger profile l mod 10), each on 128 households, trained by the
grouped launches -- fused act/step for all 51 200 households + the throughput form of the grouped replay() (csrc/shems_gupd.hip) -- for
argv[2] episodes of 72 hours, then every learner's deterministic evaluation score on its own charger's eval table (100 starts, 72
hours), next to the rule-based controller on the same starts.  argv[3] = "latency" runs the same protocol on the five-launch form
(fewer episodes advised).  argv[5] = households per learner (default 128; any multiple of 32: with argv[4] = 1 only household 0 feeds the learner, so 32 -- the smallest tile of
the fused kernel -- is the closest this framework comes to the reference's ONE household per learner).
Writes one JSON document to argv[1]."""
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

out_path = sys.argv[1] if len(sys.argv) > 1 else "group_protocol.json"
episodes = int(sys.argv[2]) if len(sys.argv) > 2 else 120
form = sys.argv[3] if len(sys.argv) > 3 else "throughput"
window = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] not in ("", "default") else None
SEEDS = 40
E = int(sys.argv[5]) if len(sys.argv) > 5 else 128
ids = (1, 2, 3, 4, 5, 6, 7, 8, 9, 98)
L = SEEDS * len(ids)
tabs = [S.tables.synthetic_table("train", c) for c in ids]
row0 = np.cumsum([0] + [t.shape[0] for t in tabs])
cfgs = [S.make_config(c, row0[k], tabs[k].shape[0]) for k, c in enumerate(ids)]
co = ((np.arange(L * E) // E) % len(ids)).astype(np.uint16)
env = S.ShemsBatch(L * E, 72, tabs, cfgs, co).use_torch_stream()
grp = G.LearnerGroup(L, E, seed=1231, rng_seed=99, form=form)
grp.populate_memory(env)
grp.min_max_buffer()
t0 = time.perf_counter()
first = last = None
for ep in range(1, episodes + 1):
    ret = grp.episode_(env, train=True, rng_ep=7, episode=ep, window_count=window).view(L, E).mean(1).cpu().numpy()
    first = ret if ep == 1 else first
    last = ret
torch.cuda.synchronize()
wall = time.perf_counter() - t0
grp.flux_()             # (tiled working layout: the learners' Flux-order tensors are made current before anything reads them)
finite = bool(torch.isfinite(grp.slab[:, :grp.layout["ws"][0]]).all())
# evaluation: every learner on its own charger's eval table; the rule-based controller on the same starts
scores, rule = np.zeros(L), {}
for k, cid in enumerate(ids):
    ev = S.tables.synthetic_table("eval", cid)
    env_eval = S.ShemsBatch(100, 1439, [ev], [S.make_config(cid, 0, ev.shape[0])]).use_torch_stream()
    env_eval.reset_(123, episode=1)
    rule[cid] = float(env_eval.rollout("rule", 72).mean().item())
    for l in range(k, L, len(ids)):
        scores[l] = float(grp.learners[l].episode_(env_eval, None, train=False, num_steps=72, rng_ep=123, episode=1).mean().item())
    env_eval.close()
per_charger = {}
for k, cid in enumerate(ids):
    sc = scores[k::len(ids)]
    per_charger[str(cid)] = {"rule_based": rule[cid], "learners": SEEDS, "score_mean": float(sc.mean()), "score_best": float(sc.max()),
                             "score_worst": float(sc.min()), "beat_rule_based": int((sc > rule[cid]).sum())}
wc = grp.ring_window(72, window)[0]
doc = {"protocol": f"40 seeds x 10 chargers = 400 learners x {E} households, grouped launches", "households_per_learner": E, "form": grp.form, "episodes": episodes,
       "remembered_transitions_per_learner_update": wc,
       "update_to_data": ("1 update per remembered transition: the reference's ratio (DDPG.jl:229-233)" if wc == 1 else
                          f"1 update per {wc} remembered transitions ({wc} x the reference's data per update)"),
       "transitions_remembered_per_learner": episodes * 72 * wc,
       "updates_per_learner": grp.updates, "learner_updates_total": grp.updates * L, "env_steps": episodes * 72 * L * E, "wall_s": wall,
       "learner_updates_per_s": grp.updates * L / wall, "state_finite": finite,
       "train_return_first_mean": float(first.mean()), "train_return_last_mean": float(last.mean()), "per_charger": per_charger}
json.dump(doc, open(out_path, "w"), indent=1)
print(json.dumps({k: doc[k] for k in ("form", "episodes", "wall_s", "learner_updates_per_s", "state_finite", "train_return_first_mean", "train_return_last_mean")}))
print(json.dumps({c: (round(v["rule_based"], 1), round(v["score_mean"], 1), round(v["score_best"], 1), v["beat_rule_based"]) for c, v in per_charger.items()}))
