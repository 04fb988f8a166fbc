"""Actor / score snapshots: the reference's saveBSON / loadBSON (memory_plotting_saving.jl:263-281).

The reference stores `actor` (a Flux Chain moved to the CPU) and the score arrays in two BSON files named
    out/bson/[temp/]DDPG_Shems_Charger_v1_<EP>_<NUM_EP>_<L1>_<L2>_<case>_<rng>_{actor,scores}_<idx>.bson
Same names, same container here: the documents are written the way BSON.jl lowers a `Chain(Dense, Dense, Dense)` and the score
variables (bson_chain.py -- "parity unpinned": every .bson shipped with the reference is a git-LFS stub, so the layout is restated
from BSON.jl's rules and has not met a Julia-written file).  `fmt="npz"` keeps the round-1 NumPy container for tools that want it.
"""
from __future__ import annotations

import os

import numpy as np

from . import bson_chain

N_ACTOR = 129002


def stem(ep_len, num_ep, l1, l2, case, rng, out_dir="out/bson", path=""):
    return os.path.join(out_dir, path, f"DDPG_Shems_Charger_v1_{ep_len}_{num_ep}_{l1}_{l2}_{case}_{rng}")


def save(actor, total_reward, score_mean, best_run, noise_mean, *, idx, ep_len=72, num_ep=1001, l1=250, l2=500, case="",
         rng=0, out_dir="out/bson", path="", fmt="bson"):
    """saveBSON(actor, total_reward, score_mean, best_run, noise_mean; idx, path, rng)."""
    st = stem(ep_len, num_ep, l1, l2, case, rng, out_dir, path)
    os.makedirs(os.path.dirname(st), exist_ok=True)
    a = np.asarray(actor.detach().cpu().numpy() if hasattr(actor, "detach") else actor, np.float32).reshape(-1)
    want = 9 * l1 + l1 + l1 * l2 + l2 + l2 * 2 + 2              # the (9 -> l1 -> l2 -> 2) chain the file name announces
    if a.size != want:
        raise ValueError(f"actor must hold {want} parameters for (L1, L2) = ({l1}, {l2}) (Flux order W1 b1 W2 b2 W3 b3); got {a.size} "
                         "(a learner of a smaller network exports its own size: Agent.export_actor)")
    if fmt == "bson":
        bson_chain.save_chain(f"{st}_actor_{idx}.bson", a, 9, 2, "tanh", key="actor", hidden=(l1, l2))
        bson_chain.save_scores(f"{st}_scores_{idx}.bson", total_reward, score_mean, best_run, noise_mean)
    else:
        np.savez(f"{st}_actor_{idx}.npz", actor=a, layout=np.array("Flux.params order; W = [in][out] C view of Julia out x in"))
        np.savez(f"{st}_scores_{idx}.npz", total_reward=np.asarray(total_reward, np.float32), score_mean=np.asarray(score_mean, np.float64),
                 best_run=np.int64(best_run), noise_mean=np.asarray(noise_mean, np.float32))
    return st


def load(*, idx, scores_only=False, ep_len=72, num_ep=1001, l1=250, l2=500, case="", rng=0, out_dir="out/bson", path=""):
    """loadBSON(; idx, scores_only, path, rng) -> (actor,) total_reward, score_mean, best_run, noise_mean."""
    st = stem(ep_len, num_ep, l1, l2, case, rng, out_dir, path)
    if os.path.exists(f"{st}_scores_{idx}.bson"):
        tr, sm, br, nm = bson_chain.load_scores(f"{st}_scores_{idx}.bson")
        scores = (np.asarray(tr, np.float32), np.asarray(sm, np.float64), int(br), np.asarray(nm, np.float32))
        if scores_only:
            return scores
        return (bson_chain.load_chain(f"{st}_actor_{idx}.bson", key="actor", hidden=(l1, l2)),) + scores
    with np.load(f"{st}_scores_{idx}.npz", allow_pickle=False) as z:
        scores = (z["total_reward"], z["score_mean"], int(z["best_run"]), z["noise_mean"])
    if scores_only:
        return scores
    with np.load(f"{st}_actor_{idx}.npz", allow_pickle=False) as z:
        actor = z["actor"].astype(np.float32)
    return (actor,) + scores
