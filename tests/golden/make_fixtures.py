"""Regenerates the committed fixtures under tests/golden/.  Run in the BUILD container only
(`python tests/golden/make_fixtures.py`): part (1) reads a data file of the reference.

(1) charger98_test_reconstructed.csv -- the Charger98 *test* exogenous series, reconstructed from the
    reference's own MPC result file (a real data file, not an LFS stub)
        SHEMS python/single_building/results/260724_results_2999_2999_0-2999_1_5_10.0_all_test_fix_Charger98.csv
    via the LP's balance constraints (SHEMS_optimizer_cost.py:55-57):
        electkwh = PV_DE + B_DE + GR_DE,  PV_generation = PV_DE + PV_B + PV_GR + PV_EV,
        h_countdown = C_EV,  soc_ev = Soc_Ev / 35.816 on arrival rows (else 1),
        hour_cos/sin = cos/sin(2*pi*hour/23), season from month, p_buy = 0.4   (SURVEY.md App. C).
    This is input DATA (8 numeric columns), no reference source text.
(1b) <package>/data/mpc_series.npz (read through tables.real_series; `bench.py --mixed` uses it) -- the same reconstruction for EVERY MPC result file the reference holds that is not an LFS stub
    (results/260724_results_*_all_{train,eval,test}_fix_ChargerNN.csv: Chargers 01/03/04/05/08/09 train (4 319 rows),
    04/05/09 eval (1 439), 01/03/06/08/09/98 test (2 999)), one packed [nrow][8] float32 table per key
    "ChargerNN_split", with each charger's own cap_ev (shems_LU1.jl:47-59) and, for the train split, the linear
    soc_ev interpolation that Data_preparation_v2.ipynb cell 40 applies to training data only.
(2) kat_appendix_b.json -- the hand-traced known-answer vectors of SURVEY.md Appendix B.
(3) oracle_golden.npz -- outputs of the CPU oracle (C restatement) on fixed inputs: a rule-based
    72-step episode (BASELINE config 1) and 64 DRL envs x 72 steps with hashed actions.  These
    pin regressions of the oracle itself; they were NOT produced by the Julia reference
    (no Julia here: "parity unpinned by the reference").
"""
import csv
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

REF = "/root/reference/SHEMS python/single_building/results/260724_results_2999_2999_0-2999_1_5_10.0_all_test_fix_Charger98.csv"


def season_of_month(m):
    return 1 if m in (3, 4, 5) else 2 if m in (6, 7, 8) else 3 if m in (9, 10, 11) else 4


def reconstruct(path=REF, cap_ev=35.816, split="test"):
    rows = list(csv.DictReader(open(path)))
    out = []
    prev_c = -1.0
    for r in rows:
        g = lambda k: float(r[k])
        c = g("C_EV")
        d_e = g("PV_DE") + g("B_DE") + g("GR_DE")
        g_e = g("PV_DE") + g("PV_B") + g("PV_GR") + g("PV_EV")
        soc = g("Soc_Ev") / cap_ev if (c > -1 and prev_c == -1) else 1.0
        hour, month = g("hour"), int(g("month"))
        out.append([c, min(soc, 1.0), round(d_e, 3), round(g_e, 3), 0.4,
                    math.cos(2 * math.pi * hour / 23.0), math.sin(2 * math.pi * hour / 23.0),
                    float(season_of_month(month))])
        prev_c = c
    a = np.asarray(out, dtype=np.float64)
    if split == "train":
        interpolate_soc_ev(a)
    return a


def interpolate_soc_ev(a):
    """Training tables only (Data_preparation_v2.ipynb cells 40, 45): soc_ev rises linearly from its arrival value to 1.0 at the
    row whose countdown is 0.  A session starts where h_countdown > 0 follows a -1 row (or opens the table)."""
    h, soc = a[:, 0], a[:, 1]
    start = None
    for i in range(len(h)):
        if h[i] > 0 and (i == 0 or h[i - 1] == -1):
            start = i
        if h[i] == 0 and start is not None:
            s0 = soc[start]
            for j in range(start, i + 1):
                soc[j] = s0 + (1.0 - s0) * (j - start) / (i - start)
            start = None


RESULT_DIR = "/root/reference/SHEMS python/single_building/results"


def reconstruct_all(T):
    """Every non-LFS MPC result file -> {"ChargerNN_split": [nrow][8] float32}."""
    import glob
    import re
    out = {}
    for path in sorted(glob.glob(os.path.join(RESULT_DIR, "260724_results_*_all_*_fix_Charger*.csv"))):
        m = re.search(r"_all_(train|eval|test)_fix_Charger(\d\d)\.csv$", path)
        split, cid = m.group(1), int(m.group(2))
        a = reconstruct(path, T.CHARGER_PROFILES[cid][0], split)
        h = a[:, 0]
        assert ((h[1:][h[:-1] == 0]) == -1).all(), "a departure row must be followed by -1 (cell 39)"
        assert (a[:, 1] >= 0).all() and (a[:, 1] <= 1 + 1e-12).all() and (a[:, 2] >= 0).all() and (a[:, 3] >= -1e-9).all()
        out[f"Charger{cid:02d}_{split}"] = T.pack_columns(*[a[:, j] for j in range(8)])
    return out


def main():
    import util as U
    T = U.tables_mod()
    if os.path.exists(REF):
        a = reconstruct()
        tab = T.pack_columns(*[a[:, j] for j in range(8)])
        T.save_csv(os.path.join(HERE, "charger98_test_reconstructed.csv"), tab)
        print("reconstructed table", tab.shape)
        series = reconstruct_all(T)
        assert np.array_equal(series["Charger98_test"], tab)
        np.savez_compressed(os.path.join(ROOT, U.PKG_NAME, "data", "mpc_series.npz"), **series)     # shipped as package DATA
        print("mpc_series.npz:", {k: v.shape[0] for k, v in series.items()})
    else:
        print("reference data file absent: keeping the committed reconstructed table")

    kats = {
        "_source": "SURVEY.md Appendix B (hand-traced from shems_LU1.jl with explicit f32/f64 typing; NOT produced by Julia)",
        "_profile": "Charger98: cap_ev 35.816, soc_max 7.5f0*0.9f0, rate_max 3.3; p_buy 0.4",
        "cases": [
            dict(name="K1", state=[3.375, 1.0, -1, 2.128, 0.0], a=[0.5, 1.0], mode=0, h_cur=-1, next=[0.24, 0, -1, 1],
                 B=-3.3, EV=0, reward=0.0, soc_b_hex="3f914691", soc_ev=1.0),
            dict(name="K2", state=[1.0, 1.0, -1, 0.281, 5.026], a=[0.8, 1.0], mode=0, h_cur=-1, next=[0.379, 5.132, -1, 1],
                 B=3.3, EV=0, reward=0.10170525756202553, soc_b_hex="4089988b", soc_ev=1.0),
            dict(name="K3", state=[4.0, 0.4, 5, 1.2, 0.0], a=[0.0, 0.9], mode=0, h_cur=5, next=[1.0, 0, 4, 0.52],
                 B=-3.3, EV=11, reward=-3.6260001080898863, soc_b_hex="3f3331d4", soc_ev_hex="3f35062a"),
            dict(name="K4", state=[0.0, 0.7, 0, 0.5, 0.0], a=[0.5, 0.75], mode=0, h_cur=0, next=[0.6, 0.3, -1, 1],
                 B=0, EV=1.7908005714, reward=-10.747920345111496, soc_b_hex="00000000", soc_ev=1.0,
                 profit=-4.4979204848098817, discomfort=25.0),
            dict(name="K5", state=[0.0, 1.0, -1, 0.5, 0.2], a=[0.5, 0.3], mode=0, h_cur=-1, next=[0.6, 0.3, -1, 1],
                 B=0, EV=0, reward=-0.19000000685453422, soc_b_hex="00000000", soc_ev=1.0,
                 profit=-0.120000006556511, penalty=0.07000000029802322),
            dict(name="K6", state=[2.0, 0.5, 10, 0.8, 3.0], a=None, mode=-1, h_cur=10, next=[0.7, 3.5, 9, 0.6],
                 B=-1.999940037727356, EV=11.0, reward=-2.7600230032206134, soc_b_hex="387c7e10", soc_ev_hex="3f4e9fc3"),
            dict(name="K7", state=[6.75, 1.0, -1, 0.388, 1.456], a=[1.0, 1.0], mode=0, h_cur=-1,
                 next=[1.261, 0, 47, 20.315789 / 35.816],
                 B=-3.3, EV=0, reward=0.085439999265670696, soc_b_hex="40d7fe58", soc_ev_hex="3f1135c4"),
        ],
    }
    json.dump(kats, open(os.path.join(HERE, "kat_appendix_b.json"), "w"), indent=1)

    # (3) oracle goldens
    import oracle_c
    tab = T.synthetic_table("train", 98)
    prof = oracle_c.profile(98)
    b = oracle_c.Batch(1, 72, tab, prof)
    total, res = b.rule_episode(0, 72, want_results=True)
    n = 64
    b2 = oracle_c.Batch(n, 72, tab, prof)
    idx0 = 1 + (np.arange(n) * 61) % (tab.shape[0] - 72)
    soc0 = ((np.arange(n) * 37 % 100) / 100.0 * 6.75).astype(np.float32)
    b2.reset(False, idx0, soc0)
    start_idx = b2.idx()
    rew = np.zeros((72, n))
    for t in range(72):
        k = np.arange(n) * 72 + t
        act = np.stack([((k * 2654435761) % 1000) / 999.0, ((k * 40503 + 7) % 1000) / 999.0], 1).astype(np.float32)
        rc, r, obs, _ = b2.step(act, 0)
        assert rc == 0
        rew[t] = r
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), rule_total=total, rule_results=res,
                        drl_idx0=idx0, drl_soc0=soc0, drl_start_idx=start_idx, drl_rewards=rew,
                        drl_final_obs=b2.state(), table_sha=np.frombuffer(
                            __import__("hashlib").sha256(tab.tobytes()).digest(), np.uint8))
    print("rule episode total", total)


if __name__ == "__main__":
    main()
