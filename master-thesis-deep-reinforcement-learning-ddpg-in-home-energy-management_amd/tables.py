"""Exogenous input tables for the batched SHEMS environment (host side).

The reference env re-reads a 21-column CSV on every reset and every step
(shems_LU1.jl:217, 265) and uses 8 of its columns (shems_LU1.jl:251-260, 268-279).
Here a table is packed ONCE into `[nrow][8]` float32 rows

    h_countdown, soc_ev, electkwh, PV_generation, p_buy, hour_cos, hour_sin, season

(each value = Float32(Float64 csv value), i.e. the rounding the reference applies
when it stores a DataFrame cell into its Float32 state) and uploaded to HBM.

The real per-charger CSVs are not public (reference README.md:12), so a seeded
synthetic "Charger98-like" generator is provided (statistics: SURVEY.md App. C,
taken from Data_preparation_v2.ipynb and the Charger98 MPC result file).
"""
from __future__ import annotations

import math

import numpy as np

NCOL = 8
COL_H, COL_SOCEV, COL_DE, COL_GE, COL_PBUY, COL_HCOS, COL_HSIN, COL_SEASON = range(NCOL)
COLUMNS = ("h_countdown", "soc_ev", "electkwh", "PV_generation", "p_buy", "hour_cos", "hour_sin", "season")

# rows per split (Data_preparation_v2.ipynb cell 36) and episode lengths (input.jl EP_LENGTH)
SPLIT_ROWS = {"train": 4320, "eval": 1440, "test": 3000}
SPLIT_ID = {"train": 0, "eval": 1, "test": 2}

# capacities dict, shems_LU1.jl:47-59: id -> (cap_ev kWh, nominal battery kWh (x 0.9f0 in f32), rate_max kW)
CHARGER_PROFILES = {
    1: (48.250, 7.5, 3.3), 2: (36.271, 10.0, 3.3), 3: (45.508, 10.0, 3.3), 4: (78.993, 11.0, 4.6),
    5: (37.207, 10.0, 4.6), 6: (35.816, 15.0, 4.6), 7: (36.521, 12.0, 3.3), 8: (45.728, 10.0, 3.3),
    9: (21.935, 7.5, 3.3), 98: (35.816, 7.5, 3.3), 97: (78.993, 11.0, 4.6),
}


def pack_columns(h_countdown, soc_ev, electkwh, pv_generation, p_buy, hour_cos, hour_sin, season):
    cols = [np.asarray(c, dtype=np.float64) for c in
            (h_countdown, soc_ev, electkwh, pv_generation, p_buy, hour_cos, hour_sin, season)]
    n = len(cols[0])
    if any(len(c) != n for c in cols):
        raise ValueError("table columns differ in length")
    out = np.empty((n, NCOL), dtype=np.float32)
    for j, c in enumerate(cols):
        out[:, j] = c.astype(np.float32)
    if not np.all(out[:, COL_H] == np.round(out[:, COL_H])):
        # the reference does Int(c_ev_end + 1) (shems_LU1.jl:232) which throws InexactError otherwise
        raise ValueError("h_countdown must be integer valued")
    return out


def load_csv(path):
    """Read a reference-format input CSV (21 columns written by Data_preparation_v2.ipynb
    cell 42; only the 8 columns the env reads are required) into a packed table."""
    import csv

    with open(path, newline="") as fh:
        rd = csv.reader(fh)
        header = [h.strip() for h in next(rd)]
        missing = [c for c in COLUMNS if c not in header]
        if missing:
            raise KeyError(f"{path}: missing column(s) {missing}")   # Julia: ArgumentError/KeyError on df[:, :col]
        pos = [header.index(c) for c in COLUMNS]
        rows = [[float(r[p]) for p in pos] for r in rd if r]
    a = np.asarray(rows, dtype=np.float64)
    return pack_columns(*[a[:, j] for j in range(NCOL)])


def save_csv(path, table):
    with open(path, "w") as fh:
        fh.write(",".join(COLUMNS) + "\n")
        for r in np.asarray(table, dtype=np.float32):
            fh.write(",".join("%.9g" % float(v) for v in r) + "\n")   # 9 digits round-trip float32


# ----------------------------------------------------------------------------
# Seeded synthetic generator (integer-hash driven: no dependence on NumPy's RNG
# implementation, so every box regenerates the identical table).
# ----------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


class _Stream:
    def __init__(self, seed, lane):
        self.s = _splitmix64((seed * 0x100000001B3 + lane) & _M64)

    def u(self):
        """uniform in (0, 1)"""
        self.s = _splitmix64(self.s)
        return ((self.s >> 11) + 0.5) / float(1 << 53)

    def n(self):
        """standard normal (Box-Muller)"""
        u1, u2 = self.u(), self.u()
        return math.sqrt(-2.0 * math.log(u1)) * math.cos(2.0 * math.pi * u2)


def _season_of_month(m):
    # Data_preparation_v2.ipynb cell 17: 1 spring (Mar-May), 2 summer, 3 autumn, 4 winter
    return 1 if m in (3, 4, 5) else 2 if m in (6, 7, 8) else 3 if m in (9, 10, 11) else 4


def synthetic_table(split="train", charger_id=98, seed=None, nrow=None):
    """Charger98-like hourly series.  Deterministic in (split, charger_id, seed)."""
    nrow = SPLIT_ROWS[split] if nrow is None else int(nrow)
    seed = (charger_id * 100 + SPLIT_ID[split]) if seed is None else int(seed)
    cap_ev = CHARGER_PROFILES[charger_id][0]
    days_per_month = {"train": 15, "eval": 5, "test": 10}[split]
    pv_scale = 1.0 + 0.05 * ((charger_id * 7) % 5 - 2)       # mild per-profile variety

    hour = np.arange(nrow) % 24
    day = np.arange(nrow) // 24
    month_idx = np.minimum(day // days_per_month, 11)
    month = (10 + month_idx) % 12 + 1                        # series starts in November
    season = np.array([_season_of_month(int(m)) for m in month], dtype=np.float64)
    hour_cos = np.cos(2.0 * np.pi * hour / 23.0)             # cell 15: divides by maximum(hour) = 23
    hour_sin = np.sin(2.0 * np.pi * hour / 23.0)

    sd, sg, se = _Stream(seed, 1), _Stream(seed, 2), _Stream(seed, 3)
    d_e = np.empty(nrow)
    g_e = np.empty(nrow)
    cloud_day = 1.0
    for t in range(nrow):
        h = int(hour[t])
        bump = 1.0 + 0.9 * math.exp(-0.5 * ((h - 19.0) / 2.5) ** 2) + 0.3 * math.exp(-0.5 * ((h - 8.0) / 1.5) ** 2)
        d = 0.2 + math.exp(-0.45 + 0.75 * sd.n()) * bump
        d_e[t] = min(max(d, 0.195), 8.5)
        if h == 0:
            cloud_day = 0.25 + 0.75 * sg.u()
        m = int(month[t])
        amp = pv_scale * (14.0 + 8.0 * math.cos(2.0 * math.pi * (m - 6.5) / 12.0))   # ~22 midsummer, ~6 midwinter
        sun = max(0.0, math.sin(math.pi * (h - 6.0) / 12.0))
        g_e[t] = amp * sun * cloud_day * (0.7 + 0.3 * sg.u())
    d_e = np.round(d_e, 3)
    g_e = np.round(g_e, 3)

    # EV sessions: ~35 % of hours connected, countdown at arrival 1..71 h (median ~19)
    h_cd = -np.ones(nrow)
    soc = np.ones(nrow)
    t = int(6 + 30 * se.u())
    while t < nrow - 2:
        want_hour = int(round(14.0 + 7.0 * se.n())) % 24
        while t < nrow and int(hour[t]) != want_hour:
            t += 1
        if t >= nrow - 2:
            break
        dur = int(round(math.exp(math.log(19.0) + 0.85 * se.n())))
        dur = min(max(dur, 1), 71)
        dur = min(dur, nrow - 2 - t)
        if dur < 1:
            break
        soc0 = min(max(0.435 + 0.19 * se.n(), 0.03), 1.0)
        soc0 = round(soc0 * cap_ev, 3) / cap_ev                # raw data holds kWh with 3 decimals
        for k in range(dur + 1):
            h_cd[t + k] = dur - k
            if split == "train":                                # cells 40, 45: interpolate towards 1.0 in train only
                soc[t + k] = soc0 + (1.0 - soc0) * (k / float(dur + 1))
            else:
                soc[t + k] = soc0 if k == 0 else 1.0
        # row after a 0 is forced to -1 with soc_ev = 1 (cells 39, 45) -- already the default
        t = t + dur + 2 + int(-math.log(se.u()) * 34.0)
    p_buy = np.full(nrow, 0.4)
    return pack_columns(h_cd, soc, d_e, g_e, p_buy, hour_cos, hour_sin, season)


_SERIES = None


def real_series_keys():
    """("ChargerNN_split", ...) of the exogenous series shipped in data/mpc_series.npz."""
    global _SERIES
    if _SERIES is None:
        import os
        with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "mpc_series.npz"), allow_pickle=False) as z:
            _SERIES = {k: np.ascontiguousarray(z[k], dtype=np.float32) for k in z.files}
    return tuple(sorted(_SERIES))


def real_series(charger_id, split="train"):
    """The real exogenous series of a charger profile, [nrow][8] float32, or None when the reference holds none.

    The per-charger input CSVs are not public (reference README.md:12), but the reference commits the MPC benchmark's result
    files for Chargers 01/03/04/05/08/09 (train, 4 319 rows), 04/05/09 (eval, 1 439) and 01/03/06/08/09/98 (test, 2 999)
    under `SHEMS python/single_building/results/`; the 8 columns the env reads are recovered from the LP's balance
    constraints by tests/golden/make_fixtures.py (data, not source text).  One row shorter than the CSVs (the MPC horizon)."""
    real_series_keys()
    t = _SERIES.get(f"Charger{int(charger_id):02d}_{split}")
    return None if t is None else t.copy()


def profile_table(charger_id, split="train", prefer_real=True):
    """Real series where the reference holds one, the seeded synthetic generator otherwise."""
    t = real_series(charger_id, split) if prefer_real else None
    return synthetic_table(split, charger_id) if t is None else t


def pad_rows(table, nrow):
    """A table with at least `nrow` rows: missing rows are copies of the row 24 h earlier (same hour-of-day features).  The series
    reconstructed from the reference's MPC result files hold one row per DECISION (eval: 1 439, test: 2 999), while a pass of that
    many steps reads one row more (next_state! looks at row idx + 1, LU1:264-281): the last hour's exogenous data is not in the files."""
    t = np.asarray(table, dtype=np.float32)
    while t.shape[0] < nrow:
        t = np.concatenate([t, t[-24:-23] if t.shape[0] >= 24 else t[-1:]], 0)
    return t


def episode_start_table(table, maxsteps):
    """Resolved episode start for every possible first draw (pure function of the draw:
    shems_LU1.jl:227-246 redraws with the SAME seed, i.e. the same value).  Host-side helper
    for analysis; the device kernel runs the loop itself.  Returns int32 [nrow-maxsteps] (1-based)."""
    h = np.asarray(table)[:, COL_H]
    nrow = len(h)
    hi = nrow - maxsteps
    out = np.zeros(max(hi, 0), dtype=np.int32)
    for idx0 in range(1, hi + 1):
        idx = idx0
        c_end = h[idx + maxsteps - 1]
        counter = 0
        while c_end > -1 and idx < hi:
            idx += int(c_end + 1)
            if idx > hi:
                idx = idx0
            c_end = h[idx + maxsteps - 1]
            counter += 1
            if counter > 100:
                break
        out[idx0 - 1] = idx
    return out
