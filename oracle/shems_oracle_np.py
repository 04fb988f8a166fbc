"""NumPy twin of oracle/shems_oracle.c -- an INDEPENDENT restatement of shems_LU1.jl.

TEST INFRASTRUCTURE ONLY (see oracle/shems_oracle.h).  "Parity unpinned by the
reference": no Julia here, the reference has no tests and its artefacts are LFS
stubs.  This file exists so that two restatements written in different styles
(C with an explicit Int/Float32/Float64 tag per value vs. NumPy scalars whose
dtype does the promotion) can be cross-checked against each other and against
the hand-traced KATs of SURVEY.md Appendix B.

Conventions (NumPy >= 2, NEP 50):
  * Julia Float32 value      -> np.float32 scalar
  * Julia Float64 literal    -> np.float64 scalar  (Python floats are "weak" in
                                NumPy 2 and would NOT promote: never use them)
  * Julia Int literal        -> Python int (weak, adopts the other operand's dtype,
                                exactly like Julia's Int+Float32 -> Float32)
Follows /root/reference/RL-SHEMS/RL_environments/envs/shems_LU1.jl (LU1).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
f64 = np.float64

NCOL = 8
COL_H, COL_SOCEV, COL_DE, COL_GE, COL_PBUY, COL_HCOS, COL_HSIN, COL_SEASON = range(8)

# LU1:47-59  (cap_ev f32, soc_max = f32*f32, rate_max f64)
CAPACITIES = {
    1: (f32(48.250), f32(7.5) * f32(0.9), f64(3.3)),
    2: (f32(36.271), f32(10) * f32(0.9), f64(3.3)),
    3: (f32(45.508), f32(10) * f32(0.9), f64(3.3)),
    4: (f32(78.993), f32(11) * f32(0.9), f64(4.6)),
    5: (f32(37.207), f32(10) * f32(0.9), f64(4.6)),
    6: (f32(35.816), f32(15) * f32(0.9), f64(4.6)),
    7: (f32(36.521), f32(12) * f32(0.9), f64(3.3)),
    8: (f32(45.728), f32(10) * f32(0.9), f64(3.3)),
    9: (f32(21.935), f32(7.5) * f32(0.9), f64(3.3)),
    98: (f32(35.816), f32(7.5) * f32(0.9), f64(3.3)),
    97: (f32(78.993), f32(11) * f32(0.9), f64(4.6)),
}


class Profile:
    """Module globals pv/b/ev/m of LU1:92-99 for one charger id / sweep point."""

    def __init__(self, charger_id=98, disc_weight=f32(0.01), disc_pot=f32(2.0), penalty_weight=f32(0.1)):
        cap, soc_max, rate = CAPACITIES[charger_id]
        self.pv_eta = f32(1.0)
        self.b_eta = f32(0.95)
        self.b_soc_min = f32(0.0)
        self.b_soc_max = f32(soc_max)
        self.b_rate_max = f64(rate)
        self.b_loss = f32(0.00003)
        self.ev_soc_min = f32(0.0)
        self.ev_soc_max = f32(cap)
        self.ev_rate_max = f32(11.0)
        self.m_sell = f64(f32(0.2))                    # Market fields are Float64
        self.m_disc_w = f64(f32(disc_weight))
        self.m_disc_pot = f64(f32(disc_pot))
        self.penalty_weight = f32(penalty_weight)


def _is_weak(x):
    return isinstance(x, int) and not isinstance(x, (np.generic, bool))


def _ptype(*xs):
    ts = [np.asarray(x).dtype for x in xs if not _is_weak(x)]
    return np.result_type(*ts).type if ts else int


def jl_min(a, b):
    T = _ptype(a, b)
    a_, b_ = T(a), T(b)
    return b_ if b_ < a_ else a_


def jl_clamp(x, lo, hi):
    """Base.clamp: ifelse(x > hi, hi, ifelse(x < lo, lo, x)) on the promoted type."""
    T = _ptype(x, lo, hi)
    if x > hi:
        return T(hi)
    if x < lo:
        return T(lo)
    return T(x)


def jl_pow(x, y):
    x = f64(x)
    if y == 2.0:      # openlibm e_pow.c: y == 2 returns x*x
        return x * x
    if y == 1.0:
        return x
    return f64(np.power(x, y))


class Env:
    """Scalar env, fields as LU1:169-177 (idx is 1-based like Julia)."""

    def __init__(self, maxsteps, table, prof: Profile):
        self.table = np.ascontiguousarray(table, dtype=np.float32)
        assert self.table.ndim == 2 and self.table.shape[1] == NCOL
        self.nrow = self.table.shape[0]
        self.p = prof
        self.state = np.array([0, 0, -1, 0, 0, 0, 1, 0, 1], dtype=np.float32)  # LU1:115
        self.reward = f64(0.0)
        self.a = np.array([0.7, 1.0], dtype=np.float32)                        # LU1:151
        self.step = 0
        self.maxsteps = int(maxsteps)
        self.idx = 1

    def _df(self, idx1, col):
        if idx1 < 1 or idx1 > self.nrow:
            raise IndexError("BoundsError")
        return f32(self.table[idx1 - 1, col])

    # ---- reset!  LU1:206-262 (the two MersenneTwister draws are inputs) ----
    def resolve_start(self, idx0):
        idx = int(idx0)
        c_ev_end = self._df(idx + self.maxsteps, COL_H)
        counter = 0
        while c_ev_end > -1 and idx < (self.nrow - self.maxsteps):
            idx += int(c_ev_end + 1)
            if idx > (self.nrow - self.maxsteps):
                idx = int(idx0)
            c_ev_end = self._df(idx + self.maxsteps, COL_H)
            counter += 1
            if counter > 100:
                break
        return idx, counter

    def reset(self, rng_is_minus1=True, idx0=1, soc_b0=f32(0.0)):
        p = self.p
        if rng_is_minus1:
            self.state[0] = f32(f64(0.5) * (p.b_soc_min + p.b_soc_max))
            idx = 1
        else:
            self.state[0] = f32(soc_b0)
            idx, _ = self.resolve_start(idx0)
        self.state[1] = self._df(idx, COL_SOCEV)
        self.state[2] = self._df(idx, COL_H)
        self.state[3] = self._df(idx, COL_DE)
        self.state[4] = self._df(idx, COL_GE)
        self.state[5] = self._df(idx, COL_PBUY)
        self.state[8] = self._df(idx, COL_SEASON)
        self.state[6] = self._df(idx, COL_HCOS)
        self.state[7] = self._df(idx, COL_HSIN)
        self.reward = f64(0.0)
        self.a = np.array([0.7, 1.0], dtype=np.float32)
        self.step = 0
        self.idx = idx
        return self

    # ---- action(env, a::ShemsAction)  LU1:283-316 ----
    def action_drl(self, B_target, EV_target):
        p = self.p
        Soc_b, Soc_ev, c_ev, d_e, g_e = (f32(v) for v in self.state[:5])
        B_target, EV_target = f32(B_target), f32(EV_target)
        Soc_b_perc = (Soc_b - p.b_soc_min) / (p.b_soc_max - p.b_soc_min)
        if c_ev > -1 and Soc_ev < EV_target:
            EV = jl_min(p.ev_rate_max, (EV_target - Soc_ev) * (p.ev_soc_max - p.ev_soc_min))
        else:
            EV = 0
        pv_ = g_e - d_e - EV
        if pv_ > 0 and Soc_b_perc < B_target:
            B_target_value = B_target * (p.b_soc_max - p.b_soc_min) + p.b_soc_min
            B = jl_clamp(pv_, 0, jl_min(p.b_rate_max, (B_target_value - Soc_b + p.b_loss)))
        elif Soc_b > f32(1e-3):
            B = -jl_min(p.b_rate_max, ((1 - p.b_loss) * Soc_b))
        else:
            B = 0
        return np.array([f32(B), f32(EV)], dtype=np.float32)

    # ---- action(env, track)  LU1:318-340 ----
    def action_rule(self):
        p = self.p
        Soc_b, Soc_ev, c_ev, d_e, g_e = (f32(v) for v in self.state[:5])
        EV = jl_min(p.ev_rate_max, (1 - Soc_ev) * (p.ev_soc_max - p.ev_soc_min))
        pv_ = g_e - d_e - EV
        if pv_ > 0 and Soc_b < (f64(0.95) * p.b_soc_max):
            B = jl_clamp(pv_, 0, jl_min(p.b_rate_max, p.b_soc_max - Soc_b + p.b_loss))
        elif Soc_b > f32(1e-3):
            B = -jl_min(p.b_rate_max, ((1 - p.b_loss) * Soc_b))
        else:
            B = 0
        return np.array([f32(B), f32(EV)], dtype=np.float32)

    # ---- step!(env, s, a; track)  LU1:343-485 ----
    def step_(self, a, track=0):
        p = self.p
        if self.idx + 1 > self.nrow:
            raise IndexError("BoundsError")
        Soc_b, Soc_ev, c_ev, d_e, g_e, p_buy = (f32(v) for v in self.state[:6])
        if track >= 0:
            B_target, EV_target = f32(a[0]), f32(a[1])
            B, EV = self.action_drl(B_target, EV_target)
            B, EV = f32(B), f32(EV)
        else:
            B_target, EV_target = f32(0), f32(0)
            B, EV = f32(a[0]), f32(a[1])
        self.a = np.array([B_target, EV_target], dtype=np.float32)

        z = f64(0.0)
        pv_ = BD = BC = z
        PV_DE = PV_B = PV_EV = B_DE = B_EV = GR_DE = GR_EV = GR_B = z

        if B < f64(-0.01):
            BD = jl_clamp(-B, f64(0.001), jl_min(p.b_rate_max, ((1 - p.b_loss - f32(1e-7)) * Soc_b)))

        if (g_e * p.pv_eta) > d_e:
            PV_DE = d_e
            pv_ = (g_e * p.pv_eta) - PV_DE
            if pv_ > EV:
                PV_EV = EV
                pv_ = pv_ - PV_EV
            elif pv_ <= EV:
                PV_EV = pv_
                pv_ = 0
                if BD > (EV - PV_EV) / p.b_eta:
                    B_EV = (EV - PV_EV)
                    BD = BD - B_EV / p.b_eta
                elif BD <= (EV - PV_EV) / p.b_eta:
                    B_EV = BD * p.b_eta
                    BD = 0
                    GR_EV = (EV - PV_EV) - B_EV
        elif (g_e * p.pv_eta) <= d_e:
            PV_DE = g_e * p.pv_eta
            pv_ = 0
            d_e = d_e - PV_DE
            if BD > (d_e / p.b_eta):
                B_DE = d_e
                BD = BD - B_DE / p.b_eta
                if BD > (EV / p.b_eta):
                    B_EV = EV
                    BD = BD - B_EV / p.b_eta
                elif BD <= (EV / p.b_eta):
                    B_EV = BD * p.b_eta
                    BD = 0
                    GR_EV = EV - B_EV
            elif BD <= (d_e / p.b_eta):
                B_DE = BD * p.b_eta
                BD = 0
                GR_DE = d_e - B_DE
                GR_EV = EV

        if B > f64(0.01):
            BC = jl_clamp(B, f64(0.001), jl_min(p.b_rate_max, p.b_soc_max - Soc_b))
            if pv_ > (BC / p.b_eta):
                PV_B = BC
                pv_ = pv_ - (BC / p.b_eta)
            elif pv_ <= (BC / p.b_eta):
                PV_B = pv_ * p.b_eta
                pv_ = 0
                GR_B = 0

        PV_GR = pv_
        B_GR = 0

        new_soc_b = f32((1 - p.b_loss) * (Soc_b + PV_B + GR_B - ((B_DE + B_EV + B_GR) / p.b_eta)))
        new_soc_ev = f32(Soc_ev + (PV_EV + B_EV + GR_EV) / (p.ev_soc_max - p.ev_soc_min))

        discomfort = 0
        penalty = 0
        EX_EV = 0
        if c_ev == 0 and new_soc_ev < 1:
            discomfort = (1 - new_soc_ev) * 100
            EX_EV = (1 - new_soc_ev) * (p.ev_soc_max - p.ev_soc_min)
            new_soc_ev = f32(1)
        elif c_ev < 0 and EV_target < f64(0.99):
            penalty = (1 - EV_target) * p.penalty_weight

        # next_state!  LU1:264-281
        idx = self.idx + 1
        self.state[2] = self._df(idx, COL_H)
        if self.state[2] >= 0 and self._df(self.idx, COL_H) == -1:
            new_soc_ev = self._df(idx, COL_SOCEV)
        self.state[3] = self._df(idx, COL_DE)
        self.state[4] = self._df(idx, COL_GE)
        self.state[5] = self._df(idx, COL_PBUY)
        self.state[8] = self._df(idx, COL_SEASON)
        self.state[6] = self._df(idx, COL_HCOS)
        self.state[7] = self._df(idx, COL_HSIN)
        self.state[0] = new_soc_b
        self.state[1] = new_soc_ev
        self.step += 1
        self.idx += 1

        profit = (p.m_sell * p_buy * (PV_GR + B_GR)) - (p_buy * (GR_DE + GR_B + GR_EV + EX_EV))
        disc_term = p.m_disc_w * jl_pow(discomfort, p.m_disc_pot)
        if track < 0:
            reward = profit - disc_term
            penalty = 0
        else:
            reward = profit - disc_term - penalty
        self.reward = f64(reward)

        results = np.array([self.idx, c_ev, EV_target, EV, Soc_ev, self.reward, profit, discomfort, penalty,
                            PV_DE, B_DE, GR_DE, PV_B, PV_GR, PV_EV, B_EV, GR_EV, EX_EV, GR_B, B_GR, B,
                            B_target, Soc_b], dtype=np.float64)
        return self.reward, self.state.copy(), results

    def finished(self):
        return False


def scale_action(a):
    """DDPG.jl:178-184 with LO=(0f0,0f0), HI=(1f0,1f0); ones() is Float64."""
    a = np.asarray(a, dtype=np.float32)
    return (f32(0) + (a.astype(np.float64) + f64(1.0)) * f64(0.5) * f64(f32(1) - f32(0))).astype(np.float32)
