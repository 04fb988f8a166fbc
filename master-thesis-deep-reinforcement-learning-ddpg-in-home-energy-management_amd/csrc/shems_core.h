// shems_core.h -- per-env arithmetic of the batched SHEMS environment (device functions).
//
// One call = one household for one hour: the body of the reference's
//   action(env, a::ShemsAction)  shems_LU1.jl:283-316
//   action(env, track)           shems_LU1.jl:318-340
//   step!(env, s, a; track)      shems_LU1.jl:343-485   (+ next_state! :264-281)
// re-derived for a GPU thread.  The reference mixes Int, Float32 and Float64 (module globals
// `b.rate_max::Float64`, Market fields Float64, `zeros(n)` Float64 defaults, Float32 state), and
// which precision an expression runs in depends on the branch taken.  To reproduce its results
// bit for bit each flow quantity is carried as `Q` = (value, is64): a Float32 quantity is held
// exactly in a double, and add/sub/mul/div pick the f32 or the f64 instruction from the operand
// tags -- the tags are compile-time constants on almost every path, so the optimiser folds them.
// Both f32 and f64 run at full vector rate on CDNA4, and the kernel is latency/bandwidth bound.
//
// MUST be compiled with -ffp-contract=off (hipcc defaults to fast contraction, which would fuse
// a*b+c into an FMA the reference does not perform).
#pragma once

#include <stdint.h>
#include "../../include/shems_hip.h"

#if defined(__HIPCC__) || defined(__HIP__)
#define SHEMS_HD __host__ __device__ __forceinline__
#else
#define SHEMS_HD inline
#endif

#pragma clang fp contract(off)

namespace shems {

// Battery / PV / EV / Market constants, shems_LU1.jl:92-99.
constexpr float  kPvEta     = 1.0f;        // pv.eta
constexpr float  kBEta      = 0.95f;       // b.eta
constexpr float  kBLoss     = 0.00003f;    // b.loss
constexpr float  kEvRateMax = 11.0f;       // ev.rate_max
constexpr double kSell      = (double)0.2f;   // m.sell_discount = Float64(0.2f0)

struct Q {            // a Julia number: Float32 (is64 = false, value exactly representable) or Float64
    double v;
    bool   is64;
};
SHEMS_HD Q q32(float x)  { return Q{(double)x, false}; }
SHEMS_HD Q q64(double x) { return Q{x, true}; }
SHEMS_HD Q qadd(Q a, Q b) { return (a.is64 || b.is64) ? Q{a.v + b.v, true} : Q{(double)((float)a.v + (float)b.v), false}; }
SHEMS_HD Q qsub(Q a, Q b) { return (a.is64 || b.is64) ? Q{a.v - b.v, true} : Q{(double)((float)a.v - (float)b.v), false}; }
SHEMS_HD Q qmul(Q a, Q b) { return (a.is64 || b.is64) ? Q{a.v * b.v, true} : Q{(double)((float)a.v * (float)b.v), false}; }
SHEMS_HD Q qdiv(Q a, Q b) { return (a.is64 || b.is64) ? Q{a.v / b.v, true} : Q{(double)((float)a.v / (float)b.v), false}; }

// Base.clamp(x, lo, hi) = ifelse(x > hi, hi, ifelse(x < lo, lo, x)); NOT min(max()) -- when lo > hi
// the answer is hi for x > hi and lo otherwise.
SHEMS_HD double jl_clamp(double x, double lo, double hi) { return x > hi ? hi : (x < lo ? lo : x); }
SHEMS_HD double jl_min(double a, double b) { return b < a ? b : a; }
SHEMS_HD float  jl_minf(float a, float b)  { return b < a ? b : a; }

struct EnvIn {        // destructured env.state (LU1:284, 319, 344)
    float Soc_b, Soc_ev, c_ev, d_e, g_e, p_buy;
};

// action(env, a::ShemsAction), LU1:283-316.  Returns Float32.([B, EV]).
SHEMS_HD void action_drl(const shems_config &c, const EnvIn &s, float B_target, float EV_target,
                         float &B_out, float &EV_out)
{
    const float span = c.soc_max - 0.0f;                       // b.soc_max - b.soc_min
    const float Soc_b_perc = (s.Soc_b - 0.0f) / span;           // :288
    float EV;
    if (s.c_ev > -1.0f && s.Soc_ev < EV_target)                // :292
        EV = jl_minf(kEvRateMax, (EV_target - s.Soc_ev) * (c.cap_ev - 0.0f));
    else
        EV = 0.0f;                                             // Int 0; g_e - d_e - 0 is unchanged
    const float pv_ = (s.g_e - s.d_e) - EV;                     // :301
    double B;
    if (pv_ > 0.0f && Soc_b_perc < B_target) {                 // :304
        const float B_target_value = B_target * span + 0.0f;   // :306 (no FMA)
        const float room = (B_target_value - s.Soc_b) + kBLoss;
        B = jl_clamp((double)pv_, 0.0, jl_min(c.rate_max, (double)room));   // :307 -> Float64
    } else if (s.Soc_b > 1e-3f) {                              // :309
        const float t = (1.0f - kBLoss) * s.Soc_b;
        B = -jl_min(c.rate_max, (double)t);                    // :310 -> Float64
    } else {
        B = 0.0;
    }
    B_out = (float)B;                                          // :315 Float32.([B, EV])
    EV_out = EV;
}

// action(env, track), LU1:318-340 (rule-based "power mode").
SHEMS_HD void action_rule(const shems_config &c, const EnvIn &s, float &B_out, float &EV_out)
{
    const float EV = jl_minf(kEvRateMax, (1.0f - s.Soc_ev) * (c.cap_ev - 0.0f));   // :323
    const float pv_ = (s.g_e - s.d_e) - EV;                                         // :327
    double B;
    if (pv_ > 0.0f && (double)s.Soc_b < (0.95 * (double)c.soc_max)) {               // :330 (0.95 is Float64)
        const float room = (c.soc_max - s.Soc_b) + kBLoss;
        B = jl_clamp((double)pv_, 0.0, jl_min(c.rate_max, (double)room));           // :331
    } else if (s.Soc_b > 1e-3f) {
        const float t = (1.0f - kBLoss) * s.Soc_b;
        B = -jl_min(c.rate_max, (double)t);
    } else {
        B = 0.0;
    }
    B_out = (float)B;
    EV_out = EV;
}

struct StepFlows {    // everything the 23-column results row needs (LU1:476-478)
    double PV_DE, B_DE, GR_DE, PV_B, PV_GR, PV_EV, B_EV, GR_EV, EX_EV;
    double profit, discomfort, penalty;
};

// Float64 ^ Float64 (LU1:467, 470).  openlibm returns x*x for y == 2 and x for y == 1 exactly.
SHEMS_HD double jl_pow(double x, double y)
{
    if (y == 2.0) return x * x;
    if (y == 1.0) return x;
#if defined(__HIP_DEVICE_COMPILE__)
    return ::pow(x, y);
#else
    return __builtin_pow(x, y);
#endif
}

// Flows + next Soc_b / Soc_ev + comfort + reward of step!, LU1:356-471, for set-points (B, EV)
// already rounded to Float32.  `rule_mode` <=> track < 0.  Soc_ev_out is the value after the
// comfort block (LU1:442-446); the caller applies next_state!'s arrival overwrite (LU1:270-272).
SHEMS_HD void step_flows(const shems_config &c, const EnvIn &s, float EV_target, float B, float EV,
                         bool rule_mode, float &Soc_b_out, float &Soc_ev_out, double &reward,
                         StepFlows &f)
{
    const Q eta = q32(kBEta);
    const Q qEV = q32(EV);
    const Q z64 = q64(0.0);                                    // zeros(8), zeros(11): Float64 defaults
    Q PV_DE = z64, PV_B = z64, PV_EV = z64, B_DE = z64, B_EV = z64, GR_DE = z64, GR_EV = z64;
    bool GR_B_is64 = true;                                     // GR_B is 0.0 (default) or Int 0 (:420)
    double BD = 0.0;
    Q pv_ = z64;
    bool pv_is_int = false;                                    // `pv_ = 0` assigns an Int

    if ((double)B < -0.01) {                                   // :362
        const float k = (1.0f - kBLoss) - 1e-7f;               // (1 - b.loss - 1f-7), two f32 subtractions
        const float lim = k * s.Soc_b;
        BD = jl_clamp((double)(-B), 0.001, jl_min(c.rate_max, (double)lim));   // :363
    }

    const float gpe = s.g_e * kPvEta;
    if (gpe > s.d_e) {                                         // :368  PV covers the demand
        PV_DE = q32(s.d_e);
        const float pvf = gpe - s.d_e;
        if (pvf > EV) {
            PV_EV = qEV;
            pv_ = q32(pvf - EV);
        } else if (pvf <= EV) {
            PV_EV = q32(pvf);
            pv_ = q32(0.0f); pv_is_int = true;
            const float rest = EV - pvf;                       // (EV - PV_EV), Float32
            const float need = rest / kBEta;
            if (BD > (double)need) {                           // :377
                B_EV = q32(rest);
                BD = BD - (double)need;                        // BD -= B_EV / b.eta  (f32 quotient)
            } else if (BD <= (double)need) {
                B_EV = q64(BD * (double)kBEta);
                BD = 0.0;
                GR_EV = qsub(q32(rest), B_EV);
            }
        }
    } else if (gpe <= s.d_e) {                                 // :388  PV short of the demand
        PV_DE = q32(gpe);
        pv_ = q32(0.0f); pv_is_int = true;
        const float de = s.d_e - gpe;                          // d_e -= PV_DE
        const float need = de / kBEta;
        if (BD > (double)need) {                               // :392
            B_DE = q32(de);
            BD = BD - (double)need;
            const float need2 = EV / kBEta;
            if (BD > (double)need2) {
                B_EV = qEV;
                BD = BD - (double)need2;
            } else if (BD <= (double)need2) {
                B_EV = q64(BD * (double)kBEta);
                BD = 0.0;
                GR_EV = qsub(qEV, B_EV);
            }
        } else if (BD <= (double)need) {                       // :403
            B_DE = q64(BD * (double)kBEta);
            BD = 0.0;
            GR_DE = qsub(q32(de), B_DE);
            GR_EV = qEV;
        }
    }

    if ((double)B > 0.01) {                                    // :412  battery charging, from PV only
        const float room = c.soc_max - s.Soc_b;
        const double BC = jl_clamp((double)B, 0.001, jl_min(c.rate_max, (double)room));
        const double need = BC / (double)kBEta;
        if (pv_.v > need) {
            PV_B = q64(BC);
            pv_ = q64(pv_.v - need); pv_is_int = false;
        } else if (pv_.v <= need) {
            PV_B = qmul(pv_, eta);                             // Float32 (or Int 0 * f32) unless pv_ is the f64 default
            pv_ = q32(0.0f); pv_is_int = true;
            GR_B_is64 = false;                                 // GR_B = 0 (Int)
        }
    }
    const Q PV_GR = pv_;                                       // :424
    (void)pv_is_int;

    // :432  Soc_b' = (1 - b.loss) * (Soc_b + PV_B + GR_B - ((B_DE + B_EV + B_GR) / b.eta))
    Q acc = qadd(q32(s.Soc_b), PV_B);
    if (GR_B_is64) acc = qadd(acc, z64);                       // + 0.0 promotes to Float64; + Int 0 does not
    const Q drawn = qdiv(qadd(B_DE, B_EV), eta);               // B_GR = 0 (Int) changes neither value nor type
    const Q nb = qmul(q32(1.0f - kBLoss), qsub(acc, drawn));
    Soc_b_out = (float)nb.v;
    // :435  Soc_ev' = Soc_ev + (PV_EV + B_EV + GR_EV) / (ev.soc_max - ev.soc_min)
    const Q ne = qadd(q32(s.Soc_ev), qdiv(qadd(qadd(PV_EV, B_EV), GR_EV), q32(c.cap_ev - 0.0f)));
    float soc_ev_n = (float)ne.v;

    float discomfort = 0.0f, penalty = 0.0f, EX_EV = 0.0f;      // :438-440
    if (s.c_ev == 0.0f && soc_ev_n < 1.0f) {                    // :442 departure below 100 %
        discomfort = (1.0f - soc_ev_n) * 100.0f;
        EX_EV = (1.0f - soc_ev_n) * (c.cap_ev - 0.0f);
        soc_ev_n = 1.0f;
    } else if (s.c_ev < 0.0f && (double)EV_target < 0.99) {     // :447 absent EV, target below 99 %
        penalty = (1.0f - EV_target) * c.penalty_weight;
    }
    Soc_ev_out = soc_ev_n;

    // :464  profit = (sell * p_buy * (PV_GR + B_GR)) - (p_buy * (GR_DE + GR_B + GR_EV + EX_EV))   (Float64)
    const double pb = (double)s.p_buy;
    const double grid = ((GR_DE.v + 0.0) + GR_EV.v) + (double)EX_EV;
    const double profit = (kSell * pb) * PV_GR.v - pb * grid;
    const double disc = c.disc_weight * jl_pow((double)discomfort, c.disc_pot);
    if (rule_mode) {                                            // :466-468
        reward = profit - disc;
        penalty = 0.0f;
    } else {
        reward = (profit - disc) - (double)penalty;
    }
    f.PV_DE = PV_DE.v; f.B_DE = B_DE.v; f.GR_DE = GR_DE.v; f.PV_B = PV_B.v; f.PV_GR = PV_GR.v;
    f.PV_EV = PV_EV.v; f.B_EV = B_EV.v; f.GR_EV = GR_EV.v; f.EX_EV = (double)EX_EV;
    f.profit = profit; f.discomfort = (double)discomfort; f.penalty = (double)penalty;
}

// scale_action, DDPG.jl:178-184 with ACTION_BOUND_LO = (0f0, 0f0), HI = (1f0, 1f0):
// Float32(LO + (a + 1.0) * 0.5 * (HI - LO)) evaluated in Float64 (ones() is Float64).
SHEMS_HD float scale_action(float a)
{
    return (float)(0.0 + (((double)a + 1.0) * 0.5) * (double)(1.0f - 0.0f));
}

// Episode-start extension loop of reset_state!, LU1:225-246, for the first draw idx0 (1-based).
// h(idx1) returns h_countdown of 1-based row idx1.  Returns -1 on an out-of-range access.
template <class HFn>
SHEMS_HD int32_t resolve_start(int32_t idx0, int32_t nrow, int32_t maxsteps, HFn h)
{
    const int32_t hi = nrow - maxsteps;
    if (idx0 < 1 || idx0 + maxsteps > nrow) return -1;
    int32_t idx = idx0;
    float c_ev_end = h(idx + maxsteps);
    int counter = 0;
    while (c_ev_end > -1.0f && idx < hi) {
        idx += (int32_t)(c_ev_end + 1.0f);          // Int(c_ev_end + 1)
        if (idx > hi) idx = idx0;                    // redraw with the same seed = the same draw
        c_ev_end = h(idx + maxsteps);
        counter += 1;
        if (counter > 100) break;                    // "exceeded maximum iterations ... Breaking"
    }
    return idx;
}

}  // namespace shems
