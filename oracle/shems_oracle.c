/*
 * shems_oracle.c -- CPU restatement of shems_LU1.jl (see shems_oracle.h).
 * TEST INFRASTRUCTURE ONLY; "parity unpinned by the reference" (no Julia, no
 * reference tests, LFS-stub artefacts) -- pinned by SURVEY.md App. B KATs, the
 * NumPy twin and the A.4 invariants.
 *
 * Method: every Julia value on the path is carried as a `jnum` = (value, kind)
 * with kind in {Int, Float32, Float64}.  Binary operators promote exactly like
 * Julia (Int < Float32 < Float64); Float32 arithmetic is performed in C `float`,
 * Float64 arithmetic in C `double`.  Compile with -ffp-contract=off and without
 * fast-math so no FMA contraction or re-association happens.
 */
#include "shems_oracle.h"

#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------ jnum -- */
typedef enum { K_INT = 0, K_F32 = 1, K_F64 = 2 } jkind;
typedef struct { double v; jkind k; } jnum;

static inline jnum J_int(long long i) { jnum r = { (double)i, K_INT }; return r; }
static inline jnum J_f32(float f)     { jnum r = { (double)f, K_F32 }; return r; }
static inline jnum J_f64(double d)    { jnum r = { d, K_F64 }; return r; }
static inline jkind kmax(jkind a, jkind b) { return a > b ? a : b; }

static inline jnum jfinish(double exact_in_kind, jkind k) { jnum r = { exact_in_kind, k }; return r; }

static inline jnum jadd(jnum a, jnum b) {
    jkind k = kmax(a.k, b.k);
    if (k == K_F32) return jfinish((double)((float)a.v + (float)b.v), k);
    return jfinish(a.v + b.v, k);               /* Int+Int stays exact in double */
}
static inline jnum jsub(jnum a, jnum b) {
    jkind k = kmax(a.k, b.k);
    if (k == K_F32) return jfinish((double)((float)a.v - (float)b.v), k);
    return jfinish(a.v - b.v, k);
}
static inline jnum jmul(jnum a, jnum b) {
    jkind k = kmax(a.k, b.k);
    if (k == K_F32) return jfinish((double)((float)a.v * (float)b.v), k);
    return jfinish(a.v * b.v, k);
}
static inline jnum jdiv(jnum a, jnum b) {
    jkind k = kmax(a.k, b.k);
    if (k == K_INT) k = K_F64;                  /* Int / Int -> Float64 in Julia */
    if (k == K_F32) return jfinish((double)((float)a.v / (float)b.v), k);
    return jfinish(a.v / b.v, k);
}
static inline jnum jneg(jnum a) { a.v = -a.v; return a; }
/* Comparisons between Int/Float32/Float64 are exact in Julia; every Float32 and
 * every small Int is exactly representable as a double. */
static inline int jgt(jnum a, jnum b) { return a.v >  b.v; }
static inline int jlt(jnum a, jnum b) { return a.v <  b.v; }
static inline int jle(jnum a, jnum b) { return a.v <= b.v; }
static inline int jeq(jnum a, jnum b) { return a.v == b.v; }
/* min(x, y) after promotion (Base.min for floats; NaN propagates). */
static inline jnum jmin(jnum a, jnum b) {
    jkind k = kmax(a.k, b.k);
    double r;
    if (isnan(a.v)) r = a.v; else if (isnan(b.v)) r = b.v;
    else if (b.v < a.v) r = b.v;
    else if (a.v < b.v) r = a.v;
    else r = signbit(a.v) ? a.v : b.v;          /* min(0.0, -0.0) == -0.0 */
    return jfinish(r, k);
}
/* Base.clamp(x, lo, hi) = ifelse(x > hi, hi, ifelse(x < lo, lo, x)) on the promoted type. */
static inline jnum jclamp(jnum x, jnum lo, jnum hi) {
    jkind k = kmax(x.k, kmax(lo.k, hi.k));
    if (jgt(x, hi)) return jfinish(hi.v, k);
    if (jlt(x, lo)) return jfinish(lo.v, k);
    return jfinish(x.v, k);
}
/* convert(Float32, x) */
static inline float jto_f32(jnum a) { return a.k == K_F32 ? (float)a.v : (float)a.v; }
static inline jnum  jround32(jnum a) { return J_f32((float)a.v); }
static inline double jto_f64(jnum a) { return a.v; }

/* ------------------------------------------------------- module constants -- */
/* LU1:92-99.  pv = PV(1f0); b = Battery(0.95f0, 0f0, soc_max, rate_max, 0.00003f0);
 * ev = ElectricVehicle(0f0, cap, 11f0); m = Market(0.2f0, w, pot) with Float64 fields. */
#define PV_ETA      J_f32(1.0f)
#define B_ETA       J_f32(0.95f)
#define B_SOC_MIN   J_f32(0.0f)
#define B_LOSS      J_f32(0.00003f)
#define EV_SOC_MIN  J_f32(0.0f)
#define EV_RATE_MAX J_f32(11.0f)
#define M_SELL      J_f64((double)0.2f)

int orc_profile_for_charger(int id, orc_profile *out)
{
    /* LU1:47-59: (cap_ev f32, soc_max = f32*f32, rate_max f64) */
    float cap, nom; double rate;
    switch (id) {
    case 1:  cap = 48.250f; nom = 7.5f;  rate = 3.3; break;
    case 2:  cap = 36.271f; nom = 10.f;  rate = 3.3; break;
    case 3:  cap = 45.508f; nom = 10.f;  rate = 3.3; break;
    case 4:  cap = 78.993f; nom = 11.f;  rate = 4.6; break;
    case 5:  cap = 37.207f; nom = 10.f;  rate = 4.6; break;
    case 6:  cap = 35.816f; nom = 15.f;  rate = 4.6; break;
    case 7:  cap = 36.521f; nom = 12.f;  rate = 3.3; break;
    case 8:  cap = 45.728f; nom = 10.f;  rate = 3.3; break;
    case 9:  cap = 21.935f; nom = 7.5f;  rate = 3.3; break;
    case 98: cap = 35.816f; nom = 7.5f;  rate = 3.3; break;
    case 97: cap = 78.993f; nom = 11.f;  rate = 4.6; break;
    default: return -1;
    }
    volatile float prod = nom * 0.9f;          /* f32 product, as written in the Dict */
    out->cap_ev = cap;
    out->soc_max = prod;
    out->rate_max = rate;
    out->disc_weight = (double)0.01f;          /* LU1:40 -> Market Float64 field */
    out->disc_pot = (double)2.0f;              /* LU1:41 */
    out->penalty_weight = 0.1f;                /* LU1:43 */
    return 0;
}

void orc_env_init(orc_env *e, int64_t maxsteps, const float *table, int64_t nrow,
                  const orc_profile *p)
{
    /* Shems(maxsteps, path) = Shems(ShemsState(), 0.0, ShemsAction(), 0, maxsteps, 1, path)  LU1:203
     * ShemsState() = (0,0,-1,0,0,0,1,0,1) LU1:115 ; ShemsAction() = (0.7, 1) LU1:151 */
    static const float s0[ORC_NSTATE] = { 0.f, 0.f, -1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 1.f };
    memcpy(e->state, s0, sizeof s0);
    e->reward = 0.0;
    e->a[0] = 0.7f; e->a[1] = 1.0f;
    e->step = 0; e->maxsteps = maxsteps; e->idx = 1;
    e->table = table; e->nrow = nrow; e->prof = *p;
}

static inline float tab(const orc_env *e, int64_t idx1, int col)
{
    return e->table[(idx1 - 1) * ORC_NCOL + col];
}

int64_t orc_resolve_start(const float *table, int64_t nrow, int64_t maxsteps,
                          int64_t idx0, int *iterations)
{
    /* LU1:225-246 with the two rand() calls replaced by the given draw idx0. */
    int64_t idx = idx0;
    int counter = 0;
    const int max_iterations = 100;
    if (idx + maxsteps < 1 || idx + maxsteps > nrow) return -1;
    float c_ev_end = table[(idx + maxsteps - 1) * ORC_NCOL + ORC_COL_H];
    while (c_ev_end > -1.0f && idx < (nrow - maxsteps)) {
        idx += (int64_t)(c_ev_end + 1.0f);      /* Int(c_ev_end + 1) */
        if (idx > (nrow - maxsteps))            /* redraw: same seed => same value */
            idx = idx0;
        c_ev_end = table[(idx + maxsteps - 1) * ORC_NCOL + ORC_COL_H];
        counter += 1;
        if (counter > max_iterations) break;    /* println(...) ; break */
    }
    if (iterations) *iterations = counter;
    return idx;
}

int orc_reset(orc_env *e, int rng_is_minus1, int64_t idx0, float soc_b0)
{
    /* reset_state! LU1:216-262 */
    int64_t idx;
    if (rng_is_minus1) {
        /* 0.5 * (b.soc_min + b.soc_max): Float64 * Float32 -> Float64, stored as Float32 */
        jnum sb = jmul(J_f64(0.5), jadd(B_SOC_MIN, J_f32(e->prof.soc_max)));
        e->state[ORC_S_SOCB] = jto_f32(sb);
        idx = 1;
    } else {
        e->state[ORC_S_SOCB] = soc_b0;
        idx = orc_resolve_start(e->table, e->nrow, e->maxsteps, idx0, 0);
        if (idx < 0) return -1;
    }
    if (idx < 1 || idx > e->nrow) return -1;
    e->state[ORC_S_SOCEV]  = tab(e, idx, ORC_COL_SOCEV);
    e->state[ORC_S_CEV]    = tab(e, idx, ORC_COL_H);
    e->state[ORC_S_DE]     = tab(e, idx, ORC_COL_DE);
    e->state[ORC_S_GE]     = tab(e, idx, ORC_COL_GE);
    e->state[ORC_S_PBUY]   = tab(e, idx, ORC_COL_PBUY);
    e->state[ORC_S_SEASON] = tab(e, idx, ORC_COL_SEASON);
    e->state[ORC_S_HCOS]   = tab(e, idx, ORC_COL_HCOS);
    e->state[ORC_S_HSIN]   = tab(e, idx, ORC_COL_HSIN);
    /* reset! LU1:208-212 */
    e->reward = 0.0;
    e->a[0] = 0.7f; e->a[1] = 1.0f;
    e->step = 0;
    e->idx = idx;
    return 0;
}

static void action_drl(const orc_env *e, jnum B_target, jnum EV_target, jnum *B_out, jnum *EV_out)
{
    /* LU1:283-316 */
    const jnum Soc_b = J_f32(e->state[ORC_S_SOCB]), Soc_ev = J_f32(e->state[ORC_S_SOCEV]);
    const jnum c_ev = J_f32(e->state[ORC_S_CEV]);
    const jnum d_e = J_f32(e->state[ORC_S_DE]), g_e = J_f32(e->state[ORC_S_GE]);
    const jnum b_soc_max = J_f32(e->prof.soc_max), b_rate_max = J_f64(e->prof.rate_max);
    const jnum ev_soc_max = J_f32(e->prof.cap_ev);
    jnum B = J_f64(0.0), EV = J_f64(0.0);                                   /* zeros(2) */

    jnum Soc_b_perc = jdiv(jsub(Soc_b, B_SOC_MIN), jsub(b_soc_max, B_SOC_MIN));   /* :288 */
    if (jgt(c_ev, J_int(-1)) && jlt(Soc_ev, EV_target))                           /* :292 */
        EV = jmin(EV_RATE_MAX, jmul(jsub(EV_target, Soc_ev), jsub(ev_soc_max, EV_SOC_MIN)));
    else
        EV = J_int(0);
    jnum pv_ = jsub(jsub(g_e, d_e), EV);                                          /* :301 */
    if (jgt(pv_, J_int(0)) && jlt(Soc_b_perc, B_target)) {                        /* :304 */
        jnum B_target_value = jadd(jmul(B_target, jsub(b_soc_max, B_SOC_MIN)), B_SOC_MIN);
        B = jclamp(pv_, J_int(0),
                   jmin(b_rate_max, jadd(jsub(B_target_value, Soc_b), B_LOSS)));   /* :307 */
    } else if (jgt(Soc_b, J_f32(1e-3f))) {                                        /* :309 */
        B = jneg(jmin(b_rate_max, jmul(jsub(J_int(1), B_LOSS), Soc_b)));           /* :310 */
    } else {
        B = J_int(0);
    }
    *B_out = jround32(B);                                                /* Float32.([B, EV]) */
    *EV_out = jround32(EV);
}

static void action_rule(const orc_env *e, jnum *B_out, jnum *EV_out)
{
    /* LU1:318-340 */
    const jnum Soc_b = J_f32(e->state[ORC_S_SOCB]), Soc_ev = J_f32(e->state[ORC_S_SOCEV]);
    const jnum d_e = J_f32(e->state[ORC_S_DE]), g_e = J_f32(e->state[ORC_S_GE]);
    const jnum b_soc_max = J_f32(e->prof.soc_max), b_rate_max = J_f64(e->prof.rate_max);
    const jnum ev_soc_max = J_f32(e->prof.cap_ev);
    jnum B;
    jnum EV = jmin(EV_RATE_MAX, jmul(jsub(J_int(1), Soc_ev), jsub(ev_soc_max, EV_SOC_MIN)));  /* :323 */
    jnum pv_ = jsub(jsub(g_e, d_e), EV);                                                      /* :327 */
    if (jgt(pv_, J_int(0)) && jlt(Soc_b, jmul(J_f64(0.95), b_soc_max))) {                     /* :330 */
        B = jclamp(pv_, J_int(0), jmin(b_rate_max, jadd(jsub(b_soc_max, Soc_b), B_LOSS)));    /* :331 */
    } else if (jgt(Soc_b, J_f32(1e-3f))) {
        B = jneg(jmin(b_rate_max, jmul(jsub(J_int(1), B_LOSS), Soc_b)));
    } else {
        B = J_int(0);
    }
    *B_out = jround32(B);
    *EV_out = jround32(EV);
}

void orc_action_drl(const orc_env *e, float B_target, float EV_target, float out[2])
{
    jnum B, EV;
    action_drl(e, J_f32(B_target), J_f32(EV_target), &B, &EV);
    out[0] = jto_f32(B); out[1] = jto_f32(EV);
}

void orc_action_rule(const orc_env *e, float out[2])
{
    jnum B, EV;
    action_rule(e, &B, &EV);
    out[0] = jto_f32(B); out[1] = jto_f32(EV);
}

/* Float64 ^ Float64 (LU1:467/470).  openlibm's pow returns x*x for y == 2 and x for
 * y == 1 exactly; other exponents go through libm pow (tolerance-checked only). */
static double jpow64(double x, double y)
{
    if (y == 2.0) return x * x;
    if (y == 1.0) return x;
    return pow(x, y);
}

int orc_step(orc_env *e, const float a[2], int track_mode, double *reward_out, double *results23)
{
    /* LU1:343-485 */
    if (e->idx + 1 > e->nrow || e->idx < 1) return -1;   /* df[idx+1, ...] BoundsError in next_state! */

    const jnum Soc_b = J_f32(e->state[ORC_S_SOCB]), Soc_ev = J_f32(e->state[ORC_S_SOCEV]);
    const jnum c_ev = J_f32(e->state[ORC_S_CEV]);
    jnum d_e = J_f32(e->state[ORC_S_DE]);
    const jnum g_e = J_f32(e->state[ORC_S_GE]), p_buy = J_f32(e->state[ORC_S_PBUY]);
    const jnum b_soc_max = J_f32(e->prof.soc_max), b_rate_max = J_f64(e->prof.rate_max);
    const jnum ev_soc_max = J_f32(e->prof.cap_ev);
    const jnum w_disc = J_f64(e->prof.disc_weight);
    const jnum pen_w = J_f32(e->prof.penalty_weight);

    jnum B_target, EV_target, B, EV;
    if (track_mode >= 0) {                                  /* :346-349 */
        B_target = J_f32(a[0]); EV_target = J_f32(a[1]);
        action_drl(e, B_target, EV_target, &B, &EV);
    } else {                                                /* :350-354 */
        B_target = J_f32(0.f); EV_target = J_f32(0.f);
        B = J_f32(a[0]); EV = J_f32(a[1]);
    }
    e->a[0] = jto_f32(B_target); e->a[1] = jto_f32(EV_target);

    /* :356-357  zeros(8), zeros(11): every default is Float64 0.0 */
    jnum pv_ = J_f64(0), BD = J_f64(0), BC = J_f64(0);
    jnum discomfort, penalty;
    jnum PV_DE = J_f64(0), PV_B = J_f64(0), PV_EV = J_f64(0), PV_GR, B_DE = J_f64(0), B_EV = J_f64(0),
         B_GR, GR_DE = J_f64(0), GR_EV = J_f64(0), GR_B = J_f64(0), EX_EV;

    if (jlt(B, J_f64(-0.01)))                                                /* :362 */
        BD = jclamp(jneg(B), J_f64(0.001),
                    jmin(b_rate_max, jmul(jsub(jsub(J_int(1), B_LOSS), J_f32(1e-7f)), Soc_b)));

    if (jgt(jmul(g_e, PV_ETA), d_e)) {                                       /* :368 */
        PV_DE = d_e;
        pv_ = jsub(jmul(g_e, PV_ETA), PV_DE);
        if (jgt(pv_, EV)) {
            PV_EV = EV;
            pv_ = jsub(pv_, PV_EV);
        } else if (jle(pv_, EV)) {
            PV_EV = pv_;
            pv_ = J_int(0);
            if (jgt(BD, jdiv(jsub(EV, PV_EV), B_ETA))) {                     /* :377 */
                B_EV = jsub(EV, PV_EV);
                BD = jsub(BD, jdiv(B_EV, B_ETA));
            } else if (jle(BD, jdiv(jsub(EV, PV_EV), B_ETA))) {
                B_EV = jmul(BD, B_ETA);
                BD = J_int(0);
                GR_EV = jsub(jsub(EV, PV_EV), B_EV);
            }
        }
    } else if (jle(jmul(g_e, PV_ETA), d_e)) {                                /* :388 */
        PV_DE = jmul(g_e, PV_ETA);
        pv_ = J_int(0);
        d_e = jsub(d_e, PV_DE);
        if (jgt(BD, jdiv(d_e, B_ETA))) {                                     /* :392 */
            B_DE = d_e;
            BD = jsub(BD, jdiv(B_DE, B_ETA));
            if (jgt(BD, jdiv(EV, B_ETA))) {
                B_EV = EV;
                BD = jsub(BD, jdiv(B_EV, B_ETA));
            } else if (jle(BD, jdiv(EV, B_ETA))) {
                B_EV = jmul(BD, B_ETA);
                BD = J_int(0);
                GR_EV = jsub(EV, B_EV);
            }
        } else if (jle(BD, jdiv(d_e, B_ETA))) {                              /* :403 */
            B_DE = jmul(BD, B_ETA);
            BD = J_int(0);
            GR_DE = jsub(d_e, B_DE);
            GR_EV = EV;
        }
    }

    if (jgt(B, J_f64(0.01))) {                                               /* :412 */
        BC = jclamp(B, J_f64(0.001), jmin(b_rate_max, jsub(b_soc_max, Soc_b)));
        if (jgt(pv_, jdiv(BC, B_ETA))) {
            PV_B = BC;
            pv_ = jsub(pv_, jdiv(BC, B_ETA));
        } else if (jle(pv_, jdiv(BC, B_ETA))) {
            PV_B = jmul(pv_, B_ETA);
            pv_ = J_int(0);
            GR_B = J_int(0);
        }
    }
    PV_GR = pv_;                                                             /* :424 */
    B_GR = J_int(0);                                                         /* :425 */

    /* :432  env.state.Soc_b = (1 - b.loss) * (Soc_b + PV_B + GR_B - ((B_DE + B_EV + B_GR) / b.eta)) */
    jnum new_soc_b = jmul(jsub(J_int(1), B_LOSS),
                          jsub(jadd(jadd(Soc_b, PV_B), GR_B),
                               jdiv(jadd(jadd(B_DE, B_EV), B_GR), B_ETA)));
    float st_soc_b = jto_f32(new_soc_b);
    /* :435  env.state.Soc_ev = Soc_ev + (PV_EV + B_EV + GR_EV) / (ev.soc_max - ev.soc_min) */
    jnum new_soc_ev = jadd(Soc_ev, jdiv(jadd(jadd(PV_EV, B_EV), GR_EV), jsub(ev_soc_max, EV_SOC_MIN)));
    float st_soc_ev = jto_f32(new_soc_ev);

    discomfort = J_int(0); penalty = J_int(0); EX_EV = J_int(0);             /* :438-440 */
    if (jeq(c_ev, J_int(0)) && jlt(J_f32(st_soc_ev), J_int(1))) {            /* :442 */
        discomfort = jmul(jsub(J_int(1), J_f32(st_soc_ev)), J_int(100));
        EX_EV = jmul(jsub(J_int(1), J_f32(st_soc_ev)), jsub(ev_soc_max, EV_SOC_MIN));
        st_soc_ev = 1.0f;
    } else if (jlt(c_ev, J_int(0)) && jlt(EV_target, J_f64(0.99))) {         /* :447 */
        penalty = jmul(jsub(J_int(1), EV_target), pen_w);
    }

    /* next_state!(env) LU1:264-281 */
    {
        int64_t idx = e->idx + 1;
        float new_c_ev = tab(e, idx, ORC_COL_H);
        e->state[ORC_S_CEV] = new_c_ev;
        if (new_c_ev >= 0.0f && tab(e, e->idx, ORC_COL_H) == -1.0f)          /* newly connected */
            st_soc_ev = tab(e, idx, ORC_COL_SOCEV);
        e->state[ORC_S_DE]     = tab(e, idx, ORC_COL_DE);
        e->state[ORC_S_GE]     = tab(e, idx, ORC_COL_GE);
        e->state[ORC_S_PBUY]   = tab(e, idx, ORC_COL_PBUY);
        e->state[ORC_S_SEASON] = tab(e, idx, ORC_COL_SEASON);
        e->state[ORC_S_HCOS]   = tab(e, idx, ORC_COL_HCOS);
        e->state[ORC_S_HSIN]   = tab(e, idx, ORC_COL_HSIN);
    }
    e->state[ORC_S_SOCB] = st_soc_b;
    e->state[ORC_S_SOCEV] = st_soc_ev;
    e->step += 1;                                                            /* :455 */
    e->idx += 1;                                                             /* :456 */

    /* :464  profit = (sell * p_buy * (PV_GR + B_GR)) - (p_buy * (GR_DE + GR_B + GR_EV + EX_EV)) */
    jnum profit = jsub(jmul(jmul(M_SELL, p_buy), jadd(PV_GR, B_GR)),
                       jmul(p_buy, jadd(jadd(jadd(GR_DE, GR_B), GR_EV), EX_EV)));
    jnum disc_term = jmul(w_disc, J_f64(jpow64(jto_f64(discomfort), e->prof.disc_pot)));
    jnum reward;
    if (track_mode < 0) {                                                    /* :466-468 */
        reward = jsub(profit, disc_term);
        penalty = J_int(0);
    } else {
        reward = jsub(jsub(profit, disc_term), penalty);
    }
    e->reward = jto_f64(reward);
    if (reward_out) *reward_out = e->reward;

    if (results23) {                                                         /* :476-478 */
        double *r = results23;
        r[0] = (double)e->idx;      r[1] = c_ev.v;       r[2] = EV_target.v;  r[3] = EV.v;
        r[4] = Soc_ev.v;            r[5] = e->reward;    r[6] = profit.v;     r[7] = discomfort.v;
        r[8] = penalty.v;           r[9] = PV_DE.v;      r[10] = B_DE.v;      r[11] = GR_DE.v;
        r[12] = PV_B.v;             r[13] = PV_GR.v;     r[14] = PV_EV.v;     r[15] = B_EV.v;
        r[16] = GR_EV.v;            r[17] = EX_EV.v;     r[18] = GR_B.v;      r[19] = B_GR.v;
        r[20] = B.v;                r[21] = B_target.v;  r[22] = Soc_b.v;
    }
    (void)BC;
    return 0;
}

int orc_finished(const orc_env *e) { (void)e; return 0; }   /* LU1:487-502: false on both paths */

float orc_scale_action(float a)
{
    /* DDPG.jl:180-182: Float32.(LO .+ (a .+ ones(2)) .* 0.5 .* (HI .- LO)), LO=0f0, HI=1f0.
     * (a + 1.0) is Float64 because ones() is Float64. */
    jnum r = jadd(J_f32(0.f), jmul(jmul(jadd(J_f32(a), J_f64(1.0)), J_f64(0.5)),
                                   jsub(J_f32(1.f), J_f32(0.f))));
    return jto_f32(r);
}

int orc_batch_step(orc_env *envs, int64_t n, const float *actions, int track_mode,
                   double *rewards, float *obs_out, double *results)
{
    int rc = 0;
    for (int64_t i = 0; i < n; ++i) {
        double r;
        if (orc_step(&envs[i], actions + 2 * i, track_mode, &r, results ? results + 23 * i : 0) != 0) rc = -1;
        if (rewards) rewards[i] = r;
        if (obs_out) memcpy(obs_out + ORC_NSTATE * i, envs[i].state, sizeof(float) * ORC_NSTATE);
    }
    return rc;
}

double orc_rule_episode(orc_env *e, int64_t steps, double *results)
{
    /* inference(track<0) -> episode!(rng_ep=-1) DDPG.jl:186-242 with the actor call elided:
     * its result is discarded on the rule-based path (DDPG.jl:209-211). */
    double total = 0.0;       /* reward_eps: 0f0 + Float64 -> Float64 after the first add */
    if (orc_reset(e, 1, 1, 0.f) != 0) return NAN;
    for (int64_t s = 0; s < steps; ++s) {
        float a[2]; double r;
        orc_action_rule(e, a);
        if (orc_step(e, a, -1, &r, results ? results + 23 * s : 0) != 0) return NAN;
        total += r;
    }
    return total;
}

/* OpenMP variant for the all-cores CPU baseline. */
int orc_batch_step_omp(orc_env *envs, int64_t n, const float *actions, int track_mode,
                       double *rewards, float *obs_out)
{
    int rc = 0;
#pragma omp parallel for schedule(static) reduction(|:rc)
    for (int64_t i = 0; i < n; ++i) {
        double r;
        if (orc_step(&envs[i], actions + 2 * i, track_mode, &r, 0) != 0) rc |= 1;
        if (rewards) rewards[i] = r;
        if (obs_out) memcpy(obs_out + ORC_NSTATE * i, envs[i].state, sizeof(float) * ORC_NSTATE);
    }
    return rc ? -1 : 0;
}

/* Whole episodes on all host cores: envs are independent, so each thread carries its envs through all `nsteps` steps inside ONE
 * parallel region (the per-vector-step variant above pays a fork/join per step, which at 256 threads outweighs the work).
 * actions: [nsets][n][2], step t uses set t % nsets.  returns_out[i] = sum of the env's rewards. */
int orc_batch_episode_omp(orc_env *envs, int64_t n, const float *actions, int64_t nsets, int64_t nsteps, int track_mode,
                          double *returns_out)
{
    int rc = 0;
#pragma omp parallel for schedule(static) reduction(|:rc)
    for (int64_t i = 0; i < n; ++i) {
        double total = 0.0;
        for (int64_t t = 0; t < nsteps; ++t) {
            double r;
            if (orc_step(&envs[i], actions + ((t % nsets) * n + i) * 2, track_mode, &r, 0) != 0) { rc |= 1; break; }
            total += r;
        }
        if (returns_out) returns_out[i] = total;
    }
    return rc ? -1 : 0;
}

/* Array-of-envs helpers so Python (ctypes) can own a batch without mirroring the struct. */
#include <stdlib.h>
orc_env *orc_batch_alloc(int64_t n) { return (orc_env *)calloc((size_t)n, sizeof(orc_env)); }
void     orc_batch_free(orc_env *p) { free(p); }
orc_env *orc_batch_at(orc_env *p, int64_t i) { return p + i; }
int64_t  orc_env_idx(const orc_env *e) { return e->idx; }
int64_t  orc_env_step(const orc_env *e) { return e->step; }
void     orc_env_get_state(const orc_env *e, float *out9) { memcpy(out9, e->state, sizeof(float) * ORC_NSTATE); }
void     orc_env_set_state(orc_env *e, const float *in9, int64_t idx, int64_t step)
{ memcpy(e->state, in9, sizeof(float) * ORC_NSTATE); e->idx = idx; e->step = step; }

/* Whole-batch forms of the accessors above: ONE foreign call per batch, so that a timed loop (bench.py's cpu_baseline) or a
 * parity test never pays a Python/ctypes round trip per env.  NULL idx0/soc_b0 = (1, 0.f) for every env; NULL step = 0. */
void orc_batch_init(orc_env *envs, int64_t n, int64_t maxsteps, const float *const *tables, const int64_t *nrows,
                    const int64_t *table_of_env, const orc_profile *profiles, const int64_t *profile_of_env)
{
    for (int64_t i = 0; i < n; ++i) {
        int64_t t = table_of_env ? table_of_env[i] : 0, p = profile_of_env ? profile_of_env[i] : 0;
        orc_env_init(&envs[i], maxsteps, tables[t], nrows[t], &profiles[p]);
    }
}
int orc_batch_reset(orc_env *envs, int64_t n, int rng_is_minus1, const int64_t *idx0, const float *soc_b0)
{
    int rc = 0;
    for (int64_t i = 0; i < n; ++i)
        rc |= orc_reset(&envs[i], rng_is_minus1, idx0 ? idx0[i] : 1, soc_b0 ? soc_b0[i] : 0.f);
    return rc;
}
void orc_batch_get_state(const orc_env *envs, int64_t n, float *obs_out)
{ for (int64_t i = 0; i < n; ++i) memcpy(obs_out + ORC_NSTATE * i, envs[i].state, sizeof(float) * ORC_NSTATE); }
void orc_batch_set_state(orc_env *envs, int64_t n, const float *obs, const int64_t *idx, const int64_t *step)
{ for (int64_t i = 0; i < n; ++i) orc_env_set_state(&envs[i], obs + ORC_NSTATE * i, idx[i], step ? step[i] : 0); }
void orc_batch_get_idx(const orc_env *envs, int64_t n, int64_t *idx_out, int64_t *step_out)
{
    for (int64_t i = 0; i < n; ++i) {
        if (idx_out) idx_out[i] = envs[i].idx;
        if (step_out) step_out[i] = envs[i].step;
    }
}
void orc_batch_action_drl(const orc_env *envs, int64_t n, const float *targets, float *out)
{ for (int64_t i = 0; i < n; ++i) orc_action_drl(&envs[i], targets[2 * i], targets[2 * i + 1], out + 2 * i); }
void orc_batch_action_rule(const orc_env *envs, int64_t n, float *out)
{ for (int64_t i = 0; i < n; ++i) orc_action_rule(&envs[i], out + 2 * i); }
void orc_scale_actions(const float *a, int64_t count, float *out)                 /* scale_action over a whole array */
{ for (int64_t i = 0; i < count; ++i) out[i] = orc_scale_action(a[i]); }
