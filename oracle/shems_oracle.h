/*
 * shems_oracle.h -- CPU restatement of the reference's shems_LU1 environment.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker / the timed CPU baseline.
 *
 * PARITY STATUS: "parity unpinned by the reference".  The reference is Julia
 * (not installed here), has no tests, and all of its run artefacts are git-LFS
 * pointer stubs.  This restatement follows
 *   /root/reference/RL-SHEMS/RL_environments/envs/shems_LU1.jl   (LU1)
 * line by line with an explicit emulation of Julia's Int/Float32/Float64
 * promotion, and is pinned by (a) the hand-traced known-answer vectors of
 * SURVEY.md Appendix B, (b) an independent NumPy restatement
 * (oracle/shems_oracle_np.py) and (c) the conservation invariants of
 * SURVEY.md A.4.
 */
#ifndef SHEMS_ORACLE_H
#define SHEMS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Exogenous table row, 8 x f32 (values are Float32(Float64 csv value)):
 * h_countdown, soc_ev, electkwh, PV_generation, p_buy, hour_cos, hour_sin, season
 * (columns read by LU1:251-260 / LU1:268-279). */
enum { ORC_COL_H = 0, ORC_COL_SOCEV = 1, ORC_COL_DE = 2, ORC_COL_GE = 3,
       ORC_COL_PBUY = 4, ORC_COL_HCOS = 5, ORC_COL_HSIN = 6, ORC_COL_SEASON = 7,
       ORC_NCOL = 8 };

/* State vector order of ShemsState (LU1:101-111). */
enum { ORC_S_SOCB = 0, ORC_S_SOCEV = 1, ORC_S_CEV = 2, ORC_S_DE = 3, ORC_S_GE = 4,
       ORC_S_PBUY = 5, ORC_S_HCOS = 6, ORC_S_HSIN = 7, ORC_S_SEASON = 8, ORC_NSTATE = 9 };

/* Module-level constants of LU1:40-59, 92-99 (per charger profile / per sweep point). */
typedef struct {
    float  cap_ev;        /* ev.soc_max  (capacities[id][1])                 */
    float  soc_max;       /* b.soc_max   (capacities[id][2], an f32 product) */
    double rate_max;      /* b.rate_max  (capacities[id][3], Float64)        */
    double disc_weight;   /* m.discomfort_weight_ev = Float64(f32 literal)   */
    double disc_pot;      /* m.disc_pot             = Float64(f32 literal)   */
    float  penalty_weight;/* penalty_weight (LU1:43)                         */
} orc_profile;

/* One scalar environment (LU1:169-177). */
typedef struct {
    float   state[ORC_NSTATE];
    double  reward;
    float   a[2];         /* env.a = ShemsAction(B_target, EV_target) */
    int64_t step;
    int64_t maxsteps;
    int64_t idx;          /* 1-based row index, as in Julia */
    const float *table;   /* [nrow][8] */
    int64_t nrow;
    orc_profile prof;
} orc_env;

/* capacities dict LU1:47-59; returns 0 on success, -1 for an unknown id.
 * weight/pot/penalty are set to the LU1:40-43 defaults (0.01f0, 2f0, 0.1f0). */
int  orc_profile_for_charger(int charger_id, orc_profile *out);

void orc_env_init(orc_env *e, int64_t maxsteps, const float *table, int64_t nrow,
                  const orc_profile *p);                                  /* LU1:203 */

/* Episode-start resolution loop LU1:227-246 for a given initial draw idx0
 * (1-based).  Returns the final idx; *iterations gets the loop counter.
 * Returns -1 if a table access would be out of bounds (Julia BoundsError). */
int64_t orc_resolve_start(const float *table, int64_t nrow, int64_t maxsteps,
                          int64_t idx0, int *iterations);

/* reset!(env; rng) LU1:206-262.  rng_is_minus1 != 0 reproduces rng == -1
 * (Soc_b = 0.5*(soc_min+soc_max), idx = 1).  Otherwise the two MersenneTwister
 * draws are supplied by the caller: idx0 (1-based draw from 1:(nrow-maxsteps))
 * and soc_b0 (the Uniform(soc_min, soc_max) draw, already rounded to f32).
 * Returns 0, or -1 on out-of-bounds. */
int  orc_reset(orc_env *e, int rng_is_minus1, int64_t idx0, float soc_b0);

/* action(env, a::ShemsAction) LU1:283-316  -> out[0]=B, out[1]=EV (Float32). */
void orc_action_drl(const orc_env *e, float B_target, float EV_target, float out[2]);
/* action(env, track) LU1:318-340 (rule based)  -> out[0]=B, out[1]=EV. */
void orc_action_rule(const orc_env *e, float out[2]);

/* step!(env, s, a; track) LU1:343-485.
 * track_mode: 0 -> track == 0 (DRL, no results row), 1 -> track > 0 (DRL + results),
 *            -1 -> track < 0 (a = kWh set-points (B, EV); penalty zeroed).
 * results23 may be NULL; if given it receives the 23 Float64 columns of LU1:476-478.
 * Returns 0, or -1 when idx+1 would exceed nrow (Julia BoundsError). */
int  orc_step(orc_env *e, const float a[2], int track_mode, double *reward_out,
              double *results23);

/* finished(env, s') LU1:487-502: always false. */
int  orc_finished(const orc_env *e);

/* scale_action DDPG.jl:178-184 with ACTION_BOUND_LO=(0f0,0f0), HI=(1f0,1f0). */
float orc_scale_action(float a);

/* Batched helpers used by bench.py's cpu_baseline leg and by the parity tests:
 * n independent envs stored AoS.  actions is [n][2].  Returns 0 / -1. */
int  orc_batch_step(orc_env *envs, int64_t n, const float *actions, int track_mode,
                    double *rewards, float *obs_out /* [n][9] or NULL */,
                    double *results /* [n][23] or NULL */);

/* One full rule-based episode (BASELINE config 1): reset(rng=-1) then `steps`
 * x { a = action(env, track); step!(env, s, a, track=-0.5) }.  Returns sum of rewards. */
double orc_rule_episode(orc_env *e, int64_t steps, double *results /* [steps][23] or NULL */);

int  orc_batch_step_omp(orc_env *envs, int64_t n, const float *actions, int track_mode,
                        double *rewards, float *obs_out);
int  orc_batch_episode_omp(orc_env *envs, int64_t n, const float *actions /* [nsets][n][2] */, int64_t nsets, int64_t nsteps,
                           int track_mode, double *returns_out /* [n] or NULL */);
orc_env *orc_batch_alloc(int64_t n);
void     orc_batch_free(orc_env *p);
orc_env *orc_batch_at(orc_env *p, int64_t i);
int64_t  orc_env_idx(const orc_env *e);
int64_t  orc_env_step(const orc_env *e);
void     orc_env_get_state(const orc_env *e, float *out9);
void     orc_env_set_state(orc_env *e, const float *in9, int64_t idx, int64_t step);

/* Whole-batch accessors (one call per batch; NULL idx0/soc_b0 -> (1, 0.f), NULL step -> 0). */
void orc_batch_init(orc_env *envs, int64_t n, int64_t maxsteps, const float *const *tables, const int64_t *nrows,
                    const int64_t *table_of_env /* or NULL */, const orc_profile *profiles, const int64_t *profile_of_env /* or NULL */);
int  orc_batch_reset(orc_env *envs, int64_t n, int rng_is_minus1, const int64_t *idx0, const float *soc_b0);
void orc_batch_get_state(const orc_env *envs, int64_t n, float *obs_out /* [n][9] */);
void orc_batch_set_state(orc_env *envs, int64_t n, const float *obs, const int64_t *idx, const int64_t *step);
void orc_batch_get_idx(const orc_env *envs, int64_t n, int64_t *idx_out, int64_t *step_out);
void orc_batch_action_drl(const orc_env *envs, int64_t n, const float *targets /* [n][2] */, float *out /* [n][2] */);
void orc_batch_action_rule(const orc_env *envs, int64_t n, float *out /* [n][2] */);
void orc_scale_actions(const float *a, int64_t count, float *out);

#ifdef __cplusplus
}
#endif
#endif
