/*
 * shems_hip.h -- C ABI of libshems_hip.so: the batched, MI355X-native (gfx950)
 * replacement for the hot path of the reference
 *   RL-SHEMS/RL_environments/envs/shems_LU1.jl        (LU1, the environment)
 *   RL-SHEMS/algorithms/DDPG.jl                        (DDPG.jl, act/replay/episode!)
 *   RL-SHEMS/src/memory_plotting_saving.jl             (MPS, replay buffer + normalisation)
 *
 * The reference has no FFI today: LU1 extends the Reinforce.jl generics
 * `reset!/step!/action/finished` on `Shems <: AbstractEnvironment` (LU1:62-65,169).
 * Each entry point below names the reference method it replaces; the Julia
 * `ccall` stubs a maintainer would add are in INTEGRATION.md, and the Python
 * `ctypes` mirror used by this repo's tests is the package's `_capi.py`.
 *
 * Conventions
 *   - plain C types only; every function returns SHEMS_OK (0) or a negative error
 *     code, and shems_last_error() returns a thread-local message.
 *   - "host" pointers are ordinary CPU memory; "dev" pointers are HIP device memory.
 *   - row / step indices are 1-based at this boundary, exactly as in the Julia code.
 *   - observation layout is [N][9] float32 = Julia's 9xN column-major Matrix{Float32}
 *     (state order LU1:101-111: Soc_b, Soc_ev, c_ev, d_e, g_e, p_buy, h_cos, h_sin, season).
 *   - a table is [nrow][8] float32: h_countdown, soc_ev, electkwh, PV_generation,
 *     p_buy, hour_cos, hour_sin, season  (the columns LU1:251-260 / 268-279 read).
 *   - `stream` arguments are a hipStream_t passed as void* (NULL = the legacy default stream).
 */
#ifndef SHEMS_HIP_H
#define SHEMS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHEMS_ABI_VERSION 1

enum {
    SHEMS_OK          =  0,
    SHEMS_ERR_ARG     = -1,  /* bad argument (NULL, size, range)                                   */
    SHEMS_ERR_HIP     = -2,  /* a HIP runtime call failed                                           */
    SHEMS_ERR_INDEX   = -3,  /* table access out of range: Julia BoundsError (LU1:227,239,265-279) */
    SHEMS_ERR_NOMEM   = -4,
    SHEMS_ERR_NODEVICE= -5,  /* no gfx950 device visible: the library never falls back to the CPU   */
    SHEMS_ERR_STATE   = -6   /* call order (tables/configs not set, not reset)                      */
};

enum { SHEMS_NSTATE = 9, SHEMS_NACTION = 2, SHEMS_NCOL = 8, SHEMS_NRESULT = 23 };

/* track modes of step!(env, s, a; track) (LU1:343-354, 466-484) */
enum {
    SHEMS_TRACK_OFF   = 0,   /* track == 0 : a = SoC targets in [0,1]                               */
    SHEMS_TRACK_DRL   = 1,   /* track  > 0 : same, plus the 23-column results row (LU1:476-478)     */
    SHEMS_TRACK_RULE  = -1   /* track  < 0 : a = kWh set-points (B, EV); penalty forced to 0        */
};

/* One "config" = what the reference spreads over module globals and env.path:
 * charger capacities (LU1:47-59), Battery/EV/Market constants (LU1:92-99), reward weights
 * (LU1:40-43; per-env in the discomfort-weight sweep) and the input table (env.path, LU1:176). */
typedef struct shems_config {
    float   cap_ev;          /* ev.soc_max  [kWh]   capacities[id][1]                               */
    float   soc_max;         /* b.soc_max   [kWh]   capacities[id][2] (an f32 product, e.g. 7.5f0*0.9f0) */
    double  rate_max;        /* b.rate_max  [kW]    capacities[id][3] (Float64)                     */
    double  disc_weight;     /* m.discomfort_weight_ev = Float64(DISCOMFORT_WEIGHT_EV::Float32)     */
    double  disc_pot;        /* m.disc_pot             = Float64(DISC_POT::Float32)                 */
    float   penalty_weight;  /* penalty_weight::Float32 (LU1:43)                                    */
    int32_t table_row0;      /* 0-based first row of this config's table in the uploaded row array  */
    int32_t nrow;            /* nrow(df) of that table                                              */
    int32_t reserved;
} shems_config;              /* 48 bytes */

typedef struct shems_env shems_env;      /* opaque: N parallel Shems instances on one GPU */

/* ------------------------------------------------------------------ library -- */
int         shems_abi_version(void);
const char *shems_last_error(void);
int         shems_device_count(int *out_count);

/* ------------------------------------------------- handle API (host arrays) -- */
/* Shems(maxsteps, path) x n_envs (LU1:203; called from input.jl:162-164).  Selects `device`
 * (the reference: CUDA.device!(GPU_ID), DDPG_reinforce_charger_v1.jl:12-14). */
int shems_create(int64_t n_envs, int32_t maxsteps, int device, shems_env **out);
int shems_destroy(shems_env *env);
int shems_n_envs(const shems_env *env, int64_t *out);

/* Replaces CSV.read(env.path, DataFrame) (LU1:217, 265): all tables, concatenated row-wise,
 * are uploaded once.  rows = [total_rows][8] float32 host memory. */
int shems_set_tables(shems_env *env, const float *rows, int64_t total_rows);
/* Replaces the module globals LU1:40-59, 92-99.  cfg_of_env = [n_envs] host indices into cfgs
 * (NULL: every env uses cfgs[0]). */
int shems_set_configs(shems_env *env, const shems_config *cfgs, int32_t n_cfg,
                      const uint16_t *cfg_of_env);

/* reset!(env; rng) (LU1:206-262).  rng_minus1 != 0  <=>  rng == -1: Soc_b = 0.5*(soc_min+soc_max),
 * idx = 1.  Otherwise the two MersenneTwister draws are inputs: idx0[i] in 1..(nrow-maxsteps)
 * (1-based) and soc_b0[i]; the episode-extension loop LU1:227-246 runs on the device. */
int shems_reset(shems_env *env, int rng_minus1, const int32_t *idx0, const float *soc_b0);
/* Same, with both draws taken from the library's counter-based generator (Philox4x32-10):
 * idx0 = 1 + x0 mod (nrow-maxsteps), soc_b0 = float(x1 >> 8) * 2^-24 * soc_max,
 * (x0,x1,..) = philox(key = seed, counter = (env index, episode, 0, 0)). */
int shems_reset_seeded(shems_env *env, uint64_t seed, uint32_t episode);

/* step!(env, s, a; track) (LU1:343-485).  actions = [n][2] host float32.  Outputs (each may be
 * NULL): rewards [n] Float64, obs [n][9] (= Vector{Float32}(env.state)), results [n][23] Float64
 * (column order LU1:476-478; written for every track_mode if non-NULL).
 * Returns SHEMS_ERR_INDEX (and steps no env past the table) if any env has idx+1 > nrow. */
int shems_step(shems_env *env, const float *actions, int track_mode,
               double *rewards, float *obs, double *results);

/* action(env, a::ShemsAction) (LU1:283-316): targets [n][2] -> kWh set-points [n][2] (B, EV). */
int shems_action(shems_env *env, const float *targets, float *out_b_ev);
/* action(env, track) (LU1:318-340), the rule-based controller -> [n][2] (B, EV). */
int shems_rule_action(shems_env *env, float *out_b_ev);
/* finished(env, s') (LU1:487-502): always false; done = [n] bytes. */
int shems_finished(shems_env *env, uint8_t *done);

/* inference(env; track != 0) (memory_plotting_saving.jl:62-89): the whole tracking pass -- `nsteps` hours from the envs' current
 * state (call shems_reset(env, 1, NULL, NULL) first, as episode!(...; rng_ep = -1) does) -- in ONE launch, host arrays in and out:
 * track_mode > 0: the deterministic actor (`actor` = the 129002 parameters in Flux.params order, s_min / s_max [9]); track_mode < 0:
 * the rule-based controller (actor / s_min / s_max may be NULL).  results [nsteps][23] Float64 receives the rows of env 0
 * (shems_LU1.jl:476-478), returns [n_envs] the summed rewards (DDPG.jl:223); either may be NULL.  SHEMS_ERR_INDEX if the pass
 * runs off the table (Julia: BoundsError).  The device-pointer form, with one actor per env, is shems_track_dev. */
int shems_track(shems_env *env, const float *actor, const float *s_min, const float *s_max, int track_mode, int32_t nsteps,
                double *results, double *returns);

/* env.state / env.idx / env.step accessors (LU1:169-177).  Any pointer may be NULL. */
int shems_get_state(shems_env *env, float *obs, int32_t *idx, int32_t *step);
int shems_set_state(shems_env *env, const float *obs, const int32_t *idx, const int32_t *step);

/* ----------------------------------------- device-pointer (zero-copy) API -- */
/* Device views of a handle, for chaining with the policy / DDPG kernels without host copies. */
typedef struct shems_view {
    int64_t  n_envs;
    int32_t  maxsteps;
    int32_t  n_cfg;
    float   *obs;            /* dev [n][9]                         */
    int32_t *idx;            /* dev [n]   1-based row index        */
    int32_t *step;           /* dev [n]                            */
    const uint16_t     *cfg_of_env;   /* dev [n]                   */
    const shems_config *cfgs;         /* dev [n_cfg]               */
    const float        *tables;       /* dev [total_rows][8]       */
    int64_t  total_rows;
    int32_t *err;            /* dev [1]: sticky error word (SHEMS_ERR_*) set by kernels */
} shems_view;
int shems_get_view(shems_env *env, shems_view *out);
int shems_set_stream(shems_env *env, void *stream);     /* stream used by the handle API         */
int shems_check_error(shems_env *env);                  /* syncs, reads + clears view.err          */

/* step! on device buffers.  d_actions [n][2]; optional outputs: d_rewards [n] Float64,
 * d_rewards_f32 [n] (the Float32 the replay buffer keeps, MPS:37 + gpu()), d_results [n][23],
 * d_block_reward [ceil(n/256)] per-workgroup reward sums (wavefront reduction). */
int shems_step_dev(const shems_view *v, const float *d_actions, int track_mode,
                   double *d_rewards, float *d_rewards_f32, double *d_results,
                   double *d_block_reward, void *stream);
int shems_action_dev(const shems_view *v, const float *d_targets, int rule_based,
                     float *d_out_b_ev, void *stream);
int shems_reset_dev(const shems_view *v, int rng_minus1, const int32_t *d_idx0,
                    const float *d_soc_b0, void *stream);
int shems_reset_seeded_dev(const shems_view *v, uint64_t seed, uint32_t episode, void *stream);

/* Whole-episode rollout without host round trips: `nsteps` x { a = policy; step! } in ONE launch,
 * env state held in registers.  policy = SHEMS_ROLLOUT_RULE: a = action(env, track), track < 0
 * (BASELINE config 1 at scale, MPS:62-71 + DDPG.jl:209-212); SHEMS_ROLLOUT_RANDOM: uniform random
 * actions in [-1,1] -> scale_action (populate_memory, MPS:9-29), Philox keyed by (seed, env, step).
 * d_returns [n] Float64 episode sums (DDPG.jl:223).  If ring != NULL every transition is appended
 * of the first `ring_envs` envs (0 = all) is appended to the replay ring (see shems_replay below) at
 * slot (ring_pos + env*nsteps + t) mod capacity, i.e. in the reference's episode-major push order;
 * pushes that a later push of the same launch would overwrite (order < ring_envs*nsteps - capacity)
 * are skipped, so the ring content is deterministic (= what the CircularBuffer would hold). */
enum { SHEMS_ROLLOUT_RULE = 0, SHEMS_ROLLOUT_RANDOM = 1 };
struct shems_replay;
int shems_rollout_dev(const shems_view *v, int policy, int32_t nsteps, uint64_t seed,
                      double *d_returns, const struct shems_replay *ring, int64_t ring_pos,
                      int64_t ring_envs, void *stream);

/* ------------------------------------------------------------ replay ring -- */
/* memory = CircularBuffer{Any}(MEM_SIZE) of [s, a, r, s', done] (input.jl:139-140, MPS:46-47),
 * kept in HBM as struct-of-arrays.  `a` is the UNSCALED policy output in [-1,1] (DDPG.jl:229). */
typedef struct shems_replay {
    int64_t  capacity;       /* MEM_SIZE                                   */
    float   *s;              /* dev [capacity][9]                          */
    float   *a;              /* dev [capacity][2]                          */
    float   *r;              /* dev [capacity]    Float32(reward)          */
    float   *s2;             /* dev [capacity][9]                          */
    uint8_t *done;           /* dev [capacity]                             */
} shems_replay;              /* the push position is host state and is passed by value */


/* ------------------------------------------------------- policy (actor) -- */
/* Network parameters are ONE flat float32 device array per network in Flux's own order and
 * layout (Flux.params(Chain(Dense, Dense, Dense)) = W1, b1, W2, b2, W3, b3 with each W a
 * column-major out x in Matrix, DDPG.jl:30-46):
 *   actor  (9 -> 250 -> 500 -> 2, relu, relu, tanh):  W1[9][250] b1[250] W2[250][500] b2[500] W3[500][2] b3[2]  = 129 002
 *   critic (11 -> 250 -> 500 -> 1, relu, relu, id):   W1[11][250] b1[250] W2[250][500] b2[500] W3[500][1] b3[1] = 129 001
 * ([k][n] = C order of the Julia out x in column-major matrix: the out index is contiguous). */
enum { SHEMS_L1 = 250, SHEMS_L2 = 500, SHEMS_ACTOR_PARAMS = 129002, SHEMS_CRITIC_PARAMS = 129001 };

typedef struct shems_act_params {
    const float *actor;      /* dev [129002]                                                        */
    const float *s_min;      /* dev [9]  normalize(): (s - s_min) / (s_max - s_min + 1f-8), MPS:55-57 */
    const float *s_max;      /* dev [9]                                                             */
    float    noise_mu;       /* gn = GNoise(mu, sigma_act, .) / ou = OUNoise(mu, sigma, theta, dt, X)  input.jl:190-237 */
    float    noise_sigma;
    int32_t  train;          /* act(...; train): 1 = explore (DDPG.jl:151-172)                      */
    uint32_t tick;           /* counter word of the noise stream (the reference's rng_step, DDPG.jl:197) */
    uint64_t seed;           /* Philox key                                                          */
    int32_t  noise_kind;     /* SHEMS_NOISE_*: which branch of act() (noise_type "gn" / "ou" / "en") */
    float    ou_theta;       /* OUNoise theta                                                       */
    float    ou_dt;          /* OUNoise dt (1f-2)                                                   */
    float    eps;            /* EpsNoise: current xi = max(0.5 - zeta*(episode - MEM/EP), xi_min), DDPG.jl:69-72 */
    float   *ou_state;       /* dev [n][2]: OUNoise.X per env (persistent, DDPG.jl:49-55); required for SHEMS_NOISE_OU */
    float   *noise_acc;      /* dev [n] or NULL: += act()'s second return value for this step (DDPG.jl:148-176): mean(noise) of the
                              * two action noise samples (gn, ou), mean(|act_pred - act_uni|) when an epsilon step explores, else 0 */
} shems_act_params;
/* act()'s exploration branches (DDPG.jl:148-176).  GAUSS: clamp(a + N(mu, sigma)); OU: X += theta(mu - X)dt +
 * sigma sqrt(dt) N(0,1), clamp(a + X); EPS: with probability eps a uniform action in [-1,1]^2, else the
 * unperturbed a (returned unclamped by the reference -- tanh already bounds it). */
enum { SHEMS_NOISE_GAUSS = 0, SHEMS_NOISE_OU = 1, SHEMS_NOISE_EPS = 2 };

/* Which transitions of a vector step enter the replay ring: envs i with ((i - offset) mod n) < count
 * are stored at slot (pos + ((i - offset) mod n)) mod capacity.  count = 0 stores nothing. */
typedef struct shems_ring_window {
    int64_t pos;
    int64_t count;
    int64_t offset;
} shems_ring_window;

/* act(): a = clamp(actor(normalize(s)) + noise, -1, 1) for m observations (DDPG.jl:148-176).
 * d_obs [m][9] -> d_a [m][2] (UNSCALED action in [-1, 1]).  fp32 MFMA (v_mfma_f32_32x32x2_f32). */
int shems_actor_forward_dev(const shems_act_params *p, const float *d_obs, int64_t m, float *d_a,
                            void *stream);

/* One fused vector step of episode! (DDPG.jl:195-234) for every env of the view, in ONE launch:
 *   s = env.state; a = act(normalize(s)); step!(env, s, scale_action(a)); remember(s, a, r, s', false).
 * Optional outputs: d_a [n][2] unscaled actions, d_rewards [n] f64, d_rewards_f32 [n],
 * d_block_reward [grid] per-workgroup reward sums, d_returns_acc [n] f64 += reward (reward_eps, DDPG.jl:223).  ring/window may be NULL (evaluation episodes). */
int shems_act_step_dev(const shems_view *v, const shems_act_params *p, float *d_a, double *d_rewards,
                       float *d_rewards_f32, double *d_block_reward, double *d_returns_acc,
                       const shems_replay *ring, const shems_ring_window *window, void *stream);
/* The same fused step (DDPG.jl:195-234) for the envs [env_lo, env_lo + env_count) of the view only (the other envs are not touched).  Every env
 * draws its noise from (seed, tick, ITS index in the view) and the ring window is still defined on the whole batch, so stepping a
 * batch range by range -- in any order, on any streams -- leaves the same bytes as one shems_act_step_dev over the view.  The training
 * loop's order-exact pipelined mode steps the window's envs first with it (shems_train_steps). */
int shems_act_step_range_dev(const shems_view *v, const shems_act_params *p, int64_t env_lo, int64_t env_count, float *d_rewards_f32,
                             const shems_replay *ring, const shems_ring_window *window, void *stream);
/* scale_action (DDPG.jl:178-184) on device: d_a [n][2] in [-1,1] -> d_out [n][2] SoC targets in [0,1]. */
int shems_scale_action_dev(const float *d_a, int64_t n, float *d_out, void *stream);
/* The kernel shems_act_step_dev (grouped = 0) / shems_act_step_group_dev (grouped != 0) dispatches for n_envs envs, by the name a
 * profiler shows (e.g. "shems::k_act2", "shems::k_actg<1, 8, 1, 3>"), NUL-terminated into out[cap]: bench.py's roofline.kernel. */
int shems_act_step_kernel(int64_t n_envs, int grouped, char *out, int32_t cap);
/* The same for a learner group of envs_per_learner households per learner (a tile never straddles two learners; tiled != 0: the group's
 * W2 is read from its tiled regions, shems_act_step_group_tiled_dev). */
int shems_act_step_group_kernel(int64_t n_envs, int64_t envs_per_learner, int tiled, char *out, int32_t cap);
/* Number of workgroups shems_act_step_dev launches for n envs (length of d_block_reward). */
int shems_act_step_grid(int64_t n_envs, int64_t *out_blocks);

/* inference(env; track != 0) (memory_plotting_saving.jl:62-89 -> episode!(env; train = false, track, rng_ep = -1), DDPG.jl:186-242;
 * 80 such passes per job, DDPG_reinforce_charger_v1.jl:87-105) as ONE launch: every env of the view runs `nsteps` hours from
 * its current state (reset it first: shems_reset_dev(rng_minus1 = 1)), one workgroup per env, no host round trip per hour:
 *   track_mode > 0: a = scale_action(actor(normalize(s)))   -- the deterministic actor (act(...; train = false), DDPG.jl:148-184)
 *   track_mode < 0: a = action(env, track)                  -- the rule-based controller (shems_LU1.jl:318-340)
 * then step!(env, s, a; track) (shems_LU1.jl:343-485).  Env e uses the actor / s_min / s_max at p's pointers + e * actor_stride_bytes
 * (0: one actor for every env; a learner-group slab stride: 40 seeds x {last, best} in one launch); p may be NULL for track < 0.
 * d_results: the 23-column rows of shems_LU1.jl:476-478, [n_envs][nsteps][23] Float64 (results_env = -1) or [nsteps][23] of env
 * `results_env` only, or NULL.  d_returns [n_envs]: sum of rewards (DDPG.jl:223), or NULL.  The layers are tolerance-class
 * arithmetic (as shems_act_step_dev); step! is exact. */
int shems_track_dev(const shems_view *v, const shems_act_params *p, int64_t actor_stride_bytes, int track_mode, int32_t nsteps,
                    double *d_results, int64_t results_env, double *d_returns, void *stream);


/* -------------------------------------------------------- DDPG update -- */
/* replay() (DDPG.jl:121-145).  Batch <= 128 (BATCH_SIZE = 120 in the tuned config).  All pointers are device memory;
 * `ws` is a scratch block of shems_ddpg_workspace_floats() floats (zero it once).
 *
 * Single replica: shems_ddpg_update runs the whole function in five dependent launches (csrc/shems_ddpg.hip):
 *   getData (sample WITH replacement, MPS:31-42) -> normalize -> a' = actor_target(s'), q' = critic_target([s';a']),
 *   y = r + gamma(1-done)q' -> update_model!(critic, opt_crit, loss_crit, y, s, a) -> update_model!(actor, opt_act,
 *   loss_act, s) -> soft_update!(actor_target, actor), soft_update!(critic_target, critic).
 * ADAM and the soft target update of an element are applied by the workgroup that produced its gradient.
 *
 * Data-parallel replicas exchange gradients at two points; for them the same function is split there:
 *   shems_ddpg_critic_grad : ... -> d mse(critic([s;a]), y) / d critic  into grad_critic [129001]
 *   (all-reduce grad_critic across replicas here)
 *   shems_ddpg_critic_apply: ADAM(eta_crit) on critic, then soft_update!(critic_target, critic; tau)
 *   shems_ddpg_actor_grad  : d(-mean(critic([s; actor(s)]))) / d actor  into grad_actor [129002]
 *   (all-reduce grad_actor here)
 *   shems_ddpg_actor_apply : ADAM(eta_act) on actor, then soft_update!(actor_target, actor; tau)
 * Both forms run the same gradient kernels and the same ADAM arithmetic: with grad_scale = 1 they produce the same bits.
 * ADAM is Flux 0.12.1's: m,v float32 arrays, scalars in Float64, bias-correction powers beta^t are
 * host state passed by value (bp1 = beta1^t, bp2 = beta2^t for the t-th step, t >= 1). */
typedef struct shems_ddpg {
    float *actor, *critic, *actor_t, *critic_t;     /* [129002] [129001] [129002] [129001]        */
    float *m_actor, *v_actor, *m_critic, *v_critic; /* ADAM moments                                */
    float *grad_actor, *grad_critic;                /* gradient outputs (sum over the local batch / batch) */
    const float *s_min, *s_max;                     /* [9]                                         */
    float *ws;                                      /* workspace                                   */
    float *losses;                                  /* [2] out: critic mse, actor loss (-mean q)   */
    float gamma, tau;
    int32_t batch;                                  /* BATCH_SIZE (<= 128)                         */
    int32_t flags;                                  /* SHEMS_DDPG_* bits (0 = default)             */
} shems_ddpg;

/* flags: DEFER_ACTOR_E -- split form only: shems_ddpg_critic_grad leaves the actor's two E products (independent of the critic
 * update) out of its second launch; the caller issues them with shems_ddpg_actor_prepare between shems_ddpg_critic_grad and
 * shems_ddpg_actor_grad, typically right after starting the asynchronous all-reduce of grad_critic, so that they run under it. */
enum { SHEMS_DDPG_DEFER_ACTOR_E = 1 };

int shems_ddpg_workspace_floats(int64_t *out);
/* The whole replay() for one replica.  grad_actor / grad_critic still receive the complete gradients.  excl_pos / excl_count:
 * as shems_ddpg_critic_grad_ex (0, 0 = none).  d_publish: optional second copy [129002] of the updated actor (see
 * shems_ddpg_actor_apply_pub), or NULL. */
int shems_ddpg_update(const shems_ddpg *d, const shems_replay *ring, int64_t ring_len, uint64_t seed, uint32_t tick,
                      int64_t excl_pos, int64_t excl_count, double eta_crit, double bp1_crit, double bp2_crit,
                      double eta_act, double bp1_act, double bp2_act, float *d_publish, void *stream);
int shems_ddpg_critic_grad(const shems_ddpg *d, const shems_replay *ring, int64_t ring_len,
                           uint64_t seed, uint32_t tick, void *stream);
/* Pipelined variant for running replay() on a second stream WHILE the fused act/step kernel of the same vector step
 * inserts into the ring: slots [excl_pos, excl_pos + excl_count) (mod capacity) -- the window being written -- are
 * excluded from sampling, i.e. the minibatch is drawn from the buffer as it stood before this step's inserts.
 * excl_count = 0 is shems_ddpg_critic_grad. */
int shems_ddpg_critic_grad_ex(const shems_ddpg *d, const shems_replay *ring, int64_t ring_len,
                              uint64_t seed, uint32_t tick, int64_t excl_pos, int64_t excl_count, void *stream);
int shems_ddpg_actor_prepare(const shems_ddpg *d, void *stream);   /* see SHEMS_DDPG_DEFER_ACTOR_E */
int shems_ddpg_critic_apply(const shems_ddpg *d, double eta, double bp1, double bp2, double grad_scale,
                            void *stream);
int shems_ddpg_actor_grad(const shems_ddpg *d, void *stream);
int shems_ddpg_actor_apply(const shems_ddpg *d, double eta, double bp1, double bp2, double grad_scale,
                           void *stream);
/* Same, and also writes the updated actor into d_publish [129002] (a second copy the policy kernel of the NEXT
 * vector step reads while this stream already works on the next update). */
int shems_ddpg_actor_apply_pub(const shems_ddpg *d, double eta, double bp1, double bp2, double grad_scale,
                               float *d_publish, void *stream);
/* The minibatch indices of (seed, tick): host helper for tests (same Philox as the device). */
int shems_ddpg_sample_indices(uint64_t seed, uint32_t tick, int32_t batch, int64_t ring_len, int64_t *out);
/* ------------------------------------------ data-parallel replicas -- */
/* One process per GPU, each with its own env shard and ring, one learner replicated: replay() sums the gradients over the replicas at
 * its two exchange points (the reference has no collective: 40 independent processes,
 * RL-SHEMS_bs_scheduler_1179_08_on_01-98.sh:67-87).  shems_dp is one RCCL communicator; its all-reduces are issued IN THE CALLER'S
 * STREAM (no stream of their own, hence no dependency between queues -- 5-10 us each on this stack), RCCL itself is loaded with dlopen
 * at first use.  Rank 0 makes the 128-byte id and hands it to the others by any means (the Python host: torch.distributed broadcast);
 * shems_dp_create is collective (every rank calls it, on its own device). */
enum { SHEMS_DP_ID_BYTES = 128 };
typedef struct shems_dp shems_dp;
int shems_dp_unique_id(char *out128);
int shems_dp_create(const char *id128, int rank, int world, shems_dp **out);
int shems_dp_destroy(shems_dp *dp);
int shems_dp_info(const shems_dp *dp, int *rank, int *world, char *lib, int32_t cap);   /* lib: which librccl was loaded */
int shems_dp_allreduce_sum(shems_dp *dp, float *d_buf, int64_t n, void *stream);        /* in place, float32, in `stream` */
/* The same record WITHOUT RCCL (opt-in, SHEMS_DP=direct in the Python host): a direct one-shot exchange over peer-mapped memory.  Every
 * rank owns an inbox in fine-grained device memory, every peer maps it (hipIpcGetMemHandle / hipIpcOpenMemHandle: xGMI peers on one
 * node; two processes on one device for rehearsals); the ADAM sweep of each network pushes its slice of the local gradient into every
 * peer's inbox, stamps a per-slice epoch flag, waits (bounded) for the peers' slices and sums them in rank order -- no collective launch
 * ("direct one-shot ... over all 7 links").  world <= 8.  shems_dp_create_direct, then exchange the 128-byte handle
 * blocks by any means and shems_dp_direct_connect every peer on every rank BEFORE the first shems_ddpg_update_dp (a barrier of the
 * caller's).  A wait is bounded (shems_dp_direct_set_wait_ms, default 5 s).  A wait that gives up POISONS the record: that sweep and
 * every sweep already enqueued behind it apply nothing, and every later shems_ddpg_update_dp / shems_train_steps on the record
 * returns SHEMS_ERR_STATE -- a late peer makes the replicas fail loudly, it never lets them train on diverged
 * (shems_dp_direct_poisoned reads the sticky word without synchronising; shems_dp_direct_timeouts counts the waits that gave up).
 * Validated on ONE device only (two processes, tests/test_bench_gpu.py); never run across xGMI. */
int shems_dp_create_direct(int rank, int world, shems_dp **out);
int shems_dp_direct_handles(shems_dp *dp, char *out128);
int shems_dp_direct_connect(shems_dp *dp, int peer, const char *handles128);
int shems_dp_direct_timeouts(shems_dp *dp, int64_t *out, void *stream);
int shems_dp_direct_set_wait_ms(shems_dp *dp, int64_t ms);
int shems_dp_direct_poisoned(const shems_dp *dp, int32_t *out);
/* replay() of one replica: shems_ddpg_critic_grad_ex, all-reduce(grad_critic), shems_ddpg_critic_apply(grad_scale = 1 / world),
 * shems_ddpg_actor_grad, all-reduce(grad_actor), shems_ddpg_actor_apply_pub -- everything in `stream`.  dp == NULL: a single replica in
 * the split form (the bytes of shems_ddpg_update). */
int shems_ddpg_update_dp(const shems_ddpg *d, const shems_replay *ring, int64_t ring_len, uint64_t seed, uint32_t tick, int64_t excl_pos,
                         int64_t excl_count, double eta_crit, double bp1_crit, double bp2_crit, double eta_act, double bp1_act, double bp2_act,
                         float *d_publish, shems_dp *dp, void *stream);

/* ------------------------------------------------- the training loop -- */
/* The hour loop of episode! (DDPG.jl:195-234) for every env of a view, `k` vector steps enqueued by ONE call:
 *   per step t:  [t > 0 and t % ep_len == 0: episode += 1, reset!(env) with (env_seed, episode)  -- DDPG.jl:189-193]
 *                a = act(normalize(s)); step!(env, s, scale_action(a)); remember(...)               -- shems_act_step_dev
 *                updates_per_step x replay()                                                        -- shems_ddpg_update
 * exactly the calls a host loop over shems_act_step_dev / shems_ddpg_update would make, with the same arguments (the noise tick is
 * t, the ring window rotates by `window` envs per step, the sampler tick is the running update count, ADAM's beta powers advance in
 * Float64 after every update), so a run of the loop leaves the same bytes as that host loop.  What it removes is the host time per
 * launch: a Python / Julia caller pays several microseconds per foreign call, which at <= 8 192 envs is comparable to the kernels.
 *
 * mode SHEMS_LOOP_ORDERED   : program order on `stream` -- the reference's order.
 * mode SHEMS_LOOP_PIPELINED : replay(t) runs on `stream2` while the fused act/step launch of step t runs on `stream`; act(t) reads the
 *                             actor published by replay(t - 1) (actor_pub[t & 1], written by the ADAM sweep), exactly the actor the
 *                             ordered loop would use; the one deviation is that replay(t) samples the ring as it stood BEFORE step
 *                             t's inserts (the window being written is excluded, shems_ddpg_critic_grad_ex).
 * mode SHEMS_LOOP_PIPELINED_EXACT : the envs of step t's ring window are stepped FIRST (their own launch on `stream`), replay(t) then
 *                             runs on `stream2` and samples the ring WITH those inserts while the rest of the batch is stepped on
 *                             `stream`: the same bytes as SHEMS_LOOP_ORDERED.
 * The struct is caller-owned state: fields marked in/out are advanced by the call.  `sync` holds the loop's two signal-memory words (stream memory operations) and is created on
 * first use of a pipelined mode; release it with shems_train_loop_release (it does not touch the device buffers). */
enum { SHEMS_LOOP_ORDERED = 0, SHEMS_LOOP_PIPELINED = 1, SHEMS_LOOP_PIPELINED_EXACT = 2 };
typedef struct shems_train_loop {
    shems_view       view;
    shems_act_params act;            /* .tick is overwritten with t; .actor = the learner's actor (ddpg.actor)                  */
    shems_ddpg       ddpg;
    shems_replay     ring;
    float   *rewards_f32;            /* dev [n] or NULL: Float32(reward) of the last step                                        */
    float   *actor_pub[2];           /* pipelined modes: two dev [129002] copies of the actor, both equal to ddpg.actor on entry of the first call */
    int64_t  window;                 /* envs whose transitions enter the ring per vector step (<= n, <= ring.capacity)            */
    int64_t  ring_pushed;            /* in/out: transitions pushed so far (push position = ring_pushed mod capacity)              */
    int64_t  t;                      /* in/out: vector steps done so far                                                          */
    int64_t  updates;                /* in/out: replay() calls done so far (the sampler's tick)                                   */
    uint64_t env_seed;               /* reset!(env) key (shems_reset_seeded_dev)                                                  */
    uint64_t sample_seed;            /* getData key (shems_ddpg_update's seed)                                                    */
    uint32_t episode;                /* in/out                                                                                    */
    int32_t  ep_len;                 /* steps per episode (EP_LENGTH["train"] = 72)                                              */
    int32_t  updates_per_step;       /* 0 = no learning                                                                           */
    int32_t  mode;                   /* SHEMS_LOOP_*                                                                              */
    double   eta_crit, bp_crit[2];   /* in/out: ADAM(eta_crit) and its beta^t powers for the NEXT critic step                     */
    double   eta_act, bp_act[2];     /* in/out: same for the actor                                                                */
    void    *sync;                   /* opaque, NULL on first use                                                                 */
    shems_dp *dp;                    /* data-parallel replicas (SHEMS_LOOP_ORDERED only): replay() = shems_ddpg_update_dp; NULL = one replica */
} shems_train_loop;
int shems_train_steps(shems_train_loop *loop, int64_t k, void *stream, void *stream2);
/* Pipelined modes: make `stream` wait for everything the loop has in flight on stream2 (the caller then synchronises `stream`). */
int shems_train_loop_join(shems_train_loop *loop, void *stream, void *stream2);
int shems_train_loop_release(shems_train_loop *loop);

/* ------------------------------------------------- learner groups -- */
/* The thesis protocol trains many INDEPENDENT learners (40 seeds x 10 charger profiles, one OS process and one
 * batch-1 GPU stream each: run scripts + DDPG_reinforce_charger_v1.jl:10-47).  A learner
 * group advances `count` independent learners with the same launches: every device buffer of learner l -- all
 * pointers of shems_ddpg, shems_replay and the actor / s_min / s_max of shems_act_params -- is learner 0's pointer
 * plus l * stride_bytes (one slab per learner, carved identically), and learner l owns the envs
 * [l * envs_per_learner, (l + 1) * envs_per_learner) of the view.  Learners never exchange anything: per learner the
 * result is bit-identical to the single-learner entry points run on that learner's buffers with seed + l.
 *   act/step:  env i is driven by learner i / envs_per_learner's actor and normalisation; the ring window applies
 *              inside each learner's env block (n = envs_per_learner) and pushes into that learner's ring.
 *              envs_per_learner must be a multiple of 32 (tiles of 128 / 64 envs are used where they divide it).
 *   update:    grid z = learner; minibatch l is sampled with Philox key seed + l; ADAM scalars are shared (the
 *              learners advance in lockstep). */
typedef struct shems_group {
    int32_t count;             /* learners (>= 1)                                         */
    int32_t reserved;
    int64_t stride_bytes;      /* byte distance between consecutive learners' slabs (multiple of 16) */
    int64_t envs_per_learner;  /* act/step only                                           */
} shems_group;

int shems_act_step_group_dev(const shems_view *v, const shems_act_params *p0, const shems_group *g, float *d_a,
                             double *d_returns_acc, const shems_replay *ring0, const shems_ring_window *window,
                             void *stream);
int shems_ddpg_group_update(const shems_ddpg *d0, const shems_replay *ring0, const shems_group *g, int64_t ring_len,
                            uint64_t seed, uint32_t tick, double eta_crit, double bp1_crit, double bp2_crit,
                            double eta_act, double bp1_act, double bp2_act, void *stream);
/* The same function in its THROUGHPUT form (csrc/shems_gupd.hip): for groups wide enough that nothing is latency-bound any more
 * (the thesis protocol runs 40 seeds x 10 chargers = 400 learners, RL-SHEMS_bs_scheduler_1179_08_on_01-98.sh:67-87) -- plain
 * back-propagation on small tiles, 4 workgroups resident per CU, eight launches for the whole group.  Same sampler, same ADAM
 * arithmetic; the products sum in another order than the five-launch form above, so per learner the two agree to fp32
 * accumulation accuracy (every gradient block within 2e-6 of its max-abs of a float64 evaluation), not bit for bit.
 * flags: SHEMS_TP_STORE_GRAD also leaves the gradients in grad_actor / grad_critic (otherwise they are never materialised;
 * the b3 / layer-1 / W2 / b2 / W3 elements are consumed by ADAM in the lane that finishes them).  Uses d0->ws with a layout of
 * its own: do not mix the two forms on one workspace within one update. */
enum { SHEMS_TP_STORE_GRAD = 1 };
int shems_ddpg_group_update_tp(const shems_ddpg *d0, const shems_replay *ring0, const shems_group *g, int64_t ring_len,
                               uint64_t seed, uint32_t tick, double eta_crit, double bp1_crit, double bp2_crit,
                               double eta_act, double bp1_act, double bp2_act, int32_t flags, void *stream);
/* The throughput form on a TILED working layout of the layer-2 state (round 6).  The W2-gradient launches of the form above are its
 * HBM stream -- 32 B in and out per W2 parameter: ADAM moments (Flux.Optimise.ADAM state, DDPG.jl:105-108), parameter and target
 * (soft_update!, DDPG.jl:99-103) -- and in Flux order ([in][out] rows of 500 floats) a 64 x 64 tile of it is 4 x 64 pieces of 256
 * bytes: 3.85 TB/s where one contiguous 64 KB piece per tile reaches 4.78 TB/s (tools/micro/adam_stream.hip).  A tiled region holds,
 * per network, W2 / m / v / target as [4 k-tiles][8 n-tiles][m | v | p | target][64][64] floats (rows / columns padded to 256 / 512
 * with zeros), SHEMS_W2T_FLOATS per learner, at learner 0's pointer + l * stride_bytes like every other block.  While a group
 * trains through these entry points the tiled regions hold the CURRENT layer-2 state and the W2 ranges of the Flux-order blocks
 * (shems_ddpg.actor / critic / actor_t / critic_t / m_* / v_*) are stale; everything else (layer 1, b2, W3, b3) stays in the
 * Flux-order blocks.  shems_group_w2_to_tiled / _to_flux copy the W2 ranges one way or the other (checkpoints, set / get of
 * parameters, the single-learner entry points and the latency form all speak Flux order).  Same arithmetic and summation order as
 * shems_ddpg_group_update_tp: the two leave the same values (tests/test_group_gpu.py holds both to the float64 oracle and to each
 * other bit for bit). */
enum { SHEMS_W2T_FLOATS = 32 * 4 * 64 * 64 };
typedef struct shems_group_w2t {
    float *actor;              /* dev [SHEMS_W2T_FLOATS] of learner 0: actor W2, its moments and actor_target W2  */
    float *critic;             /* dev [SHEMS_W2T_FLOATS] of learner 0: critic ...                                   */
} shems_group_w2t;
int shems_group_w2_to_tiled(const shems_ddpg *d0, const shems_group *g, const shems_group_w2t *t, void *stream);
int shems_group_w2_to_flux(const shems_ddpg *d0, const shems_group *g, const shems_group_w2t *t, void *stream);
int shems_ddpg_group_update_tiled(const shems_ddpg *d0, const shems_replay *ring0, const shems_group *g, const shems_group_w2t *t,
                                  int64_t ring_len, uint64_t seed, uint32_t tick, double eta_crit, double bp1_crit,
                                  double bp2_crit, double eta_act, double bp1_act, double bp2_act, int32_t flags, void *stream);
/* shems_act_step_group_dev with every learner's actor W2 read from its tiled region (t->actor; t->critic is not used). */
int shems_act_step_group_tiled_dev(const shems_view *v, const shems_act_params *p0, const shems_group *g, const shems_group_w2t *t,
                                   float *d_a, double *d_returns_acc, const shems_replay *ring0, const shems_ring_window *window,
                                   void *stream);
int shems_ddpg_group_critic_grad(const shems_ddpg *d0, const shems_replay *ring0, const shems_group *g,
                                 int64_t ring_len, uint64_t seed, uint32_t tick, void *stream);
int shems_ddpg_group_critic_apply(const shems_ddpg *d0, const shems_group *g, double eta, double bp1, double bp2,
                                  void *stream);
int shems_ddpg_group_actor_grad(const shems_ddpg *d0, const shems_group *g, void *stream);
int shems_ddpg_group_actor_apply(const shems_ddpg *d0, const shems_group *g, double eta, double bp1, double bp2,
                                 void *stream);
/* min_max_buffer for every learner of the group (learner l: Philox key seed + l). */
int shems_minmax_group_dev(const shems_replay *ring0, const shems_group *g, int64_t ring_len, int64_t count,
                           uint64_t seed, float *d_s_min0, float *d_s_max0, void *stream);

/* Parameter noise, noise_type "pn" (struct ParamNoise input.jl:210-215; add_perturb! DDPG.jl:89-96;
 * adapt_param_noise! DDPG.jl:74-87).  The reference adds ONE scalar draw N(mu, sigma_current) to every parameter
 * array of a copy of the actor (sample_noise(pn, rng) re-seeds before each draw); act() then evaluates the copy
 * without action noise (pass it as shems_act_params.actor with train = 0), and replay() adapts sigma_current from
 * the distance between the two actors' outputs on the sampled minibatch.
 *   perturb:    d_perturbed[i] = d_params[i] + shift, i < n
 *   batch_obs:  the s rows of the minibatch the last shems_ddpg_critic_grad(_ex) sampled -> d_obs [batch][9]
 *   distance:   d_out[0] = sqrt(mean((d_a - d_b)^2)) over `count` floats */
/* BATCH_SIZE above 128 (the tuned template's 150, input.jl's 200): one update pass holds 128 minibatch columns, so replay() runs the
 * gradient calls once per SUB-BATCH (each on its own workspace and gradient buffer, `batch` = the sub-batch size: its gradient is
 * the mean over the sub-batch) and combines them, acc = w_acc * acc + w_g * g with w_g = sub-batch size / BATCH_SIZE, before the one
 * ADAM step -- Flux.mse / -mean(q) over the whole minibatch (DDPG.jl:134-140) is exactly that weighted mean of sub-batch means.
 * w_acc == 0: acc is overwritten (never read). */
int shems_ddpg_combine_dev(float *d_acc, const float *d_g, int64_t n, float w_acc, float w_g, void *stream);

int shems_ddpg_perturb_dev(const float *d_params, float *d_perturbed, int64_t n, float shift, void *stream);
int shems_ddpg_batch_obs_dev(const shems_ddpg *d, const shems_replay *ring, float *d_obs, void *stream);
int shems_action_distance_dev(const float *d_a, const float *d_b, int64_t count, float *d_out, void *stream);
/* min_max_buffer (MPS:50-53): minimum/maximum of s over a bootstrap sample of `count` ring entries. */
int shems_minmax_dev(const shems_replay *ring, int64_t ring_len, int64_t count, uint64_t seed,
                     float *d_s_min, float *d_s_max, void *stream);

/* ------------------------------------------- networks wider than (250, 500) -- */
/* The reference's grids hold one architecture larger than the tuned one: (L1, L2) = (300, 600) (input09_08_on_01-09_eval.jl:62-66,
 * input.jl:58-66; Dense widths of DDPG.jl:30-46).  Smaller networks run on the entry points above zero-padded into the (250, 500)
 * layout; a larger one runs through the entry points below (csrc/shems_wide.hip): the same functions, every layer one fp32-MFMA matrix
 * product as in the reference's Flux / CUBLAS path, parameters in the flat Flux layout OF THAT SIZE
 * (W1[in][l1] b1[l1] W2[l1][l2] b2[l2] W3[l2][out] b3[out]; shems_wide_params gives the counts), the same Philox streams (noise, minibatch
 * slots) as the tuned path.  1 <= l1, l2 <= 4096.  Not the headline path: tolerance-class like the tuned kernels, an order of magnitude
 * slower per update. */
int shems_wide_params(int32_t l1, int32_t l2, int64_t *n_actor, int64_t *n_critic);
/* floats of shems_ddpg.ws for the update entry points / of d_ws for the act entry points with m observations */
int shems_wide_workspace_floats(int32_t l1, int32_t l2, int64_t *out);
int shems_wide_act_workspace_floats(int32_t l1, int32_t l2, int64_t m, int64_t *out);
/* act() (DDPG.jl:148-176) as shems_actor_forward_dev, and the fused vector step of episode! (DDPG.jl:195-234) as shems_act_step_dev
 * (without the per-workgroup reward sums): normalize + three matrix products + one launch for tanh / noise / clamp / scale_action /
 * step! / remember. */
int shems_wide_actor_forward_dev(const shems_act_params *p, int32_t l1, int32_t l2, const float *d_obs, int64_t m, float *d_a, float *d_ws,
                                 void *stream);
int shems_wide_act_step_dev(const shems_view *v, const shems_act_params *p, int32_t l1, int32_t l2, float *d_ws, float *d_a, double *d_rewards,
                            float *d_rewards_f32, double *d_returns_acc, const shems_replay *ring, const shems_ring_window *window,
                            void *stream);
/* replay() (DDPG.jl:121-145) in the split form of shems_ddpg_critic_grad_ex / _critic_apply / _actor_grad / _actor_apply_pub (the
 * caller may all-reduce grad_critic / grad_actor in between); every buffer of shems_ddpg holds the wide network's parameter count. */
int shems_wide_critic_grad_ex(const shems_ddpg *d, int32_t l1, int32_t l2, const shems_replay *ring, int64_t ring_len, uint64_t seed,
                              uint32_t tick, int64_t excl_pos, int64_t excl_count, void *stream);
int shems_wide_critic_apply(const shems_ddpg *d, int32_t l1, int32_t l2, double eta, double bp1, double bp2, double grad_scale, void *stream);
int shems_wide_actor_grad(const shems_ddpg *d, int32_t l1, int32_t l2, void *stream);
int shems_wide_actor_apply_pub(const shems_ddpg *d, int32_t l1, int32_t l2, double eta, double bp1, double bp2, double grad_scale,
                               float *d_publish, void *stream);
/* ring slots [batch] of the minibatch the last shems_wide_critic_grad_ex drew, to host memory (adapt_param_noise!, DDPG.jl:74-87);
 * synchronises the stream */
int shems_wide_batch_slots(const shems_ddpg *d, int32_t l1, int32_t l2, int32_t *out_slots, void *stream);
/* inference(env; track > 0) (memory_plotting_saving.jl:62-89) as shems_track_dev, for actors of hidden sizes (l1, l2) */
int shems_wide_track_dev(const shems_view *v, const shems_act_params *p, int32_t l1, int32_t l2, int64_t actor_stride_bytes,
                         int32_t nsteps, double *d_results, int64_t results_env, double *d_returns, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SHEMS_HIP_H */
