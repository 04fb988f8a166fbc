// shems_env.hip -- batched SHEMS environment kernels for gfx950 + the env half of the C ABI.
//
// Replaces, for N parallel households, the reference's scalar
//   reset!/reset_state!  shems_LU1.jl:206-262      step!        shems_LU1.jl:343-485
//   next_state!          shems_LU1.jl:264-281      action x2    shems_LU1.jl:283-340
// Layout in HBM: obs [N][9] f32 (Julia's 9xN column-major), idx/step [N] i32, a u16 config id per
// env, configs [n_cfg] (48 B), tables [rows][8] f32.  One thread = one env; a 256-env workgroup
// moves its 9216-byte obs slab through LDS so that global loads/stores are fully coalesced dword
// streams (the per-env 36-byte rows are read back at stride 9 dwords: conflict-free on 32 banks).
// Table rows are gathered as 2 x 16 B from L2 (a 4320-row table is 138 KB).
//
// Compile with -ffp-contract=off (see shems_core.h).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "shems_env_dev.h"
#include "shems_internal.h"

namespace shems {

// ------------------------------------------------------------------ step! --
__global__ __launch_bounds__(kBlock) void k_step(shems_view v, const float *__restrict__ actions, int track_mode,
                                                 double *__restrict__ rewards, float *__restrict__ rewards_f32,
                                                 double *__restrict__ results, double *__restrict__ block_reward)
{
    __shared__ __attribute__((aligned(16))) float tile[kBlock * SHEMS_NSTATE];
    __shared__ double red[4];
    const int64_t base = (int64_t)blockIdx.x * kBlock;
    const int64_t i = base + threadIdx.x;
    const bool live = i < v.n_envs;

    slab_load(tile, v.obs, base, v.n_envs);
    float2 a = make_float2(0.f, 0.f);
    int32_t idx = 0, step = 0;
    if (live) {
        a = reinterpret_cast<const float2 *>(actions)[i];
        idx = v.idx[i];
        step = v.step[i];
    }
    __syncthreads();

    double reward = 0.0;
    if (live) {
        const shems_config c = load_cfg(v, i);
        float obs[SHEMS_NSTATE], pre[SHEMS_NSTATE];
#pragma unroll
        for (int k = 0; k < SHEMS_NSTATE; ++k) { obs[k] = tile[threadIdx.x * SHEMS_NSTATE + k]; pre[k] = obs[k]; }
        StepFlows f;
        float B, EV, Bt, EVt;
        if (env_advance(c, v.tables, obs, idx, step, a.x, a.y, track_mode, reward, f, B, EV, Bt, EVt)) {
#pragma unroll
            for (int k = 0; k < SHEMS_NSTATE; ++k) tile[threadIdx.x * SHEMS_NSTATE + k] = obs[k];
            v.idx[i] = idx;
            v.step[i] = step;
            if (rewards) rewards[i] = reward;
            if (rewards_f32) rewards_f32[i] = (float)reward;
            if (results) write_results(results + i * SHEMS_NRESULT, idx, pre, EVt, EV, reward, f, B, Bt);
        } else {
            reward = 0.0;
            raise(v.err, SHEMS_ERR_INDEX);
        }
    }
    __syncthreads();
    slab_store(tile, v.obs, base, v.n_envs);
    if (block_reward) {
        const double s = block_sum(reward, red);
        if (threadIdx.x == 0) block_reward[blockIdx.x] = s;
    }
}

// ----------------------------------------------------------------- action --
__global__ __launch_bounds__(kBlock) void k_action(shems_view v, const float *__restrict__ targets, int rule_based,
                                                   float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float tile[kBlock * SHEMS_NSTATE];
    const int64_t base = (int64_t)blockIdx.x * kBlock;
    const int64_t i = base + threadIdx.x;
    slab_load(tile, v.obs, base, v.n_envs);
    __syncthreads();
    if (i >= v.n_envs) return;
    const shems_config c = load_cfg(v, i);
    const float *o = tile + threadIdx.x * SHEMS_NSTATE;
    const EnvIn s{o[0], o[1], o[2], o[3], o[4], o[5]};
    float B, EV;
    if (rule_based) {
        action_rule(c, s, B, EV);
    } else {
        const float2 t = reinterpret_cast<const float2 *>(targets)[i];
        action_drl(c, s, t.x, t.y, B, EV);
    }
    reinterpret_cast<float2 *>(out)[i] = make_float2(B, EV);
}

// ----------------------------------------------------------------- reset! --
__device__ __forceinline__ bool env_reset(const shems_config &c, const float *tables, int maxsteps, int rng_minus1,
                                          int32_t idx0, float soc_b0, float (&obs)[SHEMS_NSTATE], int32_t &idx)
{
    if (rng_minus1) {                            // LU1:220-222
        obs[0] = (float)(0.5 * (double)(0.0f + c.soc_max));
        idx = 1;
    } else {                                     // LU1:224-246
        obs[0] = soc_b0;
        const int64_t row0 = c.table_row0;
        idx = resolve_start(idx0, c.nrow, maxsteps, [&](int32_t r) { return load_h(tables, row0, r); });
    }
    if (idx < 1 || idx > c.nrow) return false;
    const Row r = load_row(tables, c.table_row0, idx);   // LU1:251-260
    obs[1] = r.soc_ev; obs[2] = r.h; obs[3] = r.d_e; obs[4] = r.g_e; obs[5] = r.p_buy;
    obs[6] = r.h_cos; obs[7] = r.h_sin; obs[8] = r.season;
    return true;
}

__global__ __launch_bounds__(kBlock) void k_reset(shems_view v, int rng_minus1, const int32_t *__restrict__ idx0,
                                                  const float *__restrict__ soc_b0, int seeded, uint64_t seed,
                                                  uint32_t episode)
{
    __shared__ __attribute__((aligned(16))) float tile[kBlock * SHEMS_NSTATE];
    const int64_t base = (int64_t)blockIdx.x * kBlock;
    const int64_t i = base + threadIdx.x;
    if (i < v.n_envs) {
        const shems_config c = load_cfg(v, i);
        int32_t d_idx = 1;
        float d_soc = 0.0f;
        if (!rng_minus1) {
            if (seeded) {
                const u32x4 x = philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), episode, kStreamReset,
                                              (uint32_t)seed, (uint32_t)(seed >> 32));
                const uint32_t span = (uint32_t)(c.nrow - v.maxsteps);       // rand(1:(nrow - maxsteps))
                d_idx = span > 0 ? 1 + (int32_t)(x.x % span) : 0;
                d_soc = u01_24(x.y) * c.soc_max;                             // Uniform(soc_min = 0, soc_max)
            } else {
                d_idx = idx0[i];
                d_soc = soc_b0[i];
            }
        }
        float obs[SHEMS_NSTATE];
        int32_t idx = 0;
        if (env_reset(c, v.tables, v.maxsteps, rng_minus1, d_idx, d_soc, obs, idx)) {
#pragma unroll
            for (int k = 0; k < SHEMS_NSTATE; ++k) tile[threadIdx.x * SHEMS_NSTATE + k] = obs[k];
            v.idx[i] = idx;
        } else {
            // leave a defined state: ShemsState() of LU1:115, idx = 1
            const float d[SHEMS_NSTATE] = {0.f, 0.f, -1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 1.f};
#pragma unroll
            for (int k = 0; k < SHEMS_NSTATE; ++k) tile[threadIdx.x * SHEMS_NSTATE + k] = d[k];
            v.idx[i] = 1;
            raise(v.err, SHEMS_ERR_INDEX);
        }
        v.step[i] = 0;                           // LU1:210
    }
    __syncthreads();
    slab_store(tile, v.obs, base, v.n_envs);
}

// ---------------------------------------------------------------- rollout --
// nsteps x { a = policy(env); step! } in one launch; env state lives in registers, only the episode
// return (and, optionally, the replay transitions) leave the chip.
__global__ __launch_bounds__(kBlock) void k_rollout(shems_view v, int policy, int nsteps, uint64_t seed,
                                                    double *__restrict__ returns, shems_replay ring,
                                                    int64_t ring_pos, int64_t ring_envs, int64_t first_kept)
{
    __shared__ __attribute__((aligned(16))) float tile[kBlock * SHEMS_NSTATE];
    const int64_t base = (int64_t)blockIdx.x * kBlock;
    const int64_t i = base + threadIdx.x;
    const bool live = i < v.n_envs;
    slab_load(tile, v.obs, base, v.n_envs);
    __syncthreads();
    if (live) {
        const shems_config c = load_cfg(v, i);
        float obs[SHEMS_NSTATE];
#pragma unroll
        for (int k = 0; k < SHEMS_NSTATE; ++k) obs[k] = tile[threadIdx.x * SHEMS_NSTATE + k];
        int32_t idx = v.idx[i], step = v.step[i];
        double total = 0.0;                      // reward_eps (DDPG.jl:190, 223): Float64 after the first add
        bool ok = true;
        for (int t = 0; t < nsteps && ok; ++t) {
            float a0, a1, raw0 = 0.f, raw1 = 0.f;
            int mode;
            if (policy == SHEMS_ROLLOUT_RULE) {  // DDPG.jl:209-211
                const EnvIn s{obs[0], obs[1], obs[2], obs[3], obs[4], obs[5]};
                action_rule(c, s, a0, a1);
                mode = SHEMS_TRACK_RULE;
            } else {                             // MPS:17-19: a = Float32.(rand(2) .* 2 .- 1); scale_action(a)
                const u32x4 x = philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)step, kStreamRandAct,
                                              (uint32_t)seed, (uint32_t)(seed >> 32));
                raw0 = (float)((double)x.x * (1.0 / 4294967296.0) * 2.0 - 1.0);
                raw1 = (float)((double)x.y * (1.0 / 4294967296.0) * 2.0 - 1.0);
                a0 = scale_action(raw0);
                a1 = scale_action(raw1);
                mode = SHEMS_TRACK_OFF;
            }
            float pre[SHEMS_NSTATE];
#pragma unroll
            for (int k = 0; k < SHEMS_NSTATE; ++k) pre[k] = obs[k];
            double reward;
            StepFlows f;
            float B, EV, Bt, EVt;
            ok = env_advance(c, v.tables, obs, idx, step, a0, a1, mode, reward, f, B, EV, Bt, EVt);
            if (!ok) { raise(v.err, SHEMS_ERR_INDEX); break; }
            total += reward;
            const int64_t ord = i * (int64_t)nsteps + t;
            if (i < ring_envs && ord >= first_kept) {   // remember(s, a, r, s', done)  MPS:46-47, episode-major slots
                const int64_t slot = (ring_pos + ord) % ring.capacity;
                float *ps = ring.s + slot * SHEMS_NSTATE, *p2 = ring.s2 + slot * SHEMS_NSTATE;
#pragma unroll
                for (int k = 0; k < SHEMS_NSTATE; ++k) { ps[k] = pre[k]; p2[k] = obs[k]; }
                ring.a[slot * 2 + 0] = (policy == SHEMS_ROLLOUT_RULE) ? a0 : raw0;
                ring.a[slot * 2 + 1] = (policy == SHEMS_ROLLOUT_RULE) ? a1 : raw1;
                ring.r[slot] = (float)reward;
                ring.done[slot] = 0;             // finished() is always false, LU1:487-502
            }
        }
#pragma unroll
        for (int k = 0; k < SHEMS_NSTATE; ++k) tile[threadIdx.x * SHEMS_NSTATE + k] = obs[k];
        v.idx[i] = idx;
        v.step[i] = step;
        if (returns) returns[i] = total;
    }
    __syncthreads();
    slab_store(tile, v.obs, base, v.n_envs);
}

__global__ __launch_bounds__(kBlock) void k_scale_action(const float *__restrict__ a, int64_t n2, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n2) out[i] = scale_action(a[i]);
}

}  // namespace shems

// =================================================================== C ABI ==
using namespace shems;

static inline unsigned grid_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

int shems::check_view(const shems_view *v, const char *fn)
{
    if (!v || v->n_envs <= 0 || !v->obs || !v->idx || !v->step || !v->cfgs || !v->tables || v->n_cfg < 1)
        return set_error(SHEMS_ERR_ARG, "%s: invalid shems_view (NULL buffer or n_envs <= 0)", fn);
    if (v->n_cfg > 1 && !v->cfg_of_env)
        return set_error(SHEMS_ERR_ARG, "%s: shems_view has %d configs but no cfg_of_env map", fn, (int)v->n_cfg);
    return SHEMS_OK;
}

extern "C" {

int shems_step_dev(const shems_view *v, const float *d_actions, int track_mode, double *d_rewards,
                   float *d_rewards_f32, double *d_results, double *d_block_reward, void *stream)
{
    if (int rc = check_view(v, "shems_step_dev")) return rc;
    if (!d_actions) return set_error(SHEMS_ERR_ARG, "shems_step_dev: d_actions is NULL");
    hipLaunchKernelGGL(k_step, dim3(grid_for(v->n_envs)), dim3(kBlock), 0, (hipStream_t)stream, *v, d_actions,
                       track_mode, d_rewards, d_rewards_f32, d_results, d_block_reward);
    return hip_ok(hipGetLastError(), "k_step launch");
}

int shems_action_dev(const shems_view *v, const float *d_targets, int rule_based, float *d_out, void *stream)
{
    if (int rc = check_view(v, "shems_action_dev")) return rc;
    if (!d_out || (!rule_based && !d_targets)) return set_error(SHEMS_ERR_ARG, "shems_action_dev: NULL buffer");
    hipLaunchKernelGGL(k_action, dim3(grid_for(v->n_envs)), dim3(kBlock), 0, (hipStream_t)stream, *v, d_targets,
                       rule_based, d_out);
    return hip_ok(hipGetLastError(), "k_action launch");
}

int shems_reset_dev(const shems_view *v, int rng_minus1, const int32_t *d_idx0, const float *d_soc_b0, void *stream)
{
    if (int rc = check_view(v, "shems_reset_dev")) return rc;
    if (!rng_minus1 && (!d_idx0 || !d_soc_b0))
        return set_error(SHEMS_ERR_ARG, "shems_reset_dev: idx0/soc_b0 required unless rng == -1");
    hipLaunchKernelGGL(k_reset, dim3(grid_for(v->n_envs)), dim3(kBlock), 0, (hipStream_t)stream, *v, rng_minus1,
                       d_idx0, d_soc_b0, 0, (uint64_t)0, 0u);
    return hip_ok(hipGetLastError(), "k_reset launch");
}

int shems_reset_seeded_dev(const shems_view *v, uint64_t seed, uint32_t episode, void *stream)
{
    if (int rc = check_view(v, "shems_reset_seeded_dev")) return rc;
    hipLaunchKernelGGL(k_reset, dim3(grid_for(v->n_envs)), dim3(kBlock), 0, (hipStream_t)stream, *v, 0,
                       (const int32_t *)nullptr, (const float *)nullptr, 1, seed, episode);
    return hip_ok(hipGetLastError(), "k_reset launch");
}

int shems_scale_action_dev(const float *d_a, int64_t n, float *d_out, void *stream)
{
    if (!d_a || !d_out || n <= 0) return set_error(SHEMS_ERR_ARG, "shems_scale_action_dev: bad arguments");
    hipLaunchKernelGGL(k_scale_action, dim3(grid_for(2 * n)), dim3(kBlock), 0, (hipStream_t)stream, d_a, 2 * n, d_out);
    return hip_ok(hipGetLastError(), "k_scale_action launch");
}

int shems_rollout_dev(const shems_view *v, int policy, int32_t nsteps, uint64_t seed, double *d_returns,
                      const shems_replay *ring, int64_t ring_pos, int64_t ring_envs, void *stream)
{
    if (int rc = check_view(v, "shems_rollout_dev")) return rc;
    if (nsteps < 0 || (policy != SHEMS_ROLLOUT_RULE && policy != SHEMS_ROLLOUT_RANDOM))
        return set_error(SHEMS_ERR_ARG, "shems_rollout_dev: bad policy or nsteps");
    shems_replay r;
    std::memset(&r, 0, sizeof r);
    if (ring) {
        if (ring->capacity <= 0 || !ring->s || !ring->a || !ring->r || !ring->s2 || !ring->done)
            return set_error(SHEMS_ERR_ARG, "shems_rollout_dev: incomplete replay ring");
        r = *ring;
    }
    int64_t n_ring = 0, first_kept = 0;
    if (ring) {
        n_ring = (ring_envs <= 0 || ring_envs > v->n_envs) ? v->n_envs : ring_envs;
        first_kept = n_ring * (int64_t)nsteps - ring->capacity;
        if (first_kept < 0) first_kept = 0;
    }
    hipLaunchKernelGGL(k_rollout, dim3(grid_for(v->n_envs)), dim3(kBlock), 0, (hipStream_t)stream, *v, policy,
                       (int)nsteps, seed, d_returns, r, ring_pos, n_ring, first_kept);
    return hip_ok(hipGetLastError(), "k_rollout launch");
}

}  // extern "C"
