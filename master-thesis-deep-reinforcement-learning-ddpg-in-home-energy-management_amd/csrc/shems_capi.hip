// shems_capi.hip -- handle half of the C ABI (include/shems_hip.h): owns the device buffers of N
// parallel Shems instances and moves host arrays in/out around the kernels of shems_env.hip.
// This is the layer the Julia `ccall` shim and the Python ctypes mirror bind.  There is NO CPU
// fallback: without a HIP device every entry point fails with SHEMS_ERR_NODEVICE / SHEMS_ERR_HIP.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "shems_internal.h"

namespace shems {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int hip_ok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return SHEMS_OK;
    return set_error(e == hipErrorNoDevice ? SHEMS_ERR_NODEVICE : SHEMS_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

}  // namespace shems

using namespace shems;

struct shems_env {
    int64_t n = 0;
    int32_t maxsteps = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    shems_view v{};
    // scratch for host-array calls
    float *d_act = nullptr;        // [n][2]
    float *d_out2 = nullptr;       // [n][2]
    double *d_rew = nullptr;       // [n]
    double *d_res = nullptr;       // [n][23] (lazily allocated)
    float *d_track_par = nullptr;  // shems_track: actor [129002] | pad | s_min [9] | pad | s_max [9] (lazily allocated)
    double *d_track_res = nullptr; // shems_track: [nsteps][23] of env 0, + [n] returns behind it (lazily allocated, grown on demand)
    int64_t track_res_steps = 0;
    int32_t *d_idx0 = nullptr;     // [n]
    float *d_soc0 = nullptr;       // [n]
    bool have_tables = false, have_cfgs = false;
};

#define HIP_TRY(expr)                                              \
    do {                                                           \
        if (int rc_ = hip_ok((expr), #expr)) return rc_;           \
    } while (0)

static int need_ready(const shems_env *e, const char *fn)
{
    if (!e) return set_error(SHEMS_ERR_ARG, "%s: env is NULL", fn);
    if (!e->have_tables || !e->have_cfgs)
        return set_error(SHEMS_ERR_STATE, "%s: call shems_set_tables and shems_set_configs first", fn);
    return SHEMS_OK;
}

template <class T>
static int dev_alloc(T **p, size_t count)
{
    return hip_ok(hipMalloc((void **)p, count * sizeof(T)), "hipMalloc");
}

extern "C" {

int shems_abi_version(void) { return SHEMS_ABI_VERSION; }
const char *shems_last_error(void) { return g_err; }

int shems_device_count(int *out_count)
{
    if (!out_count) return set_error(SHEMS_ERR_ARG, "shems_device_count: NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *out_count = 0; return hip_ok(e, "hipGetDeviceCount"); }
    *out_count = n;
    return SHEMS_OK;
}

int shems_create(int64_t n_envs, int32_t maxsteps, int device, shems_env **out)
{
    if (!out) return set_error(SHEMS_ERR_ARG, "shems_create: out is NULL");
    *out = nullptr;
    if (n_envs <= 0 || maxsteps <= 0) return set_error(SHEMS_ERR_ARG, "shems_create: n_envs and maxsteps must be > 0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_error(SHEMS_ERR_NODEVICE, "shems_create: no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return set_error(SHEMS_ERR_ARG, "shems_create: device %d out of range [0,%d)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    shems_env *e = new (std::nothrow) shems_env();
    if (!e) return set_error(SHEMS_ERR_NOMEM, "shems_create: out of host memory");
    e->n = n_envs; e->maxsteps = maxsteps; e->device = device;
    int rc = hip_ok(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking), "hipStreamCreate");
    e->own_stream = (rc == SHEMS_OK);
    const size_t n = (size_t)n_envs;
    uint16_t *cfg_of = nullptr;
    if (!rc) rc = dev_alloc(&e->v.obs, n * SHEMS_NSTATE);
    if (!rc) rc = dev_alloc(&e->v.idx, n);
    if (!rc) rc = dev_alloc(&e->v.step, n);
    if (!rc) rc = dev_alloc(&cfg_of, n);
    if (!rc) rc = dev_alloc(&e->v.err, 1);
    if (!rc) rc = dev_alloc(&e->d_act, n * 2);
    if (!rc) rc = dev_alloc(&e->d_out2, n * 2);
    if (!rc) rc = dev_alloc(&e->d_rew, n);
    if (!rc) rc = dev_alloc(&e->d_idx0, n);
    if (!rc) rc = dev_alloc(&e->d_soc0, n);
    e->v.cfg_of_env = cfg_of;
    if (!rc) rc = hip_ok(hipMemsetAsync(e->v.err, 0, sizeof(int32_t), e->stream), "memset err");
    if (!rc) rc = hip_ok(hipMemsetAsync(cfg_of, 0, n * sizeof(uint16_t), e->stream), "memset cfg_of_env");
    if (!rc) rc = hip_ok(hipMemsetAsync(e->v.step, 0, n * sizeof(int32_t), e->stream), "memset step");
    if (!rc) rc = hip_ok(hipMemsetAsync(e->v.obs, 0, n * SHEMS_NSTATE * sizeof(float), e->stream), "memset obs");
    if (!rc) rc = hip_ok(hipMemsetAsync(e->v.idx, 0, n * sizeof(int32_t), e->stream), "memset idx");
    if (!rc) rc = hip_ok(hipStreamSynchronize(e->stream), "sync");
    e->v.n_envs = n_envs; e->v.maxsteps = maxsteps; e->v.n_cfg = 0;
    if (rc) { shems_destroy(e); return rc; }
    *out = e;
    return SHEMS_OK;
}

int shems_destroy(shems_env *e)
{
    if (!e) return SHEMS_OK;
    hipSetDevice(e->device);
    if (e->stream) hipStreamSynchronize(e->stream);
    hipFree(e->v.obs); hipFree(e->v.idx); hipFree(e->v.step); hipFree((void *)e->v.cfg_of_env);
    hipFree((void *)e->v.cfgs); hipFree((void *)e->v.tables); hipFree(e->v.err);
    hipFree(e->d_act); hipFree(e->d_out2); hipFree(e->d_rew); hipFree(e->d_res); hipFree(e->d_idx0); hipFree(e->d_soc0);
    hipFree(e->d_track_par); hipFree(e->d_track_res);
    if (e->own_stream && e->stream) hipStreamDestroy(e->stream);
    delete e;
    return SHEMS_OK;
}

int shems_n_envs(const shems_env *e, int64_t *out)
{
    if (!e || !out) return set_error(SHEMS_ERR_ARG, "shems_n_envs: NULL");
    *out = e->n;
    return SHEMS_OK;
}

int shems_set_stream(shems_env *e, void *stream)
{
    if (!e) return set_error(SHEMS_ERR_ARG, "shems_set_stream: env is NULL");
    if (e->stream) hipStreamSynchronize(e->stream);
    if (e->own_stream && e->stream) hipStreamDestroy(e->stream);
    e->stream = (hipStream_t)stream;
    e->own_stream = false;
    return SHEMS_OK;
}

int shems_set_tables(shems_env *e, const float *rows, int64_t total_rows)
{
    if (!e || !rows || total_rows < 2) return set_error(SHEMS_ERR_ARG, "shems_set_tables: need >= 2 rows");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->v.tables) { hipFree((void *)e->v.tables); e->v.tables = nullptr; }
    float *d = nullptr;
    const size_t bytes = (size_t)total_rows * SHEMS_NCOL * sizeof(float);
    HIP_TRY(hipMalloc((void **)&d, bytes));
    e->v.tables = d;
    HIP_TRY(hipMemcpyAsync(d, rows, bytes, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->v.total_rows = total_rows;
    e->have_tables = true;
    e->have_cfgs = false;                       // configs reference table rows: re-validate
    return SHEMS_OK;
}

int shems_set_configs(shems_env *e, const shems_config *cfgs, int32_t n_cfg, const uint16_t *cfg_of_env)
{
    if (!e || !cfgs || n_cfg < 1 || n_cfg > 65536) return set_error(SHEMS_ERR_ARG, "shems_set_configs: bad arguments");
    if (!e->have_tables) return set_error(SHEMS_ERR_STATE, "shems_set_configs: call shems_set_tables first");
    for (int32_t c = 0; c < n_cfg; ++c) {
        const shems_config &k = cfgs[c];
        if (k.table_row0 < 0 || k.nrow < 2 || (int64_t)k.table_row0 + k.nrow > e->v.total_rows)
            return set_error(SHEMS_ERR_ARG, "shems_set_configs: config %d addresses rows [%d, %d) outside the %lld uploaded rows",
                             c, k.table_row0, k.table_row0 + k.nrow, (long long)e->v.total_rows);
        if (k.nrow <= e->maxsteps)
            return set_error(SHEMS_ERR_ARG, "shems_set_configs: config %d has nrow %d <= maxsteps %d (rand(1:(nrow-maxsteps)) is empty)",
                             c, k.nrow, e->maxsteps);
        if (!(k.cap_ev > 0.f) || !(k.soc_max > 0.f) || !(k.rate_max > 0.0))
            return set_error(SHEMS_ERR_ARG, "shems_set_configs: config %d has non-positive capacities", c);
    }
    if (cfg_of_env)
        for (int64_t i = 0; i < e->n; ++i)
            if (cfg_of_env[i] >= n_cfg) return set_error(SHEMS_ERR_ARG, "shems_set_configs: cfg_of_env[%lld] = %u >= n_cfg", (long long)i, cfg_of_env[i]);
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->v.cfgs) { hipFree((void *)e->v.cfgs); e->v.cfgs = nullptr; }
    shems_config *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, (size_t)n_cfg * sizeof(shems_config)));
    e->v.cfgs = d;
    HIP_TRY(hipMemcpyAsync(d, cfgs, (size_t)n_cfg * sizeof(shems_config), hipMemcpyHostToDevice, e->stream));
    if (cfg_of_env)
        HIP_TRY(hipMemcpyAsync((void *)e->v.cfg_of_env, cfg_of_env, (size_t)e->n * sizeof(uint16_t), hipMemcpyHostToDevice, e->stream));
    else
        HIP_TRY(hipMemsetAsync((void *)e->v.cfg_of_env, 0, (size_t)e->n * sizeof(uint16_t), e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->v.n_cfg = n_cfg;
    e->have_cfgs = true;
    return SHEMS_OK;
}

int shems_get_view(shems_env *e, shems_view *out)
{
    if (!e || !out) return set_error(SHEMS_ERR_ARG, "shems_get_view: NULL");
    if (int rc = need_ready(e, "shems_get_view")) return rc;
    *out = e->v;
    return SHEMS_OK;
}

int shems_check_error(shems_env *e)
{
    if (!e) return set_error(SHEMS_ERR_ARG, "shems_check_error: env is NULL");
    int32_t code = 0;
    HIP_TRY(hipMemcpyAsync(&code, e->v.err, sizeof code, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (code != 0) {
        HIP_TRY(hipMemsetAsync(e->v.err, 0, sizeof code, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (code == SHEMS_ERR_INDEX)
            return set_error(SHEMS_ERR_INDEX, "BoundsError: an env addressed a table row outside 1..nrow (idx+1 > nrow in step!, or a bad episode start)");
        return set_error(code, "device kernel reported error %d", code);
    }
    return SHEMS_OK;
}

int shems_reset(shems_env *e, int rng_minus1, const int32_t *idx0, const float *soc_b0)
{
    if (int rc = need_ready(e, "shems_reset")) return rc;
    HIP_TRY(hipSetDevice(e->device));
    if (!rng_minus1) {
        if (!idx0 || !soc_b0) return set_error(SHEMS_ERR_ARG, "shems_reset: idx0 and soc_b0 are required unless rng == -1");
        HIP_TRY(hipMemcpyAsync(e->d_idx0, idx0, (size_t)e->n * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(e->d_soc0, soc_b0, (size_t)e->n * sizeof(float), hipMemcpyHostToDevice, e->stream));
    }
    if (int rc = shems_reset_dev(&e->v, rng_minus1, e->d_idx0, e->d_soc0, e->stream)) return rc;
    return shems_check_error(e);
}

int shems_reset_seeded(shems_env *e, uint64_t seed, uint32_t episode)
{
    if (int rc = need_ready(e, "shems_reset_seeded")) return rc;
    HIP_TRY(hipSetDevice(e->device));
    if (int rc = shems_reset_seeded_dev(&e->v, seed, episode, e->stream)) return rc;
    return shems_check_error(e);
}

int shems_step(shems_env *e, const float *actions, int track_mode, double *rewards, float *obs, double *results)
{
    if (int rc = need_ready(e, "shems_step")) return rc;
    if (!actions) return set_error(SHEMS_ERR_ARG, "shems_step: actions is NULL");
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = (size_t)e->n;
    if (results && !e->d_res) HIP_TRY(hipMalloc((void **)&e->d_res, n * SHEMS_NRESULT * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(e->d_act, actions, n * 2 * sizeof(float), hipMemcpyHostToDevice, e->stream));
    if (int rc = shems_step_dev(&e->v, e->d_act, track_mode, e->d_rew, nullptr, results ? e->d_res : nullptr, nullptr, e->stream))
        return rc;
    if (rewards) HIP_TRY(hipMemcpyAsync(rewards, e->d_rew, n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    if (obs) HIP_TRY(hipMemcpyAsync(obs, e->v.obs, n * SHEMS_NSTATE * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (results) HIP_TRY(hipMemcpyAsync(results, e->d_res, n * SHEMS_NRESULT * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    return shems_check_error(e);                // synchronises
}

// inference(env; track != 0) for the handle's envs with HOST arrays (the N = 1 drop-in: what julia/ShemsEnv_LU1.jl binds): uploads the
// actor once, runs the whole pass as one launch (shems_track_dev), downloads env 0's results rows and every env's return.
int shems_track(shems_env *e, const float *actor, const float *s_min, const float *s_max, int track_mode, int32_t nsteps,
                double *results, double *returns)
{
    if (int rc = need_ready(e, "shems_track")) return rc;
    if (track_mode == 0 || nsteps <= 0) return set_error(SHEMS_ERR_ARG, "shems_track: track_mode must be non-zero and nsteps positive");
    if (track_mode > 0 && (!actor || !s_min || !s_max)) return set_error(SHEMS_ERR_ARG, "shems_track: actor / s_min / s_max required for track > 0");
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = (size_t)e->n;
    constexpr size_t kOffMin = (SHEMS_ACTOR_PARAMS + 3) / 4 * 4, kOffMax = kOffMin + 12, kParFloats = kOffMax + 12;
    shems_act_params p;
    std::memset(&p, 0, sizeof p);
    if (track_mode > 0) {
        if (!e->d_track_par) HIP_TRY(hipMalloc((void **)&e->d_track_par, kParFloats * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(e->d_track_par, actor, SHEMS_ACTOR_PARAMS * sizeof(float), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(e->d_track_par + kOffMin, s_min, SHEMS_NSTATE * sizeof(float), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(e->d_track_par + kOffMax, s_max, SHEMS_NSTATE * sizeof(float), hipMemcpyHostToDevice, e->stream));
        p.actor = e->d_track_par; p.s_min = e->d_track_par + kOffMin; p.s_max = e->d_track_par + kOffMax;
    }
    if (nsteps > e->track_res_steps) {
        if (e->d_track_res) HIP_TRY(hipFree(e->d_track_res));
        e->d_track_res = nullptr; e->track_res_steps = 0;
        HIP_TRY(hipMalloc((void **)&e->d_track_res, ((size_t)nsteps * SHEMS_NRESULT + n) * sizeof(double)));
        e->track_res_steps = nsteps;
    }
    double *d_ret = e->d_track_res + (size_t)e->track_res_steps * SHEMS_NRESULT;
    if (int rc = shems_track_dev(&e->v, track_mode > 0 ? &p : nullptr, 0, track_mode, nsteps, results ? e->d_track_res : nullptr, 0, d_ret, e->stream))
        return rc;
    if (results) HIP_TRY(hipMemcpyAsync(results, e->d_track_res, (size_t)nsteps * SHEMS_NRESULT * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    if (returns) HIP_TRY(hipMemcpyAsync(returns, d_ret, n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    return shems_check_error(e);                // synchronises; SHEMS_ERR_INDEX if the pass ran off the table
}

static int action_common(shems_env *e, const float *targets, int rule, float *out, const char *fn)
{
    if (int rc = need_ready(e, fn)) return rc;
    if (!out || (!rule && !targets)) return set_error(SHEMS_ERR_ARG, "%s: NULL buffer", fn);
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = (size_t)e->n;
    if (!rule) HIP_TRY(hipMemcpyAsync(e->d_act, targets, n * 2 * sizeof(float), hipMemcpyHostToDevice, e->stream));
    if (int rc = shems_action_dev(&e->v, e->d_act, rule, e->d_out2, e->stream)) return rc;
    HIP_TRY(hipMemcpyAsync(out, e->d_out2, n * 2 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    return hip_ok(hipStreamSynchronize(e->stream), "sync");
}

int shems_action(shems_env *e, const float *targets, float *out) { return action_common(e, targets, 0, out, "shems_action"); }
int shems_rule_action(shems_env *e, float *out) { return action_common(e, nullptr, 1, out, "shems_rule_action"); }

int shems_finished(shems_env *e, uint8_t *done)
{
    if (!e || !done) return set_error(SHEMS_ERR_ARG, "shems_finished: NULL");
    std::memset(done, 0, (size_t)e->n);         // LU1:487-502 returns false on both branches
    return SHEMS_OK;
}

int shems_get_state(shems_env *e, float *obs, int32_t *idx, int32_t *step)
{
    if (!e) return set_error(SHEMS_ERR_ARG, "shems_get_state: env is NULL");
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = (size_t)e->n;
    if (obs) HIP_TRY(hipMemcpyAsync(obs, e->v.obs, n * SHEMS_NSTATE * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (idx) HIP_TRY(hipMemcpyAsync(idx, e->v.idx, n * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    if (step) HIP_TRY(hipMemcpyAsync(step, e->v.step, n * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    return hip_ok(hipStreamSynchronize(e->stream), "sync");
}

int shems_set_state(shems_env *e, const float *obs, const int32_t *idx, const int32_t *step)
{
    if (!e) return set_error(SHEMS_ERR_ARG, "shems_set_state: env is NULL");
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = (size_t)e->n;
    if (obs) HIP_TRY(hipMemcpyAsync(e->v.obs, obs, n * SHEMS_NSTATE * sizeof(float), hipMemcpyHostToDevice, e->stream));
    if (idx) HIP_TRY(hipMemcpyAsync(e->v.idx, idx, n * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    if (step) HIP_TRY(hipMemcpyAsync(e->v.step, step, n * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    return hip_ok(hipStreamSynchronize(e->stream), "sync");
}

}  // extern "C"
