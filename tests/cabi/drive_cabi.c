/* Plain C program against include/shems_hip.h only (no Python, no PyTorch in the process): what a compiled-language caller
 * of the drop-in boundary does.  usage: drive_cabi <table.bin> <nrow> <n_envs> <nsteps> <out.bin>
 *   table.bin : [nrow][8] float32.  Runs reset!(env; rng=-1)-style seeded resets, `nsteps` x step!(env, s, a; track=1) with a
 *   closed-form action sequence, one rule-based step, and the error path (a table that is too short); writes
 *   idx0[n] i32, obs0[n][9] f32, then per step rewards[n] f64, obs[n][9] f32, results[n][23] f64; then rule (B,EV)[n][2] f32. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "shems_hip.h"

#define CHECK(call)                                                                         \
    do {                                                                                    \
        int rc_ = (call);                                                                   \
        if (rc_ != SHEMS_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, shems_last_error()); return 10; } \
    } while (0)

int main(int argc, char **argv)
{
    if (argc != 6) { fprintf(stderr, "usage\n"); return 2; }
    const long nrow = atol(argv[2]), n = atol(argv[3]), nsteps = atol(argv[4]);
    float *rows = (float *)malloc(sizeof(float) * 8 * nrow);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(rows, sizeof(float) * 8, nrow, f) != (size_t)nrow) { fprintf(stderr, "table read failed\n"); return 3; }
    fclose(f);
    if (shems_abi_version() != SHEMS_ABI_VERSION) return 4;
    int ndev = 0;
    CHECK(shems_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no device\n"); return 5; }

    shems_env *env = NULL;
    CHECK(shems_create(n, 72, 0, &env));
    int64_t nn = 0;
    CHECK(shems_n_envs(env, &nn));
    if (nn != n) return 6;
    /* call-order error: step before tables / configs / reset */
    float *act = (float *)malloc(sizeof(float) * 2 * n);
    memset(act, 0, sizeof(float) * 2 * n);
    if (shems_step(env, act, SHEMS_TRACK_OFF, NULL, NULL, NULL) != SHEMS_ERR_STATE) { fprintf(stderr, "expected SHEMS_ERR_STATE\n"); return 7; }
    CHECK(shems_set_tables(env, rows, nrow));
    shems_config cfg;                       /* Charger98: LU1:57 (35.816 kWh EV, 7.5*0.9 kWh battery, 3.3 kW), weights LU1:40-43 */
    memset(&cfg, 0, sizeof cfg);
    cfg.cap_ev = 35.816f; cfg.soc_max = 7.5f * 0.9f; cfg.rate_max = 3.3;
    cfg.disc_weight = (double)0.01f; cfg.disc_pot = (double)2.0f; cfg.penalty_weight = 0.1f;
    cfg.table_row0 = 0; cfg.nrow = (int32_t)nrow;
    CHECK(shems_set_configs(env, &cfg, 1, NULL));
    CHECK(shems_reset_seeded(env, 20240607ull, 3));

    FILE *o = fopen(argv[5], "wb");
    if (!o) return 8;
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * n);
    float *obs = (float *)malloc(sizeof(float) * 9 * n);
    double *rew = (double *)malloc(sizeof(double) * n), *res = (double *)malloc(sizeof(double) * 23 * n);
    CHECK(shems_get_state(env, obs, idx, NULL));
    fwrite(idx, sizeof(int32_t), n, o);
    fwrite(obs, sizeof(float) * 9, n, o);
    for (long t = 0; t < nsteps; ++t) {
        for (long i = 0; i < n; ++i) {      /* targets in [0, 1]: exactly representable arithmetic, reproduced by the test */
            act[2 * i] = (float)((i * 37 + t * 11) % 101) / 100.0f;
            act[2 * i + 1] = (float)((i * 53 + t * 29) % 97) / 96.0f;
        }
        CHECK(shems_step(env, act, SHEMS_TRACK_DRL, rew, obs, res));
        fwrite(rew, sizeof(double), n, o);
        fwrite(obs, sizeof(float) * 9, n, o);
        fwrite(res, sizeof(double) * 23, n, o);
    }
    float *rule = (float *)malloc(sizeof(float) * 2 * n);
    CHECK(shems_rule_action(env, rule));
    fwrite(rule, sizeof(float) * 2, n, o);
    uint8_t *done = (uint8_t *)malloc(n);
    CHECK(shems_finished(env, done));
    for (long i = 0; i < n; ++i) if (done[i]) return 9;
    fclose(o);

    /* BoundsError path: put env 0 on the last row; step! must refuse with SHEMS_ERR_INDEX */
    CHECK(shems_get_state(env, NULL, idx, NULL));
    idx[0] = (int32_t)nrow;
    CHECK(shems_set_state(env, NULL, idx, NULL));
    const int rc = shems_step(env, act, SHEMS_TRACK_OFF, rew, NULL, NULL);
    if (rc != SHEMS_ERR_INDEX) { fprintf(stderr, "expected SHEMS_ERR_INDEX, got %d\n", rc); return 11; }
    CHECK(shems_destroy(env));
    printf("drive_cabi ok: %ld envs x %ld steps\n", n, nsteps);
    return 0;
}
