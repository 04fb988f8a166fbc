/*
 * shems_policy_omp.c -- the fused vector step of the hot path on ALL host cores, for bench.py's cpu_baseline leg.
 *
 * TEST INFRASTRUCTURE / CPU BASELINE ONLY (see shems_oracle.h): nothing in the product path links or loads this.
 *
 * One OpenMP region per vector step; a thread takes blocks of 32 households through the whole body of the reference's
 * episode! loop (DDPG.jl:195-229): normalize (MPS:55-57) -> actor Chain(Dense(9,250,relu), Dense(250,500,relu),
 * Dense(500,2,tanh)) (DDPG.jl:30-36) -> + Normal(0, sigma) noise, clamp (DDPG.jl:159-160, 172) -> scale_action
 * (DDPG.jl:178-184) -> step! (shems_LU1.jl:343-485, via orc_step of shems_oracle.c).  The dense layers are plain C loops
 * (4 households x a vectorised row of outputs per inner iteration; W2 = 500 KB stays in a core's L2), compiled with
 * -O3 -march=native; the environment arithmetic stays in shems_oracle.c with its exact-IEEE flags.
 * It is a throughput baseline: the actions it produces are within float rounding of ddpg_oracle.act, not bit-identical
 * (different summation order), and tests/test_host_logic.py checks exactly that.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "shems_oracle.h"

enum { IN = 9, H1 = 250, H2 = 500, OUT = 2, BLK = 32 };
enum { OFF_B1 = IN * H1, OFF_W2 = OFF_B1 + H1, OFF_B2 = OFF_W2 + H1 * H2, OFF_W3 = OFF_B2 + H2, OFF_B3 = OFF_W3 + H2 * OUT };

static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

/* y[b][n] = relu?(sum_k x[b][k] * W[k][n] + bias[n]) for a block of nb <= BLK rows, W in the Flux layout [in][out].
 * Register-blocked: a 4-row x 32-column tile of y lives in 16 vector accumulators over the whole k loop (per k: 4 row loads of W,
 * 4 broadcasts, 16 FMAs); gcc vector extensions, unaligned loads through an aligned(4) may_alias type. */
typedef float v8 __attribute__((vector_size(32)));
typedef float v8u __attribute__((vector_size(32), aligned(4), may_alias));

static void dense_block(const float *restrict x, int ldx, int nb, const float *restrict W, const float *restrict bias, int K, int N,
                        float *restrict y, int ldy, int relu)
{
    const v8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    int n0 = 0;
    for (; n0 + 32 <= N; n0 += 32) {
        for (int b = 0; b < nb; b += 4) {
            const int r1 = b + 1 < nb ? b + 1 : b, r2 = b + 2 < nb ? b + 2 : b, r3 = b + 3 < nb ? b + 3 : b;   /* ragged tail: repeat row b */
            v8 acc[4][4];
            for (int v = 0; v < 4; ++v) { const v8 bv = *(const v8u *)(bias + n0 + 8 * v); acc[0][v] = bv; acc[1][v] = bv; acc[2][v] = bv; acc[3][v] = bv; }
            for (int k = 0; k < K; ++k) {
                const float *w = W + (size_t)k * N + n0;
                const v8 w0 = *(const v8u *)(w), w1 = *(const v8u *)(w + 8), w2 = *(const v8u *)(w + 16), w3 = *(const v8u *)(w + 24);
                const float a0 = x[b * ldx + k], a1 = x[r1 * ldx + k], a2 = x[r2 * ldx + k], a3 = x[r3 * ldx + k];
                acc[0][0] += a0 * w0; acc[0][1] += a0 * w1; acc[0][2] += a0 * w2; acc[0][3] += a0 * w3;
                acc[1][0] += a1 * w0; acc[1][1] += a1 * w1; acc[1][2] += a1 * w2; acc[1][3] += a1 * w3;
                acc[2][0] += a2 * w0; acc[2][1] += a2 * w1; acc[2][2] += a2 * w2; acc[2][3] += a2 * w3;
                acc[3][0] += a3 * w0; acc[3][1] += a3 * w1; acc[3][2] += a3 * w2; acc[3][3] += a3 * w3;
            }
            const int rows[4] = {b, r1, r2, r3};
            for (int r = 0; r < 4; ++r)
                for (int v = 0; v < 4; ++v) {
                    v8 t = acc[r][v];
                    if (relu) t = __builtin_ia32_maxps256(t, zero);
                    *(v8u *)(y + rows[r] * ldy + n0 + 8 * v) = t;
                }
        }
    }
    for (int b = 0; b < nb; ++b)                                  /* columns left over (N % 32) */
        for (int n = n0; n < N; ++n) {
            float t = bias[n];
            for (int k = 0; k < K; ++k) t += x[b * ldx + k] * W[(size_t)k * N + n];
            y[b * ldy + n] = relu ? fmaxf(t, 0.f) : t;
        }
}

/* One vector step for n households.  obs [n][9] in (the states the envs hold), a_out [n][2] = the clamped UNSCALED actions (what
 * remember() stores), rewards [n], obs_out [n][9] = s'.  Returns 0, or -1 if any step! ran off its table. */
int orc_policy_step_omp(orc_env *envs, int64_t n, const float *actor, const float *s_min, const float *s_max, float sigma,
                        uint64_t seed, uint32_t tick, int train, const float *obs, float *a_out, double *rewards, float *obs_out)
{
    int rc = 0;
    const int64_t nblk = (n + BLK - 1) / BLK;
#pragma omp parallel for schedule(static) reduction(|:rc)
    for (int64_t blk = 0; blk < nblk; ++blk) {
        const int64_t e0 = blk * BLK;
        const int nb = (int)((n - e0) < BLK ? (n - e0) : BLK);
        float x[BLK][IN + 1], h1[BLK][H1 + 2], h2[BLK][H2 + 4], o[BLK][OUT];
        for (int b = 0; b < nb; ++b)
            for (int k = 0; k < IN; ++k) x[b][k] = (obs[(e0 + b) * IN + k] - s_min[k]) / ((s_max[k] - s_min[k]) + 1e-8f);
        dense_block(&x[0][0], IN + 1, nb, actor, actor + OFF_B1, IN, H1, &h1[0][0], H1 + 2, 1);
        dense_block(&h1[0][0], H1 + 2, nb, actor + OFF_W2, actor + OFF_B2, H1, H2, &h2[0][0], H2 + 4, 1);
        dense_block(&h2[0][0], H2 + 4, nb, actor + OFF_W3, actor + OFF_B3, H2, OUT, &o[0][0], OUT, 0);
        for (int b = 0; b < nb; ++b) {
            const int64_t i = e0 + b;
            float p0 = tanhf(o[b][0]), p1 = tanhf(o[b][1]);
            if (train) {                                           /* Box-Muller on one Philox block per env */
                uint32_t c[4] = {(uint32_t)i, (uint32_t)((uint64_t)i >> 32), tick, 0x4E4F4953u};
                philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
                const float u1 = ((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(c[1] >> 8) * (1.0f / 16777216.0f);
                const float r = sqrtf(-2.0f * logf(u1)), ang = 6.28318530717958647692f * u2;
                p0 += sigma * (r * cosf(ang));
                p1 += sigma * (r * sinf(ang));
            }
            const float a[2] = {fminf(fmaxf(p0, -1.f), 1.f), fminf(fmaxf(p1, -1.f), 1.f)};
            a_out[2 * i] = a[0]; a_out[2 * i + 1] = a[1];
            const float sc[2] = {orc_scale_action(a[0]), orc_scale_action(a[1])};
            double rew;
            if (orc_step(&envs[i], sc, 0, &rew, 0) != 0) rc |= 1;
            rewards[i] = rew;
            memcpy(obs_out + IN * i, envs[i].state, sizeof(float) * IN);
        }
    }
    return rc ? -1 : 0;
}

/* Thread count of the OpenMP legs (bench.py passes the CPUs this process may actually use: affinity mask / cgroup quota --
 * the default, one thread per visible CPU, oversubscribes a container that sees 256 CPUs and owns 16). */
#include <omp.h>
void orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int  orc_get_max_threads(void) { return omp_get_max_threads(); }
