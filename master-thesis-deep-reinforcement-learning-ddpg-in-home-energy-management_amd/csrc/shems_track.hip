// shems_track.hip -- the tracking / inference pass as ONE launch: shems_track_dev.
//
// Replaces inference(env; track != 0) (memory_plotting_saving.jl:62-89), i.e. episode!(env; train = false, track, rng_ep = -1)
// (DDPG.jl:186-242) over a whole data set: EP_LENGTH["all", "eval"] = 1439 ... 2999 sequential hours of ONE household with the
// deterministic actor (track > 0) or the rule-based controller (track < 0), every hour leaving a 23-column results row
// (shems_LU1.jl:476-478).  The reference does this 80 times per job (40 seeds x {last, best} actor,
// DDPG_reinforce_charger_v1.jl:87-105), each step a batch-1 Flux forward + a CSV re-parse.
//
// A pass is inherently sequential (hour t + 1 starts from the state hour t leaves), so the unit of parallelism is the PASS:
// one workgroup of 512 threads per env of the view, each with its own actor (base + env * stride: a slab of actors as a
// learner group has it, or stride 0 = the same actor for every env), all hours inside the kernel -- no launch, no host
// synchronisation and no copy per hour; the results rows stream to HBM and are copied once by the caller.
//
// One hour of one env, 512 threads:
//   layer 1  threads 0..249: h1[k] = relu(b1[k] + sum_j x[j] W1[j][k]), the 9 + 1 weights of column k live in registers for the whole pass
//   layer 2  thread (q, p), q = t % 125 a quad of columns, p = t / 125 one of 4 row blocks of 64: 64 x (one 16-byte row segment
//            of W2 from L2 -- a 500 KB matrix does not fit a CU -- times h1[k] from LDS) in eight batches of 8 segments through two
//            register buffers; the first batch of the NEXT hour is requested before layer 3 and the serial env step of this hour
//            and waits in registers, so the hour starts with data on hand
//   layer 3  threads 0..499: relu(b2 + the 8 partial sums) . W3 (row in registers) -> wave sums -> 8 partials in LDS
//   env      thread 0: b3, tanh, clamp, scale_action, step!(track) with the exact mixed-precision arithmetic of shems_core.h, results row;
//            the table row idx + 1 was fetched into LDS by thread 32 while the layers ran
// The layers are tolerance-class arithmetic (<= 1e-5 of the float64 evaluation, as k_act; the summation order differs from k_act's,
// so the two agree to float rounding, not to the bit); step! is exact: given the targets a results row holds, the oracle reproduces
// the row bit for bit (tests/test_harness.py).
//
// Compiled with -ffp-contract=off (shems_core.h); the dense layers use explicit fmaf.
#include <hip/hip_runtime.h>

#include <cstring>

#include "shems_env_dev.h"
#include "shems_internal.h"

namespace shems {

constexpr int kTIn = 9, kTH1 = SHEMS_L1, kTH2 = SHEMS_L2, kTOut = 2;
constexpr int kTOffB1 = kTIn * kTH1, kTOffW2 = kTOffB1 + kTH1, kTOffB2 = kTOffW2 + kTH1 * kTH2, kTOffW3 = kTOffB2 + kTH2,
              kTOffB3 = kTOffW3 + kTH2 * kTOut;
static_assert(kTOffB3 + kTOut == SHEMS_ACTOR_PARAMS, "actor layout");
constexpr int kTThreads = 512, kTQuads = kTH2 / 4, kTParts = 4, kTRows = 64, kTBatch = 8;   // 512 threads: a 256-register budget (thread 0's step! needs it)
static_assert(kTQuads * kTParts <= kTThreads && kTParts * kTRows >= kTH1 && (kTParts - 1) * kTRows < kTH1 && kTH2 % 4 == 0, "layer-2 thread map");

typedef float t_f32x4 __attribute__((ext_vector_type(4)));

struct TrackArgs {
    shems_view v;
    const float *actor, *s_min, *s_max;      // env 0's; env e: + e * stride bytes
    int64_t stride;
    int track_mode;                          // SHEMS_TRACK_DRL (actor) or SHEMS_TRACK_RULE
    int nsteps;
    double *results;                         // [n or 1][nsteps][23] or null
    int64_t results_env;                     // -1: every env; else only this env's rows
    double *returns;                         // [n] or null
    int l1, l2;                              // k_track<true> (any hidden sizes): the actor is 9 -> l1 -> l2 -> 2
};

template <class T>
__device__ __forceinline__ const T *tsh(const T *p, int64_t off)
{
    return reinterpret_cast<const T *>(reinterpret_cast<const char *>(p) + off);
}

// WIDE = false: the tuned (250, 500) actor, as described above.  WIDE = true (shems_wide_track_dev): an actor of any hidden sizes
// (the reference grids' (300, 600) point) -- same pass, same env hour, the layers as plain loops over runtime sizes: layer 1 by
// threads k, k + 512, ..., layer 2 + 3 by column n = tid, tid + 512, ... (W2 read row by row from L2, coalesced over the columns).
template <bool WIDE>
__global__ __launch_bounds__(kTThreads) void k_track(TrackArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];                  // WIDE: relu(layer 1) [l1]
    __shared__ __attribute__((aligned(16))) float s_obs[12];
    __shared__ __attribute__((aligned(16))) float s_h1[kTParts * kTRows];          // 256 (rows >= 250 unused)
    __shared__ __attribute__((aligned(16))) float s_h1t[kTRows];                   // the last row block's view: 6 zeros, then h1[192..249]
    __shared__ __attribute__((aligned(16))) float s_part[kTParts][kTH2];
    __shared__ float s_red[kTThreads / 64][2];
    __shared__ __attribute__((aligned(16))) float s_row[12];                       // row idx + 1 (8 floats), h_countdown of row idx
    __shared__ int s_stop;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t e = blockIdx.x;
    const shems_view &v = A.v;
    const bool actor_mode = A.track_mode > 0;
    const float *P = tsh(A.actor, e * A.stride);
    const float *s_min = tsh(A.s_min, e * A.stride), *s_max = tsh(A.s_max, e * A.stride);

    // ---- once per pass: per-thread weights, the env's state -------------------------------------------------------------------
    const int k1 = min(tid, kTH1 - 1);                       // layer-1 column of this thread (threads >= 250: clamped, unused)
    __shared__ float s_lo[12], s_rng[12];
    float w1r[kTIn], b1r = 0.0f;
    const int n3 = min(tid, kTH2 - 1);                       // layer-3 row of this thread (threads >= 500: clamped, unused)
    float b2r = 0.0f, w30 = 0.0f, w31 = 0.0f;
    if (actor_mode) {
        if (tid < kTIn) { s_lo[tid] = s_min[tid]; s_rng[tid] = (s_max[tid] - s_min[tid]) + 1e-8f; }
        if constexpr (!WIDE) {
#pragma unroll
            for (int j = 0; j < kTIn; ++j) w1r[j] = P[j * kTH1 + k1];
            b1r = P[kTOffB1 + k1];
            b2r = P[kTOffB2 + n3];
            w30 = P[kTOffW3 + 2 * n3];
            w31 = P[kTOffW3 + 2 * n3 + 1];
        }
    }
    const int wl1 = A.l1, wl2 = A.l2;                         // WIDE: offsets of the flat Flux layout at the runtime sizes
    const int64_t wOffB1 = (int64_t)kTIn * wl1, wOffW2 = wOffB1 + wl1, wOffB2 = wOffW2 + (int64_t)wl1 * wl2, wOffW3 = wOffB2 + wl2,
                  wOffB3 = wOffW3 + (int64_t)wl2 * kTOut;
    const int q = tid % kTQuads, part = tid / kTQuads;       // layer 2: columns 4 q .. 4 q + 3, rows 64 part .. 64 part + 63
    const bool l2 = !WIDE && tid < kTQuads * kTParts;
    // row segment r of this thread: W2[base + r][4 q ..], base = 0, 64, 128 and 186 for the last block -- 64 rows from 192 would run
    // past the matrix; its first 6 rows (186..191) belong to block 2 and meet zeros in s_h1t, the shifted copy of h1 the block reads.
    const int rbase = min(kTRows * min(part, kTParts - 1), kTH1 - kTRows);
    const t_f32x4 *w2seg = reinterpret_cast<const t_f32x4 *>(P + kTOffW2) + (size_t)rbase * (kTH2 / 4) + q;
    auto seg = [&](int r) { return w2seg[(size_t)r * (kTH2 / 4)]; };
    shems_config cfg;
    float obs[SHEMS_NSTATE];
    int32_t idx = 0, step = 0;
    double total = 0.0;                                      // reward_eps (DDPG.jl:190, 223): Float64 after the first add
    if (tid == 0) {
        cfg = load_cfg(v, e);
#pragma unroll
        for (int k = 0; k < SHEMS_NSTATE; ++k) { obs[k] = v.obs[e * SHEMS_NSTATE + k]; s_obs[k] = obs[k]; }
        idx = v.idx[e];
        step = v.step[e];
        s_stop = 0;
    }
    if (tid < kTParts * kTRows) s_h1[tid] = 0.0f;
    if (tid < kTRows) s_h1t[tid] = 0.0f;
    int32_t row0 = 0, nrow = 0, pidx = 0;                    // thread 32 keeps its own copy of idx for the row prefetch
    if (tid == 32) {
        const shems_config c2 = load_cfg(v, e);
        row0 = c2.table_row0; nrow = c2.nrow; pidx = v.idx[e];
    }
    t_f32x4 wa[kTBatch];                                     // the first 8 row segments of the coming hour
    if (actor_mode && l2) {
#pragma unroll
        for (int r = 0; r < kTBatch; ++r) wa[r] = seg(r);
    }
    __syncthreads();

    for (int t = 0; t < A.nsteps; ++t) {
        // W2 is the same 500 KB every hour: left to itself the compiler hoists all 64 row segments of a thread out of this loop (256
        // registers -> scratch).  The clobber makes each hour re-read them from L2, eight segments at a time.
        // (and the pointer is made opaque per hour, or it keeps 64 precomputed addresses per thread across the loop instead)
        asm volatile("" : "+v"(w2seg) :: "memory");
        // table row idx + 1 and h_countdown of row idx for this hour's next_state!, under the layers
        if (tid == 32) {
            const int32_t r1 = max(min(pidx + 1, nrow), 2);  // clamped: env_advance_rows rejects idx + 1 > nrow before it looks
            const t_f32x4 *rp = reinterpret_cast<const t_f32x4 *>(v.tables + ((int64_t)row0 + r1 - 1) * SHEMS_NCOL);
            const t_f32x4 ra = rp[0], rb = rp[1];
            const float hc = v.tables[((int64_t)row0 + r1 - 2) * SHEMS_NCOL];
            *reinterpret_cast<t_f32x4 *>(s_row) = ra;
            *reinterpret_cast<t_f32x4 *>(s_row + 4) = rb;
            s_row[8] = hc;
            pidx += 1;
        }
        float p0 = 0.0f, p1 = 0.0f;
        if (WIDE && actor_mode) {
            if constexpr (WIDE) {
                float xr[kTIn];
#pragma unroll
                for (int j = 0; j < kTIn; ++j) xr[j] = (s_obs[j] - s_lo[j]) / s_rng[j];
                for (int k = tid; k < wl1; k += kTThreads) {
                    float z = P[wOffB1 + k];
#pragma unroll
                    for (int j = 0; j < kTIn; ++j) z = fmaf(xr[j], P[(int64_t)j * wl1 + k], z);
                    s_dyn[k] = fmaxf(z, 0.0f);
                }
                __syncthreads();
                float o0 = 0.0f, o1 = 0.0f;
                for (int n = tid; n < wl2; n += kTThreads) {
                    const float *w = P + wOffW2 + n;
                    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;
                    int k = 0;
                    for (; k + 4 <= wl1; k += 4) {
                        c0 = fmaf(s_dyn[k], w[(int64_t)k * wl2], c0);
                        c1 = fmaf(s_dyn[k + 1], w[(int64_t)(k + 1) * wl2], c1);
                        c2 = fmaf(s_dyn[k + 2], w[(int64_t)(k + 2) * wl2], c2);
                        c3 = fmaf(s_dyn[k + 3], w[(int64_t)(k + 3) * wl2], c3);
                    }
                    for (; k < wl1; ++k) c0 = fmaf(s_dyn[k], w[(int64_t)k * wl2], c0);
                    const float h2 = fmaxf(P[wOffB2 + n] + ((c0 + c1) + (c2 + c3)), 0.0f);
                    o0 = fmaf(h2, P[wOffW3 + 2 * (int64_t)n], o0);
                    o1 = fmaf(h2, P[wOffW3 + 2 * (int64_t)n + 1], o1);
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { o0 += __shfl_down(o0, off, 64); o1 += __shfl_down(o1, off, 64); }
                if (lane == 0) { s_red[wave][0] = o0; s_red[wave][1] = o1; }
                __syncthreads();
                if (tid == 0) {
                    p0 = P[wOffB3];
                    p1 = P[wOffB3 + 1];
#pragma unroll
                    for (int w = 0; w < kTThreads / 64; ++w) { p0 += s_red[w][0]; p1 += s_red[w][1]; }
                }
            }
        } else if (actor_mode) {
            // layer 1: x = normalize(s) (MPS:55-57), every thread for itself from the 9 observations in LDS
            if (tid < kTH1) {
                float z = b1r;
#pragma unroll
                for (int j = 0; j < kTIn; ++j) z = fmaf((s_obs[j] - s_lo[j]) / s_rng[j], w1r[j], z);
                const float hk = fmaxf(z, 0.0f);
                s_h1[tid] = hk;
                if (tid >= kTRows * (kTParts - 1)) s_h1t[tid - (kTH1 - kTRows)] = hk;       // rows 192..249 -> slots 6..63
            }
            __syncthreads();
            // layer 2: 32 row segments x 4 columns per thread
            if (l2) {
                t_f32x4 wb[kTBatch];
                t_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                const float *h = part == kTParts - 1 ? s_h1t : s_h1 + kTRows * part;
                // rows 8 b .. 8 b + 7 of the block: buffer w times the matching h1 values (same LDS address in every lane: broadcast)
                auto mac8 = [&](const t_f32x4 (&w)[kTBatch], int b) {
#pragma unroll
                    for (int r4 = 0; r4 < kTBatch / 4; ++r4) {
                        const t_f32x4 hv = *reinterpret_cast<const t_f32x4 *>(h + kTBatch * b + 4 * r4);
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[c] = fmaf(hv[u], w[4 * r4 + u][c], acc[c]);
                    }
                };
                // batches of 8 row segments alternate between the two register buffers; the load of batch b + 1 is issued before the
                // FMAs of batch b, and the last load is the NEXT hour's batch 0 (it flies during layer 3 and the env step)
#pragma unroll 1                                              // a real loop: unrolled, the scheduler lifts every load to the top (256 registers)
                for (int b = 0; b < kTRows / kTBatch; b += 2) {
#pragma unroll
                    for (int r = 0; r < kTBatch; ++r) wb[r] = seg((b + 1) * kTBatch + r);
                    mac8(wa, b);
                    if (b + 2 < kTRows / kTBatch || t + 1 < A.nsteps) {
#pragma unroll
                        for (int r = 0; r < kTBatch; ++r) wa[r] = seg(((b + 2) % (kTRows / kTBatch)) * kTBatch + r);
                    }
                    mac8(wb, b + 1);
                }
                *reinterpret_cast<t_f32x4 *>(&s_part[part][4 * q]) = acc;
            }
            __syncthreads();
            // layer 3: relu(b2 + partials) . W3, summed over the workgroup
            float o0 = 0.0f, o1 = 0.0f;
            if (tid < kTH2) {
                float z = b2r;
#pragma unroll
                for (int p = 0; p < kTParts; ++p) z += s_part[p][tid];
                const float h2 = fmaxf(z, 0.0f);
                o0 = h2 * w30;
                o1 = h2 * w31;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { o0 += __shfl_down(o0, off, 64); o1 += __shfl_down(o1, off, 64); }
            if (lane == 0) { s_red[wave][0] = o0; s_red[wave][1] = o1; }
            __syncthreads();
            if (tid == 0) {
                p0 = P[kTOffB3];
                p1 = P[kTOffB3 + 1];
#pragma unroll
                for (int w = 0; w < kTH2 / 64 + 1; ++w) { p0 += s_red[w][0]; p1 += s_red[w][1]; }
            }
        } else {
            __syncthreads();                                  // s_row visible
        }
        // ---- the env's hour, one thread: DDPG.jl:199-229 with train = false --------------------------------------------------
        if (tid == 0) {
            float a0, a1;
            int mode;
            if (actor_mode) {                                 // act(...; train = false): clamp(tanh-output, -1, 1), then scale_action
                a0 = scale_action(fminf(fmaxf(tanhf(p0), -1.0f), 1.0f));
                a1 = scale_action(fminf(fmaxf(tanhf(p1), -1.0f), 1.0f));
                mode = SHEMS_TRACK_DRL;
            } else {                                          // a = action(env, track)  (DDPG.jl:209-211)
                const EnvIn s{obs[0], obs[1], obs[2], obs[3], obs[4], obs[5]};
                action_rule(cfg, s, a0, a1);
                mode = SHEMS_TRACK_RULE;
            }
            float pre[SHEMS_NSTATE];
#pragma unroll
            for (int k = 0; k < SHEMS_NSTATE; ++k) pre[k] = obs[k];
            const Row nx{s_row[0], s_row[1], s_row[2], s_row[3], s_row[4], s_row[5], s_row[6], s_row[7]};
            double reward;
            StepFlows f;
            float B, EV, Bt, EVt;
            if (env_advance_rows(cfg, nx, s_row[8], obs, idx, step, a0, a1, mode, reward, f, B, EV, Bt, EVt)) {
                total += reward;
                if (A.results && (A.results_env < 0 || A.results_env == e)) {
                    double *r = A.results + ((A.results_env < 0 ? e : 0) * (int64_t)A.nsteps + t) * SHEMS_NRESULT;
                    write_results(r, idx, pre, EVt, EV, reward, f, B, Bt);
                }
#pragma unroll
                for (int k = 0; k < SHEMS_NSTATE; ++k) s_obs[k] = obs[k];
            } else {
                raise(v.err, SHEMS_ERR_INDEX);
                s_stop = 1;
            }
        }
        __syncthreads();
        if (s_stop) break;
    }
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < SHEMS_NSTATE; ++k) v.obs[e * SHEMS_NSTATE + k] = obs[k];
        v.idx[e] = idx;
        v.step[e] = step;
        if (A.returns) A.returns[e] = total;
    }
}

}  // namespace shems

using namespace shems;

extern "C" int shems_track_dev(const shems_view *v, const shems_act_params *p, int64_t actor_stride_bytes, int track_mode,
                               int32_t nsteps, double *d_results, int64_t results_env, double *d_returns, void *stream)
{
    if (int rc = check_view(v, "shems_track_dev")) return rc;
    if (track_mode == 0) return set_error(SHEMS_ERR_ARG, "shems_track_dev: track_mode must be > 0 (actor) or < 0 (rule-based)");
    if (nsteps <= 0) return set_error(SHEMS_ERR_ARG, "shems_track_dev: nsteps must be positive");
    if (results_env >= v->n_envs) return set_error(SHEMS_ERR_ARG, "shems_track_dev: results_env %lld outside the batch", (long long)results_env);
    TrackArgs a;
    std::memset(&a, 0, sizeof a);
    a.v = *v;
    if (track_mode > 0) {
        if (!p || !p->actor || !p->s_min || !p->s_max) return set_error(SHEMS_ERR_ARG, "shems_track_dev: actor / s_min / s_max required for track > 0");
        if (((uintptr_t)p->actor & 15) != 0 || (actor_stride_bytes & 15) != 0 || actor_stride_bytes < 0)
            return set_error(SHEMS_ERR_ARG, "shems_track_dev: actor block and stride must be 16-byte aligned");
        a.actor = p->actor; a.s_min = p->s_min; a.s_max = p->s_max; a.stride = actor_stride_bytes;
    }
    a.track_mode = track_mode > 0 ? SHEMS_TRACK_DRL : SHEMS_TRACK_RULE;
    a.nsteps = nsteps;
    a.results = d_results;
    a.results_env = results_env;
    a.returns = d_returns;
    hipLaunchKernelGGL(k_track<false>, dim3((unsigned)v->n_envs), dim3(kTThreads), 0, (hipStream_t)stream, a);
    return hip_ok(hipGetLastError(), "k_track launch");
}

/* shems_track_dev for an actor of any hidden sizes (9 -> l1 -> l2 -> 2 in the flat Flux layout of that size; 16-byte aligned block
 * and stride as there). */
extern "C" int shems_wide_track_dev(const shems_view *v, const shems_act_params *p, int32_t l1, int32_t l2, int64_t actor_stride_bytes,
                                    int32_t nsteps, double *d_results, int64_t results_env, double *d_returns, void *stream)
{
    if (int rc = check_view(v, "shems_wide_track_dev")) return rc;
    if (l1 < 1 || l2 < 1 || l1 > 4096 || l2 > 4096) return set_error(SHEMS_ERR_ARG, "shems_wide_track_dev: hidden sizes must be in 1..4096");
    if (nsteps <= 0) return set_error(SHEMS_ERR_ARG, "shems_wide_track_dev: nsteps must be positive");
    if (results_env >= v->n_envs) return set_error(SHEMS_ERR_ARG, "shems_wide_track_dev: results_env %lld outside the batch", (long long)results_env);
    if (!p || !p->actor || !p->s_min || !p->s_max) return set_error(SHEMS_ERR_ARG, "shems_wide_track_dev: actor / s_min / s_max required");
    if (((uintptr_t)p->actor & 15) != 0 || (actor_stride_bytes & 15) != 0 || actor_stride_bytes < 0)
        return set_error(SHEMS_ERR_ARG, "shems_wide_track_dev: actor block and stride must be 16-byte aligned");
    TrackArgs a;
    std::memset(&a, 0, sizeof a);
    a.v = *v;
    a.actor = p->actor; a.s_min = p->s_min; a.s_max = p->s_max; a.stride = actor_stride_bytes;
    a.track_mode = SHEMS_TRACK_DRL;
    a.nsteps = nsteps;
    a.results = d_results;
    a.results_env = results_env;
    a.returns = d_returns;
    a.l1 = l1; a.l2 = l2;
    hipLaunchKernelGGL(k_track<true>, dim3((unsigned)v->n_envs), dim3(kTThreads), sizeof(float) * (size_t)((l1 + 3) / 4 * 4), (hipStream_t)stream, a);
    return hip_ok(hipGetLastError(), "k_track (wide) launch");
}
