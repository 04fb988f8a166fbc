// shems_policy.hip -- actor forward (9 -> 250 -> 500 -> 2) on fp32 MFMA, fused with observation
// normalisation, Gaussian exploration noise, clamp, scale_action, step! and the replay insert.
//
// Replaces the per-step body of the reference's episode! (DDPG.jl:195-234):
//     s = copy(env.state); a, noise = act(normalize(s |> gpu)); scaled = scale_action(a)
//     r, s' = step!(env, s, scaled); remember(s, a, r, s', finished(env, s'))
// which there costs one H2D copy, ~8 tiny kernels at batch 1, one D2H sync and a CSV parse per step.
//
// Decomposition (gfx950, wave64, 4 waves per workgroup, one workgroup per CU):
//   * a workgroup owns BM = 32*TM envs.  Everything is kept FEATURE-major ("[k][m]", env index
//     contiguous) so that layer outputs come out of the MFMA in exactly the layout the next layer
//     consumes: D'[n][m] = sum_k W[k][n] * H[k][m] with the weights as the MFMA A operand
//     (A[i = n][k] = W[k][n0+i]: Flux's column-major out x in matrix IS [k][n]) and the activations as
//     the B operand (B[k][j = m]).  v_mfma_f32_32x32x2_f32: lane l holds A[l&31][l>>5], B[l>>5][l&31].
//   * layer 2 (97.5 % of the FLOPs): wave w accumulates the 128(n) x BM(m) slab n in [128w, 128w+128)
//     = 4 x TM tiles of 32x32 (16*4*TM accumulator registers) over K = 250 = 125 MFMA k-steps.
//     Both operands are streamed through LDS in chunks of KC = 10 k-rows, double-buffered:
//     W2 rows come from L2 (global -> registers -> LDS, loads issued one chunk ahead), and the layer-1
//     activations relu(W1 x + b1) of the chunk are (re)computed on the VALU from the 9 normalised
//     inputs -- K = 9 is too thin for the matrix pipe and recomputing is cheaper than holding the
//     250 x BM activation tile (125 KB at BM = 128) in LDS.
//   * epilogue: bias + relu on the accumulators, layer 3 (500 -> 2) as per-lane partial dot products
//     reduced across lane halves (DPP) and the 4 waves (LDS), + b3, tanh, noise, clamp.
//   * one thread per env then runs scale_action + step! (shems_core.h, exact reference arithmetic)
//     and pushes the transition into the HBM replay ring.
// LDS: 2 x 20 KB (W2 chunks) + 2 x KC*BM*4 (H chunks) + x (9*BM*4) + W1/b1 10 KB + b2/W3/b3 6 KB.
#include <hip/hip_runtime.h>

#include <cstring>

#include "shems_env_dev.h"
#include "shems_internal.h"

namespace shems {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kIn = 9, kH1 = SHEMS_L1, kH2 = SHEMS_L2, kOut = 2;
constexpr int kKC = 10;                         // k-rows per staged chunk (5 MFMA k-steps)
constexpr int kChunks = kH1 / kKC;              // 25
constexpr int kWcFloats = kKC * kH2 + 16;       // + pad: the last n-tile reads 12 floats past a row
constexpr int kOffB1 = kIn * kH1, kOffW2 = kOffB1 + kH1, kOffB2 = kOffW2 + kH1 * kH2, kOffW3 = kOffB2 + kH2,
              kOffB3 = kOffW3 + kH2 * kOut;
static_assert(kOffB3 + kOut == SHEMS_ACTOR_PARAMS, "actor layout");
constexpr int kH2P = 512;                       // n padded to 16 MFMA tiles; pad rows carry zero bias / W3
constexpr int kTailFloats = kH2P + kH2P * kOut + kOut;   // LDS image: b2[512], W3[512][2], b3[2]

struct ActArgs {
    shems_view v;              // only used when do_step
    shems_act_params p;
    const float *obs;          // [m][9]
    int64_t m;
    float *a_out;              // [m][2] or null
    double *rewards;
    float *rewards_f32;
    double *block_reward;
    double *returns_acc;
    shems_replay ring;
    shems_ring_window win;
    int do_step;
    int use_ring;
};

template <int TM>
constexpr size_t act_lds_bytes()
{
    return sizeof(float) * (2 * kWcFloats + 2 * kKC * 32 * TM + kIn * 32 * TM + (kIn * kH1 + kH1) + (kTailFloats + 2) +
                            4 * 32 * TM * kOut);
}

__device__ __forceinline__ void glds16(const void *g, void *lds)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}

// Standard-normal pair from one Philox block (Box-Muller, f32).
__device__ __forceinline__ float2 gauss_pair(uint64_t seed, uint32_t tick, int64_t i)
{
    const u32x4 x = philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), tick, kStreamNoise, (uint32_t)seed,
                                  (uint32_t)(seed >> 32));
    const float u1 = ((float)(x.x >> 8) + 1.0f) * (1.0f / 16777216.0f);      // (0, 1]
    const float u2 = u01_24(x.y);                                             // [0, 1)
    const float r = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincosf(6.28318530717958647692f * u2, &s, &c);
    return make_float2(r * c, r * s);
}

template <int TM>
__global__ __launch_bounds__(256, 1) void k_act(ActArgs A)
{
    constexpr int BM = 32 * TM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Wc = reinterpret_cast<float *>(smem);             // [2][kWcFloats]
    float *Hc = Wc + 2 * kWcFloats;                          // [2][kKC][BM]
    float *xT = Hc + 2 * kKC * BM;                           // [9][BM]   normalised obs, feature-major
    float *w1 = xT + kIn * BM;                               // W1 [9][250] then b1 [250]
    float *tl = w1 + (kIn * kH1 + kH1);                      // b2 [500], W3 [500][2], b3 [2]
    float *red = tl + (kTailFloats + 2);                     // [4 waves][BM][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int64_t env0 = (int64_t)blockIdx.x * BM;
    const float *__restrict__ P = A.p.actor;

    // ---- stage 0: x = normalize(s) -> xT[k][m]; W1/b1 and b2/W3/b3 -> LDS; W2 chunk 0 -> LDS -------
    for (int e = tid; e < BM * kIn; e += 256) {
        const int m = e / kIn, k = e - m * kIn;
        const int64_t g = env0 * kIn + e;
        float x = 0.0f;
        if (g < A.m * kIn) {
            const float s = A.obs[g];
            x = (s - A.p.s_min[k]) / ((A.p.s_max[k] - A.p.s_min[k]) + 1e-8f);      // MPS:56
        }
        xT[k * BM + m] = x;
    }
    for (int e = tid; e < (kIn * kH1 + kH1) / 4; e += 256)
        reinterpret_cast<float4 *>(w1)[e] = reinterpret_cast<const float4 *>(P)[e];
    for (int e = tid; e < kH2P; e += 256) tl[e] = e < kH2 ? P[kOffB2 + e] : 0.0f;
    for (int e = tid; e < kH2P * kOut; e += 256) tl[kH2P + e] = e < kH2 * kOut ? P[kOffW3 + e] : 0.0f;
    if (tid < kOut) tl[kH2P + kH2P * kOut + tid] = P[kOffB3 + tid];
    if (tid < 16) { Wc[kKC * kH2 + tid] = 0.0f; Wc[kWcFloats + kKC * kH2 + tid] = 0.0f; }

    // W2 chunk staging, global -> LDS directly (global_load_lds_dwordx4: no staging registers).  A chunk is
    // kKC*500*4 = 20000 contiguous bytes = 19 full 1-KiB wave pieces + one of 544 B (34 lanes); wave w
    // issues pieces w, w+4, ...  The LDS image is the linear copy (destination = M0 base + lane*16).
    const char *W2g = reinterpret_cast<const char *>(P + kOffW2);
    constexpr int kChunkBytes = kKC * kH2 * 4;
    constexpr int kPieces = (kChunkBytes + 1023) / 1024;            // 20
    constexpr int kTailLanes = (kChunkBytes - (kPieces - 1) * 1024) / 16;   // 34
#define W2_ISSUE(chunk, buf)                                                                      \
    do {                                                                                          \
        const char *src_ = W2g + (size_t)(chunk) * kChunkBytes;                                   \
        char *dst_ = reinterpret_cast<char *>(Wc + (buf) * kWcFloats);                            \
        _Pragma("unroll") for (int pc_ = wave; pc_ < kPieces; pc_ += 4) {                         \
            if (pc_ < kPieces - 1 || lane < kTailLanes)                                           \
                glds16(src_ + pc_ * 1024 + lane * 16, dst_ + pc_ * 1024);                         \
        }                                                                                         \
    } while (0)
    // layer 1 for the k-rows of one chunk: h1[k][m] = relu(b1[k] + sum_j W1[j][k] x[j][m])
    auto h1_chunk = [&](int chunk, int buf) {
        float *dst = Hc + buf * (kKC * BM);
#pragma unroll 1
        for (int e = tid; e < kKC * BM; e += 256) {
            const int kl = e / BM, m = e - kl * BM;
            const int k = chunk * kKC + kl;
            float acc = w1[kIn * kH1 + k];
#pragma unroll
            for (int j = 0; j < kIn; ++j) acc = fmaf(w1[j * kH1 + k], xT[j * BM + m], acc);
            dst[e] = fmaxf(acc, 0.0f);
        }
    };

    W2_ISSUE(0, 0);
    __syncthreads();                     // xT, w1 visible
    h1_chunk(0, 0);
    __syncthreads();

    // ---- layer 2: 125 k-steps of 4 x TM MFMA tiles per wave ---------------------------------------
    f32x16 acc[4][TM];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int nbase = wave * 128;
    for (int c = 0; c < kChunks; ++c) {
        const int cur = c & 1, nxt = cur ^ 1;
        if (c + 1 < kChunks) {
            W2_ISSUE(c + 1, nxt);        // LDS-DMA in flight under the MFMAs below; __syncthreads drains it
            h1_chunk(c + 1, nxt);
        }
        const float *Wb = Wc + cur * kWcFloats + nbase + li;
        const float *Hb = Hc + cur * (kKC * BM) + li;
#pragma unroll
        for (int ks = 0; ks < kKC / 2; ++ks) {
            const int krow = 2 * ks + lh;
            float af[4], bf[TM];
#pragma unroll
            for (int a = 0; a < 4; ++a) af[a] = Wb[krow * kH2 + 32 * a];
#pragma unroll
            for (int b = 0; b < TM; ++b) bf[b] = Hb[krow * BM + 32 * b];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: relu(acc + b2), layer 3 partial dot products ------------------------------------
    float o0[TM], o1[TM];
#pragma unroll
    for (int b = 0; b < TM; ++b) { o0[b] = 0.0f; o1[b] = 0.0f; }
    const float *b2s = tl, *w3s = tl + kH2P;
    // The accumulators live in AGPRs; read them out one row at a time with v_accvgpr_read (asm volatile keeps
    // program order, which bounds VGPR pressure -- left to itself the compiler copies all 64*TM*4 values first).
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // MFMA D -> v_accvgpr_read hazard (nothing pads asm)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            __builtin_amdgcn_sched_barrier(0);  // ... and keep the bias / W3 LDS reads with their row
            const int n = nbase + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lh;     // C/D row of v_mfma_f32_32x32x*
            const bool valid = n < kH2;
            const float bias = b2s[n];
            const float2 w3 = *reinterpret_cast<const float2 *>(w3s + 2 * n);
            const float wa = w3.x, wb = w3.y;
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                float x;
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[a][b][r]));
                float h = fmaxf(x + bias, 0.0f);
                h = valid ? h : 0.0f;                                           // rows >= 500 hold garbage
                o0[b] = fmaf(h, wa, o0[b]);
                o1[b] = fmaf(h, wb, o1[b]);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        o0[b] += __shfl_xor(o0[b], 32, 64);
        o1[b] += __shfl_xor(o1[b], 32, 64);
        if (lh == 0) {
            red[(wave * BM + 32 * b + li) * 2 + 0] = o0[b];
            red[(wave * BM + 32 * b + li) * 2 + 1] = o1[b];
        }
    }
    __syncthreads();

    // ---- one thread per env: tanh, noise, clamp, scale_action, step!, remember ---------------------
    double reward = 0.0;
    const int64_t i = env0 + tid;
    if (tid < BM && i < A.m) {
        float p0 = tl[kH2P + kH2P * kOut + 0], p1 = tl[kH2P + kH2P * kOut + 1];  // b3
#pragma unroll
        for (int w = 0; w < 4; ++w) { p0 += red[(w * BM + tid) * 2 + 0]; p1 += red[(w * BM + tid) * 2 + 1]; }
        p0 = tanhf(p0);
        p1 = tanhf(p1);
        if (A.p.train) {                                   // DDPG.jl:160, 172: act_pred .+ noise
            const float2 z = gauss_pair(A.p.seed, A.p.tick, i);
            p0 += A.p.noise_mu + A.p.noise_sigma * z.x;
            p1 += A.p.noise_mu + A.p.noise_sigma * z.y;
        }
        const float a0 = fminf(fmaxf(p0, -1.0f), 1.0f);    // clamp.(., -1f0, 1f0)
        const float a1 = fminf(fmaxf(p1, -1.0f), 1.0f);
        if (A.a_out) reinterpret_cast<float2 *>(A.a_out)[i] = make_float2(a0, a1);
        if (A.do_step) {
            const shems_view &v = A.v;
            const shems_config c = load_cfg(v, i);
            float obs[SHEMS_NSTATE], pre[SHEMS_NSTATE];
#pragma unroll
            for (int k = 0; k < SHEMS_NSTATE; ++k) { obs[k] = v.obs[i * SHEMS_NSTATE + k]; pre[k] = obs[k]; }
            int32_t idx = v.idx[i], step = v.step[i];
            StepFlows f;
            float B, EV, Bt, EVt;
            if (env_advance(c, v.tables, obs, idx, step, scale_action(a0), scale_action(a1), SHEMS_TRACK_OFF, reward, f,
                            B, EV, Bt, EVt)) {
#pragma unroll
                for (int k = 0; k < SHEMS_NSTATE; ++k) v.obs[i * SHEMS_NSTATE + k] = obs[k];
                v.idx[i] = idx;
                v.step[i] = step;
                if (A.rewards) A.rewards[i] = reward;
                if (A.rewards_f32) A.rewards_f32[i] = (float)reward;
                if (A.returns_acc) A.returns_acc[i] += reward;
                if (A.use_ring) {
                    int64_t rel = i - A.win.offset;
                    rel %= v.n_envs;
                    if (rel < 0) rel += v.n_envs;
                    if (rel < A.win.count)
                        ring_push(A.ring, (A.win.pos + rel) % A.ring.capacity, pre, a0, a1, (float)reward, obs);
                }
            } else {
                reward = 0.0;
                raise(v.err, SHEMS_ERR_INDEX);
            }
        }
    }
    if (A.block_reward) {
        __syncthreads();
        double *red64 = reinterpret_cast<double *>(Wc);     // Wc is dead by now
        const double s = block_sum(reward, red64);
        if (tid == 0) A.block_reward[blockIdx.x] = s;
    }
}

// BM: as large as keeps >= 2 workgroups per CU's worth of tiles (256 CUs); smaller batches use smaller tiles.
static int pick_tm(int64_t m)
{
    if (m >= 256 * 128) return 4;
    if (m >= 256 * 64) return 2;
    return 1;
}

template <int TM>
static int launch_act(const ActArgs &a, hipStream_t st)
{
    constexpr int BM = 32 * TM;
    const size_t lds = act_lds_bytes<TM>();
    static bool attr_done = false;
    if (!attr_done) {
        if (int rc = hip_ok(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_act<TM>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                            "hipFuncSetAttribute(k_act)"))
            return rc;
        attr_done = true;
    }
    const unsigned grid = (unsigned)((a.m + BM - 1) / BM);
    hipLaunchKernelGGL(k_act<TM>, dim3(grid), dim3(256), lds, st, a);
    return hip_ok(hipGetLastError(), "k_act launch");
}

static int dispatch_act(const ActArgs &a, hipStream_t st)
{
    switch (pick_tm(a.m)) {
    case 4: return launch_act<4>(a, st);
    case 2: return launch_act<2>(a, st);
    default: return launch_act<1>(a, st);
    }
}

}  // namespace shems

using namespace shems;

extern "C" {

int shems_act_step_grid(int64_t n_envs, int64_t *out_blocks)
{
    if (n_envs <= 0 || !out_blocks) return set_error(SHEMS_ERR_ARG, "shems_act_step_grid: bad arguments");
    const int bm = 32 * pick_tm(n_envs);
    *out_blocks = (n_envs + bm - 1) / bm;
    return SHEMS_OK;
}

static int check_act(const shems_act_params *p, const char *fn)
{
    if (!p || !p->actor || !p->s_min || !p->s_max) return set_error(SHEMS_ERR_ARG, "%s: actor / s_min / s_max required", fn);
    if (((uintptr_t)p->actor & 15) != 0) return set_error(SHEMS_ERR_ARG, "%s: actor parameter block must be 16-byte aligned", fn);
    return SHEMS_OK;
}

int shems_actor_forward_dev(const shems_act_params *p, const float *d_obs, int64_t m, float *d_a, void *stream)
{
    if (int rc = check_act(p, "shems_actor_forward_dev")) return rc;
    if (!d_obs || !d_a || m <= 0) return set_error(SHEMS_ERR_ARG, "shems_actor_forward_dev: bad buffers");
    ActArgs a;
    std::memset(&a, 0, sizeof a);
    a.p = *p; a.obs = d_obs; a.m = m; a.a_out = d_a;
    return dispatch_act(a, (hipStream_t)stream);
}

int shems_act_step_dev(const shems_view *v, const shems_act_params *p, float *d_a, double *d_rewards,
                       float *d_rewards_f32, double *d_block_reward, double *d_returns_acc,
                       const shems_replay *ring, const shems_ring_window *window, void *stream)
{
    if (int rc = check_act(p, "shems_act_step_dev")) return rc;
    if (!v || v->n_envs <= 0 || !v->obs || !v->idx || !v->step || !v->cfgs || !v->tables || v->n_cfg < 1)
        return set_error(SHEMS_ERR_ARG, "shems_act_step_dev: invalid view");
    ActArgs a;
    std::memset(&a, 0, sizeof a);
    a.v = *v; a.p = *p; a.obs = v->obs; a.m = v->n_envs; a.a_out = d_a;
    a.rewards = d_rewards; a.rewards_f32 = d_rewards_f32; a.block_reward = d_block_reward;
    a.returns_acc = d_returns_acc;
    a.do_step = 1;
    if (ring && window && window->count > 0) {
        if (ring->capacity <= 0 || !ring->s || !ring->a || !ring->r || !ring->s2 || !ring->done)
            return set_error(SHEMS_ERR_ARG, "shems_act_step_dev: incomplete replay ring");
        if (window->count > ring->capacity || window->count > v->n_envs || window->pos < 0)
            return set_error(SHEMS_ERR_ARG, "shems_act_step_dev: ring window larger than the ring or the batch");
        a.ring = *ring; a.win = *window; a.use_ring = 1;
    }
    return dispatch_act(a, (hipStream_t)stream);
}

}  // extern "C"
