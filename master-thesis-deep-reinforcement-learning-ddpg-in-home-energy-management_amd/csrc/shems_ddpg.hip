// shems_ddpg.hip -- one DDPG update (the reference's replay(), DDPG.jl:121-145) as a short chain of
// gfx950 kernels: GPU-resident minibatch sampling/gather, target pass, critic forward/backward,
// actor forward/backward through the critic, Flux-style ADAM and the soft target updates.
//
// Shapes (BATCH = 120, padded to BP = 128 columns; pad columns carry zero error signals):
// everything is FEATURE-major "[k][m]" (sample index contiguous), as in shems_policy.hip, so the
// 250x500 layer runs on v_mfma_f32_32x32x2_f32 with the weights as the A operand straight out of
// Flux's [in][out] layout.  At batch 120 the update is launch/latency bound (307.8 MFLOP, ~2 us at
// the fp32 MFMA peak), so the design goal is few, wide launches: 32x32 output tiles, one per wave,
// operands read from L2 (the whole working set is < 6 MB), 15 launches per update:
//   prep(sample+gather+normalize+3x layer-1) -> L2fwd(actor_t) -> head(a', layer-1 critic_t)
//   -> L2fwd(critic_t, critic, actor) -> head(y, q, dq, dW3, D2) -> L2bwd(critic: dW2 || dH1)
//   -> L1bwd(critic)            [all-reduce]  -> adam+soft(critic)
//   -> head(a_pi, layer-1 critic) -> L2fwd(critic on [s; a_pi], emits D2) -> L2bwd(dH1)
//   -> head(da, d3, dW3, D2 actor) -> L2bwd(actor: dW2 || dH1) -> L1bwd(actor)
//                               [all-reduce]  -> adam+soft(actor)
#include <hip/hip_runtime.h>

#include <cstring>

#include "philox.h"
#include "shems_internal.h"

namespace shems {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BP = 128;            // padded batch (columns)
constexpr int H1N = SHEMS_L1, H2N = SHEMS_L2;
constexpr int H1P = 256, H2P = 512;
constexpr int SIN = 9, AIN = 2, CIN = 11;

// ---- parameter block offsets (Flux order W1 b1 W2 b2 W3 b3) -------------------------------------
__host__ __device__ constexpr int off_b1(int in) { return in * H1N; }
__host__ __device__ constexpr int off_w2(int in) { return in * H1N + H1N; }
__host__ __device__ constexpr int off_b2(int in) { return off_w2(in) + H1N * H2N; }
__host__ __device__ constexpr int off_w3(int in) { return off_b2(in) + H2N; }
__host__ __device__ constexpr int off_b3(int in, int out) { return off_w3(in) + H2N * out; }

// ---- workspace carve -----------------------------------------------------------------------------
constexpr int64_t WS_XT = 0;                         // [9][BP]  normalize(s)
constexpr int64_t WS_X2T = WS_XT + SIN * BP;          // [9][BP]  normalize(s')
constexpr int64_t WS_AT = WS_X2T + SIN * BP;          // [2][BP]  stored (unscaled) actions
constexpr int64_t WS_R = WS_AT + AIN * BP;            // [BP]
constexpr int64_t WS_DONE = WS_R + BP;
constexpr int64_t WS_Y = WS_DONE + BP;
constexpr int64_t WS_Q = WS_Y + BP;
constexpr int64_t WS_DQ = WS_Q + BP;
constexpr int64_t WS_API = WS_DQ + BP;                // [2][BP]  a_pi = actor(s)
constexpr int64_t WS_D3A = WS_API + AIN * BP;         // [2][BP]  dL/d(pre-tanh) of the actor head
constexpr int64_t WS_IDX = WS_D3A + AIN * BP;         // [BP]     sampled ring slots (as int32)
constexpr int64_t WS_SLOT0 = WS_IDX + BP;
constexpr int64_t SL_H1 = 0;                          // [250][BP]
constexpr int64_t SL_H1T = SL_H1 + H1N * BP;          // [BP][256]
constexpr int64_t SL_H2 = SL_H1T + BP * H1P;          // [500][BP]
constexpr int64_t SL_D2 = SL_H2 + H2N * BP;           // [500][BP]
constexpr int64_t SL_D2T = SL_D2 + H2N * BP;          // [BP][512]
constexpr int64_t SL_D1 = SL_D2T + BP * H2P;          // [250][BP]
constexpr int64_t SL_SIZE = SL_D1 + H1N * BP;
enum { SLOT_ACTOR_T = 0, SLOT_CRITIC_T = 1, SLOT_CRITIC = 2, SLOT_ACTOR = 3, SLOT_CRITIC2 = 4, N_SLOTS = 5 };
constexpr int64_t WS_FLOATS = WS_SLOT0 + N_SLOTS * SL_SIZE;

__host__ __device__ inline float *slot(float *ws, int s) { return ws + WS_SLOT0 + (int64_t)s * SL_SIZE; }

// ---- layer 1 for a k-range: H1[k][m] = relu(b1[k] + sum_j W1[j][k] x[j][m]) -----------------------
// x is an LDS image [in][BP].  Optionally also writes the transposed copy H1T[m][k].
__device__ __forceinline__ void layer1_range(const float *__restrict__ P, int in, const float *x, int k0, int k1,
                                             float *__restrict__ H1, float *__restrict__ H1T)
{
    const int nk = k1 - k0;
    for (int e = threadIdx.x; e < nk * BP; e += blockDim.x) {
        const int kl = e / BP, m = e - kl * BP, k = k0 + kl;
        float acc = P[off_b1(in) + k];
        for (int j = 0; j < in; ++j) acc = fmaf(P[j * H1N + k], x[j * BP + m], acc);
        const float h = fmaxf(acc, 0.0f);
        H1[k * BP + m] = h;
        if (H1T) H1T[m * H1P + k] = h;
    }
}

// ---- kernel A: sample + gather + normalize + layer 1 of actor_t(s'), critic([s;a]), actor(s) --------
__global__ __launch_bounds__(256) void k_prep(shems_ddpg d, shems_replay ring, int64_t ring_len, uint64_t seed,
                                              uint32_t tick)
{
    __shared__ float xs[CIN * BP];     // rows 0..8 normalize(s), rows 9..10 a
    __shared__ float x2[SIN * BP];     // normalize(s')
    const int t = threadIdx.x;
    float *ws = d.ws;
    if (t < BP) {
        const int m = t;
        float s[SIN], s2[SIN], a0 = 0.f, a1 = 0.f, r = 0.f, dn = 0.f;
        int64_t j = 0;
        const bool live = m < d.batch;
        if (live) {
            // StatsBase.sample(rng, memory, BATCH) -- with replacement (MPS:33)
            const u32x4 x = philox4x32_10((uint32_t)(m >> 2), 0u, tick, kStreamSample, (uint32_t)seed, (uint32_t)(seed >> 32));
            const uint32_t w = (m & 3) == 0 ? x.x : (m & 3) == 1 ? x.y : (m & 3) == 2 ? x.z : x.w;
            j = (int64_t)(w % (uint32_t)ring_len);
#pragma unroll
            for (int k = 0; k < SIN; ++k) { s[k] = ring.s[j * SIN + k]; s2[k] = ring.s2[j * SIN + k]; }
            a0 = ring.a[j * 2]; a1 = ring.a[j * 2 + 1];
            r = ring.r[j];
            dn = ring.done[j] ? 1.0f : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < SIN; ++k) {
            const float den = (d.s_max[k] - d.s_min[k]) + 1e-8f;                 // MPS:56
            xs[k * BP + m] = live ? (s[k] - d.s_min[k]) / den : 0.0f;
            x2[k * BP + m] = live ? (s2[k] - d.s_min[k]) / den : 0.0f;
        }
        xs[9 * BP + m] = a0;
        xs[10 * BP + m] = a1;
        if (blockIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < SIN; ++k) { ws[WS_XT + k * BP + m] = xs[k * BP + m]; ws[WS_X2T + k * BP + m] = x2[k * BP + m]; }
            ws[WS_AT + m] = a0; ws[WS_AT + BP + m] = a1;
            ws[WS_R + m] = r; ws[WS_DONE + m] = dn;
            reinterpret_cast<int32_t *>(ws + WS_IDX)[m] = live ? (int32_t)j : -1;
        }
    }
    __syncthreads();
    const int per = (H1N + gridDim.x - 1) / gridDim.x;
    const int k0 = blockIdx.x * per, k1 = min(H1N, k0 + per);
    if (k0 < k1) {
        layer1_range(d.actor_t, SIN, x2, k0, k1, slot(ws, SLOT_ACTOR_T) + SL_H1, nullptr);
        layer1_range(d.critic, CIN, xs, k0, k1, slot(ws, SLOT_CRITIC) + SL_H1, slot(ws, SLOT_CRITIC) + SL_H1T);
        layer1_range(d.actor, SIN, xs, k0, k1, slot(ws, SLOT_ACTOR) + SL_H1, slot(ws, SLOT_ACTOR) + SL_H1T);
    }
}

// ---- kernel B: layer 2 forward, one 32(n) x 32(m) tile per wave ------------------------------------
struct L2FwdJob {
    const float *P;        // parameter block of the network
    int in;                // its input width (9 or 11)
    const float *H1;       // [250][BP]
    float *H2;             // [500][BP]  relu(W2' h1 + b2)
    float *D2, *D2T;       // optional: D2[n][m] = d2_scale * W3[n][0] * (h2 > 0) for m < batch (critic inside the actor loss)
    float d2_scale;
};
struct L2FwdArgs { L2FwdJob job[3]; int batch; };

__global__ __launch_bounds__(256) void k_l2fwd(L2FwdArgs A)
{
    const L2FwdJob J = A.job[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * 32, m0 = wave * 32;
    const float *__restrict__ W2 = J.P + off_w2(J.in);
    const float *__restrict__ pa = W2 + n0 + li;            // A[i = n][k] = W2[k][n0 + i]  (n >= 500 reads b2: discarded)
    const float *__restrict__ pb = J.H1 + m0 + li;          // B[k][j = m] = H1[k][m0 + j]
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll 5
    for (int ks = 0; ks < H1N / 2; ++ks) {
        const int k = 2 * ks + lh;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k * H2N], pb[k * BP], acc, 0, 0, 0);
    }
    const float *b2 = J.P + off_b2(J.in), *W3 = J.P + off_w3(J.in);
    const int m = m0 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < H2N) {
            const float h = fmaxf(acc[r] + b2[n], 0.0f);
            J.H2[n * BP + m] = h;
            if (J.D2) {
                const float dv = (h > 0.0f && m < A.batch) ? J.d2_scale * W3[n] : 0.0f;
                J.D2[n * BP + m] = dv;
                J.D2T[m * H2P + n] = dv;
            }
        }
    }
}

// ---- kernel D: layer 2 backward tiles --------------------------------------------------------------
//   W tiles (8 x 16): gW2[k][n] = sum_m H1T[m][k] * D2T[m][n]                       (K = BP)
//   I tiles (8 x 4) : D1[k][m]  = (sum_n W2[k][n] * D2[n][m]) * (H1[k][m] > 0)       (K = 500)
struct L2BwdArgs {
    const float *H1T, *D2T; float *gW2;        // W part (gW2 = null: skip)
    const float *W2, *D2, *H1; float *D1;      // I part
    int n_w_tiles;                             // 128 or 0
};

__global__ __launch_bounds__(256) void k_l2bwd(L2BwdArgs A)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const int tile = blockIdx.x * 4 + wave;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    if (tile < A.n_w_tiles) {
        const int k0 = (tile >> 4) * 32, n0 = (tile & 15) * 32;
        const float *__restrict__ pa = A.H1T + k0 + li;     // A[i = k][kk = m] = H1T[m][k0 + i]
        const float *__restrict__ pb = A.D2T + n0 + li;     // B[kk = m][j = n] = D2T[m][n0 + j]
#pragma unroll 8
        for (int ms = 0; ms < BP / 2; ++ms) {
            const int m = 2 * ms + lh;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[m * H1P], pb[m * H2P], acc, 0, 0, 0);
        }
        const int n = n0 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (k < H1N && n < H2N) A.gW2[k * H2N + n] = acc[r];
        }
    } else {
        const int t = tile - A.n_w_tiles;
        if (t >= 32) return;
        const int k0 = (t >> 2) * 32, m0 = (t & 3) * 32;
        const int krow = min(k0 + li, H1N - 1);              // rows >= 250 do not exist: clamp, discard later
        const float *__restrict__ pa = A.W2 + (int64_t)krow * H2N;   // A[i = k][kk = n] = W2[k0 + i][n]
        const float *__restrict__ pb = A.D2 + m0 + li;               // B[kk = n][j = m] = D2[n][m0 + j]
#pragma unroll 8
        for (int ns = 0; ns < H2N / 2; ++ns) {
            const int n = 2 * ns + lh;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[n], pb[n * BP], acc, 0, 0, 0);
        }
        const int m = m0 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (k < H1N) A.D1[k * BP + m] = A.H1[k * BP + m] > 0.0f ? acc[r] : 0.0f;
        }
    }
}

// ---- kernel C: the heads (layer 3 and everything that hangs off it), one 1024-thread workgroup ------
enum { HEAD_TARGET_ACTOR = 0, HEAD_ACTOR = 1, HEAD_CRITIC_LOSS = 2, HEAD_ACTOR_BWD = 3 };

// out[o][m] = sum_n W3[n][o] * H2[n][m] for o < OUT, all 128 m.  8 n-groups x 128 m threads, LDS reduce.
template <int OUT>
__device__ __forceinline__ void layer3_all(const float *__restrict__ W3, const float *__restrict__ H2, float *red /*[8][OUT][BP]*/,
                                           float *out /*[OUT][BP] in LDS*/)
{
    const int g = threadIdx.x >> 7, m = threadIdx.x & 127;
    float acc[OUT];
#pragma unroll
    for (int o = 0; o < OUT; ++o) acc[o] = 0.0f;
    const int na = g * 63, nb = min(H2N, na + 63);
    for (int n = na; n < nb; ++n) {
        const float h = H2[n * BP + m];
#pragma unroll
        for (int o = 0; o < OUT; ++o) acc[o] = fmaf(W3[n * OUT + o], h, acc[o]);
    }
#pragma unroll
    for (int o = 0; o < OUT; ++o) red[(g * OUT + o) * BP + m] = acc[o];
    __syncthreads();
    if (threadIdx.x < OUT * BP) {
        const int o = threadIdx.x / BP, mm = threadIdx.x - o * BP;
        float s = 0.0f;
#pragma unroll
        for (int gg = 0; gg < 8; ++gg) s += red[(gg * OUT + o) * BP + mm];
        out[o * BP + mm] = s;
    }
    __syncthreads();
}

__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
}

// gW3[n][o] = sum_m H2[n][m] * d3[o][m] (wave per n), D2[n][m] = (sum_o W3[n][o] d3[o][m]) * (H2 > 0), D2T.
template <int OUT>
__device__ __forceinline__ void head_backward(const float *__restrict__ W3, const float *__restrict__ H2, const float *d3 /*LDS [OUT][BP]*/,
                                              float *__restrict__ gW3, float *__restrict__ D2, float *__restrict__ D2T)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int n = wave; n < H2N; n += 16) {
        const float h0 = H2[n * BP + lane], h1 = H2[n * BP + 64 + lane];
#pragma unroll
        for (int o = 0; o < OUT; ++o) {
            const float s = wave_sum(h0 * d3[o * BP + lane] + h1 * d3[o * BP + 64 + lane]);
            if (lane == 0) gW3[n * OUT + o] = s;
        }
    }
    for (int e = threadIdx.x; e < H2N * BP; e += blockDim.x) {
        const int n = e / BP, m = e - n * BP;
        float v = 0.0f;
#pragma unroll
        for (int o = 0; o < OUT; ++o) v = fmaf(W3[n * OUT + o], d3[o * BP + m], v);
        v = H2[e] > 0.0f ? v : 0.0f;
        D2[e] = v;
        D2T[m * H2P + n] = v;
    }
}

__global__ __launch_bounds__(1024) void k_head(shems_ddpg d, int mode)
{
    __shared__ float red[8 * 2 * BP];
    __shared__ float xa[CIN * BP];
    __shared__ float o3[2 * BP];
    __shared__ float d3[2 * BP];
    __shared__ float sred[16];
    float *ws = d.ws;
    const int t = threadIdx.x;
    const float invB = 1.0f / (float)d.batch;

    if (mode == HEAD_TARGET_ACTOR || mode == HEAD_ACTOR) {
        const bool tgt = mode == HEAD_TARGET_ACTOR;
        const float *P = tgt ? d.actor_t : d.actor;
        const float *H2 = slot(ws, tgt ? SLOT_ACTOR_T : SLOT_ACTOR) + SL_H2;
        layer3_all<2>(P + off_w3(SIN), H2, red, o3);
        const float *X = ws + (tgt ? WS_X2T : WS_XT);
        for (int e = t; e < SIN * BP; e += blockDim.x) xa[e] = X[e];
        if (t < 2 * BP) {
            const int o = t / BP;
            const float a = tanhf(o3[t] + P[off_b3(SIN, 2) + o]);                 // Dense(500, 2, tanh)
            xa[SIN * BP + t] = a;
            if (!tgt) ws[WS_API + t] = a;
        }
        __syncthreads();
        const float *PC = tgt ? d.critic_t : d.critic;
        float *S = slot(ws, tgt ? SLOT_CRITIC_T : SLOT_CRITIC2);
        layer1_range(PC, CIN, xa, 0, H1N, S + SL_H1, nullptr);                    // vcat(s_norm, a) -> Dense(11, 250, relu)
    } else if (mode == HEAD_CRITIC_LOSS) {
        layer3_all<1>(d.critic_t + off_w3(CIN), slot(ws, SLOT_CRITIC_T) + SL_H2, red, o3);        // q'
        layer3_all<1>(d.critic + off_w3(CIN), slot(ws, SLOT_CRITIC) + SL_H2, red, o3 + BP);       // q
        float sq = 0.0f;
        if (t < BP) {
            const float q2 = o3[t] + d.critic_t[off_b3(CIN, 1)];
            const float q = o3[BP + t] + d.critic[off_b3(CIN, 1)];
            const float y = ws[WS_R + t] + d.gamma * (1.0f - ws[WS_DONE + t]) * q2;               // DDPG.jl:133
            const bool live = t < d.batch;
            const float diff = live ? q - y : 0.0f;
            ws[WS_Y + t] = y; ws[WS_Q + t] = q;
            const float dq = 2.0f * diff * invB;                                                  // d mse / d q
            ws[WS_DQ + t] = dq;
            d3[t] = dq;
            sq = diff * diff;
        }
        // loss = mean((q - y)^2), gb3 = sum(dq)
        float s1 = wave_sum(sq), s2 = wave_sum(t < BP ? d3[t < BP ? t : 0] : 0.0f);
        if (t < BP && (t & 63) == 0) { sred[t >> 6] = s1; sred[4 + (t >> 6)] = s2; }
        __syncthreads();
        if (t == 0) {
            d.losses[0] = (sred[0] + sred[1]) * invB;
            d.grad_critic[off_b3(CIN, 1)] = sred[4] + sred[5];
        }
        float *S = slot(ws, SLOT_CRITIC);
        head_backward<1>(d.critic + off_w3(CIN), S + SL_H2, d3, d.grad_critic + off_w3(CIN), S + SL_D2, S + SL_D2T);
    } else {   // HEAD_ACTOR_BWD
        // da[o][m] = sum_k W1c[9 + o][k] * D1c[k][m]   (critic input gradient, action rows only)
        {
            const int g = t >> 7, m = t & 127;
            const float *D1 = slot(ws, SLOT_CRITIC2) + SL_D1;
            const float *W1 = d.critic;
            float a0 = 0.0f, a1 = 0.0f;
            const int ka = g * 32, kb = min(H1N, ka + 32);
            for (int k = ka; k < kb; ++k) {
                const float v = D1[k * BP + m];
                a0 = fmaf(W1[9 * H1N + k], v, a0);
                a1 = fmaf(W1[10 * H1N + k], v, a1);
            }
            red[(g * 2 + 0) * BP + m] = a0;
            red[(g * 2 + 1) * BP + m] = a1;
        }
        // q of the critic inside the actor loss (for the reported loss only)
        __syncthreads();
        if (t < 2 * BP) {
            float s = 0.0f;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += red[(g * 2 + t / BP) * BP + (t & 127)];
            const float a = ws[WS_API + t];
            d3[t] = s * (1.0f - a * a);                      // through tanh
            ws[WS_D3A + t] = d3[t];
        }
        __syncthreads();
        layer3_all<1>(d.critic + off_w3(CIN), slot(ws, SLOT_CRITIC2) + SL_H2, red, o3);
        float qs = (t < d.batch) ? o3[t] + d.critic[off_b3(CIN, 1)] : 0.0f;
        float g0 = t < BP ? d3[t] : 0.0f, g1 = t < BP ? d3[BP + t] : 0.0f;
        qs = wave_sum(qs); g0 = wave_sum(g0); g1 = wave_sum(g1);
        if (t < BP && (t & 63) == 0) { sred[t >> 6] = qs; sred[4 + (t >> 6)] = g0; sred[8 + (t >> 6)] = g1; }
        __syncthreads();
        if (t == 0) {
            d.losses[1] = -(sred[0] + sred[1]) * invB;       // loss_act = -mean(critic(vcat(s, actor(s))))
            d.grad_actor[off_b3(SIN, 2) + 0] = sred[4] + sred[5];
            d.grad_actor[off_b3(SIN, 2) + 1] = sred[8] + sred[9];
        }
        float *S = slot(ws, SLOT_ACTOR);
        head_backward<2>(d.actor + off_w3(SIN), S + SL_H2, d3, d.grad_actor + off_w3(SIN), S + SL_D2, S + SL_D2T);
    }
}

// ---- kernel E: layer-1 / bias gradients ---------------------------------------------------------------
//   gW1[j][k] = sum_m x[j][m] D1[k][m],  gb1[k] = sum_m D1[k][m],  gb2[n] = sum_m D2[n][m]
__global__ __launch_bounds__(1024) void k_l1bwd(const float *__restrict__ X /*[9][BP]*/, const float *__restrict__ XA /*[2][BP] or null*/,
                                                int in, const float *__restrict__ D1, const float *__restrict__ D2,
                                                float *__restrict__ grad)
{
    __shared__ float xs[CIN * BP];
    for (int e = threadIdx.x; e < SIN * BP; e += blockDim.x) xs[e] = X[e];
    if (XA) for (int e = threadIdx.x; e < AIN * BP; e += blockDim.x) xs[SIN * BP + e] = XA[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // one wave per k (coalesced D1 rows), 16 waves
    for (int k = wave; k < H1N; k += 16) {
        const float v0 = D1[k * BP + lane], v1 = D1[k * BP + 64 + lane];
        const float sb = wave_sum(v0 + v1);
        if (lane == 0) grad[off_b1(in) + k] = sb;
        for (int j = 0; j < in; ++j) {
            const float s = wave_sum(xs[j * BP + lane] * v0 + xs[j * BP + 64 + lane] * v1);
            if (lane == 0) grad[j * H1N + k] = s;
        }
    }
    for (int n = wave; n < H2N; n += 16) {
        const float s = wave_sum(D2[n * BP + lane] + D2[n * BP + 64 + lane]);
        if (lane == 0) grad[off_b2(in) + n] = s;
    }
}

// ---- kernel F: Flux 0.12.1 ADAM + soft target update ------------------------------------------------------
//   mt = b1*mt + (1-b1)*g ; vt = b2*vt + (1-b2)*g^2 ; delta = mt/(1-bp1) / (sqrt(vt/(1-bp2)) + eps) * eta ; p -= delta
//   (Float64 scalars broadcast over Float32 arrays: each element is computed in f64 and stored as f32)
//   then target = (1f0 - tau) * target + tau * p   (DDPG.jl:99-103)
__global__ __launch_bounds__(256) void k_adam_soft(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ mt,
                                                   float *__restrict__ vt, float *__restrict__ target, int n, double eta,
                                                   double bp1, double bp2, double gscale, float tau)
{
    const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float gf = (float)((double)g[i] * gscale);            // averaged gradient, as the replicas would hold it
        const float m1 = (float)(b1 * (double)mt[i] + (1.0 - b1) * (double)gf);
        const float v1 = (float)(b2 * (double)vt[i] + (1.0 - b2) * ((double)gf * (double)gf));
        const float delta = (float)((double)m1 / (1.0 - bp1) / (sqrt((double)v1 / (1.0 - bp2)) + eps) * eta);
        const float pn = p[i] - delta;
        mt[i] = m1; vt[i] = v1; p[i] = pn;
        const float one_m_tau = 1.0f - tau;
        target[i] = one_m_tau * target[i] + tau * pn;
    }
}

// ---- min_max_buffer (MPS:50-53) -----------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_minmax(shems_replay ring, int64_t ring_len, int64_t count, uint64_t seed,
                                                 float *s_min, float *s_max)
{
    __shared__ float lmin[16 * SIN], lmax[16 * SIN];
    float mn[SIN], mx[SIN];
#pragma unroll
    for (int k = 0; k < SIN; ++k) { mn[k] = INFINITY; mx[k] = -INFINITY; }
    for (int64_t q = threadIdx.x; q < (count + 3) / 4; q += blockDim.x) {
        const u32x4 x = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), 0xFFFFFFFFu, kStreamSample, (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (q * 4 + c < count) {
                const int64_t j = (int64_t)(w[c] % (uint32_t)ring_len);
#pragma unroll
                for (int k = 0; k < SIN; ++k) { const float v = ring.s[j * SIN + k]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < SIN; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = fminf(mn[k], __shfl_down(mn[k], off, 64));
            mx[k] = fmaxf(mx[k], __shfl_down(mx[k], off, 64));
        }
        if (lane == 0) { lmin[wave * SIN + k] = mn[k]; lmax[wave * SIN + k] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < SIN) {
        float a = INFINITY, b = -INFINITY;
        for (int w = 0; w < 16; ++w) { a = fminf(a, lmin[w * SIN + threadIdx.x]); b = fmaxf(b, lmax[w * SIN + threadIdx.x]); }
        s_min[threadIdx.x] = a; s_max[threadIdx.x] = b;
    }
}

}  // namespace shems

using namespace shems;

static int check_ddpg(const shems_ddpg *d, const char *fn)
{
    if (!d || !d->actor || !d->critic || !d->actor_t || !d->critic_t || !d->m_actor || !d->v_actor || !d->m_critic ||
        !d->v_critic || !d->grad_actor || !d->grad_critic || !d->s_min || !d->s_max || !d->ws || !d->losses)
        return set_error(SHEMS_ERR_ARG, "%s: shems_ddpg has a NULL buffer", fn);
    if (d->batch < 1 || d->batch > BP) return set_error(SHEMS_ERR_ARG, "%s: batch must be in 1..128 (got %d)", fn, d->batch);
    return SHEMS_OK;
}

extern "C" {

int shems_ddpg_workspace_floats(int64_t *out)
{
    if (!out) return set_error(SHEMS_ERR_ARG, "shems_ddpg_workspace_floats: NULL");
    *out = WS_FLOATS;
    return SHEMS_OK;
}

int shems_ddpg_sample_indices(uint64_t seed, uint32_t tick, int32_t batch, int64_t ring_len, int64_t *out)
{
    if (!out || batch < 1 || ring_len < 1 || ring_len > 0xFFFFFFFFll) return set_error(SHEMS_ERR_ARG, "shems_ddpg_sample_indices: bad arguments");
    for (int m = 0; m < batch; ++m) {
        const u32x4 x = philox4x32_10((uint32_t)(m >> 2), 0u, tick, kStreamSample, (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint32_t w = (m & 3) == 0 ? x.x : (m & 3) == 1 ? x.y : (m & 3) == 2 ? x.z : x.w;
        out[m] = (int64_t)(w % (uint32_t)ring_len);
    }
    return SHEMS_OK;
}

int shems_ddpg_critic_grad(const shems_ddpg *d, const shems_replay *ring, int64_t ring_len, uint64_t seed, uint32_t tick,
                           void *stream)
{
    if (int rc = check_ddpg(d, "shems_ddpg_critic_grad")) return rc;
    if (!ring || !ring->s || !ring->a || !ring->r || !ring->s2 || !ring->done || ring_len < 1 || ring_len > ring->capacity)
        return set_error(SHEMS_ERR_ARG, "shems_ddpg_critic_grad: bad replay ring / length");
    hipStream_t st = (hipStream_t)stream;
    float *ws = d->ws;
    hipLaunchKernelGGL(k_prep, dim3(10), dim3(256), 0, st, *d, *ring, ring_len, seed, tick);
    L2FwdArgs f;
    std::memset(&f, 0, sizeof f);
    f.batch = d->batch;
    f.job[0] = L2FwdJob{d->actor_t, SIN, slot(ws, SLOT_ACTOR_T) + SL_H1, slot(ws, SLOT_ACTOR_T) + SL_H2, nullptr, nullptr, 0.f};
    hipLaunchKernelGGL(k_l2fwd, dim3(16, 1), dim3(256), 0, st, f);
    hipLaunchKernelGGL(k_head, dim3(1), dim3(1024), 0, st, *d, (int)HEAD_TARGET_ACTOR);
    f.job[0] = L2FwdJob{d->critic_t, CIN, slot(ws, SLOT_CRITIC_T) + SL_H1, slot(ws, SLOT_CRITIC_T) + SL_H2, nullptr, nullptr, 0.f};
    f.job[1] = L2FwdJob{d->critic, CIN, slot(ws, SLOT_CRITIC) + SL_H1, slot(ws, SLOT_CRITIC) + SL_H2, nullptr, nullptr, 0.f};
    f.job[2] = L2FwdJob{d->actor, SIN, slot(ws, SLOT_ACTOR) + SL_H1, slot(ws, SLOT_ACTOR) + SL_H2, nullptr, nullptr, 0.f};
    hipLaunchKernelGGL(k_l2fwd, dim3(16, 3), dim3(256), 0, st, f);
    hipLaunchKernelGGL(k_head, dim3(1), dim3(1024), 0, st, *d, (int)HEAD_CRITIC_LOSS);
    float *S = slot(ws, SLOT_CRITIC);
    L2BwdArgs b{S + SL_H1T, S + SL_D2T, d->grad_critic + off_w2(CIN), d->critic + off_w2(CIN), S + SL_D2, S + SL_H1, S + SL_D1, 128};
    hipLaunchKernelGGL(k_l2bwd, dim3(40), dim3(256), 0, st, b);
    hipLaunchKernelGGL(k_l1bwd, dim3(1), dim3(1024), 0, st, (const float *)(ws + WS_XT), (const float *)(ws + WS_AT), (int)CIN,
                       (const float *)(S + SL_D1), (const float *)(S + SL_D2), d->grad_critic);
    return hip_ok(hipGetLastError(), "ddpg critic_grad launches");
}

static int adam_launch(float *p, const float *g, float *m, float *v, float *target, int n, double eta, double bp1, double bp2,
                       double gscale, float tau, hipStream_t st)
{
    if (!(bp1 > 0.0 && bp1 < 1.0 && bp2 > 0.0 && bp2 < 1.0)) return set_error(SHEMS_ERR_ARG, "adam: beta powers must be in (0,1)");
    hipLaunchKernelGGL(k_adam_soft, dim3(128), dim3(256), 0, st, p, g, m, v, target, n, eta, bp1, bp2, gscale, tau);
    return hip_ok(hipGetLastError(), "k_adam_soft launch");
}

int shems_ddpg_critic_apply(const shems_ddpg *d, double eta, double bp1, double bp2, double grad_scale, void *stream)
{
    if (int rc = check_ddpg(d, "shems_ddpg_critic_apply")) return rc;
    return adam_launch(d->critic, d->grad_critic, d->m_critic, d->v_critic, d->critic_t, SHEMS_CRITIC_PARAMS, eta, bp1, bp2,
                       grad_scale, d->tau, (hipStream_t)stream);
}

int shems_ddpg_actor_grad(const shems_ddpg *d, void *stream)
{
    if (int rc = check_ddpg(d, "shems_ddpg_actor_grad")) return rc;
    hipStream_t st = (hipStream_t)stream;
    float *ws = d->ws;
    hipLaunchKernelGGL(k_head, dim3(1), dim3(1024), 0, st, *d, (int)HEAD_ACTOR);
    float *C2 = slot(ws, SLOT_CRITIC2);
    L2FwdArgs f;
    std::memset(&f, 0, sizeof f);
    f.batch = d->batch;
    f.job[0] = L2FwdJob{d->critic, CIN, C2 + SL_H1, C2 + SL_H2, C2 + SL_D2, C2 + SL_D2T, -1.0f / (float)d->batch};
    hipLaunchKernelGGL(k_l2fwd, dim3(16, 1), dim3(256), 0, st, f);
    L2BwdArgs bi{nullptr, nullptr, nullptr, d->critic + off_w2(CIN), C2 + SL_D2, C2 + SL_H1, C2 + SL_D1, 0};
    hipLaunchKernelGGL(k_l2bwd, dim3(8), dim3(256), 0, st, bi);
    hipLaunchKernelGGL(k_head, dim3(1), dim3(1024), 0, st, *d, (int)HEAD_ACTOR_BWD);
    float *S = slot(ws, SLOT_ACTOR);
    L2BwdArgs b{S + SL_H1T, S + SL_D2T, d->grad_actor + off_w2(SIN), d->actor + off_w2(SIN), S + SL_D2, S + SL_H1, S + SL_D1, 128};
    hipLaunchKernelGGL(k_l2bwd, dim3(40), dim3(256), 0, st, b);
    hipLaunchKernelGGL(k_l1bwd, dim3(1), dim3(1024), 0, st, (const float *)(ws + WS_XT), (const float *)nullptr, (int)SIN,
                       (const float *)(S + SL_D1), (const float *)(S + SL_D2), d->grad_actor);
    return hip_ok(hipGetLastError(), "ddpg actor_grad launches");
}

int shems_ddpg_actor_apply(const shems_ddpg *d, double eta, double bp1, double bp2, double grad_scale, void *stream)
{
    if (int rc = check_ddpg(d, "shems_ddpg_actor_apply")) return rc;
    return adam_launch(d->actor, d->grad_actor, d->m_actor, d->v_actor, d->actor_t, SHEMS_ACTOR_PARAMS, eta, bp1, bp2, grad_scale,
                       d->tau, (hipStream_t)stream);
}

int shems_minmax_dev(const shems_replay *ring, int64_t ring_len, int64_t count, uint64_t seed, float *d_s_min, float *d_s_max,
                     void *stream)
{
    if (!ring || !ring->s || ring_len < 1 || ring_len > ring->capacity || count < 1 || !d_s_min || !d_s_max)
        return set_error(SHEMS_ERR_ARG, "shems_minmax_dev: bad arguments");
    hipLaunchKernelGGL(k_minmax, dim3(1), dim3(1024), 0, (hipStream_t)stream, *ring, ring_len, count, seed, d_s_min, d_s_max);
    return hip_ok(hipGetLastError(), "k_minmax launch");
}

}  // extern "C"
