// shems_ddpg.hip -- one DDPG update (the reference's replay(), DDPG.jl:121-145) as a short chain of
// gfx950 kernels: GPU-resident minibatch sampling/gather, target pass, critic forward/backward,
// actor forward/backward through the critic, Flux-style ADAM and the soft target updates.
//
// Shapes: BATCH = 120 padded to BP = 128 columns (pad columns carry zero error signals); everything
// is FEATURE-major "[k][m]" (sample index contiguous) as in shems_policy.hip, so the 250x500 layer
// runs on v_mfma_f32_32x32x2_f32 with the weights as the A operand straight out of Flux's [in][out]
// layout.  At batch 120 one update is 307.8 MFLOP (~2 us at the fp32 MFMA peak): it is bound by
// dependent-launch boundaries (~1.5 us each) and by L2/Infinity-Cache latency, not by the matrix
// pipe.  Design rules that follow from the first measured version (profiles/r01_train_v1_*):
//   * no single-workgroup latency chains: every phase is spread over 16-64 workgroups;
//   * operands are staged into LDS with wide, independent loads (one latency per phase), never
//     fetched per MFMA k-step;
//   * nothing derivable is stored: layer-1 activations, their relu masks and the back-propagated
//     layer-2 error D2 = (W3 d3) .* (h2 > 0) are recomputed inside the kernels that consume them;
//   * cross-workgroup reductions go through partial slabs summed in a fixed order (bitwise
//     reproducible; no float atomics).
// 13 launches per update:
//   prep -> fwd(actor_t) -> fwd(critic_t | critic | actor) -> head(loss) -> bwd(critic) -> l1bwd(critic)
//   [all-reduce] adam+soft(critic)
//   fwd(critic on [s; actor(s)]) -> bwd(input grad) -> head(actor) -> bwd(actor) -> l1bwd(actor)
//   [all-reduce] adam+soft(actor)
#include <hip/hip_runtime.h>

#include <cstring>

#include "philox.h"
#include "shems_internal.h"

namespace shems {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BP = 128;            // padded batch (columns)
constexpr int H1N = SHEMS_L1, H2N = SHEMS_L2;
constexpr int SIN = 9, AIN = 2, CIN = 11;
constexpr int NT = 16;             // n-tiles of 32 over the 500 (512) layer-2 outputs
constexpr int KT = 8;              // k-tiles of 32 over the 250 (256) layer-1 outputs
constexpr int NQ = 4;              // quarters of the n range for the input-gradient tiles
constexpr int NQW = H2N / NQ;      // 125 n per quarter

__host__ __device__ constexpr int off_b1(int in) { return in * H1N; }
__host__ __device__ constexpr int off_w2(int in) { return in * H1N + H1N; }
__host__ __device__ constexpr int off_b2(int in) { return off_w2(in) + H1N * H2N; }
__host__ __device__ constexpr int off_w3(int in) { return off_b2(in) + H2N; }
__host__ __device__ constexpr int off_b3(int in, int out) { return off_w3(in) + H2N * out; }

// ---- workspace carve (floats) --------------------------------------------------------------------
constexpr int64_t WS_XT = 0;                         // [9][BP]  normalize(s)
constexpr int64_t WS_X2T = WS_XT + SIN * BP;          // [9][BP]  normalize(s')
constexpr int64_t WS_AT = WS_X2T + SIN * BP;          // [2][BP]  stored (unscaled) actions
constexpr int64_t WS_R = WS_AT + AIN * BP;            // [BP]
constexpr int64_t WS_DONE = WS_R + BP;
constexpr int64_t WS_Y = WS_DONE + BP;
constexpr int64_t WS_Q = WS_Y + BP;
constexpr int64_t WS_API = WS_Q + BP;                 // [2][BP]  a_pi = actor(s)
constexpr int64_t WS_D3C = WS_API + AIN * BP;         // [1][BP]  dq of the critic loss
constexpr int64_t WS_D3Q = WS_D3C + BP;               // [1][BP]  -1/B (actor loss through the critic)
constexpr int64_t WS_D3A = WS_D3Q + BP;               // [2][BP]  error at the actor's pre-tanh output
constexpr int64_t WS_IDX = WS_D3A + AIN * BP;         // [BP]     sampled ring slots (int32)
constexpr int64_t WS_DAP = WS_IDX + BP;               // [KT][NQ][2][BP]  partial d loss / d a_pi
constexpr int64_t WS_SLOT0 = WS_DAP + KT * NQ * AIN * BP;
constexpr int64_t SL_H2 = 0;                          // [500][BP]          relu(W2' h1 + b2)
constexpr int64_t SL_P3 = SL_H2 + H2N * BP;           // [NT][2][BP]        per-n-tile partial sums of layer 3
constexpr int64_t SL_D1P = SL_P3 + NT * 2 * BP;       // [NQ][250][BP]      partial (unmasked) error at layer 1
constexpr int64_t SL_SIZE = SL_D1P + NQ * H1N * BP;
enum { SLOT_ACTOR_T = 0, SLOT_CRITIC_T = 1, SLOT_CRITIC = 2, SLOT_ACTOR = 3, SLOT_CRITIC2 = 4, N_SLOTS = 5 };
constexpr int64_t WS_FLOATS = WS_SLOT0 + N_SLOTS * SL_SIZE;

__host__ __device__ inline float *slot(float *ws, int s) { return ws + WS_SLOT0 + (int64_t)s * SL_SIZE; }

__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
}

// Where a network input [in][BP] comes from: rows 0..8 = a normalised-state block, rows 9..10 (critics) either the
// stored actions or tanh(b3 + sum of the layer-3 partials of an actor pass).
struct XSrc {
    const float *X;        // [9][BP]
    const float *A;        // [2][BP] stored actions, or null
    const float *P3;       // [NT][2][BP] actor partials, or null
    const float *b3;       // actor b3 (with P3)
    float *publish;        // optional [2][BP]: where workgroup 0 stores the computed action
};

template <int IN>
__device__ __forceinline__ void build_x(const XSrc &s, float *xs /*LDS [IN][BP]*/, bool publisher)
{
    for (int e = threadIdx.x; e < SIN * BP; e += blockDim.x) xs[e] = s.X[e];
    if (IN == CIN) {
        for (int e = threadIdx.x; e < AIN * BP; e += blockDim.x) {
            float a;
            if (s.A) {
                a = s.A[e];
            } else {
                const int o = e / BP, m = e - o * BP;
                float acc = s.b3[o];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc += s.P3[(t * 2 + o) * BP + m];
                a = tanhf(acc);                                   // Dense(500, 2, tanh)
                if (publisher && s.publish) s.publish[e] = a;
            }
            xs[SIN * BP + e] = a;
        }
    }
}

// h1[k][m] = relu(b1[k] + sum_j W1[j][k] x[j][m]) with W1/b1 in LDS (w1: [IN][250] then b1[250]).
template <int IN>
__device__ __forceinline__ float h1_at(const float *w1, const float *xs, int k, int m)
{
    float acc = w1[IN * H1N + k];
#pragma unroll
    for (int j = 0; j < IN; ++j) acc = fmaf(w1[j * H1N + k], xs[j * BP + m], acc);
    return fmaxf(acc, 0.0f);
}

// ---- kernel A: sample + gather + normalize -----------------------------------------------------------
__global__ __launch_bounds__(128) void k_prep(shems_ddpg d, shems_replay ring, int64_t ring_len, uint64_t seed, uint32_t tick)
{
    const int m = threadIdx.x;
    float *ws = d.ws;
    float s[SIN], s2[SIN], a0 = 0.f, a1 = 0.f, r = 0.f, dn = 0.f;
    int64_t j = -1;
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
        const float lo = d.s_min[k], den = (d.s_max[k] - lo) + 1e-8f;                 // MPS:56
        ws[WS_XT + k * BP + m] = live ? (s[k] - lo) / den : 0.0f;
        ws[WS_X2T + k * BP + m] = live ? (s2[k] - lo) / den : 0.0f;
    }
    ws[WS_AT + m] = a0; ws[WS_AT + BP + m] = a1;
    ws[WS_R + m] = r; ws[WS_DONE + m] = dn;
    ws[WS_D3Q + m] = live ? -1.0f / (float)d.batch : 0.0f;                            // d(-mean q)/dq
    reinterpret_cast<int32_t *>(ws + WS_IDX)[m] = (int32_t)j;
}

// ---- kernel B: layers 1+2 forward for one 32-wide n-tile and all 128 columns --------------------------
struct FwdJob {
    const float *P;        // parameter block
    int in;                // 9 (actor nets) or 11 (critic nets)
    int out;               // 2 or 1
    XSrc x;
    float *H2;             // [500][BP] or null (target nets: nothing downstream needs it)
    float *P3;             // [NT][2][BP] layer-3 partials of this n-tile
};
struct FwdArgs { FwdJob job[3]; };

constexpr int FWD_KC = 126;                                   // k rows per phase (63 MFMA pairs), 2 phases
constexpr int FWD_LDS = (FWD_KC * BP + FWD_KC * 32 + CIN * BP + CIN * H1N + H1N) * 4;

template <int IN>
__device__ __forceinline__ void fwd_body(const FwdJob &J, float *smem)
{
    float *Hc = smem;                          // [126][BP]
    float *Wc = Hc + FWD_KC * BP;              // [126][32]
    float *xs = Wc + FWD_KC * 32;              // [IN][BP]
    float *w1 = xs + CIN * BP;                 // W1 [IN][250], b1 [250]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const float *__restrict__ P = J.P;

    build_x<IN>(J.x, xs, blockIdx.x == 0);
    for (int e = tid; e < (IN * H1N + H1N) / 2; e += 256)
        reinterpret_cast<float2 *>(w1)[e] = reinterpret_cast<const float2 *>(P)[e];
    __syncthreads();

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const float *__restrict__ W2 = P + off_w2(IN);
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        const int k0 = ph * FWD_KC, kn = ph == 0 ? FWD_KC : H1N - FWD_KC;      // 126 + 124 rows
        // stage W2[k0..][n0..n0+31] (rows of 128 B) -- 8 float4 per row; columns >= 500 of the last tile read the
        // next row / b2 (in bounds) and only feed output rows that are discarded
        for (int e = tid; e < kn * 8; e += 256) {
            const int kl = e >> 3, c = e & 7;
            reinterpret_cast<float4 *>(Wc)[e] = *reinterpret_cast<const float4 *>(W2 + (int64_t)(k0 + kl) * H2N + n0 + 4 * c);
        }
        for (int e = tid; e < kn * BP; e += 256) {
            const int kl = e >> 7, m = e & 127;
            Hc[e] = h1_at<IN>(w1, xs, k0 + kl, m);
        }
        __syncthreads();
        const float *pa = Wc + li, *pb = Hc + wave * 32 + li;
#pragma unroll 9
        for (int s = 0; s < kn / 2; ++s) {
            const int kk = 2 * s + lh;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[kk * 32], pb[kk * BP], acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue: h2 = relu(acc + b2); store; layer-3 partial over this tile's 32 rows
    const float *b2 = P + off_b2(IN), *W3 = P + off_w3(IN);
    const int m = wave * 32 + li;
    float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < H2N) {
            const float h = fmaxf(acc[r] + b2[n], 0.0f);
            if (J.H2) J.H2[n * BP + m] = h;
            if (J.out == 2) { p0 = fmaf(h, W3[2 * n], p0); p1 = fmaf(h, W3[2 * n + 1], p1); }
            else p0 = fmaf(h, W3[n], p0);
        }
    }
    p0 += __shfl_xor(p0, 32, 64);
    p1 += __shfl_xor(p1, 32, 64);
    if (lh == 0) {
        J.P3[(blockIdx.x * 2 + 0) * BP + m] = p0;
        J.P3[(blockIdx.x * 2 + 1) * BP + m] = p1;
    }
}

__global__ __launch_bounds__(256) void k_fwd(FwdArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FwdJob &J = A.job[blockIdx.y];
    if (J.in == SIN) fwd_body<SIN>(J, smem); else fwd_body<CIN>(J, smem);
}

// ---- kernel C: critic loss head (1 workgroup) ------------------------------------------------------------
__global__ __launch_bounds__(128) void k_head_loss(shems_ddpg d)
{
    __shared__ float red[4];
    float *ws = d.ws;
    const int m = threadIdx.x;
    const float *Pt = slot(ws, SLOT_CRITIC_T) + SL_P3, *Pc = slot(ws, SLOT_CRITIC) + SL_P3;
    float q2 = d.critic_t[off_b3(CIN, 1)], q = d.critic[off_b3(CIN, 1)];
#pragma unroll
    for (int t = 0; t < NT; ++t) { q2 += Pt[(t * 2) * BP + m]; q += Pc[(t * 2) * BP + m]; }
    const float y = ws[WS_R + m] + d.gamma * (1.0f - ws[WS_DONE + m]) * q2;            // DDPG.jl:133
    const bool live = m < d.batch;
    const float diff = live ? q - y : 0.0f;
    const float dq = 2.0f * diff / (float)d.batch;                                       // d mse / d q
    ws[WS_Y + m] = y; ws[WS_Q + m] = q; ws[WS_D3C + m] = dq;
    const float s1 = wave_sum(diff * diff), s2 = wave_sum(dq);
    if ((m & 63) == 0) { red[m >> 6] = s1; red[2 + (m >> 6)] = s2; }
    __syncthreads();
    if (m == 0) {
        d.losses[0] = (red[0] + red[1]) / (float)d.batch;                                // Flux.mse
        d.grad_critic[off_b3(CIN, 1)] = red[2] + red[3];
    }
}

// ---- kernel D: layer-2 backward ---------------------------------------------------------------------------
// D2[n][m] = (sum_o W3[n][o] d3[o][m]) * (h2[n][m] > 0) is generated while staging, never stored.
//   W workgroups (kt, nq): gW2[32 k][128 n] = sum_m h1[k][m] D2[n][m]; kt == 0 also emits gb2 and gW3.
//   I workgroups (kt, nq): D1part[nq][32 k][128 m] = sum_{n in quarter} W2[k][n] D2[n][m]; for the critic inside the
//                          actor loss they also emit the partial action gradient (through the layer-1 relu mask).
struct BwdArgs {
    const float *P;        // parameter block of the network being differentiated
    int in, out;
    XSrc x;                // its input (for the layer-1 recompute)
    const float *H2;       // [500][BP]
    const float *d3;       // [out][BP] error at the layer-3 pre-activation
    float *grad;           // gradient block (W part) or null
    float *D1P;            // [NQ][250][BP]
    float *DAP;            // [KT][NQ][2][BP] or null
    int n_w;               // number of W workgroups (32 or 0)
};

constexpr int BWD_LDS = (BP * 129 + 32 * 128 + CIN * BP + CIN * H1N + H1N + AIN * BP) * 4;

template <int IN>
__device__ __forceinline__ void bwd_body(const BwdArgs &A, float *smem)
{
    float *Bt = smem;                          // W: [128 m][129] D2^T panel;  I: [126 n][128 m] D2 panel
    float *At = Bt + BP * 129;                 // W: [128 m][32 k] h1^T panel; I: [32 k][127] W2 panel
    float *xs = At + 32 * 128;                 // [IN][BP]
    float *w1 = xs + CIN * BP;                 // W1, b1
    float *d3 = w1 + (CIN * H1N + H1N);        // [2][BP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const float *__restrict__ P = A.P;
    const float *__restrict__ W3 = P + off_w3(IN);
    const bool is_w = (int)blockIdx.x < A.n_w;
    const int b = is_w ? blockIdx.x : blockIdx.x - A.n_w;
    const int kt = b >> 2, nq = b & 3;

    build_x<IN>(A.x, xs, false);
    for (int e = tid; e < (IN * H1N + H1N) / 2; e += 256)
        reinterpret_cast<float2 *>(w1)[e] = reinterpret_cast<const float2 *>(P)[e];
    for (int e = tid; e < A.out * BP; e += 256) d3[e] = A.d3[e];
    __syncthreads();

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    if (is_w) {
        const int nbase = nq * 128;
        // D2^T panel: Bt[m][nl] for the 128 n of this workgroup (n >= 500 -> 0)
        for (int e = tid; e < 128 * BP; e += 256) {
            const int nl = e >> 7, m = e & 127, n = nbase + nl;
            float v = 0.0f;
            if (n < H2N) {
                const float g = A.out == 2 ? fmaf(W3[2 * n + 1], d3[BP + m], W3[2 * n] * d3[m]) : W3[n] * d3[m];
                v = A.H2[n * BP + m] > 0.0f ? g : 0.0f;
            }
            Bt[m * 129 + nl] = v;
        }
        // h1^T panel: At[m][kl] for the 32 k of this workgroup (k >= 250 -> 0)
        for (int e = tid; e < 32 * BP; e += 256) {
            const int kl = e >> 7, m = e & 127, k = kt * 32 + kl;
            At[m * 32 + kl] = k < H1N ? h1_at<IN>(w1, xs, k, m) : 0.0f;
        }
        __syncthreads();
        const float *pa = At + li, *pb = Bt + wave * 32 + li;
#pragma unroll 8
        for (int s = 0; s < BP / 2; ++s) {
            const int mm = 2 * s + lh;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[mm * 32], pb[mm * 129], acc, 0, 0, 0);
        }
        float *gW2 = A.grad + off_w2(IN);
        const int n = nbase + wave * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (k < H1N && n < H2N) gW2[k * H2N + n] = acc[r];
        }
        if (kt == 0) {
            // gb2[n] = sum_m D2[n][m]; gW3[n][o] = sum_m h2[n][m] d3[o][m]   (4 workgroups cover the 500 rows)
            if (tid < 128 && nbase + tid < H2N) {
                float s = 0.0f;
                for (int m = 0; m < BP; ++m) s += Bt[m * 129 + tid];
                A.grad[off_b2(IN) + nbase + tid] = s;
            }
            for (int nl = wave; nl < 128; nl += 4) {
                const int nn = nbase + nl;
                if (nn >= H2N) break;
                const float h0 = A.H2[nn * BP + lane], h1 = A.H2[nn * BP + 64 + lane];
                for (int o = 0; o < A.out; ++o) {
                    const float s = wave_sum(h0 * d3[o * BP + lane] + h1 * d3[o * BP + 64 + lane]);
                    if (lane == 0) A.grad[off_w3(IN) + nn * A.out + o] = s;
                }
            }
        }
    } else {
        const int nb = nq * NQW;                // 125 n, padded with one zero row to 63 MFMA pairs
        for (int e = tid; e < 126 * BP; e += 256) {
            const int nl = e >> 7, m = e & 127, n = nb + nl;
            float v = 0.0f;
            if (nl < NQW) {
                const float g = A.out == 2 ? fmaf(W3[2 * n + 1], d3[BP + m], W3[2 * n] * d3[m]) : W3[n] * d3[m];
                v = A.H2[n * BP + m] > 0.0f ? g : 0.0f;
            }
            Bt[e] = v;
        }
        const float *__restrict__ W2 = P + off_w2(IN);
        for (int e = tid; e < 32 * 126; e += 256) {
            const int kl = e / 126, nl = e - kl * 126, k = kt * 32 + kl;
            At[kl * 127 + nl] = (k < H1N && nl < NQW) ? W2[(int64_t)k * H2N + nb + nl] : 0.0f;
        }
        __syncthreads();
        const float *pa = At + li * 127, *pb = Bt + wave * 32 + li;
#pragma unroll 9
        for (int s = 0; s < 63; ++s) {
            const int nn = 2 * s + lh;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[nn], pb[nn * BP], acc, 0, 0, 0);
        }
        const int m = wave * 32 + li;
        float *D1 = A.D1P + (int64_t)nq * H1N * BP;
        float da0 = 0.0f, da1 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (k < H1N) {
                D1[k * BP + m] = acc[r];
                if (A.DAP && IN == CIN) {
                    const float v = h1_at<IN>(w1, xs, k, m) > 0.0f ? acc[r] : 0.0f;
                    da0 = fmaf(w1[9 * H1N + k], v, da0);       // W1[9 + o][k]: the action rows of the critic's first layer
                    da1 = fmaf(w1[10 * H1N + k], v, da1);
                }
            }
        }
        if (A.DAP && IN == CIN) {
            da0 += __shfl_xor(da0, 32, 64);
            da1 += __shfl_xor(da1, 32, 64);
            if (lh == 0) {
                float *o = A.DAP + (int64_t)((kt * NQ + nq) * 2) * BP;
                o[m] = da0; o[BP + m] = da1;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_bwd(BwdArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (A.in == SIN) bwd_body<SIN>(A, smem); else bwd_body<CIN>(A, smem);
}

// ---- kernel E: layer-1 gradients, one wave per hidden unit k --------------------------------------------------
//   D1[k][m] = (h1[k][m] > 0) * sum_q D1part[q][k][m];  gb1[k] = sum_m D1;  gW1[j][k] = sum_m x[j][m] D1[k][m]
template <int IN>
__device__ __forceinline__ void l1bwd_body(const float *__restrict__ P, const XSrc &x, const float *__restrict__ D1P,
                                           float *__restrict__ grad, float *xs)
{
    build_x<IN>(x, xs, false);
    __syncthreads();
    const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= H1N) return;
    float w[IN];
#pragma unroll
    for (int j = 0; j < IN; ++j) w[j] = P[j * H1N + k];
    const float b = P[off_b1(IN) + k];
    float dv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m = lane + 64 * h;
        float pre = b;
#pragma unroll
        for (int j = 0; j < IN; ++j) pre = fmaf(w[j], xs[j * BP + m], pre);
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) s += D1P[((int64_t)q * H1N + k) * BP + m];
        dv[h] = pre > 0.0f ? s : 0.0f;
    }
    const float sb = wave_sum(dv[0] + dv[1]);
    if (lane == 0) grad[off_b1(IN) + k] = sb;
#pragma unroll
    for (int j = 0; j < IN; ++j) {
        const float s = wave_sum(xs[j * BP + lane] * dv[0] + xs[j * BP + 64 + lane] * dv[1]);
        if (lane == 0) grad[j * H1N + k] = s;
    }
}

__global__ __launch_bounds__(256) void k_l1bwd(const float *P, int in, XSrc x, const float *D1P, float *grad)
{
    __shared__ float xs[CIN * BP];
    if (in == SIN) l1bwd_body<SIN>(P, x, D1P, grad, xs); else l1bwd_body<CIN>(P, x, D1P, grad, xs);
}

// ---- kernel F: actor head backward (1 workgroup): da -> d3a, actor loss, gb3 ------------------------------------
__global__ __launch_bounds__(256) void k_head_actor(shems_ddpg d)
{
    __shared__ float red[8];
    float *ws = d.ws;
    const int t = threadIdx.x, o = t >> 7, m = t & 127;
    float da = 0.0f;
#pragma unroll 8
    for (int p = 0; p < KT * NQ; ++p) da += ws[WS_DAP + (int64_t)(p * 2 + o) * BP + m];
    const float a = ws[WS_API + t];
    const float d3 = da * (1.0f - a * a);                       // through tanh
    ws[WS_D3A + t] = d3;
    float q = 0.0f;
    if (o == 0 && m < d.batch) {
        const float *Pq = slot(ws, SLOT_CRITIC2) + SL_P3;
        q = d.critic[off_b3(CIN, 1)];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) q += Pq[(tt * 2) * BP + m];
    }
    const float sg = wave_sum(d3), sq = wave_sum(q);
    if ((t & 63) == 0) { red[t >> 6] = sg; red[4 + (t >> 6)] = sq; }
    __syncthreads();
    if (t == 0) {
        d.grad_actor[off_b3(SIN, 2) + 0] = red[0] + red[1];
        d.grad_actor[off_b3(SIN, 2) + 1] = red[2] + red[3];
        d.losses[1] = -(red[4] + red[5]) / (float)d.batch;     // loss_act = -mean(critic(vcat(s, actor(s))))
    }
}

// ---- kernel G: Flux 0.12.1 ADAM + soft target update -----------------------------------------------------------
//   mt = b1*mt + (1-b1)*g ; vt = b2*vt + (1-b2)*g^2 ; delta = mt/(1-bp1) / (sqrt(vt/(1-bp2)) + eps) * eta ; p -= delta
//   (Float64 scalars broadcast over Float32 arrays: each element is computed in f64 and stored as f32)
//   then target = (1f0 - tau) * target + tau * p   (DDPG.jl:99-103)
__global__ __launch_bounds__(256) void k_adam_soft(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ mt,
                                                   float *__restrict__ vt, float *__restrict__ target, int n, double eta,
                                                   double bp1, double bp2, double gscale, float tau)
{
    const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gf = (float)((double)g[i] * gscale);            // averaged gradient, as every replica holds it
    const float m1 = (float)(b1 * (double)mt[i] + (1.0 - b1) * (double)gf);
    const float v1 = (float)(b2 * (double)vt[i] + (1.0 - b2) * ((double)gf * (double)gf));
    const float delta = (float)((double)m1 / (1.0 - bp1) / (sqrt((double)v1 / (1.0 - bp2)) + eps) * eta);
    const float pn = p[i] - delta;
    mt[i] = m1; vt[i] = v1; p[i] = pn;
    const float one_m_tau = 1.0f - tau;
    target[i] = one_m_tau * target[i] + tau * pn;
}

// ---- min_max_buffer (MPS:50-53) -----------------------------------------------------------------------------------
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

static int set_lds_attrs()
{
    static bool done = false;
    if (done) return SHEMS_OK;
    if (int rc = hip_ok(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, FWD_LDS), "attr k_fwd")) return rc;
    if (int rc = hip_ok(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS), "attr k_bwd")) return rc;
    done = true;
    return SHEMS_OK;
}

}  // namespace shems

using namespace shems;

static int check_ddpg(const shems_ddpg *d, const char *fn)
{
    if (!d || !d->actor || !d->critic || !d->actor_t || !d->critic_t || !d->m_actor || !d->v_actor || !d->m_critic ||
        !d->v_critic || !d->grad_actor || !d->grad_critic || !d->s_min || !d->s_max || !d->ws || !d->losses)
        return set_error(SHEMS_ERR_ARG, "%s: shems_ddpg has a NULL buffer", fn);
    if (d->batch < 1 || d->batch > BP) return set_error(SHEMS_ERR_ARG, "%s: batch must be in 1..128 (got %d)", fn, d->batch);
    for (const float *p : {d->actor, d->critic, d->actor_t, d->critic_t})
        if (((uintptr_t)p & 15) != 0) return set_error(SHEMS_ERR_ARG, "%s: parameter blocks must be 16-byte aligned", fn);
    return set_lds_attrs();
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
    hipLaunchKernelGGL(k_prep, dim3(1), dim3(128), 0, st, *d, *ring, ring_len, seed, tick);
    const XSrc x_s2{ws + WS_X2T, nullptr, nullptr, nullptr, nullptr};
    const XSrc x_s{ws + WS_XT, nullptr, nullptr, nullptr, nullptr};
    const XSrc x_s2a{ws + WS_X2T, nullptr, slot(ws, SLOT_ACTOR_T) + SL_P3, d->actor_t + off_b3(SIN, 2), nullptr};
    const XSrc x_sa{ws + WS_XT, ws + WS_AT, nullptr, nullptr, nullptr};
    FwdArgs f;
    std::memset(&f, 0, sizeof f);
    f.job[0] = FwdJob{d->actor_t, SIN, 2, x_s2, nullptr, slot(ws, SLOT_ACTOR_T) + SL_P3};
    hipLaunchKernelGGL(k_fwd, dim3(NT, 1), dim3(256), FWD_LDS, st, f);
    f.job[0] = FwdJob{d->critic_t, CIN, 1, x_s2a, nullptr, slot(ws, SLOT_CRITIC_T) + SL_P3};
    f.job[1] = FwdJob{d->critic, CIN, 1, x_sa, slot(ws, SLOT_CRITIC) + SL_H2, slot(ws, SLOT_CRITIC) + SL_P3};
    f.job[2] = FwdJob{d->actor, SIN, 2, x_s, slot(ws, SLOT_ACTOR) + SL_H2, slot(ws, SLOT_ACTOR) + SL_P3};
    hipLaunchKernelGGL(k_fwd, dim3(NT, 3), dim3(256), FWD_LDS, st, f);
    hipLaunchKernelGGL(k_head_loss, dim3(1), dim3(128), 0, st, *d);
    float *S = slot(ws, SLOT_CRITIC);
    const BwdArgs b{d->critic, CIN, 1, x_sa, S + SL_H2, ws + WS_D3C, d->grad_critic, S + SL_D1P, nullptr, KT * NQ};
    hipLaunchKernelGGL(k_bwd, dim3(2 * KT * NQ), dim3(256), BWD_LDS, st, b);
    hipLaunchKernelGGL(k_l1bwd, dim3((H1N + 3) / 4), dim3(256), 0, st, (const float *)d->critic, (int)CIN, x_sa,
                       (const float *)(S + SL_D1P), d->grad_critic);
    return hip_ok(hipGetLastError(), "ddpg critic_grad launches");
}

static int adam_launch(float *p, const float *g, float *m, float *v, float *target, int n, double eta, double bp1, double bp2,
                       double gscale, float tau, hipStream_t st)
{
    if (!(bp1 > 0.0 && bp1 < 1.0 && bp2 > 0.0 && bp2 < 1.0)) return set_error(SHEMS_ERR_ARG, "adam: beta powers must be in (0,1)");
    hipLaunchKernelGGL(k_adam_soft, dim3((n + 255) / 256), dim3(256), 0, st, p, g, m, v, target, n, eta, bp1, bp2, gscale, tau);
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
    // critic (already updated, DDPG.jl:137-140) on [s; a_pi], a_pi = tanh(b3 + partials of the actor pass)
    const XSrc x_spi{ws + WS_XT, nullptr, slot(ws, SLOT_ACTOR) + SL_P3, d->actor + off_b3(SIN, 2), ws + WS_API};
    const XSrc x_spi_ro{ws + WS_XT, ws + WS_API, nullptr, nullptr, nullptr};
    const XSrc x_s{ws + WS_XT, nullptr, nullptr, nullptr, nullptr};
    float *C2 = slot(ws, SLOT_CRITIC2);
    FwdArgs f;
    std::memset(&f, 0, sizeof f);
    f.job[0] = FwdJob{d->critic, CIN, 1, x_spi, C2 + SL_H2, C2 + SL_P3};
    hipLaunchKernelGGL(k_fwd, dim3(NT, 1), dim3(256), FWD_LDS, st, f);
    const BwdArgs bi{d->critic, CIN, 1, x_spi_ro, C2 + SL_H2, ws + WS_D3Q, nullptr, C2 + SL_D1P, ws + WS_DAP, 0};
    hipLaunchKernelGGL(k_bwd, dim3(KT * NQ), dim3(256), BWD_LDS, st, bi);
    hipLaunchKernelGGL(k_head_actor, dim3(1), dim3(256), 0, st, *d);
    float *S = slot(ws, SLOT_ACTOR);
    const BwdArgs b{d->actor, SIN, 2, x_s, S + SL_H2, ws + WS_D3A, d->grad_actor, S + SL_D1P, nullptr, KT * NQ};
    hipLaunchKernelGGL(k_bwd, dim3(2 * KT * NQ), dim3(256), BWD_LDS, st, b);
    hipLaunchKernelGGL(k_l1bwd, dim3((H1N + 3) / 4), dim3(256), 0, st, (const float *)d->actor, (int)SIN, x_s,
                       (const float *)(S + SL_D1P), d->grad_actor);
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
