// shems_gupd.hip -- replay() (DDPG.jl:121-145) for a GROUP of independent learners, throughput form.
//
// The thesis protocol trains 40 seeds x 10 chargers = 400 independent learners (RL-SHEMS_bs_scheduler_1179_08_on_01-98.sh:67-87).
// shems_ddpg.hip's five launches are shaped for ONE learner's latency (one exposed memory latency per launch, 1-2 workgroups per
// CU, E-product identities that trade FLOPs and 3 MB of slabs per learner for fewer grid-wide dependencies).  With hundreds of
// learners per launch nothing is latency-bound any more: per learner the update is 308 MFLOP (ten 250x500x128 products) against
// ~11 MB of parameter / moment traffic, i.e. ~2 us at the fp32 MFMA peak and ~1.6 us at 6.3 TB/s -- both roofs within a factor
// of two of each other.  This file is the same arithmetic laid out for that regime:
//
//   * plain back-propagation (no E slabs): per learner ten products, every one a stream of 32-deep weight chunks through a small
//     LDS ring against operands that live in REGISTERS in the MFMA layout (layer 1 is recomputed per chunk on the matrix pipe --
//     K = 12 -- and its D layout IS the next product's B operand; error signals are loaded from HBM straight into operand layout);
//   * 30-47 KB of LDS and 94-143 VGPRs per workgroup: 3-4 workgroups (12-16 waves) resident per CU, so one workgroup's barrier / global
//     latency is covered by the others' MFMAs;
//   * loads retire in order (s_waitcnt vmcnt counts from the oldest): every kernel issues LAST what it will wait for last -- the next
//     weight chunk is never waited for inside the chunk that requested it, the W2 tile's ADAM state is requested after every small operand
//     and arrives under the product (what this cost before it was looked for in the ISA: profiles/NOTES.md, round-5 log);
//   * activations that must cross a launch (relu(layer 2) of the two differentiated networks, 256 KB each) are written once and
//     read from L2 / Infinity Cache; layer-1 activations are never stored; gradients are never stored (ADAM + the soft target
//     update are applied by the lane that holds the finished element; SHEMS_TP_STORE_GRAD keeps the gradient for the tests).
//
// Launches (grid y = learner; all learners advance in lockstep, sharing the ADAM scalars):
//   P0 k_tp_prep   sample (StatsBase.sample with replacement, MPS:33) + gather + normalize (MPS:56); frozen layer-1 images of the
//                  four networks and frozen output layers (the in-place updates below must not be read half-way)
//   P1 k_tp_fwd    actor_target(s') | critic(s, a) | actor(s)                                      (DDPG.jl:131, 114-119)
//   P2 k_tp_fwd    critic_target(s', a')                                                           (DDPG.jl:132)
//   P3 k_tp_d1     y, dq = d mse / dq; D1 = mask1 .* (W2 D2); layer-1 gradient + ADAM (+ the updated layer-1 image P5 reads); b3   (DDPG.jl:133-137)
//   P4 k_tp_gw2    gW2 = h1 D2' + ADAM + soft update per tile; gb2, gW3 + ADAM
//   P5 k_tp_fwd<QG> updated critic on [s; actor(s)], forward and input gradient per n-tile          (DDPG.jl:117-119, 140)
//   P6 k_tp_d1     actor: through tanh, D1, layer-1 gradient + ADAM
//   P7 k_tp_gw2    actor: gW2 + ADAM + soft update, gb2, gW3                                        (DDPG.jl:140-143)
// Summation orders differ from the latency form (64-wide n-tiles, chunked contractions): per learner the results agree with it to
// fp32 accumulation accuracy, not bit for bit -- the parity test holds every gradient block of every learner to the float64
// evaluation with the same bound as the latency form (tests/test_group_gpu.py).
#include <hip/hip_runtime.h>

#include <cstring>
#include <type_traits>

#include "philox.h"
#include "shems_adam.h"
#include "shems_internal.h"

namespace shems {
namespace tp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BP = 128;
constexpr int H1N = SHEMS_L1, H2N = SHEMS_L2;
constexpr int SIN = 9, CIN = 11;
constexpr int NT = 8;              // n-tiles of 64 over the 500 (512) layer-2 units
constexpr int W1K = 12, W1C = 256; // layer-1 image: rows 0..in-1 = W1, row 11 = b1, everything else zero; columns >= 250 zero

__host__ __device__ constexpr int off_b1(int in) { return in * H1N; }
__host__ __device__ constexpr int off_w2(int in) { return in * H1N + H1N; }
__host__ __device__ constexpr int off_b2(int in) { return off_w2(in) + H1N * H2N; }
__host__ __device__ constexpr int off_w3(int in) { return off_b2(in) + H2N; }
__host__ __device__ constexpr int off_b3(int in, int out) { return off_w3(in) + H2N * out; }

// ---- TILED working layout of a network's layer-2 state (shems_group_w2t; round 6) ----------------------------------------------
// W2 [250][500], its two ADAM moments and its target as 4 x 8 tiles of 64 x 64 (rows / columns padded to 256 / 512 with zeros), the four
// arrays of a tile ADJACENT: region [kt][nt][m | v | p | target][64 rows][64 columns], 64 KB per tile, 2 MB per network.  P4 / P7 read and
// write one tile's 64 KB as ONE contiguous piece (tools/micro/adam_stream.hip: 4.78 TB/s against 3.85 TB/s for the 256-byte row pieces
// of the Flux order and 4.56 TB/s for a device copy of the same bytes); the forward / D1 launches find a 32-row chunk of a 64-wide n-tile
// as one contiguous 8 KB piece.  The Flux-order blocks keep every other parameter (layer 1, b2, W3, b3) and are the API's view of W2:
// shems_group_w2_to_tiled / _to_flux convert (csrc below), group.py keeps track of which copy is current.
constexpr int TL_TILE = 64 * 64;                   // floats of one array of one tile
constexpr int TL_BLOCK = 4 * TL_TILE;              // one tile: m | v | p | target
enum { TL_M = 0, TL_V = 1, TL_P = 2, TL_T = 3 };
static_assert(SHEMS_W2T_FLOATS == 32 * TL_BLOCK, "shems_hip.h and the kernels agree on the tiled region's size");
__host__ __device__ constexpr int64_t tl_tile(int kt, int nt) { return (int64_t)(kt * 8 + nt) * TL_BLOCK; }

// ---- workspace carve (floats, inside shems_ddpg.ws; see kTpWsFloats) -----------------------------------------------------------
constexpr int64_t TP_X = 0;                            // [12][BP]  rows 0..8 normalize(s), 9..10 stored action, 11 = 1
constexpr int64_t TP_X2 = TP_X + W1K * BP;             // [12][BP]  rows 0..8 normalize(s'), 9..10 zero, 11 = 1
constexpr int64_t TP_R = TP_X2 + W1K * BP;             // [BP]
constexpr int64_t TP_DONE = TP_R + BP;                 // [BP]
constexpr int64_t TP_IDX = TP_DONE + BP;               // [BP] int32 sampled ring slots (-1 in the pad columns)
constexpr int64_t TP_W1I = TP_IDX + BP;                // [5][12][256] frozen layer-1 images of the four networks + the critic's AFTER its update
constexpr int64_t TP_FW3C = TP_W1I + 5 * W1K * W1C;    // [512][2] frozen critic W3 ([.][1] and rows >= 500 zero)
constexpr int64_t TP_FW3A = TP_FW3C + 1024;            // [512][2] frozen actor W3
constexpr int64_t TP_FB3 = TP_FW3A + 1024;             // [8] frozen b3: critic, critic_target, actor[0], actor[1], actor_target[0], [1]
constexpr int64_t TP_P3 = TP_FB3 + 8;                  // [5 passes][NT][2][BP] layer-3 partial sums per n-tile
constexpr int64_t TP_DAP = TP_P3 + 5 * NT * 2 * BP;    // [NT][2][BP] partial d loss / d a_pi per n-tile (P5)
constexpr int64_t TP_D3C = TP_DAP + NT * 2 * BP;       // [2][BP] dq (row 1 zero)
constexpr int64_t TP_D3A = TP_D3C + 2 * BP;            // [2][BP] error at the actor's pre-tanh output
constexpr int64_t TP_API = TP_D3A + 2 * BP;            // [2][BP] a_pi = actor(s)
constexpr int64_t TP_H2C = TP_API + 2 * BP;            // [512][BP] relu(layer 2) of the critic (rows >= 500 zero)
constexpr int64_t TP_H2A = TP_H2C + 512 * BP;          // [512][BP] ... of the actor
constexpr int64_t TP_FLOATS = TP_H2A + 512 * BP;
static_assert(TP_FLOATS <= kTpWsFloats, "the throughput form's carve fits the workspace every caller allocates");
static_assert(TP_W1I % 4 == 0 && TP_H2C % 4 == 0 && TP_P3 % 4 == 0, "16-byte aligned blocks");
enum { NET_ACTOR_T = 0, NET_CRITIC_T = 1, NET_CRITIC = 2, NET_ACTOR = 3, PASS_CRITIC2 = 4 };
constexpr int IMG_CRITIC_NEW = 4;      // image slot P3 fills with the critic's updated layer 1 (P5 reads it; P3 / P4 keep reading the frozen one)
__host__ __device__ inline float *p3_of(float *ws, int pass) { return ws + TP_P3 + (int64_t)pass * NT * 2 * BP; }
__host__ __device__ inline float *w1i_of(float *ws, int net) { return ws + TP_W1I + (int64_t)net * W1K * W1C; }

// rows of a 32 x 32 MFMA accumulator held by register r of a lane in half lh
__device__ __forceinline__ int drow(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

__device__ __forceinline__ float half_sum32(float x)      // sum over the 32 lanes of a half wave, in every lane of that half
{
    x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64); x += __shfl_xor(x, 4, 64); x += __shfl_xor(x, 8, 64); x += __shfl_xor(x, 16, 64);
    return x;
}
__device__ __forceinline__ float wave_sum64(float x) { x = half_sum32(x); return x + __shfl_xor(x, 32, 64); }

__device__ __forceinline__ void adam_at(const AdamCtx &c, int i, float g, bool store_grad)
{
    float m = c.mt[i], v = c.vt[i], p = c.p[i], t = c.target[i];
    adam_math(c, g, m, v, p, t);
    c.mt[i] = m; c.vt[i] = v; c.p[i] = p; c.target[i] = t;
    if (store_grad) const_cast<float *>(c.g)[i] = g;
}

// ================================================================================================================================
// P0: sample + gather + normalize; frozen images
// ================================================================================================================================
struct PrepArgs {
    shems_ddpg d;
    shems_replay ring;
    int64_t ring_len, gstride;
    uint64_t seed;
    uint32_t tick;
};
__device__ __forceinline__ void gshift(shems_ddpg &d, int64_t off)
{
    d.actor = gsh(d.actor, off); d.critic = gsh(d.critic, off); d.actor_t = gsh(d.actor_t, off); d.critic_t = gsh(d.critic_t, off);
    d.m_actor = gsh(d.m_actor, off); d.v_actor = gsh(d.v_actor, off); d.m_critic = gsh(d.m_critic, off); d.v_critic = gsh(d.v_critic, off);
    d.grad_actor = gsh(d.grad_actor, off); d.grad_critic = gsh(d.grad_critic, off);
    d.s_min = gsh(d.s_min, off); d.s_max = gsh(d.s_max, off); d.ws = gsh(d.ws, off); d.losses = gsh(d.losses, off);
}
__device__ __forceinline__ void gshift(shems_replay &r, int64_t off)
{
    r.s = gsh(r.s, off); r.a = gsh(r.a, off); r.r = gsh(r.r, off); r.s2 = gsh(r.s2, off); r.done = gsh(r.done, off);
}

// grid (5, learners): workgroup 0 samples / gathers / normalises and freezes the output layers, workgroups 1..4 pack one network's
// frozen layer-1 image each (one launch of 2 000 short workgroups instead of 400 long ones: 49 -> ~10 us at 400 learners)
__device__ __forceinline__ void prep_body(const PrepArgs &A, const int role, const int l)
{
    const int tid = threadIdx.x;
    const int64_t off = (int64_t)l * A.gstride;
    shems_ddpg d = A.d;
    shems_replay ring = A.ring;
    gshift(d, off); gshift(ring, off);
    float *ws = d.ws;
    if (role > 0) {
        const int net = role - 1;
        const float *P = net == NET_ACTOR_T ? d.actor_t : net == NET_CRITIC_T ? d.critic_t : net == NET_CRITIC ? d.critic : d.actor;
        const int in = (net == NET_CRITIC_T || net == NET_CRITIC) ? CIN : SIN;
        float *img = w1i_of(ws, net);
        const int k = min(tid, H1N - 1);
        float v[W1K];
#pragma unroll
        for (int j = 0; j < W1K; ++j) v[j] = P[(j == W1K - 1 ? in : min(j, in - 1)) * H1N + k];       // clamped, all twelve loads in flight
#pragma unroll
        for (int j = 0; j < W1K; ++j) img[j * W1C + tid] = ((j < in || j == W1K - 1) && tid < H1N) ? v[j] : 0.0f;
        if (net == NET_CRITIC) {                 // the zeros of the slot P3 writes the updated elements into
            float *img2 = w1i_of(ws, IMG_CRITIC_NEW);
#pragma unroll
            for (int j = 0; j < W1K; ++j) img2[j * W1C + tid] = 0.0f;
        }
        return;
    }
    const uint64_t seed = A.seed + (uint64_t)l;               // learner l: Philox key seed + l (as the latency form)
    if (tid < BP) {
        const int m = tid;
        const u32x4 x = philox4x32_10((uint32_t)(m >> 2), 0u, A.tick, kStreamSample, (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint32_t w = (m & 3) == 0 ? x.x : (m & 3) == 1 ? x.y : (m & 3) == 2 ? x.z : x.w;
        const int64_t j = (int64_t)(w % (uint32_t)A.ring_len);
        const bool live = m < d.batch;
        float sv[SIN], s2v[SIN], lo[SIN], hi[SIN];
#pragma unroll
        for (int k = 0; k < SIN; ++k) { sv[k] = ring.s[j * SIN + k]; s2v[k] = ring.s2[j * SIN + k]; lo[k] = d.s_min[k]; hi[k] = d.s_max[k]; }
        const float a0 = ring.a[j * 2], a1 = ring.a[j * 2 + 1], rr = ring.r[j];
        const bool dn = ring.done[j] != 0;
#pragma unroll
        for (int k = 0; k < SIN; ++k) {
            const float den = (hi[k] - lo[k]) + 1e-8f;                                     // MPS:56
            ws[TP_X + k * BP + m] = live ? (sv[k] - lo[k]) / den : 0.0f;
            ws[TP_X2 + k * BP + m] = live ? (s2v[k] - lo[k]) / den : 0.0f;
        }
        ws[TP_X + 9 * BP + m] = live ? a0 : 0.0f;
        ws[TP_X + 10 * BP + m] = live ? a1 : 0.0f;
        ws[TP_X + 11 * BP + m] = 1.0f;
        ws[TP_X2 + 9 * BP + m] = 0.0f; ws[TP_X2 + 10 * BP + m] = 0.0f; ws[TP_X2 + 11 * BP + m] = 1.0f;
        ws[TP_R + m] = live ? rr : 0.0f;
        ws[TP_DONE + m] = live && dn ? 1.0f : 0.0f;
        reinterpret_cast<int32_t *>(ws + TP_IDX)[m] = live ? (int32_t)j : -1;
    }
    // frozen output layers
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int e = it * 256 + tid, n = e >> 1, o = e & 1;
        const float wc = d.critic[off_w3(CIN) + min(n, H2N - 1)], wa = d.actor[off_w3(SIN) + min(e, 2 * H2N - 1)];
        ws[TP_FW3C + e] = (n < H2N && o == 0) ? wc : 0.0f;
        ws[TP_FW3A + e] = n < H2N ? wa : 0.0f;
    }
    if (tid < 8)
        ws[TP_FB3 + tid] = tid == 0 ? d.critic[off_b3(CIN, 1)] : tid == 1 ? d.critic_t[off_b3(CIN, 1)]
                         : tid == 2 ? d.actor[off_b3(SIN, 2)] : tid == 3 ? d.actor[off_b3(SIN, 2) + 1]
                         : tid == 4 ? d.actor_t[off_b3(SIN, 2)] : tid == 5 ? d.actor_t[off_b3(SIN, 2) + 1] : 0.0f;
}

// ================================================================================================================================
// P1 / P2 / P5: forward through layers 1 + 2 for one 64-wide n-tile and the whole batch; layer 3 as per-tile partial sums.
// QG (P5): the same workgroup then back-propagates the constant upstream gradient of -mean(q) through its own 64 hidden units.
// ================================================================================================================================
struct FwdJob {
    const float *w2t;      // tiled layout: the network's region + TL_P / TL_T array offset (its W2 is read from there); null: Flux order, from P
    const float *P;        // parameter block of the network
    const float *X;        // [12][BP] input block (workspace)
    const float *w1i;      // layer-1 image (workspace): frozen by P0, or written by P3 (the critic after its update)
    const float *ap3;      // rows 9, 10 = tanh(ab3 + sum of these [NT][2][BP] partials) of an actor pass, or null (rows as stored in X)
    const float *ab3;      // [2]
    float *api;            // n-tile 0 publishes the computed action here ([2][BP]), or null
    float *H2;             // [512][BP] relu(layer 2) (rows >= 500 stored as zero), or null
    float *P3;             // [NT][2][BP]
    float *DAP;            // QG: [NT][2][BP]
    int in, out;
};
struct FwdArgs { FwdJob job[3]; int64_t gstride; int batch; };
__device__ __forceinline__ void gshift(FwdJob &J, int64_t off)
{
    J.w2t = gsh(J.w2t, off); J.P = gsh(J.P, off); J.X = gsh(J.X, off); J.w1i = gsh(J.w1i, off); J.ap3 = gsh(J.ap3, off); J.ab3 = gsh(J.ab3, off);
    J.api = gsh(J.api, off); J.H2 = gsh(J.H2, off); J.P3 = gsh(J.P3, off); J.DAP = gsh(J.DAP, off);
}

// W2 chunk c of an n-tile of NTL MFMA tiles (64 or 128 columns): rows k = 32 c .. 32 c + 31, columns n0 .. (256 / 512 B per row), NTL float4
// per thread.  Rows >= 250 are
// copies of row 249 (clamped address): they only meet layer-1 activations that are exactly zero (the image has no columns >= 250)
// or output rows whose relu mask is off.  Columns >= 500 of the last tile read on into the next row / b2 (inside the parameter
// block) and only feed outputs that are discarded.
template <int NTL>
__device__ __forceinline__ void fwd_chunk_load(const float *__restrict__ W2, int n0, int c, f32x4 (&v)[NTL])
{
#pragma unroll
    for (int it = 0; it < NTL; ++it) {
        const int e = it * 256 + (int)threadIdx.x, k = min(32 * c + e / (8 * NTL), H1N - 1);
        v[it] = *reinterpret_cast<const f32x4 *>(W2 + (int64_t)k * H2N + n0 + 4 * (e % (8 * NTL)));
    }
}
// The same chunk from the tiled layout: per 64 columns the 32 rows are ONE contiguous 8 KB piece (512 float4) of the tile (kt = c / 2,
// rows 32 (c & 1) ..); pad rows / columns hold zeros.  fwd_chunk_store_t puts float4 r of piece j at LDS row r / 16, column 64 j + 4 (r % 16).
template <int NTL>
__device__ __forceinline__ void fwd_chunk_load_t(const float *__restrict__ Wt, int n0, int c, f32x4 (&v)[NTL])
{
    const float *base = Wt + tl_tile(c >> 1, n0 >> 6) + (c & 1) * (32 * 64);
#pragma unroll
    for (int it = 0; it < NTL; ++it) {
        const int e = it * 256 + (int)threadIdx.x;
        v[it] = *reinterpret_cast<const f32x4 *>(base + (int64_t)(e >> 9) * TL_BLOCK + 4 * (e & 511));
    }
}
template <int S, int NTL>
__device__ __forceinline__ void fwd_chunk_store_t(float *buf, const f32x4 (&v)[NTL])
{
#pragma unroll
    for (int it = 0; it < NTL; ++it) {
        const int e = it * 256 + (int)threadIdx.x, r = e & 511;
        float *p = buf + (r >> 4) * S + 64 * (e >> 9) + 4 * (r & 15);
        if constexpr (S % 4 == 0) *reinterpret_cast<f32x4 *>(p) = v[it];
        else { p[0] = v[it][0]; p[1] = v[it][1]; p[2] = v[it][2]; p[3] = v[it][3]; }
    }
}
template <int S, int NTL>
__device__ __forceinline__ void fwd_chunk_store(float *buf, const f32x4 (&v)[NTL])
{
#pragma unroll
    for (int it = 0; it < NTL; ++it) {
        const int e = it * 256 + (int)threadIdx.x;
        float *p = buf + (e / (8 * NTL)) * S + 4 * (e % (8 * NTL));
        if constexpr (S % 4 == 0) *reinterpret_cast<f32x4 *>(p) = v[it];
        else { p[0] = v[it][0]; p[1] = v[it][1]; p[2] = v[it][2]; p[3] = v[it][3]; }
    }
}

// layer-1 pre-activations of hidden units 32 c .. 32 c + 31 x this wave's 32 batch columns, D layout (row = unit, column = sample)
__device__ __forceinline__ f32x16 l1_tile(const float *w1s, int c, const float (&xreg)[6], int li, int lh)
{
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = 0.0f;
    float a[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) a[s] = w1s[(2 * s + lh) * W1C + 32 * c + li];
#pragma unroll
    for (int s = 0; s < 6; ++s) t = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], xreg[s], t, 0, 0, 0);
    return t;
}

// NTL = MFMA tiles per wave along n.  4 = a 128-wide n-tile: layer 1 is recomputed per n-tile, 6 MFMAs per chunk beside 64 instead of
// beside 32 -- taken where the launch still has several rounds of workgroups (P1: three networks x 4 tiles x learners); 2 = 64-wide
// n-tiles, twice the workgroups at four per CU: P2 (one network: 1 600 workgroups of the wide form would be 2.1 rounds of 768 slots)
// and the pass with the input gradient (its second sweep holds 16 more accumulators).
template <bool QG, int NTL_> struct FwdShape {
    static constexpr int NTL = NTL_;
    static constexpr int NW = 32 * NTL;            // n-tile width
    static constexpr int TILES = 512 / NW;         // n-tiles per network
    static constexpr int S = QG ? NW + 1 : NW;     // chunk row stride: the backward pass reads the chunk along n (odd stride: conflict free)
    static constexpr int LDS = (W1K * W1C + NW * 4 + 2 * 32 * S) * 4;
};

template <bool QG, int NTL_, bool TL>
__device__ __forceinline__ void fwd_body(const FwdArgs &A, const int bx, const int by, float *smem)
{
    typedef FwdShape<QG, NTL_> SH;
    constexpr int S = SH::S, NTL = SH::NTL, NW = SH::NW;
    float *w1s = smem;                           // [12][256]
    float *ep = w1s + W1K * W1C;                 // [NW][4]: b2, W3[.][0], W3[.][1], valid
    float *ring = ep + NW * 4;                   // [2][32][S]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int job = bx / SH::TILES, nt = bx % SH::TILES, n0 = NW * nt;
    FwdJob J = job == 0 ? A.job[0] : job == 1 ? A.job[1] : A.job[2];
    gshift(J, (int64_t)by * A.gstride);
    const float *__restrict__ P = J.P;
    const float *__restrict__ W2 = TL ? J.w2t : P + off_w2(J.in);
    const int m = 32 * w + li;
    auto chunk_load = [&](int c, f32x4 (&v)[NTL]) {
        if constexpr (TL) fwd_chunk_load_t<NTL>(W2, n0, c, v); else fwd_chunk_load<NTL>(W2, n0, c, v);
    };
    auto chunk_store = [&](float *buf, const f32x4 (&v)[NTL]) {
        if constexpr (TL) fwd_chunk_store_t<S, NTL>(buf, v); else fwd_chunk_store<S, NTL>(buf, v);
    };

    // Every request of the prologue is issued before the first wait: the workgroups of a CU start together (at the launch and, staying
    // in step, after each generation), so their prologues are not covered by anybody's matrix work -- one exposed round trip, not four.
    f32x4 pv[NTL];
    chunk_load(0, pv);
    f32x4 wi[3];
    {
        const f32x4 *g4 = reinterpret_cast<const f32x4 *>(J.w1i) + tid;
        wi[0] = g4[0]; wi[1] = g4[256]; wi[2] = g4[512];
    }
    float eb2 = 0.0f, ew30 = 0.0f, ew31 = 0.0f;
    if (tid < NW) {
        const int nc = min(n0 + tid, H2N - 1);
        eb2 = P[off_b2(J.in) + nc]; ew30 = P[off_w3(J.in) + nc * J.out];
        if (J.out == 2) ew31 = P[off_w3(J.in) + nc * 2 + 1];
    }
    // this lane's B operands of layer 1: x[2 s + lh][m]
    float xreg[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) xreg[s] = J.X[(2 * s + lh) * BP + m];
    float a0 = 0.0f, a1 = 0.0f, pa3[2 * NT];
    if (J.ap3) {
        a0 = J.ab3[0]; a1 = J.ab3[1];
#pragma unroll
        for (int t = 0; t < NT; ++t) { pa3[2 * t] = J.ap3[(t * 2 + 0) * BP + m]; pa3[2 * t + 1] = J.ap3[(t * 2 + 1) * BP + m]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    // layer-1 image -> LDS
    {
        f32x4 *l4 = reinterpret_cast<f32x4 *>(w1s) + tid;
        l4[0] = wi[0]; l4[256] = wi[1]; l4[512] = wi[2];
    }
    if (tid < NW) {
        const float valid = n0 + tid < H2N ? 1.0f : 0.0f;
        ep[tid * 4 + 0] = eb2; ep[tid * 4 + 1] = ew30 * valid; ep[tid * 4 + 2] = ew31 * valid; ep[tid * 4 + 3] = valid;
    }
    if (J.ap3) {                                 // rows 9, 10 from an actor pass: a[o][m] = tanh(b3[o] + partials)
#pragma unroll
        for (int t = 0; t < NT; ++t) { a0 += pa3[2 * t]; a1 += pa3[2 * t + 1]; }
        a0 = tanhf(a0); a1 = tanhf(a1);          // Dense(500, 2, tanh)
        if (lh == 1) xreg[4] = a0; else xreg[5] = a1;          // row 9 = (s 4, lh 1), row 10 = (s 5, lh 0)
        if (J.api && nt == 0) J.api[lh * BP + m] = lh ? a1 : a0;
    }
    chunk_store(ring, pv);
    // The operands above are consumed inside the chunk loop only.  Loads retire in order, so without this the compiler's wait in front of
    // their first use (counted for the loop's entry edge) also waits, on every later iteration, for the chunk prefetch issued just before it.
#pragma unroll
    for (int s = 0; s < 6; ++s) asm volatile("" : "+v"(xreg[s]));
    __syncthreads();

    f32x16 acc[NTL];
#pragma unroll
    for (int tt = 0; tt < NTL; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tt][r] = 0.0f;
    float da0 = 0.0f, da1 = 0.0f;
    unsigned mbits = 0u;                         // QG: bit 16 tt + r = (h2 > 0) of this lane's element (tt, r)
    const float d3q = m < A.batch ? -1.0f / (float)A.batch : 0.0f;          // d(-mean q)/dq
    // One pass over the n-tile's eight weight chunks; chunk it + 1 is requested before chunk it is consumed and lands in the other half
    // of the ring.  BWD = false: layer 2 forward; BWD = true (QG only): the input gradient through this n-tile.
    auto pass = [&](auto bwd, int it0, bool wrap) {
        constexpr bool BWD = decltype(bwd)::value;
#pragma unroll 1
        for (int it = it0; it < it0 + 8; ++it) {
            const int c = it & 7;
            const float *buf = ring + (it & 1) * 32 * S;
            const bool more = wrap || c < 7;
            if (more) chunk_load((it + 1) & 7, pv);
            __builtin_amdgcn_sched_barrier(0);          // (where the request is unconditional the scheduler otherwise sinks it to the end of the iteration)
            if constexpr (!BWD) {
                // layer 2: k-step r contracts over the two hidden units {32 c + drow(r, 0), 32 c + drow(r, 1)}; B = relu(t[r]) from registers.
                // The A operands of step r + 1 are read from LDS before the products of step r are issued (read -> wait -> two products, as
                // the compiler lays the plain loop out, leaves the LDS latency uncovered behind every pair: ~3/4 of the pipe for one wave).
                const float *pa = buf + 4 * lh * S + li;
                float a0[NTL], a1[NTL];
#pragma unroll
                for (int tt = 0; tt < NTL; ++tt) a0[tt] = pa[32 * tt];
                const f32x16 t = l1_tile(w1s, c, xreg, li, lh);
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
#pragma unroll
                    for (int tt = 0; tt < NTL; ++tt) a1[tt] = pa[(((r + 1) & 3) + 8 * ((r + 1) >> 2)) * S + 32 * tt];
                    __builtin_amdgcn_sched_barrier(0);
                    const float b0 = fmaxf(t[r], 0.0f);
#pragma unroll
                    for (int tt = 0; tt < NTL; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[tt], b0, acc[tt], 0, 0, 0);
                    if (r < 14) {
#pragma unroll
                        for (int tt = 0; tt < NTL; ++tt) a0[tt] = pa[(((r + 2) & 3) + 8 * ((r + 2) >> 2)) * S + 32 * tt];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const float b1 = fmaxf(t[r + 1], 0.0f);
#pragma unroll
                    for (int tt = 0; tt < NTL; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[tt], b1, acc[tt], 0, 0, 0);
                }
            } else {
                // backward through this n-tile: D1part[k][m] = sum_{n in tile} W2[k][n] M[n][m], rows k = 32 c .. + 31; M[n][m] = d3q[m] W3[n]
                // (h2[n][m] > 0) is rebuilt per k-step from one mask bit and the epilogue block in LDS (32 registers less than keeping it:
                // with those the four-waves budget spilled, and a scratch reload in this loop waits for the chunk prefetch as well)
                f32x16 g;
#pragma unroll
                for (int r = 0; r < 16; ++r) g[r] = 0.0f;
                const float *pa = buf + li * S + 4 * lh;
                const float *pe = ep + 16 * lh + 1;
                // (operands of step i + 1 read before the product of step i is issued, as in the forward pass)
                auto nbof = [](int i) { return 32 * (i >> 4) + (i & 3) + 8 * ((i & 15) >> 2); };
                float aA = pa[0], wA = pe[0], aB, wB;
#pragma unroll
                for (int i = 0; i < 16 * NTL; i += 2) {
                    aB = pa[nbof(i + 1)]; wB = pe[nbof(i + 1) * 4];
                    __builtin_amdgcn_sched_barrier(0);
                    g = __builtin_amdgcn_mfma_f32_32x32x2f32(aA, (mbits >> i) & 1u ? wA * d3q : 0.0f, g, 0, 0, 0);
                    if (i + 2 < 16 * NTL) { aA = pa[nbof(i + 2)]; wA = pe[nbof(i + 2) * 4]; }
                    __builtin_amdgcn_sched_barrier(0);
                    g = __builtin_amdgcn_mfma_f32_32x32x2f32(aB, (mbits >> (i + 1)) & 1u ? wB * d3q : 0.0f, g, 0, 0, 0);
                }
                // layer-1 pre-activations for the mask only now: 16 fewer live registers under the product (with them the 128-register
                // budget of four waves per SIMD spilled two of M's values, and a scratch reload waits for the chunk prefetch as well)
                __builtin_amdgcn_sched_barrier(0);
                const f32x16 t = l1_tile(w1s, c, xreg, li, lh);
                const float *pw = w1s + 9 * W1C + 32 * c + 4 * lh;             // W1[9 + o][k]: the action rows (columns >= 250 zero)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = t[r] > 0.0f ? g[r] : 0.0f;                // layer-1 relu mask
                    da0 = fmaf(pw[(r & 3) + 8 * (r >> 2)], v, da0);
                    da1 = fmaf(pw[W1C + (r & 3) + 8 * (r >> 2)], v, da1);
                }
            }
            if (more) chunk_store(ring + ((it + 1) & 1) * 32 * S, pv);
            __syncthreads();
        }
    };
    pass(std::false_type{}, 0, QG);
    // epilogue of the forward pass: bias, relu, store, layer-3 partials; QG: the relu mask of the tile is kept, one bit per element
    auto epilogue = [&](auto keep) {
        constexpr bool KEEP = decltype(keep)::value;
        const float *epl = ep + 16 * lh;
        float *hp = J.H2 + (n0 + 4 * lh) * BP + m;
#pragma unroll
        for (int hf = 0; hf < NTL / 2; ++hf) {                                  // one pair of layer-3 partials per 64 columns: the consumers add
            float p0 = 0.0f, p1 = 0.0f;                                         // NT = 8 of them whatever the tile width
#pragma unroll
            for (int th = 0; th < 2; ++th)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tt = 2 * hf + th;
                    const int nb = 32 * tt + (r & 3) + 8 * (r >> 2);          // + 4 lh = the tile row
                    const f32x4 e = *reinterpret_cast<const f32x4 *>(epl + nb * 4);
                    const float h = fmaxf(acc[tt][r] + e[0], 0.0f) * e[3];
                    if constexpr (KEEP) hp[nb * BP] = h;
                    p0 = fmaf(h, e[1], p0);
                    p1 = fmaf(h, e[2], p1);
                    if (QG && h > 0.0f) mbits |= 1u << (16 * tt + r);
                }
            p0 += __shfl_xor(p0, 32, 64);
            p1 += __shfl_xor(p1, 32, 64);
            const int slot = (NTL / 2) * nt + hf;
            if (lh == 0) { J.P3[(slot * 2 + 0) * BP + m] = p0; J.P3[(slot * 2 + 1) * BP + m] = p1; }
        }
    };
    if (J.H2) epilogue(std::true_type{}); else epilogue(std::false_type{});
    if constexpr (QG) pass(std::true_type{}, 8, false);
    if (QG) {
        da0 += __shfl_xor(da0, 32, 64);
        da1 += __shfl_xor(da1, 32, 64);
        if (lh == 0) { J.DAP[(nt * 2 + 0) * BP + m] = da0; J.DAP[(nt * 2 + 1) * BP + m] = da1; }
    }
}

// ================================================================================================================================
// The network being differentiated (P3 / P4: critic, P6 / P7: actor)
// ================================================================================================================================
struct NetArgs {
    shems_ddpg d;          // learner 0's record (heads, losses, workspace)
    float *w2t;            // tiled layout: the network's region (learner 0's); null: Flux order
    AdamCtx c;
    int64_t gstride;
    int head;              // 1 = critic loss head, 2 = actor head
    int store_grad;
    int learners;
};
template <int IN> struct NetOf {
    static constexpr bool critic = IN == CIN;
    static constexpr int OUT = critic ? 1 : 2;
    __device__ static const float *X(const float *ws) { return ws + TP_X; }
    __device__ static const float *w1i(float *ws) { return w1i_of(ws, critic ? NET_CRITIC : NET_ACTOR); }
    __device__ static const float *w3f(const float *ws) { return ws + (critic ? TP_FW3C : TP_FW3A); }
    __device__ static const float *H2(const float *ws) { return ws + (critic ? TP_H2C : TP_H2A); }
    __device__ static float *d3(float *ws) { return ws + (critic ? TP_D3C : TP_D3A); }
};

// ---- heads: the error signal at the output layer, d3 [2][BP] into LDS (every thread of the workgroup takes part); the publishing
// workgroup also stores it for P4 / P7, reports the loss and applies the b3 gradient ----
__device__ __forceinline__ void head_critic(const NetArgs &A, const shems_ddpg &d, const AdamCtx &c, float *d3s, float *red, bool publisher)
{
    float *ws = d.ws;
    const int t = threadIdx.x, m = t & 127;
    float dq = 0.0f, diff = 0.0f;
    if (t < BP) {
        const float *Pt = p3_of(ws, NET_CRITIC_T), *Pc = p3_of(ws, NET_CRITIC);
        float q2 = ws[TP_FB3 + 1], q = ws[TP_FB3 + 0];
#pragma unroll
        for (int i = 0; i < NT; ++i) { q2 += Pt[(i * 2) * BP + m]; q += Pc[(i * 2) * BP + m]; }
        const float y = ws[TP_R + m] + d.gamma * (1.0f - ws[TP_DONE + m]) * q2;            // DDPG.jl:133
        diff = m < d.batch ? q - y : 0.0f;
        dq = 2.0f * diff / (float)d.batch;                                                  // d mse / d q
    }
    d3s[t] = t < BP ? dq : 0.0f;
    if (publisher) {
        d.ws[TP_D3C + t] = t < BP ? dq : 0.0f;
        const float s1 = wave_sum64(diff * diff), s2 = wave_sum64(dq);
        if ((t & 63) == 0) { red[t >> 6] = s1; red[4 + (t >> 6)] = s2; }
        __syncthreads();
        if (t == 0) {
            d.losses[0] = (red[0] + red[1]) / (float)d.batch;                               // Flux.mse
            adam_at(c, off_b3(CIN, 1), red[4] + red[5], A.store_grad != 0);
        }
    }
}
__device__ __forceinline__ void head_actor(const NetArgs &A, const shems_ddpg &d, const AdamCtx &c, float *d3s, float *red, bool publisher)
{
    float *ws = d.ws;
    const int t = threadIdx.x, o = t >> 7, m = t & 127;
    float da = 0.0f;
#pragma unroll
    for (int p = 0; p < NT; ++p) da += ws[TP_DAP + (int64_t)(p * 2 + o) * BP + m];
    const float a = ws[TP_API + t];
    const float g = da * (1.0f - a * a);                       // through tanh
    d3s[t] = g;
    if (publisher) {
        ws[TP_D3A + t] = g;
        float q = 0.0f;
        if (o == 0 && m < d.batch) {
            const float *Pq = p3_of(ws, PASS_CRITIC2);
            q = d.critic[off_b3(CIN, 1)];
#pragma unroll
            for (int i = 0; i < NT; ++i) q += Pq[(i * 2) * BP + m];
        }
        const float sg = wave_sum64(g), sq = wave_sum64(q);
        if ((t & 63) == 0) { red[t >> 6] = sg; red[4 + (t >> 6)] = sq; }
        __syncthreads();
        if (t == 0) {
            d.losses[1] = -(red[4] + red[5]) / (float)d.batch;     // loss_act = -mean(critic(vcat(s, actor(s))))
            adam_at(c, off_b3(SIN, 2), red[0] + red[1], A.store_grad != 0);
            adam_at(c, off_b3(SIN, 2) + 1, red[2] + red[3], A.store_grad != 0);
        }
    }
}

// ================================================================================================================================
// P3 / P6: D1' [m][k] = mask1 .* sum_n D2[n][m] W2[k][n] for one 64-wide k-tile and the whole batch, D2[n][m] = (sum_o W3[n][o]
// d3[o][m]) (h2[n][m] > 0) generated from relu(layer 2) as it is loaded -- straight into MFMA operand layout, no LDS; the weights
// stream through an LDS ring in 32-deep n-chunks.  The product is computed TRANSPOSED (rows = samples, columns = hidden units): its
// D layout is then the B operand of the layer-1 gradient gW1[j][k] = sum_m x[j][m] D1[k][m], which follows on the matrix pipe without
// any LDS round trip (the bias gradient is the row of ones of the input block).  ADAM + soft update for the k-tile's layer-1 columns.
// ================================================================================================================================
#ifndef SHEMS_D1_PAD
#define SHEMS_D1_PAD 0      // (diagnostic builds: extra dynamic LDS = fewer resident workgroups, to read the launch's sensitivity to occupancy)
#endif
constexpr int D1_S = 33;
constexpr unsigned kNarrowBelow = 48;
// KT = 32-wide sub-tiles per workgroup: 2 = a 64-wide k-tile (four workgroups per learner -- the form for wide groups, fewest reads of
// relu(layer 2)); 1 = a 32-wide one (eight per learner: with few learners four do not fill the chip -- 128 workgroups at 32 learners)
constexpr int d1_ring(int KT) { return 2 * 32 * KT * D1_S > 4 * KT * 8 * 64 ? 2 * 32 * KT * D1_S : 4 * KT * 8 * 64; }      // floats: the ring, later the four waves' gW1 partials
constexpr int d1_lds(int KT) { return (d1_ring(KT) + 1024 + 2 * BP + 8 + W1K * BP + W1K * 32 * KT) * 4; }

template <int IN, int KT, bool TL>
__device__ __forceinline__ void d1_body(const NetArgs &A, const int bx, const int by, float *smem)
{
    typedef NetOf<IN> N;
    constexpr int OUT = N::OUT;
    constexpr int KW = 32 * KT;                  // k-tile width
    float *ring = smem;                          // [2][KW k][33]
    float *w3s = ring + d1_ring(KT);             // [512][2] frozen W3
    float *d3s = w3s + 1024;                     // [2][BP]
    float *red = d3s + 2 * BP;                   // [8]
    float *xs = red + 8;                         // [12][BP] the input block        } operands of what follows the chunk loop: staged by the
    float *wls = xs + W1K * BP;                  // [12][KW] layer-1 image, k-tile  } prologue's requests, no round trip after the loop
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int kt = bx, k0 = KW * kt;
    const int64_t off = (int64_t)by * A.gstride;
    shems_ddpg d = A.d;
    AdamCtx c = A.c;
    gshift(d, off); gshift(c, off);
    float *ws = d.ws;
    const float *__restrict__ P = c.p;
    const float *__restrict__ W2 = P + off_w2(IN);
    const float *__restrict__ H2 = N::H2(ws);
    const int m = 32 * w + li;

    // A chunk q: W2[k0 .. k0 + 63][32 q .. 32 q + 31] (128 B per row), two float4 per thread; rows >= 250 clamped (their outputs are
    // never stored), columns >= 500 of the last chunk meet D2 rows that are exactly zero
    // (tiled layout: the same 32 columns of rows k0 .. are 128-byte pieces 256 bytes apart inside tile (k0 / 64, q / 2); pad rows are zeros)
    const float *__restrict__ Wt = TL ? gsh(A.w2t, off) + TL_P * TL_TILE + (int64_t)(k0 >> 6) * 8 * TL_BLOCK + (k0 & 63) * 64 : nullptr;
    auto a_load = [&](int q, f32x4 (&v)[KT]) {
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            const int e = it * 256 + tid;
            if constexpr (TL) v[it] = *reinterpret_cast<const f32x4 *>(Wt + (int64_t)(q >> 1) * TL_BLOCK + (e >> 3) * 64 + 32 * (q & 1) + 4 * (e & 7));
            else {
                const int k = min(k0 + (e >> 3), H1N - 1);
                v[it] = *reinterpret_cast<const f32x4 *>(W2 + (int64_t)k * H2N + 32 * q + 4 * (e & 7));
            }
        }
    };
    auto a_store = [&](float *buf, const f32x4 (&v)[KT]) {
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            const int e = it * 256 + tid;
            float *p = buf + (e >> 3) * D1_S + 4 * (e & 7);
            p[0] = v[it][0]; p[1] = v[it][1]; p[2] = v[it][2]; p[3] = v[it][3];
        }
    };
    // B operand source of chunk q: h2[32 q + 2 s + lh][m]
    auto h_load = [&](int q, float (&h)[16]) {
#pragma unroll
        for (int s = 0; s < 16; ++s) h[s] = H2[(int64_t)(32 * q + 2 * s + lh) * BP + m];
    };
    f32x4 pv[KT];
    float hc[16], hn[16];
    a_load(0, pv);
    h_load(0, hc);
    const f32x4 w3v = reinterpret_cast<const f32x4 *>(N::w3f(ws))[tid];
    const float *w1i = N::w1i(ws);
    const float *X = N::X(ws);
    constexpr int WLN = (W1K * KW + 255) / 256;          // 3 (KT = 2) / 2 (KT = 1: 384 floats)
    float xv[6], wlv[WLN];
#pragma unroll
    for (int s = 0; s < 6; ++s) xv[s] = X[s * 256 + tid];
#pragma unroll
    for (int s = 0; s < WLN; ++s) { const int e = min(s * 256 + tid, W1K * KW - 1); wlv[s] = w1i[(e / KW) * W1C + k0 + e % KW]; }
    if (A.head == 1) head_critic(A, d, c, d3s, red, kt == 0); else head_actor(A, d, c, d3s, red, kt == 0);
    reinterpret_cast<f32x4 *>(w3s)[tid] = w3v;
#pragma unroll
    for (int s = 0; s < 6; ++s) xs[s * 256 + tid] = xv[s];
#pragma unroll
    for (int s = 0; s < WLN; ++s) if (s * 256 + tid < W1K * KW) wls[s * 256 + tid] = wlv[s];
    a_store(ring, pv);
    __syncthreads();
    const float d30 = d3s[m], d31 = d3s[BP + m];

    f32x16 acc[KT];
#pragma unroll
    for (int tt = 0; tt < KT; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tt][r] = 0.0f;
    // One n-chunk: the next chunk's weights and error-signal source are requested first and are not touched before the following step
    // (two register sets taken in turn: a copy `hc = hn` at the end of the step is moved up by the scheduler into the product, where its
    // wait for the just-issued loads -- loads retire in order -- stalls the matrix pipe every chunk).
    auto step = [&](auto prefetch, int q, const float (&hcur)[16], float (&hnext)[16]) {
        constexpr bool PF = decltype(prefetch)::value;
        const float *buf = ring + (q & 1) * KW * D1_S;
        // (never behind a run-time branch: the compiler's wait in front of hcur would be the one of the path WITHOUT new requests, which
        // on the other path waits for them; the last step is a separate instance instead)
        if constexpr (PF) { a_load(q + 1, pv); h_load(q + 1, hnext); }
        __builtin_amdgcn_sched_barrier(0);
        const float *pb = buf + li * D1_S + lh;
        const float *pw = w3s + 2 * (32 * q + lh);
        // (the LDS operands of k-step s + 1 are read before the products of step s are issued: read -> wait -> products leaves the LDS
        // latency uncovered behind every pair)
        float2 wA = *reinterpret_cast<const float2 *>(pw), wB;
        constexpr int T1 = KT == 2 ? 32 * D1_S : 0;          // second sub-tile's rows (KT = 1: the same operand, unused)
        float bA0 = pb[0], bA1 = pb[T1], bB0, bB1;
        auto prod = [&](int s, const float2 w3, float b0, float b1) {
            const float gsum = OUT == 2 ? fmaf(w3.y, d31, w3.x * d30) : w3.x * d30;
            const float d2 = hcur[s] > 0.0f ? gsum : 0.0f;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d2, b0, acc[0], 0, 0, 0);
            if constexpr (KT == 2) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d2, b1, acc[1], 0, 0, 0);
        };
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
            wB = *reinterpret_cast<const float2 *>(pw + 4 * (s + 1)); bB0 = pb[2 * (s + 1)]; bB1 = pb[T1 + 2 * (s + 1)];
            __builtin_amdgcn_sched_barrier(0);
            prod(s, wA, bA0, bA1);
            if (s < 14) { wA = *reinterpret_cast<const float2 *>(pw + 4 * (s + 2)); bA0 = pb[2 * (s + 2)]; bA1 = pb[T1 + 2 * (s + 2)]; }
            __builtin_amdgcn_sched_barrier(0);
            prod(s + 1, wB, bB0, bB1);
        }
        if constexpr (PF) a_store(ring + ((q + 1) & 1) * KW * D1_S, pv);
        __syncthreads();
    };
#pragma unroll 1
    for (int q = 0; q < 14; q += 2) {
        step(std::true_type{}, q, hc, hn);
        step(std::true_type{}, q + 1, hn, hc);
    }
    step(std::true_type{}, 14, hc, hn);
    step(std::false_type{}, 15, hn, hc);
    // ---- this thread's four layer-1 elements: their ADAM state is requested now and arrives under the products below ----
    constexpr int NE = 2 * KT;                   // elements per thread: [KT tt][8 r][64 lanes] over 256 threads
    int ei[NE];
    float em[NE], ev[NE], epp[NE], et[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int sidx = u * 256 + tid, ln = sidx & 63, rr = (sidx >> 6) & 7, tt = sidx >> 9;
        const int j = drow(rr, ln >> 5), k = k0 + 32 * tt + (ln & 31);
        ei[u] = (k < H1N && (j < IN || j == W1K - 1)) ? (j == W1K - 1 ? off_b1(IN) + k : j * H1N + k) : -1;
        const int e = max(ei[u], 0);
        em[u] = c.mt[e]; ev[u] = c.vt[e]; epp[u] = c.p[e]; et[u] = c.target[e];
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- layer-1 relu mask: pre-activations transposed, rows = samples of this wave's column tile, columns = the k-tile's units ----
    float xa[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) xa[s] = xs[(2 * s + lh) * BP + m];
    // A operands of the layer-1 gradient: x[j = li][32 w + drow(r, lh)] (rows >= 12 zero)
    f32x4 xq[4];
    {
        const float keep = li < W1K ? 1.0f : 0.0f;
        const float *px = xs + min(li, W1K - 1) * BP + 32 * w + 4 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q) xq[q] = *reinterpret_cast<const f32x4 *>(px + 8 * q) * keep;
    }
    float *redp = ring;                          // [4 waves][KT tt][8 r][64 lanes]  (the ring is free: the loop ended with a barrier)
#pragma unroll
    for (int tt = 0; tt < KT; ++tt) {
        float wb[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) wb[s] = wls[(2 * s + lh) * KW + 32 * tt + li];
        f32x16 t;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 6; ++s) t = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[s], wb[s], t, 0, 0, 0);
        f32x16 g;
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float dv = t[r] > 0.0f ? acc[tt][r] : 0.0f;                 // D1'[m = drow(r, lh)][k = li]
            g = __builtin_amdgcn_mfma_f32_32x32x2f32(xq[r >> 2][r & 3], dv, g, 0, 0, 0);
        }
        // rows j = drow(r, lh) < 12: r = 0..3 (both halves), r = 4..7 (lh = 0)
#pragma unroll
        for (int r = 0; r < 8; ++r) redp[((w * KT + tt) * 8 + r) * 64 + lane] = g[r];
    }
    __syncthreads();
    // (consumed unconditionally: left to itself the compiler moves the requests above behind the `ei >= 0` that guards the stores)
#pragma unroll
    for (int u = 0; u < NE; ++u) asm volatile("" : "+v"(em[u]), "+v"(ev[u]), "+v"(epp[u]), "+v"(et[u]));
    float *gP = const_cast<float *>(c.g);
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int sidx = u * 256 + tid, ln = sidx & 63, rr = (sidx >> 6) & 7, tt = sidx >> 9;
        const float *pr = redp + (tt * 8 + rr) * 64 + ln;
        const float gsum = ((pr[0] + pr[KT * 8 * 64]) + pr[2 * KT * 8 * 64]) + pr[3 * KT * 8 * 64];
        adam_math(c, gsum, em[u], ev[u], epp[u], et[u]);
        if (ei[u] >= 0) {
            const int e = ei[u];
            c.mt[e] = em[u]; c.vt[e] = ev[u]; c.p[e] = epp[u]; c.target[e] = et[u];
            if (A.store_grad != 0) gP[e] = gsum;
            if constexpr (N::critic) {           // P5 runs the UPDATED critic: its layer-1 image, element by element (zeros from P0)
                const int j = drow(rr, ln >> 5), k = k0 + 32 * tt + (ln & 31);
                w1i_of(ws, IMG_CRITIC_NEW)[j * W1C + k] = epp[u];
            }
        }
    }
}

// ================================================================================================================================
// P4 / P7: gW2[k][n] = sum_m h1[k][m] D2[n][m] for one 64 x 64 tile, ADAM + soft target update on it; the k-tile-0 workgroups also
// finish gb2[n] = sum_m D2[n][m] and gW3[n][o] = sum_m h2[n][m] d3[o][m] of their n-tile.  h1' (samples x units) is recomputed on the
// matrix pipe per 32-sample block -- its D layout is the A operand; D2 goes through LDS.
// This phase is the update's HBM stream: 32 B in and out per parameter of W2 (moments, parameter, target) against 64 FLOP.  What it
// reaches depends on how the four arrays are touched and on how many bytes a CU keeps in flight (measured with the matrix work switched
// off, 400 learners, 1.65 GB per launch; profiles/r05_gw2_stream_forms.txt): 4-byte accesses in the accumulator's layout (128-byte row
// pieces) 3.85 TB/s; 16-byte accesses in row-major order of the tile, 256-byte pieces 4.1 TB/s, 1 KB pieces 4.7 TB/s = the rate of a
// plain elementwise ADAM sweep (a device copy of the same bytes: 5.5 TB/s) -- but only with ~256 KB in flight per CU: a 32 x 256 tile
// whose state is requested in quarters (to fit the registers beside the product) fell to 1.9 TB/s.  So: the whole tile's state is
// requested FIRST, 16 bytes per lane in row-major order (thread -> four consecutive columns of one row), the product runs under those
// loads, and the finished tile changes hands through LDS (accumulator layout -> rows) before ADAM.
// ================================================================================================================================
constexpr int GW_S = BP + 1;
constexpr int GW_TS = 68;          // row stride of the row-major copy of the finished tile [64 k][68] (over the D2 panel)
constexpr int GW_WGS = 4 * NT;     // k-tiles x n-tiles of 64 per learner
// (Round 6 also built a four-workgroups-per-CU form -- the error signal from registers instead of LDS: 40 448 B, and a 128-register budget, which
// spills 12 registers -- no gain at 400 learners (1.926-1.938 against 1.927-1.935 ms per step), 3 % slower at 32 (profiles/r06_gw2_ab.txt); removed.)
constexpr int GW_LDS = (64 * GW_S + 2 * BP + 64 * 2 + 64 * 3 + W1K * BP) * 4;
static_assert(64 * GW_TS <= 64 * GW_S, "the row-major copy of a tile fits the D2 panel it replaces");

template <int IN, bool TL>
__device__ __forceinline__ void gw2_body(const NetArgs &A, const int bx, const int by, float *smem)
{
    typedef NetOf<IN> N;
    constexpr int OUT = N::OUT;
    float *d2s = smem;                           // [64 n][129]; later the tile [64 k][68]
    float *d3s = d2s + 64 * GW_S;                // [2][BP]
    float *w3s = d3s + 2 * BP;                   // [64][2]
    float *rs = w3s + 64 * 2;                    // [64][3] row sums: gb2, gW3[.][0], gW3[.][1]
    float *xs = rs + 64 * 3;                     // [12][BP] the input block
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int nt = bx & 7, kt = bx >> 3, n0 = 64 * nt, k0 = 64 * kt;
    const int kw = w >> 1, nw = w & 1;
    const int64_t off = (int64_t)by * A.gstride;
    shems_ddpg d = A.d;
    AdamCtx c = A.c;
    gshift(d, off); gshift(c, off);
    float *ws = d.ws;
    const float *__restrict__ H2 = N::H2(ws);
    const bool store_grad = A.store_grad != 0;

    // Requests, in the order the workgroup needs them: loads retire in order, so a wait for any operand also waits for everything requested
    // before it.  The small operands of the panel and of the product come first; the tile's state -- 16 float4, the kernel's HBM stream --
    // is requested LAST and is still arriving while the panel is built and the product runs (nothing later in the kernel issues a load).
    f32x4 hv[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int e = it * 256 + tid;
        hv[it] = *reinterpret_cast<const f32x4 *>(H2 + (int64_t)(n0 + (e >> 5)) * BP + 4 * (e & 31));
    }
    const float d3v = N::d3(ws)[tid];
    const float w3v = N::w3f(ws)[2 * n0 + (tid & 127)];
    const float *w1i = N::w1i(ws);
    float wb[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) wb[s] = w1i[(2 * s + lh) * W1C + k0 + 32 * kw + li];
    const float *X = N::X(ws);
    float xv[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) xv[s] = X[s * 256 + tid];
    __builtin_amdgcn_sched_barrier(0);
    // this thread's 16 elements of the tile, row-major: float4 it = row 16 it + (tid >> 4), columns 4 (tid & 15) .. + 3
    // Tiled layout: the same float4 (row 16 it + tid / 16, columns 4 (tid % 16) ..) is float4 number 256 it + tid of each of the tile's
    // four arrays, which lie one behind the other: the workgroup's whole state is ONE contiguous 64 KB piece, read here, written below.
    int eidx[4];
    f32x4 am[4], av[4], ap[4], at[4];
    float *tile = TL ? gsh(A.w2t, off) + tl_tile(kt, nt) + 4 * tid : nullptr;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int k = k0 + 16 * it + (tid >> 4), n = n0 + 4 * (tid & 15);
        eidx[it] = (k < H1N && n < H2N) ? off_w2(IN) + k * H2N + n : -1;          // (500 is a multiple of 4: a float4 is all in or all out)
        if constexpr (TL) {
            // (non-temporal: a tile's state is next touched one whole grouped update -- 4 GB of other traffic -- later; on the 64 KB
            // pieces the hint is worth 4.90 -> 5.13 TB/s, a device copy's rate; on the Flux order's 256-byte pieces it costs a third:
            // tools/micro/adam_stream.hip, profiles/r06_micro_adam_stream.txt)
            const float *q = tile + 1024 * it;
            am[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(q + TL_M * TL_TILE));
            av[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(q + TL_V * TL_TILE));
            ap[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(q + TL_P * TL_TILE));
            at[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(q + TL_T * TL_TILE));
        } else {
            const int e = off_w2(IN) + min(k, H1N - 1) * H2N + min(n, H2N - 4);
            am[it] = *reinterpret_cast<const f32x4 *>(c.mt + e); av[it] = *reinterpret_cast<const f32x4 *>(c.vt + e);
            ap[it] = *reinterpret_cast<const f32x4 *>(c.p + e); at[it] = *reinterpret_cast<const f32x4 *>(c.target + e);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    d3s[tid] = d3v;
    if (tid < 128) w3s[tid] = w3v;
#pragma unroll
    for (int s = 0; s < 6; ++s) xs[s * 256 + tid] = xv[s];
    __syncthreads();
    // D2 panel + row sums
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int e = it * 256 + tid, nl = e >> 5, m4 = 4 * (e & 31);
        const float2 w3 = *reinterpret_cast<const float2 *>(w3s + 2 * nl);
        float sb = 0.0f, s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float h = hv[it][i], e0 = d3s[m4 + i], e1 = d3s[BP + m4 + i];
            const float gsum = OUT == 2 ? fmaf(w3.y, e1, w3.x * e0) : w3.x * e0;
            const float d2 = h > 0.0f ? gsum : 0.0f;
            d2s[nl * GW_S + m4 + i] = d2;
            sb += d2; s0 = fmaf(h, e0, s0); s1 = fmaf(h, e1, s1);
        }
        if (kt == 0) {                           // (workgroup-uniform)
            sb = half_sum32(sb); s0 = half_sum32(s0); s1 = half_sum32(s1);
            if (li == 0) { rs[nl * 3 + 0] = sb; rs[nl * 3 + 1] = s0; rs[nl * 3 + 2] = s1; }
        }
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const float *pb = d2s + (32 * nw + li) * GW_S + 4 * lh;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float xa[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) xa[s] = xs[(2 * s + lh) * BP + 32 * mt + li];
        f32x16 t;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 6; ++s) t = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[s], wb[s], t, 0, 0, 0);       // h1'[m][k] pre-activations
        // (B operand of k-step r + 1 read from LDS before the product of step r is issued)
        float bA = pb[32 * mt], bB;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            bB = pb[32 * mt + ((r + 1) & 3) + 8 * ((r + 1) >> 2)];
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fmaxf(t[r], 0.0f), bA, acc, 0, 0, 0);
            if (r < 14) bA = pb[32 * mt + ((r + 2) & 3) + 8 * ((r + 2) >> 2)];
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fmaxf(t[r + 1], 0.0f), bB, acc, 0, 0, 0);
        }
    }
    // the tile changes hands: accumulator layout -> LDS [64 k][68] (over the D2 panel, which every wave has finished reading) -> rows
    __syncthreads();
    float *T = d2s;
#pragma unroll
    for (int r = 0; r < 16; ++r) T[(32 * kw + drow(r, lh)) * GW_TS + 32 * nw + li] = acc[r];
    __syncthreads();
    {
        float *gW = const_cast<float *>(c.g);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const f32x4 g4 = *reinterpret_cast<const f32x4 *>(T + (16 * it + (tid >> 4)) * GW_TS + 4 * (tid & 15));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float m_ = am[it][i], v_ = av[it][i], p_ = ap[it][i], t_ = at[it][i];
                adam_math(c, g4[i], m_, v_, p_, t_);
                am[it][i] = m_; av[it][i] = v_; ap[it][i] = p_; at[it][i] = t_;
            }
            if (eidx[it] >= 0) {                 // (pad rows / columns of the tiled layout stay zero: never written)
                const int e = eidx[it];
                if constexpr (TL) {
                    float *q = tile + 1024 * it;
                    __builtin_nontemporal_store(am[it], reinterpret_cast<f32x4 *>(q + TL_M * TL_TILE));
                    __builtin_nontemporal_store(av[it], reinterpret_cast<f32x4 *>(q + TL_V * TL_TILE));
                    __builtin_nontemporal_store(ap[it], reinterpret_cast<f32x4 *>(q + TL_P * TL_TILE));
                    __builtin_nontemporal_store(at[it], reinterpret_cast<f32x4 *>(q + TL_T * TL_TILE));
                } else {
                    *reinterpret_cast<f32x4 *>(c.mt + e) = am[it]; *reinterpret_cast<f32x4 *>(c.vt + e) = av[it];
                    *reinterpret_cast<f32x4 *>(c.p + e) = ap[it]; *reinterpret_cast<f32x4 *>(c.target + e) = at[it];
                }
                if (store_grad) *reinterpret_cast<f32x4 *>(gW + e) = g4;
            }
        }
    }
    if (kt == 0 && tid < 64 * 3) {               // one element per thread: (row, b2 | W3[.][0] | W3[.][1])
        const int nl = tid / 3, col = tid - nl * 3, n = n0 + nl;
        if (n < H2N && col - 1 < OUT) adam_at(c, col == 0 ? off_b2(IN) + n : off_w3(IN) + n * OUT + (col - 1), rs[tid], store_grad);
    }
}

// ---- one launch per phase (the plain form: every learner of the group in the same phase) ---------------------------------------
__global__ __launch_bounds__(256) void k_tp_prep(PrepArgs A) { prep_body(A, blockIdx.x, blockIdx.y); }
template <bool QG, int NTL, bool TL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NTL == 4 ? 3 : 4, 4))) void k_tp_fwd(FwdArgs A)      // (the wide form's LDS allows three workgroups per CU)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    fwd_body<QG, NTL, TL>(A, blockIdx.x, blockIdx.y, smem);
}
// (Round 6 also built the forward launches as PERSISTENT workgroups -- 768 / 1 024 of them drawing their (n-tile, learner) items from a counter, which
// tools/micro/fwd_anatomy.hip rates 0.86 against 0.82 of the peak for loops without prologues -- verified against the float64 oracle, measured slower
// with the real prologues and epilogues (P1 373.5 against 355.3 us, P2 143.0 / 133.6, P5 278.9 / 273.8; 204.4 against 207.6 k updates/s), removed:
// profiles/r06_fwd_persistent_ab.txt.)
template <int IN, int KT, bool TL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_tp_d1(NetArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // A learner's k-tile workgroups (T = 8 / KT of them) read the same 256 KB of relu(layer 2): they are placed on ONE XCD (workgroup id
    // mod 8 picks the XCD and its L2), consecutive in its dispatch order -- learner = 8 (q / T) + xcd, k-tile = q mod T with q = id / 8.
    constexpr unsigned T = 8 / KT;
    const unsigned id = blockIdx.x, xcd = id & 7u, q = id >> 3;
    const unsigned learner = 8u * (q / T) + xcd;
    if (learner >= (unsigned)A.learners) return;
    d1_body<IN, KT, TL>(A, (int)(q % T), (int)learner, smem);
}
template <int IN, bool TL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_tp_gw2(NetArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gw2_body<IN, TL>(A, blockIdx.x, blockIdx.y, smem);
}

// (Round 5 also built two ways of running the HBM-bound phases (P4, P7) under the MFMA-bound ones (P1, P2, P5) of OTHER learners: cohorts
// of learners on separate streams, and a launch that runs different phases for different cohorts with their workgroups interleaved.
// Both verified, both measured slower than the plain sequence above (2.11-3.27 ms and 2.17-2.34 ms against 2.08-2.11 ms), both removed:
// profiles/r05_tp_overlap_probe.json.)

// ================================================================================================================================
// Flux order <-> tiled layout of the two networks' layer-2 state (W2, m, v, target): one workgroup per (tile, network, learner).
// TO_TILED writes the pad rows / columns as zeros (the kernels above rely on that and never write them); TO_FLUX skips them.
// ================================================================================================================================
struct LayoutArgs { shems_ddpg d; float *w2t_actor, *w2t_critic; int64_t gstride; };
template <bool TO_TILED>
__global__ __launch_bounds__(256) void k_tp_w2_layout(LayoutArgs A)
{
    const int tid = threadIdx.x, nt = blockIdx.x & 7, kt = blockIdx.x >> 3;
    const bool critic = blockIdx.y == 1;
    const int64_t off = (int64_t)blockIdx.z * A.gstride;
    shems_ddpg d = A.d;
    gshift(d, off);
    float *flux[4] = {critic ? d.m_critic : d.m_actor, critic ? d.v_critic : d.v_actor, critic ? d.critic : d.actor, critic ? d.critic_t : d.actor_t};
    float *tile = gsh(critic ? A.w2t_critic : A.w2t_actor, off) + tl_tile(kt, nt) + 4 * tid;
    const int w2 = off_w2(critic ? CIN : SIN);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int k = 64 * kt + 16 * it + (tid >> 4), n = 64 * nt + 4 * (tid & 15);
        const bool in = k < H1N && n < H2N;
        const int e = w2 + min(k, H1N - 1) * H2N + min(n, H2N - 4);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            f32x4 *t4 = reinterpret_cast<f32x4 *>(tile + 1024 * it + a * TL_TILE), *f4 = reinterpret_cast<f32x4 *>(flux[a] + e);
            if constexpr (TO_TILED) *t4 = in ? *f4 : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            else if (in) *f4 = *t4;
        }
    }
}

}  // namespace tp
}  // namespace shems

using namespace shems;
using namespace shems::tp;

static int check_w2t(const shems_group_w2t *t, const char *fn)
{
    if (!t || !t->actor || !t->critic) return set_error(SHEMS_ERR_ARG, "%s: shems_group_w2t needs both regions", fn);
    if ((((uintptr_t)t->actor | (uintptr_t)t->critic) & 15) != 0) return set_error(SHEMS_ERR_ARG, "%s: the tiled regions must be 16-byte aligned", fn);
    return SHEMS_OK;
}
static int check_group_tp(const shems_group *g, const char *fn)
{
    if (!g || g->count < 1 || g->count > 65535 || g->stride_bytes < 0 || (g->stride_bytes & 15) != 0 || (g->count > 1 && g->stride_bytes == 0))
        return set_error(SHEMS_ERR_ARG, "%s: shems_group needs 1 <= count <= 65535 and a 16-byte-multiple stride", fn);
    return SHEMS_OK;
}

// t == null: every array in Flux order (round 5's form); else the layer-2 state of both networks lives in the tiled regions.
template <bool TL>
static int group_update_tp(const char *fn, const shems_ddpg *d, const shems_replay *ring, const shems_group *g, const shems_group_w2t *t, int64_t ring_len,
                           uint64_t seed, uint32_t tick, double eta_crit, double bp1_crit, double bp2_crit, double eta_act, double bp1_act, double bp2_act,
                           int32_t flags, void *stream)
{
    if (int rc = check_group_tp(g, fn)) return rc;
    if (!d || !d->actor || !d->critic || !d->actor_t || !d->critic_t || !d->m_actor || !d->v_actor || !d->m_critic || !d->v_critic ||
        !d->s_min || !d->s_max || !d->ws || !d->losses)
        return set_error(SHEMS_ERR_ARG, "%s: shems_ddpg has a NULL buffer", fn);
    if ((flags & SHEMS_TP_STORE_GRAD) && (!d->grad_actor || !d->grad_critic)) return set_error(SHEMS_ERR_ARG, "%s: STORE_GRAD needs gradient buffers", fn);
    if (flags & ~SHEMS_TP_STORE_GRAD) return set_error(SHEMS_ERR_ARG, "%s: unknown flag bits", fn);
    if (d->batch < 1 || d->batch > BP) return set_error(SHEMS_ERR_ARG, "%s: batch must be in 1..128 (got %d)", fn, d->batch);
    for (const float *p : {(const float *)d->actor, (const float *)d->critic, (const float *)d->actor_t, (const float *)d->critic_t, (const float *)d->ws})
        if (((uintptr_t)p & 15) != 0) return set_error(SHEMS_ERR_ARG, "%s: parameter blocks and the workspace must be 16-byte aligned", fn);
    if (!ring || !ring->s || !ring->a || !ring->r || !ring->s2 || !ring->done || ring_len < 1 || ring_len > ring->capacity)
        return set_error(SHEMS_ERR_ARG, "%s: bad replay ring / length", fn);
    for (double bp : {bp1_crit, bp2_crit, bp1_act, bp2_act})
        if (!(bp > 0.0 && bp < 1.0)) return set_error(SHEMS_ERR_ARG, "%s: beta powers must be in (0,1)", fn);
    if (TL) if (int rc = check_w2t(t, fn)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const unsigned L = (unsigned)g->count;
    const int64_t gs = g->count > 1 ? g->stride_bytes : 0;
    const int sg = (flags & SHEMS_TP_STORE_GRAD) ? 1 : 0;
    float *ws = d->ws;
    float *ta = TL ? t->actor : nullptr, *tc = TL ? t->critic : nullptr;
    auto arr = [](float *region, int a) -> const float * { return region ? region + a * TL_TILE : nullptr; };

    auto adam_ctx = [&](bool critic) {
        const double eta = critic ? eta_crit : eta_act, bp1 = critic ? bp1_crit : bp1_act, bp2 = critic ? bp2_crit : bp2_act;
        return critic ? AdamCtx{d->critic, d->grad_critic, d->m_critic, d->v_critic, d->critic_t, nullptr, SHEMS_CRITIC_PARAMS, (int)CIN,
                                eta, bp1, bp2, 1.0, eta / (1.0 - bp1), 1.0 / (1.0 - bp2), d->tau}
                      : AdamCtx{d->actor, d->grad_actor, d->m_actor, d->v_actor, d->actor_t, nullptr, SHEMS_ACTOR_PARAMS, (int)SIN,
                                eta, bp1, bp2, 1.0, eta / (1.0 - bp1), 1.0 / (1.0 - bp2), d->tau};
    };
    struct { PrepArgs pa; FwdArgs f1, f2, f5; NetArgs nc, na; } U;
    std::memset(&U, 0, sizeof U);
    U.pa = PrepArgs{*d, *ring, ring_len, gs, seed, tick};
    for (FwdArgs *f : {&U.f1, &U.f2, &U.f5}) { f->gstride = gs; f->batch = d->batch; }
    // P1: three independent forward passes
    U.f1.job[0] = FwdJob{arr(ta, TL_T), d->actor_t, ws + TP_X2, w1i_of(ws, NET_ACTOR_T), nullptr, nullptr, nullptr, nullptr, p3_of(ws, NET_ACTOR_T), nullptr, SIN, 2};
    U.f1.job[1] = FwdJob{arr(tc, TL_P), d->critic, ws + TP_X, w1i_of(ws, NET_CRITIC), nullptr, nullptr, nullptr, ws + TP_H2C, p3_of(ws, NET_CRITIC), nullptr, CIN, 1};
    U.f1.job[2] = FwdJob{arr(ta, TL_P), d->actor, ws + TP_X, w1i_of(ws, NET_ACTOR), nullptr, nullptr, nullptr, ws + TP_H2A, p3_of(ws, NET_ACTOR), nullptr, SIN, 2};
    // P2: critic_target on [s'; actor_target(s')]
    U.f2.job[0] = FwdJob{arr(tc, TL_T), d->critic_t, ws + TP_X2, w1i_of(ws, NET_CRITIC_T), p3_of(ws, NET_ACTOR_T), ws + TP_FB3 + 4, nullptr, nullptr,
                         p3_of(ws, NET_CRITIC_T), nullptr, CIN, 1};
    // P5: updated critic on [s; actor(s)], forward + input gradient
    U.f5.job[0] = FwdJob{arr(tc, TL_P), d->critic, ws + TP_X, w1i_of(ws, IMG_CRITIC_NEW), p3_of(ws, NET_ACTOR), ws + TP_FB3 + 2, ws + TP_API, nullptr, p3_of(ws, PASS_CRITIC2),
                         ws + TP_DAP, CIN, 1};
    U.nc = NetArgs{*d, tc, adam_ctx(true), gs, 1, sg, (int)L};
    U.na = NetArgs{*d, ta, adam_ctx(false), gs, 2, sg, (int)L};
    hipLaunchKernelGGL(k_tp_prep, dim3(5, L), dim3(256), 0, st, U.pa);
    typedef FwdShape<false, 4> SW;           // (typedefs: the launch macro splits its arguments at the commas of a template argument list)
    typedef FwdShape<false, 2> SN;
    typedef FwdShape<true, 2> SQ;
    // Few learners: the shapes with twice the workgroups (P1 on 64-wide n-tiles, P3 / P6 on 32-wide k-tiles).  Below kNarrowBelow learners
    // the wide shapes leave CUs without work (P3 at 32 learners: 128 workgroups); measured per grouped update, wide / narrow: 32 learners
    // 247 / 226 us, 48 learners 322 / 320, 64 learners 362 / 372, 128 learners 669 / 695 (profiles/NOTES.md, round-5 log).
    const bool narrow = L < kNarrowBelow;
    if (narrow) hipLaunchKernelGGL((k_tp_fwd<false, 2, TL>), dim3(3 * SN::TILES, L), dim3(256), SN::LDS, st, U.f1);
    else hipLaunchKernelGGL((k_tp_fwd<false, 4, TL>), dim3(3 * SW::TILES, L), dim3(256), SW::LDS, st, U.f1);
    hipLaunchKernelGGL((k_tp_fwd<false, 2, TL>), dim3(SN::TILES, L), dim3(256), SN::LDS, st, U.f2);
    const unsigned g8 = 8 * ((L + 7) / 8);
    if (narrow) hipLaunchKernelGGL((k_tp_d1<CIN, 1, TL>), dim3(8 * g8), dim3(256), d1_lds(1), st, U.nc);
    else hipLaunchKernelGGL((k_tp_d1<CIN, 2, TL>), dim3(4 * g8), dim3(256), d1_lds(2) + SHEMS_D1_PAD, st, U.nc);
    hipLaunchKernelGGL((k_tp_gw2<CIN, TL>), dim3(GW_WGS, L), dim3(256), GW_LDS, st, U.nc);
    hipLaunchKernelGGL((k_tp_fwd<true, 2, TL>), dim3(SQ::TILES, L), dim3(256), SQ::LDS, st, U.f5);
    if (narrow) hipLaunchKernelGGL((k_tp_d1<SIN, 1, TL>), dim3(8 * g8), dim3(256), d1_lds(1), st, U.na);
    else hipLaunchKernelGGL((k_tp_d1<SIN, 2, TL>), dim3(4 * g8), dim3(256), d1_lds(2) + SHEMS_D1_PAD, st, U.na);
    hipLaunchKernelGGL((k_tp_gw2<SIN, TL>), dim3(GW_WGS, L), dim3(256), GW_LDS, st, U.na);
    return hip_ok(hipGetLastError(), "grouped update (throughput form) launches");
}

extern "C" int shems_ddpg_group_update_tp(const shems_ddpg *d, const shems_replay *ring, const shems_group *g, int64_t ring_len, uint64_t seed,
                                          uint32_t tick, double eta_crit, double bp1_crit, double bp2_crit, double eta_act, double bp1_act,
                                          double bp2_act, int32_t flags, void *stream)
{
    return group_update_tp<false>("shems_ddpg_group_update_tp", d, ring, g, nullptr, ring_len, seed, tick, eta_crit, bp1_crit, bp2_crit, eta_act, bp1_act, bp2_act,
                                  flags, stream);
}

extern "C" int shems_ddpg_group_update_tiled(const shems_ddpg *d, const shems_replay *ring, const shems_group *g, const shems_group_w2t *t, int64_t ring_len,
                                             uint64_t seed, uint32_t tick, double eta_crit, double bp1_crit, double bp2_crit, double eta_act, double bp1_act,
                                             double bp2_act, int32_t flags, void *stream)
{
    return group_update_tp<true>("shems_ddpg_group_update_tiled", d, ring, g, t, ring_len, seed, tick, eta_crit, bp1_crit, bp2_crit, eta_act, bp1_act, bp2_act,
                                 flags, stream);
}

static int w2_layout(const char *fn, bool to_tiled, const shems_ddpg *d, const shems_group *g, const shems_group_w2t *t, void *stream)
{
    if (int rc = check_group_tp(g, fn)) return rc;
    if (int rc = check_w2t(t, fn)) return rc;
    if (!d || !d->actor || !d->critic || !d->actor_t || !d->critic_t || !d->m_actor || !d->v_actor || !d->m_critic || !d->v_critic)
        return set_error(SHEMS_ERR_ARG, "%s: shems_ddpg has a NULL network / moment buffer", fn);
    for (const float *p : {(const float *)d->actor, (const float *)d->critic, (const float *)d->actor_t, (const float *)d->critic_t, (const float *)d->m_actor,
                           (const float *)d->v_actor, (const float *)d->m_critic, (const float *)d->v_critic})
        if (((uintptr_t)p & 15) != 0) return set_error(SHEMS_ERR_ARG, "%s: parameter and moment blocks must be 16-byte aligned", fn);
    LayoutArgs A{*d, t->actor, t->critic, g->count > 1 ? g->stride_bytes : 0};
    const dim3 grid(32, 2, (unsigned)g->count);
    if (to_tiled) hipLaunchKernelGGL(k_tp_w2_layout<true>, grid, dim3(256), 0, (hipStream_t)stream, A);
    else hipLaunchKernelGGL(k_tp_w2_layout<false>, grid, dim3(256), 0, (hipStream_t)stream, A);
    return hip_ok(hipGetLastError(), fn);
}
extern "C" int shems_group_w2_to_tiled(const shems_ddpg *d0, const shems_group *g, const shems_group_w2t *t, void *stream)
{
    return w2_layout("shems_group_w2_to_tiled", true, d0, g, t, stream);
}
extern "C" int shems_group_w2_to_flux(const shems_ddpg *d0, const shems_group *g, const shems_group_w2t *t, void *stream)
{
    return w2_layout("shems_group_w2_to_flux", false, d0, g, t, stream);
}
