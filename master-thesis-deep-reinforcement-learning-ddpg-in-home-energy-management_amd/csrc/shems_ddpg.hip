// shems_ddpg.hip -- one DDPG update (the reference's replay(), DDPG.jl:121-145) as FIVE dependent gfx950 launches:
// GPU-resident minibatch sampling/gather, target pass, critic forward/backward, actor forward/backward through the
// critic, Flux-style ADAM and the soft target updates.
//
// Shapes: BATCH = 120 padded to BP = 128 columns (pad columns carry zero error signals); everything is FEATURE-major
// "[k][m]" (sample index contiguous) as in shems_policy.hip, so the 250x500 layer runs on v_mfma_f32_32x32x2_f32 with the
// weights as the A operand straight out of Flux's [in][out] layout.  At batch 120 one update is 307.8 MFLOP (~2 us at the
// fp32 MFMA peak): it is bound by dependent-launch boundaries and by L2/Infinity-Cache latency, not by the matrix pipe, so
// the design minimises the NUMBER of grid-wide dependencies (round 1 had 8 launches; this has 5):
//
//   K1  k_fwd (opens the update: every tile workgroup samples / gathers / normalises the minibatch columns it needs)
//         actor_target(s') | critic(s, a) | actor(s)                              three independent forward passes
//   K2  k_mid   critic_target(s', a')  |  E_c = (W3c .* mask2c) W2c^T  |  E_a0, E_a1 = (W3a[.,j] .* mask2a) W2a^T
//   K3  k_grad  critic: every gradient block from batch contractions only + ADAM + soft update IN THE SAME workgroup
//   K4  k_fwd<QG> critic(s, actor(s)) with the updated critic: forward AND input gradient fused per n-tile
//   K5  k_grad  actor: as K3
//
// Two identities remove the cross-workgroup reductions that used to force extra launches:
//   * layer-1 error through a network with output width o:  D1[k][m] = mask1[k][m] * sum_j d3[j][m] * E_j[k][m] with
//     E_j[k][m] = sum_n W2[k][n] W3[n][j] mask2[n][m], which does NOT depend on the error signal d3 -- it is computed in K2,
//     next to the target critic's forward pass, as soon as the forward masks exist.  The gradient launches K3 / K5 then
//     contract over the BATCH only, so each gradient tile is complete inside one workgroup and that workgroup applies ADAM
//     and the soft target update to exactly the elements it produced (no ADAM launch, no gradient round trip);
//   * the actor loss -mean(q) has a constant upstream gradient (-1/B), so the critic's backward pass on [s; actor(s)] needs
//     no finished forward pass: each n-tile workgroup of K4 back-propagates through its own 32 hidden units right after
//     computing them (same W2 panel, still in LDS) and emits a partial d loss / d a; K5's prologue adds the 16 partials.
// In place updates are race free because no launch reads a parameter another workgroup of the same launch writes: K3 / K5
// take W3 / b3 from a frozen copy K1 puts in the workspace, layer 1 from the packed images K1 builds, and nothing else of
// the network being updated.
// When replicas exchange gradients (data parallel) K3 / K5 only store the gradient and a k_adam_soft sweep follows the
// all-reduce: 7 launches.  Both forms give the same bits (same gradient code, same ADAM function).
// Other rules kept from round 1: operands staged into LDS with wide independent loads, ALL of a phase's global loads issued
// before the first wait, clamped and never predicated; partial slabs summed in a fixed order (bitwise reproducible; no float
// atomics, no device-scope fences).
#include <hip/hip_runtime.h>

#include <cstring>
#include <type_traits>

#include "philox.h"
#include "shems_internal.h"
#include "shems_adam.h"

namespace shems {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Diagnostic build only (-DSHEMS_STAMP, tools/stamp_update.py): thread 0 of every workgroup records (s_memtime, s_memrealtime) at the
// phase boundaries into a buffer of its own -- no product code reads it, the product library contains none of this.
#ifdef SHEMS_STAMP
__device__ unsigned long long *g_stamps = nullptr;             // [5 launches][1024 workgroups][16 stamps][2]
#define STAMP(region, i) do { if (threadIdx.x == 0 && g_stamps) { unsigned long long *sp_ = g_stamps + ((((size_t)(region) * 1024 + blockIdx.x) * 16 + (i)) * 2); \
        sp_[0] = __builtin_amdgcn_s_memtime(); sp_[1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define STAMP(region, i)
#endif

constexpr int BP = 128;            // padded batch (columns)
constexpr int H1N = SHEMS_L1, H2N = SHEMS_L2;
constexpr int SIN = 9, AIN = 2, CIN = 11;
constexpr int NT = 16;             // n-tiles of 32 over the 500 (512) layer-2 outputs
constexpr int KT = 8;              // k-tiles of 32 over the 250 (256) layer-1 outputs
constexpr int NQ = 8;              // blocks of the n range for the E / gradient tiles
constexpr int NQW = 64;            // n per block (the last block holds the 52 columns 448..499)

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
constexpr int NT16 = 32;           // K4 works on 16-wide n-tiles
constexpr int64_t WS_DAP = WS_IDX + BP;               // [NT16][2][BP]  per-n-tile partial d loss / d a_pi (K4)
constexpr int64_t WS_FW3C = WS_DAP + NT16 * AIN * BP; // [512]     frozen critic W3 (rows >= 500 zero)
constexpr int64_t WS_FW3A = WS_FW3C + 512;            // [512][2]  frozen actor W3
constexpr int64_t WS_FB3 = WS_FW3A + 1024;            // [8]       frozen b3: critic, critic_target, actor[0], actor[1]
constexpr int64_t WS_W1T = WS_FB3 + 8;                // [4 nets][12][256] packed layer-1 images (see w1m below)
constexpr int64_t WS_EA1 = WS_W1T + 4 * 12 * 256;     // [NQ][250][BP]  partial E of the actor's second output
constexpr int64_t WS_SLOT0 = WS_EA1 + NQ * H1N * BP;
constexpr int64_t SL_H2 = 0;                          // [500][BP]          relu(W2' h1 + b2)
constexpr int64_t SL_P3 = SL_H2 + H2N * BP;           // [NT][2][BP]        per-n-tile partial sums of layer 3
constexpr int64_t SL_EP = SL_P3 + NT * 2 * BP;        // [NQ][250][BP]      partial E (first output) over the n blocks
constexpr int64_t SL_SIZE = SL_EP + NQ * H1N * BP;
enum { SLOT_ACTOR_T = 0, SLOT_CRITIC_T = 1, SLOT_CRITIC = 2, SLOT_ACTOR = 3, SLOT_CRITIC2 = 4, N_SLOTS = 5 };
static_assert(WS_W1T % 4 == 0 && WS_SLOT0 % 4 == 0 && SL_SIZE % 4 == 0, "16-byte aligned blocks");
__host__ __device__ inline float *w1t_of(float *ws, int net) { return ws + WS_W1T + (int64_t)net * 12 * 256; }   // net = SLOT_* < 4
constexpr int64_t WS_SYNC = WS_SLOT0 + N_SLOTS * SL_SIZE;      // 96 reserved words (rounds 3 / 4 kept the bookkeeping of two removed launch forms here;
constexpr int64_t WS_FLOATS = WS_SYNC + 96;                    // the size stays so that existing callers' allocations and snapshots keep their layout)
static_assert(WS_FLOATS == kTpWsFloats, "shems_internal.h states the workspace size for shems_gupd.hip");

__host__ __device__ inline float *slot(float *ws, int s) { return ws + WS_SLOT0 + (int64_t)s * SL_SIZE; }

// Sum over the 64 lanes, returned in EVERY lane.  Data-parallel-primitive adds on the VALU (quad swaps, row mirrors, the two row
// broadcasts), no ds_bpermute round trips through the LDS crossbar: six dependent v_add_f32 instead of six ~100-cycle shuffles.
// Fixed association: ((pairs) quads) half rows) rows), then (row0 + row1) + (row2 + row3).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float x)
{
    const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xF, false);   // masked rows read 0.0f
    return x + __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float wave_sum(float x)
{
    x = dpp_add<0xB1, 0xF>(x);         // quad_perm [1,0,3,2]
    x = dpp_add<0x4E, 0xF>(x);         // quad_perm [2,3,0,1]
    x = dpp_add<0x141, 0xF>(x);        // row_half_mirror
    x = dpp_add<0x140, 0xF>(x);        // row_mirror: every lane of a row of 16 now holds the row's sum
    x = dpp_add<0x142, 0xA>(x);        // row_bcast15 into rows 1, 3
    x = dpp_add<0x143, 0xC>(x);        // row_bcast31 into rows 2, 3: lane 63 = total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// A value a workgroup of ANOTHER launch still running reads (the published actor copy of the pipelined training loop) is stored
// write-through -- relaxed agent-scope atomic store = global_store ... sc1 -- so that no release fence is needed (a device-scope release
// writes the whole L2 back: measured in round 1, it costs more than a launch boundary); see DevSync in shems_internal.h.
__device__ __forceinline__ void pub_store(float *p, float v, bool wt)
{
    if (wt) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
__device__ __forceinline__ float pub_load(const float *p, bool wt)
{
    return wt ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
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

__device__ __forceinline__ void gshift(XSrc &x, int64_t off)
{
    x.X = gsh(x.X, off); x.A = gsh(x.A, off); x.P3 = gsh(x.P3, off); x.b3 = gsh(x.b3, off); x.publish = gsh(x.publish, off);
}
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

// Split in two so that a kernel can issue these loads together with everything else it fetches and only then start consuming
// (one exposed global latency per kernel phase instead of one per helper).  blockDim = 256: one action element per thread.
template <int IN> struct XRegs { float v[5]; float ab[2]; float p[NT]; };      // ab = {stored action | actor b3}: an array member, like the others (scalar members of this by-reference aggregate were left in scratch)
template <int IN>
__device__ __forceinline__ void build_x_load(const XSrc &s, XRegs<IN> &R)
{
#pragma unroll
    for (int it = 0; it < 5; ++it) { const int e = it * 256 + threadIdx.x; R.v[it] = s.X[min(e, SIN * BP - 1)]; }   // clamped, never predicated:
    // a guarded load becomes a branch + its own s_waitcnt, which serialises the batch
    R.ab[0] = 0.0f; R.ab[1] = 0.0f;
    if (IN == CIN) {
        const int e = threadIdx.x, o = e / BP, m = e - o * BP;
        if (s.A) {
            R.ab[0] = s.A[e];
        } else {
            R.ab[1] = s.b3[o];
#pragma unroll
            for (int t = 0; t < NT; ++t) R.p[t] = s.P3[(t * 2 + o) * BP + m];
        }
    }
}
template <int IN>
__device__ __forceinline__ void build_x_store(const XSrc &s, const XRegs<IN> &R, float *xs /*LDS [IN][BP]*/, bool publisher)
{
#pragma unroll
    for (int it = 0; it < 5; ++it) { const int e = it * 256 + threadIdx.x; if (e < SIN * BP) xs[e] = R.v[it]; }
    if (threadIdx.x < BP) xs[11 * BP + threadIdx.x] = 1.0f;                     // bias row
    if (IN == SIN) {
        xs[9 * BP + threadIdx.x] = 0.0f;                                          // rows 9, 10 (2 * BP == blockDim)
    }
    if (IN == CIN) {
        const int e = threadIdx.x;                                                // AIN * BP == blockDim
        float a;
        if (s.A) {
            a = R.ab[0];
        } else {
            float acc = R.ab[1];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc += R.p[t];
            a = tanhf(acc);                                       // Dense(500, 2, tanh)
            if (publisher && s.publish) s.publish[e] = a;
        }
        xs[SIN * BP + e] = a;
    }
}

// The same for ONE 32-column tile [mbase, mbase + 32) (forward workgroups): xs is [12][32].  Threads < 64 own one action element
// (o = tid >> 5, column tid & 31) each; the state rows are 288 elements, two per thread at most.
template <int IN> struct XTRegs { float v[2]; float ab[2]; float p[NT]; };
template <int IN>
__device__ __forceinline__ void xt_load(const XSrc &s, int mbase, XTRegs<IN> &R)
{
    // rows 0..7: one element per thread; row 8: every thread loads (and later stores) the element of column tid & 31 -- eight threads
    // write the same value to the same LDS word.  Nothing is predicated: a store under `e < 288` pulls its load into that branch,
    // behind a full `s_waitcnt vmcnt(0)`.
    R.v[0] = s.X[((int)threadIdx.x >> 5) * BP + mbase + ((int)threadIdx.x & 31)];
    R.v[1] = s.X[8 * BP + mbase + ((int)threadIdx.x & 31)];
    R.ab[0] = 0.0f; R.ab[1] = 0.0f;
    if (IN == CIN) {                           // the critic's action rows: always an actor pass here (stored actions only enter through
                                               // K1's gather) -- one straight path, no second branch whose registers the loads must respect
        const int t = min((int)threadIdx.x, 63), o = t >> 5, m = mbase + (t & 31);
        R.ab[1] = s.b3[o];
#pragma unroll
        for (int q = 0; q < NT; ++q) R.p[q] = s.P3[(q * 2 + o) * BP + m];
    }
}
template <int IN>
__device__ __forceinline__ void xt_store(const XSrc &s, const XTRegs<IN> &R, int mbase, float *xs /*LDS [12][32]*/, bool publisher, bool wt = false)
{
    const int tid = threadIdx.x;
    xs[tid] = R.v[0];
    xs[256 + (tid & 31)] = R.v[1];
    if (tid < 32) xs[11 * 32 + tid] = 1.0f;                                       // bias row
    if (tid < 64) {
        float a = 0.0f;                                                           // rows 9, 10: zero for the actor nets
        if (IN == CIN) {
            float acc = R.ab[1];
#pragma unroll
            for (int q = 0; q < NT; ++q) acc += R.p[q];
            a = tanhf(acc);                                                       // Dense(500, 2, tanh)
            if (publisher && s.publish) pub_store(s.publish + (tid >> 5) * BP + mbase + (tid & 31), a, wt);
        }
        xs[SIN * 32 + tid] = a;
    }
}

// Layer 1 also runs on the matrix pipe: pre[k][m] = sum_j w1m[j][k] * xs[j][m] with K = 12 = 6 MFMA k-steps, where
//   w1m [12][256] = rows 0..in-1: W1[j][k]; row 11: b1[k]; everything else (rows in..10, columns 250..255) zero
//   xs  [12][BP]  = rows 0..in-1: the network input; row 11: 1.0 (bias); rows in..10 zero.
// K1 builds the packed images of the four networks in the workspace (from the weights as they stand when the update starts);
// staging one is a straight 12 KB float4 copy, three loads per thread.  Workgroups that need the image of weights changed
// earlier in the same update (K4) pack it themselves from the parameter block.
constexpr int W1K = 12, W1C = 256;
// (a native vector type, not HIP's float4 struct: copies of that struct from global memory into a by-reference aggregate become
// memcpy intrinsics into a private alloca that is never promoted -- 64 B of scratch per lane and a scratch round trip per launch)
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct W1mRegs { f32x4 v0, v1, v2; };
__device__ __forceinline__ void stage_w1m_load(const float *__restrict__ g, W1mRegs &R)
{
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g) + threadIdx.x;
    R.v0 = g4[0]; R.v1 = g4[256]; R.v2 = g4[512];
}
__device__ __forceinline__ void stage_w1m_store(const W1mRegs &R, float *l)
{
    f32x4 *l4 = reinterpret_cast<f32x4 *>(l) + threadIdx.x;
    l4[0] = R.v0; l4[256] = R.v1; l4[512] = R.v2;
}
struct PackRegs { float v[12]; };
__device__ __forceinline__ void pack_w1m_load(const float *__restrict__ P, int in, PackRegs &R)
{
#pragma unroll
    for (int j = 0; j < 12; ++j) {                       // thread = column k (blockDim 256), unconditional clamped loads
        const int k = min((int)threadIdx.x, H1N - 1);
        R.v[j] = P[(j == W1K - 1 ? in : min(j, in - 1)) * H1N + k];        // row `in` of the block is b1
    }
}
__device__ __forceinline__ void pack_w1m_store(const PackRegs &R, int in, float *__restrict__ g)
{
#pragma unroll
    for (int j = 0; j < 12; ++j)
        g[j * W1C + threadIdx.x] = ((j < in || j == W1K - 1) && (int)threadIdx.x < H1N) ? R.v[j] : 0.0f;
}
__device__ __forceinline__ void pack_w1m(const float *__restrict__ P, int in, float *__restrict__ g)
{
    PackRegs R;
    pack_w1m_load(P, in, R);
    pack_w1m_store(R, in, g);
}
// One 32(k) x 32(m) tile of layer-1 pre-activations, D layout (row k = (r&3)+8(r>>2)+4*lh, column m = lane&31).
template <int WSTRIDE>
__device__ __forceinline__ f32x16 l1_tile(const float *w1m, const float *xs, int kbase, int mbase, int li, int lh)
{
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = 0.0f;
    float a[W1K / 2], b[W1K / 2];                 // all 12 operand reads in one batch: with one wave per SIMD nothing else hides them
#pragma unroll
    for (int s = 0; s < W1K / 2; ++s) {
        const int j = 2 * s + lh;
        a[s] = w1m[j * WSTRIDE + kbase + li];
        b[s] = xs[j * BP + mbase + li];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < W1K / 2; ++s) t = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], t, 0, 0, 0);
    return t;
}

// ---- sample + gather + normalize ---------------------------------------------------------------------
struct PrepArgs {
    shems_ddpg d;
    shems_replay ring;
    int64_t ring_len;
    uint64_t seed;
    uint32_t tick;
    int64_t excl_pos, excl_count;
};
// Thread m < BP: minibatch column m.  Samples the ring slot (StatsBase.sample with replacement, MPS:33), gathers the transition and
// normalises.  `which` says what the calling tile workgroup needs in its LDS input block xs [.][BP]: 0 = normalize(s') (target
// actor), 1 = normalize(s) + the stored action (critic), 2 = normalize(s) (actor); xs == null with `publish`: the publishing
// workgroup, which writes everything later launches read to the workspace: normalize(s), normalize(s'), a, r, done,
// d(-mean q)/dq and the sampled slots.  Split in a load and a store half so that the gather (whose addresses need no memory: the
// slot comes from the counter RNG) is in flight together with every other load of the workgroup's first phase.
struct PrepRegs { float s[SIN], s2[SIN], lo[SIN], hi[SIN], a[2], r, dn; int64_t j; };
__device__ __forceinline__ void prep_load(const PrepArgs &A, int m, int which, bool publish, PrepRegs &R)
{
    const shems_ddpg &d = A.d;
    const shems_replay &ring = A.ring;
    const bool want_s2 = publish || which == 0, want_s = publish || which != 0, want_a = publish || which == 1;
    const u32x4 x = philox4x32_10((uint32_t)(m >> 2), 0u, A.tick, kStreamSample, (uint32_t)A.seed, (uint32_t)(A.seed >> 32));
    const uint32_t w = (m & 3) == 0 ? x.x : (m & 3) == 1 ? x.y : (m & 3) == 2 ? x.z : x.w;
    int64_t j = (int64_t)(w % (uint32_t)(A.ring_len - A.excl_count));
    if (A.excl_count > 0) j = (A.excl_pos + A.excl_count + j) % ring.capacity;         // skip the window another stream is writing
    R.j = j;                                                                            // (pad columns m >= batch gather a valid slot too and discard it)
#pragma unroll
    for (int k = 0; k < SIN; ++k) {
        R.s2[k] = want_s2 ? ring.s2[j * SIN + k] : 0.0f;
        R.s[k] = want_s ? ring.s[j * SIN + k] : 0.0f;
        R.lo[k] = d.s_min[k]; R.hi[k] = d.s_max[k];
    }
    R.a[0] = want_a ? ring.a[j * 2] : 0.0f;
    R.a[1] = want_a ? ring.a[j * 2 + 1] : 0.0f;
    R.r = publish ? ring.r[j] : 0.0f;
    R.dn = publish ? (ring.done[j] ? 1.0f : 0.0f) : 0.0f;
}
// xs (LDS input block, or null) has `xstride` columns; this thread's column goes to slot `xcol`.
__device__ __forceinline__ void prep_store(const PrepArgs &A, int m, float *xs, int xcol, int xstride, int which, bool publish, const PrepRegs &R)
{
    const shems_ddpg &d = A.d;
    float *ws = d.ws;
    const bool live = m < d.batch;
#pragma unroll
    for (int k = 0; k < SIN; ++k) {
        const float lo = R.lo[k], den = (R.hi[k] - lo) + 1e-8f;                       // MPS:56
        const float x2 = live ? (R.s2[k] - lo) / den : 0.0f;
        const float x1 = live ? (R.s[k] - lo) / den : 0.0f;
        if (xs) xs[k * xstride + xcol] = which == 0 ? x2 : x1;
        if (publish) {
            ws[WS_XT + k * BP + m] = x1;
            ws[WS_X2T + k * BP + m] = x2;
        }
    }
    const float a0 = live ? R.a[0] : 0.0f, a1 = live ? R.a[1] : 0.0f;
    if (xs) {
        xs[9 * xstride + xcol] = which == 1 ? a0 : 0.0f;
        xs[10 * xstride + xcol] = which == 1 ? a1 : 0.0f;
        xs[11 * xstride + xcol] = 1.0f;                                                // bias row
    }
    if (publish) {
        ws[WS_AT + m] = a0; ws[WS_AT + BP + m] = a1;
        ws[WS_R + m] = live ? R.r : 0.0f; ws[WS_DONE + m] = live ? R.dn : 0.0f;
        ws[WS_D3Q + m] = live ? -1.0f / (float)d.batch : 0.0f;                        // d(-mean q)/dq
        reinterpret_cast<int32_t *>(ws + WS_IDX)[m] = live ? (int32_t)R.j : -1;
    }
}

// ---- layers 1+2 forward for one 32-wide n-tile -------------------------------------------------------------------------
struct FwdJob {
    const float *w1t;      // packed layer-1 image [12][256] of this network (PREP == 0)
    const float *P;        // parameter block
    int in;                // 9 (actor nets) or 11 (critic nets)
    int out;               // 2 or 1
    int which;             // K1: what prep_column gathers for this job (0 s', 1 s + a, 2 s)
    XSrc x;
    float *H2;             // [500][BP] or null (target nets: nothing downstream needs it)
    float *P3;             // [NT][2][BP] layer-3 partials of this n-tile
    const float *d3q;      // QG: [BP] upstream gradient of q (-1/B, 0 in the pad columns)
    float *DAP;            // QG: [NT][2][BP] partial d loss / d a_pi
};
struct FwdArgs { FwdJob job[3]; int64_t gstride; int prep; PrepArgs pa; };   // prep: 1 = this launch opens the update (K1), 2 = K4
__device__ __forceinline__ void gshift(FwdJob &J, int64_t off)
{
    J.w1t = gsh(J.w1t, off); J.P = gsh(J.P, off); gshift(J.x, off); J.H2 = gsh(J.H2, off); J.P3 = gsh(J.P3, off);
    J.d3q = gsh(J.d3q, off); J.DAP = gsh(J.DAP, off);
}

// Workgroup = one 32-wide n-tile x 32 columns of the batch (64 workgroups per network).  K is cut into four quarters of 64 hidden
// units (rows 250..255 are zero), one per wave, each accumulated as its own 32-MFMA chain; the quarters are added in the fixed
// order ((q0 + q1) + q2) + q3.
//   * Layer 1 (K = 12, matrix pipe) leaves wave w with the pre-activations of ITS 64 hidden units x its 32 columns in registers, in
//     the MFMA D layout: lane (column, half) holds rows (r & 3) + 8 (r >> 2) + 4 half of each 32-row tile.  Layer 2 contracts over
//     those 64 units in any order, so its k-step r simply takes relu(t[r]) as the B operand -- straight from the registers -- and
//     fetches the matching W2 rows (r & 3) + 8 (r >> 2) + 4 half as the A operand: no LDS round trip for the activations.
//   * Every wave hands its partial tile to the others through LDS and finishes four of the sixteen accumulator rows (bias, relu,
//     store, layer-3 partial), so the epilogue is spread over the four SIMDs.
// QG (K4, the updated critic on [s; actor(s)]): after the forward tile the workgroup back-propagates the constant upstream gradient
// through its own 32 hidden units -- M[n][m] = d3q[m] W3[n] (h2 > 0), D1part[k][m] = sum_{n in tile} W2[k][n] M[n][m] (the W2 panel is
// still in LDS; wave w produces the k rows of its quarter, whose layer-1 tiles it still holds in registers for the relu mask) --
// and emits the tile's share of d loss / d a = W1[9.., k] (mask1 .* D1part).
template <bool QG> struct FwdShape {
    static constexpr int WST = QG ? 36 : 32;                  // row stride of the W2 panel (QG also reads it along n: 36 keeps the
                                                              // float4 staging stores aligned and that second read 2-way at worst)
    static constexpr int LDS = (256 * WST + W1K * 32 + W1K * W1C + 256 + 4 * 16 * 64 + 4 * 2 * 32 + (QG ? 32 * 32 + 4 * 2 * 32 : 0)) * 4;
};

template <int IN, int PREP, bool QG>
__device__ __forceinline__ void fwd_body(const FwdJob &J, float *smem, const PrepArgs *pa, int bx, int job)
{
    typedef FwdShape<QG> SH;
    constexpr int WST = SH::WST;
    float *Wc = smem;                          // [256][WST]  W2 panel (rows >= 250: zero)
    float *xs = Wc + 256 * WST;                // [12][32]  the network input, this workgroup's 32 columns
    float *w1 = xs + W1K * 32;                 // w1m [12][256]
    float *ep = w1 + W1K * W1C;                // [32][3]: b2, W3[.][0], W3[.][1] of this n-tile (+ 160 unused slots: every thread stores)
    float *xch = ep + 256;                     // [4 quarters][16 rows][64 lanes]
    float *pp = xch + 4 * 16 * 64;             // [4 row groups][2][32] layer-3 partials
    float *Mt = pp + 4 * 2 * 32;               // QG: [32 n][32 m]
    float *red = Mt + 32 * 32;                 // QG: [4 quarters][2][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    constexpr int kMTiles = BP / 32;           // workgroups per n-tile
    constexpr int kRegion = PREP == 1 ? 0 : PREP == 0 ? 1 : 3;
    (void)kRegion;
    STAMP(kRegion, 0);
    if (PREP == 1 && bx >= NT * kMTiles) {
        // The six extra workgroups of an update's first launch (job 0 only): one publishes what the later launches read from the
        // workspace (the sampled, gathered and normalised minibatch), four pack the layer-1 images of the four networks, one
        // freezes the output layers the gradient launches must read while their owners are updated in place.  Kept off the tile
        // workgroups so that none of those runs longer than the others.
        if (job != 0) return;
        const int duty = bx - NT * kMTiles;
        const shems_ddpg &d = pa->d;
        if (duty == 0) {
            if (tid < BP) {
                PrepRegs pr;
                prep_load(*pa, tid, 0, true, pr);
                prep_store(*pa, tid, nullptr, 0, 0, 0, true, pr);
            }
        } else if (duty <= 4) {
            const int net = duty - 1;
            const float *Pn = net == SLOT_ACTOR_T ? d.actor_t : net == SLOT_CRITIC_T ? d.critic_t : net == SLOT_CRITIC ? d.critic : d.actor;
            pack_w1m(Pn, (net == SLOT_CRITIC_T || net == SLOT_CRITIC) ? CIN : SIN, w1t_of(d.ws, net));
        } else {
            float *ws = d.ws;
            for (int e = tid; e < 512; e += 256) ws[WS_FW3C + e] = e < H2N ? d.critic[off_w3(CIN) + e] : 0.0f;
            for (int e = tid; e < 1024; e += 256) ws[WS_FW3A + e] = e < 2 * H2N ? d.actor[off_w3(SIN) + e] : 0.0f;
            if (tid < 8)
                ws[WS_FB3 + tid] = tid == 0 ? d.critic[off_b3(CIN, 1)] : tid == 1 ? d.critic_t[off_b3(CIN, 1)]
                                 : tid == 2 ? d.actor[off_b3(SIN, 2)] : tid == 3 ? d.actor[off_b3(SIN, 2) + 1] : 0.0f;
        }
        return;
    }
    // Tile -> (n-tile, m-tile): the four batch-column tiles of one n-tile read the same 32-KB W2 panel.  Workgroup ids go round the 8 XCDs, so
    // (round 4) an n-tile's four workgroups sit 8 ids apart -- one XCD, one L2: the panel is fetched from the Infinity Cache once, not four times.
    static_assert(kMTiles == 4 && NT % 8 == 0, "tile map: 4 m-tiles per n-tile, n-tiles in blocks of 8");
    const int ntile = (bx & 7) + 8 * (bx >> 5), n0 = ntile * 32;
    const int mbase = 32 * ((bx >> 3) & 3);
    const float *__restrict__ P = J.P;

    // ---- every global load of the workgroup goes out before the first one is consumed (one exposed latency).  The small operands of
    // layer 1 (input tile, layer-1 image, epilogue constants) are requested FIRST and the 32 KB W2 panel last: loads return in order,
    // so layer 1 runs on the matrix pipe while the panel is still arriving ----
    // (No load of this batch sits under a lane predicate, and no loaded value meets a select right at the load: either makes the
    // compiler put the load in a branch of its own with an `s_waitcnt vmcnt(0)` inside, which drains everything issued before it.
    // Padding is applied where the value is CONSUMED, or not at all where the other operand of the product is zero anyway.)
    const int ep_t = min(tid, 95), ep_nl = ep_t / 3, ep_c = ep_t - ep_nl * 3, ep_nc = min(n0 + ep_nl, H2N - 1);
    const float epv = P[ep_c == 0 ? off_b2(IN) + ep_nc : off_w3(IN) + ep_nc * J.out + min(ep_c - 1, J.out - 1)];   // epilogue constants
    const float ep_keep = (n0 + ep_nl >= H2N || ep_c - 1 >= J.out) ? 0.0f : 1.0f;     // (a product, not a select: see above)
    XTRegs<IN> xr;
    W1mRegs wr;
    PackRegs pk;
    PrepRegs pr;
    float d3q = 0.0f;
    if (QG) d3q = J.d3q[mbase + li];
    if (PREP != 1) xt_load<IN>(J.x, mbase, xr);
    if (PREP == 0) stage_w1m_load(J.w1t, wr); else pack_w1m_load(P, IN, pk);
    // First launch of an update: no separate sample/gather/pack launch.  Every tile workgroup samples the minibatch and gathers +
    // normalises what its network reads straight into its LDS input block, and packs the layer-1 image it needs from the
    // parameter block; the extra workgroups (above) publish the workspace copies for the later launches.
    if (PREP == 1 && tid < 32) prep_load(*pa, mbase + tid, J.which, false, pr);
    // W2[0..255][n0..n0+31] (rows of 128 B, 8 float4 each): 2048 float4, 8 per thread.  Rows 250..255 are copies of row 249 (clamped
    // address): they only ever multiply layer-1 activations that are exactly zero (the image has no columns >= 250), in the forward
    // product, and in QG's backward product they feed output rows whose relu mask is off.  Columns >= 500 of the last tile read the
    // next row / b2 (in bounds) and only feed output rows that are discarded.
    const float *__restrict__ W2 = P + off_w2(IN);
    f32x4 wv[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int e = it * 256 + tid, k = e >> 3, c = e & 7;
        wv[it] = *reinterpret_cast<const f32x4 *>(W2 + (int64_t)min(k, H1N - 1) * H2N + n0 + 4 * c);
    }
    // ---- consume the layer-1 operands ----
    if (PREP == 1) {
        if (tid < 32) prep_store(*pa, mbase + tid, xs, tid, 32, J.which, false, pr);
    } else {
        xt_store<IN>(J.x, xr, mbase, xs, ntile == 0);          // (the four column tiles of n-tile 0 publish actor(s) between them)
    }
    if (PREP == 0) stage_w1m_store(wr, w1); else pack_w1m_store(pk, IN, w1);
    ep[tid] = epv * ep_keep;                   // unconditional (threads >= 96 write copies of entry 95 into the unused slots): a store under
                                               // `tid < 96` would pull the load into that branch, behind a full drain
    STAMP(kRegion, 1);
    __syncthreads();
    STAMP(kRegion, 2);

    // layer 1 on the matrix pipe: the two 32-row tiles of this wave's K quarter x the workgroup's 32 columns; the accumulator chains
    // are independent, so their MFMAs interleave
    f32x16 t[2];
    {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) t[q][r] = 0.0f;
        float xb[W1K / 2], wa[W1K / 2][2];        // operand reads in one batch (see l1_tile)
#pragma unroll
        for (int sidx = 0; sidx < W1K / 2; ++sidx) {
            const int j = 2 * sidx + lh;
            xb[sidx] = xs[j * 32 + li];
#pragma unroll
            for (int q = 0; q < 2; ++q) wa[sidx][q] = w1[j * W1C + 64 * wave + 32 * q + li];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sidx = 0; sidx < W1K / 2; ++sidx)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                t[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[sidx][q], xb[sidx], t[q], 0, 0, 0);
    }
    // the W2 panel, now landed, goes to LDS (its own rows only are read by this wave in the forward pass, but the staging threads
    // are spread over all rows: barrier)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int e = it * 256 + tid, k = e >> 3, c = e & 7;
        *reinterpret_cast<f32x4 *>(Wc + k * WST + 4 * c) = wv[it];
    }
    __syncthreads();
    STAMP(kRegion, 3);
    // layer 2 over this wave's K quarter: 32 MFMA pairs, B operand = relu of the layer-1 registers, A operand = the W2 rows those
    // registers stand for
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    {
        const float *pa_ = Wc + (64 * wave + 4 * lh) * WST + li;
        float av[32];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) av[q * 16 + r] = pa_[(32 * q + (r & 3) + 8 * (r >> 2)) * WST];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q * 16 + r], fmaxf(t[q][r], 0.0f), acc, 0, 0, 0);
    }
    STAMP(kRegion, 4);
    // hand the partial tile over: xch[quarter = wave][row][lane]
#pragma unroll
    for (int r = 0; r < 16; ++r) xch[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    STAMP(kRegion, 5);
    // this wave finishes rows 4 wave .. 4 wave + 3 of the tile, i.e. hidden units n0 + r + 8 wave + 4 lh, column mbase + li
    {
        float xv[4][4], eb[4], e0[4], e1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) xv[q][r] = xch[(q * 16 + 4 * wave + r) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nl = r + 8 * wave + 4 * lh;
            eb[r] = ep[nl * 3]; e0[r] = ep[nl * 3 + 1]; e1[r] = ep[nl * 3 + 2];
        }
        __builtin_amdgcn_sched_barrier(0);
        const int m = mbase + li;
        float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nl = r + 8 * wave + 4 * lh, n = n0 + nl;
            const float sum = ((xv[0][r] + xv[1][r]) + xv[2][r]) + xv[3][r];
            const float h = n < H2N ? fmaxf(sum + eb[r], 0.0f) : 0.0f;
            if (J.H2 && n < H2N) J.H2[n * BP + m] = h;
            p0 = fmaf(h, e0[r], p0);
            p1 = fmaf(h, e1[r], p1);
            if (QG) Mt[nl * 32 + li] = h > 0.0f ? e0[r] * d3q : 0.0f;           // rows >= 500: h == 0
        }
        p0 += __shfl_xor(p0, 32, 64);
        p1 += __shfl_xor(p1, 32, 64);
        if (lh == 0) { pp[(wave * 2 + 0) * 32 + li] = p0; pp[(wave * 2 + 1) * 32 + li] = p1; }
    }
    __syncthreads();
    STAMP(kRegion, 6);
    if (tid < 64) {                                            // layer-3 partial of the tile: the four row groups in a fixed order
        const int o = tid >> 5, c = tid & 31;
        const float s = ((pp[(0 * 2 + o) * 32 + c] + pp[(1 * 2 + o) * 32 + c]) + pp[(2 * 2 + o) * 32 + c]) + pp[(3 * 2 + o) * 32 + c];
        J.P3[(ntile * 2 + o) * BP + mbase + c] = s;
    }
    if (!QG) return;
    STAMP(kRegion, 7);
    // backward through this n-tile: wave w -> rows k of its quarter (two tiles of 32), K = the tile's 32 hidden units.  The action
    // gradient of the quarters is added in the fixed order ((q0 + q1) + q2) + q3 by the threads that store it.
    {
        float bq[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) bq[u] = Mt[(2 * u + lh) * 32 + li];
        float da0 = 0.0f, da1 = 0.0f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int kb = 64 * wave + 32 * q;
            float aq[16], wa0[16], wa1[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) aq[u] = Wc[(kb + li) * WST + 2 * u + lh];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = kb + (r & 3) + 8 * (r >> 2) + 4 * lh;
                wa0[r] = w1[9 * W1C + k]; wa1[r] = w1[10 * W1C + k];               // W1[9 + o][k]: the action rows (columns >= 250 zero)
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16 g;
#pragma unroll
            for (int r = 0; r < 16; ++r) g[r] = 0.0f;
#pragma unroll
            for (int u = 0; u < 16; ++u) g = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u], bq[u], g, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = t[q][r] > 0.0f ? g[r] : 0.0f;                       // layer-1 relu mask (pre-activations still in registers)
                da0 = fmaf(wa0[r], v, da0);
                da1 = fmaf(wa1[r], v, da1);
            }
        }
        da0 += __shfl_xor(da0, 32, 64);
        da1 += __shfl_xor(da1, 32, 64);
        if (lh == 0) { red[(wave * 2 + 0) * 32 + li] = da0; red[(wave * 2 + 1) * 32 + li] = da1; }
    }
    STAMP(kRegion, 8);
    __syncthreads();
    if (tid < 64) {                                            // (o, column) of this workgroup
        const int o = tid >> 5, c = tid & 31;
        const float s = ((red[(0 * 2 + o) * 32 + c] + red[(1 * 2 + o) * 32 + c]) + red[(2 * 2 + o) * 32 + c]) + red[(3 * 2 + o) * 32 + c];
        J.DAP[(ntile * 2 + o) * BP + mbase + c] = s;
    }
    STAMP(kRegion, 9);
}

// ---- K4 on 16-wide n-tiles: 32 x 4 = 128 workgroups instead of 64, half the W2 panel and half the MFMA chains per workgroup ----
// Same scheme as fwd_body<QG> on v_mfma_f32_16x16x4_f32 (lane l: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D[i = 4 (l >> 4) + r]
// [j = l & 15], r < 4).  Workgroup = 16 hidden units x 32 columns; wave w = K quarter w (64 hidden units of layer 1 = 4 blocks of 16),
// both 16-column halves.  Layer 1's D block (kb, mb) holds pre[64 w + 16 kb + 4 g + r][16 mb + c] in lane (g, c): layer-2 step
// (kb, r) contracts over the four units {.. + 4 g + r : g} with relu(t) straight from the registers as the B operand and the
// matching W2 rows as the A operand; the backward product D1part[k][m] = sum_{n in tile} W2[k][n] M[n][m] comes out in the same
// (kb, mb) block layout, so the relu mask is again the layer-1 registers.  Two independent accumulators (the column halves) cover
// the 40-cycle dependent latency of this instruction.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int QG16_WST = 20;                                   // W2 panel row stride: 16-B aligned rows, both operand reads conflict-free
constexpr int QG16_LDS = (256 * QG16_WST + W1K * 32 + W1K * W1C + 256 + 4 * 2 * 4 * 64 + 4 * 64 + 16 * 32 + 4 * 2 * 2 * 64) * 4;

__device__ __forceinline__ void qg16_body(const FwdJob &J, float *smem, int bx)
{
    constexpr bool wt = false;                 // (round 3's merged K4 + K5 launch stored these write-through; removed in round 4, see WS_SYNC)
    constexpr int WST = QG16_WST;
    float *Wc = smem;                          // [256][WST]  W2[k][n0 .. n0 + 15] (rows >= 250: copies of row 249, never effective)
    float *xs = Wc + 256 * WST;                // [12][32]
    float *w1 = xs + W1K * 32;                 // w1m [12][256]
    float *ep = w1 + W1K * W1C;                // [16][2]: b2, W3 of this n-tile (+ unused slots: every thread stores)
    float *xch = ep + 256;                     // [4 quarters][2 mb][4 r][64 lanes]
    float *pp = xch + 4 * 2 * 4 * 64;          // [4 waves][64 lanes] layer-3 partials
    float *Mt = pp + 4 * 64;                   // [16 n][32 m]
    float *red = Mt + 16 * 32;                 // [4 quarters][2 j][2 mb][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    const int ntile = (bx & 7) + 8 * (bx >> 5), n0 = ntile * 16, mbase = 32 * ((bx >> 3) & 3);      // an n-tile's four workgroups on one XCD (see fwd_body)
    const float *__restrict__ P = J.P;
    STAMP(3, 0);
    // ---- one burst of loads: epilogue constants, input tile, layer-1 image, upstream gradient, then the W2 panel ----
    const int ep_t = min(tid, 31), ep_nl = ep_t >> 1, ep_c = ep_t & 1, ep_nc = min(n0 + ep_nl, H2N - 1);
    const float epv = P[ep_c == 0 ? off_b2(CIN) + ep_nc : off_w3(CIN) + ep_nc];
    const float ep_keep = n0 + ep_nl >= H2N ? 0.0f : 1.0f;
    XTRegs<CIN> xr;
    PackRegs pk;
    xt_load<CIN>(J.x, mbase, xr);
    pack_w1m_load(P, CIN, pk);
    const float d3q = J.d3q[mbase + (lane & 31)];              // column 16 mb + c of the finishing lane (see below)
    const float *__restrict__ W2 = P + off_w2(CIN);
    f32x4 wv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int e = it * 256 + tid, k = e >> 2, c4 = e & 3;
        wv[it] = *reinterpret_cast<const f32x4 *>(W2 + (int64_t)min(k, H1N - 1) * H2N + n0 + 4 * c4);
    }
    xt_store<CIN>(J.x, xr, mbase, xs, ntile == 0, wt);        // (the four column tiles of n-tile 0 publish actor(s) between them)
    pack_w1m_store(pk, CIN, w1);
    ep[tid] = epv * ep_keep;
    STAMP(3, 1);
    __syncthreads();
    STAMP(3, 2);
    // ---- layer 1: 4 k-blocks x 2 column halves, K = 12 = 3 steps ----
    f32x4 t[4][2];
    {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) t[kb][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float av[3][4], bv[3][2];
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) {
            const int j = 4 * sx + g;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) av[sx][kb] = w1[j * W1C + 64 * wave + 16 * kb + c];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) bv[sx][mb] = xs[j * 32 + 16 * mb + c];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sx = 0; sx < 3; ++sx)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) t[kb][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[sx][kb], bv[sx][mb], t[kb][mb], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int e = it * 256 + tid, k = e >> 2, c4 = e & 3;
        *reinterpret_cast<f32x4 *>(Wc + k * WST + 4 * c4) = wv[it];
    }
    __syncthreads();
    STAMP(3, 3);
    // ---- layer 2 over this wave's K quarter: 16 steps x 2 column halves ----
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    {
        float aw[16];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) aw[kb * 4 + r] = Wc[(64 * wave + 16 * kb + 4 * g + r) * WST + c];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
                    acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[kb * 4 + r], fmaxf(t[kb][mb][r], 0.0f), acc[mb], 0, 0, 0);
    }
    STAMP(3, 4);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) xch[((wave * 2 + mb) * 4 + r) * 64 + lane] = acc[mb][r];
    __syncthreads();
    STAMP(3, 5);
    // this wave finishes column half mb = wave >> 1, rows r in {2 (wave & 1), 2 (wave & 1) + 1} of every lane: hidden unit n0 + 4 g + r
    {
        const int mb = wave >> 1, r0 = 2 * (wave & 1);
        float xv[4][2], eb[2], ew[2];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) xv[q][rr] = xch[((q * 2 + mb) * 4 + r0 + rr) * 64 + lane];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) { const int nl = 4 * g + r0 + rr; eb[rr] = ep[nl * 2]; ew[rr] = ep[nl * 2 + 1]; }
        const float dq = __shfl(d3q, 16 * mb + c, 64);          // upstream gradient of column 16 mb + c (lanes 0..31 hold columns 0..31)
        float p0 = 0.0f;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int nl = 4 * g + r0 + rr, n = n0 + nl;
            const float sum = ((xv[0][rr] + xv[1][rr]) + xv[2][rr]) + xv[3][rr];
            const float h = n < H2N ? fmaxf(sum + eb[rr], 0.0f) : 0.0f;
            p0 = fmaf(h, ew[rr], p0);
            Mt[nl * 32 + 16 * mb + c] = h > 0.0f ? ew[rr] * dq : 0.0f;
        }
        pp[wave * 64 + lane] = p0;
    }
    __syncthreads();
    STAMP(3, 6);
    if (tid < 32) {                                            // layer-3 partial (q, for the loss report): 2 waves x 4 lane groups per column, fixed order
        const int mb = tid >> 4, cc = tid & 15;
        float sres = 0.0f;
#pragma unroll
        for (int wv_ = 0; wv_ < 2; ++wv_)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) sres += pp[(2 * mb + wv_) * 64 + 16 * gg + cc];
        pub_store(J.P3 + ntile * BP + mbase + tid, sres, wt);
    }
    // ---- backward through this n-tile: rows k of the wave's quarter (4 blocks) x 2 column halves, K = 16 = 4 steps ----
    {
        float bm[4][2];
#pragma unroll
        for (int sx = 0; sx < 4; ++sx)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) bm[sx][mb] = Mt[(4 * sx + g) * 32 + 16 * mb + c];
        float da[2][2] = {{0.f, 0.f}, {0.f, 0.f}};             // [j][mb]
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            float ak[4], wa0[4], wa1[4];
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) ak[sx] = Wc[(64 * wave + 16 * kb + c) * WST + 4 * sx + g];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 64 * wave + 16 * kb + 4 * g + r;
                wa0[r] = w1[9 * W1C + k]; wa1[r] = w1[10 * W1C + k];               // W1[9 + j][k]: the action rows (columns >= 250 zero)
            }
            f32x4 gk[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int sx = 0; sx < 4; ++sx)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) gk[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ak[sx], bm[sx][mb], gk[mb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = t[kb][mb][r] > 0.0f ? gk[mb][r] : 0.0f;         // layer-1 relu mask (pre-activations still in registers)
                    da[0][mb] = fmaf(wa0[r], v, da[0][mb]);
                    da[1][mb] = fmaf(wa1[r], v, da[1][mb]);
                }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) red[((wave * 2 + j) * 2 + mb) * 64 + lane] = da[j][mb];
    }
    STAMP(3, 8);
    __syncthreads();
    if (tid < 64) {                                            // (j, column): the four quarters, each the sum of its four lane groups, fixed order
        const int j = tid >> 5, mm = tid & 31, mb = mm >> 4, cc = mm & 15;
        float sres = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float *rq = red + ((q * 2 + j) * 2 + mb) * 64 + cc;
            sres += ((rq[0] + rq[16]) + rq[32]) + rq[48];
        }
        pub_store(J.DAP + (ntile * 2 + j) * BP + mbase + mm, sres, wt);
    }
    STAMP(3, 9);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_fwd(FwdArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int kPerJob = NT * (BP / 32) + 6;    // K1: a one-dimensional grid of 3 x (64 tile + 6 publishing) workgroups
    const int job = A.prep == 1 ? (int)blockIdx.x / kPerJob : 0, bx = (int)blockIdx.x - job * kPerJob;
    FwdJob J = job == 0 ? A.job[0] : job == 1 ? A.job[1] : A.job[2];      // (no dynamic indexing of the kernarg block: that goes through scratch)
    gshift(J, blockIdx.z * A.gstride);             // learner blockIdx.z (stride 0 for a single learner)
    if (A.prep == 1) {                             // K1: three jobs, each workgroup gathers its own input block
        PrepArgs pa = A.pa;
        gshift(pa.d, blockIdx.z * A.gstride); gshift(pa.ring, blockIdx.z * A.gstride); pa.seed += blockIdx.z;
        if (J.in == SIN) fwd_body<SIN, 1, false>(J, smem, &pa, bx, job); else fwd_body<CIN, 1, false>(J, smem, &pa, bx, job);
        return;
    }
    qg16_body(J, smem, bx);                        // K4: the updated critic on [s; actor(s)], forward + input gradient
}

// ---- K2: critic_target forward | the three E products ---------------------------------------------------------------------
// E workgroup (kt, nq) of set j: Epart[nq][32 k][128 m] = sum_{n in block nq} W2[k][n] * (W3[n][j] * (h2[n][m] > 0)).
struct EJob {
    const float *P;        // parameter block of the network
    int in, out, col;      // col = j
    const float *H2;       // [500][BP]
    float *EP;             // [NQ][250][BP]
};
struct MidArgs { FwdJob fwd; EJob e[3]; int64_t gstride; int nfwd; };
constexpr int E_LDS = (NQW * BP + 32 * (NQW + 1) + NQW) * 4;

__device__ __forceinline__ void e_body(const EJob &E, int b, float *smem)
{
    float *Bt = smem;                          // [64 n][128 m]  M panel
    float *At = Bt + NQW * BP;                 // [32 k][65]     W2 panel
    float *w3s = At + 32 * (NQW + 1);          // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int kt = b >> 3, nq = b & 7, nb = nq * NQW;
    const int mcol = tid & 127, half = tid >> 7;
    const float *__restrict__ P = E.P;
    const float *__restrict__ W2p = P + off_w2(E.in);
    STAMP(1, 0);
    float hv[32], wvp[8], w3v = 0.0f;
#pragma unroll
    for (int u = 0; u < 32; ++u) hv[u] = E.H2[min(nb + 2 * u + half, H2N - 1) * BP + mcol];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = u * 256 + tid, kl = e >> 6, nl = e & 63, k = kt * 32 + kl, n = nb + nl;
        wvp[u] = W2p[(int64_t)min(k, H1N - 1) * H2N + min(n, H2N - 1)];     // clamped copies: rows k >= 250 are never stored, columns
                                                                              // n >= 500 meet the zero rows of the M panel below
    }
    w3v = P[off_w3(E.in) + min(nb + min(tid, NQW - 1), H2N - 1) * E.out + E.col];
    if (tid < NQW) w3s[tid] = nb + tid < H2N ? w3v : 0.0f;
    STAMP(1, 1);
    __syncthreads();
    STAMP(1, 2);
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        const int nl = 2 * u + half;
        Bt[nl * BP + mcol] = hv[u] > 0.0f ? w3s[nl] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = u * 256 + tid, kl = e >> 6, nl = e & 63;
        At[kl * (NQW + 1) + nl] = wvp[u];
    }
    STAMP(1, 3);
    __syncthreads();
    STAMP(1, 4);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const float *pa = At + li * (NQW + 1), *pb = Bt + wave * 32 + li;
    {   // 32 MFMA pairs (4 groups of 8), operand fetch one group ahead
        float ac[8], bc[8], an[8], bn[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { ac[u] = pa[2 * u + lh]; bc[u] = pb[(2 * u + lh) * BP]; }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int nn = 2 * ((g + 1) * 8 + u) + lh;
                    an[u] = pa[nn]; bn[u] = pb[nn * BP];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], bc[u], acc, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 8; ++u) { ac[u] = an[u]; bc[u] = bn[u]; }
        }
    }
    STAMP(1, 5);
    const int m = wave * 32 + li;
    float *D = E.EP + (int64_t)nq * H1N * BP;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (k < H1N) D[k * BP + m] = acc[r];
    }
    STAMP(1, 6);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_mid(MidArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int64_t off = blockIdx.z * A.gstride;
    if ((int)blockIdx.x < A.nfwd) {
        FwdJob J = A.fwd;
        gshift(J, off);
        fwd_body<CIN, 0, false>(J, smem, nullptr, (int)blockIdx.x, 0);
        return;
    }
    const int e = (int)blockIdx.x - A.nfwd;
    const int set = e / (KT * NQ);
    EJob E = set == 0 ? A.e[0] : set == 1 ? A.e[1] : A.e[2];
    E.P = gsh(E.P, off); E.H2 = gsh(E.H2, off); E.EP = gsh(E.EP, off);
    e_body(E, e % (KT * NQ), smem);
}

// ---- Flux 0.12.1 ADAM + soft target update: AdamCtx / adam_math live in shems_adam.h (shared with shems_gupd.hip) ----
__device__ __forceinline__ void adam_elem(const AdamCtx &c, int i, float graw)
{
    float m = c.mt[i], v = c.vt[i], p = c.p[i], t = c.target[i];
    adam_math(c, graw, m, v, p, t);
    c.mt[i] = m; c.vt[i] = v; c.p[i] = p; c.target[i] = t;
    if (c.publish) pub_store(c.publish + i, p, true);
}
// N independent elements of one lane: all loads first (one exposed latency), then the arithmetic, then the stores.  idx < 0: skip.
template <int N>
__device__ __forceinline__ void adam_batch(const AdamCtx &c, const int (&idx)[N], const float (&g)[N])
{
    float m[N], v[N], p[N], t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int e = max(idx[i], 0);
        m[i] = c.mt[e]; v[i] = c.vt[e]; p[i] = c.p[e]; t[i] = c.target[e];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) adam_math(c, g[i], m[i], v[i], p[i], t[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (idx[i] >= 0) {
            const int e = idx[i];
            c.mt[e] = m[i]; c.vt[e] = v[i]; c.p[e] = p[i]; c.target[e] = t[i];
            if (c.publish) pub_store(c.publish + e, p[i], true);
        }
    }
}
// The same in two halves, so that a gradient tile can request its moments / parameters / targets with its first burst of loads,
// long before the gradient exists (the elements are its own: nobody else touches them), and only computes + stores at the end.
template <int N> struct AdamRegs { float m[N], v[N], p[N], t[N]; };
template <int N>
__device__ __forceinline__ void adam_load(const AdamCtx &c, const int (&idx)[N], AdamRegs<N> &R)
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int e = max(idx[i], 0);
        R.m[i] = c.mt[e]; R.v[i] = c.vt[e]; R.p[i] = c.p[e]; R.t[i] = c.target[e];
    }
}
template <int N>
__device__ __forceinline__ void adam_apply(const AdamCtx &c, const int (&idx)[N], const float (&g)[N], AdamRegs<N> &R)
{
#pragma unroll
    for (int i = 0; i < N; ++i) adam_math(c, g[i], R.m[i], R.v[i], R.p[i], R.t[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (idx[i] >= 0) {
            const int e = idx[i];
            c.mt[e] = R.m[i]; c.vt[e] = R.v[i]; c.p[e] = R.p[i]; c.target[e] = R.t[i];
            if (c.publish) pub_store(c.publish + e, R.p[i], true);
        }
    }
}
// Four consecutive elements per thread (16-byte accesses: a quarter of the workgroups, the same bits).  i0 is a multiple of 4.
__device__ __forceinline__ void adam_vec4(const AdamCtx &c, int i0)
{
    if (i0 + 3 < c.n) {
        const float4 g4 = *reinterpret_cast<const float4 *>(c.g + i0);
        float4 m4 = *reinterpret_cast<const float4 *>(c.mt + i0), v4 = *reinterpret_cast<const float4 *>(c.vt + i0);
        float4 p4 = *reinterpret_cast<const float4 *>(c.p + i0), t4 = *reinterpret_cast<const float4 *>(c.target + i0);
        adam_math(c, g4.x, m4.x, v4.x, p4.x, t4.x);
        adam_math(c, g4.y, m4.y, v4.y, p4.y, t4.y);
        adam_math(c, g4.z, m4.z, v4.z, p4.z, t4.z);
        adam_math(c, g4.w, m4.w, v4.w, p4.w, t4.w);
        *reinterpret_cast<float4 *>(c.mt + i0) = m4; *reinterpret_cast<float4 *>(c.vt + i0) = v4;
        *reinterpret_cast<float4 *>(c.p + i0) = p4; *reinterpret_cast<float4 *>(c.target + i0) = t4;
        if (c.publish) *reinterpret_cast<float4 *>(c.publish + i0) = p4;
    } else {
        for (int i = i0; i < c.n; ++i) adam_elem(c, i, c.g[i]);
    }
}

// Data-parallel form only (a gradient all-reduce sits between the gradient launch and this sweep).
__global__ __launch_bounds__(256) void k_adam_soft(AdamCtx c, int64_t gstride)
{
    gshift(c, blockIdx.z * gstride);
    const int i0 = 4 * ((int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x);
    if (i0 < c.n) adam_vec4(c, i0);
}
// (two elements per thread -- twice the workgroups, 252 of the 256 CUs covered -- was measured inside the noise of the 7-launch form in
// round 4, 40.1-43.6 against 40.2-41.4 us, and removed)

// Data-parallel form with the DIRECT exchange (XchgArgs, shems_internal.h): push my range of the gradient to every peer, wait for
// theirs, sum in rank order, ADAM.  One workgroup = 1 024 consecutive parameters = 256 threads x 4; the same range on every rank.
__global__ __launch_bounds__(256) void k_adam_xchg(AdamCtx c, XchgArgs x)
{
    const int tid = threadIdx.x, wg = blockIdx.x, i0 = 4 * (wg * 256 + tid);
    const int par = (int)(x.epoch & 1ull);
    // a record poisoned by an earlier exchange: this sweep is a no-op (nothing pushed, nothing waited for, nothing applied) -- the host
    // answers every later call with SHEMS_ERR_STATE, and whatever was already enqueued drains without another bounded wait
    __shared__ unsigned s_dead;
    if (tid == 0) s_dead = __hip_atomic_load(x.poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (s_dead) return;
    const bool full = i0 + 3 < c.n;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    if (full) { const float4 v = *reinterpret_cast<const float4 *>(c.g + i0); g[0] = v.x; g[1] = v.y; g[2] = v.z; g[3] = v.w; }
    else for (int e = 0; e < 4; ++e) if (i0 + e < c.n) g[e] = c.g[i0 + e];
    // push: slot [par][my rank] of every peer's inbox (16-byte stores; the pad beyond n carries zeros)
    const int64_t my_slot = ((int64_t)par * x.world + x.rank) * kXchgNmax + i0;
    for (int q = 0; q < x.world; ++q)
        if (q != x.rank) *reinterpret_cast<float4 *>(x.inbox[q] + my_slot) = make_float4(g[0], g[1], g[2], g[3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                      // system scope: the pushes are visible to the peers before the flags
    __syncthreads();
    if (tid < x.world && tid != x.rank)
        __hip_atomic_store(x.flags[tid] + ((int64_t)par * x.world + x.rank) * kXchgWgs + wg, x.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // wait: the same workgroup of every peer has pushed this epoch into MY inbox (bounded: a peer that never comes is counted, not waited for)
    if (tid < x.world && tid != x.rank) {
        const unsigned long long *f = x.flags[x.rank] + ((int64_t)par * x.world + tid) * kXchgWgs + wg;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < x.epoch) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > x.wait_ticks) {
                // gave up: count it, and poison the record for good (host-visible at once; the replicas can no longer be trusted)
                __hip_atomic_fetch_add(x.timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(x.poison, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                s_dead = 1u;
                break;
            }
            __builtin_amdgcn_s_sleep(32);
        }
    }
    __syncthreads();
    if (s_dead) return;                                                // an incomplete sum is never applied
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    // sum in rank order
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < x.world; ++r) {
        float v[4];
        if (r == x.rank) { for (int e = 0; e < 4; ++e) v[e] = g[e]; }
        else {
            const float4 t = *reinterpret_cast<const float4 *>(x.inbox[x.rank] + ((int64_t)par * x.world + r) * kXchgNmax + i0);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        }
        for (int e = 0; e < 4; ++e) s4[e] = r == 0 ? v[e] : s4[e] + v[e];
    }
    // the summed gradient replaces the local one (what an all-reduce leaves), then ADAM + soft update on the owner's four elements
    for (int e = 0; e < 4; ++e) {
        const int i = i0 + e;
        if (i < c.n) {
            float m = c.mt[i], v = c.vt[i], p = c.p[i], t = c.target[i];
            adam_math(c, s4[e], m, v, p, t);
            c.mt[i] = m; c.vt[i] = v; c.p[i] = p; c.target[i] = t;
            const_cast<float *>(c.g)[i] = s4[e];
            if (c.publish) pub_store(c.publish + i, p, true);
        }
    }
}

// ---- critic loss head, evaluated in the prologue of every K3 workgroup (cheaper than a launch boundary) --------
// d3[0][m] = dq[m] = 2 (q - y) / B into LDS; workgroup 0 also publishes y, q, the loss and gb3 (+ its ADAM step when fused).
// The global operands of the two heads, fetched with the rest of a workgroup's first batch of loads (all threads load;
// the critic head uses the values of threads < BP only).  b3 of both critics comes from the copy K1 froze: the owner of that
// element updates it (and the target's) in place during this launch.
struct HeadRegs { float v[2 * NT]; float s[4]; };
static_assert(2 * NT >= NT16, "HeadRegs.v holds the actor head's NT16 partials");       // s = {a, b, c, e} scalars of the heads
__device__ __forceinline__ void head_loss_load(const shems_ddpg &d, HeadRegs &R)
{
    const float *ws = d.ws;
    const int m = threadIdx.x & 127;
    const float *Pt = slot(d.ws, SLOT_CRITIC_T) + SL_P3, *Pc = slot(d.ws, SLOT_CRITIC) + SL_P3;
#pragma unroll
    for (int i = 0; i < NT; ++i) { R.v[i] = Pt[(i * 2) * BP + m]; R.v[NT + i] = Pc[(i * 2) * BP + m]; }
    R.s[0] = ws[WS_FB3 + 1];
    R.s[1] = ws[WS_FB3 + 0];
    R.s[2] = ws[WS_R + m];
    R.s[3] = ws[WS_DONE + m];
}
__device__ __forceinline__ void head_actor_load(const shems_ddpg &d, HeadRegs &R, bool wt = false)
{
    const float *ws = d.ws;
    const int t = threadIdx.x, o = t >> 7, m = t & 127;
#pragma unroll
    for (int p = 0; p < NT16; ++p) R.v[p] = pub_load(ws + WS_DAP + (int64_t)(p * 2 + o) * BP + m, wt);
    R.s[0] = pub_load(ws + WS_API + t, wt);
    R.s[1] = R.s[2] = R.s[3] = 0.0f;
}

__device__ __forceinline__ void head_loss(const shems_ddpg &d, const HeadRegs &R, float *d3 /*LDS [2][BP]*/, float *red /*LDS [8]*/, bool publisher,
                                          const AdamCtx *fuse)
{
    float *ws = d.ws;
    const int t = threadIdx.x, m = t & 127;
    float dq = 0.0f, diff = 0.0f;
    if (t < BP) {
        float q2 = R.s[0], q = R.s[1];
#pragma unroll
        for (int i = 0; i < NT; ++i) { q2 += R.v[i]; q += R.v[NT + i]; }
        const float y = R.s[2] + d.gamma * (1.0f - R.s[3]) * q2;                                  // DDPG.jl:133
        diff = m < d.batch ? q - y : 0.0f;
        dq = 2.0f * diff / (float)d.batch;                                                  // d mse / d q
        if (publisher) { ws[WS_Y + m] = y; ws[WS_Q + m] = q; ws[WS_D3C + m] = dq; }
    }
    d3[t] = t < BP ? dq : 0.0f;
    if (publisher) {
        const float s1 = wave_sum(diff * diff), s2 = wave_sum(dq);
        if ((t & 63) == 0) { red[t >> 6] = s1; red[4 + (t >> 6)] = s2; }
        __syncthreads();
        if (t == 0) {
            d.losses[0] = (red[0] + red[1]) / (float)d.batch;                               // Flux.mse
            const float g = red[4] + red[5];
            d.grad_critic[off_b3(CIN, 1)] = g;
            if (fuse) adam_elem(*fuse, off_b3(CIN, 1), g);
        }
    }
}

// ---- actor head backward, evaluated in the prologue of every K5 workgroup ------------------------------------
// d3[o][m] = (sum of the NT partial d loss / d a_pi) * (1 - a_pi^2); workgroup 0 publishes d3, the actor loss and gb3.
__device__ __forceinline__ void head_actor(const shems_ddpg &d, const HeadRegs &R, float *d3 /*LDS [2][BP]*/, float *red /*LDS [8]*/, bool publisher,
                                           const AdamCtx *fuse, bool wt = false)
{
    float *ws = d.ws;
    const int t = threadIdx.x, o = t >> 7, m = t & 127;
    const float a = R.s[0];
    float da = 0.0f;
#pragma unroll
    for (int p = 0; p < NT16; ++p) da += R.v[p];
    const float g = da * (1.0f - a * a);                       // through tanh
    d3[t] = g;
    if (publisher) {
        pub_store(ws + WS_D3A + t, g, wt);
        float q = 0.0f;
        if (o == 0 && m < d.batch) {
            const float *Pq = slot(ws, SLOT_CRITIC2) + SL_P3;
            q = d.critic[off_b3(CIN, 1)];
#pragma unroll
            for (int i = 0; i < NT16; ++i) q += pub_load(Pq + i * BP + m, wt);     // K4's 32 tiles of 16 hidden units
        }
        const float sg = wave_sum(g), sq = wave_sum(q);
        if ((t & 63) == 0) { red[t >> 6] = sg; red[4 + (t >> 6)] = sq; }
        __syncthreads();
        if (t == 0) {
            const float g0 = red[0] + red[1], g1 = red[2] + red[3];
            d.grad_actor[off_b3(SIN, 2) + 0] = g0;
            d.grad_actor[off_b3(SIN, 2) + 1] = g1;
            d.losses[1] = -(red[4] + red[5]) / (float)d.batch;     // loss_act = -mean(critic(vcat(s, actor(s))))
            if (fuse) { adam_elem(*fuse, off_b3(SIN, 2), g0); adam_elem(*fuse, off_b3(SIN, 2) + 1, g1); }
        }
    }
}

// ---- K3 / K5: every gradient block of one network, by batch contractions only -------------------------------------------
// D2[n][m] = (sum_o W3[n][o] d3[o][m]) * (h2[n][m] > 0) is generated while staging, never stored.
//   W workgroups (kt, nt): gW2[32 k][32 n] = sum_m h1[k][m] D2[n][m], one 16 x 16 block per wave over the whole batch
//                 (v_mfma_f32_16x16x4_f32), owned by that wave for ADAM;
//   G workgroups: gb2[n] = sum_m D2[n][m], gW3[n][o] = sum_m h2[n][m] d3[o][m], 32 rows each;
//   R workgroups: one wave per hidden unit k: D1[k][m] = mask1 * sum_j d3[j][m] * (sum_q Epart_j[q][k][m]); gb1[k] = sum_m D1;
//                 gW1[j][k] = sum_m x[j][m] D1[k][m].
// With A.fuse every workgroup then applies ADAM + the soft target update to exactly the elements it produced.
struct GradArgs {
    const float *w1t;      // packed layer-1 image of that network (as K1 built it)
    const float *P;        // parameter block of the network being differentiated (read only by the owner of each element)
    int in, out;
    XSrc x;                // its input (workspace data)
    const float *H2;       // [500][BP]
    const float *w3f;      // frozen W3 [512][out]
    float *grad;           // gradient block
    const float *E0, *E1;  // [NQ][250][BP] partial E of output 0 / 1 (E1 null for the critic)
    int head;              // 1 = critic loss head, 2 = actor head
    int fuse;              // apply ADAM + soft update here
    shems_ddpg dd;         // for the heads
    AdamCtx c;
    int64_t gstride;       // learner groups: byte stride between learners (0 = single learner)
};
__device__ __forceinline__ void gshift(GradArgs &B, int64_t off)
{
    B.w1t = gsh(B.w1t, off); B.P = gsh(B.P, off); gshift(B.x, off); B.H2 = gsh(B.H2, off); B.w3f = gsh(B.w3f, off);
    B.grad = gsh(B.grad, off); B.E0 = gsh(B.E0, off); B.E1 = gsh(B.E1, off); gshift(B.dd, off); gshift(B.c, off);
}
// W tiles are 32 k x 32 n (128 workgroups), G workgroups take 32 rows (16 workgroups).  (A finer tiling -- W tiles 32 k x 16 n, 256
// workgroups, G workgroups of 16 rows -- was built in round 3, verified and measured slower, 36.1-36.2 us per update against 35.8-35.9:
// 351 workgroups on 256 CUs double up, and the head's partials every workgroup re-reads do not shrink with the tile.  Removed in round 5.)
constexpr int GR_WN = 32;
enum { GR_NTW = 512 / GR_WN, GR_NW = KT * GR_NTW, GR_NG = 16, GR_GROWS = 512 / GR_NG, GR_GU = GR_GROWS / 4, GR_NR = (H1N + 3) / 4 };

constexpr int GR_PS = BP + 4;                                // W: row stride of the two operand panels (b128 reads of 16 rows: conflict free)
constexpr int GR_BT = GR_WN * GR_PS;                         // W: [GR_WN n][132] D2 panel
constexpr int GR_AT = 32 * GR_PS;                            // W: [32 k][132] h1 panel
constexpr int GR_LDS = (GR_BT + GR_AT + W1K * BP + W1K * 32 + AIN * BP + 2 * 32 + 8 + GR_GROWS * 3) * 4;

// D2 element: (sum_o W3[n][o] d3[o][m]) * (h2 > 0)
__device__ __forceinline__ float d2_val(float h2, float w3a, float w3b, float d3a, float d3b)
{
    const float g = fmaf(w3b, d3b, w3a * d3a);
    return h2 > 0.0f ? g : 0.0f;
}

// One wave, one hidden unit k.  Everything a row needs comes straight from global memory in ONE batch of loads per lane -- its W1
// column and b1 (owned by this wave: nobody else touches them), the NQ partial slabs of each E and the two input columns (lane,
// lane + 64) of the network input, which for both differentiated networks is plain workspace data -- so there is no LDS block
// beyond the error signal and one exposed latency.
template <int IN, int OUT> struct L1Row { float w[IN]; float b; float e[OUT][2][NQ]; float x[2][IN]; };
template <int IN, int OUT>
__device__ __forceinline__ void l1row_load(const GradArgs &A, int k, int lane, L1Row<IN, OUT> &R)
{
#pragma unroll
    for (int j = 0; j < IN; ++j) R.w[j] = A.P[j * H1N + k];
    R.b = A.P[off_b1(IN) + k];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m = lane + 64 * h;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            R.e[0][h][q] = A.E0[((int64_t)q * H1N + k) * BP + m];
            if constexpr (OUT == 2) R.e[1][h][q] = A.E1[((int64_t)q * H1N + k) * BP + m];
        }
#pragma unroll
        for (int j = 0; j < IN; ++j) R.x[h][j] = j < SIN ? A.x.X[j * BP + m] : A.x.A[(j - SIN) * BP + m];
    }
}
template <int IN, int OUT>
__device__ __forceinline__ void l1row_wave(const L1Row<IN, OUT> &R, const float *d3 /*LDS [2][BP]*/, int lane, float (&gout)[IN + 1])
{
    float dv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m = lane + 64 * h;
        float pre = R.b;
#pragma unroll
        for (int j = 0; j < IN; ++j) pre = fmaf(R.w[j], R.x[h][j], pre);
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) { s0 += R.e[0][h][q]; if constexpr (OUT == 2) s1 += R.e[1][h][q]; }
        float g = d3[m] * s0;
        if (OUT == 2) g = fmaf(d3[BP + m], s1, g);
        dv[h] = pre > 0.0f ? g : 0.0f;
    }
    gout[IN] = wave_sum(dv[0] + dv[1]);
#pragma unroll
    for (int j = 0; j < IN; ++j) gout[j] = wave_sum(R.x[0][j] * dv[0] + R.x[1][j] * dv[1]);
}

template <int IN, int OUT>
__device__ __forceinline__ void grad_body(const GradArgs &A, float *smem, const int bx)
{
    // bx: this workgroup's index within the gradient grid
    float *Bt = smem;                          // W: [32 n][GR_PS] D2 panel
    float *At = Bt + GR_BT;                    // W: [32 k][GR_PS] h1 panel
    float *xs = At + GR_AT;                    // [12][BP]
    float *w1 = xs + W1K * BP;                 // W: w1m columns of the k-tile, [12][32]
    float *d3 = w1 + W1K * 32;                 // [2][BP]
    float *w3s = d3 + AIN * BP;                // W3 rows of the n-tile (W) / of the 32 rows (G), [32][2] (out == 1: [.][1] zero)
    float *red = w3s + 2 * 32;                 // [8]
    float *gbuf = red + 8;                     // G: [GR_GROWS rows][3] sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const bool is_w = bx < GR_NW, is_g = !is_w && bx < GR_NW + GR_NG;
    const int kt = bx / GR_NTW, nt = bx % GR_NTW;                  // W: (k-tile of 32, n-tile of GR_WN)
    const int mcol = tid & 127, half = tid >> 7;
    const AdamCtx *fz = A.fuse ? &A.c : nullptr;
    const bool publisher = bx == 0;
    constexpr int kRegion = OUT == 1 ? 2 : 4;
    (void)kRegion;
    STAMP(kRegion, 0);

    if (!is_w && !is_g) {
        // ---- R: layer-1 rows ----
        const int k = (bx - GR_NW - GR_NG) * 4 + wave;
        HeadRegs hr;
        L1Row<IN, OUT> R;
        l1row_load<IN, OUT>(A, min(k, H1N - 1), lane, R);
        if (A.head == 1) head_loss_load(A.dd, hr); else head_actor_load(A.dd, hr);
        STAMP(kRegion, 1);
        if (A.head == 1) head_loss(A.dd, hr, d3, red, false, nullptr); else head_actor(A.dd, hr, d3, red, false, nullptr);
        __syncthreads();
        STAMP(kRegion, 2);
        if (k >= H1N) return;
        float gout[IN + 1];
        l1row_wave<IN, OUT>(R, d3, lane, gout);
        STAMP(kRegion, 3);
        float mine = 0.0f;
#pragma unroll
        for (int j = 0; j <= IN; ++j) mine = lane == j ? gout[j] : mine;
        if (lane <= IN) {
            const int e = lane < IN ? lane * H1N + k : off_b1(IN) + k;
            A.grad[e] = mine;
            if (fz) adam_elem(*fz, e, mine);
        }
        STAMP(kRegion, 4);
        return;
    }

    // The H2 panel does not depend on the error signal: its loads go out before the head is evaluated, so the two global latencies
    // overlap instead of following each other.
    constexpr int HVN = GR_WN / 2;              // H2 rows per thread
    constexpr int WOWN = 4;                     // elements of gW2 a lane finishes and owns
    float hv[HVN];
    int widx[WOWN];
#pragma unroll
    for (int r = 0; r < WOWN; ++r) widx[r] = -1;
    AdamRegs<WOWN> ar;
    if (is_w) {
#pragma unroll
        for (int u = 0; u < HVN; ++u) hv[u] = A.H2[min(nt * GR_WN + 2 * u + half, H2N - 1) * BP + mcol];
#pragma unroll
        for (int r = 0; r < WOWN; ++r) {
            // wave w finishes the 16 x 16 block (k half w >> 1, n half w & 1), D[i = 4 g + r][j = c], r < 4
            const int k = kt * 32 + 16 * (wave >> 1) + 4 * (lane >> 4) + r, n = nt * GR_WN + 16 * (wave & 1) + (lane & 15);
            widx[r] = (k < H1N && n < H2N) ? off_w2(IN) + k * H2N + n : -1;
        }
        adam_load<WOWN>(A.c, widx, ar);         // moments, parameter, target: requested with the first burst, consumed after the tile (requested
                                                // BEHIND the head's loads instead, round 4: 36.9 against 36.2 us per update -- slower)
    }
    XRegs<IN> xr;
    f32x4 wq = {0.f, 0.f, 0.f, 0.f};
    float w3v = 0.0f;
    const int nrow0 = is_w ? nt * GR_WN : (bx - GR_NW) * GR_GROWS;               // first of the W3 rows this workgroup needs (GR_WN / GR_GROWS of them)
    if (is_w) {
        build_x_load<IN>(A.x, xr);
        const int t = min(tid, 95);                                                           // 12 rows x 8 float4: the k-tile's 32 image columns
        wq = *reinterpret_cast<const f32x4 *>(A.w1t + (t >> 3) * W1C + kt * 32 + 4 * (t & 7));
    }
    {
        const int t = min(tid, 2 * 32 - 1), o = t & 1;                                        // (W and G need the same number of rows)
        w3v = A.w3f[(nrow0 + (t >> 1)) * OUT + min(o, OUT - 1)];                              // frozen copy: 512 rows, zero padded
    }
    HeadRegs hr;
    if (A.head == 1) head_loss_load(A.dd, hr); else head_actor_load(A.dd, hr);
    if (is_w) {
        build_x_store<IN>(A.x, xr, xs, false);
        if (tid < 96) *reinterpret_cast<f32x4 *>(w1 + (tid >> 3) * 32 + 4 * (tid & 7)) = wq;
    }
    if (A.head == 1) head_loss(A.dd, hr, d3, red, publisher, fz); else head_actor(A.dd, hr, d3, red, publisher, fz);
    if (tid < 2 * 32) w3s[tid] = (tid & 1) < OUT ? w3v : 0.0f;
    STAMP(kRegion, 1);
    __syncthreads();
    STAMP(kRegion, 2);
    const float d3a = d3[mcol], d3b = d3[BP + mcol];

    if (is_w) {
        // h1 panel At[kl][m] (row stride GR_PS): this wave's 32 batch columns of the k-tile, layer 1 on the matrix pipe.  Its MFMA chain is
        // issued first so that it runs under the VALU work of the D2 panel below.
        const f32x16 t = l1_tile<32>(w1, xs, 0, wave * 32, li, lh);
        // D2 panel Bt[nl][m] (row stride GR_PS), nl = 2*u + half: 16 rows per thread from the loads issued above (W3 comes from LDS;
        // rows >= 500 meet the zero rows of its image)
        {
#pragma unroll
            for (int u = 0; u < HVN; ++u) {
                const int nl = 2 * u + half;
                const float2 w = *reinterpret_cast<const float2 *>(w3s + 2 * nl);
                Bt[nl * GR_PS + mcol] = d2_val(hv[u], w.x, w.y, d3a, d3b);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) At[((r & 3) + 8 * (r >> 2) + 4 * lh) * GR_PS + wave * 32 + li] = fmaxf(t[r], 0.0f);
        STAMP(kRegion, 3);
        __syncthreads();
        STAMP(kRegion, 4);
        // wave w: the 16 x 16 block (k half w >> 1, n half w & 1) of the tile over the WHOLE batch: 32 steps of v_mfma_f32_16x16x4_f32.
        // The contraction index may be visited in any order as long as both operands agree: lane (g, c) takes the batch columns
        // 16 q + 4 g + e (q < 8, e < 4) -- one b128 read per operand and q -- and step (q, e) contracts {16 q + 4 g' + e : g' < 4}.
        // Four accumulators (one per e), added at the end.  Nothing is exchanged between waves: each owns its 256 elements
        // (D[i = 4 g + r][j = c], 4 per lane) for ADAM.
        f32x4 blk;
        {
            const int g = lane >> 4, c = lane & 15;
            const float *pa = At + (16 * (wave >> 1) + c) * GR_PS + 4 * g, *pb = Bt + (16 * (wave & 1) + c) * GR_PS + 4 * g;
            f32x4 av[8], bv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                av[q] = *reinterpret_cast<const f32x4 *>(pa + 16 * q);
                bv[q] = *reinterpret_cast<const f32x4 *>(pb + 16 * q);
            }
            f32x4 acc[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][e], bv[q][e], acc[e], 0, 0, 0);
            blk = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
        STAMP(kRegion, 5);
        {
            float *gW2 = A.grad + off_w2(IN);
            float val[WOWN];
#pragma unroll
            for (int r = 0; r < WOWN; ++r) {
                val[r] = blk[r];
                if (widx[r] >= 0) gW2[widx[r] - off_w2(IN)] = val[r];
            }
            STAMP(kRegion, 6);
            if (fz) adam_apply<WOWN>(*fz, widx, val, ar);
            STAMP(kRegion, 7);
        }
    } else {
        // gb2[n] = sum_m D2[n][m]; gW3[n][o] = sum_m h2[n][m] d3[o][m]: GR_NG workgroups x GR_GROWS rows, one wave per row, all of a
        // wave's rows in flight
        const int g = bx - GR_NW;
        const float e0a = d3[lane], e0b = d3[64 + lane], e1a = d3[BP + lane], e1b = d3[BP + 64 + lane];
        {
            float h0[GR_GU], h1[GR_GU];
#pragma unroll
            for (int u = 0; u < GR_GU; ++u) {
                const int nc = min(g * GR_GROWS + wave + 4 * u, H2N - 1);
                h0[u] = A.H2[nc * BP + lane];
                h1[u] = A.H2[nc * BP + 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < GR_GU; ++u) {
                const int rl = wave + 4 * u;
                const float2 w = *reinterpret_cast<const float2 *>(w3s + 2 * rl);
                const float s0 = wave_sum(h0[u] * e0a + h1[u] * e0b);
                const float s1 = wave_sum(h0[u] * e1a + h1[u] * e1b);
                const float sb = wave_sum(d2_val(h0[u], w.x, w.y, e0a, e1a) + d2_val(h1[u], w.x, w.y, e0b, e1b));
                if (lane == 0) { gbuf[rl * 3 + 0] = sb; gbuf[rl * 3 + 1] = s0; gbuf[rl * 3 + 2] = s1; }
            }
        }
        STAMP(kRegion, 5);
        __syncthreads();
        if (tid < GR_GROWS * 3) {                   // one element per thread: (row, b2 | W3[.][0] | W3[.][1])
            const int rl = tid / 3, c = tid - rl * 3, nn = g * GR_GROWS + rl;
            if (nn < H2N && c - 1 < OUT) {
                const int e = c == 0 ? off_b2(IN) + nn : off_w3(IN) + nn * OUT + (c - 1);
                const float v = gbuf[tid];
                A.grad[e] = v;
                if (fz) adam_elem(*fz, e, v);
            }
        }
        STAMP(kRegion, 7);
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_grad(GradArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gshift(A, blockIdx.z * A.gstride);             // learner blockIdx.z (stride 0 for a single learner)
    if (A.in == SIN) grad_body<SIN, 2>(A, smem, (int)blockIdx.x); else grad_body<CIN, 1>(A, smem, (int)blockIdx.x);
}

// ---- min_max_buffer (MPS:50-53) -----------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_minmax(shems_replay ring, int64_t ring_len, int64_t count, uint64_t seed,
                                                 float *s_min, float *s_max, int64_t gstride)
{
    if (gstride) { const int64_t off = blockIdx.x * gstride; gshift(ring, off); s_min = gsh(s_min, off); s_max = gsh(s_max, off); seed += blockIdx.x; }
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

// ---- parameter noise (input.jl:210-215 ParamNoise; DDPG.jl:74-96) -----------------------------------------------
// add_perturb!: every parameter array of the copy gets the SAME scalar (sample_noise(pn, rng) re-seeds before each draw).
__global__ __launch_bounds__(256) void k_perturb(const float *__restrict__ src, float *__restrict__ dst, int64_t n, float shift)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i] + shift;
}

// acc = wa * acc + wg * g (wa == 0: acc is not read, so it may hold anything, NaNs included): gradients of the sub-batches of a minibatch
// wider than the 128 columns one update pass holds, each the MEAN over its sub-batch, weighted by sub-batch size / batch size
__global__ __launch_bounds__(256) void k_combine(float *__restrict__ acc, const float *__restrict__ g, int64_t n, float wa, float wg)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) acc[i] = wa == 0.0f ? wg * g[i] : fmaf(wa, acc[i], wg * g[i]);
}

// s of the minibatch the last update sampled (ring slots kept in the workspace) -> obs [batch][9]
__global__ __launch_bounds__(256) void k_batch_obs(shems_replay ring, const float *__restrict__ ws, int batch, float *__restrict__ obs)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= batch * SIN) return;
    const int m = t / SIN, k = t - m * SIN;
    const int64_t j = reinterpret_cast<const int32_t *>(ws + WS_IDX)[m];
    obs[t] = ring.s[j * SIN + k];
}

// distance = sqrt(Flux.mse(a, a_perturb)) (DDPG.jl:79): one workgroup, fixed summation order
__global__ __launch_bounds__(256) void k_action_distance(const float *__restrict__ a, const float *__restrict__ b, int64_t count, float *out)
{
    __shared__ float part[4];
    float acc = 0.0f;
    for (int64_t i = threadIdx.x; i < count; i += 256) { const float d = a[i] - b[i]; acc += d * d; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = sqrtf(((part[0] + part[1]) + (part[2] + part[3])) / (float)count);
}

constexpr int FWD_LDS = FwdShape<false>::LDS, QG_LDS = FWD_LDS > QG16_LDS ? FWD_LDS : QG16_LDS;
constexpr int MID_LDS = FWD_LDS > E_LDS ? FWD_LDS : E_LDS;
static_assert(QG_LDS <= 160 * 1024, "one workgroup's LDS");
static int set_lds_attrs()
{
    static std::atomic<uint64_t> m_fwd{0}, m_mid{0}, m_grad{0};          // per device: see lds_optin
    if (int rc = lds_optin(m_fwd, reinterpret_cast<const void *>(&k_fwd), QG_LDS, "attr k_fwd")) return rc;
    if (int rc = lds_optin(m_mid, reinterpret_cast<const void *>(&k_mid), MID_LDS, "attr k_mid")) return rc;
    if (int rc = lds_optin(m_grad, reinterpret_cast<const void *>(&k_grad), GR_LDS, "attr k_grad")) return rc;
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

static int check_group(const shems_group *g, const char *fn)
{
    if (!g || g->count < 1 || g->count > 65535 || g->stride_bytes < 0 || (g->stride_bytes & 15) != 0 || (g->count > 1 && g->stride_bytes == 0))
        return set_error(SHEMS_ERR_ARG, "%s: shems_group needs 1 <= count <= 65535 and a 16-byte-multiple stride", fn);
    return SHEMS_OK;
}

static int check_adam(double bp1, double bp2, const char *fn)
{
    if (!(bp1 > 0.0 && bp1 < 1.0 && bp2 > 0.0 && bp2 < 1.0)) return set_error(SHEMS_ERR_ARG, "%s: beta powers must be in (0,1)", fn);
    return SHEMS_OK;
}

struct AdamScalars { double eta, bp1, bp2, gscale; float *publish; };

static AdamCtx adam_ctx(const shems_ddpg *d, bool critic, const AdamScalars &s)
{
    const double k1 = s.eta / (1.0 - s.bp1), ic2 = 1.0 / (1.0 - s.bp2);
    return critic ? AdamCtx{d->critic, d->grad_critic, d->m_critic, d->v_critic, d->critic_t, nullptr, SHEMS_CRITIC_PARAMS, (int)CIN,
                            s.eta, s.bp1, s.bp2, s.gscale, k1, ic2, d->tau}
                  : AdamCtx{d->actor, d->grad_actor, d->m_actor, d->v_actor, d->actor_t, s.publish, SHEMS_ACTOR_PARAMS, (int)SIN,
                            s.eta, s.bp1, s.bp2, s.gscale, k1, ic2, d->tau};
}

// K1 + K2 + K3: the critic side of replay() (DDPG.jl:123-135).  fuse: K3 applies ADAM + the soft target update itself.
// Data-parallel form only: the actor's two E products, when shems_ddpg.flags asked critic_grad to leave them out of K2 so that they
// can run under the critic's gradient all-reduce (they need nothing the critic update produces).
static int actor_e_launch(const shems_ddpg *d, unsigned L, int64_t gs, hipStream_t st)
{
    float *ws = d->ws, *SA = slot(ws, SLOT_ACTOR);
    MidArgs m;
    std::memset(&m, 0, sizeof m);
    m.gstride = gs; m.nfwd = 0;
    m.e[0] = EJob{d->actor, SIN, 2, 0, SA + SL_H2, SA + SL_EP};
    m.e[1] = EJob{d->actor, SIN, 2, 1, SA + SL_H2, ws + WS_EA1};
    m.e[2] = m.e[1];
    hipLaunchKernelGGL(k_mid, dim3(2 * KT * NQ, 1, L), dim3(256), E_LDS, st, m);
    return hip_ok(hipGetLastError(), "k_mid (actor E) launch");
}

static int critic_side(const shems_ddpg *d, const shems_replay *ring, int64_t ring_len, uint64_t seed, uint32_t tick,
                       int64_t excl_pos, int64_t excl_count, unsigned L, int64_t gs, const AdamScalars *fuse, void *stream)
{
    if (int rc = check_ddpg(d, "shems_ddpg_critic_grad")) return rc;
    if (!ring || !ring->s || !ring->a || !ring->r || !ring->s2 || !ring->done || ring_len < 1 || ring_len > ring->capacity)
        return set_error(SHEMS_ERR_ARG, "shems_ddpg_critic_grad: bad replay ring / length");
    if (excl_count < 0 || excl_pos < 0 || (excl_count > 0 && (ring_len != ring->capacity || excl_count >= ring_len)))
        return set_error(SHEMS_ERR_ARG, "shems_ddpg_critic_grad_ex: an exclusion window needs a full ring and 0 <= count < capacity");
    hipStream_t st = (hipStream_t)stream;
    float *ws = d->ws;
    const XSrc none{nullptr, nullptr, nullptr, nullptr, nullptr};
    const XSrc x_s2a{ws + WS_X2T, nullptr, slot(ws, SLOT_ACTOR_T) + SL_P3, d->actor_t + off_b3(SIN, 2), nullptr};
    const XSrc x_sa{ws + WS_XT, ws + WS_AT, nullptr, nullptr, nullptr};
    float *SC = slot(ws, SLOT_CRITIC), *SA = slot(ws, SLOT_ACTOR);
    FwdArgs f;
    std::memset(&f, 0, sizeof f);
    f.gstride = gs;
    const unsigned fgx = NT * (BP / 32);
    const int flds = FWD_LDS;
    // K1: three independent forward passes; sample + gather + normalise + image packing + head freezing ride in this launch
    f.job[0] = FwdJob{nullptr, d->actor_t, SIN, 2, 0, none, nullptr, slot(ws, SLOT_ACTOR_T) + SL_P3, nullptr, nullptr};
    f.job[1] = FwdJob{nullptr, d->critic, CIN, 1, 1, none, SC + SL_H2, SC + SL_P3, nullptr, nullptr};
    f.job[2] = FwdJob{nullptr, d->actor, SIN, 2, 2, none, SA + SL_H2, SA + SL_P3, nullptr, nullptr};
    f.prep = 1;
    f.pa = PrepArgs{*d, *ring, ring_len, seed, tick, excl_pos, excl_count};
    hipLaunchKernelGGL(k_fwd, dim3(3 * (fgx + 6), 1, L), dim3(256), flds, st, f);      // + 6 publishing workgroups per job (see fwd_body)
    // K2: critic_target on [s'; actor_target(s')] | E of the critic | E of the actor's two outputs
    MidArgs m;
    std::memset(&m, 0, sizeof m);
    m.gstride = gs; m.nfwd = (int)fgx;
    m.fwd = FwdJob{w1t_of(ws, SLOT_CRITIC_T), d->critic_t, CIN, 1, 0, x_s2a, nullptr, slot(ws, SLOT_CRITIC_T) + SL_P3, nullptr, nullptr};
    m.e[0] = EJob{d->critic, CIN, 1, 0, SC + SL_H2, SC + SL_EP};
    m.e[1] = EJob{d->actor, SIN, 2, 0, SA + SL_H2, SA + SL_EP};
    m.e[2] = EJob{d->actor, SIN, 2, 1, SA + SL_H2, ws + WS_EA1};
    const bool defer_ea = !fuse && (d->flags & SHEMS_DDPG_DEFER_ACTOR_E) != 0;       // the caller runs shems_ddpg_actor_prepare later
    hipLaunchKernelGGL(k_mid, dim3(fgx + (defer_ea ? 1 : 3) * KT * NQ, 1, L), dim3(256), MID_LDS, st, m);
    // K3: critic gradient (+ ADAM + soft update)
    GradArgs g;
    std::memset(&g, 0, sizeof g);
    g.w1t = w1t_of(ws, SLOT_CRITIC); g.P = d->critic; g.in = CIN; g.out = 1; g.x = x_sa; g.H2 = SC + SL_H2; g.w3f = ws + WS_FW3C;
    g.grad = d->grad_critic; g.E0 = SC + SL_EP; g.E1 = nullptr; g.head = 1; g.fuse = fuse ? 1 : 0; g.dd = *d; g.gstride = gs;
    g.c = adam_ctx(d, true, fuse ? *fuse : AdamScalars{0, 0.5, 0.5, 1.0, nullptr});
    hipLaunchKernelGGL(k_grad, dim3(GR_NW + GR_NG + GR_NR, 1, L), dim3(256), GR_LDS, st, g);
    return hip_ok(hipGetLastError(), "ddpg critic-side launches");
}

// K4 + K5: the actor side (DDPG.jl:137-140), through the critic as it stands now (already updated)
static int actor_side(const shems_ddpg *d, unsigned L, int64_t gs, const AdamScalars *fuse, void *stream)
{
    if (int rc = check_ddpg(d, "shems_ddpg_actor_grad")) return rc;
    hipStream_t st = (hipStream_t)stream;
    float *ws = d->ws;
    // critic on [s; a_pi], a_pi = tanh(b3 + partials of the actor pass); workgroup 0 publishes a_pi for K5's head
    const XSrc x_spi{ws + WS_XT, nullptr, slot(ws, SLOT_ACTOR) + SL_P3, d->actor + off_b3(SIN, 2), ws + WS_API};
    const XSrc x_s{ws + WS_XT, nullptr, nullptr, nullptr, nullptr};
    float *C2 = slot(ws, SLOT_CRITIC2), *SA = slot(ws, SLOT_ACTOR);
    FwdArgs f;
    std::memset(&f, 0, sizeof f);
    f.gstride = gs;
    const unsigned fgx = NT16 * (BP / 32);
    f.prep = 2;
    f.job[0] = FwdJob{nullptr, d->critic, CIN, 1, 0, x_spi, nullptr, C2 + SL_P3, ws + WS_D3Q, ws + WS_DAP};
    GradArgs g;
    std::memset(&g, 0, sizeof g);
    g.w1t = w1t_of(ws, SLOT_ACTOR); g.P = d->actor; g.in = SIN; g.out = 2; g.x = x_s; g.H2 = SA + SL_H2; g.w3f = ws + WS_FW3A;
    g.grad = d->grad_actor; g.E0 = SA + SL_EP; g.E1 = ws + WS_EA1; g.head = 2; g.fuse = fuse ? 1 : 0; g.dd = *d; g.gstride = gs;
    g.c = adam_ctx(d, false, fuse ? *fuse : AdamScalars{0, 0.5, 0.5, 1.0, nullptr});
    hipLaunchKernelGGL(k_fwd, dim3(fgx, 1, L), dim3(256), QG16_LDS, st, f);
    hipLaunchKernelGGL(k_grad, dim3(GR_NW + GR_NG + GR_NR, 1, L), dim3(256), GR_LDS, st, g);
    return hip_ok(hipGetLastError(), "ddpg actor-side launches");
}

static int adam_launch(const shems_ddpg *d, bool critic, const AdamScalars &s, hipStream_t st, unsigned L, int64_t gs)
{
    if (int rc = check_adam(s.bp1, s.bp2, "adam")) return rc;
    const AdamCtx c = adam_ctx(d, critic, s);
    hipLaunchKernelGGL(k_adam_soft, dim3((c.n + 1023) / 1024, 1, L), dim3(256), 0, st, c, gs);
    return hip_ok(hipGetLastError(), "k_adam_soft launch");
}

namespace shems {
int adam_soft_sweep(float *p, const float *g, float *m, float *v, float *target, float *publish, int n, double eta, double bp1, double bp2,
                    double gscale, float tau, hipStream_t st)
{
    if (int rc = check_adam(bp1, bp2, "adam_soft_sweep")) return rc;
    if (!p || !g || !m || !v || !target || n < 1) return set_error(SHEMS_ERR_ARG, "adam_soft_sweep: bad buffers");
    for (const void *q : {(const void *)p, (const void *)g, (const void *)m, (const void *)v, (const void *)target, (const void *)publish})
        if (((uintptr_t)q & 15) != 0) return set_error(SHEMS_ERR_ARG, "adam_soft_sweep: buffers must be 16-byte aligned");
    const AdamCtx c{p, g, m, v, target, publish, n, 0, eta, bp1, bp2, gscale, eta / (1.0 - bp1), 1.0 / (1.0 - bp2), tau};
    hipLaunchKernelGGL(k_adam_soft, dim3((n + 1023) / 1024, 1, 1), dim3(256), 0, st, c, (int64_t)0);
    return hip_ok(hipGetLastError(), "k_adam_soft launch");
}
}  // namespace shems

extern "C" {

#ifdef SHEMS_STAMP
/* diagnostic build only: d_buf = device buffer of 5 * 1024 * 16 * 2 uint64 (or NULL to switch the stamps off) */
int shems_debug_set_stamps(void *d_buf)
{
    unsigned long long *p = (unsigned long long *)d_buf;
    return hip_ok(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof p), "set stamps");
}
#endif

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

/* replay() in one call, single replica: K1..K5 with ADAM + soft updates inside the gradient launches. */
int shems_ddpg_update(const shems_ddpg *d, const shems_replay *ring, int64_t ring_len, uint64_t seed, uint32_t tick,
                      int64_t excl_pos, int64_t excl_count, double eta_crit, double bp1_crit, double bp2_crit,
                      double eta_act, double bp1_act, double bp2_act, float *d_publish, void *stream)
{
    if (int rc = check_adam(bp1_crit, bp2_crit, "shems_ddpg_update")) return rc;
    if (int rc = check_adam(bp1_act, bp2_act, "shems_ddpg_update")) return rc;
    const AdamScalars sc{eta_crit, bp1_crit, bp2_crit, 1.0, nullptr}, sa{eta_act, bp1_act, bp2_act, 1.0, d_publish};
    if (int rc = critic_side(d, ring, ring_len, seed, tick, excl_pos, excl_count, 1, 0, &sc, stream)) return rc;
    return actor_side(d, 1, 0, &sa, stream);
}

}  // extern "C"

namespace shems {
int ddpg_apply_xchg(const shems_ddpg *d, bool critic, double eta, double bp1, double bp2, float *d_publish, const XchgArgs &x, hipStream_t st)
{
    if (int rc = check_ddpg(d, "ddpg_apply_xchg")) return rc;
    if (int rc = check_adam(bp1, bp2, "ddpg_apply_xchg")) return rc;
    if (x.world < 1 || x.world > kXchgMaxWorld || x.rank < 0 || x.rank >= x.world || x.epoch < 1 || !x.timeouts || !x.poison || x.wait_ticks < 1)
        return set_error(SHEMS_ERR_ARG, "ddpg_apply_xchg: bad exchange record");
    static_assert(SHEMS_ACTOR_PARAMS <= kXchgNmax && SHEMS_CRITIC_PARAMS <= kXchgNmax, "inbox slot holds either gradient");
    const AdamCtx c = adam_ctx(d, critic, AdamScalars{eta, bp1, bp2, 1.0 / (double)x.world, d_publish});
    hipLaunchKernelGGL(k_adam_xchg, dim3((c.n + 1023) / 1024), dim3(256), 0, st, c, x);
    return hip_ok(hipGetLastError(), "k_adam_xchg launch");
}
}  // namespace shems

extern "C" {

int shems_ddpg_group_update(const shems_ddpg *d0, const shems_replay *ring0, const shems_group *g, int64_t ring_len, uint64_t seed,
                            uint32_t tick, double eta_crit, double bp1_crit, double bp2_crit, double eta_act, double bp1_act,
                            double bp2_act, void *stream)
{
    if (int rc = check_group(g, "shems_ddpg_group_update")) return rc;
    if (int rc = check_adam(bp1_crit, bp2_crit, "shems_ddpg_group_update")) return rc;
    if (int rc = check_adam(bp1_act, bp2_act, "shems_ddpg_group_update")) return rc;
    const AdamScalars sc{eta_crit, bp1_crit, bp2_crit, 1.0, nullptr}, sa{eta_act, bp1_act, bp2_act, 1.0, nullptr};
    const int64_t gs = g->count > 1 ? g->stride_bytes : 0;
    if (int rc = critic_side(d0, ring0, ring_len, seed, tick, 0, 0, (unsigned)g->count, gs, &sc, stream)) return rc;
    return actor_side(d0, (unsigned)g->count, gs, &sa, stream);
}

int shems_ddpg_critic_grad(const shems_ddpg *d, const shems_replay *ring, int64_t ring_len, uint64_t seed, uint32_t tick,
                           void *stream)
{
    return critic_side(d, ring, ring_len, seed, tick, 0, 0, 1, 0, nullptr, stream);
}

int shems_ddpg_critic_grad_ex(const shems_ddpg *d, const shems_replay *ring, int64_t ring_len, uint64_t seed, uint32_t tick,
                              int64_t excl_pos, int64_t excl_count, void *stream)
{
    return critic_side(d, ring, ring_len, seed, tick, excl_pos, excl_count, 1, 0, nullptr, stream);
}

int shems_ddpg_group_critic_grad(const shems_ddpg *d0, const shems_replay *ring0, const shems_group *g, int64_t ring_len,
                                 uint64_t seed, uint32_t tick, void *stream)
{
    if (int rc = check_group(g, "shems_ddpg_group_critic_grad")) return rc;
    return critic_side(d0, ring0, ring_len, seed, tick, 0, 0, (unsigned)g->count, g->count > 1 ? g->stride_bytes : 0, nullptr, stream);
}

int shems_ddpg_actor_prepare(const shems_ddpg *d, void *stream)
{
    if (int rc = check_ddpg(d, "shems_ddpg_actor_prepare")) return rc;
    return actor_e_launch(d, 1, 0, (hipStream_t)stream);
}

int shems_ddpg_critic_apply(const shems_ddpg *d, double eta, double bp1, double bp2, double grad_scale, void *stream)
{
    if (int rc = check_ddpg(d, "shems_ddpg_critic_apply")) return rc;
    return adam_launch(d, true, AdamScalars{eta, bp1, bp2, grad_scale, nullptr}, (hipStream_t)stream, 1, 0);
}

int shems_ddpg_group_critic_apply(const shems_ddpg *d, const shems_group *g, double eta, double bp1, double bp2, void *stream)
{
    if (int rc = check_ddpg(d, "shems_ddpg_group_critic_apply")) return rc;
    if (int rc = check_group(g, "shems_ddpg_group_critic_apply")) return rc;
    return adam_launch(d, true, AdamScalars{eta, bp1, bp2, 1.0, nullptr}, (hipStream_t)stream, (unsigned)g->count,
                       g->count > 1 ? g->stride_bytes : 0);
}

int shems_ddpg_actor_grad(const shems_ddpg *d, void *stream) { return actor_side(d, 1, 0, nullptr, stream); }

int shems_ddpg_group_actor_grad(const shems_ddpg *d0, const shems_group *g, void *stream)
{
    if (int rc = check_group(g, "shems_ddpg_group_actor_grad")) return rc;
    return actor_side(d0, (unsigned)g->count, g->count > 1 ? g->stride_bytes : 0, nullptr, stream);
}

int shems_ddpg_group_actor_apply(const shems_ddpg *d, const shems_group *g, double eta, double bp1, double bp2, void *stream)
{
    if (int rc = check_ddpg(d, "shems_ddpg_group_actor_apply")) return rc;
    if (int rc = check_group(g, "shems_ddpg_group_actor_apply")) return rc;
    return adam_launch(d, false, AdamScalars{eta, bp1, bp2, 1.0, nullptr}, (hipStream_t)stream, (unsigned)g->count,
                       g->count > 1 ? g->stride_bytes : 0);
}

int shems_ddpg_actor_apply_pub(const shems_ddpg *d, double eta, double bp1, double bp2, double grad_scale, float *d_publish,
                               void *stream)
{
    if (int rc = check_ddpg(d, "shems_ddpg_actor_apply")) return rc;
    return adam_launch(d, false, AdamScalars{eta, bp1, bp2, grad_scale, d_publish}, (hipStream_t)stream, 1, 0);
}

int shems_ddpg_actor_apply(const shems_ddpg *d, double eta, double bp1, double bp2, double grad_scale, void *stream)
{
    return shems_ddpg_actor_apply_pub(d, eta, bp1, bp2, grad_scale, nullptr, stream);
}

int shems_ddpg_perturb_dev(const float *d_params, float *d_perturbed, int64_t n, float shift, void *stream)
{
    if (!d_params || !d_perturbed || n < 1) return set_error(SHEMS_ERR_ARG, "shems_ddpg_perturb_dev: bad arguments");
    hipLaunchKernelGGL(k_perturb, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_params, d_perturbed, n, shift);
    return hip_ok(hipGetLastError(), "k_perturb launch");
}

int shems_ddpg_combine_dev(float *d_acc, const float *d_g, int64_t n, float w_acc, float w_g, void *stream)
{
    if (!d_acc || !d_g || n < 1) return set_error(SHEMS_ERR_ARG, "shems_ddpg_combine_dev: bad arguments");
    hipLaunchKernelGGL(k_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_acc, d_g, n, w_acc, w_g);
    return hip_ok(hipGetLastError(), "k_combine launch");
}

int shems_ddpg_batch_obs_dev(const shems_ddpg *d, const shems_replay *ring, float *d_obs, void *stream)
{
    if (int rc = check_ddpg(d, "shems_ddpg_batch_obs_dev")) return rc;
    if (!ring || !ring->s || !d_obs) return set_error(SHEMS_ERR_ARG, "shems_ddpg_batch_obs_dev: bad arguments");
    hipLaunchKernelGGL(k_batch_obs, dim3((d->batch * SIN + 255) / 256), dim3(256), 0, (hipStream_t)stream, *ring, d->ws, d->batch, d_obs);
    return hip_ok(hipGetLastError(), "k_batch_obs launch");
}

int shems_action_distance_dev(const float *d_a, const float *d_b, int64_t count, float *d_out, void *stream)
{
    if (!d_a || !d_b || !d_out || count < 1) return set_error(SHEMS_ERR_ARG, "shems_action_distance_dev: bad arguments");
    hipLaunchKernelGGL(k_action_distance, dim3(1), dim3(256), 0, (hipStream_t)stream, d_a, d_b, count, d_out);
    return hip_ok(hipGetLastError(), "k_action_distance launch");
}

int shems_minmax_dev(const shems_replay *ring, int64_t ring_len, int64_t count, uint64_t seed, float *d_s_min, float *d_s_max,
                     void *stream)
{
    if (!ring || !ring->s || ring_len < 1 || ring_len > ring->capacity || count < 1 || !d_s_min || !d_s_max)
        return set_error(SHEMS_ERR_ARG, "shems_minmax_dev: bad arguments");
    hipLaunchKernelGGL(k_minmax, dim3(1), dim3(1024), 0, (hipStream_t)stream, *ring, ring_len, count, seed, d_s_min, d_s_max, (int64_t)0);
    return hip_ok(hipGetLastError(), "k_minmax launch");
}

int shems_minmax_group_dev(const shems_replay *ring, const shems_group *g, int64_t ring_len, int64_t count, uint64_t seed,
                           float *d_s_min, float *d_s_max, void *stream)
{
    if (int rc = check_group(g, "shems_minmax_group_dev")) return rc;
    if (!ring || !ring->s || ring_len < 1 || ring_len > ring->capacity || count < 1 || !d_s_min || !d_s_max)
        return set_error(SHEMS_ERR_ARG, "shems_minmax_group_dev: bad arguments");
    hipLaunchKernelGGL(k_minmax, dim3((unsigned)g->count), dim3(1024), 0, (hipStream_t)stream, *ring, ring_len, count, seed, d_s_min,
                       d_s_max, g->count > 1 ? g->stride_bytes : (int64_t)0);
    return hip_ok(hipGetLastError(), "k_minmax launch");
}

}  // extern "C"
