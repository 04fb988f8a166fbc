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
//   * no single-workgroup latency chains: every phase is spread over 64-144 workgroups, shaped so that no workgroup type
//     is the straggler of its launch (the 8 x 64-row gb2/gW3 workgroups once were: -4 us for halving them);
//   * operands are staged into LDS with wide, independent loads, ALL of a phase's global loads issued before the first
//     wait (one exposed latency per phase), never fetched per MFMA k-step; phases that only one wave finishes read
//     their LDS constants in one batch;
//   * nothing derivable is stored: layer-1 activations, their relu masks and the back-propagated
//     layer-2 error D2 = (W3 d3) .* (h2 > 0) are recomputed inside the kernels that consume them;
//   * cross-workgroup reductions go through partial slabs summed in a fixed order (bitwise
//     reproducible; no float atomics, no device-scope fences -- on this 8-XCD part a release writes the L2 back).
// 8 launches per update on a single replica (the loss / actor heads run in the prologue of the bwd workgroups, the
// minibatch sampling / gather and the layer-1 image packing inside the first forward launch, the layer-1 gradient rows
// inside the ADAM launch):
//   fwd(actor_t) -> fwd(critic_t | critic | actor) -> bwd(critic) -> adam+soft(critic)
//   fwd(critic on [s; actor(s)]) -> bwd(input grad) -> bwd(actor) -> adam+soft(actor)
// and 10 when replicas exchange gradients (the all-reduce needs the complete gradient before ADAM):
//   ... bwd(critic) -> l1bwd(critic) [all-reduce] adam+soft(critic) ... bwd(actor) -> l1bwd(actor) [all-reduce] adam+soft(actor)
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
constexpr int NQ = 8;              // blocks of the n range for the backward tiles
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
constexpr int64_t WS_DAP = WS_IDX + BP;               // [KT][NQ][2][BP]  partial d loss / d a_pi
constexpr int64_t WS_W1T = WS_DAP + KT * NQ * AIN * BP;   // [4 nets][12][256] packed layer-1 images (see w1m below)
constexpr int64_t WS_SLOT0 = WS_W1T + 4 * 12 * 256;
constexpr int64_t SL_H2 = 0;                          // [500][BP]          relu(W2' h1 + b2)
constexpr int64_t SL_P3 = SL_H2 + H2N * BP;           // [NT][2][BP]        per-n-tile partial sums of layer 3
constexpr int64_t SL_D1P = SL_P3 + NT * 2 * BP;       // [NQ][250][BP]      partial (unmasked) error at layer 1
constexpr int64_t SL_SIZE = SL_D1P + NQ * H1N * BP;
enum { SLOT_ACTOR_T = 0, SLOT_CRITIC_T = 1, SLOT_CRITIC = 2, SLOT_ACTOR = 3, SLOT_CRITIC2 = 4, N_SLOTS = 5 };
__host__ __device__ inline float *w1t_of(float *ws, int net) { return ws + WS_W1T + (int64_t)net * 12 * 256; }   // net = SLOT_* < 4
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

// Learner groups (shems_group): learner l's copy of every device buffer is learner 0's pointer + l * stride bytes.
template <class T>
__device__ __forceinline__ T *gsh(T *p, int64_t off) { return p ? reinterpret_cast<T *>(reinterpret_cast<uintptr_t>(p) + off) : p; }
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
template <int IN> struct XRegs { float v[5]; float a; float b3v; float p[NT]; };
template <int IN>
__device__ __forceinline__ void build_x_load(const XSrc &s, XRegs<IN> &R)
{
#pragma unroll
    for (int it = 0; it < 5; ++it) { const int e = it * 256 + threadIdx.x; R.v[it] = s.X[min(e, SIN * BP - 1)]; }   // clamped, never predicated:
    // a guarded load becomes a branch + its own s_waitcnt, which serialises the batch
    if (IN == CIN) {
        const int e = threadIdx.x, o = e / BP, m = e - o * BP;
        if (s.A) {
            R.a = s.A[e];
        } else {
            R.b3v = s.b3[o];
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
            a = R.a;
        } else {
            float acc = R.b3v;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc += R.p[t];
            a = tanhf(acc);                                       // Dense(500, 2, tanh)
            if (publisher && s.publish) s.publish[e] = a;
        }
        xs[SIN * BP + e] = a;
    }
}
template <int IN>
__device__ __forceinline__ void build_x(const XSrc &s, float *xs /*LDS [IN][BP]*/, bool publisher)
{
    XRegs<IN> R;
    build_x_load<IN>(s, R);
    build_x_store<IN>(s, R, xs, publisher);
}

// Layer 1 also runs on the matrix pipe: pre[k][m] = sum_j w1m[j][k] * xs[j][m] with K = 12 = 6 MFMA k-steps, where
//   w1m [12][256] = rows 0..in-1: W1[j][k]; row 11: b1[k]; everything else (rows in..10, columns 250..255) zero
//   xs  [12][BP]  = rows 0..in-1: the network input; row 11: 1.0 (bias); rows in..10 zero.
// The packed image w1m lives in the workspace (the update's first launch builds it for all four networks, the
// critic's ADAM launch refreshes the critic's); staging it is a straight 12 KB float4 copy, three loads per thread.
constexpr int W1K = 12, W1C = 256;
struct W1mRegs { float4 v[3]; };
__device__ __forceinline__ void stage_w1m_load(const float *__restrict__ g, W1mRegs &R)
{
#pragma unroll
    for (int it = 0; it < 3; ++it) R.v[it] = reinterpret_cast<const float4 *>(g)[it * 256 + threadIdx.x];
}
__device__ __forceinline__ void stage_w1m_store(const W1mRegs &R, float *l)
{
#pragma unroll
    for (int it = 0; it < 3; ++it) reinterpret_cast<float4 *>(l)[it * 256 + threadIdx.x] = R.v[it];
}
__device__ __forceinline__ void stage_w1m(const float *__restrict__ g, float *l)
{
    W1mRegs R;
    stage_w1m_load(g, R);
    stage_w1m_store(R, l);
}
__device__ __forceinline__ void pack_w1m(const float *__restrict__ P, int in, float *__restrict__ g)
{
    float v[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {                       // thread = column k (blockDim 256), unconditional clamped loads
        const int k = min((int)threadIdx.x, H1N - 1);
        v[j] = P[(j == W1K - 1 ? in : min(j, in - 1)) * H1N + k];          // row `in` of the block is b1
    }
#pragma unroll
    for (int j = 0; j < 12; ++j)
        g[j * W1C + threadIdx.x] = ((j < in || j == W1K - 1) && (int)threadIdx.x < H1N) ? v[j] : 0.0f;
}
// One 32(k) x 32(m) tile of layer-1 pre-activations, D layout (row k = (r&3)+8(r>>2)+4*lh, column m = lane&31).
__device__ __forceinline__ f32x16 l1_tile(const float *w1m, const float *xs, int kbase, int mbase, int li, int lh)
{
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = 0.0f;
    float a[W1K / 2], b[W1K / 2];                 // all 12 operand reads in one batch: with one wave per SIMD nothing else hides them
#pragma unroll
    for (int s = 0; s < W1K / 2; ++s) {
        const int j = 2 * s + lh;
        a[s] = w1m[j * W1C + kbase + li];
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
// normalises.  xs2 (LDS [.][BP], may be null) receives normalize(s'); when `publish`, everything later launches read goes to the
// workspace: normalize(s), normalize(s'), a, r, done, d(-mean q)/dq and the sampled slots.
__device__ __forceinline__ void prep_column(const PrepArgs &A, int m, float *xs2, bool publish)
{
    const shems_ddpg &d = A.d;
    const shems_replay &ring = A.ring;
    float *ws = d.ws;
    float s[SIN], s2[SIN], a0 = 0.f, a1 = 0.f, r = 0.f, dn = 0.f;
    int64_t j = -1;
    const bool live = m < d.batch;
    if (live) {
        const u32x4 x = philox4x32_10((uint32_t)(m >> 2), 0u, A.tick, kStreamSample, (uint32_t)A.seed, (uint32_t)(A.seed >> 32));
        const uint32_t w = (m & 3) == 0 ? x.x : (m & 3) == 1 ? x.y : (m & 3) == 2 ? x.z : x.w;
        j = (int64_t)(w % (uint32_t)(A.ring_len - A.excl_count));
        if (A.excl_count > 0) j = (A.excl_pos + A.excl_count + j) % ring.capacity;     // skip the window another stream is writing
#pragma unroll
        for (int k = 0; k < SIN; ++k) s2[k] = ring.s2[j * SIN + k];
        if (publish) {
#pragma unroll
            for (int k = 0; k < SIN; ++k) s[k] = ring.s[j * SIN + k];
            a0 = ring.a[j * 2]; a1 = ring.a[j * 2 + 1];
            r = ring.r[j];
            dn = ring.done[j] ? 1.0f : 0.0f;
        }
    }
#pragma unroll
    for (int k = 0; k < SIN; ++k) {
        const float lo = d.s_min[k], den = (d.s_max[k] - lo) + 1e-8f;                 // MPS:56
        const float x2 = live ? (s2[k] - lo) / den : 0.0f;
        if (xs2) xs2[k * BP + m] = x2;
        if (publish) {
            ws[WS_XT + k * BP + m] = live ? (s[k] - lo) / den : 0.0f;
            ws[WS_X2T + k * BP + m] = x2;
        }
    }
    if (publish) {
        ws[WS_AT + m] = a0; ws[WS_AT + BP + m] = a1;
        ws[WS_R + m] = r; ws[WS_DONE + m] = dn;
        ws[WS_D3Q + m] = live ? -1.0f / (float)d.batch : 0.0f;                        // d(-mean q)/dq
        reinterpret_cast<int32_t *>(ws + WS_IDX)[m] = (int32_t)j;
    }
}

// ---- kernel B: layers 1+2 forward for one 32-wide n-tile and all 128 columns --------------------------
struct FwdJob {
    const float *w1t;      // packed layer-1 image [250][12] of this network
    const float *P;        // parameter block
    int in;                // 9 (actor nets) or 11 (critic nets)
    int out;               // 2 or 1
    XSrc x;
    float *H2;             // [500][BP] or null (target nets: nothing downstream needs it)
    float *P3;             // [NT][2][BP] layer-3 partials of this n-tile
};
struct FwdArgs { FwdJob job[3]; int64_t gstride; int prep; int mt; PrepArgs pa; };    // prep: this launch opens the update (see fwd_body)
__device__ __forceinline__ void gshift(FwdJob &J, int64_t off)
{
    J.w1t = gsh(J.w1t, off); J.P = gsh(J.P, off); gshift(J.x, off); J.H2 = gsh(J.H2, off); J.P3 = gsh(J.P3, off);
}

// Workgroup = one 32-wide n-tile x MT columns of the batch.  K is always cut into four quarters of 64 hidden units (rows 250..255 are
// zero), each accumulated as its own 32-MFMA chain and added in the fixed order ((q0 + q1) + q2) + q3, so both shapes give the
// same bits:
//   MT = 32 (a single learner: 64 workgroups per network, latency): the 4 waves take one K quarter each;
//   MT = 64 (learner groups of >= 8: 32 workgroups per network, half the redundant panel loads): wave w takes the column tile
//            w & 1 and the K half w >> 1, i.e. two quarters with an accumulator each.
// Waves with a K part other than the first hand their accumulators to the first through LDS at the end.
constexpr int FWD_KQ = 64;                                    // k rows per K-quarter (32 MFMA pairs)
template <int MT> struct FwdShape {
    static constexpr int NMW = MT / 32;                       // column tiles (waves) per workgroup
    static constexpr int NKW = 4 / NMW;                       // K parts (waves) per column tile
    static constexpr int KPW = 256 / NKW;                     // k rows per wave
    static constexpr int QPW = KPW / FWD_KQ;                  // K quarters per wave
    static constexpr int LDS = (4 * KPW * 32 + 256 * 32 + W1K * BP + W1K * W1C + 96) * 4;
};
constexpr int FWD_LDS = FwdShape<64>::LDS;                    // the larger of the two

#ifdef ABL_STAMP
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) { stamps[2*(i)] = __builtin_amdgcn_s_memtime(); stamps[2*(i)+1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define STAMP(i)
#endif
template <int IN, bool PREP, int MT>
__device__ __forceinline__ void fwd_body(const FwdJob &J, float *smem, const PrepArgs *pa)
{
    typedef FwdShape<MT> SH;
#ifdef ABL_STAMP
    unsigned long long *stamps = reinterpret_cast<unsigned long long *>(const_cast<float *>(J.P3) + 100000);   // unused part of the slot
#endif
    STAMP(0);
    float *Hc = smem;                          // [4 waves][KPW][32]  relu(layer 1): this wave's K part x its 32 columns
    float *Wc = Hc + 4 * SH::KPW * 32;         // [256][32]  W2 panel (rows >= 250: zero)
    float *xs = Wc + 256 * 32;                 // [12][BP]
    float *w1 = xs + W1K * BP;                 // w1m [12][256]
    float *ep = w1 + W1K * W1C;                // [32][3]: b2, W3[.][0], W3[.][1] of this n-tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    constexpr int kMTiles = BP / MT;           // workgroups per n-tile
    if (PREP && (int)blockIdx.x >= NT * kMTiles) {
        // The five extra workgroups of an update's first launch: one publishes what the later launches read from the workspace
        // (the sampled, gathered and normalised minibatch), four pack the layer-1 images of the four networks.  Kept off the tile
        // workgroups so that none of those runs longer than the others.
        const int duty = (int)blockIdx.x - NT * kMTiles;
        if (duty == 0) {
            if (tid < BP) prep_column(*pa, tid, nullptr, true);
        } else {
            const int net = duty - 1;
            const shems_ddpg &d = pa->d;
            const float *Pn = net == SLOT_ACTOR_T ? d.actor_t : net == SLOT_CRITIC_T ? d.critic_t : net == SLOT_CRITIC ? d.critic : d.actor;
            pack_w1m(Pn, (net == SLOT_CRITIC_T || net == SLOT_CRITIC) ? CIN : SIN, w1t_of(d.ws, net));
        }
        return;
    }
    const int n0 = ((int)blockIdx.x / kMTiles) * 32;
    const int mt = wave % SH::NMW, kh = wave / SH::NMW, mbase = MT * ((int)blockIdx.x % kMTiles) + 32 * mt;
    const float *__restrict__ P = J.P;

    float epv = 0.0f;
    if (tid < 96) {                            // epilogue constants: in flight while the inputs are staged
        const int nl = tid / 3, c = tid - nl * 3, nc = min(n0 + nl, H2N - 1);
        epv = c == 0 ? P[off_b2(IN) + nc] : P[off_w3(IN) + nc * J.out + min(c - 1, J.out - 1)];
        if (n0 + nl >= H2N || c - 1 >= J.out) epv = 0.0f;
    }
    // W2[0..255][n0..n0+31] (rows of 128 B, 8 float4 each): 2048 float4, 8 per thread, all issued before the first store.
    // Columns >= 500 of the last tile read the next row / b2 (in bounds) and only feed output rows that are discarded.
    const float *__restrict__ W2 = P + off_w2(IN);
    float4 wv[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int e = it * 256 + tid, k = e >> 3, c = e & 7;
        const float4 t4 = *reinterpret_cast<const float4 *>(W2 + (int64_t)min(k, H1N - 1) * H2N + n0 + 4 * c);
        wv[it] = k < H1N ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    XRegs<IN> xr;
    W1mRegs wr;
    if (!PREP) {                                   // every global load of the stage goes out before the first one is consumed
        build_x_load<IN>(J.x, xr);
        stage_w1m_load(J.w1t, wr);
    }
    if (PREP) {
        // First launch of an update (actor_target on s'): no separate sample/gather/pack launch.  Every tile workgroup samples the
        // minibatch and gathers + normalises s' straight into its LDS input block, and packs the layer-1 image it needs from the
        // parameter block; five extra workgroups (above) publish the workspace copies for the later launches.
        if (tid < BP) prep_column(*pa, tid, xs, false);
        xs[9 * BP + tid] = 0.0f;                                                    // rows 9, 10 (2 * BP == blockDim)
        if (tid < BP) xs[11 * BP + tid] = 1.0f;                                     // bias row
        pack_w1m(P, IN, w1);
    } else {
        build_x_store<IN>(J.x, xr, xs, blockIdx.x == 0);
        stage_w1m_store(wr, w1);
    }
    if (tid < 96) ep[tid] = epv;
#pragma unroll
    for (int it = 0; it < 8; ++it) reinterpret_cast<float4 *>(Wc)[it * 256 + tid] = wv[it];
    __syncthreads();
    STAMP(1);

    // layer 1 on the matrix pipe: KPW / 32 tiles (the rows of this wave's K part) x its 32 columns; the accumulator chains are
    // independent, so their MFMAs interleave
    float *Hw = Hc + wave * (SH::KPW * 32);
    {
        constexpr int NT1 = SH::KPW / 32;
        f32x16 t[NT1];
#pragma unroll
        for (int q = 0; q < NT1; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) t[q][r] = 0.0f;
        float xb[W1K / 2], wa[W1K / 2][NT1];      // operand reads in one batch (see l1_tile)
#pragma unroll
        for (int sidx = 0; sidx < W1K / 2; ++sidx) {
            const int j = 2 * sidx + lh;
            xb[sidx] = xs[j * BP + mbase + li];
#pragma unroll
            for (int q = 0; q < NT1; ++q) wa[sidx][q] = w1[j * W1C + SH::KPW * kh + 32 * q + li];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sidx = 0; sidx < W1K / 2; ++sidx)
#pragma unroll
            for (int q = 0; q < NT1; ++q)
                t[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[sidx][q], xb[sidx], t[q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NT1; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) Hw[(32 * q + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = fmaxf(t[q][r], 0.0f);
    }
    STAMP(2);
    // this wave reads only its own Hw columns: no barrier needed between the layer-1 writes and the main loop
    f32x16 accq[SH::QPW];
#pragma unroll
    for (int qq = 0; qq < SH::QPW; ++qq) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accq[qq][r] = 0.0f;
        // 32 MFMA pairs over one K quarter, operand fetch software-pipelined one group (8 pairs) ahead
        const float *pa = Wc + (SH::KPW * kh + FWD_KQ * qq) * 32 + li, *pb = Hw + FWD_KQ * qq * 32 + li;
        float ac[8], bc[8], an[8], bn[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { ac[u] = pa[(2 * u + lh) * 32]; bc[u] = pb[(2 * u + lh) * 32]; }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int kk = 2 * ((g + 1) * 8 + u) + lh;
                    an[u] = pa[kk * 32]; bn[u] = pb[kk * 32];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) accq[qq] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], bc[u], accq[qq], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 8; ++u) { ac[u] = an[u]; bc[u] = bn[u]; }
        }
    }
    STAMP(3);
    // add the four K quarters in the fixed order ((q0 + q1) + q2) + q3: the waves of the later K parts hand their accumulators to the
    // first one of their column tile through LDS (Hc is free once everybody is here)
    __syncthreads();
    float *xch = Hc + mt * (3 * 16 * 64);                     // [quarter 1..3][16][64] of this column tile
    if (kh != 0) {
#pragma unroll
        for (int qq = 0; qq < SH::QPW; ++qq)
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[((SH::QPW * kh + qq - 1) * 16 + r) * 64 + lane] = accq[qq][r];
    }
    __syncthreads();
    if (kh != 0) return;
    f32x16 acc = accq[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += (q < SH::QPW) ? accq[q < SH::QPW ? q : 0][r] : xch[((q - 1) * 16 + r) * 64 + lane];
    }
    // epilogue: h2 = relu(acc + b2); store; layer-3 partial over this tile's 32 rows
    const int m = mbase + li;
    float p0 = 0.0f, p1 = 0.0f;
    // only this wave is still running: nothing hides an LDS latency, so all 48 epilogue constants are read in one batch (left to
    // the compiler each row's three reads sit right in front of their use: 16 exposed round trips)
    float eb[16], e0[16], e1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int nl = (r & 3) + 8 * (r >> 2) + 4 * lh;
        eb[r] = ep[nl * 3]; e0[r] = ep[nl * 3 + 1]; e1[r] = ep[nl * 3 + 2];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int nl = (r & 3) + 8 * (r >> 2) + 4 * lh, n = n0 + nl;
        const float h = n < H2N ? fmaxf(acc[r] + eb[r], 0.0f) : 0.0f;
        if (J.H2 && n < H2N) J.H2[n * BP + m] = h;
        p0 = fmaf(h, e0[r], p0);
        p1 = fmaf(h, e1[r], p1);
    }
    p0 += __shfl_xor(p0, 32, 64);
    p1 += __shfl_xor(p1, 32, 64);
    if (lh == 0) {
        J.P3[(((int)blockIdx.x / kMTiles) * 2 + 0) * BP + m] = p0;
        J.P3[(((int)blockIdx.x / kMTiles) * 2 + 1) * BP + m] = p1;
    }
    STAMP(12);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_fwd(FwdArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    FwdJob J = A.job[blockIdx.y];
    if (A.gstride) gshift(J, blockIdx.z * A.gstride);
    if (A.prep) {                                  // one job, an actor network
        PrepArgs pa = A.pa;
        if (A.gstride) { gshift(pa.d, blockIdx.z * A.gstride); gshift(pa.ring, blockIdx.z * A.gstride); pa.seed += blockIdx.z; }   // learner blockIdx.z
        if (A.mt == 64) fwd_body<SIN, true, 64>(J, smem, &pa); else fwd_body<SIN, true, 32>(J, smem, &pa);
        return;
    }
    if (A.mt == 64) {
        if (J.in == SIN) fwd_body<SIN, false, 64>(J, smem, nullptr); else fwd_body<CIN, false, 64>(J, smem, nullptr);
    } else {
        if (J.in == SIN) fwd_body<SIN, false, 32>(J, smem, nullptr); else fwd_body<CIN, false, 32>(J, smem, nullptr);
    }
}

// ---- critic loss head, evaluated in the prologue of every bwd(critic) workgroup (cheaper than a launch boundary) --------
// d3[0][m] = dq[m] = 2 (q - y) / B into LDS; workgroup 0 also publishes y, q, the loss and gb3.
// The global operands of the two heads, fetched with the rest of a backward workgroup's first batch of loads (all threads load;
// the critic head uses the values of threads < BP only).
struct HeadRegs { float v[KT * NQ]; float a, b, c, e; };
static_assert(KT * NQ >= 2 * NT, "HeadRegs.v holds the 2 x NT layer-3 partials of the critic head");
__device__ __forceinline__ void head_loss_load(const shems_ddpg &d, HeadRegs &R)
{
    const float *ws = d.ws;
    const int m = threadIdx.x & 127;
    const float *Pt = slot(d.ws, SLOT_CRITIC_T) + SL_P3, *Pc = slot(d.ws, SLOT_CRITIC) + SL_P3;
#pragma unroll
    for (int i = 0; i < NT; ++i) { R.v[i] = Pt[(i * 2) * BP + m]; R.v[NT + i] = Pc[(i * 2) * BP + m]; }
    R.a = d.critic_t[off_b3(CIN, 1)];
    R.b = d.critic[off_b3(CIN, 1)];
    R.c = ws[WS_R + m];
    R.e = ws[WS_DONE + m];
}
__device__ __forceinline__ void head_actor_load(const shems_ddpg &d, HeadRegs &R)
{
    const float *ws = d.ws;
    const int t = threadIdx.x, o = t >> 7, m = t & 127;
#pragma unroll
    for (int p = 0; p < KT * NQ; ++p) R.v[p] = ws[WS_DAP + (int64_t)(p * 2 + o) * BP + m];
    R.a = ws[WS_API + t];
}

__device__ __forceinline__ void head_loss(const shems_ddpg &d, const HeadRegs &R, float *d3 /*LDS [2][BP]*/, float *red /*LDS [8]*/, bool publisher)
{
    float *ws = d.ws;
    const int t = threadIdx.x, m = t & 127;
    float dq = 0.0f, diff = 0.0f;
    if (t < BP) {
        float q2 = R.a, q = R.b;
#pragma unroll
        for (int i = 0; i < NT; ++i) { q2 += R.v[i]; q += R.v[NT + i]; }
        const float y = R.c + d.gamma * (1.0f - R.e) * q2;                                  // DDPG.jl:133
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
            d.grad_critic[off_b3(CIN, 1)] = red[4] + red[5];
        }
    }
}

// ---- actor head backward, evaluated in the prologue of every bwd(actor) workgroup ------------------------------------
// d3[o][m] = (sum of the 32 partial d loss / d a_pi) * (1 - a_pi^2); workgroup 0 publishes d3, the actor loss and gb3.
__device__ __forceinline__ void head_actor(const shems_ddpg &d, const HeadRegs &R, float *d3 /*LDS [2][BP]*/, float *red /*LDS [8]*/, bool publisher)
{
    float *ws = d.ws;
    const int t = threadIdx.x, o = t >> 7, m = t & 127;
    const float a = R.a;
    float da = 0.0f;
#pragma unroll
    for (int p = 0; p < KT * NQ; ++p) da += R.v[p];
    const float g = da * (1.0f - a * a);                       // through tanh
    d3[t] = g;
    if (publisher) {
        ws[WS_D3A + t] = g;
        float q = 0.0f;
        if (o == 0 && m < d.batch) {
            const float *Pq = slot(ws, SLOT_CRITIC2) + SL_P3;
            q = d.critic[off_b3(CIN, 1)];
#pragma unroll
            for (int i = 0; i < NT; ++i) q += Pq[(i * 2) * BP + m];
        }
        const float sg = wave_sum(g), sq = wave_sum(q);
        if ((t & 63) == 0) { red[t >> 6] = sg; red[4 + (t >> 6)] = sq; }
        __syncthreads();
        if (t == 0) {
            d.grad_actor[off_b3(SIN, 2) + 0] = red[0] + red[1];
            d.grad_actor[off_b3(SIN, 2) + 1] = red[2] + red[3];
            d.losses[1] = -(red[4] + red[5]) / (float)d.batch;     // loss_act = -mean(critic(vcat(s, actor(s))))
        }
    }
}

// ---- kernel D: layer-2 backward ---------------------------------------------------------------------------
// D2[n][m] = (sum_o W3[n][o] d3[o][m]) * (h2[n][m] > 0) is generated while staging, never stored.
//   W workgroups (kt, nq): gW2[32 k][128 n] = sum_m h1[k][m] D2[n][m]; kt == 0 also emits gb2 and gW3.
//   I workgroups (kt, nq): D1part[nq][32 k][128 m] = sum_{n in quarter} W2[k][n] D2[n][m]; for the critic inside the
//                          actor loss they also emit the partial action gradient (through the layer-1 relu mask).
struct BwdArgs {
    const float *w1t;      // packed layer-1 image of that network
    const float *P;        // parameter block of the network being differentiated
    int in, out;
    XSrc x;                // its input (for the layer-1 recompute)
    const float *H2;       // [500][BP]
    const float *d3;       // [out][BP] error at the layer-3 pre-activation
    float *grad;           // gradient block (W part) or null
    float *D1P;            // [NQ][250][BP]
    float *DAP;            // [KT][NQ][2][BP] or null
    int n_w;               // number of W workgroups (32 or 0); when > 0, 8 more "G" workgroups emit gb2 and gW3
    int head;              // how d3 is obtained: 0 = read A.d3, 1 = critic loss head, 2 = actor head
    shems_ddpg dd;         // for the heads
    int64_t gstride;       // learner groups: byte stride between learners (0 = single learner)
};
__device__ __forceinline__ void gshift(BwdArgs &B, int64_t off)
{
    B.w1t = gsh(B.w1t, off); B.P = gsh(B.P, off); gshift(B.x, off); B.H2 = gsh(B.H2, off); B.d3 = gsh(B.d3, off);
    B.grad = gsh(B.grad, off); B.D1P = gsh(B.D1P, off); B.DAP = gsh(B.DAP, off); gshift(B.dd, off);
}
enum { BWD_NG = 16, BWD_GROWS = 512 / BWD_NG, BWD_GU = BWD_GROWS / 4 };   // G workgroups: 32 rows of gb2 / gW3 each, 8 per wave (32 x 16 measured equal)

constexpr int BWD_BT = BP * (NQW + 1);                       // W: [128 m][65] D2^T panel (>= I: [64 n][128 m] D2 panel)
constexpr int BWD_AT = BP * 33;                              // W: [128 m][33] h1^T panel (>= I: [32 k][65] W2 panel)
constexpr int BWD_LDS = (BWD_BT + BWD_AT + W1K * BP + W1K * W1C + AIN * BP + 2 * 512 + 8) * 4;

// D2 element: (sum_o W3[n][o] d3[o][m]) * (h2 > 0)
__device__ __forceinline__ float d2_val(float h2, int out, float w3a, float w3b, float d3a, float d3b)
{
    const float g = out == 2 ? fmaf(w3b, d3b, w3a * d3a) : w3a * d3a;
    return h2 > 0.0f ? g : 0.0f;
}

template <int IN>
__device__ __forceinline__ void bwd_body(const BwdArgs &A, float *smem)
{
#ifdef ABL_STAMP
    unsigned long long *bst = reinterpret_cast<unsigned long long *>(A.D1P + NQ * H1N * BP - 4096) + (((int)blockIdx.x == 0) ? 0 : 32);
    const bool bst_on = threadIdx.x == 0 && ((int)blockIdx.x == 0 || (int)blockIdx.x == A.n_w + (A.n_w > 0 ? (int)BWD_NG : 0));
#define BSTAMP(i) do { if (bst_on) { bst[2*(i)] = __builtin_amdgcn_s_memtime(); bst[2*(i)+1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define BSTAMP(i)
#endif
    BSTAMP(0);
    float *Bt = smem;                          // W: [128 m][65] D2^T panel;  I: [64 n][128 m] D2 panel
    float *At = Bt + BWD_BT;                   // W: [128 m][33] h1^T panel; I: [32 k][65] W2 panel
    float *xs = At + BWD_AT;                   // [12][BP]
    float *w1 = xs + W1K * BP;                 // w1m [12][256]
    float *d3 = w1 + W1K * W1C;                // [2][BP]
    float *w3s = d3 + AIN * BP;                // W3 [512][2] (out == 1: [.][0] only), zero beyond row 499
    float *red = w3s + 2 * 512;                // [8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const float *__restrict__ P = A.P;
    const float *__restrict__ W3 = P + off_w3(IN);
    const int n_g = A.n_w > 0 ? (int)BWD_NG : 0;
    const bool is_w = (int)blockIdx.x < A.n_w, is_g = !is_w && (int)blockIdx.x < A.n_w + n_g;
    const int b = is_w ? blockIdx.x : blockIdx.x - A.n_w - n_g;
    const int kt = b >> 3, nq = b & 7;                           // (k-tile of 32, n-block of 64)
    const int mcol = tid & 127, half = tid >> 7;

    // The H2 panel (and, for the input-gradient tiles, the W2 panel) does not depend on the error signal: its loads go out before
    // the head is evaluated, so the two global latencies overlap instead of following each other.
    float hv[32], wvp[8];
    const float *__restrict__ W2p = P + off_w2(IN);
    if (!is_g) {                                  // W and I tiles of a block read the same 64 rows of H2
#pragma unroll
        for (int u = 0; u < 32; ++u) hv[u] = A.H2[min(nq * NQW + 2 * u + half, H2N - 1) * BP + mcol];
        if (!is_w) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = u * 256 + tid, kl = e >> 6, nl = e & 63, k = kt * 32 + kl, n = nq * NQW + nl;
                const float t = W2p[(int64_t)min(k, H1N - 1) * H2N + min(n, H2N - 1)];
                wvp[u] = (k < H1N && n < H2N) ? t : 0.0f;
            }
        }
    }
    XRegs<IN> xr;
    W1mRegs wr;
    float w3v[4];
    build_x_load<IN>(A.x, xr);
    stage_w1m_load(A.w1t, wr);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int e = it * 256 + tid, n = e >> 1, o = e & 1;
        w3v[it] = W3[min(n, H2N - 1) * A.out + min(o, A.out - 1)];
    }
    HeadRegs hr;
    float d3v = 0.0f;
    if (A.head == 1) head_loss_load(A.dd, hr);
    else if (A.head == 2) head_actor_load(A.dd, hr);
    else d3v = A.d3[min(tid, A.out * BP - 1)];
    build_x_store<IN>(A.x, xr, xs, false);
    stage_w1m_store(wr, w1);
    if (A.head == 1) head_loss(A.dd, hr, d3, red, blockIdx.x == 0);
    else if (A.head == 2) head_actor(A.dd, hr, d3, red, blockIdx.x == 0);
    else d3[tid] = tid < A.out * BP ? d3v : 0.0f;                                           // AIN * BP == 256 == blockDim
    {   // W3 -> LDS as [n][2] (second column 0 for the critic), from the loads issued above
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = it * 256 + tid, n = e >> 1, o = e & 1;
            w3s[e] = (n < H2N && o < A.out) ? w3v[it] : 0.0f;
        }
    }
    __syncthreads();
    BSTAMP(1);
    const float d3a = d3[mcol], d3b = d3[BP + mcol];

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    if (is_w) {
        const int nbase = nq * NQW;
        // h1^T panel At[m][kl] (row stride 33): this wave's 32 columns of the k-tile, layer 1 on the matrix pipe.  Its MFMA chain is
        // issued first so that it runs under the VALU work of the D2 panel below.
        const f32x16 t = l1_tile(w1, xs, kt * 32, wave * 32, li, lh);
        // D2^T panel Bt[m][nl] (row stride 65), nl = 2*u + half: 32 rows per thread from the loads issued above (W3 comes from LDS;
        // rows >= 500 meet the zero rows of its image)
        {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int nl = 2 * u + half, n = nbase + nl;
                const float2 w = *reinterpret_cast<const float2 *>(w3s + 2 * n);
                Bt[mcol * (NQW + 1) + nl] = d2_val(hv[u], 2, w.x, w.y, d3a, d3b);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) At[(wave * 32 + li) * 33 + (r & 3) + 8 * (r >> 2) + 4 * lh] = fmaxf(t[r], 0.0f);
        __syncthreads();
        BSTAMP(2);
        // wave (nt, mh): the 32 x 32 tile of columns [32 nt, +32) over the batch half [64 mh, +64): 32 MFMA pairs, operand fetch
        // one group (8 pairs) ahead; the two halves are added through LDS
        const int nt = wave & 1, mh = wave >> 1;
        const float *pa = At + (64 * mh) * 33 + li, *pb = Bt + (64 * mh) * (NQW + 1) + nt * 32 + li;
        {
            float ac[8], bc[8], an[8], bn[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { ac[u] = pa[(2 * u + lh) * 33]; bc[u] = pb[(2 * u + lh) * (NQW + 1)]; }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < 3) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int mm = 2 * ((g + 1) * 8 + u) + lh;
                        an[u] = pa[mm * 33]; bn[u] = pb[mm * (NQW + 1)];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], bc[u], acc, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 8; ++u) { ac[u] = an[u]; bc[u] = bn[u]; }
            }
        }
        BSTAMP(3);
        __syncthreads();                            // everybody is done with the panels: Bt doubles as the exchange buffer
        float *xch = Bt + nt * (16 * 64);
        if (mh == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[r * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (mh == 0) {
            float *gW2 = A.grad + off_w2(IN);
            const int n = nbase + nt * 32 + li;
            float xv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = xch[r * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (k < H1N && n < H2N) gW2[k * H2N + n] = acc[r] + xv[r];
            }
        }
    } else if (is_g) {
        // gb2[n] = sum_m D2[n][m]; gW3[n][o] = sum_m h2[n][m] d3[o][m]: BWD_NG workgroups x BWD_GROWS rows, one wave per row, all of a
        // wave's rows in flight
        const int g = blockIdx.x - A.n_w;
        const float e0a = d3[lane], e0b = d3[64 + lane], e1a = d3[BP + lane], e1b = d3[BP + 64 + lane];
        {
            float h0[BWD_GU], h1[BWD_GU];
#pragma unroll
            for (int u = 0; u < BWD_GU; ++u) {
                const int nc = min(g * BWD_GROWS + wave + 4 * u, H2N - 1);
                h0[u] = A.H2[nc * BP + lane];
                h1[u] = A.H2[nc * BP + 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < BWD_GU; ++u) {
                const int nn = g * BWD_GROWS + wave + 4 * u;
                const float2 w = *reinterpret_cast<const float2 *>(w3s + 2 * min(nn, H2N - 1));
                const float s0 = wave_sum(h0[u] * e0a + h1[u] * e0b);
                const float s1 = wave_sum(h0[u] * e1a + h1[u] * e1b);
                const float sb = wave_sum(d2_val(h0[u], 2, w.x, w.y, e0a, e1a) + d2_val(h1[u], 2, w.x, w.y, e0b, e1b));
                if (lane == 0 && nn < H2N) {
                    A.grad[off_w3(IN) + nn * A.out] = s0;
                    if (A.out == 2) A.grad[off_w3(IN) + nn * 2 + 1] = s1;
                    A.grad[off_b2(IN) + nn] = sb;
                }
            }
        }
    } else {
        const int nb = nq * NQW;
        // D2 panel Bt[nl][m], nl = 2*u + half < 64 (rows >= 500: zero through the W3 image)
        {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int nl = 2 * u + half;
                const float2 w = *reinterpret_cast<const float2 *>(w3s + 2 * (nb + nl));
                Bt[nl * BP + mcol] = d2_val(hv[u], 2, w.x, w.y, d3a, d3b);
            }
        }
        // W2 panel At[kl][nl] (row stride 65): 32 x 64 elements from the loads issued above
        {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = u * 256 + tid, kl = e >> 6, nl = e & 63;
                At[kl * (NQW + 1) + nl] = wvp[u];
            }
        }
        __syncthreads();
        BSTAMP(2);
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
        BSTAMP(3);
        const int m = wave * 32 + li;
        float *D1 = A.D1P + (int64_t)nq * H1N * BP;
        float da0 = 0.0f, da1 = 0.0f;
        f32x16 pre;                             // layer-1 pre-activations of this (k-tile, m-tile): the relu mask, in acc's layout
        float wa0[16], wa1[16];                 // W1[9 + o][k]: the action rows of the critic's first layer, read in one batch
        if (A.DAP && IN == CIN) {
            pre = l1_tile(w1, xs, kt * 32, wave * 32, li, lh);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = min(kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, W1C - 1);
                wa0[r] = w1[9 * W1C + k]; wa1[r] = w1[10 * W1C + k];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (k < H1N) {
                D1[k * BP + m] = acc[r];
                if (A.DAP && IN == CIN) {
                    const float v = pre[r] > 0.0f ? acc[r] : 0.0f;
                    da0 = fmaf(wa0[r], v, da0);
                    da1 = fmaf(wa1[r], v, da1);
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
    BSTAMP(4);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_bwd(BwdArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (A.gstride) {
        BwdArgs B = A;
        gshift(B, blockIdx.z * A.gstride);
        if (B.in == SIN) bwd_body<SIN>(B, smem); else bwd_body<CIN>(B, smem);
        return;
    }
    if (A.in == SIN) bwd_body<SIN>(A, smem); else bwd_body<CIN>(A, smem);
}

// ---- kernel E: layer-1 gradients, one wave per hidden unit k --------------------------------------------------
//   D1[k][m] = (h1[k][m] > 0) * sum_q D1part[q][k][m];  gb1[k] = sum_m D1;  gW1[j][k] = sum_m x[j][m] D1[k][m]
// One wave, xs = the network input [12][BP] in LDS.  gout[j] = gW1[j][k] (j < IN), gout[IN] = gb1[k], valid in every lane.
// Everything a row needs comes straight from global memory in ONE batch of loads per lane -- its W1 column and b1, the NQ partial
// slabs and the two input columns (lane, lane + 64) of the network input, which for both differentiated networks is plain
// workspace data (normalised states, stored actions) -- so there is no LDS block, no barrier and one exposed latency.
template <int IN> struct L1Row { float w[IN]; float b; float part[2][NQ]; float x[2][IN]; };
template <int IN>
__device__ __forceinline__ void l1row_load(const float *__restrict__ P, const XSrc &xsrc, const float *__restrict__ D1P, int k, int lane,
                                           L1Row<IN> &R)
{
#pragma unroll
    for (int j = 0; j < IN; ++j) R.w[j] = P[j * H1N + k];
    R.b = P[off_b1(IN) + k];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m = lane + 64 * h;
#pragma unroll
        for (int q = 0; q < NQ; ++q) R.part[h][q] = D1P[((int64_t)q * H1N + k) * BP + m];
#pragma unroll
        for (int j = 0; j < IN; ++j) R.x[h][j] = j < SIN ? xsrc.X[j * BP + m] : xsrc.A[(j - SIN) * BP + m];
    }
}
template <int IN>
__device__ __forceinline__ void l1bwd_wave(const L1Row<IN> &R, float (&gout)[IN + 1])
{
    float dv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float pre = R.b;
#pragma unroll
        for (int j = 0; j < IN; ++j) pre = fmaf(R.w[j], R.x[h][j], pre);
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) s += R.part[h][q];
        dv[h] = pre > 0.0f ? s : 0.0f;
    }
    gout[IN] = __shfl(wave_sum(dv[0] + dv[1]), 0, 64);
#pragma unroll
    for (int j = 0; j < IN; ++j) gout[j] = __shfl(wave_sum(R.x[0][j] * dv[0] + R.x[1][j] * dv[1]), 0, 64);
}

template <int IN>
__device__ __forceinline__ void l1bwd_body(const float *__restrict__ P, const XSrc &x, const float *__restrict__ D1P,
                                           float *__restrict__ grad)
{
    const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= H1N) return;
    L1Row<IN> R;
    l1row_load<IN>(P, x, D1P, k, lane, R);
    float gout[IN + 1];
    l1bwd_wave<IN>(R, gout);
    if (lane == 0) {
        grad[off_b1(IN) + k] = gout[IN];
#pragma unroll
        for (int j = 0; j < IN; ++j) grad[j * H1N + k] = gout[j];
    }
}

__global__ __launch_bounds__(256) void k_l1bwd(const float *P, int in, XSrc x, const float *D1P, float *grad, int64_t gstride)
{
    if (gstride) { const int64_t off = blockIdx.z * gstride; P = gsh(P, off); gshift(x, off); D1P = gsh(D1P, off); grad = gsh(grad, off); }
    if (in == SIN) l1bwd_body<SIN>(P, x, D1P, grad); else l1bwd_body<CIN>(P, x, D1P, grad);
}

// ---- kernel G: Flux 0.12.1 ADAM + soft target update -----------------------------------------------------------
//   mt = b1*mt + (1-b1)*g ; vt = b2*vt + (1-b2)*g^2 ; delta = mt/(1-bp1) / (sqrt(vt/(1-bp2)) + eps) * eta ; p -= delta
//   (Float64 scalars broadcast over Float32 arrays: each element is computed in f64 and stored as f32)
//   then target = (1f0 - tau) * target + tau * p   (DDPG.jl:99-103)
// Single-replica form (shems_ddpg.fuse_l1): no gradient exchange sits between backward and ADAM, so the layer-1 gradient rows are
// produced HERE instead of by a k_l1bwd launch of their own: workgroup b < 63 computes the rows of hidden units 4b..4b+3 (one wave
// each, same arithmetic as k_l1bwd), stores them to the gradient block and applies ADAM to exactly those (in + 1) * 4 elements; the
// other workgroups sweep the elements from b1's end onwards.  No element is read by one workgroup and written by another.
struct AdamCtx {
    float *p; const float *g; float *mt, *vt, *target; float *w1t_g; float *publish;
    int n, in; double eta, bp1, bp2, gscale; float tau;
};
struct L1Src { const float *P; XSrc x; const float *D1P; float *grad; int on; };

// One element of ADAM + soft update on values: (m, v, p, target) in, updated in place.
__device__ __forceinline__ void adam_math(const AdamCtx &c, float graw, float &m, float &v, float &p, float &t)
{
    // Julia evaluates the broadcast expressions without fusing multiplies into adds; keeping the compiler from contracting also
    // makes the inlined copies of this function (layer-1 rows / sweep) round identically.
#pragma clang fp contract(off)
    const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
    const float gf = (float)((double)graw * c.gscale);          // averaged gradient, as every replica holds it
    const float m1 = (float)(b1 * (double)m + (1.0 - b1) * (double)gf);
    const float g2 = gf * gf;                                    // Flux 0.12.1 `Δ^2` on a Float32 array: literal_pow = Δ*Δ in Float32, then promoted
    const float v1 = (float)(b2 * (double)v + (1.0 - b2) * (double)g2);
    const float delta = (float)((double)m1 / (1.0 - c.bp1) / (sqrt((double)v1 / (1.0 - c.bp2)) + eps) * c.eta);
    const float pn = p - delta;
    const float one_m_tau = 1.0f - c.tau;
    t = one_m_tau * t + c.tau * pn;
    m = m1; v = v1; p = pn;
}
__device__ __forceinline__ void adam_image(const AdamCtx &c, int i, float pn)
{
    if (c.w1t_g && i < c.in * H1N + H1N) {   // keep the packed layer-1 image of the updated network current
        const int j = i / H1N, k = i - j * H1N;
        c.w1t_g[(j < c.in ? j : W1K - 1) * W1C + k] = pn;
    }
}
__device__ __forceinline__ void adam_elem(const AdamCtx &c, int i, float graw)
{
    float m = c.mt[i], v = c.vt[i], p = c.p[i], t = c.target[i];
    adam_math(c, graw, m, v, p, t);
    c.mt[i] = m; c.vt[i] = v; c.p[i] = p; c.target[i] = t;
    if (c.publish) c.publish[i] = p;
    adam_image(c, i, p);
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
        if (c.w1t_g && i0 < c.in * H1N + H1N) { adam_image(c, i0, p4.x); adam_image(c, i0 + 1, p4.y); adam_image(c, i0 + 2, p4.z); adam_image(c, i0 + 3, p4.w); }
    } else {
        for (int i = i0; i < c.n; ++i) adam_elem(c, i, c.g[i]);
    }
}

template <int IN>
__device__ __forceinline__ void adam_l1_rows(const AdamCtx &c, const L1Src &l1)
{
    const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= H1N) return;
    L1Row<IN> R;
    l1row_load<IN>(l1.P, l1.x, l1.D1P, k, lane, R);
    float gout[IN + 1];
    l1bwd_wave<IN>(R, gout);
    float mine = 0.0f;
#pragma unroll
    for (int j = 0; j <= IN; ++j) mine = lane == j ? gout[j] : mine;
    if (lane <= IN) {
        const int e = lane < IN ? lane * H1N + k : off_b1(IN) + k;
        l1.grad[e] = mine;                   // the gradient block stays complete for callers that read it after the update
        adam_elem(c, e, mine);
    }
}

__global__ __launch_bounds__(256) void k_adam_soft(AdamCtx c, L1Src l1, int64_t gstride)
{
    if (gstride) {
        const int64_t off = blockIdx.z * gstride;
        c.p = gsh(c.p, off); c.g = gsh(c.g, off); c.mt = gsh(c.mt, off); c.vt = gsh(c.vt, off); c.target = gsh(c.target, off);
        c.w1t_g = gsh(c.w1t_g, off);
        l1.P = gsh(l1.P, off); gshift(l1.x, off); l1.D1P = gsh(l1.D1P, off); l1.grad = gsh(l1.grad, off);
    }
    int first = 0, blk = blockIdx.x;
    if (l1.on) {                                  // the first 63 workgroups ONLY produce + apply the layer-1 rows; the sweep follows them
        constexpr int kRowWgs = (H1N + 3) / 4;
        if (blk < kRowWgs) {
            if (c.in == SIN) adam_l1_rows<SIN>(c, l1); else adam_l1_rows<CIN>(c, l1);
            return;
        }
        first = (c.in + 1) * H1N;
        blk -= kRowWgs;
    }
    const int i0 = first + 4 * (blk * (int)blockDim.x + (int)threadIdx.x);          // first is a multiple of 4, the blocks 16-byte aligned
    if (i0 < c.n) adam_vec4(c, i0);
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

// s of the minibatch the last shems_ddpg_critic_grad sampled (ring slots kept in the workspace) -> obs [batch][9]
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

static int check_group(const shems_group *g, const char *fn)
{
    if (!g || g->count < 1 || g->count > 65535 || g->stride_bytes < 0 || (g->stride_bytes & 15) != 0 || (g->count > 1 && g->stride_bytes == 0))
        return set_error(SHEMS_ERR_ARG, "%s: shems_group needs 1 <= count <= 65535 and a 16-byte-multiple stride", fn);
    return SHEMS_OK;
}

static int critic_grad_impl(const shems_ddpg *d, const shems_replay *ring, int64_t ring_len, uint64_t seed, uint32_t tick,
                            int64_t excl_pos, int64_t excl_count, unsigned L, int64_t gs, void *stream)
{
    if (int rc = check_ddpg(d, "shems_ddpg_critic_grad")) return rc;
    if (!ring || !ring->s || !ring->a || !ring->r || !ring->s2 || !ring->done || ring_len < 1 || ring_len > ring->capacity)
        return set_error(SHEMS_ERR_ARG, "shems_ddpg_critic_grad: bad replay ring / length");
    if (excl_count < 0 || excl_pos < 0 || (excl_count > 0 && (ring_len != ring->capacity || excl_count >= ring_len)))
        return set_error(SHEMS_ERR_ARG, "shems_ddpg_critic_grad_ex: an exclusion window needs a full ring and 0 <= count < capacity");
    hipStream_t st = (hipStream_t)stream;
    float *ws = d->ws;
    const XSrc x_s2{ws + WS_X2T, nullptr, nullptr, nullptr, nullptr};
    const XSrc x_s{ws + WS_XT, nullptr, nullptr, nullptr, nullptr};
    const XSrc x_s2a{ws + WS_X2T, nullptr, slot(ws, SLOT_ACTOR_T) + SL_P3, d->actor_t + off_b3(SIN, 2), nullptr};
    const XSrc x_sa{ws + WS_XT, ws + WS_AT, nullptr, nullptr, nullptr};
    FwdArgs f;
    std::memset(&f, 0, sizeof f);
    f.gstride = gs;
    f.mt = L >= 8 ? 64 : 32;                       // column-tile width (same bits either way, see FwdShape)
    const unsigned fgx = NT * (BP / f.mt);
    const int flds = f.mt == 64 ? FwdShape<64>::LDS : FwdShape<32>::LDS;
    f.job[0] = FwdJob{w1t_of(ws, SLOT_ACTOR_T), d->actor_t, SIN, 2, x_s2, nullptr, slot(ws, SLOT_ACTOR_T) + SL_P3};
    f.prep = 1;                                    // sample + gather + normalise + layer-1 image packing ride in this launch
    f.pa = PrepArgs{*d, *ring, ring_len, seed, tick, excl_pos, excl_count};
    hipLaunchKernelGGL(k_fwd, dim3(fgx + 5, 1, L), dim3(256), flds, st, f);      // + 5 publishing workgroups (see fwd_body)
    f.prep = 0;
    f.job[0] = FwdJob{w1t_of(ws, SLOT_CRITIC_T), d->critic_t, CIN, 1, x_s2a, nullptr, slot(ws, SLOT_CRITIC_T) + SL_P3};
    f.job[1] = FwdJob{w1t_of(ws, SLOT_CRITIC), d->critic, CIN, 1, x_sa, slot(ws, SLOT_CRITIC) + SL_H2, slot(ws, SLOT_CRITIC) + SL_P3};
    f.job[2] = FwdJob{w1t_of(ws, SLOT_ACTOR), d->actor, SIN, 2, x_s, slot(ws, SLOT_ACTOR) + SL_H2, slot(ws, SLOT_ACTOR) + SL_P3};
    hipLaunchKernelGGL(k_fwd, dim3(fgx, 3, L), dim3(256), flds, st, f);
    float *S = slot(ws, SLOT_CRITIC);
    const BwdArgs b{w1t_of(ws, SLOT_CRITIC), d->critic, CIN, 1, x_sa, S + SL_H2, ws + WS_D3C, d->grad_critic, S + SL_D1P, nullptr, KT * NQ, 1, *d, gs};
    hipLaunchKernelGGL(k_bwd, dim3(2 * KT * NQ + BWD_NG, 1, L), dim3(256), BWD_LDS, st, b);
    if (!d->fuse_l1)
        hipLaunchKernelGGL(k_l1bwd, dim3((H1N + 3) / 4, 1, L), dim3(256), 0, st, (const float *)d->critic, (int)CIN, x_sa,
                           (const float *)(S + SL_D1P), d->grad_critic, gs);
    return hip_ok(hipGetLastError(), "ddpg critic_grad launches");
}

int shems_ddpg_critic_grad(const shems_ddpg *d, const shems_replay *ring, int64_t ring_len, uint64_t seed, uint32_t tick,
                           void *stream)
{
    return critic_grad_impl(d, ring, ring_len, seed, tick, 0, 0, 1, 0, stream);
}

int shems_ddpg_critic_grad_ex(const shems_ddpg *d, const shems_replay *ring, int64_t ring_len, uint64_t seed, uint32_t tick,
                              int64_t excl_pos, int64_t excl_count, void *stream)
{
    return critic_grad_impl(d, ring, ring_len, seed, tick, excl_pos, excl_count, 1, 0, stream);
}

int shems_ddpg_group_critic_grad(const shems_ddpg *d0, const shems_replay *ring0, const shems_group *g, int64_t ring_len,
                                 uint64_t seed, uint32_t tick, void *stream)
{
    if (int rc = check_group(g, "shems_ddpg_group_critic_grad")) return rc;
    return critic_grad_impl(d0, ring0, ring_len, seed, tick, 0, 0, (unsigned)g->count, g->count > 1 ? g->stride_bytes : 0, stream);
}

static int adam_launch(float *p, const float *g, float *m, float *v, float *target, int n, double eta, double bp1, double bp2,
                       double gscale, float tau, float *w1t_g, int in, float *publish, hipStream_t st, unsigned L, int64_t gs, const L1Src &l1)
{
    if (!(bp1 > 0.0 && bp1 < 1.0 && bp2 > 0.0 && bp2 < 1.0)) return set_error(SHEMS_ERR_ARG, "adam: beta powers must be in (0,1)");
    const AdamCtx c{p, g, m, v, target, w1t_g, publish, n, in, eta, bp1, bp2, gscale, tau};
    const int sweep = l1.on ? n - (in + 1) * H1N : n;             // elements the plain sweep covers
    const int row_wgs = l1.on ? (H1N + 3) / 4 : 0;                // + the workgroups that produce and apply the layer-1 rows
    static_assert(((SIN + 1) * H1N) % 4 == 0 && ((CIN + 1) * H1N) % 4 == 0, "the sweep starts on a 16-byte boundary");
    hipLaunchKernelGGL(k_adam_soft, dim3(row_wgs + (sweep + 1023) / 1024, 1, L), dim3(256), 0, st, c, l1, gs);
    return hip_ok(hipGetLastError(), "k_adam_soft launch");
}

// the layer-1 sources of the two differentiated networks (must match what *_grad_impl hands to k_bwd / k_l1bwd)
static L1Src l1_critic(const shems_ddpg *d, bool on)
{
    float *ws = d->ws;
    return L1Src{d->critic, XSrc{ws + WS_XT, ws + WS_AT, nullptr, nullptr, nullptr}, slot(ws, SLOT_CRITIC) + SL_D1P, d->grad_critic, on ? 1 : 0};
}
static L1Src l1_actor(const shems_ddpg *d, bool on)
{
    float *ws = d->ws;
    return L1Src{d->actor, XSrc{ws + WS_XT, nullptr, nullptr, nullptr, nullptr}, slot(ws, SLOT_ACTOR) + SL_D1P, d->grad_actor, on ? 1 : 0};
}

int shems_ddpg_critic_apply(const shems_ddpg *d, double eta, double bp1, double bp2, double grad_scale, void *stream)
{
    if (int rc = check_ddpg(d, "shems_ddpg_critic_apply")) return rc;
    if (d->fuse_l1 && grad_scale != 1.0)
        return set_error(SHEMS_ERR_ARG, "shems_ddpg_critic_apply: fuse_l1 is the single-replica form (grad_scale must be 1)");
    return adam_launch(d->critic, d->grad_critic, d->m_critic, d->v_critic, d->critic_t, SHEMS_CRITIC_PARAMS, eta, bp1, bp2,
                       grad_scale, d->tau, w1t_of(d->ws, SLOT_CRITIC), (int)CIN, nullptr, (hipStream_t)stream, 1, 0, l1_critic(d, d->fuse_l1 != 0));
}

int shems_ddpg_group_critic_apply(const shems_ddpg *d, const shems_group *g, double eta, double bp1, double bp2, void *stream)
{
    if (int rc = check_ddpg(d, "shems_ddpg_group_critic_apply")) return rc;
    if (int rc = check_group(g, "shems_ddpg_group_critic_apply")) return rc;
    return adam_launch(d->critic, d->grad_critic, d->m_critic, d->v_critic, d->critic_t, SHEMS_CRITIC_PARAMS, eta, bp1, bp2, 1.0,
                       d->tau, w1t_of(d->ws, SLOT_CRITIC), (int)CIN, nullptr, (hipStream_t)stream, (unsigned)g->count,
                       g->count > 1 ? g->stride_bytes : 0, l1_critic(d, d->fuse_l1 != 0));
}

static int actor_grad_impl(const shems_ddpg *d, unsigned L, int64_t gs, void *stream)
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
    f.gstride = gs;
    f.mt = L >= 8 ? 64 : 32;                       // column-tile width (same bits either way, see FwdShape)
    const unsigned fgx = NT * (BP / f.mt);
    const int flds = f.mt == 64 ? FwdShape<64>::LDS : FwdShape<32>::LDS;
    f.job[0] = FwdJob{w1t_of(ws, SLOT_CRITIC), d->critic, CIN, 1, x_spi, C2 + SL_H2, C2 + SL_P3};
    hipLaunchKernelGGL(k_fwd, dim3(fgx, 1, L), dim3(256), flds, st, f);
    const BwdArgs bi{w1t_of(ws, SLOT_CRITIC), d->critic, CIN, 1, x_spi_ro, C2 + SL_H2, ws + WS_D3Q, nullptr, C2 + SL_D1P, ws + WS_DAP, 0, 0, *d, gs};
    hipLaunchKernelGGL(k_bwd, dim3(KT * NQ, 1, L), dim3(256), BWD_LDS, st, bi);
    float *S = slot(ws, SLOT_ACTOR);
    const BwdArgs b{w1t_of(ws, SLOT_ACTOR), d->actor, SIN, 2, x_s, S + SL_H2, ws + WS_D3A, d->grad_actor, S + SL_D1P, nullptr, KT * NQ, 2, *d, gs};
    hipLaunchKernelGGL(k_bwd, dim3(2 * KT * NQ + BWD_NG, 1, L), dim3(256), BWD_LDS, st, b);
    if (!d->fuse_l1)
        hipLaunchKernelGGL(k_l1bwd, dim3((H1N + 3) / 4, 1, L), dim3(256), 0, st, (const float *)d->actor, (int)SIN, x_s,
                           (const float *)(S + SL_D1P), d->grad_actor, gs);
    return hip_ok(hipGetLastError(), "ddpg actor_grad launches");
}

int shems_ddpg_actor_grad(const shems_ddpg *d, void *stream) { return actor_grad_impl(d, 1, 0, stream); }

int shems_ddpg_group_actor_grad(const shems_ddpg *d0, const shems_group *g, void *stream)
{
    if (int rc = check_group(g, "shems_ddpg_group_actor_grad")) return rc;
    return actor_grad_impl(d0, (unsigned)g->count, g->count > 1 ? g->stride_bytes : 0, stream);
}

int shems_ddpg_group_actor_apply(const shems_ddpg *d, const shems_group *g, double eta, double bp1, double bp2, void *stream)
{
    if (int rc = check_ddpg(d, "shems_ddpg_group_actor_apply")) return rc;
    if (int rc = check_group(g, "shems_ddpg_group_actor_apply")) return rc;
    return adam_launch(d->actor, d->grad_actor, d->m_actor, d->v_actor, d->actor_t, SHEMS_ACTOR_PARAMS, eta, bp1, bp2, 1.0,
                       d->tau, nullptr, (int)SIN, nullptr, (hipStream_t)stream, (unsigned)g->count, g->count > 1 ? g->stride_bytes : 0,
                       l1_actor(d, d->fuse_l1 != 0));
}

int shems_ddpg_actor_apply_pub(const shems_ddpg *d, double eta, double bp1, double bp2, double grad_scale, float *d_publish,
                               void *stream)
{
    if (int rc = check_ddpg(d, "shems_ddpg_actor_apply")) return rc;
    if (d->fuse_l1 && grad_scale != 1.0)
        return set_error(SHEMS_ERR_ARG, "shems_ddpg_actor_apply: fuse_l1 is the single-replica form (grad_scale must be 1)");
    return adam_launch(d->actor, d->grad_actor, d->m_actor, d->v_actor, d->actor_t, SHEMS_ACTOR_PARAMS, eta, bp1, bp2, grad_scale,
                       d->tau, nullptr, (int)SIN, d_publish, (hipStream_t)stream, 1, 0, l1_actor(d, d->fuse_l1 != 0));
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
