// shems_policy.hip -- actor forward (9 -> 250 -> 500 -> 2) on fp32 MFMA, fused with observation
// normalisation, Gaussian exploration noise, clamp, scale_action, step! and the replay insert.
//
// Replaces the per-step body of the reference's episode! (DDPG.jl:195-234):
//     s = copy(env.state); a, noise = act(normalize(s |> gpu)); scaled = scale_action(a)
//     r, s' = step!(env, s, scaled); remember(s, a, r, s', finished(env, s'))
// which there costs one H2D copy, ~8 tiny kernels at batch 1, one D2H sync and a CSV parse per step.
//
// Three kernels share the arithmetic and one CANONICAL COLUMN ORDER (act_col below), so they write the same bytes, and the batch size
// picks one (dispatch_act):
//   k_act2  > 8 192 envs: 64-env tiles, < 80 KB of LDS and <= 256 registers, so TWO workgroups are resident per CU and one's
//           latency-bound phases (stage 0, layer 1, layer 3, env tail) run under the other's MFMAs -- the default for large batches;
//   k_actg  <= 8 192 envs: 32-env tiles, one wave per 64-column group; 8 waves per tile, or two workgroups per tile (<= 4 096 envs)
//           joined by one 8-byte exchange per env;
//   k_act   the round-1/2 kernel described next: learner groups, and the forms the all-forms test and A/B runs select by knob.
// (In the description below "tile (a, b) holds the columns n = 128w + 4i + a" is round 2's layout; round 3 interleaves within
// 64-column groups, see act_col.)
//
// k_act -- decomposition (gfx950, wave64, 4 waves per workgroup, one workgroup per CU):
//   * a workgroup owns BM = 32*TM envs.  Everything is kept FEATURE-major ("[k][m]", env index
//     contiguous) so that layer outputs come out of the MFMA in exactly the layout the next layer
//     consumes: D'[n][m] = sum_k W[k][n] * H[k][m] with the weights as the MFMA A operand
//     (A[i = n][k] = W[k][n0+i]: Flux's column-major out x in matrix IS [k][n]) and the activations as
//     the B operand (B[k][j = m]).  v_mfma_f32_32x32x2_f32: lane l holds A[l&31][l>>5], B[l>>5][l&31].
//   * layer 2 (97.5 % of the FLOPs): wave w accumulates the 128(n) x BM(m) slab n in [128w, 128w+128)
//     = 4 x TM tiles of 32x32 (16*4*TM accumulator registers) over K = 250 = 125 MFMA k-steps.  Tile (a, b) holds the
//     columns n = 128w + 4i + a and the envs m = TM*j + b (i, j = a lane's MFMA row / column index), so what a lane needs
//     per k-step is contiguous in LDS: one ds_read_b128 of weights, one of activations, fetched one k-step ahead.
//     Both operands are streamed through LDS, double-buffered: W2 in chunks of 16 k-rows by LDS-DMA
//     (global_load_lds_dwordx4: wave w moves the 8 consecutive 1-KiB pieces 8w..8w+7 during the first half of the previous
//     chunk, one 64-bit address and one M0 value per chunk, the piece chosen by the instruction's immediate offset), and the
//     layer-1 activations relu(W1 x + b1) in 32-row groups that are RECOMPUTED ON THE MATRIX PIPE as well (K = 9 inputs +
//     a bias row = 10 = 5 MFMA k-steps per 32x32 tile, +1.9 % MFMA work; an inline-asm chain so that its
//     accumulator stays in VGPRs) -- cheaper than holding the 250 x BM activation tile (125 KB at BM = 128) in LDS, and it
//     keeps the VALU out of the main loop (the first version computed them with FMAs: profiles/r01_train_v1_*).  N is padded
//     500 -> 512 (zero bias / W3 rows), the last chunk runs only its 5 real k-steps.
//   * epilogue: bias + relu on the accumulators, layer 3 (500 -> 2) as per-lane partial dot products
//     reduced across lane halves (DPP) and the 4 waves (LDS), + b3, tanh, noise, clamp.
//   * one thread per env then runs scale_action + step! (shems_core.h, exact reference arithmetic)
//     and pushes the transition into the HBM replay ring.
// LDS: 2 x 32 KB (W2 chunks) + 2 x 32*BM*4 (layer-1 groups) + x (10*BM*4) + layer-1 image 10 KB + b2/W3/b3 6 KB.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "shems_env_dev.h"
#include "shems_internal.h"

namespace shems {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N> struct FVec { typedef float type __attribute__((ext_vector_type(N))); };
template <> struct FVec<1> { typedef float type; };
template <int N> __device__ __forceinline__ float fvec_get(const typename FVec<N>::type &v, int i) { return v[i]; }
template <> __device__ __forceinline__ float fvec_get<1>(const float &v, int) { return v; }

constexpr int kIn = 9, kH1 = SHEMS_L1, kH2 = SHEMS_L2, kOut = 2;

// CANONICAL COLUMN ORDER of layer 2 / layer 3 (every form of the kernel follows it, so all forms write the same bytes):
// the 512 (padded) hidden-2 columns are 8 GROUPS of 64; a group is two MFMA tiles, tile a in {0, 1} of group g holding the columns
// n = 64 g + 2 i + a (i = MFMA row).  Layer 3 is summed per tile (one FMA chain over a lane's 16 rows), a lane adds the chains of the
// group's two tiles, the two lane halves are added; the group sums of each column HALF are added in order, H0 = ((g0 + g1) + g2) + g3,
// H1 = ((g4 + g5) + g6) + g7, and the output is b3 + (H0 + H1).  A wave that owns NA tiles
// owns NA / 2 consecutive groups; the weights a lane needs for a k-step are 8 contiguous bytes per group (one ds_read2_b64 for two
// groups).  Because the order is defined on groups, a group can be computed by any wave of any workgroup: the 8-wave and the
// column-split (two workgroups per env tile) forms below produce the bits of the 4-wave forms.
template <int NA> __device__ __forceinline__ int act_col(int nbase, int t, int row) { return nbase + 64 * (t >> 1) + 2 * row + (t & 1); }
template <int NA> __device__ __forceinline__ typename FVec<NA>::type act_load_a(const float *row_at_nbase, int li);
template <> __device__ __forceinline__ FVec<4>::type act_load_a<4>(const float *p, int li)
{
    const f32x2 x = *reinterpret_cast<const f32x2 *>(p + 2 * li), y = *reinterpret_cast<const f32x2 *>(p + 64 + 2 * li);
    return FVec<4>::type{x[0], x[1], y[0], y[1]};
}
template <> __device__ __forceinline__ FVec<2>::type act_load_a<2>(const float *p, int li)
{
    return *reinterpret_cast<const f32x2 *>(p + 2 * li);
}
constexpr int kKC = 16;                         // k-rows per staged W2 chunk (8 MFMA k-steps)
constexpr int kChunks = 16;                     // K = 250 padded to 256: the padded rows of layer 1 are exactly 0
constexpr int kWcFloats = 8192 + 16;             // 32 whole 1-KiB LDS-DMA pieces (16 rows = 8000 floats, + 192 of the next row) + read pad
constexpr int kW1K = 10, kW1C = 256;            // layer-1 operand image: rows 0..8 W1[j][k], row 9 b1[k] (K = 9 inputs + bias = 5 MFMA k-steps)
constexpr int kOffB1 = kIn * kH1, kOffW2 = kOffB1 + kH1, kOffB2 = kOffW2 + kH1 * kH2, kOffW3 = kOffB2 + kH2,
              kOffB3 = kOffW3 + kH2 * kOut;
static_assert(kOffB3 + kOut == SHEMS_ACTOR_PARAMS, "actor layout");
static_assert(kW1K == 10 && kIn == 9, "L1_GROUP's MFMA chain is written out for 5 k-steps: 9 inputs + the bias row");
constexpr int kH2P = 512;                       // n padded to 16 MFMA tiles; pad rows carry zero bias / W3
constexpr int kTailFloats = kH2P + kH2P * kOut + kOut;   // LDS image: b2[512], W3[512][2], b3[2]

struct ActArgs {
    shems_view v;              // only used when do_step
    shems_act_params p;
    const float *obs;          // [m][9]
    int64_t m;
    int64_t m0;                // first env of this launch (0 = the whole batch): the launch covers envs [m0, m), tile t = envs m0 + t * BM ..
    float *a_out;              // [m][2] or null
    double *rewards;
    float *rewards_f32;
    double *block_reward;
    double *returns_acc;
    shems_replay ring;
    shems_ring_window win;
    int do_step;
    int use_ring;
    // learner groups (shems_group): env i belongs to learner i / genvs, whose actor / s_min / s_max / ring are the
    // learner-0 pointers + learner * gstride bytes.  gcount <= 1: one learner.
    int gcount;
    int64_t gstride;
    int64_t genvs;
    // learner groups on the tiled working layout (shems_group_w2t): learner 0's actor region; W2 is read from its p arrays instead of
    // from the Flux-order block (the free-running forms of k_act only).  null: Flux order.
    const float *w2t;
    int tm_max;                // learner groups: largest env tile (in units of 32 envs) that divides envs_per_learner -- a tile never straddles two learners; 0 = no limit
};

template <class T>
__device__ __forceinline__ T *gsh(T *p, int64_t off)
{
    // byte arithmetic on the pointer itself (no round trip through an integer): the compiler keeps the global address space of the
    // kernel argument and emits global_load / global_store instead of flat accesses
    typedef typename std::conditional<std::is_const<T>::value, const char, char>::type B;
    return p ? reinterpret_cast<T *>(reinterpret_cast<B *>(p) + off) : p;
}

// RD = 0: the workgroup streams W2 as whole 16-row chunks through a shared double buffer and layer 1 is produced 32 rows at a time
// inside the loop.  RD >= 2 ("free-running waves", small tiles): each wave streams only ITS 128 columns of W2 through a private ring
// of RD chunks (16 rows x 512 B) and all of relu(layer 1) is laid down before the loop, so the loop has no workgroup barrier.
constexpr int kFreeChunkFloats = kKC * 128;      // one wave's chunk: 16 rows x 128 columns
constexpr int kPreDw = 16;                       // TailPre block per env (free-running form)
// Rows of relu(layer 1) resident in LDS: the shared-stream form keeps two 32-row groups; the free-running form all 256 rows for
// TM <= 2 and, for TM = 4 (128 envs per workgroup: 256 rows would be 128 KB), one half at a time -- the second half is laid down
// between chunk 7 and chunk 8, the only two workgroup barriers of the layer.
constexpr int act_h_rows(int tm, int rd) { return rd == 0 ? 64 : tm == 4 ? 128 : 256; }
constexpr bool act_tail_pre(int tm, int rd) { return rd != 0 && tm <= 2; }    // TailPre blocks: the small tiles only (LDS)
template <int TM, int NW, int RD = 0>
constexpr size_t act_lds_bytes()
{
    return sizeof(float) * ((RD ? NW * RD * kFreeChunkFloats + 16 : 2 * kWcFloats) + act_h_rows(TM, RD) * 32 * TM + kW1K * 32 * TM + kW1K * kW1C +
                            (kTailFloats + 2) + 32 * TM * kIn + (act_tail_pre(TM, RD) ? 32 * TM * kPreDw : 0));
}

__device__ __forceinline__ void glds16(const void *g, void *lds)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}

// Piece q (0..7) of a wave's 8-KiB LDS-DMA stream.  The instruction's immediate offset is added to the global address AND to
// the LDS address (M0 base + offset + lane * 16), so both bases point at piece 4 and the piece is chosen by (q - 4) KiB alone:
// one 64-bit address and one M0 value per chunk.  The builtin wants an integer constant expression, hence the switch.
template <int Q>
__device__ __forceinline__ void glds16_imm(const char *base, char *lds)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)base,
                                     (__attribute__((address_space(3))) void *)lds, 16, (Q - 4) * 1024, 0);
}
__device__ __forceinline__ void glds16_piece(const char *base, char *lds, int q)
{
    switch (q) {
    case 0: glds16_imm<0>(base, lds); break;
    case 1: glds16_imm<1>(base, lds); break;
    case 2: glds16_imm<2>(base, lds); break;
    case 3: glds16_imm<3>(base, lds); break;
    case 4: glds16_imm<4>(base, lds); break;
    case 5: glds16_imm<5>(base, lds); break;
    case 6: glds16_imm<6>(base, lds); break;
    default: glds16_imm<7>(base, lds); break;
    }
}

// LDS-DMA as an asm statement (k_actg).  Given the builtin, the compiler books a global_load_lds as a FLAT access that may touch LDS
// and VMEM both, and while one is pending every later LDS dependency is waited for with lgkmcnt(0) (and VMEM ones with vmcnt(0)): the
// operand reads of the NEXT k-steps, just issued, are waited for too.  Issued from asm the piece is invisible to that bookkeeping (its
// completion is counted by hand: the s_waitcnt vmcnt(N) of the chunk loops), the compiler's own waits stay exact, and any vmcnt wait it
// emits for its own loads can only be stricter than needed, never weaker (the counter retires in issue order).  M0 = LDS base, written
// in the same statement that reads it (the compiler does not preserve M0 across statements and uses it for nothing else here).
__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
template <int IMM>
__device__ __forceinline__ void glds16_asm(const char *sbase, uint32_t voff, uint32_t lds_base)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                 :: "v"(voff), "s"(sbase), "s"(lds_base), "i"(IMM) : "memory");
}

// Piece q (0..7) of a wave's 8-KiB stream (k_act's free-running forms), immediate (q - 4) KiB on both addresses.
__device__ __forceinline__ void glds16_asm_piece8(const char *sbase, uint32_t voff, uint32_t lds_base, int q)
{
    switch (q) {
    case 0: glds16_asm<-4096>(sbase, voff, lds_base); break;
    case 1: glds16_asm<-3072>(sbase, voff, lds_base); break;
    case 2: glds16_asm<-2048>(sbase, voff, lds_base); break;
    case 3: glds16_asm<-1024>(sbase, voff, lds_base); break;
    case 4: glds16_asm<0>(sbase, voff, lds_base); break;
    case 5: glds16_asm<1024>(sbase, voff, lds_base); break;
    case 6: glds16_asm<2048>(sbase, voff, lds_base); break;
    default: glds16_asm<3072>(sbase, voff, lds_base); break;
    }
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

#ifdef SHEMS_STAMP_ACT
#define PSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { reinterpret_cast<unsigned long long *>(A.block_reward)[2*(i)] = __builtin_amdgcn_s_memtime(); reinterpret_cast<unsigned long long *>(A.block_reward)[2*(i)+1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define PSTAMP(i)
#endif
#ifdef SHEMS_STAMP_ACT
#define TSTAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); PSTAMP(i); } while (0)
#else
#define TSTAMP(i)
#endif

// The random draws of one act() call for env i: they do not depend on the actor's output, so a caller with latency to hide makes them
// early (k_act's small-tile forms, during stage 0).  eps-greedy: the three raw Philox words; Gaussian / OU: the standard-normal pair.
struct NoiseDraw { uint32_t a, b, c; };
__device__ __forceinline__ NoiseDraw noise_draw(const shems_act_params &p, int64_t i)
{
    NoiseDraw d = {0u, 0u, 0u};
    if (!p.train) return d;
    if (p.noise_kind == SHEMS_NOISE_EPS) {
        const u32x4 x = philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), p.tick, kStreamNoise, (uint32_t)p.seed, (uint32_t)(p.seed >> 32));
        d.a = x.x; d.b = x.y; d.c = x.z;
    } else {
        const float2 z = gauss_pair(p.seed, p.tick, i);
        d.a = __float_as_uint(z.x); d.b = __float_as_uint(z.y);
    }
    return d;
}

// What the env tail reads from global memory before it can step, fetched ahead by the small-tile forms of k_act (16 dwords per env in
// LDS): the next table row, h_countdown of the current row, idx, step, the config index and the noise draw.
struct TailPre {
    Row nx;
    float h_cur;
    int32_t idx, step, ci;
    NoiseDraw nz;
};
__device__ __forceinline__ void tailpre_store(float *lds, const TailPre &t)
{
    f32x4 *q = reinterpret_cast<f32x4 *>(lds);
    q[0] = f32x4{t.nx.h, t.nx.soc_ev, t.nx.d_e, t.nx.g_e};
    q[1] = f32x4{t.nx.p_buy, t.nx.h_cos, t.nx.h_sin, t.nx.season};
    q[2] = f32x4{t.h_cur, __int_as_float(t.idx), __int_as_float(t.step), __int_as_float(t.ci)};
    q[3] = f32x4{__uint_as_float(t.nz.a), __uint_as_float(t.nz.b), __uint_as_float(t.nz.c), 0.0f};
}
__device__ __forceinline__ TailPre tailpre_load(const float *lds)
{
    const f32x4 *q = reinterpret_cast<const f32x4 *>(lds);
    const f32x4 a = q[0], b = q[1], c = q[2], d = q[3];
    TailPre t;
    t.nx = Row{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    t.h_cur = c[0]; t.idx = __float_as_int(c[1]); t.step = __float_as_int(c[2]); t.ci = __float_as_int(c[3]);
    t.nz = NoiseDraw{__float_as_uint(d[0]), __float_as_uint(d[1]), __float_as_uint(d[2])};
    return t;
}

// One env after layer 3: p0, p1 = the pre-activation outputs (b3 included).  tanh, exploration noise, clamp, scale_action, step!,
// remember (DDPG.jl:148-184, 199-229).  Returns the env's reward (0 when nothing was stepped).
// obs_lds: the env's 9 raw observations in LDS (the caller staged them), or null = read them from the view.
// pre_lds: the env's TailPre block in LDS (the caller fetched it ahead), or null = draw / read in place.
__device__ __forceinline__ double act_env_tail(const ActArgs &A, int64_t i, float p0, float p1, int64_t learner, int64_t goff,
                                               const float *obs_lds, const float *pre_lds = nullptr)
{
    double reward = 0.0;
    {
        TailPre pre;
        if (pre_lds) pre = tailpre_load(pre_lds);
        const NoiseDraw nz = pre_lds ? pre.nz : noise_draw(A.p, i);
        p0 = tanhf(p0);
        p1 = tanhf(p1);
        float a0, a1, nmean = 0.0f;                                    // nmean: act()'s second return value
        if (A.p.train && A.p.noise_kind == SHEMS_NOISE_EPS) {          // DDPG.jl:161-170
            const bool explore = !(u01_24(nz.c) > A.p.eps);            // rng > eps: greedy; rng <= eps: uniform action
            a0 = explore ? (float)((double)nz.a * (1.0 / 4294967296.0) * 2.0 - 1.0) : p0;
            a1 = explore ? (float)((double)nz.b * (1.0 / 4294967296.0) * 2.0 - 1.0) : p1;
            nmean = explore ? 0.5f * (fabsf(p0 - a0) + fabsf(p1 - a1)) : 0.0f;          // mean(abs.(act_pred .- act_uni)) / 0f0
        } else {
            if (A.p.train) {
                const float2 z = make_float2(__uint_as_float(nz.a), __uint_as_float(nz.b));
                if (A.p.noise_kind == SHEMS_NOISE_OU) {                // DDPG.jl:49-55, 157-158
                    float2 X = reinterpret_cast<float2 *>(A.p.ou_state)[i];
                    const float sdt = A.p.noise_sigma * sqrtf(A.p.ou_dt);
                    X.x += A.p.ou_theta * (A.p.noise_mu - X.x) * A.p.ou_dt + sdt * z.x;
                    X.y += A.p.ou_theta * (A.p.noise_mu - X.y) * A.p.ou_dt + sdt * z.y;
                    reinterpret_cast<float2 *>(A.p.ou_state)[i] = X;
                    p0 += X.x;
                    p1 += X.y;
                    nmean = 0.5f * (X.x + X.y);
                } else {                                               // DDPG.jl:57-61, 159-160: Normal(mu, sigma_act)
                    const float n0 = A.p.noise_mu + A.p.noise_sigma * z.x, n1 = A.p.noise_mu + A.p.noise_sigma * z.y;
                    p0 += n0;
                    p1 += n1;
                    nmean = 0.5f * (n0 + n1);
                }
            }
            a0 = fminf(fmaxf(p0, -1.0f), 1.0f);                        // clamp.(act_pred .+ noise, -1f0, 1f0)
            a1 = fminf(fmaxf(p1, -1.0f), 1.0f);
        }
        TSTAMP(5);
        if (A.a_out) reinterpret_cast<float2 *>(A.a_out)[i] = make_float2(a0, a1);
        if (A.p.noise_acc) A.p.noise_acc[i] += nmean;
        if (A.do_step) {
            const shems_view &v = A.v;
            const shems_config c = pre_lds ? v.cfgs[pre.ci] : load_cfg(v, i);
            float obs[SHEMS_NSTATE], s0[SHEMS_NSTATE];
#pragma unroll
            for (int k = 0; k < SHEMS_NSTATE; ++k) { obs[k] = obs_lds ? obs_lds[k] : v.obs[i * SHEMS_NSTATE + k]; s0[k] = obs[k]; }
            int32_t idx = pre_lds ? pre.idx : v.idx[i], step = pre_lds ? pre.step : v.step[i];
            TSTAMP(6);
            StepFlows f;
            float B, EV, Bt, EVt;
            const bool ok = pre_lds ? env_advance_rows(c, pre.nx, pre.h_cur, obs, idx, step, scale_action(a0), scale_action(a1), SHEMS_TRACK_OFF,
                                                       reward, f, B, EV, Bt, EVt)
                                    : env_advance(c, v.tables, obs, idx, step, scale_action(a0), scale_action(a1), SHEMS_TRACK_OFF, reward, f,
                                                  B, EV, Bt, EVt);
            if (ok) {
                TSTAMP(7);
#pragma unroll
                for (int k = 0; k < SHEMS_NSTATE; ++k) v.obs[i * SHEMS_NSTATE + k] = obs[k];
                v.idx[i] = idx;
                v.step[i] = step;
                if (A.rewards) A.rewards[i] = reward;
                if (A.rewards_f32) A.rewards_f32[i] = (float)reward;
                if (A.returns_acc) A.returns_acc[i] += reward;
                if (A.use_ring) {
                    const int64_t nl = A.gcount > 1 ? A.genvs : v.n_envs;          // the window rotates inside a learner's env block
                    int64_t rel = (i - learner * nl) - A.win.offset;
                    rel %= nl;
                    if (rel < 0) rel += nl;
                    if (rel < A.win.count) {
                        shems_replay ring = A.ring;
                        if (A.gcount > 1) {
                            ring.s = gsh(ring.s, goff); ring.a = gsh(ring.a, goff); ring.r = gsh(ring.r, goff);
                            ring.s2 = gsh(ring.s2, goff); ring.done = gsh(ring.done, goff);
                        }
                        ring_push(ring, (A.win.pos + rel) % ring.capacity, s0, a0, a1, (float)reward, obs);
                    }
                }
            } else {
                reward = 0.0;
                raise(v.err, SHEMS_ERR_INDEX);
            }
        }
    }
    return reward;
}

// Scheduling pipeline of one chunk (one basic block): [DS reads of k-step 0], then per k-step
// [2*TM MFMAs] [DS reads of the NEXT k-step] [2*TM MFMAs] [one LDS-DMA piece]: the operand fetch sits in the middle of an
// MFMA group, half a group (~500 cycles) ahead of its first use.  ds_read2_b32 fetches two operands, so a k-step is
// 2 + ceil(TM/2) DS instructions.
constexpr int kDmaKs = 4;   // 4 waves: the 8 LDS-DMA pieces a wave issues per chunk go out in k-steps 0..3 (two each), so the
                            // last one has half a chunk (~4000 cycles) to land before the barrier that publishes it
template <int TM, int NA, bool DMA, int NKS>
__device__ __forceinline__ void sched_chunk()
{
    constexpr int DS = 2;                       // one vector read per operand
    __builtin_amdgcn_sched_group_barrier(0x100, DS, 0);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        __builtin_amdgcn_sched_group_barrier(0x008, NA * TM / 2, 0);
        if (ks + 1 < NKS) __builtin_amdgcn_sched_group_barrier(0x100, DS, 0);
        if (DMA && NA == 4 && ks < kDmaKs) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NA * TM - NA * TM / 2, 0);
        if (DMA && ((NA == 4 && ks < kDmaKs) || (NA != 4 && (ks & 1) == 0))) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
}

// Free-running form: after chunk c the wave needs its chunk c + 1; the LDS-DMA pieces of chunks c + 2 .. c + RD - 1 (8 each, 5 for
// the short chunk 15) are younger and may stay in flight.
template <int RD>
constexpr int free_keep(int c)
{
    int n = 0;
    for (int j = c + 2; j <= c + RD - 1; ++j) n += j > kChunks - 1 ? 0 : j == kChunks - 1 ? (kH1 - (kChunks - 1) * kKC) / 2 : 8;
    return n;
}

template <int TM, int NW, int RD>
__global__ __launch_bounds__(64 * NW, NW / 4) void k_act(ActArgs A)
{
    static_assert(RD == 0 || (NW == 4 && RD >= 2 && RD <= 4), "free-running form: 4 waves, ring of 2..4 chunks");
    constexpr int NT_ = 64 * NW;            // threads per workgroup
    constexpr int NA = 16 / NW;             // 32-wide n-tiles per wave (4 waves: 4, 8 waves: 2)
    PSTAMP(0);
    constexpr int BM = 32 * TM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Wc = reinterpret_cast<float *>(smem);             // RD = 0: [2][kWcFloats] W2 chunks (16 rows x 500); else [NW][RD][16][128]
    constexpr int HR = act_h_rows(TM, RD);                   // rows of relu(layer 1) resident at a time
    constexpr bool PRE = act_tail_pre(TM, RD);               // env-tail inputs fetched in stage 0
    constexpr bool PRE2 = RD != 0 && TM == 4;                // 128-env tiles: indices in stage 0, rows + noise while the second half of h1
                                                             // is laid down, TailPre blocks over the then dead layer-1 operand image
    float *Hc = Wc + (RD ? NW * RD * kFreeChunkFloats + 16 : 2 * kWcFloats);   // [HR][BM] relu(layer 1)
    float *xT = Hc + HR * BM;                                // [10][BM]  normalised obs (rows 0..8), row 9 = 1 (bias)
    float *w1 = xT + kW1K * BM;                              // [10][256] layer-1 operand image
    float *tl = w1 + kW1K * kW1C;                            // b2 [512], W3 [512][2], b3 [2]
    // Layer-3 group sums [8 column groups][BM][2]: kept over the W2 stream, which is dead by then -- a wave's groups at the start of
    // ITS OWN ring (free-running form: nobody else touches it) / of its slice of the shared double buffer (dead for everybody after the
    // last chunk's barrier).  red_of(g) = where the owner of group g put it.
    constexpr int kRedStride = RD ? RD * kFreeChunkFloats : (16 / NW / 2) * BM * kOut;
    static_assert((16 / NW / 2) * BM * kOut <= (RD ? RD * kFreeChunkFloats : 2 * kWcFloats / NW), "group sums fit the dead W2 buffer");
#define RED_OF(g) (Wc + ((g) / (NA / 2)) * kRedStride + ((g) % (NA / 2)) * (BM * kOut))
    float *xR = tl + (kTailFloats + 2);                        // [BM][9]   the raw observations stage 0 loaded: step! starts from these, not from a
                                                             //           second (stride-36-byte) read of global memory at the end of the kernel
    [[maybe_unused]] float *xP = xR + BM * kIn;              // PRE: [BM][kPreDw] TailPre blocks

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave index as a scalar: LDS-DMA bases (M0) stay on the SALU
    const int li = lane & 31, lh = lane >> 5;
    // Learner groups: the workgroups of one learner stream the same 500 KB W2, so they should share an L2.  Workgroup ids
    // go round-robin over the 8 XCDs; remapping id -> (id % 8) * (grid / 8) + id / 8 gives each XCD a contiguous range of
    // tiles (= a few whole learners) instead of a slice of every learner.
    int64_t bid = blockIdx.x;
    if (A.gcount > 1 && (gridDim.x & 7) == 0) bid = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int64_t env0 = A.m0 + bid * BM;
    const int64_t learner = A.gcount > 1 ? env0 / A.genvs : 0;
    const int64_t goff = learner * A.gstride;
    const float *__restrict__ P = gsh(A.p.actor, goff);
    const float *__restrict__ s_min = gsh(A.p.s_min, goff), *__restrict__ s_max = gsh(A.p.s_max, goff);

    // ---- stage 0: x = normalize(s) -> xT[k][m]; layer-1 image, b2/W3/b3 -> LDS; W2 chunk 0 -> LDS --------------
    // W2 chunk staging, global -> LDS directly (global_load_lds_dwordx4: no staging registers).  A chunk is 16 rows =
    // 32000 contiguous bytes = 31 full 1-KiB wave pieces + one of 256 B (16 lanes); wave w issues pieces w, w+4, ...
    // The LDS image is the linear copy (destination = M0 base + lane*16).  The last chunk holds only rows 240..249
    // (20000 B): its rows 10..15 receive clamped-source filler that is never read (the last chunk runs 5 k-steps).
    const char *W2g = reinterpret_cast<const char *>(P + kOffW2);
    constexpr int kChunkBytes = kKC * kH2 * 4;
    constexpr int kTotalBytes = kH1 * kH2 * 4;
#define W2_PIECE(chunk, buf, pc)                                                                  \
    do {                                                                                          \
        const int off_ = (pc) * 1024 + lane * 16;                                                 \
        if (off_ < kChunkBytes && (chunk) * kChunkBytes + off_ < kTotalBytes)                     \
            glds16(W2g + (chunk) * kChunkBytes + off_,                                            \
                   reinterpret_cast<char *>(Wc + (buf) * kWcFloats) + (pc) * 1024);               \
    } while (0)
#define W2_ISSUE(chunk, buf)                                                                      \
    do { _Pragma("unroll") for (int pc_ = wave; pc_ < 32; pc_ += NW) W2_PIECE(chunk, buf, pc_); } while (0)
    // Free-running form: wave w moves rows [16 c, +16) x columns [128 w, +128) of W2 -- piece q = rows 2q, 2q + 1 (lanes 0..31 /
    // 32..63), 512 B each -- into its own ring.  Columns 500..511 of wave 3 run into the next row (the last row: into b2); those
    // accumulator columns meet zero W3 rows.  Rows >= 250 (pieces 5..7 of chunk 15) are never read and never fetched.
    float *Wf = Wc + wave * (RD * kFreeChunkFloats);
    // One address pair per chunk: the chunk's base is wave-uniform (SGPR pair), a lane's 32-bit offset for piece q is constant over the
    // whole kernel (8 registers), and the instruction's immediate -- added to the global AND the LDS address -- selects the LDS piece:
    //   global = (W2 + 512 w + 32000 c) + [lane part + 4000 q - 1024 (q - 4)] + 1024 (q - 4),   LDS = (ring buffer + 4096) + 1024 (q - 4)
    // Tiled working layout of a learner group (A.w2t: [kt][nt][m | v | p | target][64][64], 64 KB per tile): the wave's 128 columns are the
    // n-tiles 2 w and 2 w + 1, chunk c = rows 16 (c & 3) .. of k-tile c / 4, a row of a tile 256 contiguous bytes.  The same three-part
    // address -- wave-uniform chunk base, a lane's constant per piece, the immediate -- with other constants: a piece (rows 2 q, 2 q + 1
    // x 128 columns = 1 KB of LDS) takes lanes 0..15 / 16..31 from the two tiles' row 2 q and lanes 32..63 likewise from row 2 q + 1.
    // Pad rows / columns are zeros there (nothing runs into the next row).
    const bool tiled = A.w2t != nullptr;
    constexpr size_t kTlTile = 4 * 64 * 64 * 4, kTlP = 2 * 64 * 64 * 4;          // bytes: one tile's four arrays; offset of its p array
    const char *wsbase = tiled ? reinterpret_cast<const char *>(gsh(A.w2t, goff)) + kTlP + (size_t)wave * (2 * kTlTile) : W2g + wave * 512;
    uint32_t wvoff[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
        wvoff[q] = tiled ? (uint32_t)(((lane & 31) >> 4) * kTlTile + (lane >> 5) * 256 + (lane & 15) * 16 + q * 512 - (q - 4) * 1024)
                         : (uint32_t)((lane >> 5) * (kH2 * 4) + (lane & 31) * 16 + q * (2 * kH2 * 4) - (q - 4) * 1024);
    const uint32_t wf_lds = lds_addr(Wf) + 4 * 1024;          // piece 4 of ring buffer 0
    // (asm LDS-DMA: see glds16_asm -- the compiler's LDS / VMEM waits stay exact)
#define FREE_CHUNK_OFF(chunk) (tiled ? (size_t)((chunk) >> 2) * (8 * kTlTile) + (size_t)((chunk) & 3) * (kKC * 256) : (size_t)(chunk) * (kKC * kH2 * 4))
#define FREE_PIECE(chunk, q)                                                                      \
    glds16_asm_piece8(wsbase + FREE_CHUNK_OFF(chunk), wvoff[q],                                   \
                      wf_lds + ((chunk) % (RD ? RD : 1)) * (kFreeChunkFloats * 4), (q))
#define STAGE0_DMA()                                                                              \
    do {                                                                                          \
        if constexpr (RD == 0) W2_ISSUE(0, 0);                                                    \
        else {                                                                                    \
            _Pragma("unroll") for (int ch_ = 0; ch_ < RD - 1; ++ch_)                              \
                _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) FREE_PIECE(ch_, q_);             \
        }                                                                                         \
    } while (0)
    // Every global load of the stage is issued before the first value is used (obs + normalisation, layer-1 image, b2|W3|b3;
    // the first W2 chunk goes out by LDS-DMA right behind them): one exposed latency instead of one per block.  Addresses are
    // clamped, never predicated -- a guarded load becomes a branch with its own s_waitcnt and serialises the batch.
    constexpr int kIt = (BM * kIn + NT_ - 1) / NT_;
    float sv[kIt], lo[kIt], hi[kIt], wv[kW1K], tv[6];
    const int64_t last = A.m * kIn - 1;
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int e = min(it * NT_ + tid, BM * kIn - 1), k = e % kIn;
        sv[it] = A.obs[min(env0 * kIn + e, last)];
        lo[it] = s_min[k];
        hi[it] = s_max[k];
    }
    {   // w1[j][k]: j < 9 -> W1[j][k], j == 9 -> b1[k]; columns 250..255 zero (thread = column k)
        const int kc = min(tid, kH1 - 1);
#pragma unroll
        for (int j = 0; j < kW1K; ++j) wv[j] = P[(j == kW1K - 1 ? kIn : min(j, kIn - 1)) * kH1 + kc];
    }
    // b2 | W3 | b3 are contiguous in the parameter block (1502 floats)
#pragma unroll
    for (int it = 0; it < 6; ++it) tv[it] = P[kOffB2 + min(it * 256 + (tid & 255), kH2 + kH2 * kOut + kOut - 1)];
    // Small tiles: what the env tail needs from global memory is fetched now, and its random draws are made while stage 0 waits for
    // its loads (TailPre; at one 32-env tile per workgroup the tail is ~10 % of the kernel and a chain of three dependent global reads).
    // Every thread does it for env tid % BM (identical values, unconditional LDS stores: a predicated load would be sunk into a branch
    // behind an s_waitcnt vmcnt(0)); without a view (actor forward only) the pointers fall back to the parameter block.
    [[maybe_unused]] TailPre tp;
    [[maybe_unused]] const float *tp_tables = nullptr;
    [[maybe_unused]] int64_t tp_row = 0;
    if constexpr (PRE || PRE2) {
        const int64_t pe = min(env0 + (tid & (BM - 1)), A.m - 1);
        const bool view = A.do_step != 0;
        const int32_t *pidx = view ? A.v.idx : reinterpret_cast<const int32_t *>(P), *pstep = view ? A.v.step : reinterpret_cast<const int32_t *>(P);
        const uint16_t *pci = view && A.v.n_cfg > 1 ? A.v.cfg_of_env : reinterpret_cast<const uint16_t *>(P);
        tp.idx = pidx[view ? pe : 0];
        tp.step = pstep[view ? pe : 0];
        tp.ci = view && A.v.n_cfg > 1 ? (int)pci[pe] : 0;
    }
    if constexpr (RD == 0) STAGE0_DMA();
    if constexpr (PRE) tp.nz = noise_draw(A.p, env0 + (tid & (BM - 1)));
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int e = it * NT_ + tid, m = e / kIn, k = e - m * kIn;
        const float x = (sv[it] - lo[it]) / ((hi[it] - lo[it]) + 1e-8f);      // MPS:56
        if (e < BM * kIn) { xT[k * BM + m] = env0 * kIn + e <= last ? x : 0.0f; xR[e] = sv[it]; }
    }
    for (int e = tid; e < BM; e += NT_) xT[kIn * BM + e] = 1.0f;                               // row 9 = 1: the bias input
    if (tid < kW1C) {
#pragma unroll
        for (int j = 0; j < kW1K; ++j) w1[j * kW1C + tid] = ((j < kIn || j == kW1K - 1) && tid < kH1) ? wv[j] : 0.0f;
    }
    {
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int e = tid < 256 ? it * 256 + tid : 1 << 20;   // source index: [0,500) b2, [500,1500) W3, [1500,1502) b3
            if (e < kH2) tl[e] = tv[it];
            else if (e < kH2 + kH2 * kOut) tl[kH2P + (e - kH2)] = tv[it];
            else if (e < kH2 + kH2 * kOut + kOut) tl[kH2P + kH2P * kOut + (e - kH2 - kH2 * kOut)] = tv[it];
        }
        if (tid < kH2P - kH2) tl[kH2 + tid] = 0.0f;                                   // pad rows of b2
        if (tid < (kH2P - kH2) * kOut) tl[kH2P + kH2 * kOut + tid] = 0.0f;            // pad rows of W3
    }
    if (RD == 0 && tid < 16) { Wc[kKC * kH2 + tid] = 0.0f; Wc[kWcFloats + kKC * kH2 + tid] = 0.0f; }

    // Layer 1 on the matrix pipe (K = 10 = 5 k-steps): rows [32g, 32g+32) x this workgroup's BM columns into Hc[g & 1];
    // wave w owns column tile w (TM <= 4 tiles).  D layout: row (r&3)+8(r>>2)+4*lh, column lane&31.
/* A operand of k-step s_ (input row j_ = 2 s_ + lh) of row group g: from the LDS image w1 here; k_actg redefines it (registers). */
#define L1_A(s_, j_, g, which) w1[(j_) * kW1C + 32 * (g) + li]
#define L1_TILE(g, b, dst)                                                                        \
    do {                                                                                          \
            float a_[kW1K / 2], b_[kW1K / 2];                                                     \
            _Pragma("unroll") for (int s_ = 0; s_ < kW1K / 2; ++s_) {                             \
                const int j_ = 2 * s_ + lh;                                                       \
                a_[s_] = L1_A(s_, j_, g, 0);                                                      \
                b_[s_] = xT[j_ * BM + TM * li + (b)];             /* env column m = TM*j + tile */  \
            }                                                                                     \
            /* The accumulator of this tile must stay in VGPRs: given the builtin, the compiler parks it in a[0:15] and moves */ \
            /* a layer-2 accumulator tile out and back around every group (drain + 48 register moves).  Wait states are ours */ \
            /* inside the string: VALU-written operand -> MFMA (1), chain C = previous D (0), D -> VALU reader (16-pass: 19+). */ \
            f32x16 t_;                                                                            \
            asm volatile("s_nop 1\n\t"                                                            \
                         "v_mfma_f32_32x32x2_f32 %0, %1, %6, 0\n\t"                               \
                         "v_mfma_f32_32x32x2_f32 %0, %2, %7, %0\n\t"                              \
                         "v_mfma_f32_32x32x2_f32 %0, %3, %8, %0\n\t"                              \
                         "v_mfma_f32_32x32x2_f32 %0, %4, %9, %0\n\t"                              \
                         "v_mfma_f32_32x32x2_f32 %0, %5, %10, %0\n\t"                             \
                         "s_nop 15\n\ts_nop 7"                                                    \
                         : "=&v"(t_)                                                              \
                         : "v"(a_[0]), "v"(a_[1]), "v"(a_[2]), "v"(a_[3]), "v"(a_[4]),            \
                           "v"(b_[0]), "v"(b_[1]), "v"(b_[2]), "v"(b_[3]), "v"(b_[4]));           \
            float *dst_ = (dst);                                                                  \
            _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_)                                     \
                dst_[((r_ & 3) + 8 * (r_ >> 2) + 4 * lh) * BM] = fmaxf(t_[r_], 0.0f);             \
    } while (0)
/* Two tiles at once: the two 5-MFMA chains interleaved (each MFMA's C is the D of the one two back: no dependent-issue stall), */
/* one operand fetch and one D -> VALU wait for both. */
#define L1_TILE2(g0, b0, dst0, g1, b1, dst1)                                                      \
    do {                                                                                          \
            float a0_[kW1K / 2], c0_[kW1K / 2], a1_[kW1K / 2], c1_[kW1K / 2];                     \
            _Pragma("unroll") for (int s_ = 0; s_ < kW1K / 2; ++s_) {                             \
                const int j_ = 2 * s_ + lh;                                                       \
                a0_[s_] = L1_A(s_, j_, g0, 0);                                                    \
                c0_[s_] = xT[j_ * BM + TM * li + (b0)];                                           \
                a1_[s_] = L1_A(s_, j_, g1, 1);                                                    \
                c1_[s_] = xT[j_ * BM + TM * li + (b1)];                                           \
            }                                                                                     \
            f32x16 t0_, t1_;                                                                      \
            asm volatile("s_nop 1\n\t"                                                            \
                         "v_mfma_f32_32x32x2_f32 %0, %2, %7, 0\n\t"                               \
                         "v_mfma_f32_32x32x2_f32 %1, %12, %17, 0\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %0, %3, %8, %0\n\t"                              \
                         "v_mfma_f32_32x32x2_f32 %1, %13, %18, %1\n\t"                            \
                         "v_mfma_f32_32x32x2_f32 %0, %4, %9, %0\n\t"                              \
                         "v_mfma_f32_32x32x2_f32 %1, %14, %19, %1\n\t"                            \
                         "v_mfma_f32_32x32x2_f32 %0, %5, %10, %0\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %1, %15, %20, %1\n\t"                            \
                         "v_mfma_f32_32x32x2_f32 %0, %6, %11, %0\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %1, %16, %21, %1\n\t"                            \
                         "s_nop 15\n\ts_nop 7"                                                    \
                         : "=&v"(t0_), "=&v"(t1_)                                                 \
                         : "v"(a0_[0]), "v"(a0_[1]), "v"(a0_[2]), "v"(a0_[3]), "v"(a0_[4]),       \
                           "v"(c0_[0]), "v"(c0_[1]), "v"(c0_[2]), "v"(c0_[3]), "v"(c0_[4]),       \
                           "v"(a1_[0]), "v"(a1_[1]), "v"(a1_[2]), "v"(a1_[3]), "v"(a1_[4]),       \
                           "v"(c1_[0]), "v"(c1_[1]), "v"(c1_[2]), "v"(c1_[3]), "v"(c1_[4]));      \
            float *d0_ = (dst0), *d1_ = (dst1);                                                   \
            _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) {                                   \
                d0_[((r_ & 3) + 8 * (r_ >> 2) + 4 * lh) * BM] = fmaxf(t0_[r_], 0.0f);             \
                d1_[((r_ & 3) + 8 * (r_ >> 2) + 4 * lh) * BM] = fmaxf(t1_[r_], 0.0f);             \
            }                                                                                     \
    } while (0)
#define L1_GROUP(g)                                                                               \
    do { if (wave < TM) L1_TILE(g, wave, Hc + ((g) & 1) * (32 * BM) + TM * li + wave); } while (0)

    if constexpr (PRE || PRE2) {
        const bool view = A.do_step != 0;
        const int32_t *pc = view ? reinterpret_cast<const int32_t *>(A.v.cfgs + tp.ci) : reinterpret_cast<const int32_t *>(P);
        constexpr int o_row0 = offsetof(shems_config, table_row0) / 4, o_nrow = offsetof(shems_config, nrow) / 4;
        const int32_t row0 = pc[o_row0], nrow = pc[o_nrow];
        tp_tables = view ? A.v.tables : P;
        // row idx + 1 (1-based) of the env's table, clamped into the table: env_advance_rows rejects idx + 1 > nrow before it looks at it
        tp_row = view ? (int64_t)row0 + max(min(tp.idx + 1, nrow), 2) - 1 : 1;
    }
    // xT, w1 visible.  Only LDS stores are being published: a barrier that does not also drain the W2 pieces in flight (what
    // __syncthreads() would do with its vmcnt(0))
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // The ring's first chunks go out only now: issued with stage 0's loads, 256 workgroups x 64 KB of pieces queue in front of the
    // few KB every workgroup is actually waiting for (measured: -0.6 % at 65 536 envs, -1.7 % at 8 192).
    // (small tiles: behind the TailPre row loads below, so that waiting for those does not mean waiting for the pieces)
    if constexpr (RD != 0 && !PRE) STAGE0_DMA();
    PSTAMP(1);
    /* TM = 4: one wave lays down a whole 32-row group for all four column tiles.  A lane's four env columns m = 4 j + b are adjacent: */
    /* the B operands come as one b128 read per k-step and the result leaves as one b128 store per row (the per-tile form's b32      */
    /* accesses at a 16-byte lane stride are 4-way bank conflicts).  20 MFMAs, k-step outer / tile inner: C is the D of four back.   */
#define L1_GROUP4(g, dst)                                                                         \
    do {                                                                                          \
            float a_[kW1K / 2];                                                                   \
            f32x4 x_[kW1K / 2];                                                                   \
            _Pragma("unroll") for (int s_ = 0; s_ < kW1K / 2; ++s_) {                             \
                const int j_ = 2 * s_ + lh;                                                       \
                a_[s_] = w1[j_ * kW1C + 32 * (g) + li];                                           \
                x_[s_] = *reinterpret_cast<const f32x4 *>(xT + j_ * BM + 4 * li);                 \
            }                                                                                     \
            f32x16 t0_, t1_, t2_, t3_;                                                            \
            asm volatile("s_nop 1\n\t"                                                            \
                         "v_mfma_f32_32x32x2_f32 %0, %4, %9, 0\n\t"                               \
                         "v_mfma_f32_32x32x2_f32 %1, %4, %10, 0\n\t"                              \
                         "v_mfma_f32_32x32x2_f32 %2, %4, %11, 0\n\t"                              \
                         "v_mfma_f32_32x32x2_f32 %3, %4, %12, 0\n\t"                              \
                         "v_mfma_f32_32x32x2_f32 %0, %5, %13, %0\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %1, %5, %14, %1\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %2, %5, %15, %2\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %3, %5, %16, %3\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %0, %6, %17, %0\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %1, %6, %18, %1\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %2, %6, %19, %2\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %3, %6, %20, %3\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %0, %7, %21, %0\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %1, %7, %22, %1\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %2, %7, %23, %2\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %3, %7, %24, %3\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %0, %8, %25, %0\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %1, %8, %26, %1\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %2, %8, %27, %2\n\t"                             \
                         "v_mfma_f32_32x32x2_f32 %3, %8, %28, %3\n\t"                             \
                         "s_nop 15\n\ts_nop 7"                                                    \
                         : "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_)                         \
                         : "v"(a_[0]), "v"(a_[1]), "v"(a_[2]), "v"(a_[3]), "v"(a_[4]),            \
                           "v"(x_[0][0]), "v"(x_[0][1]), "v"(x_[0][2]), "v"(x_[0][3]),            \
                           "v"(x_[1][0]), "v"(x_[1][1]), "v"(x_[1][2]), "v"(x_[1][3]),            \
                           "v"(x_[2][0]), "v"(x_[2][1]), "v"(x_[2][2]), "v"(x_[2][3]),            \
                           "v"(x_[3][0]), "v"(x_[3][1]), "v"(x_[3][2]), "v"(x_[3][3]),            \
                           "v"(x_[4][0]), "v"(x_[4][1]), "v"(x_[4][2]), "v"(x_[4][3]));           \
            float *d_ = (dst) + 4 * li;                                                           \
            _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_)                                     \
                *reinterpret_cast<f32x4 *>(d_ + ((r_ & 3) + 8 * (r_ >> 2) + 4 * lh) * BM) =       \
                    f32x4{fmaxf(t0_[r_], 0.0f), fmaxf(t1_[r_], 0.0f), fmaxf(t2_[r_], 0.0f), fmaxf(t3_[r_], 0.0f)}; \
    } while (0)
    // One phase of relu(layer 1) in the free-running form: row groups gbase .. gbase + HR / 32 - 1, TM column tiles each, over the four
    // waves in interleaved pairs
#define L1_PHASE(gbase)                                                                           \
    do {                                                                                          \
        if constexpr (TM == 4) {                       /* HR / 32 = 4 groups: one per wave */      \
            const int g_ = (gbase) + wave;                                                        \
            L1_GROUP4(g_, Hc + ((g_ * 32) % HR) * BM);                                            \
        } else                                                                                    \
        _Pragma("unroll") for (int u = 0; u < (HR / 32) * TM / 8; ++u) {                          \
            const int t0 = wave + 8 * u, g0 = (gbase) + t0 / TM, b0 = t0 % TM, t1 = t0 + 4, g1 = (gbase) + t1 / TM, b1 = t1 % TM; \
            L1_TILE2(g0, b0, Hc + ((g0 * 32) % HR) * BM + TM * li + b0, g1, b1, Hc + ((g1 * 32) % HR) * BM + TM * li + b1); \
        }                                                                                         \
    } while (0)
    if constexpr (RD == 0) L1_GROUP(0);
    else if constexpr (PRE) {
        const f32x4 *rp = reinterpret_cast<const f32x4 *>(tp_tables + tp_row * SHEMS_NCOL);
        const f32x4 ra = rp[0], rb = rp[1];                                    // row idx + 1
        tp.h_cur = tp_tables[(tp_row - 1) * SHEMS_NCOL];                       // h_countdown of row idx
        STAGE0_DMA();
        L1_PHASE(0);
        tp.nx = Row{ra[0], ra[1], ra[2], ra[3], rb[0], rb[1], rb[2], rb[3]};
        tailpre_store(xP + (tid & (BM - 1)) * kPreDw, tp);
    } else L1_PHASE(0);
    if constexpr (RD == 0) __syncthreads();                                   // + chunk 0 of the shared stream
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // h1 only: every wave waits for its own ring below
    PSTAMP(2);

    // ---- layer 2: 128 k-steps of 4 x TM MFMA tiles per wave ----------------------------------------------------
    f32x16 acc[NA][TM];
    // Tile (a, b) of a wave covers the canonical columns act_col(nbase, a, i) = nbase + 64 (a >> 1) + 2 i + (a & 1) and the envs
    // m = TM*j + b (i, j = the MFMA row / column index of a lane): the weights a lane needs for a k-step are 8 contiguous bytes per
    // column group (NA = 4: two groups, one ds_read2_b64), its TM activations one vector read -- a fixed DS count per k-step for the
    // pinned schedule.
    typedef typename FVec<NA>::type AVec;
    typedef typename FVec<TM>::type BVec;
    const int nbase = wave * (32 * NA);
    const char *wbase_ = W2g + kChunkBytes + (8 * wave + 4) * 1024 + lane * 16;      // piece 8w+4 of chunk 1: the base of this wave's LDS-DMA stream
    // the accumulators start at b2[n] (rows >= 500: 0), so the epilogue is relu + two FMAs per element
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bias = tl[act_col<NA>(nbase, a, (r & 3) + 8 * (r >> 2) + 4 * lh)];   // canonical column of tile a, MFMA row i
#pragma unroll
            for (int b = 0; b < TM; ++b) acc[a][b][r] = bias;
        }

    // One chunk = 8 k-steps x (4 x TM) MFMAs.  The operands of k-step ks+1 are fetched from LDS before the MFMAs of
    // k-step ks issue (register double buffer; sched_group_barrier pins that order -- left alone the compiler sinks each
    // ds_read to just before its first use and the wave, alone on its SIMD, eats the LDS latency 16 times per chunk).
    // ISSUE: 0 = the next chunk is a whole 32-piece chunk (branch-free LDS-DMA, one piece per k-step), 1 = the next chunk is
    // the last one (rows 240..249 only: source addresses clamped to the end of W2, still branch-free), 2 = nothing to fetch.  ODD: this chunk also produces the next layer-1 group.
#define CHUNK_BODY(c, ISSUE, ODD, NKS)                                                                             \
    do {                                                                                                        \
        const int cur_ = (c) & 1, nxt_ = cur_ ^ 1;                                                              \
        if (ODD) L1_GROUP(((c) + 1) >> 1);                                                                      \
        const float *Wb_ = Wc + cur_ * kWcFloats + nbase;                                                       \
        const float *Hb_ = Hc + (((c) >> 1) & 1) * (32 * BM) + ((c) & 1) * (kKC * BM) + TM * li;                \
        AVec af_[2];                                                                                            \
        BVec bf_[2];                                                                                            \
        af_[0] = act_load_a<NA>(Wb_ + lh * kH2, li);                                                            \
        bf_[0] = *reinterpret_cast<const BVec *>(Hb_ + lh * BM);                                                \
        _Pragma("unroll") for (int ks = 0; ks < (NKS); ++ks) {                                                  \
            if (ks + 1 < (NKS)) {                                                                               \
                const int kr_ = 2 * (ks + 1) + lh;                                                              \
                af_[(ks + 1) & 1] = act_load_a<NA>(Wb_ + kr_ * kH2, li);                                        \
                bf_[(ks + 1) & 1] = *reinterpret_cast<const BVec *>(Hb_ + kr_ * BM);                            \
            }                                                                                                   \
            _Pragma("unroll") for (int a = 0; a < NA; ++a)                                                      \
                _Pragma("unroll") for (int b = 0; b < TM; ++b)                                                  \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fvec_get<NA>(af_[ks & 1], a), fvec_get<TM>(bf_[ks & 1], b), acc[a][b], 0, 0, 0); \
            _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                  \
                if (NA == 4 ? ks < kDmaKs : ((ks & 1) == 0 && h_ == 0)) {                                       \
                    const int pc_ = NA == 4 ? wave + 4 * (2 * ks + h_) : wave + 8 * (ks >> 1);                  \
                    if (ISSUE == 0 && NA == 4) {                                                                \
                        /* wave w moves the 8 consecutive pieces 8w..8w+7: one 64-bit address per chunk, the */   \
                        /* piece selected by the instruction's immediate offset (-4096 .. 3072, both sides)  */   \
                        glds16_piece(wbase_ + (size_t)(c) * kChunkBytes,                                        \
                                     reinterpret_cast<char *>(Wc + nxt_ * kWcFloats) + (8 * wave + 4) * 1024, 2 * ks + h_); \
                    } else if (ISSUE == 0)                                                                      \
                        glds16(W2g + ((c) + 1) * kChunkBytes + pc_ * 1024 + lane * 16,                          \
                               reinterpret_cast<char *>(Wc + nxt_ * kWcFloats) + pc_ * 1024);                   \
                    else if (ISSUE == 1)                 /* short last chunk: source clamped to the end of W2 */ \
                        glds16(W2g + min(((c) + 1) * kChunkBytes + pc_ * 1024 + lane * 16, kTotalBytes - 16),   \
                               reinterpret_cast<char *>(Wc + nxt_ * kWcFloats) + pc_ * 1024);                   \
                }                                                                                               \
            }                                                                                                   \
        }                                                                                                       \
        sched_chunk<TM, NA, ISSUE != 2, NKS>();                                                   \
        __syncthreads();                                                                                        \
    } while (0)

    // Free-running form: chunk c of a wave = 8 k-steps on its own ring buffer c % RD and rows [16 c, +16) of the resident h1; the
    // pieces of chunk c + RD - 1 go out during k-steps 0..3 into the buffer chunk c - 1 was read from; at the end the wave waits for
    // ITS OWN chunk c + 1 (vmcnt counts LDS-DMA in issue order: the pieces of younger chunks may stay in flight).  No barrier.
#define FREE_NPIECES(ch) ((ch) > kChunks - 1 ? 0 : (ch) == kChunks - 1 ? (kH1 - (kChunks - 1) * kKC) / 2 : 8)
    /* k-step K = 8 c + ks of the whole layer (0 .. 124): operands sit in register buffer K % 3 and were requested two k-steps ago. */ \
    /* One scheduling region per k-step: [first MFMA: its lgkmcnt wait comes before the new reads are issued] [operand reads of    */ \
    /* k-step K + 2] [the other MFMAs] [two LDS-DMA pieces].                                                                     */
#define FREE_KSTEP(c, ks)                                                                                       \
    do {                                                                                                        \
        const int K_ = 8 * (c) + (ks), K2_ = K_ + 2;                                                            \
        /* a half-resident h1 (TM = 4): nothing of the second half is requested before it has been laid down */ \
        const int klim_ = (HR < 256 && (c) < kChunks / 2) ? (kChunks / 2) * (kKC / 2) : kFreeKsteps;            \
        if (K2_ < klim_) {                                                                                      \
            const int c2_ = K2_ >> 3, kr_ = 2 * (K2_ & 7) + lh;                                                 \
            af_[K2_ % 3] = act_load_a<NA>(Wf + (c2_ % RD) * kFreeChunkFloats + kr_ * 128, li);                  \
            bf_[K2_ % 3] = *reinterpret_cast<const BVec *>(Hc + ((c2_ * kKC) % HR + kr_) * BM + TM * li);       \
        }                                                                                                       \
        /* quarters of the k-step's MFMAs = the four tile rows a; the operand reads of k-step K + 2 go out between the first two, */ \
        /* an LDS-DMA piece (asm) after the third and after the fourth */                                       \
        _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                           \
            _Pragma("unroll") for (int b = 0; b < TM; ++b)                                                      \
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fvec_get<NA>(af_[K_ % 3], a), fvec_get<TM>(bf_[K_ % 3], b), acc[a][b], 0, 0, 0); \
        {                                                                                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                  \
            __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
        }                                                                                                       \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                      \
            _Pragma("unroll") for (int b = 0; b < TM; ++b)                                                      \
                acc[2 + h_][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fvec_get<NA>(af_[K_ % 3], 2 + h_), fvec_get<TM>(bf_[K_ % 3], b), acc[2 + h_][b], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            if ((ks) < kDmaKs && 2 * (ks) + h_ < FREE_NPIECES((c) + RD - 1)) {                                  \
                FREE_PIECE((c) + RD - 1, 2 * (ks) + h_);                                                        \
                __builtin_amdgcn_sched_barrier(0);                                                              \
            }                                                                                                   \
        }                                                                                                       \
    } while (0)
    /* Chunk c: k-steps 0..5 read inside the chunk; before k-step 6 (whose reads open chunk c + 1) the wave waits for ITS OWN next   */ \
    /* chunk, issued RD - 1 chunks ago -- the pieces of younger chunks may stay in flight.                                         */
#define FREE_CHUNK(c, NKS)                                                                                      \
    do {                                                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < ((NKS) < 6 ? (NKS) : 6); ++ks) FREE_KSTEP(c, ks);               \
        if ((c) + 1 < kChunks) {                                                                                \
            constexpr int keep_ = free_keep<RD>(c);                                                             \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(keep_) : "memory");                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
        }                                                                                                       \
        _Pragma("unroll") for (int ks = 6; ks < (NKS); ++ks) FREE_KSTEP(c, ks);                                 \
    } while (0)

    if constexpr (RD == 0) {
#pragma unroll 1
    for (int cp = 0; cp < (kChunks - 2) / 2; ++cp) {     // chunks 0..13: the next chunk (1..14) is whole
        CHUNK_BODY(2 * cp, 0, false, kKC / 2);
        CHUNK_BODY(2 * cp + 1, 0, true, kKC / 2);
    }
    PSTAMP(3);
    CHUNK_BODY(kChunks - 2, 1, false, kKC / 2);           // chunk 14 fetches the short last chunk
    PSTAMP(4);
    CHUNK_BODY(kChunks - 1, 2, false, (kH1 - (kChunks - 1) * kKC) / 2);   // chunk 15: rows 240..249 only = 5 k-steps
    } else {
        {
            constexpr int keep0_ = free_keep<RD>(-1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(keep0_) : "memory");        // this wave's chunk 0 has landed
            __builtin_amdgcn_sched_barrier(0);
        }
        constexpr int kFreeKsteps = (kChunks - 1) * (kKC / 2) + (kH1 - (kChunks - 1) * kKC) / 2;      // 125 k-steps of two rows
        AVec af_[3];                                                             // operand ring of three k-steps, carried from chunk to chunk
        BVec bf_[3];
#pragma unroll
        for (int k0 = 0; k0 < 2; ++k0) {
            af_[k0] = act_load_a<NA>(Wf + (2 * k0 + lh) * 128, li);
            bf_[k0] = *reinterpret_cast<const BVec *>(Hc + TM * li + (2 * k0 + lh) * BM);
        }
        __builtin_amdgcn_sched_barrier(0);
        FREE_CHUNK(0, kKC / 2);  PSTAMP(3); FREE_CHUNK(1, kKC / 2);  FREE_CHUNK(2, kKC / 2);  FREE_CHUNK(3, kKC / 2);
        FREE_CHUNK(4, kKC / 2);  FREE_CHUNK(5, kKC / 2);  FREE_CHUNK(6, kKC / 2);  FREE_CHUNK(7, kKC / 2);
        if constexpr (HR < 256) {
            // every wave has read the first half of h1 to the end (its last requests were waited for): lay down rows 128..255 over it
            // and restart the operand ring at k-step 64.  The W2 pieces in flight keep flying.
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            [[maybe_unused]] f32x4 ra, rb;
            if constexpr (PRE2) {
                const f32x4 *rp = reinterpret_cast<const f32x4 *>(tp_tables + tp_row * SHEMS_NCOL);
                ra = rp[0]; rb = rp[1];                                                // row idx + 1
                tp.h_cur = tp_tables[(tp_row - 1) * SHEMS_NCOL];                       // h_countdown of row idx
            }
            L1_PHASE(HR / 32);
            if constexpr (PRE2) tp.nz = noise_draw(A.p, env0 + (tid & (BM - 1)));
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if constexpr (PRE2) {                                                      // w1 / xT are dead from here on
                tp.nx = Row{ra[0], ra[1], ra[2], ra[3], rb[0], rb[1], rb[2], rb[3]};
                tailpre_store(w1 + (tid & (BM - 1)) * kPreDw, tp);
            }
#pragma unroll
            for (int k0 = 0; k0 < 2; ++k0) {
                constexpr int Kp = (kChunks / 2) * (kKC / 2);
                af_[(Kp + k0) % 3] = act_load_a<NA>(Wf + ((kChunks / 2) % RD) * kFreeChunkFloats + (2 * k0 + lh) * 128, li);
                bf_[(Kp + k0) % 3] = *reinterpret_cast<const BVec *>(Hc + TM * li + (2 * k0 + lh) * BM);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        FREE_CHUNK(8, kKC / 2);  FREE_CHUNK(9, kKC / 2);  FREE_CHUNK(10, kKC / 2); FREE_CHUNK(11, kKC / 2);
        FREE_CHUNK(12, kKC / 2); FREE_CHUNK(13, kKC / 2); FREE_CHUNK(14, kKC / 2);
        PSTAMP(4);
        FREE_CHUNK(15, (kH1 - (kChunks - 1) * kKC) / 2);                       // rows 240..249 only = 5 k-steps
    }
    PSTAMP(10);

    // ---- epilogue: relu(acc + b2), layer 3 in the canonical order: one FMA chain per tile, lane halves, tile pairs -> group sums ----
    // (scalar FMAs: v_pk_fma_f32 on splat operands measured slower, 5 524 -> 5 856 cycles at TM = 4)
    const float *w3s = tl + kH2P;
    // The accumulators live in AGPRs; the sched_barrier every four rows keeps the compiler from hoisting all 64*TM*4
    // v_accvgpr_reads to the top (VGPR pressure).  The reads are the compiler's own, so it pads the MFMA -> read hazard itself
    // (hand-written asm reads also made it shuffle accumulators between AGPRs to satisfy the operand constraints).
#pragma unroll
    for (int gl = 0; gl < NA / 2; ++gl) {                        // this wave's column groups
        float u0[TM], u1[TM];                                    // a lane's two tile chains, added
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int a = 2 * gl + t;
            float o0[TM], o1[TM];
#pragma unroll
            for (int b = 0; b < TM; ++b) { o0[b] = 0.0f; o1[b] = 0.0f; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if ((r & (TM == 4 ? 3 : 15)) == 0) __builtin_amdgcn_sched_barrier(0);   // TM = 4: keep the W3 LDS reads near their rows (VGPR pressure); small tiles: a whole tile's reads in flight
                const int n = act_col<NA>(nbase, a, (r & 3) + 8 * (r >> 2) + 4 * lh);   // C/D row i of v_mfma_f32_32x32x* -> column n
                const float2 w3 = *reinterpret_cast<const float2 *>(w3s + 2 * n);       // rows >= 500: zero weights (and finite h)
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const float h = fmaxf(acc[a][b][r], 0.0f);
                    o0[b] = fmaf(h, w3.x, o0[b]);
                    o1[b] = fmaf(h, w3.y, o1[b]);
                }
            }
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                u0[b] = t == 0 ? o0[b] : u0[b] + o0[b];
                u1[b] = t == 0 ? o1[b] : u1[b] + o1[b];
            }
        }
        const int g = nbase / 64 + gl;
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            const float g0 = u0[b] + __shfl_xor(u0[b], 32, 64), g1 = u1[b] + __shfl_xor(u1[b], 32, 64);   // + the other lane half
            if (lh == 0) *reinterpret_cast<float2 *>(RED_OF(g) + (TM * li + b) * 2) = make_float2(g0, g1);
        }
    }
    __syncthreads();

    PSTAMP(11);
    // ---- one thread per env: tanh, noise, clamp, scale_action, step!, remember ---------------------
    double reward = 0.0;
    const int64_t i = env0 + tid;
    if (tid < BM && i < A.m) {
        float hs[2][2];                                                          // canonical: the two column halves, groups in order
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                hs[hf][j] = ((RED_OF(4 * hf)[tid * 2 + j] + RED_OF(4 * hf + 1)[tid * 2 + j]) + RED_OF(4 * hf + 2)[tid * 2 + j]) + RED_OF(4 * hf + 3)[tid * 2 + j];
        const float p0 = tl[kH2P + kH2P * kOut + 0] + (hs[0][0] + hs[1][0]), p1 = tl[kH2P + kH2P * kOut + 1] + (hs[0][1] + hs[1][1]);   // b3 + (H0 + H1)
        reward = act_env_tail(A, i, p0, p1, learner, goff, A.obs == A.v.obs ? xR + tid * kIn : nullptr, PRE ? xP + tid * kPreDw : PRE2 ? w1 + tid * kPreDw : nullptr);
    }
    PSTAMP(12);
#ifndef SHEMS_STAMP_ACT
    if (A.block_reward) {
        __syncthreads();
        double *red64 = reinterpret_cast<double *>(Wc);     // Wc is dead by now
        const double s = block_sum(reward, red64, NW);
        if (tid == 0) A.block_reward[bid] = s;
    }
#endif
}

// =====================================================================================================================
// Column-group forms for small batches: k_actg<TM, NW, NS, RD>.
//
// A launch of <= 8 192 envs is one 32-env tile per CU or less, and k_act's 4-wave tile then pays its fixed phases (stage 0, layer 1,
// layer 3, env tail) and the issue cost of its own LDS-DMA pieces in full.  Here the unit of work is the canonical COLUMN GROUP
// (64 hidden-2 columns = two MFMA tiles, see act_col): a wave owns exactly one group, streams only that group's 256-byte row
// segments of W2 through a private ring (free-running: it waits for nobody but itself), and
//   NS = 1, NW = 8: one workgroup of 8 waves per env tile -- two waves per SIMD, so one wave's LDS-DMA issue slots, operand waits
//                   and epilogue run under the other's MFMAs (4 096 < envs <= 8 192; above that k_act2's 64-env tiles take over);
//   NS = 2, NW = 4: TWO workgroups per env tile, each with four of the eight groups (envs <= 4 096: every CU gets work down to
//                   128 tiles; a 1-env tracking pass runs on two CUs).  Each half leaves its four group sums [4][BM][2] in a scratch
//                   slab and takes a ticket (agent-scope acq_rel atomic); the half that arrives second reads the other's sums and
//                   finishes b3 + g0 + .. + g7 in the canonical order, tanh / noise / step! / remember -- the same bits whichever
//                   half that is, and the bits of every other form.  Nobody waits: the first half simply exits.  Placing both halves
//                   on one XCD (workgroup ids b and b + 8) is for speed only; nothing depends on it.
// Both halves run stage 0 and layer 1 for their env tile (layer 1 is 2 % of the FLOPs).
constexpr int kGChunkFloats = kKC * 64;          // one wave's chunk: 16 rows x 64 columns = 4 KiB = 4 LDS-DMA pieces of 4 rows each
constexpr int kGPieces = 4, kGLastPieces = 3;    // chunk 15: rows 240..251 (rows 250, 251 lie in b2 | W3: in bounds, never multiplied)
constexpr int g_pieces(int ch) { return ch > kChunks - 1 ? 0 : ch == kChunks - 1 ? kGLastPieces : kGPieces; }
template <int RD>
constexpr int g_keep(int c)                      // pieces younger than chunk c + 1's that may stay in flight when the wave waits for it
{
    int n = 0;
    for (int j = c + 2; j <= c + RD - 1; ++j) n += g_pieces(j);
    return n;
}

struct ActSplit {                                // NS = 2 only
    unsigned long long *slot;                    // [tiles][BM] one 8-byte exchange slot per env, kSplitEmpty between launches
};
constexpr unsigned long long kSplitEmpty = ~0ull;         // no pair of layer-3 sums has these bits (made sure of below)

template <int TM, int NW, int RD>
constexpr size_t actg_lds_bytes()
{
    return sizeof(float) * (NW * RD * kGChunkFloats + 16 + 256 * 32 * TM + kW1K * 32 * TM + (kTailFloats + 2) +
                            8 * 32 * TM * kOut + 32 * TM * kIn + 32 * TM * kPreDw + 4);
}

// Piece q (0..3) of a group chunk by asm LDS-DMA: the immediate (q - 2) KiB is added to the global AND the LDS address.
__device__ __forceinline__ void g_piece(const char *sbase, uint32_t voff, uint32_t lds_base, int q)
{
    switch (q) {
    case 0: glds16_asm<-2048>(sbase, voff, lds_base); break;
    case 1: glds16_asm<-1024>(sbase, voff, lds_base); break;
    case 2: glds16_asm<0>(sbase, voff, lds_base); break;
    default: glds16_asm<1024>(sbase, voff, lds_base); break;
    }
}

template <int TM, int NW, int NS, int RD>
__global__ __launch_bounds__(64 * NW) void k_actg(ActArgs A, ActSplit X)
{
    static_assert(NW * NS == 8 && (NW == 4 || NW == 8), "8 column groups per env tile: 8 waves, or two workgroups of 4");
#ifdef SHEMS_STAMP_ACT
#define GSTAMP(i, cond) do { if (threadIdx.x == 0 && (cond)) { reinterpret_cast<unsigned long long *>(A.block_reward)[2*(i)] = __builtin_amdgcn_s_memtime(); reinterpret_cast<unsigned long long *>(A.block_reward)[2*(i)+1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define GSTAMP(i, cond)
#endif
    GSTAMP(0, blockIdx.x == 0);
    static_assert(RD >= 2 && RD <= 4 && TM >= 1 && TM <= 2, "ring of 2..4 chunks; relu(layer 1) fully resident (TM <= 2)");
    constexpr int NT_ = 64 * NW, BM = 32 * TM, HR = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Wc = reinterpret_cast<float *>(smem);             // [NW][RD][16][64] private W2 rings
    float *Hc = Wc + NW * RD * kGChunkFloats + 16;           // [256][BM] relu(layer 1)
    float *xT = Hc + HR * BM;                                // [10][BM]  normalised obs (rows 0..8), row 9 = 1 (bias)
    float *tl = xT + kW1K * BM;                              // b2 [512], W3 [512][2], b3 [2]   (no layer-1 operand image: see wa_ below)
    float *red = tl + (kTailFloats + 2);                     // [8 groups][BM][2] layer-3 group sums
    float *xR = red + 8 * BM * kOut;                         // [BM][9] raw observations
    float *xP = xR + BM * kIn;                               // [BM][kPreDw] TailPre blocks

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    // workgroup -> (env tile, column half).  NS = 2: ids b and b + 8 (same XCD under the round-robin dispatch) are the two halves of
    // tile 8 (b / 16) + b % 8; the ragged end of the grid pairs neighbours.
    int64_t tile = blockIdx.x;
    int half = 0;
    if constexpr (NS == 2) {
        const unsigned b = blockIdx.x, T = gridDim.x >> 1, full = (T >> 3) << 4;
        if (b < full) { tile = (int64_t)(b >> 4) * 8 + (b & 7); half = (b >> 3) & 1; }
        else { tile = (int64_t)(T >> 3) * 8 + ((b - full) >> 1); half = (b - full) & 1; }
    }
    const int64_t env0 = A.m0 + tile * BM;
    const int64_t learner = A.gcount > 1 ? env0 / A.genvs : 0;
    const int64_t goff = learner * A.gstride;
    const float *__restrict__ P = gsh(A.p.actor, goff);
    const float *__restrict__ s_min = gsh(A.p.s_min, goff), *__restrict__ s_max = gsh(A.p.s_max, goff);
    const int g = half * NW + wave;                          // this wave's column group
    const int nbase = 64 * g;

    // ---- stage 0: every global load issued before the first use, addresses clamped, never predicated ------------------------------
    const char *W2g = reinterpret_cast<const char *>(P + kOffW2);
    float *Wg = Wc + wave * (RD * kGChunkFloats);
    const uint32_t ring_lds = lds_addr(Wg) + 2 * 1024;       // piece 2 of ring buffer 0
    // piece q of a chunk = its rows 4 q .. 4 q + 3, 16 lanes (256 B) per row; LDS image linear.  One wave-uniform base per chunk, four
    // constant lane offsets, the instruction's immediate (added to the global AND the LDS address) selects the LDS piece.
    const char *gsbase = W2g + g * 256;
    uint32_t gvoff[kGPieces];
#pragma unroll
    for (int q = 0; q < kGPieces; ++q) gvoff[q] = (uint32_t)((4 * q + (lane >> 4)) * (kH2 * 4) + (lane & 15) * 16 - (q - 2) * 1024);
#define G_PIECE(chunk, q)                                                                         \
    g_piece(gsbase + (size_t)(chunk) * (kKC * kH2 * 4), gvoff[q], ring_lds + ((chunk) % RD) * (kGChunkFloats * 4), (q))
    // The env tail's inputs lead the burst (TailPre; every thread for env tid % BM): the table row idx + 1 hangs off idx, so idx is
    // asked for first and the row right behind it -- the row then lands during stage 0 instead of holding up the layer-1 phase.
    // One config for the whole batch (the usual case): its table_row0 / nrow come by scalar loads, not behind a per-env config index.
    TailPre tp;
    const bool view = A.do_step != 0, multi = view && A.v.n_cfg > 1;
    {
        const int64_t pe = min(env0 + (tid & (BM - 1)), A.m - 1);
        const int32_t *pidx = view ? A.v.idx : reinterpret_cast<const int32_t *>(P), *pstep = view ? A.v.step : reinterpret_cast<const int32_t *>(P);
        const uint16_t *pci = multi ? A.v.cfg_of_env : reinterpret_cast<const uint16_t *>(P);
        tp.idx = pidx[view ? pe : 0];
        tp.ci = multi ? (int)pci[pe] : 0;
        tp.step = pstep[view ? pe : 0];
    }
    constexpr int kIt = (BM * kIn + NT_ - 1) / NT_;
    float sv[kIt], lo[kIt], hi[kIt], tv[6];
    const int64_t last = A.m * kIn - 1;
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int e = min(it * NT_ + tid, BM * kIn - 1), k = e % kIn;
        sv[it] = A.obs[min(env0 * kIn + e, last)];
        lo[it] = s_min[k];
        hi[it] = s_max[k];
    }
    // Layer-1 A operands straight from the parameter block (round 4; k_act2 does the same): a wave lays down row groups wave (and
    // wave + 4 when it has two), and lane (li, lh) of a tile of group g multiplies W1[2 s + lh][32 g + li], s = 0..4 (input row 9 = b1)
    // -- ten words per lane, requested here with everything else.  No [10][256] image in LDS: 10 KB less per workgroup, which is what
    // lets a 64-KB workgroup of the update share the CU with the two-workgroups-per-tile form (92 + 64 KB <= 160 KB).
    constexpr int kL1PerWave = (8 * TM) / NW;                // 2 (NW = 4) or 1 (NW = 8)
    static_assert(TM == 1 && (kL1PerWave == 1 || kL1PerWave == 2), "k_actg lays layer 1 down for 32-env tiles");
    float wa_[kL1PerWave][kW1K / 2];
    bool wok_[kL1PerWave];
#pragma unroll
    for (int u = 0; u < kL1PerWave; ++u) {
        const int k = 32 * (wave + NW * u) + li;
        wok_[u] = k < kH1;
#pragma unroll
        for (int s2 = 0; s2 < kW1K / 2; ++s2) {
            const int j = 2 * s2 + lh;                        // 0..9; row 9 of the operand = b1 = parameter row kIn
            wa_[u][s2] = P[j * kH1 + min(k, kH1 - 1)];
        }
    }
#pragma unroll
    for (int it = 0; it < 6; ++it) tv[it] = P[kOffB2 + min(it * 256 + (tid & 255), kH2 + kH2 * kOut + kOut - 1)];
    tp.nz = noise_draw(A.p, env0 + (tid & (BM - 1)));       // Philox + Box-Muller under the loads
    f32x4 ra, rb;
    {
        const int32_t *pc = view ? reinterpret_cast<const int32_t *>(A.v.cfgs + tp.ci) : reinterpret_cast<const int32_t *>(P);
        constexpr int o_row0 = offsetof(shems_config, table_row0) / 4, o_nrow = offsetof(shems_config, nrow) / 4;
        const int32_t row0 = pc[o_row0], nrow = pc[o_nrow];
        const float *tp_tables = view ? A.v.tables : P;
        // row idx + 1 (1-based) of the env's table, clamped into the table: env_advance_rows rejects idx + 1 > nrow before it looks at it
        const int64_t tp_row = view ? (int64_t)row0 + max(min(tp.idx + 1, nrow), 2) - 1 : 1;
        const f32x4 *rp = reinterpret_cast<const f32x4 *>(tp_tables + tp_row * SHEMS_NCOL);
        ra = rp[0]; rb = rp[1];                                                // row idx + 1
        tp.h_cur = tp_tables[(tp_row - 1) * SHEMS_NCOL];                       // h_countdown of row idx
    }
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int e = it * NT_ + tid, m = e / kIn, k = e - m * kIn;
        const float x = (sv[it] - lo[it]) / ((hi[it] - lo[it]) + 1e-8f);      // MPS:56
        if (e < BM * kIn) { xT[k * BM + m] = env0 * kIn + e <= last ? x : 0.0f; xR[e] = sv[it]; }
    }
    for (int e = tid; e < BM; e += NT_) xT[kIn * BM + e] = 1.0f;                               // row 9 = 1: the bias input
    {
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int e = tid < 256 ? it * 256 + tid : 1 << 20;   // source index: [0,500) b2, [500,1500) W3, [1500,1502) b3
            if (e < kH2) tl[e] = tv[it];
            else if (e < kH2 + kH2 * kOut) tl[kH2P + (e - kH2)] = tv[it];
            else if (e < kH2 + kH2 * kOut + kOut) tl[kH2P + kH2P * kOut + (e - kH2 - kH2 * kOut)] = tv[it];
        }
        if (tid < kH2P - kH2) tl[kH2 + tid] = 0.0f;                                   // pad rows of b2
        if (tid < (kH2P - kH2) * kOut) tl[kH2P + kH2 * kOut + tid] = 0.0f;            // pad rows of W3
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");            // xT visible (LDS stores only; the row loads keep flying)
    GSTAMP(1, blockIdx.x == 0);
#undef L1_A
#define L1_A(s_, j_, g, which) (wok_[which] ? wa_[which][s_] : 0.0f)           /* rows of W1 / b1 held in registers since stage 0 */
    {
        // the ring's first chunks go out only now: 256 workgroups x 64 KB of pieces would otherwise queue in front of the few KB
        // every workgroup is waiting for
#pragma unroll
        for (int ch = 0; ch < RD - 1; ++ch)
#pragma unroll
            for (int q = 0; q < kGPieces; ++q) G_PIECE(ch, q);
        // layer 1 on the matrix pipe: 8 row groups x TM env tiles over the NW waves (pairs interleaved where a wave has two)
        constexpr int kL1Tiles = 8 * TM;
        if constexpr (kL1Tiles / NW >= 2) {
#pragma unroll
            for (int u = 0; u < kL1Tiles / (2 * NW); ++u) {
                const int t0 = wave + 2 * NW * u, g0 = t0 / TM, b0 = t0 % TM, t1 = t0 + NW, g1 = t1 / TM, b1 = t1 % TM;
                L1_TILE2(g0, b0, Hc + (g0 * 32) * BM + TM * li + b0, g1, b1, Hc + (g1 * 32) * BM + TM * li + b1);
            }
        } else {
            const int g0 = wave / TM, b0 = wave % TM;
            L1_TILE(g0, b0, Hc + (g0 * 32) * BM + TM * li + b0);
        }
        // the table row has had stage 0's tail and the layer-1 phase to arrive (the compiler's wait for it also covers the ring's
        // first pieces, which it cannot see: they were issued later and the counter retires in order)
        tp.nx = Row{ra[0], ra[1], ra[2], ra[3], rb[0], rb[1], rb[2], rb[3]};
        tailpre_store(xP + (tid & (BM - 1)) * kPreDw, tp);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");            // h1 complete; every wave waits for its own ring below
    GSTAMP(2, blockIdx.x == 0);

    // ---- layer 2: this wave's group = 2 x TM tiles over 125 k-steps, no barrier ---------------------------------------------------
    typedef typename FVec<TM>::type BVec;
    f32x16 acc[2][TM];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bias = tl[act_col<2>(nbase, a, (r & 3) + 8 * (r >> 2) + 4 * lh)];
#pragma unroll
            for (int b = 0; b < TM; ++b) acc[a][b][r] = bias;
        }
    constexpr int kGKsteps = (kChunks - 1) * (kKC / 2) + (kH1 - (kChunks - 1) * kKC) / 2;      // 125 k-steps of two rows
    f32x2 af_[3];
    BVec bf_[3];
    /* k-step K = 8 c + ks: operands in register buffer K % 3, requested two k-steps ago.  One scheduling region per k-step:        */
    /* [TM MFMAs] [A read of K + 2] [TM MFMAs] [B read of K + 2]; one LDS-DMA piece (asm) of chunk c + RD - 1 closes k-steps 0..3.   */
#define G_KSTEP(c, ks)                                                                                          \
    do {                                                                                                        \
        const int K_ = 8 * (c) + (ks), K2_ = K_ + 2;                                                            \
        if (K2_ < kGKsteps) {                                                                                   \
            const int c2_ = K2_ >> 3, kr_ = 2 * (K2_ & 7) + lh;                                                 \
            af_[K2_ % 3] = *reinterpret_cast<const f32x2 *>(Wg + (c2_ % RD) * kGChunkFloats + kr_ * 64 + 2 * li); \
            bf_[K2_ % 3] = *reinterpret_cast<const BVec *>(Hc + (c2_ * kKC + kr_) * BM + TM * li);              \
        }                                                                                                       \
        _Pragma("unroll") for (int b = 0; b < TM; ++b)                                                          \
            acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af_[K_ % 3][0], fvec_get<TM>(bf_[K_ % 3], b), acc[0][b], 0, 0, 0); \
        _Pragma("unroll") for (int b = 0; b < TM; ++b)                                                          \
            acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af_[K_ % 3][1], fvec_get<TM>(bf_[K_ % 3], b), acc[1][b], 0, 0, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);                                                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);                                                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        if ((ks) < g_pieces((c) + RD - 1)) { G_PIECE((c) + RD - 1, (ks) < kGPieces ? (ks) : 0); __builtin_amdgcn_sched_barrier(0); } \
    } while (0)
    /* Two waves per SIMD (NW = 8; waves w and w + 4 share one): the issue arbiter serves the older wave first -- waves 0..3 finish the  */
    /* layer after 19.7 k cycles, waves 4..7 after 35.3 k (32 k of matrix work per SIMD).  Swapping s_setprio between the two every    */
    /* chunk was measured: the early waves slow down (24.5 k), the late ones end at the same 35.4 k -- the 10 % are not arbitration but  */
    /* the SIMD's 8 LDS-DMA pieces per 2048 MFMA cycles (the split form, half the pieces per CU, runs its layer at 94 %).               */
#define G_CHUNK(c, NKS)                                                                                         \
    do {                                                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < ((NKS) < 6 ? (NKS) : 6); ++ks) G_KSTEP(c, ks);                  \
        if ((c) + 1 < kChunks) {                                                                                \
            constexpr int keep_ = g_keep<RD>(c);                                                                \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(keep_) : "memory");      /* this wave's chunk c + 1 has landed */ \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
        }                                                                                                       \
        _Pragma("unroll") for (int ks = 6; ks < (NKS); ++ks) G_KSTEP(c, ks);                                    \
    } while (0)
    {
        constexpr int keep0_ = g_keep<RD>(-1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(keep0_) : "memory");          // this wave's chunk 0 has landed
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef SHEMS_STAMP_ACT
    if (lane == 0 && blockIdx.x == 0) reinterpret_cast<unsigned long long *>(A.block_reward)[48 + wave] = __builtin_amdgcn_s_memtime();    // every wave's loop start
#endif
#pragma unroll
    for (int k0 = 0; k0 < 2; ++k0) {
        af_[k0] = *reinterpret_cast<const f32x2 *>(Wg + (2 * k0 + lh) * 64 + 2 * li);
        bf_[k0] = *reinterpret_cast<const BVec *>(Hc + (2 * k0 + lh) * BM + TM * li);
    }
    __builtin_amdgcn_sched_barrier(0);
    G_CHUNK(0, kKC / 2);  GSTAMP(3, blockIdx.x == 0); G_CHUNK(1, kKC / 2);  G_CHUNK(2, kKC / 2);  G_CHUNK(3, kKC / 2);
    G_CHUNK(4, kKC / 2);  G_CHUNK(5, kKC / 2);  G_CHUNK(6, kKC / 2);  G_CHUNK(7, kKC / 2);
    G_CHUNK(8, kKC / 2);  G_CHUNK(9, kKC / 2);  G_CHUNK(10, kKC / 2); G_CHUNK(11, kKC / 2);
    G_CHUNK(12, kKC / 2); G_CHUNK(13, kKC / 2); G_CHUNK(14, kKC / 2);
    GSTAMP(4, blockIdx.x == 0);
    G_CHUNK(15, (kH1 - (kChunks - 1) * kKC) / 2);                              // rows 240..249 only = 5 k-steps

    GSTAMP(10, blockIdx.x == 0);
#ifdef SHEMS_STAMP_ACT
    if (lane == 0 && blockIdx.x == 0) reinterpret_cast<unsigned long long *>(A.block_reward)[32 + wave] = __builtin_amdgcn_s_memtime();    // every wave's loop end
#endif
    // ---- layer 3 of this group, canonical order (see act_col) ---------------------------------------------------------------------
    {
        const float *w3s = tl + kH2P;
        float u0[TM], u1[TM];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float o0[TM], o1[TM];
#pragma unroll
            for (int b = 0; b < TM; ++b) { o0[b] = 0.0f; o1[b] = 0.0f; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = act_col<2>(nbase, t, (r & 3) + 8 * (r >> 2) + 4 * lh);
                const float2 w3 = *reinterpret_cast<const float2 *>(w3s + 2 * n);
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const float h = fmaxf(acc[t][b][r], 0.0f);
                    o0[b] = fmaf(h, w3.x, o0[b]);
                    o1[b] = fmaf(h, w3.y, o1[b]);
                }
            }
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                u0[b] = t == 0 ? o0[b] : u0[b] + o0[b];
                u1[b] = t == 0 ? o1[b] : u1[b] + o1[b];
            }
        }
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            const float g0 = u0[b] + __shfl_xor(u0[b], 32, 64), g1 = u1[b] + __shfl_xor(u1[b], 32, 64);
            if (lh == 0) *reinterpret_cast<float2 *>(red + (g * BM + TM * li + b) * 2) = make_float2(g0, g1);
        }
    }
    __syncthreads();
    GSTAMP(8, blockIdx.x == 0);

    // ---- one thread per env: b3 + (H0 + H1), tanh, noise, clamp, scale_action, step!, remember -----------------------------------
    double reward = 0.0;
    const int64_t i = env0 + tid;
    bool fin = tid < BM && i < A.m;
    float p0 = 0.0f, p1 = 0.0f;
    if (fin) {
        // sums of a column half, groups in canonical order (NS = 2: this workgroup holds only its own half's groups)
        auto half_sum = [&](int hf, int j) {
            return ((red[((4 * hf) * BM + tid) * 2 + j] + red[((4 * hf + 1) * BM + tid) * 2 + j]) + red[((4 * hf + 2) * BM + tid) * 2 + j]) +
                   red[((4 * hf + 3) * BM + tid) * 2 + j];
        };
        float h00, h01, h10, h11;                                                  // H0 (outputs 0, 1), H1 (outputs 0, 1)
        if constexpr (NS == 1) {
            h00 = half_sum(0, 0); h01 = half_sum(0, 1); h10 = half_sum(1, 0); h11 = half_sum(1, 1);
        } else {
            // The data is the hand-off: ONE 8-byte exchange per env on its slot (agent scope, performed at the memory side, so the two
            // halves meet there whatever XCDs they run on).  Whoever finds the slot empty leaves its sums and is done with this env;
            // whoever finds the other half's sums finishes the env and empties the slot for the next launch.  Nothing is waited for.
            float m0 = half_sum(half, 0), m1 = half_sum(half, 1);
            unsigned long long mine = ((unsigned long long)__float_as_uint(m1) << 32) | __float_as_uint(m0);
            if (mine == kSplitEmpty) { mine = 0x7FC000007FC00000ull; m0 = m1 = __uint_as_float(0x7FC00000u); }   // a NaN pair never takes the empty pattern
            unsigned long long *slot = X.slot + tile * BM + tid;
            const unsigned long long got = __hip_atomic_exchange(slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got == kSplitEmpty) fin = false;
            else __hip_atomic_store(slot, kSplitEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float o0 = __uint_as_float((uint32_t)got), o1 = __uint_as_float((uint32_t)(got >> 32));
            h00 = half == 0 ? m0 : o0; h01 = half == 0 ? m1 : o1;
            h10 = half == 0 ? o0 : m0; h11 = half == 0 ? o1 : m1;
        }
        p0 = tl[kH2P + kH2P * kOut + 0] + (h00 + h10);                              // b3 + (H0 + H1)
        p1 = tl[kH2P + kH2P * kOut + 1] + (h01 + h11);
    }
    GSTAMP(9, blockIdx.x == 0);
    GSTAMP(11, tile == 0 && fin);
    if (fin) reward = act_env_tail(A, i, p0, p1, learner, goff, A.obs == A.v.obs ? xR + tid * kIn : nullptr, xP + tid * kPreDw);
    GSTAMP(12, tile == 0 && fin);
#ifndef SHEMS_STAMP_ACT
    if (NS == 1 && A.block_reward) {
        __syncthreads();
        double *red64 = reinterpret_cast<double *>(Wc);     // the rings are dead by now
        const double s = block_sum(reward, red64, NW);
        if (tid == 0) A.block_reward[tile] = s;
    }
#endif
}

// =====================================================================================================================
// k_act2: the fused step for large batches with TWO workgroups resident per CU.
//
// k_act's 128-env tile fills a CU's LDS (161 KB), so a CU holds one workgroup at a time and everything outside the MFMA loop -- stage
// 0, the two layer-1 phases, layer 3, the env tail: 12.5 % of a workgroup's cycles at 65 536 envs, most of it exposed latency -- leaves
// the matrix pipe idle (the loop itself runs at 99.5 %).  Here a workgroup is a 64-env tile that needs < 80 KB of LDS and <= 256
// registers per lane, so TWO of them share a CU: while one is in its latency-bound phases the other's waves own the matrix pipe.
// What makes it fit:
//   * W2 ring of 4 chunks x 4 rows per wave (8 KB instead of 32): the same 3-chunk prefetch distance in k-steps, a quarter of the bytes;
//   * relu(layer 1) resident one half at a time (32 KB), as k_act's 128-env tile does;
//   * no layer-1 operand image in LDS: the 10 weights a lane needs per layer-1 tile come straight from the parameter block (L2) --
//     their latency is what the other workgroup's MFMAs are for;
//   * the env tail reads its observations from global memory, not from an LDS copy.
// Same canonical column order, same arithmetic: the bytes of every other form.
constexpr int k2CR = 4, k2RD = 4, k2CF = k2CR * 128;                       // chunk rows, ring depth, floats per wave chunk
constexpr int k2NCH = (kH1 + k2CR - 1) / k2CR;                             // 63 chunks hold rows 0 .. 251 (250, 251: in bounds, never multiplied)
constexpr int k2_pieces(int ch) { return ch < 0 || ch >= k2NCH ? 0 : ch == k2NCH - 1 ? (kH1 - (k2NCH - 1) * k2CR + 1) / 2 : k2CR / 2; }
constexpr size_t act2_lds_bytes()
{
    return sizeof(float) * (4 * k2RD * k2CF + 16 + 128 * 64 + kW1K * 64 + (kTailFloats + 2) + 64 * kPreDw + 4);
}

__device__ __forceinline__ void k2_piece(const char *sbase, uint32_t voff, uint32_t lds_base, int q)
{
    if (q == 0) glds16_asm<-1024>(sbase, voff, lds_base); else glds16_asm<0>(sbase, voff, lds_base);
}

__global__ __launch_bounds__(256, 2) void k_act2(ActArgs A)
{
    constexpr int TM = 2, BM = 64, NA = 4, NT_ = 256, HR = 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Wc = reinterpret_cast<float *>(smem);             // [4 waves][RD][4 rows][128] private W2 rings
    float *Hc = Wc + 4 * k2RD * k2CF + 16;                   // [128][BM] relu(layer 1), one half at a time
    float *xT = Hc + HR * BM;                                // [10][BM]
    float *tl = xT + kW1K * BM;                              // b2 [512], W3 [512][2], b3 [2]
    float *xP = tl + (kTailFloats + 2);                      // [BM][kPreDw] TailPre blocks
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    int64_t bid = blockIdx.x;
    if (A.gcount > 1 && (gridDim.x & 7) == 0) bid = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);     // a learner's tiles share an XCD
    const int64_t env0 = A.m0 + bid * BM;
    const int64_t learner = A.gcount > 1 ? env0 / A.genvs : 0;
    const int64_t goff = learner * A.gstride;
    const float *__restrict__ P = gsh(A.p.actor, goff);
    const float *__restrict__ s_min = gsh(A.p.s_min, goff), *__restrict__ s_max = gsh(A.p.s_max, goff);
    const int nbase = 128 * wave;                            // this wave's two column groups

    // ---- stage 0 --------------------------------------------------------------------------------------------------------------
    const char *W2g = reinterpret_cast<const char *>(P + kOffW2);
    float *Wf = Wc + wave * (k2RD * k2CF);
    const uint32_t ring_lds = lds_addr(Wf) + 1024;           // piece 1 of ring buffer 0
    const char *wsbase = W2g + wave * 512;
    uint32_t wvoff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) wvoff[q] = (uint32_t)((lane >> 5) * (kH2 * 4) + (lane & 31) * 16 + q * (2 * kH2 * 4) - (q - 1) * 1024);
#define K2_PIECE(chunk, q) k2_piece(wsbase + (size_t)(chunk) * (k2CR * kH2 * 4), wvoff[q], ring_lds + ((chunk) % k2RD) * (k2CF * 4), (q))
    TailPre tp;
    const bool view = A.do_step != 0, multi = view && A.v.n_cfg > 1;
    {
        const int64_t pe = min(env0 + (tid & (BM - 1)), A.m - 1);
        const int32_t *pidx = view ? A.v.idx : reinterpret_cast<const int32_t *>(P), *pstep = view ? A.v.step : reinterpret_cast<const int32_t *>(P);
        const uint16_t *pci = multi ? A.v.cfg_of_env : reinterpret_cast<const uint16_t *>(P);
        tp.idx = pidx[view ? pe : 0];
        tp.ci = multi ? (int)pci[pe] : 0;
        tp.step = pstep[view ? pe : 0];
    }
    constexpr int kIt = (BM * kIn + NT_ - 1) / NT_;          // 3
    float sv[kIt], lo[kIt], hi[kIt], tv[6];
    const int64_t last = A.m * kIn - 1;
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int e = min(it * NT_ + tid, BM * kIn - 1), k = e % kIn;
        sv[it] = A.obs[min(env0 * kIn + e, last)];
        lo[it] = s_min[k];
        hi[it] = s_max[k];
    }
#pragma unroll
    for (int it = 0; it < 6; ++it) tv[it] = P[kOffB2 + min(it * 256 + tid, kH2 + kH2 * kOut + kOut - 1)];
    // layer-1 weights of this wave's two tiles of the FIRST half, straight from the parameter block: tile t = wave + 4 u (u = 0, 1) of the
    // 8 (row group, env tile) pairs; W1[j][k] and b1[k] are contiguous (row 9 of the "image" is b1), columns >= 250 are zero
#define K2_L1_LOAD(gbase, a0_, a1_, kA_, kB_)                                                     \
    _Pragma("unroll") for (int s_ = 0; s_ < kW1K / 2; ++s_) {                                     \
        const int j_ = 2 * s_ + lh;                                                               \
        a0_[s_] = P[j_ * kH1 + min(kA_, kH1 - 1)];                                                \
        a1_[s_] = P[j_ * kH1 + min(kB_, kH1 - 1)];                                                \
    }
    const int t0 = wave, t1 = wave + 4;                      // tiles of a phase: (g, b) = (t / 2, t % 2)
    const int kA0 = 32 * (t0 >> 1) + li, kB0 = 32 * (t1 >> 1) + li;
    float wa0[kW1K / 2], wb0[kW1K / 2];
    K2_L1_LOAD(0, wa0, wb0, kA0, kB0)
    tp.nz = noise_draw(A.p, env0 + (tid & (BM - 1)));
    f32x4 ra, rb;
    {
        const int32_t *pc = view ? reinterpret_cast<const int32_t *>(A.v.cfgs + tp.ci) : reinterpret_cast<const int32_t *>(P);
        constexpr int o_row0 = offsetof(shems_config, table_row0) / 4, o_nrow = offsetof(shems_config, nrow) / 4;
        const int32_t row0 = pc[o_row0], nrow = pc[o_nrow];
        const float *tp_tables = view ? A.v.tables : P;
        const int64_t tp_row = view ? (int64_t)row0 + max(min(tp.idx + 1, nrow), 2) - 1 : 1;
        const f32x4 *rp = reinterpret_cast<const f32x4 *>(tp_tables + tp_row * SHEMS_NCOL);
        ra = rp[0]; rb = rp[1];
        tp.h_cur = tp_tables[(tp_row - 1) * SHEMS_NCOL];
    }
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int e = it * NT_ + tid, m = e / kIn, k = e - m * kIn;
        const float x = (sv[it] - lo[it]) / ((hi[it] - lo[it]) + 1e-8f);      // MPS:56
        if (e < BM * kIn) xT[k * BM + m] = env0 * kIn + e <= last ? x : 0.0f;
    }
    if (tid < BM) xT[kIn * BM + tid] = 1.0f;                                                   // row 9 = 1: the bias input
    {
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int e = it * 256 + tid;                        // source index: [0,500) b2, [500,1500) W3, [1500,1502) b3
            if (e < kH2) tl[e] = tv[it];
            else if (e < kH2 + kH2 * kOut) tl[kH2P + (e - kH2)] = tv[it];
            else if (e < kH2 + kH2 * kOut + kOut) tl[kH2P + kH2P * kOut + (e - kH2 - kH2 * kOut)] = tv[it];
        }
        if (tid < kH2P - kH2) tl[kH2 + tid] = 0.0f;
        if (tid < (kH2P - kH2) * kOut) tl[kH2P + kH2 * kOut + tid] = 0.0f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");            // xT visible
    // the ring's first three chunks
#pragma unroll
    for (int ch = 0; ch < k2RD - 1; ++ch)
#pragma unroll
        for (int q = 0; q < 2; ++q) K2_PIECE(ch, q);

    // one phase of layer 1: row groups gbase .. gbase + 3 of the half, two tiles per wave interleaved (the chains of L1_TILE2)
#define K2_L1_PHASE(gbase, a0_, a1_)                                                              \
    do {                                                                                          \
        const int gA_ = (gbase) + (t0 >> 1), bA_ = t0 & 1, gB_ = (gbase) + (t1 >> 1), bB_ = t1 & 1; \
        float c0_[kW1K / 2], c1_[kW1K / 2], z0_[kW1K / 2], z1_[kW1K / 2];                         \
        _Pragma("unroll") for (int s_ = 0; s_ < kW1K / 2; ++s_) {                                 \
            const int j_ = 2 * s_ + lh;                                                           \
            c0_[s_] = xT[j_ * BM + TM * li + bA_];                                                \
            c1_[s_] = xT[j_ * BM + TM * li + bB_];                                                \
            z0_[s_] = 32 * gA_ + li < kH1 ? a0_[s_] : 0.0f;                                       \
            z1_[s_] = 32 * gB_ + li < kH1 ? a1_[s_] : 0.0f;                                       \
        }                                                                                         \
        f32x16 u0_, u1_;                                                                          \
        asm volatile("s_nop 1\n\t"                                                                \
                     "v_mfma_f32_32x32x2_f32 %0, %2, %7, 0\n\t"                                   \
                     "v_mfma_f32_32x32x2_f32 %1, %12, %17, 0\n\t"                                 \
                     "v_mfma_f32_32x32x2_f32 %0, %3, %8, %0\n\t"                                  \
                     "v_mfma_f32_32x32x2_f32 %1, %13, %18, %1\n\t"                                \
                     "v_mfma_f32_32x32x2_f32 %0, %4, %9, %0\n\t"                                  \
                     "v_mfma_f32_32x32x2_f32 %1, %14, %19, %1\n\t"                                \
                     "v_mfma_f32_32x32x2_f32 %0, %5, %10, %0\n\t"                                 \
                     "v_mfma_f32_32x32x2_f32 %1, %15, %20, %1\n\t"                                \
                     "v_mfma_f32_32x32x2_f32 %0, %6, %11, %0\n\t"                                 \
                     "v_mfma_f32_32x32x2_f32 %1, %16, %21, %1\n\t"                                \
                     "s_nop 15\n\ts_nop 7"                                                        \
                     : "=&v"(u0_), "=&v"(u1_)                                                     \
                     : "v"(z0_[0]), "v"(z0_[1]), "v"(z0_[2]), "v"(z0_[3]), "v"(z0_[4]),           \
                       "v"(c0_[0]), "v"(c0_[1]), "v"(c0_[2]), "v"(c0_[3]), "v"(c0_[4]),           \
                       "v"(z1_[0]), "v"(z1_[1]), "v"(z1_[2]), "v"(z1_[3]), "v"(z1_[4]),           \
                       "v"(c1_[0]), "v"(c1_[1]), "v"(c1_[2]), "v"(c1_[3]), "v"(c1_[4]));          \
        float *d0_ = Hc + ((gA_ * 32) % HR) * BM + TM * li + bA_, *d1_ = Hc + ((gB_ * 32) % HR) * BM + TM * li + bB_; \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) {                                       \
            d0_[((r_ & 3) + 8 * (r_ >> 2) + 4 * lh) * BM] = fmaxf(u0_[r_], 0.0f);                 \
            d1_[((r_ & 3) + 8 * (r_ >> 2) + 4 * lh) * BM] = fmaxf(u1_[r_], 0.0f);                 \
        }                                                                                         \
    } while (0)
    K2_L1_PHASE(0, wa0, wb0);
    // weights of the second half's two tiles: requested now, consumed after k-step 63
    float wa1[kW1K / 2], wb1[kW1K / 2];
    {
        const int kA1 = 128 + kA0, kB1 = 128 + kB0;
        K2_L1_LOAD(4, wa1, wb1, kA1, kB1)
    }
    tp.nx = Row{ra[0], ra[1], ra[2], ra[3], rb[0], rb[1], rb[2], rb[3]};
    tailpre_store(xP + (tid & (BM - 1)) * kPreDw, tp);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");            // first half of h1 complete

    // ---- layer 2 ----------------------------------------------------------------------------------------------------------------
    typedef FVec<NA>::type AVec;
    typedef FVec<TM>::type BVec;
    f32x16 acc[NA][TM];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bias = tl[act_col<NA>(nbase, a, (r & 3) + 8 * (r >> 2) + 4 * lh)];
#pragma unroll
            for (int b = 0; b < TM; ++b) acc[a][b][r] = bias;
        }
    constexpr int kKsteps = kH1 / 2;                         // 125 k-steps of two rows; chunk c = k-steps 2 c, 2 c + 1
    AVec af_[3];
    BVec bf_[3];
    /* k-step K: operands in register buffer K % 3, requested two k-steps ago.  Before the reads of K + 2 open a new chunk the wave waits  */
    /* for that chunk (its own ring: vmcnt retires in order; the pieces of the chunk after it stay in flight).  The pieces of chunk          */
    /* (K >> 1) + 3 go out one per k-step, after the k-step's MFMAs.  Nothing of h1's second half is requested before it is laid down.       */
#define K2_KSTEP(K)                                                                                             \
    do {                                                                                                        \
        constexpr int K2_ = (K) + 2, c2_ = K2_ >> 1, klim_ = (K) < 64 ? 64 : kKsteps;                           \
        if (K2_ < klim_) {                                                                                      \
            if ((K2_ & 1) == 0) {                                                                               \
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(k2_pieces(c2_ + 1)) : "memory");                       \
                __builtin_amdgcn_sched_barrier(0);                                                              \
            }                                                                                                   \
            const int kr_ = 2 * (K2_ & 1) + lh;                                                                 \
            af_[K2_ % 3] = act_load_a<NA>(Wf + (c2_ % k2RD) * k2CF + kr_ * 128, li);                            \
            bf_[K2_ % 3] = *reinterpret_cast<const BVec *>(Hc + ((2 * K2_) % HR + lh) * BM + TM * li);          \
        }                                                                                                       \
        _Pragma("unroll") for (int a = 0; a < NA; ++a)                                                          \
            _Pragma("unroll") for (int b = 0; b < TM; ++b)                                                      \
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fvec_get<NA>(af_[(K) % 3], a), fvec_get<TM>(bf_[(K) % 3], b), acc[a][b], 0, 0, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                      \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                      \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        if (((K) & 1) < k2_pieces(((K) >> 1) + k2RD - 1)) { K2_PIECE(((K) >> 1) + k2RD - 1, (K) & 1); __builtin_amdgcn_sched_barrier(0); } \
    } while (0)
#define K2_K4(K) K2_KSTEP(K); K2_KSTEP((K) + 1); K2_KSTEP((K) + 2); K2_KSTEP((K) + 3)
#define K2_K16(K) K2_K4(K); K2_K4((K) + 4); K2_K4((K) + 8); K2_K4((K) + 12)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(k2_pieces(1) + k2_pieces(2)) : "memory");      // chunk 0 has landed
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k0 = 0; k0 < 2; ++k0) {
        af_[k0] = act_load_a<NA>(Wf + (2 * k0 + lh) * 128, li);
        bf_[k0] = *reinterpret_cast<const BVec *>(Hc + (2 * k0 + lh) * BM + TM * li);
    }
    __builtin_amdgcn_sched_barrier(0);
    K2_K16(0); K2_K16(16); K2_K16(32); K2_K16(48);
    // every wave has read the first half of h1 to the end: rows 128..255 over it, then the operand ring restarts at k-step 64
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    K2_L1_PHASE(4, wa1, wb1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(k2_pieces(33) + k2_pieces(34)) : "memory");    // chunk 32 has landed (33, 34 may fly)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k0 = 0; k0 < 2; ++k0) {
        af_[(64 + k0) % 3] = act_load_a<NA>(Wf + ((32 + (k0 >> 1)) % k2RD) * k2CF + (2 * (k0 & 1) + lh) * 128, li);
        bf_[(64 + k0) % 3] = *reinterpret_cast<const BVec *>(Hc + (2 * k0 + lh) * BM + TM * li);
    }
    __builtin_amdgcn_sched_barrier(0);
    K2_K16(64); K2_K16(80); K2_K16(96); K2_K4(112); K2_K4(116); K2_K4(120); K2_KSTEP(124);

    // ---- layer 3, canonical order; the group sums go to the start of this wave's own (dead) ring ---------------------------------
    const float *w3s = tl + kH2P;
#pragma unroll
    for (int gl = 0; gl < 2; ++gl) {
        float u0[TM], u1[TM];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int a = 2 * gl + t;
            float o0[TM], o1[TM];
#pragma unroll
            for (int b = 0; b < TM; ++b) { o0[b] = 0.0f; o1[b] = 0.0f; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = act_col<NA>(nbase, a, (r & 3) + 8 * (r >> 2) + 4 * lh);
                const float2 w3 = *reinterpret_cast<const float2 *>(w3s + 2 * n);
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const float h = fmaxf(acc[a][b][r], 0.0f);
                    o0[b] = fmaf(h, w3.x, o0[b]);
                    o1[b] = fmaf(h, w3.y, o1[b]);
                }
            }
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                u0[b] = t == 0 ? o0[b] : u0[b] + o0[b];
                u1[b] = t == 0 ? o1[b] : u1[b] + o1[b];
            }
        }
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            const float g0 = u0[b] + __shfl_xor(u0[b], 32, 64), g1 = u1[b] + __shfl_xor(u1[b], 32, 64);
            if (lh == 0) *reinterpret_cast<float2 *>(Wf + gl * (BM * kOut) + (TM * li + b) * 2) = make_float2(g0, g1);
        }
    }
    __syncthreads();
    double reward = 0.0;
    const int64_t i = env0 + tid;
    if (tid < BM && i < A.m) {
        float hs[2][2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // group g = 4 hf + q lives in wave g / 2's ring, slot g % 2
                const float *r0 = Wc + (2 * hf) * (k2RD * k2CF), *r1 = Wc + (2 * hf + 1) * (k2RD * k2CF);
                hs[hf][j] = ((r0[tid * 2 + j] + r0[BM * kOut + tid * 2 + j]) + r1[tid * 2 + j]) + r1[BM * kOut + tid * 2 + j];
            }
        const float p0 = tl[kH2P + kH2P * kOut + 0] + (hs[0][0] + hs[1][0]), p1 = tl[kH2P + kH2P * kOut + 1] + (hs[0][1] + hs[1][1]);
        reward = act_env_tail(A, i, p0, p1, learner, goff, nullptr, xP + tid * kPreDw);
    }
#ifndef SHEMS_STAMP_ACT
    if (A.block_reward) {
        __syncthreads();
        double *red64 = reinterpret_cast<double *>(Hc);      // h1 is dead by now
        const double s = block_sum(reward, red64, 4);
        if (tid == 0) A.block_reward[bid] = s;
    }
#endif
}

static int launch_act2(const ActArgs &a, hipStream_t st)
{
    constexpr size_t lds = act2_lds_bytes();
    static_assert(lds <= 80 * 1024, "k_act2: two workgroups must fit a CU's 160 KB");
    static std::atomic<uint64_t> optin{0};
    if (int rc = lds_optin(optin, reinterpret_cast<const void *>(&k_act2), (int)lds, "hipFuncSetAttribute(k_act2)")) return rc;
    hipLaunchKernelGGL(k_act2, dim3((unsigned)((a.m - a.m0 + 63) / 64)), dim3(256), lds, st, a);
    return hip_ok(hipGetLastError(), "k_act2 launch");
}

// BM: as large as keeps >= 2 workgroups per CU's worth of tiles (256 CUs); smaller batches use smaller tiles.
static int pick_tm(int64_t m, int tm_max = 0)
{
    const int tm = m >= 256 * 128 ? 4 : m >= 256 * 64 ? 2 : 1;
    return tm_max > 0 && tm > tm_max ? tm_max : tm;
}
// learner groups: envs_per_learner is a multiple of 32; tiles of 128 / 64 envs only where they divide it
static int group_tm_max(int64_t envs_per_learner) { return envs_per_learner % 128 == 0 ? 4 : envs_per_learner % 64 == 0 ? 2 : 1; }

template <int TM, int NW, int RD = 0>
static int launch_act(const ActArgs &a, hipStream_t st)
{
    constexpr int BM = 32 * TM;
    const size_t lds = act_lds_bytes<TM, NW, RD>();
    static_assert(act_lds_bytes<TM, NW, RD>() <= 160 * 1024, "k_act: LDS image exceeds 160 KB");
    static std::atomic<uint64_t> optin{0};                   // per device: see lds_optin
    if (int rc = lds_optin(optin, reinterpret_cast<const void *>(&k_act<TM, NW, RD>), (int)lds, "hipFuncSetAttribute(k_act)")) return rc;
    const unsigned grid = (unsigned)((a.m - a.m0 + BM - 1) / BM);
    hipLaunchKernelGGL((k_act<TM, NW, RD>), dim3(grid), dim3(64 * NW), lds, st, a);
    return hip_ok(hipGetLastError(), "k_act launch");
}

// Exchange slots of the column-split form, one slab per (device, stream): two streams may run the kernel at once and must not share
// slots.  Allocated and set to "empty" on first use (a synchronous hipMalloc + hipMemset, once), never freed: 256 KB per slab.
constexpr int kSplitMaxTiles = 512;
static int split_scratch(hipStream_t st, ActSplit *out)
{
    struct Slot { int dev; hipStream_t st; ActSplit x; };
    static Slot slots[32];
    static int nslots = 0;
    static std::atomic_flag lock = ATOMIC_FLAG_INIT;
    int dev = 0;
    if (int rc = hip_ok(hipGetDevice(&dev), "hipGetDevice")) return rc;
    while (lock.test_and_set(std::memory_order_acquire)) {}
    int rc = SHEMS_OK;
    int found = -1;
    for (int i = 0; i < nslots; ++i)
        if (slots[i].dev == dev && slots[i].st == st) { found = i; break; }
    if (found < 0) {
        if (nslots == 32) rc = set_error(SHEMS_ERR_ARG, "k_actg: more than 32 (device, stream) pairs use the column-split form");
        else {
            ActSplit x = {nullptr};
            const size_t bytes = (size_t)kSplitMaxTiles * 64 * sizeof(unsigned long long);
            rc = hip_ok(hipMalloc((void **)&x.slot, bytes), "hipMalloc(split slots)");
            if (!rc) rc = hip_ok(hipMemset(x.slot, 0xFF, bytes), "hipMemset(split slots)");       // synchronous: every slot empty before any launch
            if (!rc) { slots[nslots] = Slot{dev, st, x}; found = nslots++; }
        }
    }
    if (!rc) *out = slots[found].x;
    lock.clear(std::memory_order_release);
    return rc;
}

template <int TM, int NW, int NS, int RD>
static int launch_actg(const ActArgs &a, hipStream_t st)
{
    constexpr int BM = 32 * TM;
    constexpr size_t lds = actg_lds_bytes<TM, NW, RD>();
    static_assert(lds <= 160 * 1024, "k_actg: LDS image exceeds 160 KB");
    static std::atomic<uint64_t> optin{0};                   // per device: see lds_optin
    if (int rc = lds_optin(optin, reinterpret_cast<const void *>(&k_actg<TM, NW, NS, RD>), (int)lds, "hipFuncSetAttribute(k_actg)")) return rc;
    const int64_t tiles = (a.m - a.m0 + BM - 1) / BM;
    ActSplit x = {nullptr};
    if constexpr (NS == 2) {
        if (tiles > kSplitMaxTiles) return set_error(SHEMS_ERR_ARG, "k_actg: %lld env tiles exceed the split form's scratch", (long long)tiles);
        if (split_scratch(st, &x) != SHEMS_OK) {
            // no exchange slab for this (device, stream) -- the 33rd distinct stream of a long-lived process, or hipMalloc failed: the
            // one-workgroup-per-tile form needs none and writes the same bytes
            return launch_actg<1, 8, 1, RD>(a, st);
        }
    }
    hipLaunchKernelGGL((k_actg<TM, NW, NS, RD>), dim3((unsigned)(tiles * NS)), dim3(64 * NW), lds, st, a, x);
    return hip_ok(hipGetLastError(), "k_actg launch");
}

// Which form runs a launch of m envs (the knobs exist for the all-forms test and for A/B runs):
//   m > 8 192            k_act2           64-env tiles, two workgroups resident per CU            (SHEMS_ACT_FORM4 = 1 / 0: k_act, below)
//   4 096 < m <= 8 192   k_actg<1, 4, 2, 2>  32-env tiles, two workgroups per tile, ring of 2 chunks: both resident on one CU
//                        (per-tile reward sums asked for, or SHEMS_ACT_FORM = 8: k_actg<1, 8, 1>, 8 waves = 8 column groups; 10: force the ring-2 split form;
//                         SHEMS_ACT_FORM = 0 / 2 / 3: k_act<1, 4, .>)
//   m <= 4 096           k_actg<1, 4, 2>  32-env tiles, two workgroups (4 groups each) per tile   (SHEMS_ACT_FORM = 8 / 9: force one of the two)
// k_act (SHEMS_ACT_FORM4 = 1: free-running waves, 0: shared W2 stream): 128-env tiles from 32 768 envs, 64-env tiles from 16 384, 32 below.
static int act_form() { static const int f = []() { const char *e = getenv("SHEMS_ACT_FORM"); return e ? atoi(e) : -1; }(); return f; }
static int act_form4() { static const int f = []() { const char *e = getenv("SHEMS_ACT_FORM4"); return e ? atoi(e) : 2; }(); return f; }
// envs per workgroup tile (= per entry of block_reward) of the form that runs m envs
// (learner groups keep k_act's 128-env tiles: with 32 weight sets in flight two co-resident workgroups of different learners cost more
// in L2 than they win -- 151.9 against 145.7 us at 32 x 2 048 envs)
static int act_tile_envs(int64_t m, bool grouped = false)
{
    if (act_form4() == 2 && ((act_form() < 0 && m > 8192 && !grouped) || act_form() == 12)) return 64;
    return 32 * pick_tm(m);
}

static int dispatch_act(const ActArgs &a, hipStream_t st)
{
    const int form = act_form(), form4 = act_form4();
#ifdef SHEMS_STAMP_ACT
    const bool want_sum = false;                              // stamp builds: block_reward is the stamp buffer
#else
    const bool want_sum = a.block_reward != nullptr;          // per-tile reward sums: a form whose one workgroup finishes the whole tile
#endif
    const int64_t cnt = a.m - a.m0;                           // envs of this launch (a range launch: every form writes the same bytes)
    if (a.w2t) {                                              // tiled learner group: the free-running forms read W2 from the tiled regions
        const int tmt = pick_tm(cnt, a.tm_max);
        return tmt == 4 ? launch_act<4, 4, 2>(a, st) : tmt == 2 ? launch_act<2, 4, 2>(a, st) : launch_act<1, 4, 2>(a, st);
    }
    if (form4 == 2 && form < 0 && cnt > 8192 && a.gcount <= 1) return launch_act2(a, st);
    if (form4 == 2 && form == 12 && (a.tm_max == 0 || a.tm_max >= 2)) return launch_act2(a, st);      // A/B: the two-per-CU form (64-env tiles) at any size
    const int tm = pick_tm(cnt, a.tm_max);
    if (tm == 4) return form4 == 0 ? launch_act<4, 4>(a, st) : launch_act<4, 4, 2>(a, st);
    if (tm == 2) return form == 0 || form4 == 0 ? launch_act<2, 4>(a, st) : launch_act<2, 4, 2>(a, st);
    if (form == 0) return launch_act<1, 4>(a, st);
    if (form == 2) return launch_act<1, 4, 2>(a, st);
    if (form == 3) return launch_act<1, 4, 3>(a, st);
    // 4 096 < envs <= 8 192 (round 4): the split form with a ring of TWO chunks -- 76.5 KB per workgroup, so the two halves of a tile's work
    // are two independent workgroups resident on one CU (8 waves, two per SIMD, as the 8-wave form) without that form's coupling (its early
    // waves wait at the barrier for the late ones): 23.1 against 23.8 us at 8 192 envs.  At <= 4 096 envs (one workgroup per CU) the
    // shallower ring costs more than it wins (17.7 against 16.2 us): ring of three there.  Per-tile reward sums need ONE workgroup per tile.
    if ((form == 10 || (form < 0 && cnt > 128 * 32)) && !want_sum && (cnt + 31) / 32 <= kSplitMaxTiles) return launch_actg<1, 4, 2, 2>(a, st);
    // (a learner group whose env block admits only 32-env tiles can be large: beyond the split forms' exchange scratch the one-workgroup-per-tile form runs)
    const bool split_fits = (cnt + 31) / 32 <= kSplitMaxTiles;
    if (form == 8 || !split_fits || (form != 9 && (cnt > 128 * 32 || want_sum))) return launch_actg<1, 8, 1, 3>(a, st);
    return launch_actg<1, 4, 2, 3>(a, st);
}

// Wide networks (shems_wide.hip): the layers ran as matrix products and left partial sums of the output layer; this is the rest of the
// fused step -- b3 + the partials in index order, tanh, noise, clamp, scale_action, step!, remember -- one thread per env, the same
// act_env_tail as every form above (the same draws for the same (seed, tick, env)).
__global__ __launch_bounds__(256) void k_act_tail(ActArgs A, const float *__restrict__ part, int n_part, const float *__restrict__ b3)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.m) return;
    float p0 = 0.0f, p1 = 0.0f;
    for (int p = 0; p < n_part; ++p) {
        const float2 v = reinterpret_cast<const float2 *>(part)[(int64_t)p * A.m + i];
        p0 += v.x; p1 += v.y;
    }
    act_env_tail(A, i, b3[0] + p0, b3[1] + p1, 0, 0, nullptr);
}

static int wide_act(const ActArgs &a, int l1, int l2, float *d_ws, hipStream_t st)
{
    if (!d_ws || ((uintptr_t)d_ws & 15) != 0) return set_error(SHEMS_ERR_ARG, "shems_wide_act: 16-byte aligned workspace required");
    float *part = d_ws + wide_act_part_offset(l1, a.m);
    int n_part = 0;
    if (int rc = wide_actor_pre(a.p.actor, a.p.s_min, a.p.s_max, l1, l2, a.obs, a.m, d_ws, part, &n_part, st)) return rc;
    const float *b3 = a.p.actor + ((int64_t)kIn * l1 + l1 + (int64_t)l1 * l2 + l2 + (int64_t)l2 * kOut);
    hipLaunchKernelGGL(k_act_tail, dim3((unsigned)((a.m + 255) / 256)), dim3(256), 0, st, a, part, n_part, b3);
    return hip_ok(hipGetLastError(), "k_act_tail launch");
}

}  // namespace shems

using namespace shems;

extern "C" {

int shems_act_step_grid(int64_t n_envs, int64_t *out_blocks)
{
    if (n_envs <= 0 || !out_blocks) return set_error(SHEMS_ERR_ARG, "shems_act_step_grid: bad arguments");
    const int bm = act_tile_envs(n_envs);
    *out_blocks = (n_envs + bm - 1) / bm;
    return SHEMS_OK;
}

/* Which kernel shems_act_step_dev / shems_act_step_group_dev dispatches for n_envs envs (grouped != 0: a learner group), as the
 * profiler prints it: the same decisions as dispatch_act, environment overrides included. */
int shems_act_step_kernel(int64_t n_envs, int grouped, char *out, int32_t cap)
{
    if (n_envs <= 0 || !out || cap < 2) return set_error(SHEMS_ERR_ARG, "shems_act_step_kernel: bad arguments");
    const int form = act_form(), form4 = act_form4();
    const char *name;
    if (grouped == 2) {                                       // a learner group on the tiled working layout
        const int tmt = pick_tm(n_envs);
        name = tmt == 4 ? "shems::k_act<4, 4, 2>" : tmt == 2 ? "shems::k_act<2, 4, 2>" : "shems::k_act<1, 4, 2>";
    } else if (form4 == 2 && ((form < 0 && n_envs > 8192 && !grouped) || form == 12)) name = "shems::k_act2";
    else {
        const int tm = pick_tm(n_envs);
        if (tm == 4) name = form4 == 0 ? "shems::k_act<4, 4, 0>" : "shems::k_act<4, 4, 2>";
        else if (tm == 2) name = form == 0 || form4 == 0 ? "shems::k_act<2, 4, 0>" : "shems::k_act<2, 4, 2>";
        else if (form == 0) name = "shems::k_act<1, 4, 0>";
        else if (form == 2) name = "shems::k_act<1, 4, 2>";
        else if (form == 3) name = "shems::k_act<1, 4, 3>";
        else if ((form == 10 || (form < 0 && n_envs > 128 * 32)) && (n_envs + 31) / 32 <= kSplitMaxTiles) name = "shems::k_actg<1, 4, 2, 2>";
        else if (form == 8 || (n_envs + 31) / 32 > kSplitMaxTiles || (form != 9 && n_envs > 128 * 32)) name = "shems::k_actg<1, 8, 1, 3>";
        else name = "shems::k_actg<1, 4, 2, 3>";
    }
    snprintf(out, (size_t)cap, "%s", name);
    return SHEMS_OK;
}

/* The same for a learner group: the tile never straddles two learners, so envs_per_learner limits it. */
int shems_act_step_group_kernel(int64_t n_envs, int64_t envs_per_learner, int tiled, char *out, int32_t cap)
{
    if (n_envs <= 0 || envs_per_learner < 32 || envs_per_learner % 32 != 0 || !out || cap < 2)
        return set_error(SHEMS_ERR_ARG, "shems_act_step_group_kernel: bad arguments");
    const int form = act_form(), form4 = act_form4(), tmax = group_tm_max(envs_per_learner);
    const int tm = pick_tm(n_envs, tmax);
    const char *name;
    if (tiled) name = tm == 4 ? "shems::k_act<4, 4, 2>" : tm == 2 ? "shems::k_act<2, 4, 2>" : "shems::k_act<1, 4, 2>";
    else if (form4 == 2 && form == 12 && tmax >= 2) name = "shems::k_act2";
    else if (tm == 4) name = form4 == 0 ? "shems::k_act<4, 4, 0>" : "shems::k_act<4, 4, 2>";
    else if (tm == 2) name = form == 0 || form4 == 0 ? "shems::k_act<2, 4, 0>" : "shems::k_act<2, 4, 2>";
    else return shems_act_step_kernel(n_envs, 1, out, cap);          // 32-env tiles: the ungrouped decision tree below tm = 2
    snprintf(out, (size_t)cap, "%s", name);
    return SHEMS_OK;
}

static int check_act(const shems_act_params *p, const char *fn)
{
    if (!p || !p->actor || !p->s_min || !p->s_max) return set_error(SHEMS_ERR_ARG, "%s: actor / s_min / s_max required", fn);
    if (p->noise_kind < SHEMS_NOISE_GAUSS || p->noise_kind > SHEMS_NOISE_EPS) return set_error(SHEMS_ERR_ARG, "%s: unknown noise_kind %d", fn, p->noise_kind);
    if (p->train && p->noise_kind == SHEMS_NOISE_OU && !p->ou_state) return set_error(SHEMS_ERR_ARG, "%s: OU noise needs ou_state", fn);
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
    if (int rc = check_view(v, "shems_act_step_dev")) return rc;
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
        if (window->offset < 0 || window->offset >= v->n_envs)
            return set_error(SHEMS_ERR_ARG, "shems_act_step_dev: ring window offset %lld outside the batch of %lld envs",
                             (long long)window->offset, (long long)v->n_envs);
        a.ring = *ring; a.win = *window; a.use_ring = 1;
    }
    return dispatch_act(a, (hipStream_t)stream);
}

}  // extern "C"


extern "C" {

int shems_act_step_range_dev(const shems_view *v, const shems_act_params *p, int64_t env_lo, int64_t env_count, float *d_rewards_f32,
                             const shems_replay *ring, const shems_ring_window *window, void *stream)
{
    if (int rc = check_act(p, "shems_act_step_range_dev")) return rc;
    if (int rc = check_view(v, "shems_act_step_range_dev")) return rc;
    if (env_lo < 0 || env_count <= 0 || env_lo + env_count > v->n_envs)
        return set_error(SHEMS_ERR_ARG, "shems_act_step_range_dev: envs [%lld, %lld + %lld) outside the batch of %lld", (long long)env_lo,
                         (long long)env_lo, (long long)env_count, (long long)v->n_envs);
    ActArgs a;
    std::memset(&a, 0, sizeof a);
    a.v = *v; a.p = *p; a.obs = v->obs; a.m0 = env_lo; a.m = env_lo + env_count;
    a.rewards_f32 = d_rewards_f32;
    a.do_step = 1;
    if (ring && window && window->count > 0) {
        if (ring->capacity <= 0 || !ring->s || !ring->a || !ring->r || !ring->s2 || !ring->done)
            return set_error(SHEMS_ERR_ARG, "shems_act_step_range_dev: incomplete replay ring");
        if (window->count > ring->capacity || window->count > v->n_envs || window->pos < 0 || window->offset < 0 || window->offset >= v->n_envs)
            return set_error(SHEMS_ERR_ARG, "shems_act_step_range_dev: ring window outside the ring or the batch");
        a.ring = *ring; a.win = *window; a.use_ring = 1;
    }
    return dispatch_act(a, (hipStream_t)stream);
}

int shems_wide_actor_forward_dev(const shems_act_params *p, int32_t l1, int32_t l2, const float *d_obs, int64_t m, float *d_a, float *d_ws,
                                 void *stream)
{
    if (int rc = check_act(p, "shems_wide_actor_forward_dev")) return rc;
    if (!d_obs || !d_a || m <= 0) return set_error(SHEMS_ERR_ARG, "shems_wide_actor_forward_dev: bad buffers");
    ActArgs a;
    std::memset(&a, 0, sizeof a);
    a.p = *p; a.obs = d_obs; a.m = m; a.a_out = d_a;
    return wide_act(a, l1, l2, d_ws, (hipStream_t)stream);
}

int shems_wide_act_step_dev(const shems_view *v, const shems_act_params *p, int32_t l1, int32_t l2, float *d_ws, float *d_a, double *d_rewards,
                            float *d_rewards_f32, double *d_returns_acc, const shems_replay *ring, const shems_ring_window *window,
                            void *stream)
{
    if (int rc = check_act(p, "shems_wide_act_step_dev")) return rc;
    if (int rc = check_view(v, "shems_wide_act_step_dev")) return rc;
    ActArgs a;
    std::memset(&a, 0, sizeof a);
    a.v = *v; a.p = *p; a.obs = v->obs; a.m = v->n_envs; a.a_out = d_a;
    a.rewards = d_rewards; a.rewards_f32 = d_rewards_f32; a.returns_acc = d_returns_acc;
    a.do_step = 1;
    if (ring && window && window->count > 0) {
        if (ring->capacity <= 0 || !ring->s || !ring->a || !ring->r || !ring->s2 || !ring->done)
            return set_error(SHEMS_ERR_ARG, "shems_wide_act_step_dev: incomplete replay ring");
        if (window->count > ring->capacity || window->count > v->n_envs || window->pos < 0 || window->offset < 0 || window->offset >= v->n_envs)
            return set_error(SHEMS_ERR_ARG, "shems_wide_act_step_dev: ring window outside the ring or the batch");
        a.ring = *ring; a.win = *window; a.use_ring = 1;
    }
    return wide_act(a, l1, l2, d_ws, (hipStream_t)stream);
}

static int act_step_group(const char *fn, const shems_view *v, const shems_act_params *p0, const shems_group *g, const float *w2t, float *d_a,
                          double *d_returns_acc, const shems_replay *ring0, const shems_ring_window *window, void *stream)
{
    if (int rc = check_act(p0, fn)) return rc;
    if (int rc = check_view(v, fn)) return rc;
    if (!g || g->count < 1 || g->stride_bytes < 0 || (g->stride_bytes & 15) != 0 || (g->count > 1 && g->stride_bytes == 0))
        return set_error(SHEMS_ERR_ARG, "%s: shems_group needs count >= 1 and a 16-byte-multiple stride", fn);
    if (g->envs_per_learner < 32 || g->envs_per_learner % 32 != 0 || g->envs_per_learner * g->count != v->n_envs)
        return set_error(SHEMS_ERR_ARG, "%s: envs_per_learner must be a multiple of 32 and count * envs_per_learner == n_envs "
                         "(got %lld x %d for %lld envs)", fn, (long long)g->envs_per_learner, g->count, (long long)v->n_envs);
    ActArgs a;
    std::memset(&a, 0, sizeof a);
    a.v = *v; a.p = *p0; a.obs = v->obs; a.m = v->n_envs; a.a_out = d_a;
    a.returns_acc = d_returns_acc;
    a.do_step = 1;
    a.gcount = g->count; a.gstride = g->count > 1 ? g->stride_bytes : 0; a.genvs = g->envs_per_learner;
    a.w2t = w2t;
    a.tm_max = group_tm_max(g->envs_per_learner);
    if (ring0 && window && window->count > 0) {
        if (ring0->capacity <= 0 || !ring0->s || !ring0->a || !ring0->r || !ring0->s2 || !ring0->done)
            return set_error(SHEMS_ERR_ARG, "%s: incomplete replay ring", fn);
        if (window->count > ring0->capacity || window->count > g->envs_per_learner || window->pos < 0)
            return set_error(SHEMS_ERR_ARG, "%s: ring window larger than the ring or a learner's env block", fn);
        if (window->offset < 0 || window->offset >= g->envs_per_learner)
            return set_error(SHEMS_ERR_ARG, "%s: ring window offset %lld outside a learner's block of %lld envs", fn,
                             (long long)window->offset, (long long)g->envs_per_learner);
        a.ring = *ring0; a.win = *window; a.use_ring = 1;
    }
    return dispatch_act(a, (hipStream_t)stream);
}

int shems_act_step_group_dev(const shems_view *v, const shems_act_params *p0, const shems_group *g, float *d_a,
                             double *d_returns_acc, const shems_replay *ring0, const shems_ring_window *window, void *stream)
{
    return act_step_group("shems_act_step_group_dev", v, p0, g, nullptr, d_a, d_returns_acc, ring0, window, stream);
}

int shems_act_step_group_tiled_dev(const shems_view *v, const shems_act_params *p0, const shems_group *g, const shems_group_w2t *t, float *d_a,
                                   double *d_returns_acc, const shems_replay *ring0, const shems_ring_window *window, void *stream)
{
    if (!t || !t->actor || ((uintptr_t)t->actor & 15) != 0)
        return set_error(SHEMS_ERR_ARG, "shems_act_step_group_tiled_dev: shems_group_w2t.actor must be a 16-byte aligned device pointer");
    return act_step_group("shems_act_step_group_tiled_dev", v, p0, g, t->actor, d_a, d_returns_acc, ring0, window, stream);
}

}  // extern "C"
