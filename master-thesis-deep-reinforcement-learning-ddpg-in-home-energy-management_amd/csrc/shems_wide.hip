// shems_wide.hip -- networks LARGER than the (250, 500) the tile maps of shems_policy.hip / shems_ddpg.hip are written for.
//
// The reference's hyper-parameter grids hold one such point: (L1, L2) = (300, 600) (input09_08_on_01-09_eval.jl:62-66 digit 3 = 0,
// input.jl:58-66).  Smaller networks run on the tuned kernels by zero padding (ddpg.pad_net); a larger one cannot, so it runs
// here, layer by layer, exactly as the reference's Flux / CUBLAS path does (Dense = W*x .+ b, Zygote's pullbacks: DDPG.jl:21-46,
// 99-145): every layer, forward or backward, is ONE general matrix product with a fused bias / relu / relu' epilogue.
//
//   k_wgemm   C[M][N] = epilogue(sum_k A(i, k) B(k, j)) on v_mfma_f32_32x32x2_f32, 64 x 64 tile per workgroup of 4 waves, K in
//             stages of 16 through LDS with the next stage's global loads in flight; both operands by (row, column) element
//             strides, so W (Flux layout [in][out]), its transpose, and the sample-major activations [m][features] all go in
//             without a copy.  fp32 accumulation in a fixed order: results are reproducible bit for bit, and tolerance-class
//             against the oracle like the tuned kernels (tests/test_wide_gpu.py).
//   the rest  elementwise / column-sum kernels: normalize, minibatch sample + gather (the same Philox sampler as the tuned path:
//             the same (seed, tick) draws the same slots), tanh + concatenation, the two loss heads.
// ADAM + soft target update are shems_ddpg.hip's sweep (adam_soft_sweep: the same arithmetic on any parameter count).
//
// This path is about running the grid point, not about the roofline: ~45 launches per replay() (~0.3 ms), a vector step of 65 536
// envs is three GEMMs through HBM-resident activations.  The headline configuration never comes here.
#include <hip/hip_runtime.h>

#include <cstring>

#include "philox.h"
#include "shems_internal.h"

namespace shems {

typedef float wf32x16 __attribute__((ext_vector_type(16)));

constexpr int WBP = 128;                 // minibatch rows one update pass holds (as shems_ddpg.hip's BP)
constexpr int WSIN = 9, WAIN = 2, WCIN = 11;
constexpr int GT = 64, GK = 16, GKB = 4, GLD = GT + 4, GR = GT * GK / 256;   // tile, K per stage (2^GKB), LDS row stride, elements per thread and operand
static_assert((1 << GKB) == GK, "GK");

struct GemmArgs {
    const float *A, *B;
    float *C;
    int M, N, K;
    int64_t sai, sak, sbk, sbj, ldc;     // A(i, k) = A[i sai + k sak], B(k, j) = B[k sbk + j sbj], C[i ldc + j]
    const float *bias;                   // [N] or null
    const float *gate;                   // [M][ldg] or null: C = gate > 0 ? C : 0  (relu' read off the stored post-relu activations)
    int64_t ldg;
    int relu;
};

__global__ __launch_bounds__(256) void k_wgemm(GemmArgs G)
{
    __shared__ float As[2][GK][GLD], Bs[2][GK][GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.x * GT, n0 = (int64_t)blockIdx.y * GT;
    const bool a_kfast = G.sak == 1, b_jfast = G.sbj == 1;
    // element e = tid + 256 r of a 64 x GK operand tile: (row, k) with the memory-contiguous index fastest across threads
    int ai[GR], ak[GR], bj[GR], bk[GR];
#pragma unroll
    for (int r = 0; r < GR; ++r) {
        const int e = tid + 256 * r;
        ai[r] = a_kfast ? e >> GKB : e & 63;  ak[r] = a_kfast ? e & (GK - 1) : e >> 6;
        bj[r] = b_jfast ? e & 63 : e >> GKB;  bk[r] = b_jfast ? e >> 6 : e & (GK - 1);
    }
    // the global loads of stage s + 1 are in flight while stage s is multiplied; double-buffered LDS, one barrier per stage.  (With
    // thousands of workgroups the latency is hidden by occupancy: a deeper register ring and unpredicated clamped loads, which pay off
    // in the small-M kernel below, measured slower here -- 480-490 against 443 us for the vector step of 65 536 envs.)
    float ra[GR], rb[GR];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int r = 0; r < GR; ++r) {
            const int64_t i = m0 + ai[r], j = n0 + bj[r];
            const int ka = k0 + ak[r], kb = k0 + bk[r];
            ra[r] = (i < G.M && ka < G.K) ? G.A[i * G.sai + ka * G.sak] : 0.0f;
            rb[r] = (j < G.N && kb < G.K) ? G.B[kb * G.sbk + j * G.sbj] : 0.0f;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int r = 0; r < GR; ++r) { As[buf][ak[r]][ai[r]] = ra[r]; Bs[buf][bk[r]][bj[r]] = rb[r]; }
    };
    wf32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    fetch(0);
    stash(0);
    __syncthreads();
    const int nst = (G.K + GK - 1) / GK;
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        if (s + 1 < nst) fetch((s + 1) * GK);
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][kk + lh][wi * 32 + li], Bs[buf][kk + lh][wj * 32 + li], acc, 0, 0, 0);
        if (s + 1 < nst) stash(buf ^ 1);
        __syncthreads();
    }
    const int64_t j = n0 + wj * 32 + li;
    if (j < G.N) {
        const float bj_ = G.bias ? G.bias[j] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t i = m0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (i < G.M) {
                float v = acc[r] + bj_;
                if (G.relu) v = fmaxf(v, 0.0f);
                if (G.gate) v = G.gate[i * G.ldg + j] > 0.0f ? v : 0.0f;
                G.C[i * G.ldc + j] = v;
            }
        }
    }
}

// The minibatch-sized products (M <= a few hundred rows): few tiles and a long K.  One wave's chain of 32x32x2 MFMAs costs 32 cycles
// per k whatever the memory system does (K = 600: 9 us), and a 64 x 64 tile per workgroup leaves most CUs idle (N = 600: 20
// workgroups).  So here a workgroup owns a 32 x 32 tile and its four waves split every 64-deep K stage four ways (16 k each); the
// four partial tiles meet in LDS at the end and are added in wave order (a fixed order).  Same operands, strides and epilogue.
constexpr int SK = 64, SKB = 6, SLD = 32 + 4, SR = 32 * SK / 256;
__global__ __launch_bounds__(256) void k_wgemm_sk(GemmArgs G)
{
    __shared__ float As[2][SK][SLD], Bs[2][SK][SLD];
    __shared__ float red[4][16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * 32, n0 = (int64_t)blockIdx.y * 32;
    const bool a_kfast = G.sak == 1, b_jfast = G.sbj == 1;
    int ai[SR], ak[SR], bj[SR], bk[SR];
#pragma unroll
    for (int r = 0; r < SR; ++r) {
        const int e = tid + 256 * r;
        ai[r] = a_kfast ? e >> SKB : e & 31;  ak[r] = a_kfast ? e & (SK - 1) : e >> 5;
        bj[r] = b_jfast ? e & 31 : e >> SKB;  bk[r] = b_jfast ? e >> 5 : e & (SK - 1);
    }
    // global loads run GP stages ahead in a ring of register slots; every load is unconditional, from a clamped (always valid) address,
    // and the zero of an out-of-range element is selected when the slot is stashed (a load under a lane predicate is sunk into a branch
    // behind s_waitcnt vmcnt(0), which serialises the ring).  One barrier per stage: stash(s) -> barrier -> MFMAs(s); buffer s & 1 was
    // last read in stage s - 2, which every thread left before anyone passed barrier s - 1.
    constexpr int GP = 2;
    float ra[GP][SR], rb[GP][SR];
    unsigned oka[GP], okb[GP];
    auto fetch = [&](int k0, float (&xa)[SR], float (&xb)[SR], unsigned &ma, unsigned &mb) {
        ma = mb = 0u;
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            const int64_t i = m0 + ai[r], j = n0 + bj[r];
            const int ka = k0 + ak[r], kb = k0 + bk[r];
            ma |= (i < G.M && ka < G.K) ? 1u << r : 0u;
            mb |= (j < G.N && kb < G.K) ? 1u << r : 0u;
            xa[r] = G.A[min(i, (int64_t)G.M - 1) * G.sai + min(ka, G.K - 1) * G.sak];
            xb[r] = G.B[min(kb, G.K - 1) * G.sbk + min(j, (int64_t)G.N - 1) * G.sbj];
        }
    };
    auto stash = [&](int buf, const float (&xa)[SR], const float (&xb)[SR], unsigned ma, unsigned mb) {
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            As[buf][ak[r]][ai[r]] = (ma >> r & 1u) ? xa[r] : 0.0f;
            Bs[buf][bk[r]][bj[r]] = (mb >> r & 1u) ? xb[r] : 0.0f;
        }
    };
    wf32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int nst = (G.K + SK - 1) / SK;
#pragma unroll
    for (int u = 0; u < GP; ++u) fetch(u * SK, ra[u], rb[u], oka[u], okb[u]);
    for (int s0 = 0; s0 < nst; s0 += GP) {
#pragma unroll
        for (int u = 0; u < GP; ++u) {
            const int s = s0 + u;
            if (s < nst) {
                const int buf = s & 1;
                stash(buf, ra[u], rb[u], oka[u], okb[u]);
                __syncthreads();
                if (s + GP < nst) fetch((s + GP) * SK, ra[u], rb[u], oka[u], okb[u]);
#pragma unroll
                for (int kk = 0; kk < SK / 4; kk += 2)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][wave * (SK / 4) + kk + lh][li], Bs[buf][wave * (SK / 4) + kk + lh][li], acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q, r = e >> 6, l2 = e & 63;
        const int64_t i = m0 + (r & 3) + 8 * (r >> 2) + 4 * (l2 >> 5), j = n0 + (l2 & 31);
        if (i < G.M && j < G.N) {
            float v = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
            if (G.bias) v += G.bias[j];
            if (G.relu) v = fmaxf(v, 0.0f);
            if (G.gate) v = G.gate[i * G.ldg + j] > 0.0f ? v : 0.0f;
            G.C[i * G.ldc + j] = v;
        }
    }
}

static int gemm(hipStream_t st, const float *A, int64_t sai, int64_t sak, const float *B, int64_t sbk, int64_t sbj, float *C, int64_t ldc,
                int64_t M, int N, int K, const float *bias = nullptr, int relu = 0, const float *gate = nullptr, int64_t ldg = 0)
{
    GemmArgs g{A, B, C, (int)M, N, K, sai, sak, sbk, sbj, ldc, bias, gate, ldg, relu};
    if (M <= 512) {
        hipLaunchKernelGGL(k_wgemm_sk, dim3((unsigned)((M + 31) / 32), (unsigned)((N + 31) / 32)), dim3(256), 0, st, g);
        return hip_ok(hipGetLastError(), "k_wgemm_sk launch");
    }
    hipLaunchKernelGGL(k_wgemm, dim3((unsigned)((M + GT - 1) / GT), (unsigned)((N + GT - 1) / GT)), dim3(256), 0, st, g);
    return hip_ok(hipGetLastError(), "k_wgemm launch");
}

// A network (in -> l1 -> l2 -> out) in the flat Flux layout.
struct WNet {
    const float *W1, *b1, *W2, *b2, *W3, *b3;
    int in, l1, l2, out;
};
static WNet wnet(const float *P, int in, int l1, int l2, int out)
{
    const float *W1 = P, *b1 = W1 + (int64_t)in * l1, *W2 = b1 + l1, *b2 = W2 + (int64_t)l1 * l2, *W3 = b2 + l2, *b3 = W3 + (int64_t)l2 * out;
    return WNet{W1, b1, W2, b2, W3, b3, in, l1, l2, out};
}
static int64_t wnet_size(int in, int l1, int l2, int out) { return (int64_t)in * l1 + l1 + (int64_t)l1 * l2 + l2 + (int64_t)l2 * out + out; }

// X [m][in] -> H1 [m][l1], H2 [m][l2] (post-relu), P [m][out] (pre-activation of the last layer, b3 included)
static int net_forward(hipStream_t st, const WNet &n, const float *X, int64_t m, float *H1, float *H2, float *P)
{
    if (int rc = gemm(st, X, n.in, 1, n.W1, n.l1, 1, H1, n.l1, m, n.l1, n.in, n.b1, 1)) return rc;
    if (int rc = gemm(st, H1, n.l1, 1, n.W2, n.l2, 1, H2, n.l2, m, n.l2, n.l1, n.b2, 1)) return rc;
    return gemm(st, H2, n.l2, 1, n.W3, n.out, 1, P, n.out, m, n.out, n.l2, n.b3, 0);
}

// obs [m][9] -> (obs - s_min) / ((s_max - s_min) + 1f-8)  (normalize, MPS:55-57)
__global__ __launch_bounds__(256) void k_wnorm(const float *__restrict__ obs, const float *__restrict__ lo, const float *__restrict__ hi,
                                               float *__restrict__ out, int64_t count)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < count) {
        const int k = (int)(e % WSIN);
        out[e] = (obs[e] - lo[k]) / ((hi[k] - lo[k]) + 1e-8f);
    }
}

// ---- update workspace (floats; sample-major, WBP rows) ----------------------------------------------------------------------
struct WWs {
    float *XS, *XS2, *XC, *XC2, *XQ;          // [WBP][9] normalize(s), normalize(s'); [WBP][11] [s; a], [s'; actor_target(s')], [s; actor(s)]
    float *R, *DONE, *Y, *Q, *Q2, *DQ, *DQA;  // [WBP]
    int32_t *IDX;                             // [WBP] sampled ring slots
    float *PA, *PT, *API, *DA, *D3;           // [WBP][2] actor pre-activation, target actor's, a_pi; [WBP][11] d loss / d [s; a_pi]; [WBP][2]
    float *T1, *T2, *H1c, *H2c, *H1a, *H2a, *H1q, *H2q, *G1, *G2;
    int64_t total;
};
static WWs wws(float *base, int l1, int l2)
{
    WWs w;
    int64_t o = 0;
    auto take = [&](int64_t n) { float *p = base ? base + o : nullptr; o += (n + 3) / 4 * 4; return p; };
    w.XS = take(WBP * WSIN); w.XS2 = take(WBP * WSIN); w.XC = take(WBP * WCIN); w.XC2 = take(WBP * WCIN); w.XQ = take(WBP * WCIN);
    w.R = take(WBP); w.DONE = take(WBP); w.Y = take(WBP); w.Q = take(WBP); w.Q2 = take(WBP); w.DQ = take(WBP); w.DQA = take(WBP);
    w.IDX = reinterpret_cast<int32_t *>(take(WBP));
    w.PA = take(WBP * WAIN); w.PT = take(WBP * WAIN); w.API = take(WBP * WAIN); w.DA = take(WBP * WCIN); w.D3 = take(WBP * WAIN);
    w.T1 = take((int64_t)WBP * l1); w.T2 = take((int64_t)WBP * l2);
    w.H1c = take((int64_t)WBP * l1); w.H2c = take((int64_t)WBP * l2);
    w.H1a = take((int64_t)WBP * l1); w.H2a = take((int64_t)WBP * l2);
    w.H1q = take((int64_t)WBP * l1); w.H2q = take((int64_t)WBP * l2);
    w.G1 = take((int64_t)WBP * l1); w.G2 = take((int64_t)WBP * l2);
    w.total = o;
    return w;
}

struct WPrep {
    shems_replay ring;
    int64_t ring_len, excl_pos, excl_count;
    uint64_t seed;
    uint32_t tick;
    int batch;
    const float *lo, *hi;
    WWs w;
};
// getData (MPS:31-42) + normalize: thread m = minibatch row m.  The sampler is shems_ddpg.hip's prep_load.
__global__ __launch_bounds__(WBP) void k_wprep(WPrep A)
{
    const int m = threadIdx.x;
    const bool live = m < A.batch;
    const u32x4 x = philox4x32_10((uint32_t)(m >> 2), 0u, A.tick, kStreamSample, (uint32_t)A.seed, (uint32_t)(A.seed >> 32));
    const uint32_t wd = (m & 3) == 0 ? x.x : (m & 3) == 1 ? x.y : (m & 3) == 2 ? x.z : x.w;
    int64_t j = (int64_t)(wd % (uint32_t)(A.ring_len - A.excl_count));
    if (A.excl_count > 0) j = (A.excl_pos + A.excl_count + j) % A.ring.capacity;
    const WWs &w = A.w;
#pragma unroll
    for (int k = 0; k < WSIN; ++k) {
        const float lo = A.lo[k], den = (A.hi[k] - lo) + 1e-8f;
        const float x1 = live ? (A.ring.s[j * WSIN + k] - lo) / den : 0.0f;
        const float x2 = live ? (A.ring.s2[j * WSIN + k] - lo) / den : 0.0f;
        w.XS[m * WSIN + k] = x1; w.XS2[m * WSIN + k] = x2;
        w.XC[m * WCIN + k] = x1; w.XC2[m * WCIN + k] = x2; w.XQ[m * WCIN + k] = x1;
    }
    w.XC[m * WCIN + 9] = live ? A.ring.a[j * 2] : 0.0f;
    w.XC[m * WCIN + 10] = live ? A.ring.a[j * 2 + 1] : 0.0f;
    w.R[m] = live ? A.ring.r[j] : 0.0f;
    w.DONE[m] = live && A.ring.done[j] ? 1.0f : 0.0f;
    w.DQA[m] = live ? -1.0f / (float)A.batch : 0.0f;                 // d(-mean q) / dq
    w.IDX[m] = live ? (int32_t)j : -1;
}

// a = tanh(P) for the live rows -> A_out [WBP][2] (may be null) and the action columns of a [WBP][11] critic input
__global__ __launch_bounds__(WBP) void k_wtanh_cat(const float *__restrict__ P, float *__restrict__ a_out, float *__restrict__ cat, int batch)
{
    const int m = threadIdx.x;
    const float a0 = m < batch ? tanhf(P[2 * m]) : 0.0f, a1 = m < batch ? tanhf(P[2 * m + 1]) : 0.0f;
    if (a_out) { a_out[2 * m] = a0; a_out[2 * m + 1] = a1; }
    cat[m * WCIN + 9] = a0;
    cat[m * WCIN + 10] = a1;
}

// out[j] = sum_m D[m][j], m ascending (one thread per column: a fixed order)
__global__ __launch_bounds__(256) void k_wcolsum(const float *__restrict__ D, int n, float *__restrict__ out)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    float s = 0.0f;
    for (int m = 0; m < WBP; ++m) s += D[(int64_t)m * n + j];
    out[j] = s;
}

// critic loss head (DDPG.jl:131-135): y = r + gamma (1 - done) q', dq = 2 (q - y) / B, loss = mean((q - y)^2)
__global__ __launch_bounds__(WBP) void k_wloss(WWs w, float gamma, int batch, float *loss)
{
    __shared__ float red[WBP];
    const int m = threadIdx.x;
    const float y = w.R[m] + gamma * (1.0f - w.DONE[m]) * w.Q2[m];
    const float diff = m < batch ? w.Q[m] - y : 0.0f;
    w.Y[m] = y;
    w.DQ[m] = 2.0f * diff / (float)batch;
    red[m] = diff * diff;
    __syncthreads();
    if (m == 0) {
        float s = 0.0f;
        for (int i = 0; i < WBP; ++i) s += red[i];
        loss[0] = s / (float)batch;
    }
}

// actor loss head (DDPG.jl:137-140): loss = -mean(q); error at the actor's pre-tanh output = d loss / d a_pi * (1 - a_pi^2)
__global__ __launch_bounds__(WBP) void k_wactor_head(WWs w, int batch, float *loss)
{
    __shared__ float red[WBP];
    const int m = threadIdx.x;
    const bool live = m < batch;
#pragma unroll
    for (int o = 0; o < WAIN; ++o) {
        const float a = w.API[2 * m + o];
        w.D3[2 * m + o] = live ? w.DA[m * WCIN + 9 + o] * (1.0f - a * a) : 0.0f;
    }
    red[m] = live ? w.Q[m] : 0.0f;
    __syncthreads();
    if (m == 0) {
        float s = 0.0f;
        for (int i = 0; i < WBP; ++i) s += red[i];
        loss[1] = -s / (float)batch;
    }
}

// Zygote's pullback of Chain(Dense, Dense, Dense) for the error d3 [WBP][out] at the last layer's pre-activation: parameter
// gradients into `grad` (flat Flux layout); dX [WBP][in] = d loss / d input if asked for.  X, H1, H2: what the forward pass kept.
static int net_backward(hipStream_t st, const WNet &n, const float *X, const float *H1, const float *H2, const float *d3, float *grad,
                        float *G1, float *G2, float *dX)
{
    const int64_t oW1 = 0, ob1 = oW1 + (int64_t)n.in * n.l1, oW2 = ob1 + n.l1, ob2 = oW2 + (int64_t)n.l1 * n.l2, oW3 = ob2 + n.l2,
                  ob3 = oW3 + (int64_t)n.l2 * n.out;
    float *gW1 = grad ? grad + oW1 : nullptr, *gb1 = grad ? grad + ob1 : nullptr, *gW2 = grad ? grad + oW2 : nullptr,
          *gb2 = grad ? grad + ob2 : nullptr, *gW3 = grad ? grad + oW3 : nullptr, *gb3 = grad ? grad + ob3 : nullptr;
    if (grad) {
        if (int rc = gemm(st, H2, 1, n.l2, d3, n.out, 1, gW3, n.out, n.l2, n.out, WBP)) return rc;                       // gW3 = H2' d3
        hipLaunchKernelGGL(k_wcolsum, dim3(1), dim3(256), 0, st, d3, n.out, gb3);
    }
    if (int rc = gemm(st, d3, n.out, 1, n.W3, 1, n.out, G2, n.l2, WBP, n.l2, n.out, nullptr, 0, H2, n.l2)) return rc;     // (d3 W3') .* relu'
    if (grad) {
        hipLaunchKernelGGL(k_wcolsum, dim3((n.l2 + 255) / 256), dim3(256), 0, st, G2, n.l2, gb2);
        if (int rc = gemm(st, H1, 1, n.l1, G2, n.l2, 1, gW2, n.l2, n.l1, n.l2, WBP)) return rc;                          // gW2 = H1' G2
    }
    if (int rc = gemm(st, G2, n.l2, 1, n.W2, 1, n.l2, G1, n.l1, WBP, n.l1, n.l2, nullptr, 0, H1, n.l1)) return rc;        // (G2 W2') .* relu'
    if (grad) {
        hipLaunchKernelGGL(k_wcolsum, dim3((n.l1 + 255) / 256), dim3(256), 0, st, G1, n.l1, gb1);
        if (int rc = gemm(st, X, 1, n.in, G1, n.l1, 1, gW1, n.l1, n.in, n.l1, WBP)) return rc;                            // gW1 = X' G1
    }
    if (dX)
        if (int rc = gemm(st, G1, n.l1, 1, n.W1, 1, n.l1, dX, n.in, WBP, n.in, n.l1)) return rc;                          // G1 W1'
    return hip_ok(hipGetLastError(), "wide backward launches");
}

static int check_shape(int l1, int l2, const char *fn)
{
    if (l1 < 1 || l2 < 1 || l1 > 4096 || l2 > 4096) return set_error(SHEMS_ERR_ARG, "%s: hidden sizes must be in 1..4096 (got %d, %d)", fn, l1, l2);
    return SHEMS_OK;
}
static int check_wide(const shems_ddpg *d, int l1, int l2, const char *fn)
{
    if (int rc = check_shape(l1, l2, fn)) return rc;
    if (!d || !d->actor || !d->critic || !d->actor_t || !d->critic_t || !d->m_actor || !d->v_actor || !d->m_critic ||
        !d->v_critic || !d->grad_actor || !d->grad_critic || !d->s_min || !d->s_max || !d->ws || !d->losses)
        return set_error(SHEMS_ERR_ARG, "%s: shems_ddpg has a NULL buffer", fn);
    if (d->batch < 1 || d->batch > WBP) return set_error(SHEMS_ERR_ARG, "%s: batch must be in 1..128 (got %d)", fn, d->batch);
    return SHEMS_OK;
}

// used by shems_policy.hip's wide act entry points
int wide_actor_pre(const float *actor, const float *s_min, const float *s_max, int l1, int l2, const float *d_obs, int64_t m, float *d_ws,
                   float *d_pre, hipStream_t st)
{
    if (int rc = check_shape(l1, l2, "shems_wide_act")) return rc;
    float *xn = d_ws, *H1 = xn + (m * WSIN + 3) / 4 * 4, *H2 = H1 + m * l1;
    const int64_t cnt = m * WSIN;
    hipLaunchKernelGGL(k_wnorm, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, d_obs, s_min, s_max, xn, cnt);
    return net_forward(st, wnet(actor, WSIN, l1, l2, WAIN), xn, m, H1, H2, d_pre);
}
int64_t wide_act_ws_floats(int l1, int l2, int64_t m) { return (m * WSIN + 3) / 4 * 4 + m * ((int64_t)l1 + l2) + 2 * m; }

}  // namespace shems

using namespace shems;

extern "C" {

int shems_wide_params(int32_t l1, int32_t l2, int64_t *n_actor, int64_t *n_critic)
{
    if (int rc = check_shape(l1, l2, "shems_wide_params")) return rc;
    if (n_actor) *n_actor = wnet_size(WSIN, l1, l2, WAIN);
    if (n_critic) *n_critic = wnet_size(WCIN, l1, l2, 1);
    return SHEMS_OK;
}

int shems_wide_workspace_floats(int32_t l1, int32_t l2, int64_t *out)
{
    if (int rc = check_shape(l1, l2, "shems_wide_workspace_floats")) return rc;
    if (!out) return set_error(SHEMS_ERR_ARG, "shems_wide_workspace_floats: NULL");
    *out = wws(nullptr, l1, l2).total;
    return SHEMS_OK;
}

int shems_wide_act_workspace_floats(int32_t l1, int32_t l2, int64_t m, int64_t *out)
{
    if (int rc = check_shape(l1, l2, "shems_wide_act_workspace_floats")) return rc;
    if (!out || m <= 0) return set_error(SHEMS_ERR_ARG, "shems_wide_act_workspace_floats: bad arguments");
    *out = wide_act_ws_floats(l1, l2, m);
    return SHEMS_OK;
}

int shems_wide_critic_grad_ex(const shems_ddpg *d, int32_t l1, int32_t l2, const shems_replay *ring, int64_t ring_len, uint64_t seed,
                              uint32_t tick, int64_t excl_pos, int64_t excl_count, void *stream)
{
    if (int rc = check_wide(d, l1, l2, "shems_wide_critic_grad_ex")) return rc;
    if (!ring || !ring->s || !ring->a || !ring->r || !ring->s2 || !ring->done || ring_len < 1 || ring_len > ring->capacity)
        return set_error(SHEMS_ERR_ARG, "shems_wide_critic_grad_ex: bad replay ring / length");
    if (excl_count < 0 || excl_pos < 0 || (excl_count > 0 && (ring_len != ring->capacity || excl_count >= ring_len)))
        return set_error(SHEMS_ERR_ARG, "shems_wide_critic_grad_ex: an exclusion window needs a full ring and 0 <= count < capacity");
    hipStream_t st = (hipStream_t)stream;
    const WWs w = wws(d->ws, l1, l2);
    WPrep p{*ring, ring_len, excl_pos, excl_count, seed, tick, d->batch, d->s_min, d->s_max, w};
    hipLaunchKernelGGL(k_wprep, dim3(1), dim3(WBP), 0, st, p);
    // a' = actor_target(s'), q' = critic_target([s'; a'])  (DDPG.jl:131-132)
    if (int rc = net_forward(st, wnet(d->actor_t, WSIN, l1, l2, WAIN), w.XS2, WBP, w.T1, w.T2, w.PT)) return rc;
    hipLaunchKernelGGL(k_wtanh_cat, dim3(1), dim3(WBP), 0, st, w.PT, (float *)nullptr, w.XC2, d->batch);
    if (int rc = net_forward(st, wnet(d->critic_t, WCIN, l1, l2, 1), w.XC2, WBP, w.T1, w.T2, w.Q2)) return rc;
    // q = critic([s; a]); loss_crit = mse(q, y) and its pullback (DDPG.jl:133-135)
    const WNet c = wnet(d->critic, WCIN, l1, l2, 1);
    if (int rc = net_forward(st, c, w.XC, WBP, w.H1c, w.H2c, w.Q)) return rc;
    hipLaunchKernelGGL(k_wloss, dim3(1), dim3(WBP), 0, st, w, d->gamma, d->batch, d->losses);
    return net_backward(st, c, w.XC, w.H1c, w.H2c, w.DQ, d->grad_critic, w.G1, w.G2, nullptr);
}

int shems_wide_actor_grad(const shems_ddpg *d, int32_t l1, int32_t l2, void *stream)
{
    if (int rc = check_wide(d, l1, l2, "shems_wide_actor_grad")) return rc;
    hipStream_t st = (hipStream_t)stream;
    const WWs w = wws(d->ws, l1, l2);
    // loss_act = -mean(critic([s; actor(s)])) through the critic as it stands now (already updated, DDPG.jl:137-140)
    const WNet a = wnet(d->actor, WSIN, l1, l2, WAIN), c = wnet(d->critic, WCIN, l1, l2, 1);
    if (int rc = net_forward(st, a, w.XS, WBP, w.H1a, w.H2a, w.PA)) return rc;
    hipLaunchKernelGGL(k_wtanh_cat, dim3(1), dim3(WBP), 0, st, w.PA, w.API, w.XQ, d->batch);
    if (int rc = net_forward(st, c, w.XQ, WBP, w.H1q, w.H2q, w.Q)) return rc;
    if (int rc = net_backward(st, c, w.XQ, w.H1q, w.H2q, w.DQA, nullptr, w.G1, w.G2, w.DA)) return rc;
    hipLaunchKernelGGL(k_wactor_head, dim3(1), dim3(WBP), 0, st, w, d->batch, d->losses);
    return net_backward(st, a, w.XS, w.H1a, w.H2a, w.D3, d->grad_actor, w.G1, w.G2, nullptr);
}

int shems_wide_critic_apply(const shems_ddpg *d, int32_t l1, int32_t l2, double eta, double bp1, double bp2, double grad_scale, void *stream)
{
    if (int rc = check_wide(d, l1, l2, "shems_wide_critic_apply")) return rc;
    return adam_soft_sweep(d->critic, d->grad_critic, d->m_critic, d->v_critic, d->critic_t, nullptr, (int)wnet_size(WCIN, l1, l2, 1), eta, bp1,
                           bp2, grad_scale, d->tau, (hipStream_t)stream);
}

int shems_wide_actor_apply_pub(const shems_ddpg *d, int32_t l1, int32_t l2, double eta, double bp1, double bp2, double grad_scale,
                               float *d_publish, void *stream)
{
    if (int rc = check_wide(d, l1, l2, "shems_wide_actor_apply_pub")) return rc;
    return adam_soft_sweep(d->actor, d->grad_actor, d->m_actor, d->v_actor, d->actor_t, d_publish, (int)wnet_size(WSIN, l1, l2, WAIN), eta, bp1,
                           bp2, grad_scale, d->tau, (hipStream_t)stream);
}

/* the s rows of the minibatch the last shems_wide_critic_grad_ex sampled (adapt_param_noise!, DDPG.jl:74-87) */
int shems_wide_batch_slots(const shems_ddpg *d, int32_t l1, int32_t l2, int32_t *out_slots, void *stream)
{
    if (int rc = check_wide(d, l1, l2, "shems_wide_batch_slots")) return rc;
    if (!out_slots) return set_error(SHEMS_ERR_ARG, "shems_wide_batch_slots: NULL");
    const WWs w = wws(d->ws, l1, l2);
    if (int rc = hip_ok(hipMemcpyAsync(out_slots, w.IDX, sizeof(int32_t) * d->batch, hipMemcpyDeviceToHost, (hipStream_t)stream), "memcpy slots")) return rc;
    return hip_ok(hipStreamSynchronize((hipStream_t)stream), "sync");
}

}  // extern "C"
