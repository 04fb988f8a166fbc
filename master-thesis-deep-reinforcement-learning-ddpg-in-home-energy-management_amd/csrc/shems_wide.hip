// shems_wide.hip -- networks LARGER than the (250, 500) the tile maps of shems_policy.hip / shems_ddpg.hip are written for.
//
// The reference's hyper-parameter grids hold one such point: (L1, L2) = (300, 600) (input09_08_on_01-09_eval.jl:62-66 digit 3 = 0,
// input.jl:58-66).  Smaller networks run on the tuned kernels by zero padding (ddpg.pad_net); a larger one cannot, so it runs
// here, layer by layer, exactly as the reference's Flux / CUBLAS path does (Dense = W*x .+ b, Zygote's pullbacks: DDPG.jl:21-46,
// 99-145): every layer, forward or backward, is ONE general matrix product with a fused bias / relu / relu' epilogue.
//
// All on v_mfma_f32_32x32x2_f32 with fp32 accumulation in fixed orders (reproducible bit for bit; tolerance-class against the oracle like
// the tuned kernels, tests/test_wide_gpu.py); both operands by (row, column) element strides, so W (Flux layout [in][out]), its
// transpose, and the sample-major activations [m][features] go in without a copy.
//   k_wgemm       64 x 64 tile per workgroup of 4 waves, K in stages of 16 through double-buffered LDS: the vector step's layer 1.
//   k_wgemm128    128 x 128 tile, each wave a 64 x 64 quarter as 2 x 2 MFMA blocks: the vector step's layer 2 -- with the output layer
//                 folded into its epilogue, so relu(layer 2) never leaves the registers.
//   k_wgemm_sk    32 x 32 tile, K stage split over the four waves, up to four independent products per launch: everything in replay(),
//                 where one of M, N, K is the 128-row minibatch.
//   the rest      elementwise kernels: normalize, minibatch sample + gather (the same Philox sampler as the tuned path: the same
//                 (seed, tick) draws the same slots), tanh + concatenation, the two loss heads.
// ADAM + soft target update are shems_ddpg.hip's sweep (adam_soft_sweep: the same arithmetic on any parameter count).
//
// This path is about running the grid point, not about the roofline: sampler + 24 launches per replay() (0.22 ms at (300, 600)), a
// vector step of 65 536 envs 0.28 ms (87 TFLOP/s).  The headline configuration never comes here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <initializer_list>

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
    // k_wgemm128<true> only: the output layer folded into this product's epilogue.  With C = relu(A B + bias) never stored,
    // head_out[(2 t + h) * M + i][o] = sum over the 64 columns j of half h of column tile t of C[i][j] * head_w[j * head_n + o]
    const float *head_w;
    float *head_out;
    int head_n;
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

// The vector step's layer 2 (M = tens of thousands of envs, N = l2, K = l1: 98 % of its FLOPs): 128 x 128 tile, each wave a 64 x 64
// quarter as 2 x 2 MFMA blocks, so a k-step's four operand reads feed four MFMAs (the 64 x 64 kernel above: two reads per MFMA).
// HEAD: the output layer (l2 -> 2) is folded into the epilogue -- every wave reduces its 64 columns of relu(C) against W3 and leaves one
// partial per (row, output); relu(layer 2), 157 MB at 65 536 envs x 600, is never written or read back, and the N = 2 product, which
// a 64-wide tile pads 32-fold, disappears.  k_act_tail adds the partials in index order.
constexpr int BT = 128, BLD = BT + 4, BR = BT * GK / 256;
// V4: both operands are contiguous along the index that runs across a stage (A along k, B along j), 16-byte aligned, with row strides, K
// and N multiples of 4 (the host checks): a stage is staged with two 16-byte loads per operand and thread instead of eight predicated
// 4-byte ones -- a quarter of the load / address / bounds instructions next to the MFMAs.
typedef float wf32x4 __attribute__((ext_vector_type(4)));
template <bool HEAD, bool V4>
__global__ __launch_bounds__(256) void k_wgemm128(GemmArgs G)
{
    __shared__ __attribute__((aligned(16))) float As[2][GK][BLD], Bs[2][GK][BLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.x * BT, n0 = (int64_t)blockIdx.y * BT;
    const bool a_kfast = G.sak == 1, b_jfast = G.sbj == 1;
    int ai[BR], ak[BR], bj[BR], bk[BR];
#pragma unroll
    for (int r = 0; r < BR; ++r) {
        const int e = tid + 256 * r;
        ai[r] = a_kfast ? e >> GKB : e & (BT - 1);  ak[r] = a_kfast ? e & (GK - 1) : e >> 7;
        bj[r] = b_jfast ? e & (BT - 1) : e >> GKB;  bk[r] = b_jfast ? e >> 7 : e & (GK - 1);
    }
    float ra[BR], rb[BR];
    wf32x4 va[2], vb[2];
    auto fetch = [&](int k0) {
        if constexpr (V4) {
            // quad f = tid + 256 r: A row f >> 2, k = 4 (f & 3) ..; B row k = f >> 5, j = 4 (f & 31) ..
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int f = tid + 256 * r;
                const int64_t i = m0 + (f >> 2), j = n0 + 4 * (f & 31);
                const int ka = k0 + 4 * (f & 3), kb = k0 + (f >> 5);
                const wf32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
                va[r] = (i < G.M && ka < G.K) ? *reinterpret_cast<const wf32x4 *>(G.A + i * G.sai + ka) : z;
                vb[r] = (j < G.N && kb < G.K) ? *reinterpret_cast<const wf32x4 *>(G.B + kb * G.sbk + j) : z;
            }
        } else {
#pragma unroll
            for (int r = 0; r < BR; ++r) {
                const int64_t i = m0 + ai[r], j = n0 + bj[r];
                const int ka = k0 + ak[r], kb = k0 + bk[r];
                ra[r] = (i < G.M && ka < G.K) ? G.A[i * G.sai + ka * G.sak] : 0.0f;
                rb[r] = (j < G.N && kb < G.K) ? G.B[kb * G.sbk + j * G.sbj] : 0.0f;
            }
        }
    };
    auto stash = [&](int buf) {
        if constexpr (V4) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int f = tid + 256 * r;
#pragma unroll
                for (int c = 0; c < 4; ++c) As[buf][4 * (f & 3) + c][f >> 2] = va[r][c];
                *reinterpret_cast<wf32x4 *>(&Bs[buf][f >> 5][4 * (f & 31)]) = vb[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < BR; ++r) { As[buf][ak[r]][ai[r]] = ra[r]; Bs[buf][bk[r]][bj[r]] = rb[r]; }
        }
    };
    wf32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;
    fetch(0);
    stash(0);
    __syncthreads();
    const int nst = (G.K + GK - 1) / GK;
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        if (s + 1 < nst) fetch((s + 1) * GK);
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2) {
            const float a0 = As[buf][kk + lh][wi * 64 + li], a1 = As[buf][kk + lh][wi * 64 + 32 + li];
            const float b0 = Bs[buf][kk + lh][wj * 64 + li], b1 = Bs[buf][kk + lh][wj * 64 + 32 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (s + 1 < nst) stash(buf ^ 1);
        __syncthreads();
    }
    if constexpr (!HEAD) {
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int64_t j = n0 + wj * 64 + y * 32 + li;
            if (j >= G.N) continue;
            const float bj_ = G.bias ? G.bias[j] : 0.0f;
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t i = m0 + wi * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (i < G.M) {
                        float v = acc[x][y][r] + bj_;
                        if (G.relu) v = fmaxf(v, 0.0f);
                        if (G.gate) v = G.gate[i * G.ldg + j] > 0.0f ? v : 0.0f;
                        G.C[i * G.ldc + j] = v;
                    }
                }
        }
    } else {
        // this lane's two columns: bias and the head's weights (zero beyond N, so padded columns add nothing)
        float bb[2], w3[2][2];
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int64_t j = n0 + wj * 64 + y * 32 + li;
            const bool in = j < G.N;
            const int64_t jc = in ? j : 0;
            bb[y] = in && G.bias ? G.bias[jc] : 0.0f;
#pragma unroll
            for (int o = 0; o < 2; ++o) w3[y][o] = in && o < G.head_n ? G.head_w[jc * G.head_n + o] : 0.0f;
        }
        float *out = G.head_out + ((int64_t)blockIdx.y * 2 + wj) * G.M * G.head_n;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v0 = fmaxf(acc[x][0][r] + bb[0], 0.0f), v1 = fmaxf(acc[x][1][r] + bb[1], 0.0f);
                float s0 = v0 * w3[0][0] + v1 * w3[1][0], s1 = v0 * w3[0][1] + v1 * w3[1][1];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }     // over the half's 32 lanes
                const int64_t i = m0 + wi * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (li == 0 && i < G.M) {
                    out[i * G.head_n] = s0;
                    if (G.head_n > 1) out[i * G.head_n + 1] = s1;
                }
            }
    }
}

// The minibatch-sized products (M <= a few hundred rows): few tiles and a long K.  One wave's chain of 32x32x2 MFMAs costs 32 cycles
// per k whatever the memory system does (K = 600: 9 us), and a 64 x 64 tile per workgroup leaves most CUs idle (N = 600: 20
// workgroups).  So here a workgroup owns a 32 x 32 tile and its four waves split every 64-deep K stage four ways (16 k each); the
// four partial tiles meet in LDS at the end and are added in wave order (a fixed order).  Same operands, strides and epilogue.
// Up to four INDEPENDENT products ride in one launch (grid z): the update is a chain of small dependent launches, and e.g. the layer-1
// products of three forward passes, or a layer's weight gradient, bias gradient and back-propagated error, need nothing from each other.
constexpr int SK = 64, SKB = 6, SLD = 32 + 4, SR = 32 * SK / 256, GMAX = 4;
struct GemmBatch { GemmArgs g[GMAX]; };
__global__ __launch_bounds__(256) void k_wgemm_sk(GemmBatch B)
{
    const GemmArgs &G = B.g[blockIdx.z];
    if ((int64_t)blockIdx.x * 32 >= G.M || (int64_t)blockIdx.y * 32 >= G.N) return;      // (the grid is the largest problem's)
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

static GemmArgs prod(const float *A, int64_t sai, int64_t sak, const float *B, int64_t sbk, int64_t sbj, float *C, int64_t ldc,
                     int64_t M, int N, int K, const float *bias = nullptr, int relu = 0, const float *gate = nullptr, int64_t ldg = 0)
{
    return GemmArgs{A, B, C, (int)M, N, K, sai, sak, sbk, sbj, ldc, bias, gate, ldg, relu};
}
// independent minibatch-sized products (one of M, N, K is the minibatch) in one launch
static int gemm_multi(hipStream_t st, std::initializer_list<GemmArgs> list)
{
    GemmBatch b;
    std::memset(&b, 0, sizeof b);
    unsigned gx = 1, gy = 1, n = 0;
    for (const GemmArgs &g : list) {
        if (n >= GMAX) return set_error(SHEMS_ERR_ARG, "gemm_multi: at most %d products per launch", GMAX);
        b.g[n++] = g;
        gx = std::max(gx, (unsigned)((g.M + 31) / 32));
        gy = std::max(gy, (unsigned)((g.N + 31) / 32));
    }
    hipLaunchKernelGGL(k_wgemm_sk, dim3(gx, gy, n), dim3(256), 0, st, b);
    return hip_ok(hipGetLastError(), "k_wgemm_sk launch");
}
static int gemm(hipStream_t st, const float *A, int64_t sai, int64_t sak, const float *B, int64_t sbk, int64_t sbj, float *C, int64_t ldc,
                int64_t M, int N, int K, const float *bias = nullptr, int relu = 0, const float *gate = nullptr, int64_t ldg = 0)
{
    GemmArgs g{A, B, C, (int)M, N, K, sai, sak, sbk, sbj, ldc, bias, gate, ldg, relu};
    if (M <= 512) return gemm_multi(st, {g});
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

// The three layer products of a forward pass over WBP rows, as values: independent passes are launched layer by layer together.
struct Fwd { GemmArgs l1, l2, l3; };
static Fwd fwd_of(const WNet &n, const float *X, float *H1, float *H2, float *P)
{
    return Fwd{prod(X, n.in, 1, n.W1, n.l1, 1, H1, n.l1, WBP, n.l1, n.in, n.b1, 1), prod(H1, n.l1, 1, n.W2, n.l2, 1, H2, n.l2, WBP, n.l2, n.l1, n.b2, 1),
               prod(H2, n.l2, 1, n.W3, n.out, 1, P, n.out, WBP, n.out, n.l2, n.b3, 0)};
}

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
    float *ONES;                              // [WBP] ones: a bias gradient sum_m dY[m][n] is the product ones' dY, one more tile in a launch that runs anyway
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
    w.ONES = take(WBP);
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
    w.ONES[m] = 1.0f;
}

// a = tanh(P) for the live rows -> a_out [WBP][2] (may be null) and the action columns of a [WBP][11] critic input; workgroup 0: the
// target actor's head into [s'; a'], workgroup 1: the actor's into a_pi and [s; a_pi]
__global__ __launch_bounds__(WBP) void k_wtanh_cat(const float *__restrict__ P0, float *__restrict__ a0_out, float *__restrict__ cat0,
                                                   const float *__restrict__ P1, float *__restrict__ a1_out, float *__restrict__ cat1, int batch)
{
    const int m = threadIdx.x;
    const float *P = blockIdx.x == 0 ? P0 : P1;
    float *a_out = blockIdx.x == 0 ? a0_out : a1_out, *cat = blockIdx.x == 0 ? cat0 : cat1;
    const float a0 = m < batch ? tanhf(P[2 * m]) : 0.0f, a1 = m < batch ? tanhf(P[2 * m + 1]) : 0.0f;
    if (a_out) { a_out[2 * m] = a0; a_out[2 * m + 1] = a1; }
    cat[m * WCIN + 9] = a0;
    cat[m * WCIN + 10] = a1;
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
// gradients into `grad` (flat Flux layout; null = input gradient only); dX [WBP][in] = d loss / d input if asked for.  X, H1, H2: what
// the forward pass kept.  Three launches: {gW3, gb3, G2}, {gW2, gb2, G1}, {gW1, gb1, dX} -- within one, nothing depends on anything.
static int net_backward(hipStream_t st, const WNet &n, const float *X, const float *H1, const float *H2, const float *d3, float *grad,
                        float *G1, float *G2, float *dX, const float *ones)
{
    const GemmArgs g2 = prod(d3, n.out, 1, n.W3, 1, n.out, G2, n.l2, WBP, n.l2, n.out, nullptr, 0, H2, n.l2);       // (d3 W3') .* relu'
    const GemmArgs g1 = prod(G2, n.l2, 1, n.W2, 1, n.l2, G1, n.l1, WBP, n.l1, n.l2, nullptr, 0, H1, n.l1);          // (G2 W2') .* relu'
    if (!grad) {
        if (int rc = gemm_multi(st, {g2})) return rc;
        if (int rc = gemm_multi(st, {g1})) return rc;
        return dX ? gemm_multi(st, {prod(G1, n.l1, 1, n.W1, 1, n.l1, dX, n.in, WBP, n.in, n.l1)}) : SHEMS_OK;      // G1 W1'
    }
    float *gW1 = grad, *gb1 = gW1 + (int64_t)n.in * n.l1, *gW2 = gb1 + n.l1, *gb2 = gW2 + (int64_t)n.l1 * n.l2, *gW3 = gb2 + n.l2,
          *gb3 = gW3 + (int64_t)n.l2 * n.out;
    auto colsum = [&](const float *D, int cols, float *out) { return prod(ones, WBP, 1, D, cols, 1, out, cols, 1, cols, WBP); };   // ones' D
    if (int rc = gemm_multi(st, {prod(H2, 1, n.l2, d3, n.out, 1, gW3, n.out, n.l2, n.out, WBP), colsum(d3, n.out, gb3), g2})) return rc;       // gW3 = H2' d3
    if (int rc = gemm_multi(st, {prod(H1, 1, n.l1, G2, n.l2, 1, gW2, n.l2, n.l1, n.l2, WBP), colsum(G2, n.l2, gb2), g1})) return rc;           // gW2 = H1' G2
    if (dX)
        return gemm_multi(st, {prod(X, 1, n.in, G1, n.l1, 1, gW1, n.l1, n.in, n.l1, WBP), colsum(G1, n.l1, gb1),                             // gW1 = X' G1
                               prod(G1, n.l1, 1, n.W1, 1, n.l1, dX, n.in, WBP, n.in, n.l1)});
    return gemm_multi(st, {prod(X, 1, n.in, G1, n.l1, 1, gW1, n.l1, n.in, n.l1, WBP), colsum(G1, n.l1, gb1)});
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

// used by shems_policy.hip's wide act entry points: the vector step's forward pass.  Outputs `n_partials` partial pre-activation sums
// [p][m][2] into d_part (b3 NOT included: k_act_tail adds b3 and the partials in index order).
int wide_actor_pre(const float *actor, const float *s_min, const float *s_max, int l1, int l2, const float *d_obs, int64_t m, float *d_ws,
                   float *d_part, int *n_partials, hipStream_t st)
{
    if (int rc = check_shape(l1, l2, "shems_wide_act")) return rc;
    float *xn = d_ws, *H1 = xn + (m * WSIN + 3) / 4 * 4;
    const int64_t cnt = m * WSIN;
    const WNet n = wnet(actor, WSIN, l1, l2, WAIN);
    hipLaunchKernelGGL(k_wnorm, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, d_obs, s_min, s_max, xn, cnt);
    if (int rc = gemm(st, xn, n.in, 1, n.W1, n.l1, 1, H1, n.l1, m, n.l1, n.in, n.b1, 1)) return rc;
    // layer 2 with the output layer in its epilogue: relu(layer 2) stays in registers
    GemmArgs g = prod(H1, n.l1, 1, n.W2, n.l2, 1, nullptr, 0, m, n.l2, n.l1, n.b2, 1);
    g.head_w = n.W3; g.head_out = d_part; g.head_n = WAIN;
    const unsigned ty = (unsigned)((l2 + BT - 1) / BT);
    *n_partials = 2 * (int)ty;
    const bool v4 = (l1 % 4) == 0 && (l2 % 4) == 0 && ((uintptr_t)H1 & 15) == 0 && ((uintptr_t)n.W2 & 15) == 0;     // (300, 600): yes
    if (v4) hipLaunchKernelGGL((k_wgemm128<true, true>), dim3((unsigned)((m + BT - 1) / BT), ty), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((k_wgemm128<true, false>), dim3((unsigned)((m + BT - 1) / BT), ty), dim3(256), 0, st, g);
    return hip_ok(hipGetLastError(), "k_wgemm128 launch");
}
// floats of d_ws: normalised observations, layer 1, and the partial sums of the output layer
int64_t wide_act_part_offset(int l1, int64_t m) { return (m * WSIN + 3) / 4 * 4 + (m * (int64_t)l1 + 3) / 4 * 4; }      // 16-byte aligned
int64_t wide_act_ws_floats(int l1, int l2, int64_t m) { return wide_act_part_offset(l1, m) + 2 * ((l2 + BT - 1) / BT) * m * WAIN; }

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
    // Three forward passes need only the minibatch: a' = actor_target(s') (DDPG.jl:131), q = critic([s; a]) (:134) and a_pi = actor(s)
    // (:138; the actor does not change before shems_wide_actor_apply_pub) -- launched together, layer by layer.
    const WNet c = wnet(d->critic, WCIN, l1, l2, 1);
    const Fwd ft = fwd_of(wnet(d->actor_t, WSIN, l1, l2, WAIN), w.XS2, w.T1, w.T2, w.PT), fc = fwd_of(c, w.XC, w.H1c, w.H2c, w.Q),
              fa = fwd_of(wnet(d->actor, WSIN, l1, l2, WAIN), w.XS, w.H1a, w.H2a, w.PA);
    if (int rc = gemm_multi(st, {ft.l1, fc.l1, fa.l1})) return rc;
    if (int rc = gemm_multi(st, {ft.l2, fc.l2, fa.l2})) return rc;
    if (int rc = gemm_multi(st, {ft.l3, fc.l3, fa.l3})) return rc;
    hipLaunchKernelGGL(k_wtanh_cat, dim3(2), dim3(WBP), 0, st, w.PT, (float *)nullptr, w.XC2, w.PA, w.API, w.XQ, d->batch);
    // q' = critic_target([s'; a'])  (DDPG.jl:132); T1 / T2 are free again
    if (int rc = net_forward(st, wnet(d->critic_t, WCIN, l1, l2, 1), w.XC2, WBP, w.T1, w.T2, w.Q2)) return rc;
    // loss_crit = mse(q, y) and its pullback (DDPG.jl:133-135)
    hipLaunchKernelGGL(k_wloss, dim3(1), dim3(WBP), 0, st, w, d->gamma, d->batch, d->losses);
    return net_backward(st, c, w.XC, w.H1c, w.H2c, w.DQ, d->grad_critic, w.G1, w.G2, nullptr, w.ONES);
}

int shems_wide_actor_grad(const shems_ddpg *d, int32_t l1, int32_t l2, void *stream)
{
    if (int rc = check_wide(d, l1, l2, "shems_wide_actor_grad")) return rc;
    hipStream_t st = (hipStream_t)stream;
    const WWs w = wws(d->ws, l1, l2);
    // loss_act = -mean(critic([s; actor(s)])) through the critic as it stands now (already updated, DDPG.jl:137-140); a_pi = actor(s) and
    // the actor's hidden layers were laid down by shems_wide_critic_grad_ex
    const WNet a = wnet(d->actor, WSIN, l1, l2, WAIN), c = wnet(d->critic, WCIN, l1, l2, 1);
    if (int rc = net_forward(st, c, w.XQ, WBP, w.H1q, w.H2q, w.Q)) return rc;
    if (int rc = net_backward(st, c, w.XQ, w.H1q, w.H2q, w.DQA, nullptr, w.G1, w.G2, w.DA, w.ONES)) return rc;
    hipLaunchKernelGGL(k_wactor_head, dim3(1), dim3(WBP), 0, st, w, d->batch, d->losses);
    return net_backward(st, a, w.XS, w.H1a, w.H2a, w.D3, d->grad_actor, w.G1, w.G2, nullptr, w.ONES);
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
