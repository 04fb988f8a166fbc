// The ADAM-state stream of the grouped update's W2-gradient phase (k_tp_gw2, csrc/shems_gupd.hip) WITHOUT its matrix work, as a
// function of how the four arrays (moments m, v, parameter p, target t) of a learner's 250 x 500 layer-2 matrix are laid out in the
// learner's slab.  One workgroup per 64 x 64 tile, 16 float4 per lane requested up front (the kept form of round 5), the real
// adam_math on every element, the tile written back in place.  Per launch: learners x 125 000 x 32 B.
//
//   layout 0  row-major [250][500] (Flux order), 256-byte row pieces                      <- what round 5 ships
//   layout 1  tile-major, one 16 KB piece per array and tile (arrays separate)
//   layout 2  tile-major, the four arrays' tiles adjacent (one 64 KB block per tile)
//   layout 3  m, v tile-major; p, t row-major
//   layout 4  m, v, t tile-major; p row-major
//   layout 5  element-interleaved {m, v, p, t} (one float4 per element), tile-major: one 64 KB read-modify-write stream per tile
//   copy      device copy of the same bytes (read L x 125 000 x 16 B, write as many)
//
// Build + run (from the repo root, through gpurun):
//   hipcc -O3 --offload-arch=gfx950 -I include -I <package>/csrc -o abl/adam_stream tools/micro/adam_stream.hip && abl/adam_stream [learners]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "shems_adam.h"

using namespace shems;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 250, N = 500, KT = 4, NT = 8, TILE = 64 * 64;
constexpr int64_t NET = 131072;            // floats reserved per array (tile-major: 32 tiles x 4 096)

struct Args {
    float *base;           // learner 0's slab
    int64_t stride;        // floats between learners
    int64_t om, ov, op, ot;// offsets of the four arrays inside a slab (layout 2 / 5: om = the block's base)
    AdamCtx c;
};

template <int LAYOUT, bool NTMP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_stream(Args A)
{
    extern __shared__ float smem[];
    const int tid = threadIdx.x, nt = blockIdx.x & 7, kt = blockIdx.x >> 3, tile = kt * NT + nt;
    float *S = A.base + (int64_t)blockIdx.y * A.stride;
    f32x4 am[4], av[4], ap[4], at[4];
    int64_t im[4], iv[4], ip[4], itg[4];
    bool ok[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int kl = 16 * it + (tid >> 4), nl = 4 * (tid & 15), k = 64 * kt + kl, n = 64 * nt + nl;
        ok[it] = k < K && n < N;
        const int64_t rm = (int64_t)min(k, K - 1) * N + min(n, N - 4);           // row-major element
        const int64_t tm = (int64_t)tile * TILE + kl * 64 + nl;                   // tile-major element
        if (LAYOUT == 0) { im[it] = A.om + rm; iv[it] = A.ov + rm; ip[it] = A.op + rm; itg[it] = A.ot + rm; }
        else if (LAYOUT == 1) { im[it] = A.om + tm; iv[it] = A.ov + tm; ip[it] = A.op + tm; itg[it] = A.ot + tm; }
        else if (LAYOUT == 2) {
            const int64_t b = A.om + (int64_t)tile * 4 * TILE + kl * 64 + nl;
            im[it] = b; iv[it] = b + TILE; ip[it] = b + 2 * TILE; itg[it] = b + 3 * TILE;
        } else if (LAYOUT == 3) { im[it] = A.om + tm; iv[it] = A.ov + tm; ip[it] = A.op + rm; itg[it] = A.ot + rm; }
        else if (LAYOUT == 4) { im[it] = A.om + tm; iv[it] = A.ov + tm; ip[it] = A.op + rm; itg[it] = A.ot + tm; }
        else {                                                                     // 5: four coalesced float4 loads, one element {m, v, p, t} each
            const int64_t b = A.om + ((int64_t)tile * TILE + it * 1024 + tid) * 4;         // element e = 1 024 it + 256 j + tid of the tile
            im[it] = b; iv[it] = b + 256 * 4; ip[it] = b + 512 * 4; itg[it] = b + 768 * 4;
            const int e3 = it * 1024 + 768 + tid;
            ok[it] = 64 * kt + (e3 >> 6) < K + 6 && true;                                   // (micro: rows only; the tail columns are streamed)
        }
        if (NTMP) {
            am[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(S + im[it])); av[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(S + iv[it]));
            ap[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(S + ip[it])); at[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(S + itg[it]));
        } else {
            am[it] = *reinterpret_cast<const f32x4 *>(S + im[it]); av[it] = *reinterpret_cast<const f32x4 *>(S + iv[it]);
            ap[it] = *reinterpret_cast<const f32x4 *>(S + ip[it]); at[it] = *reinterpret_cast<const f32x4 *>(S + itg[it]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (smem[tid] == 123.0f) return;             // (keeps the dynamic LDS allocation: it sets the workgroups per CU)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (LAYOUT == 5) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 &e = i == 0 ? am[it] : i == 1 ? av[it] : i == 2 ? ap[it] : at[it];
                const float g = 1e-3f * (float)((tid * 4 + i) & 31) - 0.015f;
                float m_ = e[0], v_ = e[1], p_ = e[2], t_ = e[3];
                adam_math(A.c, g, m_, v_, p_, t_);
                e[0] = m_; e[1] = v_; e[2] = p_; e[3] = t_;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float g = 1e-3f * (float)((tid * 4 + i) & 31) - 0.015f;
                float m_ = am[it][i], v_ = av[it][i], p_ = ap[it][i], t_ = at[it][i];
                adam_math(A.c, g, m_, v_, p_, t_);
                am[it][i] = m_; av[it][i] = v_; ap[it][i] = p_; at[it][i] = t_;
            }
        }
        if (ok[it] && NTMP) {
            __builtin_nontemporal_store(am[it], reinterpret_cast<f32x4 *>(S + im[it])); __builtin_nontemporal_store(av[it], reinterpret_cast<f32x4 *>(S + iv[it]));
            __builtin_nontemporal_store(ap[it], reinterpret_cast<f32x4 *>(S + ip[it])); __builtin_nontemporal_store(at[it], reinterpret_cast<f32x4 *>(S + itg[it]));
        } else if (ok[it]) {
            *reinterpret_cast<f32x4 *>(S + im[it]) = am[it]; *reinterpret_cast<f32x4 *>(S + iv[it]) = av[it];
            *reinterpret_cast<f32x4 *>(S + ip[it]) = ap[it]; *reinterpret_cast<f32x4 *>(S + itg[it]) = at[it];
        }
    }
}

// copy with 16 float4 per thread requested before the first store (64 KB per workgroup in flight, as the stream kernels)
__global__ __launch_bounds__(256) void k_copy(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, int64_t n4)
{
    for (int64_t b = (int64_t)blockIdx.x * 4096; b + 4096 <= n4; b += (int64_t)gridDim.x * 4096) {
        f32x4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = src[b + j * 256 + threadIdx.x];
#pragma unroll
        for (int j = 0; j < 16; ++j) dst[b + j * 256 + threadIdx.x] = v[j];
    }
}

template <int LAYOUT, bool NTMP = false>
static void run(const char *what, Args A, int L, int lds)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stream<LAYOUT, NTMP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_stream<LAYOUT, NTMP>), dim3(KT * NT, L), dim3(256), lds, 0, A);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_stream<LAYOUT, NTMP>), dim3(KT * NT, L), dim3(256), lds, 0, A);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, bytes = (double)L * K * N * 32.0;
    printf("layout %d  %-58s lds %6d  %7.1f us  %5.2f TB/s\n", LAYOUT, what, lds, us, bytes / us * 1e-6);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int L = argc > 1 ? atoi(argv[1]) : 400;
    // a learner's slab as group.py carves it: ~3.7 M floats; the arrays of one network at the offsets of actor / actor_t / m_actor / v_actor
    const int64_t stride = 3700000 + 4 * NET;          // (room for the padded / interleaved forms behind the real carve)
    float *base;
    if (hipMalloc(&base, (size_t)L * stride * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(base, 0, (size_t)L * stride * 4);
    AdamCtx c{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 9, 1e-4, 0.9, 0.999, 1.0, 1e-4 / (1.0 - 0.9), 1.0 / (1.0 - 0.999), 1e-3f};
    const int64_t W2 = 2500;                             // off_w2(9)
    // Flux-order carve: actor 0, critic 129 004, actor_t 258 008, critic_t 387 012, m_actor 516 016, v_actor 645 020 (floats, padded to 4)
    Args rm{base, stride, 516016 + W2, 645020 + W2, 0 + W2, 258008 + W2, c};
    Args tm{base, stride, 3700000, 3700000 + NET, 3700000 + 2 * NET, 3700000 + 3 * NET, c};
    Args mix3{base, stride, 3700000, 3700000 + NET, 0 + W2, 258008 + W2, c};
    Args mix4{base, stride, 3700000, 3700000 + NET, 0 + W2, 3700000 + 3 * NET, c};
    Args blk{base, stride, 3700000, 0, 0, 0, c};
    printf("ADAM-state stream of one network's W2, %d learners: %.2f GB per launch (read + write)\n", L, (double)L * K * N * 32.0 * 1e-9);
    for (int lds : {42 * 1024, 32 * 1024, 20 * 1024}) {        // 3 / 5 / 8 workgroups per CU (160 KB of LDS; registers allow 4 waves per SIMD = 4 workgroups)
        run<0>("row-major, 256-byte pieces (round 5)", rm, L, lds);
        run<1>("tile-major, arrays separate (16 KB pieces)", tm, L, lds);
        run<2>("tile-major, one 64 KB block per tile", blk, L, lds);
        run<3>("m, v tile-major; p, t row-major", mix3, L, lds);
        run<4>("m, v, t tile-major; p row-major", mix4, L, lds);
        run<5>("element-interleaved {m,v,p,t}, tile-major", blk, L, lds);
        run<2, true>("tile-major 64 KB blocks, non-temporal loads + stores", blk, L, lds);
        run<0, true>("row-major, non-temporal loads + stores", rm, L, lds);
    }
    // device copy of the same bytes
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int64_t n4 = (int64_t)L * K * N;           // x 16 B read, x 16 B written
        f32x4 *src = reinterpret_cast<f32x4 *>(base), *dst = src + n4;
        if ((size_t)2 * n4 * 16 <= (size_t)L * stride * 4) {
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, src, dst, n4);
            hipEventRecord(e0, 0);
            for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, src, dst, n4);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("copy      %-58s             %7.1f us  %5.2f TB/s\n", "grid-stride float4 copy of the same bytes", ms * 100.0, (double)n4 * 32.0 / (ms * 100.0) * 1e-6);
        }
    }
    hipFree(base);
    return 0;
}
