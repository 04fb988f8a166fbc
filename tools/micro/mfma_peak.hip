// Sustained fp32 MFMA rate of the whole chip, no memory traffic: what fraction of the 157.3 TFLOP/s datasheet figure
// (256 CUs x 4 SIMDs x 2.4 GHz x 64 FLOP/cycle) a kernel made of nothing but v_mfma_f32_32x32x2_f32 reaches for a given launch
// length and waves per SIMD.  Build + run (from the repo root, through gpurun):
//   hipcc -O3 --offload-arch=gfx950 -o abl/mfma_peak tools/micro/mfma_peak.hip && abl/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// shader cycles (s_memtime) per 100-MHz tick (s_memrealtime) seen by workgroup 0's first wave over the whole loop: the clock under this load
__device__ unsigned long long g_clk[2];
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(float *out, int iters, float a, float b, int stagger = 0)
{
    if (stagger) for (int i = 0; i < (int)((blockIdx.x >> 8) & 3) * stagger; ++i) __builtin_amdgcn_s_sleep(16);
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + i);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = __builtin_readcyclecounter() - c0; g_clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

static int g_lds = 0, g_stagger = 0;       // dynamic LDS per workgroup: caps the workgroups a CU can hold (160 KB / g_lds) without touching the kernel
template <int NACC>
static void run(int wgs_per_cu, int iters, float *out, int rounds = 1)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu * rounds;
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), g_lds, 0, out, iters, 1.0f, 0.5f, g_stagger);
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.0f;
    const int reps = 20;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), g_lds, 0, out, iters, 1.0f, 0.5f, g_stagger);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    unsigned long long clk[2];
    (void)hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof clk);
    const double ghz = (double)clk[0] / ((double)clk[1] * 10e-9) * 1e-9;
    const double flop = (double)grid * 4.0 * iters * 4.0 * NACC * 4096.0;
    printf("lds %6d rounds %d accs/wave %d  waves/SIMD %d  iters %6d  avg %.1f us  best %.1f us  %.1f TFLOP/s avg (%.3f of 157.3)  %.1f best (%.3f)  shader clock %.2f GHz\n", g_lds, rounds, NACC, wgs_per_cu, iters,
           sum / reps * 1e3, best * 1e3, flop / (sum / reps * 1e-3) * 1e-12, flop / (sum / reps * 1e-3) * 1e-12 / 157.3, flop / (best * 1e-3) * 1e-12,
           flop / (best * 1e-3) * 1e-12 / 157.3, ghz);
}

int main()
{
    float *out; hipMalloc(&out, 256 * 64 * 256 * sizeof(float));
    // one exact generation of workgroups: 16 accumulator tiles per SIMD in flight however they are split over waves?
    for (int st : {0, 1, 7}) {
        g_stagger = st;
        printf("stagger %d (workgroup generation b / 256 sleeps b * %d * 1024 cycles first)\n", st, st);
        g_lds = 33792; run<4>(4, 400, out); run<2>(4, 800, out); run<4>(3, 400, out);
        g_lds = 70000; run<8>(2, 200, out); run<4>(2, 400, out); run<8>(2, 50, out, 4);
        g_lds = 150000; run<8>(1, 400, out); 
    }
    g_stagger = 0;
    g_lds = 0;
    for (int iters : {100, 1600}) { run<4>(1, iters, out); run<4>(2, iters, out); run<2>(4, iters, out); run<1>(4, iters, out); }
    for (int r = 0; r < 500; ++r) hipLaunchKernelGGL(k_mfma<4>, dim3(1024), dim3(256), 0, 0, out, 3200, 1.0f, 0.5f, 0);
    hipDeviceSynchronize();
    printf("after ~2 s of continuous matrix work:\n");
    run<2>(4, 6400, out);
    return 0;
}
