// micro-benchmark: v_mfma_f32_32x32x2_f32 issued in program order (inline asm) with NACC accumulators used round robin:
// NACC = 1 is a fully dependent chain (every MFMA's C is the previous D), NACC = 2 / 4 / 16 leave 1 / 3 / 15 independent MFMAs between
// an accumulator's consecutive uses.  Answers: what does the matrix pipe charge for back-to-back dependent MFMAs of this shape?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *stamps, int iters)
{
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float af = 0.001f * threadIdx.x, bf = 0.002f * threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u)
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[u % NACC]) : "v"(af), "v"(bf));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}
template <int NACC> void run(int grid)
{
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, sizeof(float) * 256 * 4096); (void)hipMalloc(&st, 16);
    const int iters = 400;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, st, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[2]; (void)hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * 64;
    printf("accumulators %2d  grid %4d: %.2f cycles/MFMA, %.2f ns/MFMA, clock %.3f GHz\n", NACC, grid, h[0] / n, h[1] * 10.0 / n, h[0] / (h[1] * 10.0));
    (void)hipFree(out); (void)hipFree(st);
}
int main()
{
    for (int grid : {1, 256}) { run<1>(grid); run<2>(grid); run<4>(grid); run<16>(grid); }
    return 0;
}
