// micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 with 16 independent accumulators per wave, 4 waves per WG (1 per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *stamps, int iters)
{
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 0.001f * (i & 63);
    __syncthreads();
    f32x16 acc[16];
    for (int a = 0; a < 16; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float af[4], bf[4];
    for (int a = 0; a < 4; ++a) { af[a] = lds[threadIdx.x + 64 * a]; bf[a] = lds[threadIdx.x + 999 + 64 * a]; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (MODE >= 1) {
#pragma unroll
                for (int a = 0; a < 4; ++a) { af[a] = lds[((it * 8 + ks) * 512 + threadIdx.x + 64 * a) & 16383]; bf[a] = lds[((it * 8 + ks) * 128 + threadIdx.x + 32 * a + 7) & 16383]; }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a * 4 + b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a * 4 + b], 0, 0, 0);
        }
        if (MODE >= 2) __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a = 0; a < 16; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}
template <int MODE> void run(const char *name, int grid)
{
    float *out; unsigned long long *st;
    hipMalloc(&out, sizeof(float) * 256 * 4096); hipMalloc(&st, 16);
    const int iters = 200;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, st, iters);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * 128;
    printf("%-28s grid %4d: %.2f memtime-cycles/MFMA, %.2f ns/MFMA, ratio %.3f GHz\n", name, grid, h[0] / n, h[1] * 10.0 / n, h[0] / (h[1] * 10.0));
    hipFree(out); hipFree(st);
}
int main()
{
    for (int grid : {1, 256, 512}) {
        run<0>("mfma only", grid);
        run<1>("mfma + 8 ds_read/kstep", grid);
        run<2>("mfma + ds_read + barrier/8ks", grid);
    }
    return 0;
}
