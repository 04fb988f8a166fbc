// micro-benchmark: a dedicated loader wave.  Waves 0..3 (one per SIMD) run 4 MFMAs + ds_read_b128 + ds_read_b32 per k-step and issue no
// vector-memory instruction; wave 4 (second wave on SIMD 0) issues the 8 global_load_lds_dwordx4 pieces per k-step that the four of them
// would have issued themselves (2 each).  Prints the compute waves' cycles per k-step (256 = MFMA bound) and the loader's own pace.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
template <int PER>   // DMA pieces the loader issues per k-step
__global__ __launch_bounds__(320) void k(float *out, const float *src, unsigned long long *stamps, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 24576; i += 320) lds[i] = 0.001f * (i & 63);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave == 4) {
        const char *g = reinterpret_cast<const char *>(src) + (lane >> 5) * 2000 + (lane & 31) * 16;
        char *dma = reinterpret_cast<char *>(lds + 20480);
#pragma unroll 1
        for (int it = 0; it < iters * 8; ++it) {
#pragma unroll
            for (int q = 0; q < PER; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (size_t)((it * 8 + q) & 127) * 4000),
                                                 (__attribute__((address_space(3))) void *)(dma + 1024 * q), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0 && blockIdx.x == 0) stamps[1] = t1 - t0;
        return;
    }
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f32x4 av = {0.1f, 0.2f, 0.3f, 0.4f};
    float bv = 0.5f;
    const float *pa = lds + wave * 4096 + lane * 4;
    const float *pb = lds + 16384 + lane;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            f32x4 an;
            float bn;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            MFMA(acc[0], av[0], bv);
            asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(an) : "v"((unsigned)(size_t)(pa + ks * 256)) : "memory");
            MFMA(acc[1], av[1], bv);
            asm volatile("ds_read_b32 %0, %1 offset:0" : "=v"(bn) : "v"((unsigned)(size_t)(pb + ks * 64)) : "memory");
            MFMA(acc[2], av[2], bv);
            MFMA(acc[3], av[3], bv);
            av = an; bv = bn;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = av[0] + bv;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0 && blockIdx.x == 0 && (wave == 0 || wave == 1)) stamps[wave == 0 ? 0 : 2] = t1 - t0;
}
template <int PER> void run(int grid, float *out, float *src, unsigned long long *st)
{
    const int iters = 200;
    const int ldsb = 24576 * 4 + 8192 + 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<PER>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<PER>, dim3(grid), dim3(320), ldsb, 0, out, src, st, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[3];
    (void)hipMemcpy(h, st, 24, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8;
    printf("loader issues %d pieces per k-step, grid %4d: wave 0 (shares its SIMD with the loader) %.1f, wave 1 %.1f cycles per k-step; loader %.1f cycles per k-step\n",
           PER, grid, h[0] / n, h[2] / n, h[1] / n);
}
int main()
{
    float *out, *src;
    unsigned long long *st;
    (void)hipMalloc(&out, sizeof(float) * 256 * 4096);
    (void)hipMalloc(&src, 1 << 20);
    (void)hipMemset(src, 0, 1 << 20);
    (void)hipMalloc(&st, 24);
    for (int grid : {1, 256}) { run<8>(grid, out, src, st); run<4>(grid, out, src, st); }
    return 0;
}
