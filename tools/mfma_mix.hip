// micro-benchmark: what do LDS operand reads and LDS-DMA pieces cost a wave that issues v_mfma_f32_32x32x2_f32 back to back?
// One workgroup of 4 waves per CU (1 wave per SIMD, as k_act), 4 accumulators per wave (the TM = 1 tile: 4 MFMAs = 256 cycles per
// k-step), instruction order pinned with inline asm.  Variants per k-step:
//   0  4 MFMA                                   1  + ds_read_b128 + ds_read_b32 (results waited for one k-step later)
//   2  + 2 x ds_read_b32                        3  as 1 + 2 x global_load_lds_dwordx4 (1 KiB each, L2-resident source)
//   4  4 MFMA + 2 x global_load_lds_dwordx4     5  as 1, reads never waited for
//   6  4 MFMA + 4 x global_load_lds_dword (256 B each, one per MFMA gap)
//   7  4 MFMA + 1 x global_load_dwordx4 into registers (the A operand straight from L2, used 8 k-steps later) + ds_read_b32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
template <int V>
__global__ __launch_bounds__(256) void k(float *out, const float *src, unsigned long long *stamps, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 24576; i += 256) lds[i] = 0.001f * (i & 63);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f32x4 av = {0.1f, 0.2f, 0.3f, 0.4f};
    f32x4 ring[8];
    for (int q = 0; q < 8; ++q) ring[q] = av;
    float bv = 0.5f;
    const float *pa = lds + wave * 4096 + lane * 4;              // 1 KiB contiguous per wave instruction
    const float *pb = lds + 16384 + lane;
    const char *g = reinterpret_cast<const char *>(src) + wave * 512 + (lane >> 5) * 2000 + (lane & 31) * 16;
    char *dma = reinterpret_cast<char *>(lds + 20480) + wave * 2048;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            f32x4 an = av;
            float bn = bv;
            if (V == 1 || V == 3 || V == 5) {
                if (V != 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                MFMA(acc[0], av[0], bv);
                asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(an) : "v"((unsigned)(size_t)(pa + ks * 256)) : "memory");
                MFMA(acc[1], av[1], bv);
                asm volatile("ds_read_b32 %0, %1 offset:0" : "=v"(bn) : "v"((unsigned)(size_t)(pb + ks * 64)) : "memory");
            } else if (V == 2) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                MFMA(acc[0], av[0], bv);
                asm volatile("ds_read_b32 %0, %1 offset:0" : "=v"(an[0]) : "v"((unsigned)(size_t)(pb + ks * 64 + 1024)) : "memory");
                MFMA(acc[1], av[1], bv);
                asm volatile("ds_read_b32 %0, %1 offset:0" : "=v"(bn) : "v"((unsigned)(size_t)(pb + ks * 64)) : "memory");
            } else {
                MFMA(acc[0], av[0], bv);
                MFMA(acc[1], av[1], bv);
            }
            if (V == 6) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (size_t)((it * 16 + 2 * ks) & 127) * 4000 + q * 4),
                                                     (__attribute__((address_space(3))) void *)(dma + 256 * q), 4, 0, 0);
            }
            if (V == 7) {
                ring[ks] = *reinterpret_cast<const f32x4 *>(g + (size_t)((it * 8 + ks) & 127) * 4000);
                asm volatile("ds_read_b32 %0, %1 offset:0" : "=v"(bn) : "v"((unsigned)(size_t)(pb + ks * 64)) : "memory");
            }
            MFMA(acc[2], av[2], bv);
            if (V == 6)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (size_t)((it * 16 + 2 * ks) & 127) * 4000 + 8),
                                                 (__attribute__((address_space(3))) void *)(dma + 512), 4, 0, 0);
            if (V == 3 || V == 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (size_t)((it * 16 + 2 * ks) & 127) * 4000),
                                                 (__attribute__((address_space(3))) void *)dma, 16, 0, 0);
            MFMA(acc[3], av[3], bv);
            if (V == 3 || V == 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (size_t)((it * 16 + 2 * ks + 1) & 127) * 4000),
                                                 (__attribute__((address_space(3))) void *)(dma + 1024), 16, 0, 0);
            if (V == 6)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (size_t)((it * 16 + 2 * ks) & 127) * 4000 + 12),
                                                 (__attribute__((address_space(3))) void *)(dma + 768), 4, 0, 0);
            if (V == 7) { an = ring[(ks + 1) & 7]; }
            if (V == 5) { asm volatile("" : "+v"(an), "+v"(bn)); }
            else { av = an; bv = bn; }
        }
        if (V == 3 || V == 4 || V == 6) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = av[0] + bv;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}
template <int V> void run(const char *name, int grid, float *out, float *src, unsigned long long *st)
{
    const int iters = 200;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 24576 * 4 + 8192 + 1024);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 24576 * 4 + 8192 + 1024, 0, out, src, st, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[2];
    (void)hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8;
    printf("%-44s grid %4d: %7.1f cycles per k-step (4 MFMA = 256), clock %.3f GHz\n", name, grid, h[0] / n, h[0] / (h[1] * 10.0));
}
int main()
{
    float *out, *src;
    unsigned long long *st;
    (void)hipMalloc(&out, sizeof(float) * 256 * 4096);
    (void)hipMalloc(&src, 1 << 20);
    (void)hipMemset(src, 0, 1 << 20);
    (void)hipMalloc(&st, 16);
    for (int grid : {1, 256}) {
        run<0>("4 MFMA", grid, out, src, st);
        run<1>("+ ds_read_b128 + ds_read_b32, waited", grid, out, src, st);
        run<5>("+ ds_read_b128 + ds_read_b32, never waited", grid, out, src, st);
        run<2>("+ 2 ds_read_b32, waited", grid, out, src, st);
        run<4>("+ 2 global_load_lds_dwordx4", grid, out, src, st);
        run<3>("+ both reads + 2 global_load_lds_dwordx4", grid, out, src, st);
        run<6>("+ 4 global_load_lds_dword (1 per MFMA gap)", grid, out, src, st);
        run<7>("+ global_load_dwordx4 -> regs (8 deep) + ds_read_b32", grid, out, src, st);
    }
    return 0;
}
