// Which ingredient of the grouped update's forward kernel (k_tp_fwd, shems_gupd.hip) costs matrix-pipe time?  The same loop shape --
// per chunk: 6 dependent layer-1 products, 16 x NTL layer-2 products on NTL accumulators, one barrier -- with ingredients switched on one
// at a time (bit mask F): 1 = barrier per chunk, 2 = A operands read from LDS (one ds_read per product pair), 4 = B operand through
// v_max (relu) of the layer-1 tile, 8 = next chunk requested from global memory at the top and stored to the LDS ring at the bottom.
//   hipcc -O3 --offload-arch=gfx950 -o abl/fwd_anatomy tools/micro/fwd_anatomy.hip && abl/fwd_anatomy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int F, int NTL>
__global__ __launch_bounds__(256) void k(const float *__restrict__ W, float *out, int chunks)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int S = 32 * NTL;
    float *ring = smem;                      // [2][32][S]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    f32x16 acc[NTL];
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    for (int i = tid; i < 2 * 32 * S; i += 256) ring[i] = 1.0f / (float)(i + 1);
    __syncthreads();
    const float *Wb = W + (size_t)blockIdx.x * 4096;
    float x[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) x[s] = 0.25f * (float)(s + lane);
    f32x4 pv[NTL];
#pragma unroll 1
    for (int c = 0; c < chunks; ++c) {
        const float *buf = ring + (c & 1) * 32 * S;
        if (F & 8) {
#pragma unroll
            for (int it = 0; it < NTL; ++it) pv[it] = *reinterpret_cast<const f32x4 *>(Wb + ((c & 7) * NTL + it) * 1024 + 4 * tid);
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x16 t;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 6; ++s) t = __builtin_amdgcn_mfma_f32_32x32x2f32((F & 2) ? buf[s * 32 + li] : x[s], x[s], t, 0, 0, 0);
        const float *pa = buf + 4 * lh * S + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float b = (F & 4) ? fmaxf(t[r], 0.0f) : t[r];
#pragma unroll
            for (int tt = 0; tt < NTL; ++tt)
                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32((F & 2) ? pa[((r & 3) + 8 * (r >> 2)) * S + 32 * tt] : x[tt], b, acc[tt], 0, 0, 0);
        }
        if (F & 8) {
#pragma unroll
            for (int it = 0; it < NTL; ++it) *reinterpret_cast<f32x4 *>(ring + ((c + 1) & 1) * 32 * S + it * 1024 + 4 * tid) = pv[it];
        }
        if (F & 1) __syncthreads();
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
}

template <int F, int NTL>
static void run(int wgs, int chunks, int lds_extra, const float *W, float *out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int lds = 2 * 32 * 32 * NTL * 4 + lds_extra;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<F, NTL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k<F, NTL>), dim3(wgs), dim3(256), lds, 0, W, out, chunks);
    (void)hipDeviceSynchronize();
    float sum = 0.0f;
    const int reps = 10;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<F, NTL>), dim3(wgs), dim3(256), lds, 0, W, out, chunks);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        sum += ms;
    }
    const double flop = (double)wgs * 4.0 * chunks * (6.0 + 16.0 * NTL) * 4096.0;
    const double tf = flop / (sum / reps * 1e-3) * 1e-12;
    printf("F=%2d NTL=%d wgs %5d chunks %3d lds %6d: %8.1f us  %.1f TFLOP/s (%.3f of 157.3, all products counted)\n", F, NTL, wgs, chunks, lds, sum / reps * 1e3, tf, tf / 157.3);
}

int main()
{
    float *W, *out;
    (void)hipMalloc(&W, (size_t)8192 * 4096 * 4);
    (void)hipMemset(W, 0, (size_t)8192 * 4096 * 4);
    (void)hipMalloc(&out, 8192 * 256 * 4);
    // the shapes of P1 (NTL 4: 4 800 workgroups, 8 chunks, 46 KB -> 3 per CU) and P2 (NTL 2: 3 200 workgroups, 30 KB -> 4 per CU)
    const int x4 = 47104 - 2 * 32 * 128 * 4, x2 = 30464 - 2 * 32 * 64 * 4;
    run<0, 4>(4800, 8, x4, W, out); run<1, 4>(4800, 8, x4, W, out); run<3, 4>(4800, 8, x4, W, out); run<7, 4>(4800, 8, x4, W, out); run<15, 4>(4800, 8, x4, W, out);
    run<0, 2>(3200, 8, x2, W, out); run<1, 2>(3200, 8, x2, W, out); run<3, 2>(3200, 8, x2, W, out); run<7, 2>(3200, 8, x2, W, out); run<15, 2>(3200, 8, x2, W, out);
    // the wide form with the LDS of four (then two) workgroups per CU: what residency is worth
    run<15, 4>(4800, 8, 2048, W, out); run<0, 4>(4800, 8, 2048, W, out); run<15, 4>(4800, 8, 47104, W, out); run<15, 2>(3200, 8, 20000, W, out); run<15, 2>(3200, 8, 40000, W, out);
    // the same work in fewer, longer workgroups (64 chunks): what the launch shape costs
    run<15, 4>(768, 50, x4, W, out); run<15, 2>(1024, 25, x2, W, out);
    run<0, 4>(768, 50, x4, W, out); run<0, 2>(1024, 25, x2, W, out);
    return 0;
}
