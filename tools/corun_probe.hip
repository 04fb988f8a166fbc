// Round 4: does the DDPG update slow down beside ANY MFMA-bound kernel, or only beside the (large, straight-line) fused step kernel?
// A co-runner with a tiny rolled loop: 256 workgroups x 4 waves, 90 KB of LDS each (one per CU, leaving room for a 63.5-KB update workgroup),
// every wave issuing dependent-free v_mfma_f32_32x32x2_f32 back to back for `iters` iterations, optionally with LDS operand reads.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o libcorun.so tools/corun_probe.hip ; driven by tools/corun_probe.py
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_corun(float *out, int iters, int use_lds)
{
    extern __shared__ float lds[];
    const int t = threadIdx.x;
    for (int i = t; i < 4096; i += 256) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    float x = lds[t], y = lds[t + 256];
    for (int it = 0; it < iters; ++it) {
        if (use_lds) { x = lds[(t + 8 * it) & 4095]; y = lds[(t + 8 * it + 1024) & 4095]; }
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.0f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.678f) out[t] = s;
}
extern "C" int corun_launch(void *stream, float *out, int iters, int use_lds, int lds_bytes, int grid)
{
    static bool opt = false;
    if (!opt) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_corun), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1; opt = true; }
    hipLaunchKernelGGL(k_corun, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, out, iters, use_lds);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
