// What tells a workgroup which of a CU's resident slots it got?  512 workgroups of 256 threads with 78 KB of LDS (two per CU, as k_act2):
// each records HW_REG_LDS_ALLOC, HW_REG_HW_ID and XCC_ID.  hipcc -O3 --offload-arch=gfx950 -o abl/slot_probe tools/micro/slot_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned *out)
{
    extern __shared__ float smem[];
    smem[threadIdx.x] = 1.0f;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6);      // HW_REG_LDS_ALLOC, 32 bits
        out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID
        out[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);     // HW_REG_XCC_ID
        out[blockIdx.x * 4 + 3] = (unsigned)smem[5];
    }
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(127);      // stay resident while the others start
}
int main()
{
    unsigned *d; (void)hipMalloc(&d, 512 * 16);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 78 * 1024);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 78 * 1024, 0, d);
    std::vector<unsigned> h(2048);
    (void)hipMemcpy(h.data(), d, 512 * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, int> bases;
    for (int b = 0; b < 512; ++b) bases[h[b * 4]]++;
    for (auto &kv : bases) printf("LDS_ALLOC %08x : %d workgroups\n", kv.first, kv.second);
    for (int b = 0; b < 24; ++b) printf("wg %3d lds_alloc %08x hw_id %08x (cu %u sh %u se %u simd %u wave %u) xcc %x\n", b, h[b * 4], h[b * 4 + 1], (h[b * 4 + 1] >> 8) & 15,
                                        (h[b * 4 + 1] >> 12) & 1, (h[b * 4 + 1] >> 13) & 7, (h[b * 4 + 1] >> 4) & 3, h[b * 4 + 1] & 15, h[b * 4 + 2]);
    for (int b = 256; b < 272; ++b) printf("wg %3d lds_alloc %08x hw_id %08x (cu %u sh %u se %u simd %u wave %u) xcc %x\n", b, h[b * 4], h[b * 4 + 1], (h[b * 4 + 1] >> 8) & 15,
                                        (h[b * 4 + 1] >> 12) & 1, (h[b * 4 + 1] >> 13) & 7, (h[b * 4 + 1] >> 4) & 3, h[b * 4 + 1] & 15, h[b * 4 + 2]);
    return 0;
}
