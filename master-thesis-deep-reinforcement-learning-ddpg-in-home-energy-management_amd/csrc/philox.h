// philox.h -- Philox4x32-10 counter-based generator (Salmon et al., SC'11), used for every random
// draw the batched path needs: episode starts (reference: MersenneTwister(rng), shems_LU1.jl:224-225),
// random pre-fill actions (memory_plotting_saving.jl:17), exploration noise (DDPG.jl:57-61) and
// minibatch indices (memory_plotting_saving.jl:33).  The reference's dSFMT streams are not
// reproducible outside Julia; parity is defined on (state, action) -> (state', reward).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define PHILOX_HD __host__ __device__ __forceinline__
#else
#define PHILOX_HD inline
#endif

namespace shems {

struct u32x4 { uint32_t x, y, z, w; };

PHILOX_HD u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0;
        const uint64_t p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return u32x4{c0, c1, c2, c3};
}

// Stream tags (counter word 3) so the different consumers never share a counter.
enum : uint32_t { kStreamReset = 0x52455345u, kStreamRandAct = 0x52414354u, kStreamNoise = 0x4E4F4953u,
                  kStreamSample = 0x53414D50u, kStreamInit = 0x494E4954u };

PHILOX_HD float u01_24(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }   // [0,1), exact in f32

}  // namespace shems
