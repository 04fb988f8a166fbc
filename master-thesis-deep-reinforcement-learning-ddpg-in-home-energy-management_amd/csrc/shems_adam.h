// shems_adam.h -- Flux 0.12.1 ADAM + soft target update on one element, and the learner-group pointer shift: shared by the
// latency form (shems_ddpg.hip) and the throughput form (shems_gupd.hip) of the update so that both round identically.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>

namespace shems {

// Learner groups (shems_group): learner l's copy of every device buffer is learner 0's pointer + l * stride bytes.
template <class T>
__device__ __forceinline__ T *gsh(T *p, int64_t off)
{
    // byte arithmetic on the pointer itself (no round trip through an integer): the compiler keeps the global address space of the
    // kernel argument it came from and emits global_load / global_store instead of flat accesses
    typedef typename std::conditional<std::is_const<T>::value, const char, char>::type B;
    return p ? reinterpret_cast<T *>(reinterpret_cast<B *>(p) + off) : p;
}
// ---- Flux 0.12.1 ADAM + soft target update -----------------------------------------------------------------------------
//   mt = b1*mt + (1-b1)*g ; vt = b2*vt + (1-b2)*g^2 ; delta = mt/(1-bp1) / (sqrt(vt/(1-bp2)) + eps) * eta ; p -= delta
//   (Float64 scalars broadcast over Float32 arrays: each element is computed in f64 and stored as f32; g^2 is the Float32 square)
//   then target = (1f0 - tau) * target + tau * p   (DDPG.jl:99-103)
struct AdamCtx {
    float *p; const float *g; float *mt, *vt, *target; float *publish;
    int n, in; double eta, bp1, bp2, gscale;
    double k1, ic2;        // eta / (1 - bp1), 1 / (1 - bp2): host-side Float64 quotients
    float tau;
};
__device__ __forceinline__ void gshift(AdamCtx &c, int64_t off)
{
    c.p = gsh(c.p, off); c.g = gsh(c.g, off); c.mt = gsh(c.mt, off); c.vt = gsh(c.vt, off); c.target = gsh(c.target, off);
    c.publish = gsh(c.publish, off);
}

// One element of ADAM + soft update on values: (m, v, p, target) in, updated in place.
__device__ __forceinline__ void adam_math(const AdamCtx &c, float graw, float &m, float &v, float &p, float &t)
{
    // Julia evaluates the broadcast expressions without fusing multiplies into adds; keeping the compiler from contracting also
    // makes the inlined copies of this function (gradient tiles / sweep) round identically.
#pragma clang fp contract(off)
    const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
    const float gf = (float)((double)graw * c.gscale);          // averaged gradient, as every replica holds it
    const float m1 = (float)(b1 * (double)m + (1.0 - b1) * (double)gf);
    const float g2 = gf * gf;                                    // Flux 0.12.1 `Δ^2` on a Float32 array: literal_pow = Δ*Δ in Float32, then promoted
    const float v1 = (float)(b2 * (double)v + (1.0 - b2) * (double)g2);
    // delta = Float32(mt / (1 - bp1) / (sqrt(vt / (1 - bp2)) + eps) * eta), every operation in Float64 (Flux's scalars are Float64).
    // Evaluated as (mt * k1) / s with s = sqrt(vt * ic2) + eps, k1 = eta / (1 - bp1), ic2 = 1 / (1 - bp2), the quotient by a
    // reciprocal refined to <= 1 ulp (two Newton steps + a residual correction): three Float64 divisions become none.  The Float64
    // value differs from the reference's left-to-right evaluation by a few ulp(Float64) at most, i.e. the Float32 it rounds to
    // differs only when it falls within ~1e-15 of a Float32 rounding boundary (about one element in 1e7, by one Float32 ulp of delta).
    const double sq = sqrt((double)v1 * c.ic2) + eps;
    double y = __builtin_amdgcn_rcp(sq);
    y = __builtin_fma(__builtin_fma(-sq, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-sq, y, 1.0), y, y);
    const double tnum = (double)m1 * c.k1;
    double qd = tnum * y;
    qd = __builtin_fma(__builtin_fma(-sq, qd, tnum), y, qd);
    const float delta = (float)qd;
    const float pn = p - delta;
    const float one_m_tau = 1.0f - c.tau;
    t = one_m_tau * t + c.tau * pn;
    m = m1; v = v1; p = pn;
}
}  // namespace shems
