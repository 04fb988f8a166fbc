// shems_env_dev.h -- device-side building blocks shared by the env kernels (shems_env.hip) and the
// fused policy+step kernel (shems_policy.hip): config/table access, one-env advance (action ->
// flows -> next_state! -> bookkeeping), reset, results row, replay push.
// Reference: shems_LU1.jl:206-485, memory_plotting_saving.jl:46-47.
#pragma once
#include <hip/hip_runtime.h>
#include "shems_core.h"
#include "philox.h"

namespace shems {

constexpr int kBlock = 256;

// ---------------------------------------------------------------- helpers --
__device__ __forceinline__ shems_config load_cfg(const shems_view &v, int64_t i)
{
    // n_cfg == 1: wave-uniform address -> scalar loads; otherwise a 48-byte gather from L2.
    const int c = (v.n_cfg > 1) ? (int)v.cfg_of_env[i] : 0;
    return v.cfgs[c];
}

struct Row { float h, soc_ev, d_e, g_e, p_buy, h_cos, h_sin, season; };

__device__ __forceinline__ Row load_row(const float *tables, int64_t row0, int32_t idx1)
{
    const float4 *p = reinterpret_cast<const float4 *>(tables + (row0 + (int64_t)idx1 - 1) * SHEMS_NCOL);
    const float4 a = p[0], b = p[1];
    return Row{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
}

__device__ __forceinline__ float load_h(const float *tables, int64_t row0, int32_t idx1)
{
    return tables[(row0 + (int64_t)idx1 - 1) * SHEMS_NCOL];
}

__device__ __forceinline__ void raise(int32_t *err, int code)
{
    if (err) atomicCAS(err, 0, code);          // sticky: first error wins
}

// obs slab of this workgroup: global <-> LDS.  A full 256-env slab is 9216 contiguous, 16-byte aligned bytes = 576 float4:
// three 16-B-per-lane passes (the widest coalesced access); the ragged last workgroup falls back to dword passes.
__device__ __forceinline__ void slab_load(float *lds, const float *g, int64_t base_env, int64_t n)
{
    const int64_t first = base_env * SHEMS_NSTATE;
    if (base_env + kBlock <= n) {
        const float4 *src = reinterpret_cast<const float4 *>(g + first);
        float4 v[3];
#pragma unroll
        for (int it = 0; it < 3; ++it) v[it] = src[min(it * kBlock + (int)threadIdx.x, kBlock * SHEMS_NSTATE / 4 - 1)];
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int e = it * kBlock + threadIdx.x;
            if (e < kBlock * SHEMS_NSTATE / 4) reinterpret_cast<float4 *>(lds)[e] = v[it];
        }
        return;
    }
    const int64_t total = n * SHEMS_NSTATE;
#pragma unroll
    for (int k = 0; k < SHEMS_NSTATE; ++k) {
        const int o = k * kBlock + threadIdx.x;
        if (first + o < total) lds[o] = g[first + o];
    }
}
__device__ __forceinline__ void slab_store(const float *lds, float *g, int64_t base_env, int64_t n)
{
    const int64_t first = base_env * SHEMS_NSTATE;
    if (base_env + kBlock <= n) {
        float4 *dst = reinterpret_cast<float4 *>(g + first);
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int e = it * kBlock + threadIdx.x;
            if (e < kBlock * SHEMS_NSTATE / 4) dst[e] = reinterpret_cast<const float4 *>(lds)[e];
        }
        return;
    }
    const int64_t total = n * SHEMS_NSTATE;
#pragma unroll
    for (int k = 0; k < SHEMS_NSTATE; ++k) {
        const int o = k * kBlock + threadIdx.x;
        if (first + o < total) g[first + o] = lds[o];
    }
}

// Sum of `x` over the workgroup (wavefront butterfly, then 4 partials through LDS).
__device__ __forceinline__ double block_sum(double x, double *lds4, int nwaves = 4)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) lds4[w] = x;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) {
        s = ((lds4[0] + lds4[1]) + lds4[2]) + lds4[3];
        for (int w = 4; w < nwaves; ++w) s += lds4[w];
    }
    return s;
}

// One env, one hour: action -> flows -> next_state! -> bookkeeping.  obs is updated in place.
// Returns false (and leaves obs untouched) when row idx+1 does not exist (Julia: BoundsError).
// env_advance_rows takes the two table reads of next_state! (row idx+1 and h_countdown of row idx) as values, so that a caller
// with latency to hide can fetch them ahead (k_act does, behind its layer-3 epilogue); env_advance reads them in place.
__device__ __forceinline__ bool env_advance_rows(const shems_config &c, const Row &nx, float h_cur, float (&obs)[SHEMS_NSTATE],
                                                 int32_t &idx, int32_t &step, float a0, float a1, int track_mode,
                                                 double &reward, StepFlows &f, float &B, float &EV,
                                                 float &B_target, float &EV_target)
{
    if (idx < 1 || idx + 1 > c.nrow) return false;
    const EnvIn s{obs[0], obs[1], obs[2], obs[3], obs[4], obs[5]};
    if (track_mode >= 0) {                       // LU1:346-349
        B_target = a0; EV_target = a1;
        action_drl(c, s, B_target, EV_target, B, EV);
    } else {                                     // LU1:350-354
        B_target = 0.0f; EV_target = 0.0f;
        B = a0; EV = a1;
    }
    float soc_b_n, soc_ev_n;
    step_flows(c, s, EV_target, B, EV, track_mode < 0, soc_b_n, soc_ev_n, reward, f);

    // next_state!  LU1:264-281
    if (nx.h >= 0.0f && h_cur == -1.0f) soc_ev_n = nx.soc_ev;     // newly connected EV
    obs[0] = soc_b_n; obs[1] = soc_ev_n; obs[2] = nx.h; obs[3] = nx.d_e; obs[4] = nx.g_e;
    obs[5] = nx.p_buy; obs[6] = nx.h_cos; obs[7] = nx.h_sin; obs[8] = nx.season;
    step += 1;                                   // LU1:455
    idx += 1;                                    // LU1:456
    return true;
}

__device__ __forceinline__ bool env_advance(const shems_config &c, const float *tables, float (&obs)[SHEMS_NSTATE],
                                            int32_t &idx, int32_t &step, float a0, float a1, int track_mode,
                                            double &reward, StepFlows &f, float &B, float &EV,
                                            float &B_target, float &EV_target)
{
    if (idx < 1 || idx + 1 > c.nrow) return false;
    const Row nx = load_row(tables, c.table_row0, idx + 1);
    const float h_cur = load_h(tables, c.table_row0, idx);
    return env_advance_rows(c, nx, h_cur, obs, idx, step, a0, a1, track_mode, reward, f, B, EV, B_target, EV_target);
}

__device__ __forceinline__ void write_results(double *r, int32_t idx_after, const float (&pre)[SHEMS_NSTATE],
                                              float EV_target, float EV, double reward, const StepFlows &f,
                                              float B, float B_target)
{
    // LU1:476-478 column order
    r[0] = (double)idx_after; r[1] = (double)pre[2]; r[2] = (double)EV_target; r[3] = (double)EV;
    r[4] = (double)pre[1];    r[5] = reward;         r[6] = f.profit;         r[7] = f.discomfort;
    r[8] = f.penalty;         r[9] = f.PV_DE;        r[10] = f.B_DE;          r[11] = f.GR_DE;
    r[12] = f.PV_B;           r[13] = f.PV_GR;       r[14] = f.PV_EV;         r[15] = f.B_EV;
    r[16] = f.GR_EV;          r[17] = f.EX_EV;       r[18] = 0.0;             r[19] = 0.0;
    r[20] = (double)B;        r[21] = (double)B_target; r[22] = (double)pre[0];
}


// remember(s, a, r, s', done)  (memory_plotting_saving.jl:46-47) into the HBM ring.
__device__ __forceinline__ void ring_push(const shems_replay &ring, int64_t slot, const float (&s)[SHEMS_NSTATE],
                                          float a0, float a1, float r, const float (&s2)[SHEMS_NSTATE])
{
    float *ps = ring.s + slot * SHEMS_NSTATE, *p2 = ring.s2 + slot * SHEMS_NSTATE;
#pragma unroll
    for (int k = 0; k < SHEMS_NSTATE; ++k) { ps[k] = s[k]; p2[k] = s2[k]; }
    reinterpret_cast<float2 *>(ring.a)[slot] = make_float2(a0, a1);
    ring.r[slot] = r;
    ring.done[slot] = 0;                         // finished() is always false, shems_LU1.jl:487-502
}

}  // namespace shems
