// shems_internal.h -- shared by the translation units of libshems_hip.so (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include "../../include/shems_hip.h"

namespace shems {
int set_error(int code, const char *fmt, ...);          // records the thread-local message, returns code
int hip_ok(hipError_t e, const char *what);             // SHEMS_OK or SHEMS_ERR_HIP (+ message)
// Kernels with more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize, and the attribute is PER DEVICE: a
// C-ABI caller may drive several GPUs from one process (shems_create(..., device, ...)), so the opt-in is remembered per
// (kernel, current device) -- `mask` is the kernel's own static bit set, bit = device ordinal.
inline int lds_optin(std::atomic<uint64_t> &mask, const void *fn, int bytes, const char *what)
{
    int dev = 0;
    if (int rc = hip_ok(hipGetDevice(&dev), "hipGetDevice")) return rc;
    const uint64_t bit = 1ull << (dev & 63);
    if (mask.load(std::memory_order_acquire) & bit) return SHEMS_OK;
    if (int rc = hip_ok(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), what)) return rc;
    mask.fetch_or(bit, std::memory_order_release);
    return SHEMS_OK;
}
int check_view(const shems_view *v, const char *fn);    // every entry point that dereferences a caller-built view (shems_env.hip)
// ADAM (Flux 0.12.1) + soft target update over n consecutive parameters, the data-parallel path's sweep (shems_ddpg.hip: k_adam_soft)
// on any parameter count: the wide-network path (shems_wide.hip) applies its gradients with it.
int adam_soft_sweep(float *p, const float *g, float *m, float *v, float *target, float *publish, int n, double eta, double bp1, double bp2,
                    double gscale, float tau, hipStream_t st);
// shems_wide.hip: forward pass of an actor (9 -> l1 -> l2 -> 2) of any hidden sizes for m observations, up to the output layer's partial
// sums: d_part [*n_partials][m][2] (b3 not included; the caller adds b3 and the partials in index order).  d_ws holds
// wide_act_ws_floats(l1, l2, m) floats of scratch (normalised observations, layer 1, then -- at wide_act_part_offset -- the partials).
int wide_actor_pre(const float *actor, const float *s_min, const float *s_max, int l1, int l2, const float *d_obs, int64_t m, float *d_ws,
                   float *d_part, int *n_partials, hipStream_t st);
int64_t wide_act_ws_floats(int l1, int l2, int64_t m);
int64_t wide_act_part_offset(int l1, int64_t m);

// ---- device-side dependencies between launches of DIFFERENT queues (the pipelined training loop, shems_train.hip) ----------------
// A dependency between two HIP queues costs the waiting queue 5-10 us on this stack whatever carries it (events 9.6 us, stream memory
// operations 4.9 us -- they are one-thread kernels, tools/xqueue_sync.hip), which is a quarter of a vector step at <= 8 192 envs.  So
// the two queues of the pipelined loop never wait for each other: the CONSUMING launch is enqueued early and its workgroups wait, in
// the kernel, for a count of producer workgroups that have finished (one 8-byte word per direction, on a cache line of its own).
//   producer workgroup, last thing:  make its stores visible device-wide (write-through stores + s_waitcnt, or a release fence),
//                                    barrier, ONE lane adds 1 to the word (relaxed, agent scope: performed at the memory side);
//   consumer workgroup, first thing: lane 0 polls the word past the caches until it reaches the target, bounded (a consumer that
//                                    cannot be satisfied gives up after ~tens of ms, counts itself in *timeouts and carries on with
//                                    stale data: it never hangs; the host reads the counter, shems_ddpg_sync_timeouts), barrier.
// The consumer needs no acquire fence for what it reads afterwards: those lines were last touched on its XCD before its own launch
// started (every launch starts with an invalidated L2), and nobody on the device reads them between the producer's previous and
// current write (actor_pub[i] is read by act(t), act(t + 2), ... only; the ring window of step t is excluded from replay(t)'s sampler).
// Two kinds of words, never on one cache line: producers ADD to a counter; the one whose add completes the launch's total then stores
// that total into kDevFlagCopies FLAG words (a line each), and consumer workgroup b polls flag copy b % kDevFlagCopies.  Hundreds of
// pollers on the line the producers add to would queue those adds behind their loads (round 3 measured it inside one launch: 41 -> 37 us
// when the polled word got a line of its own); spread over 16 lines and polled every ~0.5 us the pollers cost the producers nothing.
constexpr int kDevFlagCopies = 16, kDevLineWords = 16;        // 16 x 8 bytes = one 128-byte line
struct DevSync {
    const unsigned long long *wait_flags;  // null: nothing to wait for; else [kDevFlagCopies][kDevLineWords]
    unsigned long long wait_target;        // satisfied when flag >= target
    unsigned long long *arrive_count;      // null: nobody is told
    unsigned long long *arrive_flags;      // [kDevFlagCopies][kDevLineWords], written by the arrival that makes the count reach arrive_total
    unsigned long long arrive_total;       // the count when every workgroup of this launch (and of all earlier ones) has arrived
    unsigned *timeouts;
};
constexpr size_t kDevSyncBytes = (size_t)(2 * (1 + kDevFlagCopies) * kDevLineWords) * 8;   // two directions x (counter line + flag lines)
#ifdef __HIPCC__
constexpr unsigned kDevWaitSpins = 1u << 16;      // x (one sc1 load + s_sleep 8) ~ tens of ms
__device__ __forceinline__ void dev_wait(const DevSync &sy)
{
    if (!sy.wait_flags) return;                     // kernel-argument uniform
    if (threadIdx.x == 0) {
        const unsigned long long *w = sy.wait_flags + (blockIdx.x % kDevFlagCopies) * kDevLineWords;
        unsigned spins = 0;
        while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sy.wait_target) {
            if (++spins > kDevWaitSpins) { __hip_atomic_fetch_add(sy.timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(16);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       // no instruction: keeps the payload loads below the barrier
}
// fence: true = this workgroup's plain stores must become visible (agent-scope release: writes the XCD's dirty L2 lines back);
// false = everything the consumer reads was stored write-through (sc1) already, draining the wave's stores is enough.
__device__ __forceinline__ void dev_arrive(const DevSync &sy, bool fence)
{
    if (!sy.arrive_count) return;
    if (fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64) {
        unsigned long long old = 0;
        if (threadIdx.x == 0) old = __hip_atomic_fetch_add(sy.arrive_count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __shfl(old, 0, 64);
        // the arrival that completes the total tells the consumers: lanes 0..15 store one flag copy each
        if (old + 1 == sy.arrive_total && threadIdx.x < kDevFlagCopies)
            __hip_atomic_store(sy.arrive_flags + threadIdx.x * kDevLineWords, sy.arrive_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
#endif
// The fused step with a device-side dependency (shems_policy.hip): as shems_act_step_dev; split = 1 forces the two-workgroups-per-tile
// form (92 KB of LDS per workgroup, one per CU: a 64-KB workgroup of the update fits beside it); *grid_out = workgroups launched.
int act_step_sync(const shems_view *v, const shems_act_params *p, float *d_rewards_f32, const shems_replay *ring, const shems_ring_window *window,
                  const DevSync &sy, int split, int64_t *grid_out, hipStream_t st);
// replay() in one call (shems_ddpg.hip: shems_ddpg_update) whose first launch waits for `first.wait_*` and whose last launch arrives at
// `last.arrive_word`; *k5_grid = workgroups of that last launch.
int ddpg_update_sync(const shems_ddpg *d, const shems_replay *ring, int64_t ring_len, uint64_t seed, uint32_t tick, int64_t excl_pos,
                     int64_t excl_count, double eta_crit, double bp1_crit, double bp2_crit, double eta_act, double bp1_act, double bp2_act,
                     float *d_publish, const DevSync &first, const DevSync &last, hipStream_t st);
int ddpg_last_launch_grid();
unsigned *ddpg_timeout_word(const shems_ddpg *d);

// ---- data-parallel replicas: the direct gradient exchange (csrc/shems_dp.hip owns the memory, csrc/shems_ddpg.hip the kernel) ---------
// Every rank owns an INBOX [2 parities][world][kXchgNmax] float32 and FLAGS [2][world][kXchgWgs] uint64 in fine-grained device memory that
// every peer has mapped (hipIpcOpenMemHandle).  The ADAM sweep of an exchange (k_adam_xchg, one workgroup per 1 024 consecutive
// parameters) PUSHES its range of the local gradient into slot [parity][my rank] of every peer's inbox, releases at system scope, stamps
// the peers' flag [parity][my rank][workgroup] with the exchange's epoch, WAITS (bounded) for the same workgroup's flag from every peer
// in its own flags, then sums the slots IN RANK ORDER (its own gradient in its own position: every replica adds the same numbers in the
// same order, so the replicas stay bit-identical) and applies ADAM with grad_scale 1 / world.  No collective launch, no launch at all
// beyond the sweep the data-parallel form has anyway.  Epochs count exchanges (critic, actor, critic, ...), parity = epoch & 1: a slot
// is rewritten two exchanges later, by which time its owner has consumed it (a rank pushes epoch e + 1 only after consuming epoch e,
// and epoch e + 2 only after every peer's e + 1 has arrived).
constexpr int kXchgMaxWorld = 8;
constexpr int kXchgNmax = 129024;                     // 126 x 1 024 >= SHEMS_ACTOR_PARAMS
constexpr int kXchgWgs = kXchgNmax / 1024;
struct XchgArgs {
    float *inbox[kXchgMaxWorld];                      // inbox[q]: rank q's inbox as mapped in this process (inbox[rank]: the local allocation)
    unsigned long long *flags[kXchgMaxWorld];
    int rank, world;
    unsigned long long epoch;                         // this exchange (>= 1)
    unsigned *timeouts;                               // device word: waits that gave up
};
// ADAM + soft update of one network behind a direct exchange of its gradient (grad_scale 1 / world inside).
int ddpg_apply_xchg(const shems_ddpg *d, bool critic, double eta, double bp1, double bp2, float *d_publish, const XchgArgs &x, hipStream_t st);
}
