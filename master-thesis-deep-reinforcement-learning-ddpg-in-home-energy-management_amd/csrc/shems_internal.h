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
// Floats of shems_ddpg.ws every caller allocates (shems_ddpg_workspace_floats): the latency form's carve (shems_ddpg.hip) defines it, the
// throughput form of the grouped update (shems_gupd.hip) carves its own, smaller, layout out of the same block.
constexpr int64_t kTpWsFloats = 1902568;
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

// ---- data-parallel replicas: the direct gradient exchange (csrc/shems_dp.hip owns the memory, csrc/shems_ddpg.hip the kernel) ---------
// Every rank owns an INBOX [2 parities][world][kXchgNmax] float32 and FLAGS [2][world][kXchgWgs] uint64 in fine-grained device memory that
// every peer has mapped (hipIpcOpenMemHandle).  The ADAM sweep of an exchange (k_adam_xchg, one workgroup per 1 024 consecutive
// parameters) PUSHES its range of the local gradient into slot [parity][my rank] of every peer's inbox, releases at system scope, stamps
// the peers' flag [parity][my rank][workgroup] with the exchange's epoch, WAITS (bounded: seconds; a wait that gives up POISONS the record, see XchgArgs::poison) for the same workgroup's flag from every peer
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
    unsigned *poison;                                 // pinned host word (device address): set when a wait gives up, never cleared
    unsigned long long wait_ticks;                    // bound of one wait, s_memrealtime ticks (100 MHz)
};
// ADAM + soft update of one network behind a direct exchange of its gradient (grad_scale 1 / world inside).
int ddpg_apply_xchg(const shems_ddpg *d, bool critic, double eta, double bp1, double bp2, float *d_publish, const XchgArgs &x, hipStream_t st);
}
