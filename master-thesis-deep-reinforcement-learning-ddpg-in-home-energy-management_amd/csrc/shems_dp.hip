// shems_dp.hip -- data-parallel replicas: the two gradient all-reduces of replay() (DDPG.jl:134-140) as RCCL calls IN THE
// UPDATE'S OWN STREAM, from native code.
//
// Why not torch.distributed's all_reduce: ProcessGroupNCCL runs every collective on a stream of its own, so each one costs the update two
// dependencies between queues (compute -> RCCL stream -> compute).  On this stack such a dependency costs 5-10 us whatever carries it
// (tools/xqueue_sync.hip), and round 3 measured 8.4 us per collective on a one-rank group before a byte crosses a link: 16.8 of the
// 80.6 us of a vector step at config 4's shard size (8 192 envs).  Nothing in replay() can overlap the exchange anyway (the critic's
// all-reduce feeds the critic's ADAM step, whose result feeds the actor's gradient): in-stream is the natural place.  ncclAllReduce on
// the caller's stream is one more launch in the chain and no queue hop.
//
// RCCL is loaded with dlopen at first use (librccl.so.1: the copy the process already mapped -- PyTorch-ROCm bundles one -- else
// /opt/rocm's), so libshems_hip.so has no link-time dependency on it: a single-GPU or Julia N = 1 host never touches RCCL.
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <new>
#include <rccl/rccl.h>          // types only (ncclComm_t, ncclUniqueId, enums); every function is resolved with dlsym
#include "shems_internal.h"

using namespace shems;

namespace {
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    char where[160] = "";
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (h) { snprintf(g_rccl.where, sizeof g_rccl.where, "%s (already mapped)", n); break; } }
    if (!h)
        for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) { snprintf(g_rccl.where, sizeof g_rccl.where, "%s", n); break; } }
    if (!h) return;
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.GetErrorString) g_rccl.h = h;
}
int need_rccl(const char *fn)
{
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.h) {
        const char *why = dlerror();
        return set_error(SHEMS_ERR_STATE, "%s: RCCL (librccl.so.1) could not be loaded: %s", fn, why ? why : "library or symbols missing");
    }
    return SHEMS_OK;
}
int nccl_ok(ncclResult_t r, const char *what)
{
    if (r == ncclSuccess) return SHEMS_OK;
    return set_error(SHEMS_ERR_HIP, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
}
}  // namespace

// The direct exchange's memory (XchgArgs, shems_internal.h): this rank's inbox and flags (fine-grained device memory, exported with
// hipIpcGetMemHandle) and the peers' as mapped here.
struct Direct {
    float *inbox[kXchgMaxWorld] = {};
    unsigned long long *flags[kXchgMaxWorld] = {};
    bool opened[kXchgMaxWorld] = {};
    int connected = 0;                    // peers mapped so far
    unsigned long long epoch = 0;         // exchanges enqueued so far
    unsigned *timeouts = nullptr;         // device word
    unsigned *poison_host = nullptr;      // pinned host word the kernels can store to: non-zero = a wait gave up (sticky, see shems_ddpg_update_dp)
    unsigned *poison_dev = nullptr;       // the same word as the device addresses it
    unsigned long long wait_ticks = 500000000ull;   // bound of one wait in s_memrealtime ticks (100 MHz): 5 s
    size_t inbox_bytes = 0, flags_bytes = 0;
};

struct shems_dp {
    ncclComm_t comm;
    int rank, world, device;
    Direct *direct;                       // non-null: gradients travel through peer-mapped inboxes, not RCCL
};

extern "C" {

int shems_dp_unique_id(char *out128)
{
    if (!out128) return set_error(SHEMS_ERR_ARG, "shems_dp_unique_id: NULL");
    if (int rc = need_rccl("shems_dp_unique_id")) return rc;
    ncclUniqueId id;
    if (int rc = nccl_ok(g_rccl.GetUniqueId(&id), "ncclGetUniqueId")) return rc;
    static_assert(sizeof id == SHEMS_DP_ID_BYTES, "ncclUniqueId is 128 bytes");
    std::memcpy(out128, &id, sizeof id);
    return SHEMS_OK;
}

int shems_dp_create(const char *id128, int rank, int world, shems_dp **out)
{
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return set_error(SHEMS_ERR_ARG, "shems_dp_create: bad arguments (rank %d of %d)", rank, world);
    if (int rc = need_rccl("shems_dp_create")) return rc;
    shems_dp *dp = new (std::nothrow) shems_dp;
    if (!dp) return set_error(SHEMS_ERR_NOMEM, "shems_dp_create: out of host memory");
    dp->rank = rank; dp->world = world; dp->comm = nullptr; dp->direct = nullptr;
    if (int rc = hip_ok(hipGetDevice(&dp->device), "hipGetDevice")) { delete dp; return rc; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    if (int rc = nccl_ok(g_rccl.CommInitRank(&dp->comm, world, id, rank), "ncclCommInitRank")) { delete dp; return rc; }
    *out = dp;
    return SHEMS_OK;
}

int shems_dp_destroy(shems_dp *dp)
{
    if (!dp) return SHEMS_OK;
    int rc = SHEMS_OK;
    if (dp->comm && g_rccl.h) rc = nccl_ok(g_rccl.CommDestroy(dp->comm), "ncclCommDestroy");
    if (Direct *x = dp->direct) {
        for (int q = 0; q < dp->world; ++q) {
            if (q == dp->rank) continue;
            if (x->opened[q]) { (void)hipIpcCloseMemHandle(x->inbox[q]); (void)hipIpcCloseMemHandle(x->flags[q]); }
        }
        if (x->inbox[dp->rank]) (void)hipFree(x->inbox[dp->rank]);
        if (x->flags[dp->rank]) (void)hipFree(x->flags[dp->rank]);
        if (x->timeouts) (void)hipFree(x->timeouts);
        if (x->poison_host) (void)hipHostFree(x->poison_host);
        delete x;
    }
    delete dp;
    return rc;
}

/* The direct exchange (no RCCL): every rank creates its record, publishes its two IPC handles (128 bytes: inbox, flags), maps every
 * peer's, and from then on shems_ddpg_update_dp exchanges gradients through the inboxes (k_adam_xchg, csrc/shems_ddpg.hip).  The caller
 * must make sure every rank has connected every peer before the first update (a barrier of its own). */
int shems_dp_create_direct(int rank, int world, shems_dp **out)
{
    if (!out || world < 1 || world > kXchgMaxWorld || rank < 0 || rank >= world)
        return set_error(SHEMS_ERR_ARG, "shems_dp_create_direct: rank %d of %d (at most %d replicas)", rank, world, kXchgMaxWorld);
    shems_dp *dp = new (std::nothrow) shems_dp;
    Direct *x = new (std::nothrow) Direct;
    if (!dp || !x) { delete dp; delete x; return set_error(SHEMS_ERR_NOMEM, "shems_dp_create_direct: out of host memory"); }
    dp->rank = rank; dp->world = world; dp->comm = nullptr; dp->direct = x;
    int rc = hip_ok(hipGetDevice(&dp->device), "hipGetDevice");
    x->inbox_bytes = (size_t)2 * world * kXchgNmax * sizeof(float);
    x->flags_bytes = (size_t)2 * world * kXchgWgs * sizeof(unsigned long long);
    // fine-grained: stores of a peer (another GPU, or another process on this one) become visible while kernels run
    if (!rc) rc = hip_ok(hipExtMallocWithFlags((void **)&x->inbox[rank], x->inbox_bytes, hipDeviceMallocFinegrained), "hipExtMallocWithFlags(inbox)");
    if (!rc) rc = hip_ok(hipExtMallocWithFlags((void **)&x->flags[rank], x->flags_bytes, hipDeviceMallocFinegrained), "hipExtMallocWithFlags(flags)");
    if (!rc) rc = hip_ok(hipMalloc((void **)&x->timeouts, sizeof(unsigned)), "hipMalloc(timeouts)");
    // the poison word lives in pinned, coherent host memory: the sweep stores to it at system scope, the host reads it with a plain load
    if (!rc) rc = hip_ok(hipHostMalloc((void **)&x->poison_host, sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc(poison)");
    if (!rc) { *x->poison_host = 0u; rc = hip_ok(hipHostGetDevicePointer((void **)&x->poison_dev, x->poison_host, 0), "hipHostGetDevicePointer(poison)"); }
    if (!rc) rc = hip_ok(hipMemset(x->inbox[rank], 0, x->inbox_bytes), "hipMemset(inbox)");
    if (!rc) rc = hip_ok(hipMemset(x->flags[rank], 0, x->flags_bytes), "hipMemset(flags)");
    if (!rc) rc = hip_ok(hipMemset(x->timeouts, 0, sizeof(unsigned)), "hipMemset(timeouts)");
    if (!rc) rc = hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
    if (rc) { shems_dp_destroy(dp); return rc; }
    *out = dp;
    return SHEMS_OK;
}

int shems_dp_direct_handles(shems_dp *dp, char *out128)
{
    if (!dp || !dp->direct || !out128) return set_error(SHEMS_ERR_ARG, "shems_dp_direct_handles: not a direct-exchange record");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "two 64-byte IPC handles");
    hipIpcMemHandle_t h[2];
    if (int rc = hip_ok(hipIpcGetMemHandle(&h[0], dp->direct->inbox[dp->rank]), "hipIpcGetMemHandle(inbox)")) return rc;
    if (int rc = hip_ok(hipIpcGetMemHandle(&h[1], dp->direct->flags[dp->rank]), "hipIpcGetMemHandle(flags)")) return rc;
    std::memcpy(out128, h, sizeof h);
    return SHEMS_OK;
}

int shems_dp_direct_connect(shems_dp *dp, int peer, const char *handles128)
{
    if (!dp || !dp->direct || !handles128 || peer < 0 || peer >= dp->world || peer == dp->rank)
        return set_error(SHEMS_ERR_ARG, "shems_dp_direct_connect: bad peer %d", peer);
    Direct *x = dp->direct;
    if (x->opened[peer]) return SHEMS_OK;
    hipIpcMemHandle_t h[2];
    std::memcpy(h, handles128, sizeof h);
    if (int rc = hip_ok(hipIpcOpenMemHandle((void **)&x->inbox[peer], h[0], hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle(inbox)")) return rc;
    if (int rc = hip_ok(hipIpcOpenMemHandle((void **)&x->flags[peer], h[1], hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle(flags)")) {
        (void)hipIpcCloseMemHandle(x->inbox[peer]); x->inbox[peer] = nullptr;
        return rc;
    }
    x->opened[peer] = true;
    x->connected += 1;
    return SHEMS_OK;
}

/* Exchange waits that gave up since the last call (read and cleared; synchronises `stream`).  Non-zero: a peer never delivered -- the
 * updates since then used an incomplete sum and the replicas have diverged. */
int shems_dp_direct_timeouts(shems_dp *dp, int64_t *out, void *stream)
{
    if (!dp || !dp->direct || !out) return set_error(SHEMS_ERR_ARG, "shems_dp_direct_timeouts: not a direct-exchange record");
    unsigned v = 0;
    if (int rc = hip_ok(hipMemcpyAsync(&v, dp->direct->timeouts, sizeof v, hipMemcpyDeviceToHost, (hipStream_t)stream), "memcpy timeouts")) return rc;
    if (int rc = hip_ok(hipStreamSynchronize((hipStream_t)stream), "sync")) return rc;
    if (v) if (int rc = hip_ok(hipMemsetAsync(dp->direct->timeouts, 0, sizeof v, (hipStream_t)stream), "memset timeouts")) return rc;
    *out = (int64_t)v;
    return SHEMS_OK;
}

/* Bound of one exchange wait in milliseconds (default 5 000; 1 .. 600 000).  A wait that reaches it POISONS the record. */
int shems_dp_direct_set_wait_ms(shems_dp *dp, int64_t ms)
{
    if (!dp || !dp->direct || ms < 1 || ms > 600000) return set_error(SHEMS_ERR_ARG, "shems_dp_direct_set_wait_ms: a direct-exchange record and 1..600000 ms");
    dp->direct->wait_ticks = (unsigned long long)ms * 100000ull;        // s_memrealtime: 100 MHz
    return SHEMS_OK;
}

/* 1 if an exchange wait of this record ever gave up (sticky; no synchronisation: the word is host memory the kernels store to). */
int shems_dp_direct_poisoned(const shems_dp *dp, int32_t *out)
{
    if (!dp || !dp->direct || !out) return set_error(SHEMS_ERR_ARG, "shems_dp_direct_poisoned: not a direct-exchange record");
    *out = *(volatile unsigned *)dp->direct->poison_host ? 1 : 0;
    return SHEMS_OK;
}

int shems_dp_info(const shems_dp *dp, int *rank, int *world, char *lib, int32_t cap)
{
    if (!dp) return set_error(SHEMS_ERR_ARG, "shems_dp_info: NULL");
    if (rank) *rank = dp->rank;
    if (world) *world = dp->world;
    if (lib && cap > 1) snprintf(lib, (size_t)cap, "%s", dp->direct ? "direct exchange (peer-mapped inboxes, no RCCL)" : g_rccl.where);
    return SHEMS_OK;
}

int shems_dp_allreduce_sum(shems_dp *dp, float *d_buf, int64_t n, void *stream)
{
    if (!dp || !d_buf || n < 1) return set_error(SHEMS_ERR_ARG, "shems_dp_allreduce_sum: bad arguments");
    if (!dp->comm) return set_error(SHEMS_ERR_STATE, "shems_dp_allreduce_sum: this record carries the direct exchange (no RCCL communicator): gradients are exchanged inside shems_ddpg_update_dp");
    return nccl_ok(g_rccl.AllReduce(d_buf, d_buf, (size_t)n, ncclFloat32, ncclSum, dp->comm, (hipStream_t)stream), "ncclAllReduce");
}

/* replay() of one replica among `world` (DDPG.jl:121-145 with the gradient exchange of SURVEY.md 8(e)): the split form of
 * shems_ddpg_update with both all-reduces in `stream`.  dp == NULL: one replica (the split form alone, grad_scale 1). */
int shems_ddpg_update_dp(const shems_ddpg *d, const shems_replay *ring, int64_t ring_len, uint64_t seed, uint32_t tick, int64_t excl_pos,
                         int64_t excl_count, double eta_crit, double bp1_crit, double bp2_crit, double eta_act, double bp1_act, double bp2_act,
                         float *d_publish, shems_dp *dp, void *stream)
{
    if (!d) return set_error(SHEMS_ERR_ARG, "shems_ddpg_update_dp: NULL");
    // The split form below runs K2 WITH the actor's two E products: a caller's "leave them to shems_ddpg_actor_prepare" flag (the
    // asynchronous torch.distributed form of replay()) must not reach it -- nobody here would ever issue them.
    shems_ddpg dl = *d;
    dl.flags &= ~SHEMS_DDPG_DEFER_ACTOR_E;
    d = &dl;
    if (dp && dp->direct) {
        // direct exchange: the ADAM sweep of each network pushes / waits / sums in rank order (k_adam_xchg); two launches less than RCCL's form
        Direct *x = dp->direct;
        if (x->connected != dp->world - 1) return set_error(SHEMS_ERR_STATE, "shems_ddpg_update_dp: %d of %d peers connected", x->connected, dp->world - 1);
        // Sticky: once a wait has given up (a peer never delivered within the bound), this replica has applied -- or skipped -- an
        // incomplete sum and the replicas have diverged; every later call fails instead of training on.  The sweeps already enqueued
        // see the same word and turn into no-ops (they neither push, wait nor apply), so the queue drains at once.
        if (*(volatile unsigned *)x->poison_host)
            return set_error(SHEMS_ERR_STATE, "shems_ddpg_update_dp: a gradient exchange wait gave up earlier (a peer did not deliver within %.1f s): "
                             "the replicas have diverged -- this record is poisoned, restart the replicas from a common snapshot", (double)x->wait_ticks * 1e-8);
        XchgArgs a;
        std::memset(&a, 0, sizeof a);
        for (int q = 0; q < dp->world; ++q) { a.inbox[q] = x->inbox[q]; a.flags[q] = x->flags[q]; }
        a.rank = dp->rank; a.world = dp->world; a.timeouts = x->timeouts; a.poison = x->poison_dev; a.wait_ticks = x->wait_ticks;
        if (int rc = shems_ddpg_critic_grad_ex(d, ring, ring_len, seed, tick, excl_pos, excl_count, stream)) return rc;
        a.epoch = ++x->epoch;
        if (int rc = ddpg_apply_xchg(d, true, eta_crit, bp1_crit, bp2_crit, nullptr, a, (hipStream_t)stream)) return rc;
        if (int rc = shems_ddpg_actor_grad(d, stream)) return rc;
        a.epoch = ++x->epoch;
        return ddpg_apply_xchg(d, false, eta_act, bp1_act, bp2_act, d_publish, a, (hipStream_t)stream);
    }
    const double gs = dp ? 1.0 / (double)dp->world : 1.0;
    if (int rc = shems_ddpg_critic_grad_ex(d, ring, ring_len, seed, tick, excl_pos, excl_count, stream)) return rc;
    if (dp) if (int rc = shems_dp_allreduce_sum(dp, d->grad_critic, SHEMS_CRITIC_PARAMS, stream)) return rc;
    if (int rc = shems_ddpg_critic_apply(d, eta_crit, bp1_crit, bp2_crit, gs, stream)) return rc;
    if (int rc = shems_ddpg_actor_grad(d, stream)) return rc;
    if (dp) if (int rc = shems_dp_allreduce_sum(dp, d->grad_actor, SHEMS_ACTOR_PARAMS, stream)) return rc;
    return shems_ddpg_actor_apply_pub(d, eta_act, bp1_act, bp2_act, gs, d_publish, stream);
}

}  // extern "C"
