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
    if (!g_rccl.h) return set_error(SHEMS_ERR_STATE, "%s: RCCL (librccl.so.1) could not be loaded: %s", fn, dlerror() ? dlerror() : "symbols missing");
    return SHEMS_OK;
}
int nccl_ok(ncclResult_t r, const char *what)
{
    if (r == ncclSuccess) return SHEMS_OK;
    return set_error(SHEMS_ERR_HIP, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
}
}  // namespace

struct shems_dp {
    ncclComm_t comm;
    int rank, world, device;
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
    dp->rank = rank; dp->world = world; dp->comm = nullptr;
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
    delete dp;
    return rc;
}

int shems_dp_info(const shems_dp *dp, int *rank, int *world, char *lib, int32_t cap)
{
    if (!dp) return set_error(SHEMS_ERR_ARG, "shems_dp_info: NULL");
    if (rank) *rank = dp->rank;
    if (world) *world = dp->world;
    if (lib && cap > 1) snprintf(lib, (size_t)cap, "%s", g_rccl.where);
    return SHEMS_OK;
}

int shems_dp_allreduce_sum(shems_dp *dp, float *d_buf, int64_t n, void *stream)
{
    if (!dp || !d_buf || n < 1) return set_error(SHEMS_ERR_ARG, "shems_dp_allreduce_sum: bad arguments");
    return nccl_ok(g_rccl.AllReduce(d_buf, d_buf, (size_t)n, ncclFloat32, ncclSum, dp->comm, (hipStream_t)stream), "ncclAllReduce");
}

/* replay() of one replica among `world` (DDPG.jl:121-145 with the gradient exchange of SURVEY.md 8(e)): the split form of
 * shems_ddpg_update with both all-reduces in `stream`.  dp == NULL: one replica (the split form alone, grad_scale 1). */
int shems_ddpg_update_dp(const shems_ddpg *d, const shems_replay *ring, int64_t ring_len, uint64_t seed, uint32_t tick, int64_t excl_pos,
                         int64_t excl_count, double eta_crit, double bp1_crit, double bp2_crit, double eta_act, double bp1_act, double bp2_act,
                         float *d_publish, shems_dp *dp, void *stream)
{
    if (!d) return set_error(SHEMS_ERR_ARG, "shems_ddpg_update_dp: NULL");
    const double gs = dp ? 1.0 / (double)dp->world : 1.0;
    if (int rc = shems_ddpg_critic_grad_ex(d, ring, ring_len, seed, tick, excl_pos, excl_count, stream)) return rc;
    if (dp) if (int rc = shems_dp_allreduce_sum(dp, d->grad_critic, SHEMS_CRITIC_PARAMS, stream)) return rc;
    if (int rc = shems_ddpg_critic_apply(d, eta_crit, bp1_crit, bp2_crit, gs, stream)) return rc;
    if (int rc = shems_ddpg_actor_grad(d, stream)) return rc;
    if (dp) if (int rc = shems_dp_allreduce_sum(dp, d->grad_actor, SHEMS_ACTOR_PARAMS, stream)) return rc;
    return shems_ddpg_actor_apply_pub(d, eta_act, bp1_act, bp2_act, gs, d_publish, stream);
}

}  // extern "C"
