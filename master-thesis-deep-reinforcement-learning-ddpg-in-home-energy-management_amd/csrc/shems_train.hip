// shems_train.hip -- the hour loop of episode! (DDPG.jl:195-234) enqueued natively: shems_train_steps.
//
// Host code only.  Every launch goes through the library's own entry points (shems_act_step_dev / shems_act_step_range_dev /
// shems_ddpg_update / shems_reset_seeded_dev) with the arguments a host loop over them would pass, so the loop is those calls minus the
// foreign-call cost per launch (a ctypes call costs ~6 us; at <= 8 192 envs the six launches of a vector step take 50-60 us, and in the
// pipelined modes the host would otherwise be the bound: tools/r04_b2.sh traced act(t) and replay(t) strictly one after the other
// because the second stream's launches were not even enqueued when the first kernel ended).
//
// Stream / event protocol of the pipelined modes (A = `stream`, B = `stream2`; pub[i] = actor_pub[i]):
//   SHEMS_LOOP_PIPELINED        A: wait U(t-1); act(t) reading pub[t & 1]; record S(t)
//                               B: wait S(t-1); replay(t) sampling the ring without step t's window, publishing pub[(t+1) & 1]; record U(t)
//       act(t) and replay(t) are independent of each other and both follow replay(t-1): they run concurrently.  pub[(t+1) & 1] is
//       last read by act(t-1), which B has waited for; step t's ring writes land in the window replay(t) does not sample; replay(t+1)
//       follows S(t).
//   SHEMS_LOOP_PIPELINED_EXACT  A: wait U(t-1); act_window(t) (the envs whose transitions enter the ring, <= 2 range launches);
//                                  record W(t); act_rest(t) (<= 2 range launches) -- all reading pub[t & 1]
//                               B: wait W(t); replay(t) sampling the whole ring, publishing pub[(t+1) & 1]; record U(t)
//       replay(t) sees step t's inserts like the ordered loop, act(t+1) the actor replay(t) produced: the bytes of the ordered loop.
//       act_rest(t) runs under replay(t).  pub[(t+1) & 1] was last read by act_rest(t-1): A ran it before act_window(t), whose W(t) B
//       waits for.
// Two events of each kind alternate; an event is re-recorded only after its last waiter has been enqueued (stream order), which is all
// hipStreamWaitEvent needs (the wait captures the record that precedes it).
#include <cstring>
#include <new>
#include "shems_internal.h"

using namespace shems;

namespace {
struct LoopSync {
    int device = -1;
    hipEvent_t stepped[2] = {nullptr, nullptr};     // S(t) / W(t): the ring holds everything replay(t + 1) / replay(t) may sample
    hipEvent_t updated[2] = {nullptr, nullptr};     // U(t): replay(t) done, pub[(t + 1) & 1] published
    bool have_updated[2] = {false, false}, have_stepped[2] = {false, false};
};

int sync_of(shems_train_loop *L, LoopSync **out)
{
    if (!L->sync) {
        LoopSync *s = new (std::nothrow) LoopSync;
        if (!s) return set_error(SHEMS_ERR_NOMEM, "shems_train_steps: out of host memory");
        if (int rc = hip_ok(hipGetDevice(&s->device), "hipGetDevice")) { delete s; return rc; }
        for (int i = 0; i < 2; ++i) {
            if (int rc = hip_ok(hipEventCreateWithFlags(&s->stepped[i], hipEventDisableTiming), "hipEventCreate")) { delete s; return rc; }
            if (int rc = hip_ok(hipEventCreateWithFlags(&s->updated[i], hipEventDisableTiming), "hipEventCreate")) { delete s; return rc; }
        }
        L->sync = s;
    }
    *out = static_cast<LoopSync *>(L->sync);
    return SHEMS_OK;
}

int check_loop(const shems_train_loop *L, void *stream, void *stream2)
{
    if (!L) return set_error(SHEMS_ERR_ARG, "shems_train_steps: loop is NULL");
    const int64_t n = L->view.n_envs;
    if (L->window < 0 || L->window > n || L->window > L->ring.capacity)
        return set_error(SHEMS_ERR_ARG, "shems_train_steps: window of %lld envs outside the batch (%lld) or the ring (%lld)", (long long)L->window,
                         (long long)n, (long long)L->ring.capacity);
    if (L->ep_len <= 0 || L->updates_per_step < 0 || L->t < 0 || L->updates < 0 || L->ring_pushed < 0)
        return set_error(SHEMS_ERR_ARG, "shems_train_steps: ep_len must be positive, counters non-negative");
    if (L->mode < SHEMS_LOOP_ORDERED || L->mode > SHEMS_LOOP_PIPELINED_EXACT) return set_error(SHEMS_ERR_ARG, "shems_train_steps: unknown mode %d", L->mode);
    if (L->updates_per_step > 0) {
        for (int i = 0; i < 2; ++i)
            if (!(L->bp_crit[i] > 0.0 && L->bp_crit[i] < 1.0 && L->bp_act[i] > 0.0 && L->bp_act[i] < 1.0))
                return set_error(SHEMS_ERR_ARG, "shems_train_steps: ADAM beta powers must lie in (0, 1)");
        if (L->act.actor != L->ddpg.actor) return set_error(SHEMS_ERR_ARG, "shems_train_steps: act.actor must be the learner's actor (ddpg.actor)");
    }
    if (L->mode != SHEMS_LOOP_ORDERED) {
        if (L->updates_per_step < 1) return set_error(SHEMS_ERR_ARG, "shems_train_steps: a pipelined mode needs updates_per_step >= 1");
        if (!L->actor_pub[0] || !L->actor_pub[1] || L->actor_pub[0] == L->actor_pub[1])
            return set_error(SHEMS_ERR_ARG, "shems_train_steps: a pipelined mode needs two distinct actor_pub buffers");
        if (((uintptr_t)L->actor_pub[0] & 15) || ((uintptr_t)L->actor_pub[1] & 15))
            return set_error(SHEMS_ERR_ARG, "shems_train_steps: actor_pub buffers must be 16-byte aligned");
        if (stream == stream2) return set_error(SHEMS_ERR_ARG, "shems_train_steps: a pipelined mode needs two different streams");
    }
    return SHEMS_OK;
}

// The envs of a step's ring window, offset .. offset + count - 1 (mod n), as at most two ascending ranges; `rest` = the complement.
struct Ranges { int64_t lo[2], cnt[2]; int k; };
void window_ranges(int64_t n, int64_t offset, int64_t count, Ranges *win, Ranges *rest)
{
    win->k = rest->k = 0;
    if (count <= 0) { rest->lo[0] = 0; rest->cnt[0] = n; rest->k = 1; return; }
    if (offset + count <= n) {
        win->lo[0] = offset; win->cnt[0] = count; win->k = 1;
        if (offset > 0) { rest->lo[rest->k] = 0; rest->cnt[rest->k] = offset; ++rest->k; }
        if (offset + count < n) { rest->lo[rest->k] = offset + count; rest->cnt[rest->k] = n - offset - count; ++rest->k; }
    } else {
        const int64_t head = offset + count - n;               // [0, head) and [offset, n)
        win->lo[0] = 0; win->cnt[0] = head; win->lo[1] = offset; win->cnt[1] = n - offset; win->k = 2;
        if (offset > head) { rest->lo[0] = head; rest->cnt[0] = offset - head; rest->k = 1; }
    }
}
}  // namespace

extern "C" {

int shems_train_steps(shems_train_loop *L, int64_t k, void *stream, void *stream2)
{
    if (int rc = check_loop(L, stream, stream2)) return rc;
    if (k < 0) return set_error(SHEMS_ERR_ARG, "shems_train_steps: k < 0");
    hipStream_t A = (hipStream_t)stream, B = (hipStream_t)stream2;
    LoopSync *S = nullptr;
    if (L->mode != SHEMS_LOOP_ORDERED)
        if (int rc = sync_of(L, &S)) return rc;
    const int64_t n = L->view.n_envs, cap = L->ring.capacity;
    for (int64_t it = 0; it < k; ++it) {
        const int64_t t = L->t;
        if (L->mode != SHEMS_LOOP_ORDERED && S->have_updated[(t - 1) & 1])
            if (int rc = hip_ok(hipStreamWaitEvent(A, S->updated[(t - 1) & 1], 0), "hipStreamWaitEvent")) return rc;   // pub[t & 1] is replay(t - 1)'s
        if (t > 0 && t % L->ep_len == 0) {                                  // DDPG.jl:189-193: the next episode's reset!(env)
            L->episode += 1;
            if (int rc = shems_reset_seeded_dev(&L->view, L->env_seed, L->episode, A)) return rc;
        }
        shems_act_params p = L->act;
        p.tick = (uint32_t)(t & 0xFFFFFFFFll);
        if (L->mode != SHEMS_LOOP_ORDERED) p.actor = L->actor_pub[t & 1];
        const bool use_ring = L->window > 0 && cap > 0;
        shems_ring_window w = {use_ring ? L->ring_pushed % cap : 0, use_ring ? L->window : 0, use_ring ? (t * L->window) % n : 0};
        if (L->mode == SHEMS_LOOP_PIPELINED_EXACT) {
            Ranges win, rest;
            window_ranges(n, w.offset, w.count, &win, &rest);
            for (int i = 0; i < win.k; ++i)
                if (int rc = shems_act_step_range_dev(&L->view, &p, win.lo[i], win.cnt[i], L->rewards_f32, &L->ring, &w, A)) return rc;
            if (int rc = hip_ok(hipEventRecord(S->stepped[t & 1], A), "hipEventRecord")) return rc;
            S->have_stepped[t & 1] = true;
            for (int i = 0; i < rest.k; ++i)
                if (int rc = shems_act_step_range_dev(&L->view, &p, rest.lo[i], rest.cnt[i], L->rewards_f32, &L->ring, &w, A)) return rc;
        } else {
            if (int rc = shems_act_step_dev(&L->view, &p, nullptr, nullptr, L->rewards_f32, nullptr, nullptr, use_ring ? &L->ring : nullptr,
                                            use_ring ? &w : nullptr, A))
                return rc;
            if (L->mode == SHEMS_LOOP_PIPELINED) {
                if (int rc = hip_ok(hipEventRecord(S->stepped[t & 1], A), "hipEventRecord")) return rc;
                S->have_stepped[t & 1] = true;
            }
        }
        if (use_ring) L->ring_pushed += L->window;
        const int64_t ring_len = L->ring_pushed < cap ? L->ring_pushed : cap;
        hipStream_t U = L->mode == SHEMS_LOOP_ORDERED ? A : B;
        if (L->mode == SHEMS_LOOP_PIPELINED) {
            // replay(t) may not touch the slots step t is writing and republishes pub[(t + 1) & 1], last read by act(t - 1)
            if (S->have_stepped[(t - 1) & 1])
                if (int rc = hip_ok(hipStreamWaitEvent(B, S->stepped[(t - 1) & 1], 0), "hipStreamWaitEvent")) return rc;
        } else if (L->mode == SHEMS_LOOP_PIPELINED_EXACT) {
            if (int rc = hip_ok(hipStreamWaitEvent(B, S->stepped[t & 1], 0), "hipStreamWaitEvent")) return rc;
        }
        for (int u = 0; u < L->updates_per_step; ++u) {
            const bool last = u == L->updates_per_step - 1;
            const bool excl = L->mode == SHEMS_LOOP_PIPELINED && use_ring;
            float *pub = (L->mode != SHEMS_LOOP_ORDERED && last) ? L->actor_pub[(t + 1) & 1] : nullptr;
            if (int rc = shems_ddpg_update(&L->ddpg, &L->ring, ring_len, L->sample_seed, (uint32_t)(L->updates & 0xFFFFFFFFll),
                                           excl ? w.pos : 0, excl ? w.count : 0, L->eta_crit, L->bp_crit[0], L->bp_crit[1], L->eta_act,
                                           L->bp_act[0], L->bp_act[1], pub, U))
                return rc;
            L->bp_crit[0] *= 0.9; L->bp_crit[1] *= 0.999;       // Flux ADAM: beta^t advanced after every step (Float64)
            L->bp_act[0] *= 0.9;  L->bp_act[1] *= 0.999;
            L->updates += 1;
        }
        if (L->mode != SHEMS_LOOP_ORDERED) {
            if (int rc = hip_ok(hipEventRecord(S->updated[t & 1], B), "hipEventRecord")) return rc;
            S->have_updated[t & 1] = true;
        }
        L->t += 1;
    }
    return SHEMS_OK;
}

int shems_train_loop_join(shems_train_loop *L, void *stream, void *stream2)
{
    if (!L) return set_error(SHEMS_ERR_ARG, "shems_train_loop_join: loop is NULL");
    if (!L->sync || L->mode == SHEMS_LOOP_ORDERED) return SHEMS_OK;
    LoopSync *S = static_cast<LoopSync *>(L->sync);
    (void)stream2;
    for (int i = 0; i < 2; ++i)
        if (S->have_updated[i])
            if (int rc = hip_ok(hipStreamWaitEvent((hipStream_t)stream, S->updated[i], 0), "hipStreamWaitEvent")) return rc;
    return SHEMS_OK;
}

int shems_train_loop_release(shems_train_loop *L)
{
    if (!L) return set_error(SHEMS_ERR_ARG, "shems_train_loop_release: loop is NULL");
    if (L->sync) {
        LoopSync *S = static_cast<LoopSync *>(L->sync);
        for (int i = 0; i < 2; ++i) {
            if (S->stepped[i]) (void)hipEventDestroy(S->stepped[i]);
            if (S->updated[i]) (void)hipEventDestroy(S->updated[i]);
        }
        delete S;
        L->sync = nullptr;
    }
    return SHEMS_OK;
}

}  // extern "C"
