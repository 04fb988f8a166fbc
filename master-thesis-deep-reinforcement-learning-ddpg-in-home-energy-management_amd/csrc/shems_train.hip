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
// What carries S / W / U: stream memory operations -- two 8-byte words of signal memory holding "steps stepped" and "steps updated",
// written by the producing queue behind its launches (hipStreamWriteValue64) and waited for (>=) by the other (hipStreamWaitValue64):
// 4.9 us per hop on this stack (tools/xqueue_sync.hip, MI355X, ROCm 7.2), against 9.6 us for an event record + hipStreamWaitEvent.
// Round 4 also built an event form and a form with NO queue-level dependency (in-kernel waits on counts of finished producer
// workgroups); both were measured slower (profiles/r04_overlap_forms.json, r04_overlap_timeline_4096.txt, r04_xqueue_sync.txt) and
// were removed in round 5 -- as every pipelined form is slower than program order on this part (52.9 us per step at 4 096 envs in
// program order, 54.3 with stream memory operations), SHEMS_LOOP_ORDERED stays the default and the pipelined modes stay an option.
// A device without stream memory operations (hipDeviceAttributeCanUseStreamWaitValue) gets SHEMS_ERR_STATE for a pipelined mode.
#include <cstdlib>
#include <cstring>
#include <new>
#include "shems_internal.h"

using namespace shems;

namespace {
struct LoopSync {
    uint64_t *n_stepped = nullptr, *n_updated = nullptr;   // signal memory: S(t) / W(t) = "n_stepped >= t + 1", U(t) = "n_updated >= t + 1"
    int64_t base = 0;                               // loop->t when the words were last zeroed (values are counted from there)
    int64_t last_stepped = -1, last_updated = -1;   // newest t whose S / U has been enqueued
    int64_t next_t = -1;                            // loop->t this record expects at the next call (-1: none yet)

    // The caller moved loop->t (a restored snapshot, another run on the same record), or a call failed half-way: nothing in flight may
    // refer to the old numbering.
    int restart(hipStream_t a, hipStream_t b, int64_t t)
    {
        if (int rc = hip_ok(hipStreamSynchronize(a), "hipStreamSynchronize")) return rc;
        if (int rc = hip_ok(hipStreamSynchronize(b), "hipStreamSynchronize")) return rc;
        if (int rc = hip_ok(hipMemset(n_stepped, 0, 8), "hipMemset")) return rc;
        if (int rc = hip_ok(hipMemset(n_updated, 0, 8), "hipMemset")) return rc;
        base = t; last_stepped = last_updated = -1;
        return SHEMS_OK;
    }
    // "the producing queue has finished step t's act (or window) launch" / "... replay(t)"
    int signal_stepped(hipStream_t q, int64_t t)
    {
        last_stepped = t;
        return hip_ok(hipStreamWriteValue64(q, n_stepped, (uint64_t)(t - base + 1), 0), "hipStreamWriteValue64");
    }
    int signal_updated(hipStream_t q, int64_t t)
    {
        last_updated = t;
        return hip_ok(hipStreamWriteValue64(q, n_updated, (uint64_t)(t - base + 1), 0), "hipStreamWriteValue64");
    }
    int wait_stepped(hipStream_t q, int64_t t)      // no-op when S(t) was never signalled (the loop's first steps)
    {
        if (t < base || t > last_stepped) return SHEMS_OK;
        return hip_ok(hipStreamWaitValue64(q, n_stepped, (uint64_t)(t - base + 1), hipStreamWaitValueGte, ~0ull), "hipStreamWaitValue64");
    }
    int wait_updated(hipStream_t q, int64_t t)
    {
        if (t < base || t > last_updated) return SHEMS_OK;
        return hip_ok(hipStreamWaitValue64(q, n_updated, (uint64_t)(t - base + 1), hipStreamWaitValueGte, ~0ull), "hipStreamWaitValue64");
    }
};

void free_sync(LoopSync *s)
{
    if (s->n_stepped) (void)hipFree(s->n_stepped);
    if (s->n_updated) (void)hipFree(s->n_updated);
    delete s;
}

int sync_of(shems_train_loop *L, LoopSync **out)
{
    if (!L->sync) {
        int dev = 0, can = 0;
        if (int rc = hip_ok(hipGetDevice(&dev), "hipGetDevice")) return rc;
        if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev) != hipSuccess || !can) {
            (void)hipGetLastError();
            return set_error(SHEMS_ERR_STATE, "shems_train_steps: the pipelined modes need stream memory operations (hipStreamWaitValue64), which this device does not offer; use SHEMS_LOOP_ORDERED");
        }
        LoopSync *s = new (std::nothrow) LoopSync;
        if (!s) return set_error(SHEMS_ERR_NOMEM, "shems_train_steps: out of host memory");
        // signal memory is handed out 8 bytes at a time; zeroed synchronously before any queue looks at it
        if (hipExtMallocWithFlags((void **)&s->n_stepped, 8, hipMallocSignalMemory) != hipSuccess ||
            hipExtMallocWithFlags((void **)&s->n_updated, 8, hipMallocSignalMemory) != hipSuccess ||
            hipMemset(s->n_stepped, 0, 8) != hipSuccess || hipMemset(s->n_updated, 0, 8) != hipSuccess) {
            (void)hipGetLastError();
            free_sync(s);
            return set_error(SHEMS_ERR_HIP, "shems_train_steps: could not allocate signal memory for the pipelined loop");
        }
        s->base = L->t;
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
    if (L->dp && L->mode != SHEMS_LOOP_ORDERED) return set_error(SHEMS_ERR_ARG, "shems_train_steps: data-parallel replicas run in program order");
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
    if (S && S->next_t != -1 && S->next_t != L->t)           // (-1: first call; -2: the previous call failed half-way)
        if (int rc = S->restart(A, B, L->t)) return rc;
    // A step's effect on the caller's record (episode, ring position, update count, ADAM beta powers, t) is committed only once the whole
    // step has been enqueued.  A failure half-way leaves the record at the last complete step and marks the queue-level words for a
    // restart (drain both queues, zero the words) at the next call: a retry never runs against a half-advanced ring position.
    auto failed = [&](int rc) { if (S) S->next_t = -2; return rc; };
    for (int64_t it = 0; it < k; ++it) {
        const int64_t t = L->t;
        int64_t episode = L->episode, ring_pushed = L->ring_pushed, updates = L->updates;
        double bpc[2] = {L->bp_crit[0], L->bp_crit[1]}, bpa[2] = {L->bp_act[0], L->bp_act[1]};
        if (L->mode != SHEMS_LOOP_ORDERED)
            if (int rc = S->wait_updated(A, t - 1)) return failed(rc);          // pub[t & 1] is replay(t - 1)'s
        if (t > 0 && t % L->ep_len == 0) {                                  // DDPG.jl:189-193: the next episode's reset!(env)
            episode += 1;
            if (int rc = shems_reset_seeded_dev(&L->view, L->env_seed, episode, A)) return failed(rc);
        }
        shems_act_params p = L->act;
        p.tick = (uint32_t)(t & 0xFFFFFFFFll);
        if (L->mode != SHEMS_LOOP_ORDERED) p.actor = L->actor_pub[t & 1];
        const bool use_ring = L->window > 0 && cap > 0;
        shems_ring_window w = {use_ring ? ring_pushed % cap : 0, use_ring ? L->window : 0, use_ring ? (t * L->window) % n : 0};
        if (L->mode == SHEMS_LOOP_PIPELINED_EXACT) {
            Ranges win, rest;
            window_ranges(n, w.offset, w.count, &win, &rest);
            for (int i = 0; i < win.k; ++i)
                if (int rc = shems_act_step_range_dev(&L->view, &p, win.lo[i], win.cnt[i], L->rewards_f32, &L->ring, &w, A)) return failed(rc);
            if (int rc = S->signal_stepped(A, t)) return failed(rc);
            for (int i = 0; i < rest.k; ++i)
                if (int rc = shems_act_step_range_dev(&L->view, &p, rest.lo[i], rest.cnt[i], L->rewards_f32, &L->ring, &w, A)) return failed(rc);
        } else {
            if (int rc = shems_act_step_dev(&L->view, &p, nullptr, nullptr, L->rewards_f32, nullptr, nullptr, use_ring ? &L->ring : nullptr,
                                            use_ring ? &w : nullptr, A))
                return failed(rc);
            if (L->mode == SHEMS_LOOP_PIPELINED)
                if (int rc = S->signal_stepped(A, t)) return failed(rc);
        }
        if (use_ring) ring_pushed += L->window;
        const int64_t ring_len = ring_pushed < cap ? ring_pushed : cap;
        hipStream_t U = L->mode == SHEMS_LOOP_ORDERED ? A : B;
        if (L->mode == SHEMS_LOOP_PIPELINED) {
            // replay(t) may not touch the slots step t is writing and republishes pub[(t + 1) & 1], last read by act(t - 1)
            if (int rc = S->wait_stepped(B, t - 1)) return failed(rc);
        } else if (L->mode == SHEMS_LOOP_PIPELINED_EXACT) {
            if (int rc = S->wait_stepped(B, t)) return failed(rc);
        }
        for (int u = 0; u < L->updates_per_step; ++u) {
            const bool last = u == L->updates_per_step - 1;
            const bool excl = L->mode == SHEMS_LOOP_PIPELINED && use_ring;
            float *pub = (L->mode != SHEMS_LOOP_ORDERED && last) ? L->actor_pub[(t + 1) & 1] : nullptr;
            if (L->dp) {                                          // replicas: the split form with both all-reduces in this stream
                if (int rc = shems_ddpg_update_dp(&L->ddpg, &L->ring, ring_len, L->sample_seed, (uint32_t)(updates & 0xFFFFFFFFll), 0, 0,
                                                  L->eta_crit, bpc[0], bpc[1], L->eta_act, bpa[0], bpa[1], nullptr, L->dp, U))
                    return failed(rc);
            } else if (int rc = shems_ddpg_update(&L->ddpg, &L->ring, ring_len, L->sample_seed, (uint32_t)(updates & 0xFFFFFFFFll),
                                           excl ? w.pos : 0, excl ? w.count : 0, L->eta_crit, bpc[0], bpc[1], L->eta_act,
                                           bpa[0], bpa[1], pub, U))
                return failed(rc);
            bpc[0] *= 0.9; bpc[1] *= 0.999;                     // Flux ADAM: beta^t advanced after every step (Float64)
            bpa[0] *= 0.9; bpa[1] *= 0.999;
            updates += 1;
        }
        if (L->mode != SHEMS_LOOP_ORDERED)
            if (int rc = S->signal_updated(B, t)) return failed(rc);
        L->episode = episode; L->ring_pushed = ring_pushed; L->updates = updates;
        L->bp_crit[0] = bpc[0]; L->bp_crit[1] = bpc[1]; L->bp_act[0] = bpa[0]; L->bp_act[1] = bpa[1];
        L->t = t + 1;
        if (S) S->next_t = L->t;
    }
    return SHEMS_OK;
}

int shems_train_loop_join(shems_train_loop *L, void *stream, void *stream2)
{
    if (!L) return set_error(SHEMS_ERR_ARG, "shems_train_loop_join: loop is NULL");
    if (!L->sync || L->mode == SHEMS_LOOP_ORDERED) return SHEMS_OK;
    LoopSync *S = static_cast<LoopSync *>(L->sync);
    return S->wait_updated((hipStream_t)stream, S->last_updated);        // replay(t) follows everything else the loop enqueued on stream2
}

int shems_train_loop_release(shems_train_loop *L)
{
    if (!L) return set_error(SHEMS_ERR_ARG, "shems_train_loop_release: loop is NULL");
    if (L->sync) {
        free_sync(static_cast<LoopSync *>(L->sync));
        L->sync = nullptr;
    }
    return SHEMS_OK;
}

}  // extern "C"
