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
//
// What carries S / W / U: a dependency between two HIP queues is expensive on this stack (tools/xqueue_sync.hip, MI355X, ROCm 7.2: an
// event record + hipStreamWaitEvent costs the waiting queue 9.6 us, 2.9 us even when already satisfied; hipStreamWriteValue64 +
// hipStreamWaitValue64 on signal memory 4.9 us), and the loop pays two per vector step, so the counters are stream memory operations
// where the device supports them (hipDeviceAttributeCanUseStreamWaitValue): two 8-byte signal words holding "steps stepped" and
// "steps updated", written by the producing queue behind its launches and waited for (>=) by the other.  SHEMS_LOOP_SYNC=events keeps
// the event form (A/B runs; devices without stream memory operations use it anyway).
//
// SHEMS_LOOP_SYNC=device (opt-in; SHEMS_LOOP_PIPELINED at <= 16 384 envs): NO queue-level dependency at all (DevSync, shems_internal.h).
// Even the cheapest queue-level wait is a one-thread kernel plus two launch gaps (~5 us per hop), and the update's chain carries two of
// them per step.  In this form both queues run free and the dependent LAUNCHES wait in the kernel: the workgroups of act(t)
// poll a count of finished K5 workgroups of replay(t - 1) before they read actor_pub[t & 1]; the workgroups of K1 of replay(t) poll a
// count of finished workgroups of act(t - 1) before they sample.  Waiting workgroups must not keep their producers off the CUs, hence
// the size limit and the forms: up to 8 192 envs the two-workgroups-per-tile form (92 KB of LDS, one workgroup per CU at a time),
// up to 16 384 the 64-env tiles (78 KB, <= 256 workgroups) -- either leaves room for a 64-KB workgroup of the update on every CU
// (or, for 64-env tiles that landed two to a CU, on the CUs that got none).  Larger batches fill every CU's LDS with step workgroups:
// they keep the queue-level form.  Every wait is bounded (it gives up after tens of ms and counts itself: shems_ddpg_sync_timeouts).
// Measured (profiles/r04_overlap_forms.json, r04_overlap_timeline_4096.txt): correct -- the bytes of the host-side pipelined loop -- and
// the update's five launches do run gap-free beside the waiting step kernel, but once the step kernel starts computing it and K1 / K2
// slow each other down on the shared CUs (K1 6 -> 26 us, the step kernel 17 -> 30 us): 55.5 us per step at 4 096 envs against 54.3 with
// stream memory operations and 52.9 in program order.  Not the default.
#include <cstdlib>
#include <cstring>
#include <new>
#include "shems_internal.h"

using namespace shems;

namespace {
struct LoopSync {
    int dev_id = -1;
    bool values = false;                            // stream memory operations instead of events
    bool device = false;                            // in-kernel waits on arrival counts (DevSync): no queue-level dependency
    unsigned long long *d_words = nullptr;          // device: per direction a counter line + kDevFlagCopies flag lines (step words first, then update words)
    unsigned long long *cnt(int dir) const { return d_words + dir * (1 + kDevFlagCopies) * kDevLineWords; }
    unsigned long long *flags(int dir) const { return cnt(dir) + kDevLineWords; }
    unsigned long long step_wgs = 0, upd_wgs = 0;   // workgroups enqueued so far (= what the words will hold when they are done)
    uint64_t *n_stepped = nullptr, *n_updated = nullptr;   // signal memory: S(t) / W(t) = "n_stepped >= t + 1", U(t) = "n_updated >= t + 1"
    int64_t base = 0;                               // loop->t when the words were last zeroed (values are counted from there)
    hipEvent_t stepped[2] = {nullptr, nullptr};     // S(t) / W(t): the ring holds everything replay(t + 1) / replay(t) may sample
    hipEvent_t updated[2] = {nullptr, nullptr};     // U(t): replay(t) done, pub[(t + 1) & 1] published
    bool have_updated[2] = {false, false}, have_stepped[2] = {false, false};
    int64_t last_stepped = -1, last_updated = -1;   // newest t whose S / U has been enqueued
    int64_t next_t = -1;                            // loop->t this record expects at the next call (-1: none yet)

    // The caller moved loop->t (a restored snapshot, another run on the same record): nothing in flight may refer to the old numbering.
    int restart(hipStream_t a, hipStream_t b, int64_t t)
    {
        if (int rc = hip_ok(hipStreamSynchronize(a), "hipStreamSynchronize")) return rc;
        if (int rc = hip_ok(hipStreamSynchronize(b), "hipStreamSynchronize")) return rc;
        if (values) {
            if (int rc = hip_ok(hipMemset(n_stepped, 0, 8), "hipMemset")) return rc;
            if (int rc = hip_ok(hipMemset(n_updated, 0, 8), "hipMemset")) return rc;
        }
        if (device) {
            if (int rc = hip_ok(hipMemset(d_words, 0, kDevSyncBytes), "hipMemset")) return rc;
            step_wgs = upd_wgs = 0;
        }
        base = t; last_stepped = last_updated = -1;
        have_stepped[0] = have_stepped[1] = have_updated[0] = have_updated[1] = false;
        return SHEMS_OK;
    }

    // "the producing queue has finished step t's act (or window) launch" / "... replay(t)"
    int signal_stepped(hipStream_t q, int64_t t)
    {
        last_stepped = t;
        if (values) return hip_ok(hipStreamWriteValue64(q, n_stepped, (uint64_t)(t - base + 1), 0), "hipStreamWriteValue64");
        have_stepped[t & 1] = true;
        return hip_ok(hipEventRecord(stepped[t & 1], q), "hipEventRecord");
    }
    int signal_updated(hipStream_t q, int64_t t)
    {
        last_updated = t;
        if (values) return hip_ok(hipStreamWriteValue64(q, n_updated, (uint64_t)(t - base + 1), 0), "hipStreamWriteValue64");
        have_updated[t & 1] = true;
        return hip_ok(hipEventRecord(updated[t & 1], q), "hipEventRecord");
    }
    int wait_stepped(hipStream_t q, int64_t t)      // no-op when S(t) was never signalled (the loop's first steps)
    {
        if (t < base || t > last_stepped) return SHEMS_OK;
        if (values) return hip_ok(hipStreamWaitValue64(q, n_stepped, (uint64_t)(t - base + 1), hipStreamWaitValueGte, ~0ull), "hipStreamWaitValue64");
        return have_stepped[t & 1] ? hip_ok(hipStreamWaitEvent(q, stepped[t & 1], 0), "hipStreamWaitEvent") : SHEMS_OK;
    }
    int wait_updated(hipStream_t q, int64_t t)
    {
        if (t < base || t > last_updated) return SHEMS_OK;
        if (values) return hip_ok(hipStreamWaitValue64(q, n_updated, (uint64_t)(t - base + 1), hipStreamWaitValueGte, ~0ull), "hipStreamWaitValue64");
        return have_updated[t & 1] ? hip_ok(hipStreamWaitEvent(q, updated[t & 1], 0), "hipStreamWaitEvent") : SHEMS_OK;
    }
};

void free_sync(LoopSync *s)
{
    for (int i = 0; i < 2; ++i) {
        if (s->stepped[i]) (void)hipEventDestroy(s->stepped[i]);
        if (s->updated[i]) (void)hipEventDestroy(s->updated[i]);
    }
    if (s->n_stepped) (void)hipFree(s->n_stepped);
    if (s->n_updated) (void)hipFree(s->n_updated);
    if (s->d_words) (void)hipFree(s->d_words);
    delete s;
}

// The in-kernel form serves SHEMS_LOOP_PIPELINED at <= 16 384 envs when asked for; everything else is queue-level.
bool want_device(const shems_train_loop *L)
{
    const char *how = getenv("SHEMS_LOOP_SYNC");
    return how && !strcmp(how, "device") && L->mode == SHEMS_LOOP_PIPELINED && L->view.n_envs <= 16384;
}

int sync_of(shems_train_loop *L, LoopSync **out, hipStream_t a, hipStream_t b)
{
    if (L->sync && static_cast<LoopSync *>(L->sync)->device != want_device(L)) {
        // the caller changed the mode (or the batch) on a live record: the other form's objects do not exist -- drain and start over
        if (int rc = hip_ok(hipStreamSynchronize(a), "hipStreamSynchronize")) return rc;
        if (int rc = hip_ok(hipStreamSynchronize(b), "hipStreamSynchronize")) return rc;
        free_sync(static_cast<LoopSync *>(L->sync));
        L->sync = nullptr;
    }
    if (!L->sync) {
        LoopSync *s = new (std::nothrow) LoopSync;
        if (!s) return set_error(SHEMS_ERR_NOMEM, "shems_train_steps: out of host memory");
        if (int rc = hip_ok(hipGetDevice(&s->dev_id), "hipGetDevice")) { delete s; return rc; }
        int can = 0;
        const char *how = getenv("SHEMS_LOOP_SYNC");
        if (want_device(L)) {
            if (int rc = hip_ok(hipMalloc((void **)&s->d_words, kDevSyncBytes), "hipMalloc(sync words)")) { delete s; return rc; }
            if (int rc = hip_ok(hipMemset(s->d_words, 0, kDevSyncBytes), "hipMemset(sync words)")) { free_sync(s); return rc; }
            s->device = true;
        }
        if (!s->device && !(how && !strcmp(how, "events")) && hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, s->dev_id) == hipSuccess && can) {
            // signal memory is handed out 8 bytes at a time; zeroed synchronously before any queue looks at it
            if (hipExtMallocWithFlags((void **)&s->n_stepped, 8, hipMallocSignalMemory) == hipSuccess &&
                hipExtMallocWithFlags((void **)&s->n_updated, 8, hipMallocSignalMemory) == hipSuccess &&
                hipMemset(s->n_stepped, 0, 8) == hipSuccess && hipMemset(s->n_updated, 0, 8) == hipSuccess)
                s->values = true;
            else
                (void)hipGetLastError();
        }
        s->base = L->t;
        if (!s->values && !s->device)
            for (int i = 0; i < 2; ++i) {
                if (int rc = hip_ok(hipEventCreateWithFlags(&s->stepped[i], hipEventDisableTiming), "hipEventCreate")) { free_sync(s); return rc; }
                if (int rc = hip_ok(hipEventCreateWithFlags(&s->updated[i], hipEventDisableTiming), "hipEventCreate")) { free_sync(s); return rc; }
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
        if (int rc = sync_of(L, &S, A, B)) return rc;
    const int64_t n = L->view.n_envs, cap = L->ring.capacity;
    if (S && S->next_t >= 0 && S->next_t != L->t)
        if (int rc = S->restart(A, B, L->t)) return rc;
    for (int64_t it = 0; it < k; ++it) {
        const int64_t t = L->t;
        if (S && S->device) {
            // both queues run free; the launches carry their dependencies (DevSync)
            if (t > 0 && t % L->ep_len == 0) {
                L->episode += 1;
                if (int rc = shems_reset_seeded_dev(&L->view, L->env_seed, L->episode, A)) return rc;
            }
            shems_act_params p = L->act;
            p.tick = (uint32_t)(t & 0xFFFFFFFFll);
            p.actor = L->actor_pub[t & 1];
            const bool use_ring = L->window > 0 && cap > 0;
            shems_ring_window w = {use_ring ? L->ring_pushed % cap : 0, use_ring ? L->window : 0, use_ring ? (t * L->window) % n : 0};
            unsigned *tmo = ddpg_timeout_word(&L->ddpg);
            const unsigned long long step_wgs_before = S->step_wgs;           // act(0) .. act(t - 1)
            // act(t): replay(t - 1) published pub[t & 1]; its workgroups arrive at the step counter
            const bool split = n <= 8192;
            const int64_t grid = split ? 2 * ((n + 31) / 32) : (n + 63) / 64;
            S->step_wgs += (unsigned long long)grid;
            DevSync sa = {S->flags(1), S->upd_wgs, S->cnt(0), S->flags(0), S->step_wgs, tmo};
            int64_t grid_l = 0;
            if (int rc = act_step_sync(&L->view, &p, L->rewards_f32, use_ring ? &L->ring : nullptr, use_ring ? &w : nullptr, sa, split, &grid_l, A)) return rc;
            if (grid_l != grid) return set_error(SHEMS_ERR_STATE, "shems_train_steps: the step launch has %lld workgroups, %lld expected", (long long)grid_l, (long long)grid);
            if (use_ring) L->ring_pushed += L->window;
            const int64_t ring_len = L->ring_pushed < cap ? L->ring_pushed : cap;
            for (int u = 0; u < L->updates_per_step; ++u) {
                const bool last = u == L->updates_per_step - 1;
                // K1 of the step's first update: act(t - 1) has finished (ring rows; pub[(t + 1) & 1] no longer read); only the step's last
                // update publishes and arrives
                if (last) S->upd_wgs += (unsigned long long)ddpg_last_launch_grid();
                static const bool nowait1 = []() { const char *e = getenv("SHEMS_LOOP_DIAG_NOWAIT_K1"); return e && atoi(e) == 1; }();   // timing diagnostics only: drops a real dependency
                DevSync first = {u == 0 && !nowait1 ? S->flags(0) : nullptr, step_wgs_before, nullptr, nullptr, 0, tmo};
                DevSync lastl = {nullptr, 0, last ? S->cnt(1) : nullptr, S->flags(1), S->upd_wgs, tmo};
                if (int rc = ddpg_update_sync(&L->ddpg, &L->ring, ring_len, L->sample_seed, (uint32_t)(L->updates & 0xFFFFFFFFll), use_ring ? w.pos : 0,
                                              use_ring ? w.count : 0, L->eta_crit, L->bp_crit[0], L->bp_crit[1], L->eta_act, L->bp_act[0], L->bp_act[1],
                                              last ? L->actor_pub[(t + 1) & 1] : nullptr, first, lastl, B))
                    return rc;
                L->bp_crit[0] *= 0.9; L->bp_crit[1] *= 0.999;
                L->bp_act[0] *= 0.9;  L->bp_act[1] *= 0.999;
                L->updates += 1;
            }
            S->last_updated = t;
            L->t += 1;
            S->next_t = L->t;
            continue;
        }
        if (L->mode != SHEMS_LOOP_ORDERED)
            if (int rc = S->wait_updated(A, t - 1)) return rc;                  // pub[t & 1] is replay(t - 1)'s
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
            if (int rc = S->signal_stepped(A, t)) return rc;
            for (int i = 0; i < rest.k; ++i)
                if (int rc = shems_act_step_range_dev(&L->view, &p, rest.lo[i], rest.cnt[i], L->rewards_f32, &L->ring, &w, A)) return rc;
        } else {
            if (int rc = shems_act_step_dev(&L->view, &p, nullptr, nullptr, L->rewards_f32, nullptr, nullptr, use_ring ? &L->ring : nullptr,
                                            use_ring ? &w : nullptr, A))
                return rc;
            if (L->mode == SHEMS_LOOP_PIPELINED)
                if (int rc = S->signal_stepped(A, t)) return rc;
        }
        if (use_ring) L->ring_pushed += L->window;
        const int64_t ring_len = L->ring_pushed < cap ? L->ring_pushed : cap;
        hipStream_t U = L->mode == SHEMS_LOOP_ORDERED ? A : B;
        if (L->mode == SHEMS_LOOP_PIPELINED) {
            // replay(t) may not touch the slots step t is writing and republishes pub[(t + 1) & 1], last read by act(t - 1)
            if (int rc = S->wait_stepped(B, t - 1)) return rc;
        } else if (L->mode == SHEMS_LOOP_PIPELINED_EXACT) {
            if (int rc = S->wait_stepped(B, t)) return rc;
        }
        for (int u = 0; u < L->updates_per_step; ++u) {
            const bool last = u == L->updates_per_step - 1;
            const bool excl = L->mode == SHEMS_LOOP_PIPELINED && use_ring;
            float *pub = (L->mode != SHEMS_LOOP_ORDERED && last) ? L->actor_pub[(t + 1) & 1] : nullptr;
            if (L->dp) {                                          // replicas: the split form with both all-reduces in this stream
                if (int rc = shems_ddpg_update_dp(&L->ddpg, &L->ring, ring_len, L->sample_seed, (uint32_t)(L->updates & 0xFFFFFFFFll), 0, 0,
                                                  L->eta_crit, L->bp_crit[0], L->bp_crit[1], L->eta_act, L->bp_act[0], L->bp_act[1], nullptr, L->dp, U))
                    return rc;
            } else if (int rc = shems_ddpg_update(&L->ddpg, &L->ring, ring_len, L->sample_seed, (uint32_t)(L->updates & 0xFFFFFFFFll),
                                           excl ? w.pos : 0, excl ? w.count : 0, L->eta_crit, L->bp_crit[0], L->bp_crit[1], L->eta_act,
                                           L->bp_act[0], L->bp_act[1], pub, U))
                return rc;
            L->bp_crit[0] *= 0.9; L->bp_crit[1] *= 0.999;       // Flux ADAM: beta^t advanced after every step (Float64)
            L->bp_act[0] *= 0.9;  L->bp_act[1] *= 0.999;
            L->updates += 1;
        }
        if (L->mode != SHEMS_LOOP_ORDERED)
            if (int rc = S->signal_updated(B, t)) return rc;
        L->t += 1;
        if (S) S->next_t = L->t;
    }
    return SHEMS_OK;
}

int shems_train_loop_join(shems_train_loop *L, void *stream, void *stream2)
{
    if (!L) return set_error(SHEMS_ERR_ARG, "shems_train_loop_join: loop is NULL");
    if (!L->sync || L->mode == SHEMS_LOOP_ORDERED) return SHEMS_OK;
    LoopSync *S = static_cast<LoopSync *>(L->sync);
    if (S->device) {                                   // no queue-level object exists: one event, once per join
        hipEvent_t e;
        if (int rc = hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate")) return rc;
        int rc = hip_ok(hipEventRecord(e, (hipStream_t)stream2), "hipEventRecord");
        if (!rc) rc = hip_ok(hipStreamWaitEvent((hipStream_t)stream, e, 0), "hipStreamWaitEvent");
        (void)hipEventDestroy(e);
        return rc;
    }
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
