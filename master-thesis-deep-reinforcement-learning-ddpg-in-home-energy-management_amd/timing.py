"""HIP-event timing of one kernel on the stream it is launched on (bench.py's roofline.achieved).

A pair of events around a single launch also times the command processor's hand-off (≈3-5 µs on this stack: a 6.6 µs kernel
reads 11.6 µs), so launches are timed in back-to-back groups and the group time is divided by its length; rocprofv3's
per-kernel average of the same command is the cross-check (profiles/).  `launch(i)` may be a whole vector step: the train workload
times groups of steps and groups of replay() alone and reports their difference, because k_act launched back to back runs 4-5 %
slower than the same kernel between the update's launches (ddpg.TrainWorkload.kernel_pass)."""
from __future__ import annotations


def time_launches(torch, launch, reps, group=8, before_group=None):
    """launch(i) enqueues launch i on the current stream.  Returns (avg_us, median_us, launches) per single launch.
    before_group(g, i0) runs untimed before group g (e.g. an env reset every few groups)."""
    ngroups = max(1, int(reps) // group)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(ngroups)]
    torch.cuda.synchronize()
    i = 0
    for g, (a, b) in enumerate(ev):
        if before_group is not None:
            before_group(g, i)
        a.record()
        for _ in range(group):
            launch(i)
            i += 1
        b.record()
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) * 1e3 / group for a, b in ev)
    return sum(us) / len(us), us[len(us) // 2], ngroups * group
