"""What bounds the ADAM + soft-update stream of the grouped update?  Times, on 400 learner slabs: (a) the plain elementwise sweep of the
latency form's split path (k_adam_soft: 20 B in, 16 B out per element, contiguous 16-byte accesses), (b) a device copy of the same
number of bytes, (c) the throughput form's whole update.  HIP events over back-to-back launches."""
import ctypes as C
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import util as U

S = U.pkg()
G = importlib.import_module(U.PKG_NAME + ".group")
_capi = S._capi
L, E = 400, 128
tab = S.tables.synthetic_table("train", 98)
env = S.ShemsBatch(L * E, 72, [tab], [S.make_config(98, 0, tab.shape[0])]).use_torch_stream()
grp = G.LearnerGroup(L, E, seed=21, rng_seed=77, capacity=2400, form="throughput")
grp.populate_memory(env, seed=5)
grp.min_max_buffer()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


a0, g = grp.learners[0], grp.struct()
d = a0._ddpg_args()
st = grp._stream()
out = {}
sweep = lambda: _capi.check(grp.L.shems_ddpg_group_critic_apply(C.byref(d), C.byref(g), 1e-3, 0.9, 0.999, st))
out["adam_sweep_us"] = timeit(sweep)
out["adam_sweep_bytes"] = 36 * 129001 * L
out["adam_sweep_tbs"] = out["adam_sweep_bytes"] / out["adam_sweep_us"] / 1e6
src = torch.empty(out["adam_sweep_bytes"] // 8, dtype=torch.float32, device="cuda")
dst = torch.empty_like(src)
out["copy_us"] = timeit(lambda: dst.copy_(src))
out["copy_tbs"] = out["adam_sweep_bytes"] / out["copy_us"] / 1e6
out["tp_update_us"] = timeit(grp.replay, 10)
print(json.dumps(out))
