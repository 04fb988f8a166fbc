"""replay() (five launches, stream B) timed alone and beside a small-code MFMA-bound co-runner on stream A (tools/corun_probe.hip): is the slowdown of the
update beside the fused step kernel a matter of the matrix pipe / issue slots, or of that kernel in particular?
    hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o /tmp/libcorun.so tools/corun_probe.hip && python3 tools/corun_probe.py /tmp/libcorun.so"""
import ctypes as C, importlib, json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
PKG = "master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"
S = importlib.import_module(PKG); D = importlib.import_module(PKG + ".ddpg")
co = C.CDLL(sys.argv[1])
co.corun_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
wl = D.TrainWorkload(S, torch, 4096, seed=3, updates=1, loop="host")
out = torch.zeros(1024, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
res = {}
def timed(label, iters, use_lds, lds_bytes=90 * 1024, grid=256, reps=40):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if iters:
            with torch.cuda.stream(sa):
                c0.record(sa)
                assert co.corun_launch(C.c_void_p(sa.cuda_stream), C.c_void_p(out.data_ptr()), iters, use_lds, lds_bytes, grid) == 0
                c1.record(sa)
        with torch.cuda.stream(sb):
            e0.record(sb)
            wl.agent.replay(wl.ring)
            e1.record(sb)
        torch.cuda.synchronize()
        ts.append((e0.elapsed_time(e1) * 1e3, c0.elapsed_time(c1) * 1e3 if iters else 0.0))
    ts.sort()
    res[label] = {"replay_us_median": ts[len(ts) // 2][0], "corunner_us_median": sorted(t[1] for t in ts)[len(ts) // 2]}
    print(label, res[label], flush=True)
for _ in range(50):
    wl.agent.replay(wl.ring)
timed("replay alone", 0, 0)
timed("beside MFMA loop (no LDS reads), ~100 us", 1500, 0)
timed("beside MFMA loop + LDS operand reads, ~100 us", 1500, 1)
timed("beside MFMA loop, 128 workgroups only", 1500, 0, grid=128)
timed("beside MFMA loop, 40 KB LDS (two co-runner workgroups could share a CU)", 1500, 0, lds_bytes=40 * 1024)
print(json.dumps(res))
