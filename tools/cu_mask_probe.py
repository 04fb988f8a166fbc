"""What would an overlapped vector step cost if the update had CUs of its own?  (replay(t) on a second stream while the
fused kernel of step t runs; without a reservation the update's workgroups find no free CU slot between the fused kernel's rounds.)
Streams created with hipExtStreamCreateWithCUMask: the fused kernel on 256 - R CUs, the five update launches on the other R (mask bit i
= CU i / 8 of XCD i % 8, so both sets are spread evenly over the 8 XCDs).  Timing only: the pipelined mode's ordering (5b) is kept as it is.
    [CU_MASK_R=64,96,128] python3 tools/cu_mask_probe.py [envs]"""
import ctypes as C, importlib, json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
PKG = "master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"
S = importlib.import_module(PKG); D = importlib.import_module(PKG + ".ddpg")
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int

def masked(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (w * 32 + b) in bits) for w in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
out = {}
def run(label, overlap, R):
    torch.cuda.set_stream(torch.cuda.default_stream())
    wl = D.TrainWorkload(S, torch, n, seed=1231, updates=1, overlap=overlap)
    if R is not None:
        main = masked(set(range(0, 256 - R))) if R > 0 else masked(set(range(256)))
        if overlap:
            wl.upd_stream = masked(set(range(256 - R, 256))) if R > 0 else masked(set(range(256)))
        torch.cuda.synchronize()
        torch.cuda.set_stream(main)
    for _ in range(3000): wl.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1440): wl.step()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 1440 * 1e6
    wl.finish()
    out[label] = {"us_per_step": us, "env_steps_per_s": n / us * 1e6}
    print(label, "%.1f us/step" % us, "%.1f M env-steps/s" % (n / us), flush=True)
    torch.cuda.set_stream(torch.cuda.default_stream())
    del wl

run("sequential, default stream", False, None)
run("sequential, fused kernel + update on a stream masked to 224 CUs", False, 32)
run("pipelined (5b), two unmasked streams", True, None)
for R in [int(x) for x in os.environ.get("CU_MASK_R", "16,24,32,40,48,64").split(",")]:
    run(f"pipelined, update on {R} CUs / fused kernel on {256 - R}", True, R)
print(json.dumps(out, indent=1))
