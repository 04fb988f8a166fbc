"""Can two processes on ONE device map each other's device memory (hipIpcGetMemHandle / hipIpcOpenMemHandle) and see each other's
stores while kernels run?  (the rehearsal vehicle a peer-mapped gradient exchange would need on a one-GPU box.)
Parent allocates a buffer, exports the handle, starts a child that opens it, writes a pattern with a kernel-free hipMemcpy and sets a flag word;
the parent polls the flag through its own mapping.  Prints one JSON line."""
import ctypes as C, json, os, subprocess, sys, time
hip = C.CDLL("libamdhip64.so")
class H(C.Structure):
    _fields_ = [("reserved", C.c_char * 64)]
def ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc}")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    raw = bytes.fromhex(sys.argv[2])
    h = H(); C.memmove(C.byref(h), raw, 64)
    p = C.c_void_p()
    rc = hip.hipIpcOpenMemHandle(C.byref(p), h, 1)      # hipIpcMemLazyEnablePeerAccess
    if rc != 0:
        print(json.dumps({"child_open_rc": rc})); sys.exit(0)
    host = (C.c_uint32 * 1024)(*range(1, 1025))
    ck(hip.hipMemcpy(C.c_void_p(p.value + 4096), host, 4096, 1), "child memcpy payload")
    flag = (C.c_uint32 * 1)(0xC0FFEE)
    ck(hip.hipMemcpy(p, flag, 4, 1), "child memcpy flag")
    ck(hip.hipDeviceSynchronize(), "child sync")
    hip.hipIpcCloseMemHandle(p)
    print(json.dumps({"child_open_rc": 0}))
    sys.exit(0)
out = {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
d = C.c_void_p()
ck(hip.hipMalloc(C.byref(d), 1 << 16), "hipMalloc")
ck(hip.hipMemset(d, 0, 1 << 16), "hipMemset")
h = H()
rc = hip.hipIpcGetMemHandle(C.byref(h), d)
out["get_handle_rc"] = rc
if rc == 0:
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", bytes(h).hex()], capture_output=True, text=True, timeout=120)
    out["child"] = (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]
    flag = (C.c_uint32 * 1)(0)
    t0 = time.time()
    while time.time() - t0 < 5:
        ck(hip.hipMemcpy(flag, d, 4, 2), "parent read flag")
        if flag[0] == 0xC0FFEE:
            break
        time.sleep(0.01)
    pay = (C.c_uint32 * 1024)()
    ck(hip.hipMemcpy(pay, C.c_void_p(d.value + 4096), 4096, 2), "parent read payload")
    out["flag_seen"] = flag[0] == 0xC0FFEE
    out["payload_ok"] = list(pay) == list(range(1, 1025))
print(json.dumps(out))
