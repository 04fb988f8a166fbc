"""A late peer in the direct gradient exchange (SHEMS_DP=direct, csrc/shems_ddpg.hip: k_adam_xchg) -- run by tests/test_bench_gpu.py.

    python tests/dp_direct_late_peer.py <wait_ms> <sleep_s>

Two rank processes on device 0 (the IPC rehearsal form) train 6 data-parallel vector steps, then rank 1 stalls on the HOST for <sleep_s>
seconds while rank 0 enqueues 6 more steps (its exchange sweeps wait, in the kernel, for rank 1's slices), then rank 1 catches up.
Prints ONE line: "LATE ok <crc0> <crc1>" when both ranks finished (the waits were long enough: the replicas must be identical), or
"LATE poisoned <ranks that failed>" when every rank failed loudly with SHEMS_ERR_STATE / a finish() error.  Any other outcome -- one rank
finishing while the other failed, different checksums -- exits non-zero.  A failed rank EXITS; nothing re-executes a process that has
touched the GPU."""
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rank_main(rank, wait_ms, sleep_s, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import torch
    import torch.distributed as dist
    import util as U
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SHEMS_DP="direct")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    S = U.pkg()
    D = importlib.import_module(U.PKG_NAME + ".ddpg")
    wl = D.TrainWorkload(S, torch, 2048, seed=77, updates=1, dist=dist)
    assert wl.agent.sync.direct and wl.loop == "native", (wl.agent.sync.direct, wl.loop)
    L = wl.agent.L
    L.shems_dp_direct_set_wait_ms.argtypes = [C.c_void_p, C.c_int64]
    L.shems_dp_direct_set_wait_ms.restype = C.c_int
    assert L.shems_dp_direct_set_wait_ms(wl.agent.sync.native, int(wait_ms)) == 0
    status, err = "ok", ""
    try:
        wl.steps(6)
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 1:
            time.sleep(sleep_s)                     # a descheduled process / a rank busy writing a snapshot
        wl.steps(6)
        torch.cuda.synchronize()
        wl.steps(2)                                 # a poisoned record answers the NEXT call with SHEMS_ERR_STATE
        torch.cuda.synchronize()
    except Exception as e:                          # noqa: BLE001
        status, err = "failed", repr(e)[:300]
    try:
        wl.finish()                                 # votes: raises on EVERY rank if any rank saw a problem
    except Exception as e:                          # noqa: BLE001
        status, err = "failed", err or repr(e)[:300]
    print("RANK " + json.dumps({"rank": rank, "status": status, "crc": wl._learner_crc(), "poisoned": wl.dp_poisoned(), "err": err}), flush=True)
    wl.close()
    dist.destroy_process_group()
    sys.exit(0 if status == "ok" else 3)


def main():
    if len(sys.argv) >= 2 and sys.argv[1] == "--rank":
        return rank_main(int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5]))
    wait_ms, sleep_s = float(sys.argv[1]), float(sys.argv[2])
    port = 29700 + os.getpid() % 200
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), str(wait_ms), str(sleep_s), str(port)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    recs = []
    for (o, e), p in zip(outs, procs):
        line = [l for l in o.splitlines() if l.startswith("RANK ")]
        if not line:
            print("LATE broken: a rank printed nothing", e[-1500:], file=sys.stderr)
            sys.exit(2)
        recs.append(json.loads(line[-1][5:]))
    st = [r["status"] for r in recs]
    if st == ["ok", "ok"]:
        if recs[0]["crc"] != recs[1]["crc"] or any(r["poisoned"] for r in recs):
            print("LATE diverged", recs, file=sys.stderr)
            sys.exit(4)
        print("LATE ok", recs[0]["crc"], recs[1]["crc"])
    elif st == ["failed", "failed"]:
        print("LATE poisoned", [r["rank"] for r in recs if r["poisoned"]], "|", recs[0]["err"][:120])
    else:
        print("LATE silent third outcome", recs, file=sys.stderr)
        sys.exit(5)


if __name__ == "__main__":
    main()
