"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately, with --kernel-trace only) into
profiles/<round>_pmc_counters_train.csv and profiles/pmc_traffic.json (read by bench.py for roofline.traffic).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <repo>/gpurun_out/pmc_fetch -- python3 <repo>/bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <repo>/gpurun_out/pmc_write -- python3 <repo>/bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write

Units and corrections follow MI355X_MICROARCH.md (HBM section): both counters are in KB; on gfx950 FETCH_SIZE reports half of a
wide coalesced read stream (doubled here), WRITE_SIZE is exact."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(d):
    acc = defaultdict(list)
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:                                   # gpurun_out/ keeps earlier runs: the newest pass only
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    acc = {}
    tag = "r03"
    outdir = os.path.join(ROOT, "profiles")        # on the GPU box pass --outdir=gpurun_out/...: only gpurun_out/ travels back
    dirs = []
    group = None                                   # --group=LEARNERS,ENVS: the passes are of bench.py --mode group (tools/profile_group.sh)
    for a in sys.argv[1:]:
        if a.startswith("--group="):
            group = tuple(int(x) for x in a.split("=", 1)[1].split(","))
        elif a.startswith("--round="):
            tag = a.split("=", 1)[1]
        elif a.startswith("--outdir="):
            outdir = a.split("=", 1)[1]
        else:
            dirs.append(a)
    for d in dirs:
        acc.update(collect(d))
    rows = sorted(acc.items())
    with open(os.path.join(outdir, f"{tag}_pmc_counters_train.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "Dispatches", "Mean_KB", "Min_KB", "Max_KB"])
        for (k, c), v in rows:
            if k.startswith("shems::") or "shems::" in k:
                w.writerow([k, c, len(v), sum(v) / len(v), min(v), max(v)])
    path = os.path.join(outdir, "pmc_traffic.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    if not doc and os.path.exists(os.path.join(ROOT, "profiles", "pmc_traffic.json")):
        doc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))          # records of the other modes are kept
    if group is not None:
        def gmean(name, ctr):
            v = [x for (k, c), vals in rows if c == ctr and name in k for x in vals]
            return sum(v) / len(v)
        # launches of one grouped replay() at >= 48 learners: prep, fwd<false, 4>, fwd<false, 2>, fwd<true, 2>, d1<., 2> x2, gw2 x2
        # (name prefixes: round 6 added the layout flag as a last template argument, e.g. k_tp_gw2<11, true>)
        names = ("k_tp_prep", "k_tp_fwd<false, 4", "k_tp_fwd<false, 2", "k_tp_fwd<true, 2", "k_tp_d1<11,", "k_tp_d1<9,", "k_tp_gw2<11", "k_tp_gw2<9")
        tiled = any("k_tp_gw2<11, true>" in k for (k, c), vals in rows)
        fetch_kb, write_kb = sum(gmean(n, "FETCH_SIZE") for n in names), sum(gmean(n, "WRITE_SIZE") for n in names)
        doc["group"] = {"round": tag.split("_")[0], "learners": group[0], "envs_per_gpu": group[1], "form": "throughput", "w2_layout": "tiled" if tiled else "flux",
                        "launches": "k_tp_prep + k_tp_fwd x3 + k_tp_d1 x2 + k_tp_gw2 x2", "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
                        "bytes_as_read": (fetch_kb + write_kb) * 1024.0, "bytes_fetch_x2": (2.0 * fetch_kb + write_kb) * 1024.0,
                        "source": f"profiles/{tag}_pmc_counters_train.csv (two rocprofv3 --pmc passes of bench.py --mode group, tools/profile_group.sh)"}
        json.dump(doc, open(path, "w"), indent=1)
        print(json.dumps(doc["group"]))
        return
    act = {c: v for (k, c), v in rows if "k_act2" in k or "k_act<4" in k}
    # only the train-loop launches at 65 536 envs (populate / smoke launches of other sizes are other template instances)
    fetch, write = act["FETCH_SIZE"], act["WRITE_SIZE"]
    fk, wk = sum(fetch) / len(fetch), sum(write) / len(write)
    rec = {"envs_per_gpu": 65536, "round": tag, "kernel": "shems::k_act2 (64-env tiles, two workgroups per CU)", "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
           "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0,
           "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE/WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports 1/2 of a wide (16 B/lane) "
                         "coalesced read stream -> doubled; WRITE_SIZE exact. The 4-byte-per-lane obs reads of this kernel are an uncalibrated "
                         "width, so 2x is an upper bound for the read side.",
           "commands": ["rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline",
                        "rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline"]}
    # the learner's update (train mode, one replica): k_fwd x2 + k_mid + k_grad x2 per replay()
    def mean(name, ctr):
        v = [x for (k, c), vals in rows if c == ctr and name in k for x in vals]
        return sum(v) / len(v) if v else None
    upd = None
    f = {n: mean(n, "FETCH_SIZE") for n in ("k_fwd(", "k_mid(", "k_grad(")}
    w = {n: mean(n, "WRITE_SIZE") for n in ("k_fwd(", "k_mid(", "k_grad(")}
    if all(v is not None for v in list(f.values()) + list(w.values())):
        fetch_kb = 2 * f["k_fwd("] + f["k_mid("] + 2 * f["k_grad("]
        write_kb = 2 * w["k_fwd("] + w["k_mid("] + 2 * w["k_grad("]
        upd = {"round": tag, "launches": "k_fwd x2 + k_mid + k_grad x2", "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
               "bytes_as_read": (fetch_kb + write_kb) * 1024.0, "bytes_fetch_x2": (2.0 * fetch_kb + write_kb) * 1024.0}
    doc.update({"train": rec, "policy": rec, "update": upd})
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps({"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "dispatches": len(fetch)}))


if __name__ == "__main__":
    main()
