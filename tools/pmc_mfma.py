"""Summarise a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv` pass
of bench.py into profiles/<round>_pmc_mfma.csv: per kernel the MFMA-busy fraction and the effective clock.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <out> -- python3 <repo>/bench.py --steps 144 --warmup 8 --prewarm-s 0.2 --no-cpu-baseline [--envs N]
    python tools/pmc_mfma.py <out> [<out2> ...] --out profiles/r03_pmc_mfma.csv

Reading the counters (MI355X_MICROARCH.md, rocprofv3 PMC slots / DVFS give-back):
  GRBM_GUI_ACTIVE            summed over the 8 XCDs -> cycles the chip was busy = GUI / 8; effective clock = GUI / 8 / duration
                             (reads high on dispatches shorter than ~0.3 ms: for k_act's 25-140 us dispatches it is an upper bound)
  SQ_VALU_MFMA_BUSY_CYCLES   summed over all SIMDs: cycles a SIMD's matrix pipe was busy.  MFMA-busy fraction of the dispatch =
                             MFMA_BUSY / (4 SIMDs x 256 CUs x GUI / 8)  (the gfx94x MfmaUtil formula; ROCm 7.2 ships no gfx950 one)
"""
import csv
import glob
import os
import sys
from collections import defaultdict

N_SIMD = 4 * 256


def newest(d, pat):
    f = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True), key=os.path.getmtime)
    return f[-1] if f else None


def short(name):
    name = name.replace("void ", "")
    return name[:name.index("(")] if "(" in name else name[:80]


def summarise(d):
    cc, kt = newest(d, "*counter_collection.csv"), newest(d, "*kernel_trace.csv")
    ctr = defaultdict(lambda: defaultdict(float))          # dispatch id -> counter -> value
    kname = {}
    for r in csv.DictReader(open(cc)):
        did = r["Dispatch_Id"]
        ctr[did][r["Counter_Name"]] += float(r["Counter_Value"])
        kname[did] = short(r["Kernel_Name"])
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3     # us
    per = defaultdict(list)
    for did, c in ctr.items():
        per[kname[did]].append((c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_BUSY_CU_CYCLES", 0.0),
                                dur.get(did)))
    rows = []
    for k, v in per.items():
        v = v[len(v) // 4:]                                # skip the pre-warm quarter: the first dispatches run on a cold chip
        n = len(v)
        mf, gui, bcu = (sum(x[i] for x in v) / n for i in range(3))
        ds = [x[3] for x in v if x[3] is not None]
        du = sum(ds) / len(ds) if ds else float("nan")
        rows.append(dict(kernel=k, dispatches=n, avg_us=round(du, 2), mfma_busy_cycles=round(mf), grbm_gui_active=round(gui),
                         sq_busy_cu_cycles=round(bcu),
                         mfma_busy_frac=round(mf / (N_SIMD * gui / 8), 4) if gui else None,
                         effective_clock_ghz=round(gui / 8 / (du * 1e3), 3) if ds and du else None))
    rows.sort(key=lambda r: -r["avg_us"] * r["dispatches"])
    return rows


def main():
    out, dirs, label = None, [], {}
    a = sys.argv[1:]
    while a:
        x = a.pop(0)
        if x == "--out":
            out = a.pop(0)
        elif "=" in x:                                     # label=dir
            l, d = x.split("=", 1)
            dirs.append(d)
            label[d] = l
        else:
            dirs.append(x)
    rows = []
    for d in dirs:
        for r in summarise(d):
            rows.append(dict(run=label.get(d, os.path.basename(d.rstrip("/"))), **r))
    w = csv.DictWriter(open(out, "w", newline="") if out else sys.stdout, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)


if __name__ == "__main__":
    main()
