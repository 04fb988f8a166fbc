"""Per kernel: where the waves' cycles go (one rocprofv3 --pmc pass of the eight SQ counters below, --kernel-trace only).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
              SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d <out> -- python3 <repo>/bench.py ...
    python tools/pmc_stalls.py <out> --out profiles/rNN_pmc_stalls.csv [--match k_tp]

Buckets (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAIT_ANY = wave parked (s_waitcnt / barrier), SQ_WAIT_INST_ANY = issue stall
(MFMA RAW / pipe busy; SQ_WAIT_INST_LDS is a sub-bucket), SQ_ACTIVE_INST_ANY = issuing; the three add up to about SQ_WAVE_CYCLES.
Fractions below are of SQ_WAVE_CYCLES."""
import csv
import glob
import os
import sys
from collections import defaultdict

NAMES = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT",
         "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"]


def short(name):
    name = name.replace("void ", "")
    return name[:name.index("(")] if "(" in name else name[:80]


def main():
    out, match, dirs = None, "", []
    it = iter(sys.argv[1:])
    for a in it:
        if a == "--out":
            out = next(it)
        elif a == "--match":
            match = next(it)
        else:
            dirs.append(a)
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))          # kernel -> dispatch -> counter -> value
    for d in dirs:
        f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if match in k:
                per[k][r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    rows = []
    for k, disp in per.items():
        v = list(disp.values())
        v = v[len(v) // 4:]
        mean = {n: sum(x.get(n, 0.0) for x in v) / len(v) for n in NAMES}
        wc = mean["SQ_WAVE_CYCLES"] or float("nan")
        rows.append(dict(kernel=k, dispatches=len(v), wave_cycles=round(wc), parked=round(mean["SQ_WAIT_ANY"] / wc, 4),
                         issue_stall=round(mean["SQ_WAIT_INST_ANY"] / wc, 4), issuing=round(mean["SQ_ACTIVE_INST_ANY"] / wc, 4),
                         lds_issue_stall=round(mean["SQ_WAIT_INST_LDS"] / wc, 4), lds_bank_conflict_cycles=round(mean["SQ_LDS_BANK_CONFLICT"]),
                         mfma_busy_cycles=round(mean["SQ_VALU_MFMA_BUSY_CYCLES"]), busy_cu_cycles=round(mean["SQ_BUSY_CU_CYCLES"])))
    rows.sort(key=lambda r: -r["wave_cycles"])
    w = csv.DictWriter(open(out, "w", newline="") if out else sys.stdout, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)


if __name__ == "__main__":
    main()
