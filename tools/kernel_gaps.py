"""Idle time between consecutive kernels of one rocprofv3 --kernel-trace (csv): per (previous kernel -> next kernel) pair the mean gap
between the end of one dispatch and the start of the next.  python tools/kernel_gaps.py <dir with *kernel_trace.csv> [--match k_tp]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void ", "")
    return n[:n.index("(")] if "(" in n else n[:60]


d = sys.argv[1]
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f))), key=lambda x: x[0])
rows = rows[len(rows) // 4:]
gaps = defaultdict(list)
for (s0, e0, k0), (s1, e1, k1) in zip(rows, rows[1:]):
    if match in k0 or match in k1:
        gaps[(k0, k1)].append((s1 - e0) * 1e-3)
print(f"{'previous':34s} {'next':34s} {'n':>5s} {'mean gap us':>12s} {'min':>8s} {'max':>8s}")
tot = 0.0
for (k0, k1), v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) >= 5:
        print(f"{k0[-34:]:34s} {k1[-34:]:34s} {len(v):5d} {sum(v) / len(v):12.2f} {min(v):8.2f} {max(v):8.2f}")
