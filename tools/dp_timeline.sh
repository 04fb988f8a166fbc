set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_dp_tl; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for how in torch native; do
  rocprofv3 --kernel-trace --output-format csv -d $O/kt_$how -- python3 $R/tools/dp_timeline.py $how > $O/$how.log 2>&1
  cp $(find $O/kt_$how -name "*kernel_trace.csv" | head -1) $O/${how}_trace.csv; rm -rf $O/kt_$how
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/${how}_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if 'k_actg' in r["Kernel_Name"]]
i0=idx[len(idx)*2//3]
t0=int(rows[i0]["Start_Timestamp"])
print("== $how")
for r in rows[i0:i0+20]:
    print("%-44s q=%s %8.2f -> %8.2f (%6.2f us)"%(r["Kernel_Name"][:44], r.get("Queue_Id"), (int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
done
