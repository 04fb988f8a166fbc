set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_b4; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for m in pipelined; do
rocprofv3 --kernel-trace --output-format csv -d $O/kt_$m -- python3 $R/bench.py --envs 4096 --overlap $m --steps 300 --warmup 50 --prewarm-s 0.1 --no-cpu-baseline > $O/kt_$m.log 2>&1
cp $(find $O/kt_$m -name "*kernel_trace.csv" | head -1) $O/${m}_trace.csv; rm -rf $O/kt_$m
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/${m}_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if 'k_actg' in r["Kernel_Name"]]
i0=idx[len(idx)//3]
t0=int(rows[i0]["Start_Timestamp"])
print("== $m")
for r in rows[i0:i0+36]:
    print("%-28s q=%s %8.2f -> %8.2f (%.2f) grid %s"%(r["Kernel_Name"][:28], r.get("Queue_Id"), (int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Grid_Size_X"]))
PY
done
