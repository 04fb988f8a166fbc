set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_b2; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt_ovl4096 -- python3 $R/bench.py --envs 4096 --overlap --steps 200 --warmup 50 --prewarm-s 0.1 --no-cpu-baseline > $O/kt.log 2>&1
cp $(find $O/kt_ovl4096 -name "*kernel_trace.csv" | head -1) $O/ovl4096_trace.csv; rm -rf $O/kt_ovl4096
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/ovl4096_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-60:]
t0=int(rows[0]["Start_Timestamp"])
for r in rows:
    print("%-40s q=%s  %8.2f -> %8.2f  (%.2f us)"%(r["Kernel_Name"][:40], r.get("Queue_Id"), (int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
CU_MASK_R=64,96,128,160 python3 $R/tools/cu_mask_probe.py 4096 > $O/mask4096.txt 2>&1; grep "us/step" $O/mask4096.txt
CU_MASK_R=64,96,128,160 python3 $R/tools/cu_mask_probe.py 8192 > $O/mask8192.txt 2>&1; grep "us/step" $O/mask8192.txt
