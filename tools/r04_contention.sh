# Round 4: what do the step kernel and the update contend for when they share CUs?  Pipelined loop with in-kernel waits (SHEMS_LOOP_SYNC=device)
# at 4 096 envs, rocprofv3 kernel trace per ablation of the step kernel's layer-2 loop (diagnostic libraries: wrong results by construction).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_contention; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
P=$R/master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd
for v in "" _nodma _nolds _nomfma; do
  SHEMS_HIP_LIB=$P/libshems_hip$v.so SHEMS_LOOP_SYNC=device rocprofv3 --kernel-trace --output-format csv -d $O/kt$v -- python3 $R/bench.py --envs 4096 --overlap pipelined --steps 300 --warmup 50 --prewarm-s 0.1 --no-cpu-baseline > $O/log$v.txt 2>&1 || { tail -5 $O/log$v.txt; continue; }
  cp $(find $O/kt$v -name "*kernel_trace.csv" | head -1) $O/trace$v.csv; rm -rf $O/kt$v
  python3 - <<PY
import csv, statistics as st
rows=list(csv.DictReader(open("$O/trace$v.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
act=[r for r in rows if 'k_actg' in r["Kernel_Name"]]
lo=int(act[len(act)//3]["Start_Timestamp"]); hi=int(act[len(act)*2//3]["Start_Timestamp"])
sel=[r for r in rows if lo<=int(r["Start_Timestamp"])<hi]
def dur(name, grid=None):
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in sel if name in r["Kernel_Name"] and (grid is None or r["Grid_Size_X"]==grid)]
    return round(st.median(d),2) if d else None
a=[r for r in sel if 'k_actg' in r["Kernel_Name"]]
period=(int(a[-1]["Start_Timestamp"])-int(a[0]["Start_Timestamp"]))/1e3/(len(a)-1)
print("variant '%s': period %.1f us  act span %s  K1 %s  K2 %s  K3/K5 %s  K4 %s"%("$v", period, dur("k_actg"), dur("k_fwd","53760"), dur("k_mid"), dur("k_grad"), dur("k_fwd","32768")))
PY
done

# ---- second part: the same loop with K1's in-kernel wait switched off (timing diagnostics only) ----
for nw in 0 1; do
  SHEMS_LOOP_DIAG_NOWAIT_K1=$nw SHEMS_LOOP_SYNC=device timeout -k 10 200 python3 $R/bench.py --envs 4096 --overlap pipelined --steps 2880 --prewarm-s 1 --no-cpu-baseline > $O/b$nw.json 2>$O/err.txt || tail -3 $O/err.txt
  python3 -c "
import json; d=json.loads(open('$O/b$nw.json').read().strip().splitlines()[-1]); print('nowait_k1=$nw bench', round(d['ms_per_step']*1e3,2),'us/step')"
  SHEMS_LOOP_DIAG_NOWAIT_K1=$nw SHEMS_LOOP_SYNC=device rocprofv3 --kernel-trace --output-format csv -d $O/ktw$nw -- python3 $R/bench.py --envs 4096 --overlap pipelined --steps 300 --warmup 50 --prewarm-s 0.1 --no-cpu-baseline > $O/logw$nw.txt 2>&1 || { tail -5 $O/logw$nw.txt; continue; }
  cp $(find $O/ktw$nw -name "*kernel_trace.csv" | head -1) $O/tracew$nw.csv; rm -rf $O/ktw$nw
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/tracew$nw.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if 'k_actg' in r["Kernel_Name"]]
i0=idx[len(idx)//2]; t0=int(rows[i0]["Start_Timestamp"])
print("== nowait_k1=$nw")
for r in rows[i0:i0+13]:
    print("%-30s q=%s %8.2f -> %8.2f (%6.2f us) grid %s"%(r["Kernel_Name"][:30], r.get("Queue_Id"), (int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Grid_Size_X"]))
PY
done
