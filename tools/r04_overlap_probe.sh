# Round 4, VERDICT item 1(a): the pipelined mode at the sizes where the update dominates, next to the sequential loop, on one box.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_overlap; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for n in 4096 8192 16384 65536; do
  python3 $R/bench.py --envs $n --steps 720 --no-cpu-baseline > $O/seq_$n.json 2>$O/seq_$n.err
  python3 $R/bench.py --envs $n --steps 720 --no-cpu-baseline --overlap > $O/ovl_$n.json 2>$O/ovl_$n.err
  python3 - <<PY
import json
for k in ("seq","ovl"):
    d=json.loads(open("$O/%s_$n.json"%k).read().strip().splitlines()[-1])
    print($n,k,round(d["value"]/1e6,1),"M env-steps/s", d["ms_per_step"]*1e3,"us/step", d.get("update_us"), d.get("roofline",{}).get("kernel_avg_us"))
PY
done
