# Per-kernel times of the grouped update under different builds of the library, one gpurun call:
#   gpurun -- 'bash tools/ab_kernel_times.sh abl/lib_A.so abl/lib_B.so'
# rocprofv3 kernel-trace of `bench.py --mode group` (400 learners x 128) per library; prints the average of every k_tp_* kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  O=$R/gpurun_out/abk_$(basename $lib .so)
  rm -rf $O; mkdir -p $O
  SHEMS_HIP_LIB=$R/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --mode group --learners ${LEARNERS:-400} --envs ${ENVS:-51200} --mixed --no-cpu-baseline --steps 72 --warmup 8 --prewarm-s 0.5 > $O/kt.log 2>&1
  f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
  echo "== $lib"
  python3 - "$f" <<PY
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "k_tp_" in r["Name"] or "k_act" in r["Name"]]
tot=0
for r in sorted(rows,key=lambda r:r["Name"]):
    us=float(r["AverageNs"])/1e3; print("  %-64s %5s %8.1f"%(r["Name"][:64], r["Calls"], us)); tot+=us if "k_tp_" in r["Name"] else 0
print("  sum of the update's kernels %.1f us"%tot)
PY
  rm -rf $O/kt
done
