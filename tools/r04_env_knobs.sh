# Round 4: HIP runtime knobs that touch launch latency, A/B on one box (bench.py train mode, alternating).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_knobs; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
one() { python3 - <<PY
import json
d=json.loads(open("$O/x.json").read().strip().splitlines()[-1])
print("$1", round(d["value"]/1e6,1),"M", round(d["ms_per_step"]*1e3,2),"us/step upd",round(d["update_us"],2),"k", round(d["roofline"]["kernel_avg_us"],2))
PY
}
for n in 4096 65536; do
 for rep in 1 2; do
  timeout -k 10 200 python3 $R/bench.py --envs $n --steps 1440 --no-cpu-baseline --prewarm-s 1 > $O/x.json 2>$O/err.txt; one "$n default          "
  HIP_FORCE_DEV_KERNARG=1 timeout -k 10 200 python3 $R/bench.py --envs $n --steps 1440 --no-cpu-baseline --prewarm-s 1 > $O/x.json 2>$O/err.txt; one "$n DEV_KERNARG=1     "
  HIP_FORCE_DEV_KERNARG=0 timeout -k 10 200 python3 $R/bench.py --envs $n --steps 1440 --no-cpu-baseline --prewarm-s 1 > $O/x.json 2>$O/err.txt; one "$n DEV_KERNARG=0     "
  AMD_OPT_FLUSH=0 timeout -k 10 200 python3 $R/bench.py --envs $n --steps 1440 --no-cpu-baseline --prewarm-s 1 > $O/x.json 2>$O/err.txt; one "$n AMD_OPT_FLUSH=0    "
 done
done
GPU_STREAMOPS_CP_WAIT=1 timeout -k 10 200 python3 $R/bench.py --envs 4096 --steps 1440 --no-cpu-baseline --prewarm-s 1 --overlap pipelined > $O/x.json 2>$O/err.txt; one "4096 pipelined STREAMOPS_CP_WAIT=1"
timeout -k 10 200 python3 $R/bench.py --envs 4096 --steps 1440 --no-cpu-baseline --prewarm-s 1 --overlap pipelined > $O/x.json 2>$O/err.txt; one "4096 pipelined default"
