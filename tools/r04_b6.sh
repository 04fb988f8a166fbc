set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_b6; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_train_loop_gpu.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
run() { n=$1; tag=$2; shift 2
  timeout -k 10 200 python3 $R/bench.py --envs $n --steps 2880 --no-cpu-baseline "$@" > $O/${tag}_$n.json 2>$O/${tag}_$n.err || { tail -5 $O/${tag}_$n.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("$O/${tag}_$n.json").read().strip().splitlines()[-1])
print($n,"$tag",round(d["value"]/1e6,1),"M env-steps/s", round(d["ms_per_step"]*1e3,2),"us/step upd",round(d.get("update_us") or 0,2), "k", round(d["roofline"]["kernel_avg_us"],2))
PY
}
for n in 4096 8192 16384; do
  run $n ordered
  run $n pipelined_device --overlap pipelined
  SHEMS_LOOP_SYNC=values run $n pipelined_values --overlap pipelined
done
