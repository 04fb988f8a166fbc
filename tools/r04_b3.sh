set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_b3; mkdir -p $O; cd $R
timeout -k 10 500 python3 -m pytest tests/test_train_loop_gpu.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
for n in 4096 8192 16384 65536; do
  for m in "--loop host" "" "--overlap pipelined" "--overlap exact"; do
    tag=$(echo "$m" | tr -d ' -'); tag=${tag:-native}
    python3 $R/bench.py --envs $n --steps 1440 --no-cpu-baseline $m > $O/${tag}_$n.json 2>$O/${tag}_$n.err || { tail -5 $O/${tag}_$n.err; exit 1; }
    python3 - <<PY
import json
d=json.loads(open("$O/${tag}_$n.json").read().strip().splitlines()[-1])
print($n,"$tag",round(d["value"]/1e6,1),"M env-steps/s", round(d["ms_per_step"]*1e3,2),"us/step upd",round(d.get("update_us") or 0,2), "crc", d.get("learner_crc32"))
PY
  done
done
