set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
nproc; python3 -c "import os;print(len(os.sched_getaffinity(0)), os.cpu_count())"; cat /sys/fs/cgroup/cpu.max 2>/dev/null || true
( time python3 $R/bench.py > $O/r03_train_bench.json 2> $O/r03_train_bench.err ) 2>&1 | grep real
python3 -c "import json;d=json.load(open('$O/r03_train_bench.json'));c=d['cpu_baseline'];print(round(d['value']/1e6,1), d['roofline']['frac'], json.dumps({k:c[k] for k in ('value','cores','one_thread_value','all_cores_routes','updates_per_sec','env_only_all_cores_value','env_only_value')}))"
