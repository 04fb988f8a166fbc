set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03h
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_ddpg_gpu.py tests/test_determinism_gpu.py tests/test_train_gpu.py tests/test_group_gpu.py -m gpu -x -q > $O/gputest.log 2>&1 || { tail -60 $O/gputest.log; exit 1; }
tail -2 $O/gputest.log
cd /tmp
for m in 1 0 1 0; do
  SHEMS_DDPG_MERGE=$m timeout -k 10 300 python3 $R/bench.py --steps 288 --no-cpu-baseline > $O/bench_merge$m.json 2> $O/bench_merge$m.err
  python3 -c "import json;d=json.load(open('$O/bench_merge$m.json'));r=d['roofline'];print('merge=$m', round(d['value']/1e6,1),'M/s  k_act',round(r['kernel_avg_us'],2),'us  upd',round(d['update_us'],2),'us  updates/s',round(d['updates_per_sec']))"
done
