set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_policy_gpu.py tests/test_group_gpu.py tests/test_determinism_gpu.py tests/test_harness.py tests/test_env_gpu.py -m gpu -x -q > $O/gputest.log 2>&1 || { tail -60 $O/gputest.log; exit 1; }
tail -3 $O/gputest.log
cd /tmp && export TMPDIR=/tmp
for n in 4096 8192 65536; do
  timeout -k 10 300 python3 $R/bench.py --envs $n --steps 288 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
  python3 -c "import json;d=json.load(open('$O/bench_$n.json'));r=d['roofline'];print($n, round(d['value']/1e6,1),'M/s  k_act',round(r['kernel_avg_us'],2),'us frac',round(r['frac'],3),'upd',round(d['update_us'],2))"
done
for f in 3; do
  for n in 4096 8192; do
  SHEMS_ACT_FORM=$f timeout -k 10 300 python3 $R/bench.py --envs $n --steps 288 --no-cpu-baseline > $O/bench_${n}_form$f.json 2>/dev/null
  python3 -c "import json;d=json.load(open('$O/bench_${n}_form$f.json'));r=d['roofline'];print('form$f',$n, round(d['value']/1e6,1),'M/s  k_act',round(r['kernel_avg_us'],2),'us frac',round(r['frac'],3))"
  done
done
SHEMS_ACT_FORM=8 timeout -k 10 300 python3 $R/bench.py --envs 4096 --steps 288 --no-cpu-baseline > $O/bench_4096_form8.json 2>/dev/null
python3 -c "import json;d=json.load(open('$O/bench_4096_form8.json'));r=d['roofline'];print('form8 4096', round(d['value']/1e6,1),'M/s  k_act',round(r['kernel_avg_us'],2),'us frac',round(r['frac'],3))"
SHEMS_ACT_FORM=9 timeout -k 10 300 python3 $R/bench.py --envs 8192 --steps 288 --no-cpu-baseline > $O/bench_8192_form9.json 2>/dev/null
python3 -c "import json;d=json.load(open('$O/bench_8192_form9.json'));r=d['roofline'];print('form9 8192', round(d['value']/1e6,1),'M/s  k_act',round(r['kernel_avg_us'],2),'us frac',round(r['frac'],3))"
