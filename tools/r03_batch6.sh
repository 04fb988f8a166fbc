set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03f
mkdir -p $O
cd $R
P=master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1 || { tail -60 $O/gputest.log; exit 1; }
tail -2 $O/gputest.log
export SHEMS_HIP_LIB=$R/$P/libshems_hip_abl.so
timeout -k 10 120 python3 tools/stamp_k_act_line.py 65536 2>&1 | tail -1
timeout -k 10 120 python3 tools/stamp_k_act_line.py 16384 2>&1 | tail -1
SHEMS_ACT_FORM=3 timeout -k 10 120 python3 tools/stamp_k_act_line.py 8192 2>&1 | tail -1
unset SHEMS_HIP_LIB
cd /tmp
for n in 65536 32768 16384 8192 4096; do
  timeout -k 10 300 python3 $R/bench.py --envs $n --steps 288 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
  python3 -c "import json;d=json.load(open('$O/bench_$n.json'));r=d['roofline'];print($n, round(d['value']/1e6,1),'M/s  k_act',round(r['kernel_avg_us'],2),'us frac',round(r['frac'],3),'upd',round(d['update_us'],2))"
done
for n in 8192 12288; do
  SHEMS_ACT_FORM=3 timeout -k 10 300 python3 $R/bench.py --envs $n --steps 288 --no-cpu-baseline > $O/bench_${n}_form3.json 2>/dev/null
  python3 -c "import json;d=json.load(open('$O/bench_${n}_form3.json'));r=d['roofline'];print('form3',$n, round(d['value']/1e6,1),'M/s  k_act',round(r['kernel_avg_us'],2),'us frac',round(r['frac'],3))"
done
