set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_policy_gpu.py -m gpu -x -q -k "every_form" 2>&1 | tail -12
cd /tmp
for rep in 1 2; do
for n in 65536 32768 16384; do
  for f in 1 2; do
    SHEMS_ACT_FORM4=$f timeout -k 10 300 python3 $R/bench.py --envs $n --steps 288 --no-cpu-baseline > /tmp/ab.json 2>/dev/null
    python3 -c "import json;d=json.load(open('/tmp/ab.json'));r=d['roofline'];print('form4=$f',$n, round(d['value']/1e6,1),'M/s  k_act',round(r['kernel_avg_us'],2),'us frac',round(r['frac'],3),'upd',round(d['update_us'],2))"
  done
done
done
