set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03j
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SHEMS_ACT_FORM4=2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 144 --warmup 72 --no-cpu-baseline > $O/kt.log 2>&1
head -4 $(find $O/kt -name "*kernel_stats.csv" | head -1) | cut -c1-120
rm -rf $O/kt
for n in 10240 12288 14336; do
  timeout -k 10 300 python3 $R/bench.py --envs $n --steps 288 --no-cpu-baseline > /tmp/ab.json 2>/dev/null
  python3 -c "import json;d=json.load(open('/tmp/ab.json'));r=d['roofline'];print('default',$n, round(d['value']/1e6,1),'M/s  k_act',round(r['kernel_avg_us'],2),'us frac',round(r['frac'],3))"
  SHEMS_ACT_FORM4=2 SHEMS_ACT_FORM=12 timeout -k 10 300 python3 $R/bench.py --envs $n --steps 288 --no-cpu-baseline > /tmp/ab.json 2>/dev/null
  python3 -c "import json;d=json.load(open('/tmp/ab.json'));r=d['roofline'];print('k_act2 ',$n, round(d['value']/1e6,1),'M/s  k_act',round(r['kernel_avg_us'],2),'us frac',round(r['frac'],3))"
done
