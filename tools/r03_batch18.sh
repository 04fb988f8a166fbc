set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_bench_gpu.py -m gpu -x -q 2>&1 | tail -3
cd /tmp
for n in 65536 8192 4096; do
  timeout -k 10 300 python3 $R/bench.py --envs $n --steps 288 --no-cpu-baseline > /tmp/ab.json 2>/dev/null
  python3 -c "import json;d=json.load(open('/tmp/ab.json'));r=d['roofline'];print($n, round(d['value']/1e6,1),'M/s  k_act avg',round(r['kernel_avg_us'],2),'med',round(r['kernel_median_us'],2),'frac',round(r['frac'],3),'upd',round(d['update_us'],2),'implied',round(d['ms_per_step']*1e3-d['update_us'],2))"
done
