set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 4096 8192; do
  python3 $R/bench.py --envs $n --steps 288 --no-cpu-baseline > $O/r03_train_${n}_bench.json 2>/dev/null
  python3 -c "import json;d=json.load(open('$O/r03_train_${n}_bench.json'));r=d['roofline'];print($n, round(d['value']/1e6,1),'M/s  k_act',round(r['kernel_avg_us'],2),'us frac',round(r['frac'],3),'upd',round(d['update_us'],2), 'implied', round(d['ms_per_step']*1e3-d['update_us'],2))"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$n -- python3 $R/bench.py --envs $n --steps 288 --warmup 72 --no-cpu-baseline > $O/kt$n.log 2>&1
  cp $(find $O/kt$n -name "*kernel_stats.csv" | head -1) $O/r03_train_${n}_kernel_stats.csv
  head -3 $O/r03_train_${n}_kernel_stats.csv | cut -c1-140
  rm -rf $O/kt$n
done
python3 $R/bench.py --envs 65536 --steps 288 --no-cpu-baseline > /tmp/b.json 2>/dev/null
python3 -c "import json;d=json.load(open('/tmp/b.json'));r=d['roofline'];print(65536, round(d['value']/1e6,1),'M/s  k_act',round(r['kernel_avg_us'],2),'us frac',round(r['frac'],3),'upd',round(d['update_us'],2), 'implied', round(d['ms_per_step']*1e3-d['update_us'],2))"
