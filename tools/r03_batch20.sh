R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_final
mkdir -p $O
cd /tmp
python3 $R/bench.py --mode group --learners 32 --no-cpu-baseline > $O/r03_group_bench.json 2> $O/r03_group_bench.err
tail -5 $O/r03_group_bench.err
python3 -c "import json;d=json.load(open('$O/r03_group_bench.json'));r=d['roofline'];print('group', round(d['value']/1e6,1), round(d['updates_per_sec']), round(r['kernel_avg_us'],2), round(r['frac'],3))"
