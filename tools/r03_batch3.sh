set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
P=master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd
export SHEMS_HIP_LIB=$R/$P/libshems_hip_abl.so
for n in 4096 8192; do
  for f in -1 3 8 9; do
    SHEMS_ACT_FORM=$f timeout -k 10 120 python3 tools/stamp_k_act_line.py $n 2>&1 | tail -1
  done
done
unset SHEMS_HIP_LIB
cd /tmp && export TMPDIR=/tmp
for n in 4096 8192; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$n -- python3 $R/bench.py --envs $n --steps 144 --warmup 8 --prewarm-s 0.2 --no-cpu-baseline > $O/kt$n.log 2>&1
f=$(find $O/kt$n -name "*kernel_stats.csv" | head -1)
head -8 $f | cut -c1-150
done
