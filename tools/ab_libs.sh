# A/B of library variants inside ONE gpurun call (same box): usage  bash tools/ab_libs.sh "<envs list>" libA.so libB.so ...
set -e
R=$GRAFT_REPO_ROOT
cd /tmp
ENVS="$1"; shift
for rep in 1 2; do
for n in $ENVS; do
  for lib in "$@"; do
    SHEMS_HIP_LIB=$R/master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd/$lib timeout -k 10 300 python3 $R/bench.py --envs $n --steps 1440 --prewarm-s 1 --no-cpu-baseline > /tmp/ab.json 2>/dev/null
    python3 -c "import json;d=json.load(open('/tmp/ab.json'));r=d['roofline'];print('$lib',$n, round(d['value']/1e6,1),'M/s  k_act',round(r['kernel_avg_us'],2),'us frac',round(r['frac'],3),'upd',round(d['update_us'],2))"
  done
done
done
