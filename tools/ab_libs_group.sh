# A/B of builds of the library inside ONE gpurun call (box-to-box spread is ~2.5 %, more than most single changes):
#   gpurun -- 'bash tools/ab_libs_group.sh abl/lib_A.so abl/lib_B.so'
# runs `bench.py --mode group` (400 learners x 128 envs) alternately on each library (SHEMS_HIP_LIB), three rounds, and prints ms per step.
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2 3; do
  for lib in "$@"; do
    SHEMS_HIP_LIB=$R/$lib timeout -k 10 200 python3 $R/bench.py --mode group --learners 400 --envs 51200 --mixed --no-cpu-baseline --steps 144 --warmup 16 --prewarm-s 1 2>/dev/null \
      | python3 -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$lib', round(d['ms_per_step'],4), round(d['updates_per_sec']), r.get('avg_us'))" || exit 1
  done
done
