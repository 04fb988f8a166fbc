set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03i
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_ddpg_gpu.py tests/test_determinism_gpu.py tests/test_train_gpu.py tests/test_group_gpu.py -m gpu -x -q > $O/gputest.log 2>&1 || { tail -60 $O/gputest.log; exit 1; }
tail -2 $O/gputest.log
bash tools/r03_ab.sh "65536 8192" libshems_hip.so libshems_hip_w32.so
