R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_bench_gpu.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -40
