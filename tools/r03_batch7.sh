set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03g
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_harness.py tests/test_main_entry.py -m gpu -x -q > $O/gputest.log 2>&1 || { tail -60 $O/gputest.log; exit 1; }
tail -2 $O/gputest.log
timeout -k 10 300 python3 tools/track_time.py
