set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_final
mkdir -p $O
cd $R
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/gputest_final.log 2>&1 || { tail -60 $O/gputest_final.log; exit 1; }
tail -2 $O/gputest_final.log
