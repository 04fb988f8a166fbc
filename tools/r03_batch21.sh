R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2 3; do timeout -k 10 600 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2; done
