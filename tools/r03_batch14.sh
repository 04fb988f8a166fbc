set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_padding.py tests/test_main_entry.py tests/test_harness.py -m gpu -x -q 2>&1 | tail -25
