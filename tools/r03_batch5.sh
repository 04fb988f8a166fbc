set -e
R=$GRAFT_REPO_ROOT
cd $R
P=master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd
export SHEMS_HIP_LIB=$R/$P/libshems_hip_abl.so
SHEMS_ACT_FORM=8 timeout -k 10 120 python3 tools/stamp_k_act_line.py 8192 2>&1 | tail -2
SHEMS_ACT_FORM=9 timeout -k 10 120 python3 tools/stamp_k_act_line.py 4096 2>&1 | tail -2
