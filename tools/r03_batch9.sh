set -e
R=$GRAFT_REPO_ROOT
cd $R
P=master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd
export SHEMS_HIP_LIB=$R/$P/libshems_hip_stamp.so
SHEMS_DDPG_MERGE=1 timeout -k 10 200 python3 tools/stamp_update.py 2>&1 | grep -v amdgpu.ids | tail -22
SHEMS_DDPG_MERGE=0 timeout -k 10 200 python3 tools/stamp_update.py 2>&1 | grep -v amdgpu.ids | tail -12
