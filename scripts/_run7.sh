HOUV_STAMPS_LIB=$PWD/houv_amd/lib/libhouv_hip_stamps.so P=64 ITERS=50 python scripts/stamps.py 2>&1 | grep -v amdgpu
