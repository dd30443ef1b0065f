HOUV_STAMPS_LIB=$PWD/houv_amd/lib/libhouv_hip_stamps.so PRUNED=1 P=64 ITERS=50 python scripts/stamps.py 2>&1 | grep -v amdgpu
HOUV_HIP_LIB=$PWD/houv_amd/lib/libhouv_hip_stamps.so python scripts/prune_stats.py 2>&1 | grep -v amdgpu
