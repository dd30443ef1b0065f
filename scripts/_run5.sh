mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r2_gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -4 gpurun_out/r2_gpu_suite.log
HOUV_STAMPS_LIB=$PWD/houv_amd/lib/libhouv_hip_stamps.so P=64 ITERS=10 python scripts/stamps.py > gpurun_out/r2_stamps.log 2>&1; echo "stamps rc=$?"; cat gpurun_out/r2_stamps.log | grep -v amdgpu
./scripts/ubench/misc_rate > gpurun_out/r2_instr_rates.txt 2>&1
./scripts/ubench/fma_operands >> gpurun_out/r2_instr_rates.txt 2>&1
