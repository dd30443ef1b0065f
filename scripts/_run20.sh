mkdir -p gpurun_out; rm -f gpurun_out/r2_ab_ldsboxes.log
for i in 1 2 3; do
  HOUV_HIP_LIB=$PWD/houv_amd/lib/libhouv_hip_prev.so python scripts/ab_solve.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2_ab_ldsboxes.log
  python scripts/ab_solve.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2_ab_ldsboxes.log
done
python -m pytest tests/test_gpu_solve.py -x -q -m gpu -k "pruned or prediction" 2>&1 | tail -3
