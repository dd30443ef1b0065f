mkdir -p gpurun_out; rm -f gpurun_out/r2_ab_noslp.log
for i in 1 2 3; do
  python scripts/ab_solve.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2_ab_noslp.log
  HOUV_HIP_LIB=$PWD/houv_amd/lib/libhouv_hip_noslp.so python scripts/ab_solve.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2_ab_noslp.log
done
