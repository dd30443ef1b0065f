mkdir -p gpurun_out
L=houv_amd/lib
for i in 1 2 3; do
  HOUV_HIP_LIB=$PWD/$L/libhouv_hip_r1.so python scripts/ab_solve.py >> gpurun_out/r2_ab_epilogue.log 2>&1 || exit 1
  python scripts/ab_solve.py >> gpurun_out/r2_ab_epilogue.log 2>&1 || exit 1
done
cat gpurun_out/r2_ab_epilogue.log
