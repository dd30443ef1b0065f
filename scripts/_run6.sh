mkdir -p gpurun_out
python -m pytest tests/test_gpu_solve.py -x -q -m gpu > gpurun_out/r2_solve_tests2.log 2>&1; echo "solve tests rc=$?"; tail -5 gpurun_out/r2_solve_tests2.log
rm -f gpurun_out/r2_ab_predict.log
for i in 1 2; do
  HOUV_SOLVE_PREDICT=all python scripts/ab_solve.py 2>&1 | grep -v amdgpu.ids | sed 's/^/ALL  /' | tee -a gpurun_out/r2_ab_predict.log
  python scripts/ab_solve.py 2>&1 | grep -v amdgpu.ids | sed 's/^/PRED /' | tee -a gpurun_out/r2_ab_predict.log
  HOUV_SOLVE_PREDICT=b python scripts/ab_solve.py 2>&1 | grep -v amdgpu.ids | sed 's/^/B    /' | tee -a gpurun_out/r2_ab_predict.log
done
