mkdir -p gpurun_out
python -m pytest tests/test_gpu_chamfer.py -x -q -m gpu > gpurun_out/r2_chamfer_tests.log 2>&1; echo "chamfer tests rc=$?"; tail -3 gpurun_out/r2_chamfer_tests.log
rm -f gpurun_out/r2_ab_chamfer.log
for i in 1 2; do
HOUV_CHAMFER_DIRECT=1 python scripts/ab_chamfer.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2_ab_chamfer.log
python scripts/ab_chamfer.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2_ab_chamfer.log
done
