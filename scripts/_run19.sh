mkdir -p gpurun_out
for i in 1 2; do
HOUV_HIP_LIB=$PWD/houv_amd/lib/libhouv_hip_slp.so python scripts/ab_chamfer.py 2>&1 | grep -v amdgpu.ids | sed 's/^/SLP   /'
python scripts/ab_chamfer.py 2>&1 | grep -v amdgpu.ids | sed 's/^/NOSLP /'
done | tee gpurun_out/r2_ab_chamfer_noslp.log
HOUV_HIP_LIB=$PWD/houv_amd/lib/libhouv_hip_slp.so python scripts/perf_dcp.py 2>&1 | grep -E "fused|forward|linear|conv4" | sed 's/^/SLP   /'
python scripts/perf_dcp.py 2>&1 | grep -E "fused|forward|linear|conv4" | sed 's/^/NOSLP /'
python -m pytest tests -x -q -m gpu > gpurun_out/r2_gpu_suite2.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r2_gpu_suite2.log
