mkdir -p gpurun_out
python -m pytest tests/test_gpu_dcp.py -x -q -m gpu > gpurun_out/r2_dcp_tests.log 2>&1; echo "dcp tests rc=$?"; tail -5 gpurun_out/r2_dcp_tests.log
python scripts/perf_dcp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_perf_dcp.log | tail -12
