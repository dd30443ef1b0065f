mkdir -p gpurun_out
python -m pytest tests/test_gpu_dcp.py tests/test_gpu_drivers.py tests/test_gpu_pointops.py -x -q -m gpu > gpurun_out/r2_tests3.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r2_tests3.log
python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r2_bench.json
