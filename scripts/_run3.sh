mkdir -p gpurun_out
python -m pytest tests/test_gpu_distributed.py tests/test_gpu_drivers.py tests/test_gpu_envelope.py -x -q -m gpu > gpurun_out/r2_newtests.log 2>&1; echo "new tests rc=$?"; tail -15 gpurun_out/r2_newtests.log
python scripts/load_balance.py > gpurun_out/r2_load_balance.log 2>&1; echo "lb rc=$?"; tail -3 gpurun_out/r2_load_balance.log | cut -c1-600
