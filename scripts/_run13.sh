mkdir -p gpurun_out
python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu 2>&1 | tail -5
