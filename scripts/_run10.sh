mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --backend gloo --single-device --pairs 32 > gpurun_out/r2_bench_2rank_gloo.json 2> gpurun_out/r2_bench_2rank_gloo.err; echo "2-rank rc=$?"; tail -c 600 gpurun_out/r2_bench_2rank_gloo.json; tail -3 gpurun_out/r2_bench_2rank_gloo.err
python bench.py --dcp --pairs 64 --steps 3 --warmup 1 > gpurun_out/r2_bench_dcp.json 2>/dev/null; echo "dcp rc=$?"; tail -c 700 gpurun_out/r2_bench_dcp.json
python bench.py --icp --pairs 64 --steps 1 --warmup 1 --no-cpu-baseline --no-chamfer-op > gpurun_out/r2_bench_icp.json 2>/dev/null; echo "icp rc=$?"; head -c 400 gpurun_out/r2_bench_icp.json
