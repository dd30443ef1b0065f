bash scripts/prof_r2.sh > gpurun_out/prof_r2_stdout.log 2>&1; tail -12 gpurun_out/prof_r2_stdout.log
rm -f gpurun_out/r2_ab_chamfer.log
for i in 1 2; do
HOUV_CHAMFER_DIRECT=1 python scripts/ab_chamfer.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2_ab_chamfer.log
python scripts/ab_chamfer.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2_ab_chamfer.log
done
python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err; echo "bench rc=$?"
