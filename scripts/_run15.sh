mkdir -p gpurun_out; rm -f gpurun_out/r2_ab_refresh.log
for i in 1 2; do
 for r in 1 2 4 8 1000; do
  HOUV_PRUNE_REFRESH=$r python scripts/ab_solve.py 2>&1 | grep -v amdgpu.ids | sed "s/^/refresh=$r /" | tee -a gpurun_out/r2_ab_refresh.log
 done
done
