mkdir -p gpurun_out
python -m pytest tests/test_gpu_solve.py tests/test_gpu_bench_contract.py tests/test_gpu_drivers.py tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/r2_tests4.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r2_tests4.log
for c in 1 0; do HOUV_CONC=$c python - <<'PY'
import os, sys, json, io
sys.path.insert(0, os.getcwd())
from houv_amd import solver
solver.CONCURRENT_RETRIES = os.environ["HOUV_CONC"] == "1"
import bench
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-chamfer-op"]
from contextlib import redirect_stdout
buf = io.StringIO()
with redirect_stdout(buf): bench.main()
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][0])
print("CONCURRENT" if solver.CONCURRENT_RETRIES else "SEQUENTIAL", "value %.2f pairs/s  ms_per_step %.0f  us/hyp-iter %.4f  pruned %.2f (%s)" % (d["value"], d["ms_per_step"], d["roofline"]["us_per_hypothesis_iteration"], d["pruned"]["value"], d["pruned"]["bit_identical_to_brute_force"]))
PY
done 2>&1 | grep -v amdgpu
