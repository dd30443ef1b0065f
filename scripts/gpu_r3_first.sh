#!/bin/bash
# round 3, first GPU call: full GPU suite with the pruned default, pruned-kernel diagnostics, a short bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3a/pytest.log
tail -n 5 gpurun_out/r3a/pytest.log
PRUNED=1 timeout -k 10 200 python scripts/stamps.py > gpurun_out/r3a/stamps_pruned.txt 2>&1; tail -n 30 gpurun_out/r3a/stamps_pruned.txt
timeout -k 10 300 python scripts/prune_stats.py > gpurun_out/r3a/prune_stats.txt 2>&1; tail -n 4 gpurun_out/r3a/prune_stats.txt
timeout -k 10 500 python bench.py --steps 3 --warmup 1 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/r3a/bench.json
