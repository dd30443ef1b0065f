#!/bin/bash
# A/B of library builds houv_amd/lib/libhouv_hip_<v>.so on scripts/perf_pruned.py (brute and pruned, 200 iterations), twice
for round in 1 2; do
  for v in "$@"; do
    echo "== variant $v (round $round)"
    HOUV_HIP_LIB=$GRAFT_REPO_ROOT/houv_amd/lib/libhouv_hip_$v.so timeout -k 5 200 python scripts/perf_pruned.py 2>/dev/null | grep "iters=200"
  done
done
