#!/bin/bash
# A/B of build variants in one process each, interleaved twice (same box).
for round in 1 2; do
  for v in "$@"; do
    echo "== variant $v (round $round)"
    HOUV_HIP_LIB=$GRAFT_REPO_ROOT/houv_amd/lib/libhouv_hip_$v.so P=32 ITERS=20 timeout -k 5 120 python scripts/perf_probe.py 2>/dev/null | grep "^solve"
  done
done
