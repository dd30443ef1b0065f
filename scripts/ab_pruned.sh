#!/bin/bash
# A/B of pruned-search variants built as houv_amd/lib/libhouv_hip_<v>.so (+ libhouv_hip_stamps_<v>.so for selectivity)
for round in 1 2; do
  for v in "$@"; do
    echo "== variant $v (round $round)"
    HOUV_HIP_LIB=$GRAFT_REPO_ROOT/houv_amd/lib/libhouv_hip_$v.so timeout -k 5 200 python scripts/perf_pruned.py 2>/dev/null | grep "iters=200.*pruned=True"
  done
done
for v in "$@"; do
  if [ -f $GRAFT_REPO_ROOT/houv_amd/lib/libhouv_hip_stamps_$v.so ]; then
    echo "== selectivity $v"
    HOUV_STAMPS_LIB=$GRAFT_REPO_ROOT/houv_amd/lib/libhouv_hip_stamps_$v.so timeout -k 5 200 python scripts/prune_stats.py 2>/dev/null | grep views
  fi
done
