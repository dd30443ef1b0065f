#!/bin/bash
# rocprofv3 kernel trace + stats of a reduced bench (same kernels, fewer pairs).  Run on the GPU box via gpurun.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_trace
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
find $OUT -name "*stats*.csv" | head
