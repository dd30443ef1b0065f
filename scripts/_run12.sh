cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_dcp_r2; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --dcp --pairs 64 --steps 3 --warmup 1 > $OUT/bench_dcp.json 2> $OUT/err.txt
cat $OUT/bench_dcp.json | head -c 300; echo
f=$(find $OUT -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-200
