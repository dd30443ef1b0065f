#!/bin/bash
# Round-2 profiles of the fused loop at the bench's launch shape (16384 hypotheses x 50 iterations per launch):
# kernel-trace stats of the bench command, PMC passes (each in its own run), FETCH/WRITE calibration.  Run via gpurun.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r2
rm -rf $OUT && mkdir -p $OUT
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench/pmc_calib.hip -o $OUT/pmc_calib
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/calib_$c -- $OUT/pmc_calib > $OUT/calib_$c.log 2>&1 || echo "calib $c failed"
done
run() { name=$1; shift
  P=256 ITERS=50 SOLVE_ONLY=1 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 scripts/perf_probe.py > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run fetch FETCH_SIZE
run write WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || echo "trace failed"
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1 || true
python3 scripts/pmc_calib_summary.py $OUT > $OUT/pmc_calib_summary.txt 2>&1 || true
find $OUT -name "*kernel_stats.csv" | head -3
cat $OUT/pmc_calib_summary.txt; cat $OUT/pmc_summary.txt | grep -E "FETCH|WRITE|GRBM_GUI|INSTS_VALU " 
