#!/bin/bash
# Round-3 profiles of the fused loop at the bench's launch shape (16384 hypotheses x 50 iterations per launch), both searches:
# PMC passes (each counter group in its own run, --kernel-trace only) and the kernel-trace stats of the bench command.
# Run via gpurun; writes gpurun_out/prof_r3/{pmc_summary.txt,pmc_solve.json,...} -- copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r3
rm -rf $OUT && mkdir -p $OUT
cd $R
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 scripts/pmc_probe.py > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA
run sq3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INSTS_BRANCH
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 scripts/pmc_summary_r3.py $OUT > $OUT/pmc_summary.txt 2>&1 || true
cat $OUT/pmc_summary.txt | grep -E "INSTS_VALU |GRBM_GUI|FETCH|WRITE|BUSY_CYC|us per" | head -40
if [ "$1" = "trace" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || echo "trace failed"
  find $OUT/trace -name "*kernel_stats.csv" | head -3
fi
