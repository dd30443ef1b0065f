#!/bin/bash
# PMC passes (separate runs, kernel-trace only) for the fused kernel on a small workload.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_pmc
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 scripts/perf_probe.py > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run fetch FETCH_SIZE
run write WRITE_SIZE
ls $OUT
