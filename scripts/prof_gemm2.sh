#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_gemm2; rm -rf $OUT; mkdir -p $OUT; cd $R
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 scripts/gemm_probe2.py > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 - <<'PY'
import csv, glob, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_gemm2"
for d in sorted(glob.glob(root + "/*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        last = {}
        for r in csv.DictReader(open(f)):
            if "gemm_f32" in r["Kernel_Name"]:
                last[r["Counter_Name"]] = (float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["VGPR_Count"])
        for c, v in last.items(): print(f"{os.path.basename(d):6s} {c:26s} {v[0]:.4g} dur_us={v[1]/1e3:.1f} vgpr={v[2]}")
PY
cat $OUT/sq1.log | tail -1
