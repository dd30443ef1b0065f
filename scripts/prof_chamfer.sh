#!/bin/bash
# PMC passes on the stand-alone Chamfer op (filtered vs direct kernel), uniform 4096 x 2048^2.  Run via gpurun.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_chamfer; rm -rf $OUT; mkdir -p $OUT; cd $R
export ONLY=uniform4096_2048
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 scripts/ab_chamfer.py > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
for mode in filter direct; do
  if [ $mode = direct ]; then export HOUV_CHAMFER_DIRECT=1; else unset HOUV_CHAMFER_DIRECT; fi
  run ${mode}_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
  run ${mode}_sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA
  run ${mode}_grbm GRBM_GUI_ACTIVE GRBM_COUNT
done
python3 - <<'PY'
import csv, glob, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_chamfer"
for d in sorted(glob.glob(root + "/*_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        rows = [r for r in csv.DictReader(open(f)) if "chamfer_nn" in r["Kernel_Name"]]
        last = {}
        for r in rows: last[r["Counter_Name"]] = (float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["VGPR_Count"], r["LDS_Block_Size"] if "LDS_Block_Size" in r else "")
        for c, v in last.items(): print(f"{os.path.basename(d):14s} {c:22s} {v[0]:.4g} dur_us={v[1]/1e3:.1f} vgpr={v[2]} lds={v[3]}")
PY
