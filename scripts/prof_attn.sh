#!/bin/bash
# PMC pass of the attention probe (scripts/attn_split_probe.py): where the split attention kernel's wave cycles go.  Run via gpurun.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_attn; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/p1 -- python3 scripts/attn_split_probe.py > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS --output-format csv -d $OUT/p2 -- python3 scripts/attn_split_probe.py > $OUT/p2.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
for p in ("p1", "p2"):
    f = glob.glob(f"{R}/gpurun_out/prof_attn/{p}/*/*counter_collection.csv")
    if not f: print(p, "no counters"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "attention" not in k: continue
        print(k)
        for c, v in sorted(d.items()): print(f"    {c:28s} max over launches {max(v):.4g}  (n={len(v)})")
PY
