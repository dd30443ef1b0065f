#!/bin/bash
# rocprofv3 per-kernel summary of `bench.py --dcp` (BASELINE configs[4]).  Run via gpurun.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_dcp3; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --dcp --pairs 64 --steps 3 --warmup 1 > $OUT/bench.json 2>$OUT/bench.err
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_dcp3/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), "%8.1f us avg" % (float(r["AverageNs"]) / 1e3), "%5.2f %%" % float(r["Percentage"]))
print("total ms", tot / 1e6)
PY
