"""Summarise the PMC passes of scripts/prof_r3.sh: one line per (pass, kernel, counter) and pmc_solve.json, the per-kernel
numbers bench.py quotes (with the build id of the profiled library)."""
import collections, csv, glob, json, os, re, subprocess, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r3"
P, K, ITERS = int(os.environ.get("P", 256)), 64, int(os.environ.get("ITERS", 50))
wg_iters = P * K * ITERS
N_SIMD, N_XCD = 1024, 8
last = collections.defaultdict(dict)
for name in ("sq1", "sq2", "sq3", "grbm", "fetch", "write"):
    fs = glob.glob(f"{root}/{name}/*/*counter_collection.csv")
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "solve_kernel" not in k:
            continue
        kk = "houv::solve_kernel<%s>" % k.split("solve_kernel<")[1].split(">")[0]
        agg[kk][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                          r["VGPR_Count"], r.get("Scratch_Size", ""), r["Grid_Size"]))
    for kk, d in agg.items():
        for c, v in d.items():
            x = v[-1]                                   # the LAST launch: a pruned launch that started from a valid workspace
            last[kk][c] = x[0]
            last[kk]["dur_us_" + name] = x[1] / 1e3
            last[kk]["vgpr"], last[kk]["scratch"] = x[2], x[3]
            print(f"{name:6s} {kk:42s} {c:22s} last={x[0]:.4g} dur_us={x[1]/1e3:.1f} vgpr={x[2]} scratch={x[3]} grid={x[4]} n={len(v)}")
for f in sorted(glob.glob(f"{root}/*.log")):
    for line in open(f):
        if "us per hypothesis-iteration" in line:
            print(os.path.basename(f), line.strip())
build = None
for f in glob.glob(f"{root}/sq1.log"):
    m = re.search(r"build ([0-9a-f]{16})", open(f).read())
    build = m.group(1) if m else None
out = {"build_id": build, "points": int(os.environ.get("N", 2048)),
       "launch": f"{P} pairs x {K} hypotheses x {ITERS} iterations = {wg_iters} workgroup-iterations per launch (bench.py's base-stage "
                 "launch shape), scripts/prof_r3.sh / scripts/pmc_probe.py; counters of the last of 3 launches",
       "fetch_correction": 2.0, "write_correction": 1.0,
       "calibration": "profiles/r02_pmc_calib.txt: FETCH_SIZE reports 1/2 of the bytes on this gfx950, WRITE_SIZE is exact",
       "kernels": {}}
try:
    out["git_head"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(os.path.abspath(__file__))).decode().strip()
except Exception:
    out["git_head"] = os.environ.get("GIT_HEAD", "unknown")
for kk, d in last.items():
    if "SQ_INSTS_VALU" not in d or "GRBM_GUI_ACTIVE" not in d:
        continue
    e = {"valu_insts_per_wg_iter": d["SQ_INSTS_VALU"] / wg_iters,
         "simd_clk_per_wg_iter": d["GRBM_GUI_ACTIVE"] / N_XCD * N_SIMD / wg_iters,
         "sustained_clock_ghz": d["GRBM_GUI_ACTIVE"] / N_XCD / (d["dur_us_grbm"] * 1e3),
         "us_per_wg_iter": d["dur_us_grbm"] / wg_iters,
         "fetch_bytes_per_wg_iter": d.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0 / wg_iters,
         "write_bytes_per_wg_iter": d.get("WRITE_SIZE", 0.0) * 1024.0 / wg_iters,
         "lds_insts_per_wg_iter": d.get("SQ_INSTS_LDS", 0.0) / wg_iters,
         "salu_insts_per_wg_iter": d.get("SQ_INSTS_SALU", 0.0) / wg_iters,
         "lds_bank_conflict_share_of_lds_cycles": (d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]) if d.get("SQ_LDS_IDX_ACTIVE") else None,
         "wait_inst_lds_share_of_wave_cycles": (d.get("SQ_WAIT_INST_LDS", 0.0) / d["SQ_WAVE_CYCLES"]) if d.get("SQ_WAVE_CYCLES") else None,
         "wait_any_share_of_wave_cycles": (d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"]) if d.get("SQ_WAVE_CYCLES") else None,
         "vgpr": d.get("vgpr"), "scratch_bytes_per_lane": d.get("scratch")}
    out["kernels"][kk] = e
json.dump(out, open(os.path.join(root, "pmc_solve.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
