import csv,glob,collections,sys,os
root=sys.argv[1] if len(sys.argv)>1 else "gpurun_out/prof_pmc"
for name in ("sq1","sq2","grbm","fetch","write"):
    fs=glob.glob(f"{root}/{name}/*/*counter_collection.csv")
    if not fs: continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"]
        if "solve_kernel" in k: kk="solve<%s>"%k.split("solve_kernel<")[1].split(">")[0]
        elif "chamfer_nn" in k: kk="chamfer_nn"
        else: continue
        agg[kk][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"]), r["VGPR_Count"], r["Scratch_Size"],r["Grid_Size"]))
    for kk,d in agg.items():
        for c,v in d.items():
            last=v[-1]
            print(f"{name:6s} {kk:18s} {c:22s} last={last[0]:.4g} dur_us={last[1]/1e3:.1f} vgpr={last[2]} scratch={last[3]} grid={last[4]} n={len(v)}")
