import csv, glob, sys
root = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{root}/calib_{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if r["Counter_Name"] == c and any(x in k for x in ("read_", "write_")):
                v = float(r["Counter_Value"])
                print(f"{c:10s} {k:12s} counter={v:.5g} (KB) -> {v * 1024 / 2**30:.4f} x the 1 GiB the kernel moves in its own direction")
