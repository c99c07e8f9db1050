import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/*/*_counter_collection.csv") + glob.glob(f"{d}/*_counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "mg_tile" in r["Kernel_Name"] or "mg_step" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
