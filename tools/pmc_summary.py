"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per counter: sum / dispatches."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        print("==", f)
        for k, v in agg.items():
            print(f"  {k:60s} dispatches {len(n[k]):4d}  " + "  ".join(f"{c}/disp={val/len(n[k]):.4g}" for c, val in v.items()))
