"""Summarise rocprofv3 --pmc CSV output: per kernel, per counter: mean over the last dispatches."""
import collections, csv, glob, os, sys
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
for k, cs in acc.items():
    if "warp" not in k and "stitch" not in k:
        continue
    print(k[:90])
    for c, vals in sorted(cs.items()):
        by = collections.defaultdict(float)
        for d, v in vals:
            by[d] += v
        ds = sorted(by)[-3:]
        print("   %-28s %16.0f   (mean of last %d of %d dispatches)" % (c, sum(by[d] for d in ds) / len(ds), len(ds), len(by)))
