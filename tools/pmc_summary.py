"""Summarise rocprofv3 --pmc CSV output: per (kernel, grid size), per counter: mean over the dispatches (sum over the
counter's instances per dispatch)."""
import collections, csv, glob, json, os, sys
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            key = (row["Kernel_Name"], int(row["Grid_Size"]))
            acc[key][row["Counter_Name"]][(f, int(row["Dispatch_Id"]))] += float(row["Counter_Value"])
out = {}
for (k, grid), cs in sorted(acc.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    if "rwh::" not in k:
        continue
    print("%s   grid %d" % (k[:100], grid))
    out["%s|%d" % (k, grid)] = {}
    for c, by in sorted(cs.items()):
        vals = [by[d] for d in sorted(by)]
        vals = vals[len(vals) // 2:] if len(vals) > 4 else vals      # second half: past the clock ramp
        m = sum(vals) / len(vals)
        out["%s|%d" % (k, grid)][c] = m
        print("   %-24s %18.0f   (mean of %d dispatches)" % (c, m, len(vals)))
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
