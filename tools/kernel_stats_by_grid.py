"""Per-(kernel, grid size) statistics of a rocprofv3 --kernel-trace CSV: calls, mean / median / min / max duration (us), stddev.
rocprofv3's own --stats folds every launch of a kernel into one row; bench.py launches the headline kernel on three workloads
(4K x 32, 8K x 8, 512 x 1080p), so the per-grid table is what reproduces the bench line.
   python tools/kernel_stats_by_grid.py <dir with *kernel_trace.csv> > profiles/rNN_kernel_stats_by_grid.csv"""
import collections, csv, glob, os, statistics, sys
root = sys.argv[1]
rows = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        grid = r.get("Grid_Size") or r.get("Grid_Size_X")
        rows[(r["Kernel_Name"], int(grid), int(r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or 0), r.get("VGPR_Count", ""), r.get("SGPR_Count", ""),
              r.get("LDS_Block_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
w = csv.writer(sys.stdout)
w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Calls", "Mean_us", "Median_us", "Min_us", "Max_us", "Stddev_us", "Total_us"])
for key, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    w.writerow(list(key) + [len(v), "%.2f" % statistics.mean(v), "%.2f" % statistics.median(v), "%.2f" % min(v), "%.2f" % max(v),
                            "%.2f" % (statistics.pstdev(v) if len(v) > 1 else 0.0), "%.1f" % sum(v)])
