"""Print the top rows of a rocprofv3 --stats kernel summary (helper for gpurun one-liners): python tests/gpu_prof_top.py <dir> [n]"""
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not fs:
    print("no kernel_stats.csv under", sys.argv[1]); sys.exit(0)
rows = list(csv.DictReader(open(fs[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%-64s %6s %12s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("%-64s %6s %12.1f %9.2f %6.2f" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
