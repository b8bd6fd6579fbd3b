"""Print the per-kernel summary of a rocprofv3 --kernel-trace --stats run (csv output)."""
import csv, glob, sys
d = sys.argv[1]
files = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
for f in files:
    rows = list(csv.DictReader(open(f)))
    print("%-70s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for r in rows[:14]:
        print("%-70s %8s %12.1f %10.2f %6.2f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
