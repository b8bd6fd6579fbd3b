"""Per-kernel sums of every counter found under a directory of rocprofv3 --pmc outputs (mean per launch).  usage: pmc_kernels.py DIR [name filter]"""
import collections, csv, glob, os, sys
root = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else "k_"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if flt not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); launches[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(acc):
    print(k)
    c = {n: v / max(1, len(launches[k][n])) for n, v in acc[k].items()}
    for n in sorted(c):
        print("   %-28s %.4g  (%d launches)" % (n, c[n], len(launches[k][n])))
    a, b, t = c.get("SQ_ACTIVE_INST_VALU"), c.get("SQ_BUSY_CYCLES"), c.get("SQ_THREAD_CYCLES_VALU")
    if a and b:
        print("   valu_busy %.3f  lane_utilisation %.3f" % ((4 * a / 1024) / (b / 32), t / (64 * a) if t else 0))
