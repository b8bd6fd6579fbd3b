"""For each kernel name: counters of its LARGEST dispatch (by SQ_BUSY_CYCLES or first counter).  usage: pmc_big_dispatch.py DIR"""
import csv, glob, os, sys, collections
root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"].split("(")[0]][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    for k, ds in per.items():
        key = "SQ_BUSY_CYCLES" if any("SQ_BUSY_CYCLES" in c for c in ds.values()) else "SQ_ACTIVE_INST_ANY"
        best = max(ds.items(), key=lambda kv: kv[1].get(key, 0))
        c = best[1]
        line = "%s %s dispatch %s:" % (os.path.relpath(f, root).split(os.sep)[0], k[:40], best[0])
        print(line)
        for n, v in sorted(c.items()): print("    %-30s %.5g" % (n, v))
        if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
            print("    lane utilisation %.3f   VALU busy %.3f (of kernel time, 1024 SIMDs, 32 SEs)" %
                  (c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]), c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (c["SQ_BUSY_CYCLES"] / 32)))
