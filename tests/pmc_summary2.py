"""Per kernel name: SUM of counters over all launches (for kernels launched many times).  usage: pmc_summary2.py DIR"""
import csv, glob, os, sys, collections
root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, cs in acc.items():
        print(os.path.relpath(f, root).split(os.sep)[0], k[:50], "launches", len(n[k]))
        for c, v in sorted(cs.items()): print("    %-36s %.6g" % (c, v))
