"""Summarise rocprofv3 --pmc csv output: per kernel name, mean counter value per launch.  usage: pmc_summary.py DIR [substr]"""
import csv, glob, os, sys, collections
root = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub and sub not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); launches[k].add(r["Dispatch_Id"])
    for k, cs in acc.items():
        n = len(launches[k])
        print(os.path.relpath(f, root).split(os.sep)[0], k[:60], "launches", n)
        for c, v in sorted(cs.items()): print("    %-44s %.6g per launch" % (c, v / n))
