"""Sum every counter of a rocprofv3 --pmc run per kernel.  usage: pmc_raw.py DIR"""
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_"):
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        print("   %-28s %.4g" % (n, v))
