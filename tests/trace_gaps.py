import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [r for r in rows if "k_" in r["Kernel_Name"]]
idx = [i for i, r in enumerate(ks) if r["Kernel_Name"].startswith("k_raygen")]
seg = ks[idx[-1]:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
print("span %.3f ms, busy %.3f ms, gaps %.3f ms over %d launches" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(seg)))
for r in seg:
    print("%-28s %9.1f us" % (r["Kernel_Name"].split("(")[0].replace("void ", "")[:28], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
