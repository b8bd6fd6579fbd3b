"""Where one pass spends its time by bounce (from a rocprofv3 kernel trace csv).  usage: trace_tail.py DIR"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [r for r in rows if "k_" in r["Kernel_Name"]]
idx = [i for i, r in enumerate(ks) if r["Kernel_Name"].startswith("k_raygen")]
seg = ks[idx[-1]:]
t0 = int(seg[0]["Start_Timestamp"]); tend = int(seg[-1]["End_Timestamp"])
print("pass %.2f ms, %d launches" % ((tend - t0) / 1e6, len(seg)))
b = -1
acc = {}
for r in seg:
    n = r["Kernel_Name"]
    if "k_extend" in n: b += 1
    acc.setdefault(b, [0.0, 0]); acc[b][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; acc[b][1] += 1
cum = 0
for k in sorted(acc):
    cum += acc[k][0]
    if k < 12 or k % 8 == 0: print("bounce %3d: %9.1f us in %d launches, cumulative %.2f ms" % (k, acc[k][0], acc[k][1], cum / 1e3))
