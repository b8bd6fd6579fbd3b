"""Per-launch listing of the first render pass in a rocprofv3 kernel trace.  usage: trace_pass_summary.py DIR [rows]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_raygen")]
seg = rows[idx[0]:(idx[1] if len(idx) > 1 else len(rows))]
t0 = int(seg[0]["Start_Timestamp"])
tot = collections.defaultdict(float)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for k, r in enumerate(seg):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[name] += d
    if k < n: print("%-22s %10.1f us  at %10.1f" % (name[:22], d, (int(r["Start_Timestamp"]) - t0) / 1e3))
print("pass wall ms", (int(seg[-1]["End_Timestamp"]) - t0) / 1e6, "launches", len(seg))
print({k: round(v / 1e3, 2) for k, v in tot.items()})
