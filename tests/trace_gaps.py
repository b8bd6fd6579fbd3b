"""Launch timeline of the last render in a rocprofv3 kernel trace: per stream (queue) the span, the busy time and every launch.
usage: trace_gaps.py DIR   (not collected by pytest)"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [r for r in rows if "k_" in r["Kernel_Name"]]
last_acc = [i for i, r in enumerate(ks) if r["Kernel_Name"].startswith("k_raygen")]
n_lanes = len({r["Queue_Id"] for r in ks[last_acc[-1]:]})
start = last_acc[-n_lanes] if len(last_acc) >= n_lanes else last_acc[-1]
seg = ks[start:]
t0 = min(int(r["Start_Timestamp"]) for r in seg); t1 = max(int(r["End_Timestamp"]) for r in seg)
print("render: %.3f ms from first launch to last end, %d launches on %d queues" % ((t1 - t0) / 1e6, len(seg), n_lanes))
for q in sorted({r["Queue_Id"] for r in seg}):
    qs = [r for r in seg if r["Queue_Id"] == q]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in qs)
    print("queue %s: starts at %.3f ms, ends at %.3f ms, busy %.3f ms in %d launches" %
          (q, (int(qs[0]["Start_Timestamp"]) - t0) / 1e6, (int(qs[-1]["End_Timestamp"]) - t0) / 1e6, busy / 1e6, len(qs)))
    for r in qs:
        print("   %8.3f  %-30s %9.1f us" % ((int(r["Start_Timestamp"]) - t0) / 1e6, r["Kernel_Name"].split("(")[0].replace("void ", "")[:30],
                                            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
