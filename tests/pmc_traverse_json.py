"""HBM traffic per launch of the soup traversal kernel from the PMC passes of tests/pmc_traverse.sh, in the form bench.py quotes as
roofline.traffic.  usage: pmc_traverse_json.py gpurun_out/pmc > profiles/rNN_traverse_traffic.json
Reads = sum over request sizes of TCC_EA0_RDREQ_<size> x size (FETCH_SIZE tallies the 128-byte requests of gfx950 at 64 B, so it is
reported but not used); writes = WRITE_SIZE (KiB)."""
import collections, csv, glob, json, os, sys
root = sys.argv[1]
acc = collections.defaultdict(float); launches = collections.defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace_pairs<false, false, true>" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); launches[r["Counter_Name"]].add(r["Dispatch_Id"])
per = {k: v / max(1, len(launches[k])) for k, v in acc.items()}
g = lambda k: per.get(k, per.get(k + "_sum", 0.0))
rd128, rd64, rd32, rd_all = g("TCC_EA0_RDREQ_128B"), g("TCC_EA0_RDREQ_64B"), g("TCC_EA0_RDREQ_32B"), g("TCC_EA0_RDREQ")
other = max(0.0, rd_all - rd128 - rd64 - rd32)
reads = 128.0 * rd128 + 64.0 * (rd64 + other) + 32.0 * rd32
writes = 1024.0 * g("WRITE_SIZE")
print(json.dumps({"kernel": "k_trace_pairs<false,false,true> (closest hit, quad records)", "workload": {"n_tris": 10000000, "n_rays": 16777216},
                  "per_launch": {"TCC_EA0_RDREQ_128B": rd128, "TCC_EA0_RDREQ_64B": rd64, "TCC_EA0_RDREQ_32B": rd32, "TCC_EA0_RDREQ": rd_all,
                                 "WRITE_SIZE_KiB": g("WRITE_SIZE"), "FETCH_SIZE_KiB": g("FETCH_SIZE"), "TCC_REQ": g("TCC_REQ"), "TCC_HIT": g("TCC_HIT"),
                                 "TCC_MISS": g("TCC_MISS")},
                  "l2_hit_rate": (g("TCC_HIT") / g("TCC_REQ")) if g("TCC_REQ") else None,
                  "traffic_bytes_per_launch": reads + writes,
                  "method": "reads = sum over request sizes of TCC_EA0_RDREQ_<size> x size (requests of unlisted size counted at 64 B); writes = WRITE_SIZE; "
                            "separate rocprofv3 --pmc passes over tests/prof_traverse.py (tests/pmc_traverse.sh), mean over the launches of the kernel"}, indent=1))
