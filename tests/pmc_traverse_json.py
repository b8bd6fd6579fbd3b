"""HBM traffic per launch of the soup traversal kernel from the PMC passes of tests/pmc_traverse.sh, in the form bench.py quotes as
roofline.traffic.  usage: pmc_traverse_json.py gpurun_out/pmc > profiles/rNN_traverse_traffic.json
Reads = sum over request sizes of TCC_EA0_RDREQ_<size> x size (FETCH_SIZE tallies the 128-byte requests of gfx950 at 64 B, so it is
reported but not used); writes = WRITE_SIZE (KiB)."""
import collections, csv, glob, hashlib, json, os, sys
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("k_trace_pairs<false, false"):      # closest hit, not the counting variant
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); launches[k][r["Counter_Name"]].add(r["Dispatch_Id"])
# the timed launch is the kernel that moves the most data (the order-free kernel walks its near-ties again in the reference's order itself;
# should an ordered launch of the same batch appear next to it, as earlier builds had for those rays, its means are added)
def per_launch(k):
    return {n: v / max(1, len(launches[k][n])) for n, v in acc[k].items()}
main = max(acc, key=lambda k: acc[k].get("TCC_EA0_RDREQ_sum", acc[k].get("TCC_EA0_RDREQ", 0.0)) / max(1, len(launches[k].get("TCC_EA0_RDREQ_sum", launches[k].get("TCC_EA0_RDREQ", [1])))))
per = per_launch(main)
names = {"k_trace_pairs<false, false, true, true, true>": "closest hit, order-free over eight-wide quantised records (near-ties walked again in the reference's order inside the launch)",
         "k_trace_pairs<false, false, true, true, false>": "closest hit, order-free over quad records (near-ties walked again in the reference's order inside the launch)",
         "k_trace_pairs<false, false, true, false, false>": "closest hit, quad records, reference order"}
replay = [k for k in acc if k != main and "true, true" in main and k == "k_trace_pairs<false, false, true, false, false>"]
for k in replay:
    for n, v in per_launch(k).items():
        per[n] = per.get(n, 0.0) + v
g = lambda k: per.get(k, per.get(k + "_sum", 0.0))
rd128, rd64, rd32, rd_all = g("TCC_EA0_RDREQ_128B"), g("TCC_EA0_RDREQ_64B"), g("TCC_EA0_RDREQ_32B"), g("TCC_EA0_RDREQ")
other = max(0.0, rd_all - rd128 - rd64 - rd32)
reads = 128.0 * rd128 + 64.0 * (rd64 + other) + 32.0 * rd32
writes = 1024.0 * g("WRITE_SIZE")
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracing_amd", "lib", "librt_amd.so")
print(json.dumps({"kernel": "%s (%s)" % (main, names.get(main, "")), "workload": {"n_tris": 10000000, "n_rays": 16777216},
                  "lib_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None,
                  "per_launch": {"TCC_EA0_RDREQ_128B": rd128, "TCC_EA0_RDREQ_64B": rd64, "TCC_EA0_RDREQ_32B": rd32, "TCC_EA0_RDREQ": rd_all,
                                 "WRITE_SIZE_KiB": g("WRITE_SIZE"), "FETCH_SIZE_KiB": g("FETCH_SIZE"), "TCC_REQ": g("TCC_REQ"), "TCC_HIT": g("TCC_HIT"),
                                 "TCC_MISS": g("TCC_MISS")},
                  "l2_hit_rate": (g("TCC_HIT") / g("TCC_REQ")) if g("TCC_REQ") else None,
                  "traffic_bytes_per_launch": reads + writes,
                  "method": "reads = sum over request sizes of TCC_EA0_RDREQ_<size> x size (requests of unlisted size counted at 64 B); writes = WRITE_SIZE; "
                            "separate rocprofv3 --pmc passes over tests/prof_traverse.py (tests/pmc_traverse.sh), mean over the launches of the timed kernel"
                            " (the ordered second walk of near-ties runs inside it)"}, indent=1))
