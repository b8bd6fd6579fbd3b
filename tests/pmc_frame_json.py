"""Turn the SQ counter pass of one frame (rocprofv3 --pmc ... -- prof_frame.py SCENE WIDTH SX SY) into the summary bench.py quotes as
`roofline_frame`: per kernel and for the whole frame, VALU busy, lane utilisation and their product.
usage: pmc_frame_json.py PMC_DIR SCENE WIDTH HEIGHT SX SY > profiles/rNN_frame_pmc_SCENE.json

Counters (summed over every dispatch of a kernel): SQ_ACTIVE_INST_VALU = cycles, in units of four, in which a SIMD executes a VALU
instruction, summed over the chip's 1024 SIMDs; SQ_BUSY_CYCLES = cycles a shader engine has waves, summed over 32 engines;
SQ_THREAD_CYCLES_VALU = the same VALU cycles times the number of active lanes.
  valu_busy        = 4 * ACTIVE_INST_VALU / 1024  /  (BUSY_CYCLES / 32)      share of the kernel's time a SIMD issues VALU work
  lane_utilisation = THREAD_CYCLES_VALU / (64 * ACTIVE_INST_VALU)            active lanes per issued VALU instruction
  useful           = valu_busy * lane_utilisation                            share of the chip's VALU lane-slots doing useful work"""
import collections, csv, glob, hashlib, json, os, sys
root, scene, w, h, sx, sy = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.defaultdict(set)
spans = {}      # dispatch -> (start, end) in ns, when the counter file carries timestamps
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("k_"):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); launches[k].add(r["Dispatch_Id"])
        try:
            spans[r["Dispatch_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
        except (KeyError, ValueError):
            pass
# the library these counters belong to (bench.py quotes them only for the same build)
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracing_amd", "lib", "librt_amd.so")
lib_sha16 = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None
serial_seconds = round(sum(e - b for b, e in spans.values()) * 1e-9, 6) if spans else None
def derive(c):
    a, b, t = c.get("SQ_ACTIVE_INST_VALU", 0.0), c.get("SQ_BUSY_CYCLES", 0.0), c.get("SQ_THREAD_CYCLES_VALU", 0.0)
    busy = (4 * a / 1024) / (b / 32) if b else 0.0
    util = t / (64 * a) if a else 0.0
    return {"valu_busy": round(busy, 4), "lane_utilisation": round(util, 4), "useful_lane_fraction": round(busy * util, 4)}
tot = collections.defaultdict(float)
kernels = {}
busy_all = sum(c.get("SQ_BUSY_CYCLES", 0.0) for c in acc.values()) or 1.0
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0.0)):
    for n, v in c.items():
        tot[n] += v
    d = derive(c); d["launches"] = len(launches[k]); d["share_of_busy_cycles"] = round(c.get("SQ_BUSY_CYCLES", 0.0) / busy_all, 4)
    d["valu_instructions"] = c.get("SQ_INSTS_VALU", 0.0)
    kernels[k] = d
print(json.dumps({"scene": scene, "width": w, "height": h, "sample": "%d spp (%dx%d strata) of the frame, one render under rocprofv3 --pmc" % (sx * sy, sx, sy),
                  "sample_spp": sx * sy, "lib_sha16": lib_sha16, "serial_kernel_seconds": serial_seconds,
                  "counters": ["SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU"],
                  "frame": derive(tot), "kernels": kernels}, indent=1))
