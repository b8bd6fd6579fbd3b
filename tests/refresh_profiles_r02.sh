#!/bin/bash
# Regenerates the r02_* profile summaries (copy them from gpurun_out/r02 to profiles/ afterwards).  Run on the GPU box from the repo root.
set -e
R=$PWD
OUT=$R/gpurun_out/r02
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. bench.py as the driver runs it, under a kernel trace, and plain
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $R/bench.py > $OUT/r02_bench_under_rocprof.json 2> $OUT/bench_trace.err
echo bench-trace-done
cd $R && timeout -k 10 500 python3 bench.py > $OUT/r02_bench_line.json 2> $OUT/bench_line.err
echo bench-done
cd /tmp
# 2. one frame of each configuration under a kernel trace
for cfg in "cornell 800 32 32" "helmet 1440 16 16" "spheres2 1440 16 16"; do
  set -- $cfg
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/frame_$1 -- python3 $R/tests/prof_frame.py $cfg > $OUT/frame_$1.log 2>&1
done
echo frames-done
# 3. SQ counters of a frame slice (Cornell 800x800 @ 256 spp, helmet 800x800 @ 64 spp): one pass each
pass() { local name=$1 scene=$2 w=$3 sx=$4; shift 4; timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_${name}_$scene -- python3 $R/tests/prof_frame.py $scene $w $sx $sx > $OUT/pmc_${name}_$scene.log 2>&1; }
pass sq1 cornell 800 16 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY
pass sq1 helmet 800 8 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY
pass sq2 cornell 800 16 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
echo pmc-done
cd $R
python3 tests/pmc_frame_json.py $OUT/pmc_sq1_cornell cornell 800 800 16 16 > $OUT/r02_frame_pmc_cornell.json
python3 tests/pmc_frame_json.py $OUT/pmc_sq1_helmet helmet 800 800 8 8 > $OUT/r02_frame_pmc_helmet.json
python3 tests/pmc_big_dispatch.py $OUT > $OUT/r02_frame_pmc_largest_dispatch.txt
for d in bench_trace frame_cornell frame_helmet frame_spheres2; do cp $(ls -t $OUT/$d/*/*kernel_stats.csv | head -1) $OUT/r02_${d}_kernel_stats.csv; done
python3 - <<PY
import csv, glob, os
f = max(glob.glob("$OUT/bench_trace/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
with open("$OUT/r02_bench_trace_kernel_launches.csv", "w") as o:
    o.write("kernel,duration_ms\n")
    for r in rows:
        if "k_trace_pairs" in r["Kernel_Name"]:
            o.write('"%s",%.3f\n' % (r["Kernel_Name"].split("(")[0].replace("void ", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
rm -rf $OUT/bench_trace $OUT/frame_cornell $OUT/frame_helmet $OUT/frame_spheres2 $OUT/pmc_sq1_cornell $OUT/pmc_sq1_helmet $OUT/pmc_sq2_cornell
echo all-done
