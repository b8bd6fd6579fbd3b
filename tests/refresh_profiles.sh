#!/bin/bash
# Regenerates the judged profile summaries (copied to profiles/ afterwards).  Run on the GPU box from the repo root.
set -e
R=$PWD
OUT=$R/gpurun_out/refresh
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. bench.py as the driver runs it, under a kernel trace
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $R/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/bench_trace.err
echo bench-trace-done
# 2. plain bench line
cd $R && timeout -k 10 400 python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
echo bench-done
# 3. one frame of Cornell and of the helmet under a kernel trace
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/frame_cornell -- python3 $R/tests/prof_frame.py cornell 800 32 32 > $OUT/frame_cornell.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/frame_helmet -- python3 $R/tests/prof_frame.py helmet 800 16 16 > $OUT/frame_helmet.log 2>&1
echo frames-done
# 4. SQ counters of a Cornell frame slice (two passes of counters)
pass() { local name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 $R/tests/prof_frame.py cornell 800 16 16 > $OUT/pmc_$name.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
echo pmc-done
cd $R
python3 tests/pmc_big_dispatch.py $OUT > $OUT/frame_pmc_largest_dispatch.txt
for d in bench_trace frame_cornell frame_helmet; do cp $(ls -t $OUT/$d/*/*kernel_stats.csv | head -1) $OUT/${d}_kernel_stats.csv; done
python3 - <<PY
import csv, glob, os
f = max(glob.glob("$OUT/bench_trace/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
with open("$OUT/bench_trace_kernel_launches.csv", "w") as o:
    o.write("kernel,duration_ms\n")
    for r in rows:
        if "k_trace_pairs" in r["Kernel_Name"]:
            o.write('"%s",%.3f\n' % (r["Kernel_Name"].split("(")[0].replace("void ", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
rm -rf $OUT/bench_trace $OUT/frame_cornell $OUT/frame_helmet $OUT/pmc_sq1 $OUT/pmc_sq2
echo all-done
