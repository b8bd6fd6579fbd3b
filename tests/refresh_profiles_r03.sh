#!/bin/bash
# Regenerates the r03_* profile summaries into gpurun_out/r03 (copy them to profiles/ afterwards).  Run on the GPU box from the repo root,
# with the librt_amd.so that is to be judged: bench.py quotes the counter summaries only for the build whose hash they carry.
set -e
R=$PWD
OUT=$R/gpurun_out/r03
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. bench.py as the driver runs it, under a kernel trace, and plain
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $R/bench.py > $OUT/r03_bench_under_rocprof.json 2> $OUT/bench_trace.err
echo bench-trace-done
# 2. one frame of each configuration under a kernel trace
frame() { local scene=$1 w=$2 h=$3 sx=$4; PROF_HEIGHT=$h timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/frame_$scene -- python3 $R/tests/prof_frame.py $scene $w $sx $sx > $OUT/frame_$scene.log 2>&1; }
frame cornell 800 800 32
frame helmet 1440 1440 16
frame spheres2 1440 810 16
frame mrspheres 1440 1440 16
echo frames-done
# 3. SQ counters of a slice of each frame at its own resolution: one pass each (what bench.py's roofline_frame is derived from)
pass() { local name=$1 scene=$2 w=$3 h=$4 sx=$5; shift 5; PROF_HEIGHT=$h timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_${name}_$scene -- python3 $R/tests/prof_frame.py $scene $w $sx $sx > $OUT/pmc_${name}_$scene.log 2>&1; }
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY"
pass sq1 cornell 800 800 16 $SQ1
pass sq1 helmet 1440 1440 8 $SQ1
pass sq1 spheres2 1440 810 8 $SQ1
# 4. the helmet's walk kernels: memory side (vector-memory issue and busy cycles, texture addresser, L1, L2), one pass per counter block
pass sq2 helmet 1440 1440 8 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
pass ta helmet 1440 1440 8 TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcc helmet 1440 1440 8 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
echo pmc-done
cd $R
timeout -k 10 500 python3 bench.py > $OUT/r03_bench_line.json 2> $OUT/bench_line.err
echo bench-done
python3 tests/pmc_frame_json.py $OUT/pmc_sq1_cornell cornell 800 800 16 16 > $OUT/r03_frame_pmc_cornell.json
python3 tests/pmc_frame_json.py $OUT/pmc_sq1_helmet helmet 1440 1440 8 8 > $OUT/r03_frame_pmc_helmet.json
python3 tests/pmc_frame_json.py $OUT/pmc_sq1_spheres2 spheres2 1440 810 8 8 > $OUT/r03_frame_pmc_spheres2.json
mkdir -p $OUT/helmet_walk && cp -r $OUT/pmc_sq1_helmet $OUT/pmc_sq2_helmet $OUT/pmc_ta_helmet $OUT/pmc_tcc_helmet $OUT/helmet_walk/
python3 tests/pmc_kernels.py $OUT/helmet_walk _pl > $OUT/r03_helmet_walk_counters.txt
for d in bench_trace frame_cornell frame_helmet frame_spheres2 frame_mrspheres; do cp $(ls -t $OUT/$d/*/*kernel_stats.csv | head -1) $OUT/r03_${d}_kernel_stats.csv; done
python3 - <<PY
import csv, glob, os
f = max(glob.glob("$OUT/bench_trace/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
with open("$OUT/r03_bench_trace_kernel_launches.csv", "w") as o:
    o.write("kernel,duration_ms\n")
    for r in rows:
        if "k_trace_pairs" in r["Kernel_Name"]:
            o.write('"%s",%.3f\n' % (r["Kernel_Name"].split("(")[0].replace("void ", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
rm -rf $OUT/bench_trace $OUT/frame_cornell $OUT/frame_helmet $OUT/frame_spheres2 $OUT/frame_mrspheres $OUT/pmc_sq1_* $OUT/pmc_sq2_* $OUT/pmc_ta_* $OUT/pmc_tcc_* $OUT/helmet_walk
echo all-done
