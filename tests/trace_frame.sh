#!/bin/bash
# Kernel trace of one full Cornell frame (800x800, 1024 spp).  Run on the GPU box from the repo root.
set -e
R=$PWD
OUT=$R/gpurun_out/frame_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/tests/prof_frame.py ${1:-cornell} 800 ${2:-32} ${2:-32} > $OUT/run.log 2>&1
echo done
