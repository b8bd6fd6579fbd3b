#!/bin/bash
# Kernel statistics and one SQ counter pass of a frame slice; usage: quick_prof.sh TAG SCENE WIDTH SX  (run on the GPU box from the repo root)
set -e
R=$PWD; TAG=$1; SCENE=$2; W=$3; SX=$4
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tests/prof_frame.py $SCENE $W $SX $SX > $OUT/trace.log 2>&1
cp $(ls -t $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d $OUT/pmc -- python3 $R/tests/prof_frame.py $SCENE $W $SX $SX > $OUT/pmc.log 2>&1
cd $R
python3 tests/pmc_frame_json.py $OUT/pmc $SCENE $W $W $SX $SX > $OUT/${TAG}_frame_pmc_$SCENE.json
rm -rf $OUT/trace $OUT/pmc
echo quick-prof-done
