#!/bin/bash
# Texture-addresser / L1 counters over one frame slice of a scene (default helmet, 800x800, 64 spp).  Run on the GPU box from the repo root.
set -e
R=$PWD
OUT=$R/gpurun_out/pmc_ta
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/ta -- python3 $R/tests/prof_frame.py ${SCENE:-helmet} 800 ${SX:-8} ${SX:-8} > $OUT/ta.log 2>&1
echo done
