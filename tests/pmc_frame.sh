#!/bin/bash
# PMC passes over one Cornell frame slice (800x800, 52 spp = 2 passes). Run on the GPU box from the repo root.
set -e
R=$PWD
OUT=$R/gpurun_out/pmc_frame
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tests/prof_frame.py ${SCENE:-cornell} 800 ${SX:-16} ${SX:-16} > $OUT/$name.log 2>&1
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
echo done
