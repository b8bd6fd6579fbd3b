#!/bin/bash
# PMC passes over the soup traversal workload (one counter group per rocprofv3 run). Run on the GPU box from the repo root.
set -e
R=$PWD
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tests/prof_traverse.py 10000000 16777216 2 > $OUT/$name.log 2>&1
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass tcc2 TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TCC_BUBBLE_sum
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass ta1 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
echo done
