#!/bin/bash
# Counters of the batch trace kernels (ordered and order-free) on the soup.  Run on the GPU box from the repo root.
set -e
R=$PWD; OUT=$R/gpurun_out/pmc_free; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { local name=$1; shift; timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tests/free_walk_probe.py > $OUT/$name.log 2>&1; }
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY
cd $R && python3 tests/pmc_kernels.py $OUT k_trace_pairs > $OUT/summary.txt
rm -rf $OUT/rdreq $OUT/tcc $OUT/sq1
echo done
