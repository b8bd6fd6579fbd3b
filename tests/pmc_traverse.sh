#!/bin/bash
# PMC passes over the soup traversal workload (one counter group per rocprofv3 run), then a kernel trace of bench.py.
# Run on the GPU box from the repo root; writes under gpurun_out/pmc.
set -e
R=$PWD
OUT=$R/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tests/prof_traverse.py 10000000 16777216 2 0 > $OUT/$name.log 2>&1
}
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM
if [ -z "$SKIP_BENCH" ]; then timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $R/bench.py --steps 2 --warmup 1 > $OUT/bench_trace.log 2>&1; fi
cd $R && python3 tests/pmc_traverse_json.py $OUT > $OUT/traverse_traffic.json && python3 tests/pmc_summary.py $OUT k_trace_pairs > $OUT/traverse_pmc_summary.txt
echo done
