#!/bin/bash
# Calibrate FETCH_SIZE on random 64-B and 128-B record reads of known size (tests/microbench/random64). Run on the GPU box.
set -e
R=$PWD
OUT=$R/gpurun_out/pmc_cal
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $R/tests/microbench/random64 40000000 256 > $OUT/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- $R/tests/microbench/random64 40000000 256 > $OUT/tcc.log 2>&1
echo done
